#!/bin/bash
# A/B timing of kernel variants inside ONE GPU call (boxes differ by several percent):
#   bash scripts/ab_variant.sh base TAG               kernel stats of a short cfg2 bench on the in-tree library
#   bash scripts/ab_variant.sh libpdekan_X.so TAG      the same on pde_amd/variants/libpdekan_X.so (PDEKAN_LIB)
# (pde_amd/variants is listed in .gpurunignore: take it out while variant libraries have to travel)
export TMPDIR=/tmp
if [ -n "$1" ] && [ "$1" != "base" ]; then export PDEKAN_LIB=$PWD/pde_amd/variants/$1; fi
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$2 -- python3 bench.py --steps 15 --warmup 3 --workload cfg2 --no-cpu-baseline --no-scaling-reference > gpurun_out/ab_$2.log 2>&1 || exit 1
python3 - "$2" <<PY
import glob,csv,sys
f=glob.glob("gpurun_out/ab_%s/**/*kernel_stats.csv" % sys.argv[1],recursive=True)[0]
print("==", sys.argv[1])
for r in list(csv.DictReader(open(f)))[:7]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
