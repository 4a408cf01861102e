#!/bin/bash
# A/B of library variants inside ONE GPU call on a named workload: per-kernel times of a short bench for each variant.
#   bash scripts/ab_bench_wl.sh cfg4 3 base head      (base = the in-tree library; X = pde_amd/variants/libpdekan_X.so)
WL=$1; STEPS=$2; shift; shift
for v in "$@"; do
  if [ "$v" = "base" ]; then unset PDEKAN_LIB; else export PDEKAN_LIB=$PWD/pde_amd/variants/libpdekan_$v.so; fi
  timeout -k 10 200 python3 bench.py --steps $STEPS --warmup 1 --workload $WL --no-cpu-baseline --no-scaling-reference > gpurun_out/ab_${WL}_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/ab_${WL}_$v.log; exit 1; }
  grep "^{" gpurun_out/ab_${WL}_$v.log > gpurun_out/ab_${WL}_$v.json
  echo "== $WL $v"; python3 scripts/show_bench.py gpurun_out/ab_${WL}_$v.json | head -8
done
