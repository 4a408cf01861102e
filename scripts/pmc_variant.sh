#!/bin/bash
# SQ + GRBM counters of a short cfg2 bench for library variants (see ab_bench.sh): bash scripts/pmc_variant.sh KERNEL_SUBSTR base X ...
export TMPDIR=/tmp
K=$1; shift
B="python3 bench.py --steps 5 --warmup 2 --workload cfg2 --no-cpu-baseline --no-scaling-reference"
for v in "$@"; do
  if [ "$v" = "base" ]; then unset PDEKAN_LIB; else export PDEKAN_LIB=$PWD/pde_amd/variants/libpdekan_$v.so; fi
  OUT=gpurun_out/pmcv_$v; mkdir -p $OUT
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- $B > $OUT/grbm.log 2>&1 || exit 1
  echo "== $v"; python3 scripts/pmc_summary.py $OUT/sq $OUT/grbm | grep "$K" | grep -v "true>" | cut -c1-420
done
