#!/bin/bash
# rocprofv3 counter passes (one block of counters per run) for bench.py's default workload, summary to stdout.
# usage (on the GPU box, from the repo root):  bash scripts/pmc_r2.sh TAG [bench args]
TAG=${1:-x}; shift
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
B="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-reference $*"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- python3 $B > $OUT/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/valu -- python3 $B > $OUT/valu.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- python3 $B > $OUT/grbm.log 2>&1 || exit 1
python3 scripts/pmc_summary.py $OUT/sq $OUT/valu $OUT/grbm > $OUT/summary.txt
cat $OUT/summary.txt
