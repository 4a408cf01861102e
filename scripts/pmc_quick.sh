#!/bin/bash
# two PMC passes (SQ busy/wait + VALU/LDS) of bench.py for a quick look at one build.  usage: bash scripts/pmc_quick.sh TAG
TAG=${1:-q}
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_valu -- $B > $OUT/pmc_valu.log 2>&1 || exit 1
python3 scripts/pmc_summary.py $OUT/pmc_sq $OUT/pmc_valu | grep -E "^#|bf16x3"
