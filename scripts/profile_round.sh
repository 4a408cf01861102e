#!/bin/bash
# rocprofv3 evidence for bench.py's default workload: kernel-trace stats + PMC passes (one counter block per run).
# usage (on the GPU box, from the repo root):  bash scripts/profile_round.sh r01
TAG=${1:-r01}
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# the stats pass runs longer than the counter passes: the first launches after idle run at lower clocks
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_valu -- $B > $OUT/pmc_valu.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || exit 1
grep "^{\"metric\"" $OUT/stats.log | tail -1 > $OUT/bench_line.json
echo done
