#!/bin/bash
# HBM traffic counters only (two passes), for a quick check after a kernel change.  usage: bash scripts/pmc_traffic.sh TAG
TAG=${1:-t}
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
OUT=gpurun_out/traffic_$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || exit 1
python3 scripts/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write | grep -E "mfma|sparse|head"
