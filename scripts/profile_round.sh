#!/bin/bash
# rocprofv3 evidence for one bench.py workload: kernel-trace stats + PMC passes (one counter block per run).
# usage (on the GPU box, from the repo root):  bash scripts/profile_round.sh TAG [WORKLOAD] [STEPS]
# writes gpurun_out/prof_TAG/{stats,pmc_*}, kernel_stats.csv, pmc_summary.txt (header line = what bench.py checks
# before quoting HBM traffic from it: workload, points per GPU, digest of the kernel sources) and bench_line.json
TAG=${1:-r03}
WL=${2:-cfg2}
STEPS=${3:-30}
export TMPDIR=/tmp
ARGS="--workload $WL --no-cpu-baseline --no-scaling-reference"
B="bench.py --steps 5 --warmup 2 $ARGS"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# the stats pass runs longer than the counter passes: the first launches after idle run at lower clocks
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps $STEPS --warmup 5 $ARGS > $OUT/stats.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $B > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_valu -- python3 $B > $OUT/pmc_valu.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- python3 $B > $OUT/pmc_grbm.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B > $OUT/pmc_write.log 2>&1 || exit 1
grep "^{\"metric\"" $OUT/stats.log | tail -1 > $OUT/bench_line.json
python3 - "$OUT" "$WL" <<'PY'
import glob, json, os, shutil, subprocess, sys
out, wl = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
import bench
line = json.load(open(os.path.join(out, "bench_line.json")))
n_local = line["config"]["points_per_gpu"]
head = f"# workload={wl} n_local={n_local} sources={bench.sources_digest()}"
body = subprocess.run([sys.executable, "scripts/pmc_summary.py"] + [os.path.join(out, d) for d in
                      ("pmc_sq", "pmc_valu", "pmc_grbm", "pmc_fetch", "pmc_write")], capture_output=True, text=True).stdout
open(os.path.join(out, "pmc_summary.txt"), "w").write(head + "\n" + body)
stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
print(head)
PY
echo done
