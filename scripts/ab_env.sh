#!/bin/bash
# A/B a library switch on ONE box: bench.py with and without the environment variable given as $1 (e.g. PDEKAN_BWW_NB=1).
# usage: bash scripts/ab_env.sh PDEKAN_BWW_NB=1 [extra bench.py flags]
VAR=$1; shift
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/ab_a.json || exit 1
export "$VAR"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/ab_b.json || exit 1
echo "== baseline"; python scripts/show_bench.py gpurun_out/ab_a.json | head -8
echo "== $VAR"; python scripts/show_bench.py gpurun_out/ab_b.json | head -8
