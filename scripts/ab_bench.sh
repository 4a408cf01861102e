#!/bin/bash
# A/B of library variants inside ONE GPU call: per-kernel times of a short cfg2 bench for each variant named.
#   bash scripts/ab_bench.sh base pad fixc0      (base = the in-tree library; X = pde_amd/variants/libpdekan_X.so)
for v in "$@"; do
  if [ "$v" = "base" ]; then unset PDEKAN_LIB; else export PDEKAN_LIB=$PWD/pde_amd/variants/libpdekan_$v.so; fi
  timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --workload cfg2 --no-cpu-baseline --no-scaling-reference > gpurun_out/ab_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/ab_$v.log; exit 1; }
  grep "^{" gpurun_out/ab_$v.log > gpurun_out/ab_$v.json
  echo "== $v"; python3 scripts/show_bench.py gpurun_out/ab_$v.json | head -5
done
