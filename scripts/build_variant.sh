#!/bin/bash
# Build pde_amd/variants/libpdekan_TAG.so: the in-tree objects with kan_f16.hip recompiled under extra -D flags.
#   bash scripts/build_variant.sh TAG -DPDEKAN_X=1 ...
# (run python -m pde_amd.build first; bench.py / tests pick a variant up through PDEKAN_LIB)
TAG=$1; shift
D=pde_amd/build
mkdir -p pde_amd/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c pde_amd/csrc/kan_f16.hip -o $D/kan_f16_$TAG.o -I pde_amd/csrc "$@" || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o pde_amd/variants/libpdekan_$TAG.so $D/kan_kernels.o $D/kan_mfma.o $D/kan_head.o $D/kan_sparse.o $D/kan_f16_$TAG.o $D/kan_small.o $D/pdekan_api.o || exit 1
echo pde_amd/variants/libpdekan_$TAG.so
