#!/bin/bash
# Build pde_amd/variants/libpdekan_TAG.so: the in-tree objects with ONE source recompiled under extra -D flags.
#   bash scripts/build_variant.sh TAG kan_f16 -DPDEKAN_X=1 ...
# (run python -m pde_amd.build first; bench.py / tests pick a variant up through PDEKAN_LIB)
TAG=$1; SRC=$2; shift; shift
D=pde_amd/build
mkdir -p pde_amd/variants
EXTRA=""; [ "$SRC" = "kan_f16" ] && EXTRA="-fno-slp-vectorize"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA -c pde_amd/csrc/$SRC.hip -o $D/${SRC}_$TAG.o -I pde_amd/csrc "$@" || exit 1
OBJS=""
for s in kan_kernels kan_mfma kan_head kan_sparse kan_f16 kan_small kan_deeponet pdekan_api; do
  if [ "$s" = "$SRC" ]; then OBJS="$OBJS $D/${s}_$TAG.o"; else OBJS="$OBJS $D/$s.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o pde_amd/variants/libpdekan_$TAG.so $OBJS || exit 1
echo pde_amd/variants/libpdekan_$TAG.so
