#!/bin/bash
# development helper: build_dbg/lib_<name>.so = the library with a headline-only EM build (em_kernels.hip + the one em_logl
# instantiation of em_logl_f1.hip, -DNMMA_DEV_HEADLINE_ONLY) compiled with extra -D flags; the GW and walk units are linked from the
# regular build's objects.  Usage: build_variant.sh <name> [-DX ...]
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_dbg
for u in em_kernels em_logl_f1; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-comment -DNMMA_DEV_HEADLINE_ONLY "$@" \
        -c nmma_amd/csrc/$u.hip -o build_dbg/${u}_$name.o 2>&1 | grep -i "error" -A3 &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC build_dbg/em_kernels_$name.o build_dbg/em_logl_f1_$name.o nmma_amd/csrc/build/gw_kernels.o \
    nmma_amd/csrc/build/walk_kernels.o -o build_dbg/lib_$name.so 2>&1 | grep -i "error" -A3
rm -f build_dbg/em_kernels_$name.o build_dbg/em_logl_f1_$name.o
ls -la build_dbg/lib_$name.so
