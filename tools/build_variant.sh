#!/bin/bash
# development helper: build_dbg/lib_<name>.so = the library with a headline-only EM unit built with extra -D flags (the GW and walk
# units are linked from the regular build's objects).  Usage: build_variant.sh <name> [-DX ...]
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_dbg
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-comment -DNMMA_DEV_HEADLINE_ONLY "$@" \
    -c nmma_amd/csrc/em_kernels.hip -o build_dbg/em_$name.o 2>&1 | grep -i "error" -A3
hipcc --offload-arch=gfx950 -shared -fPIC build_dbg/em_$name.o nmma_amd/csrc/build/gw_kernels.o nmma_amd/csrc/build/walk_kernels.o \
    -o build_dbg/lib_$name.so 2>&1 | grep -i "error" -A3
rm -f build_dbg/em_$name.o
ls -la build_dbg/lib_$name.so
