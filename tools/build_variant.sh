#!/bin/bash
# development helper: build_dbg/lib_<name>.so = headline-only build with extra -D flags.  Usage: build_variant.sh <name> [-DX ...]
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_dbg
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-comment -DNMMA_DEV_HEADLINE_ONLY "$@" \
    nmma_amd/csrc/em_kernels.hip -o build_dbg/lib_$name.so 2>&1 | grep -i "error" -A3
ls -la build_dbg/lib_$name.so
