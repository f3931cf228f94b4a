#!/bin/bash
# development helper: rebuild libnmma_hip.so and the stamped diagnostic build of the in-wave kernel
cd "$(dirname "$0")/.."
python -c "from nmma_amd import _lib; print(_lib.build_library(force=True))" || exit 1
mkdir -p build_dbg
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-comment -DNMMA_DEV_HEADLINE_ONLY -DNMMA_IW_STAMPS \
    nmma_amd/csrc/em_kernels.hip -o build_dbg/lib_STAMPS.so 2>&1 | grep -i "error" -A3
ls -la nmma_amd/libnmma_hip.so build_dbg/lib_STAMPS.so
