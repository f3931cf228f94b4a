#!/bin/bash
# Build libnmma_hip.so for gfx950 (cross-compiles without a GPU).  Usage: tools/build_lib.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-comment \
    nmma_amd/csrc/em_kernels.hip -o nmma_amd/libnmma_hip.so "$@"
