#!/bin/bash
# Build libnmma_hip.so for gfx950 (cross-compiles without a GPU).  Usage: tools/build_lib.sh [extra hipcc flags]
# (same recipe as nmma_amd/_lib.py:build_library: one object per translation unit under nmma_amd/csrc/build/, then the link)
set -e
cd "$(dirname "$0")/.."
if [ $# -gt 0 ]; then
    python3 -c "import sys; from nmma_amd import _lib; print(_lib.build_library(force=True, extra_flags=sys.argv[1:]))" "$@"
else
    python3 -c "from nmma_amd import _lib; print(_lib.build_library())"
fi
