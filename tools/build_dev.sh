#!/bin/bash
# development helper: rebuild libnmma_hip.so and a stamped headline-only diagnostic build
cd "$(dirname "$0")/.."
python -c "from nmma_amd import _lib; print(_lib.build_library(force=True))" || exit 1
tools/build_variant.sh STAMPS -DNMMA_DBG_TASKSTAMPS
ls -la nmma_amd/libnmma_hip.so build_dbg/lib_STAMPS.so
