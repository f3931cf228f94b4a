#!/bin/bash
# development helper: build_dbg/lib_<name>.so = the regular library with ONE translation unit recompiled with extra -D flags
# Usage: build_unit_variant.sh <unit, e.g. em_logl_f6> <name> [-DX ...]   (run it: NMMA_HIP_LIB=build_dbg/lib_<name>.so python ...)
cd "$(dirname "$0")/.."
unit=$1; name=$2; shift; shift
mkdir -p build_dbg
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-comment "$@" -c nmma_amd/csrc/$unit.hip -o build_dbg/${unit}_$name.o 2>&1 | grep -i "error" -A3
objs=$(ls nmma_amd/csrc/build/*.o | grep -v "/$unit.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs build_dbg/${unit}_$name.o -o build_dbg/lib_$name.so 2>&1 | grep -i "error" -A3
rm -f build_dbg/${unit}_$name.o
ls -la build_dbg/lib_$name.so
