#!/bin/bash
# development helper: build_dbg/lib_<name>.so = the regular library with em_logl_f5.hip (the general lean task) recompiled with extra -D flags
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_dbg
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-comment -Iinclude "$@" -c nmma_amd/csrc/em_logl_f5.hip -o build_dbg/f5_$name.o 2>&1 | grep -i "error" -A3
objs=$(ls nmma_amd/csrc/build/*.o | grep -v em_logl_f5.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs build_dbg/f5_$name.o -o build_dbg/lib_$name.so 2>&1 | grep -i "error" -A3
rm -f build_dbg/f5_$name.o
ls -la build_dbg/lib_$name.so
