#!/bin/bash
# Round 6, on the GPU box: config 4's shape with the library and with development variants of its unit that change how a task waits for
# the surrogate -- build them first (they are not tracked):
#   tools/build_unit_variant.sh em_logl_f6 sleep16 -DNMMA_DENSE_SLEEP=16      tools/build_unit_variant.sh em_logl_f6 sleep32 -DNMMA_DENSE_SLEEP=32
#   tools/build_unit_variant.sh em_logl_f6 wake64 -DNMMA_SYNC_WAKEUP -DNMMA_DENSE_SLEEP=64
#   tools/build_unit_variant.sh em_logl_f6 wake127 -DNMMA_SYNC_WAKEUP -DNMMA_DENSE_SLEEP=127 -DNMMA_SYNC_SLEEP=32
# (profiles/r06_c4.md: 135.2-136.3 us for all five -- polling is not what bounds the launch)
for lib in "" build_dbg/lib_sleep16.so build_dbg/lib_sleep32.so build_dbg/lib_wake64.so build_dbg/lib_wake127.so; do
  echo "== lib: ${lib:-default}"
  for c in "c4_shape 8192" "c4_dt05 8192" "c4_syserr 8192" "c4_shape 65536" "c4_shape 4096"; do NMMA_HIP_LIB=$lib python3 tools/perf_case.py $c | tail -1 | cut -c1-80; done
done
