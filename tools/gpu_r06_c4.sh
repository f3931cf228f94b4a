for lib in "" build_dbg/lib_sleep16.so build_dbg/lib_sleep32.so build_dbg/lib_wake64.so build_dbg/lib_wake127.so; do
  echo "== lib: ${lib:-default}"
  for c in "c4_shape 8192" "c4_dt05 8192" "c4_syserr 8192" "c4_shape 65536" "c4_shape 4096"; do NMMA_HIP_LIB=$lib python3 tools/perf_case.py $c | tail -1 | cut -c1-80; done
done
