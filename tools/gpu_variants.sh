#!/bin/bash
# Runs ON THE GPU BOX: bench kernel time for each build_dbg/lib_<name>.so given
for n in "$@"; do
  NMMA_HIP_LIB=$PWD/build_dbg/lib_$n.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline > /tmp/b_$n.json 2>/tmp/b_$n.err
  python - <<PY
import json
try:
    d=json.load(open("/tmp/b_$n.json")); print("$n", "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,2), "frac", round(d["roofline"]["frac"],3))
except Exception as e:
    print("$n failed", e, open("/tmp/b_$n.err").read()[-400:])
PY
done
