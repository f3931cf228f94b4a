#!/bin/bash
# Runs ON THE GPU BOX: bench kernel time for each build_dbg/lib_<name>.so given (NAME or NAME:ENV=VAL,...)
for spec in "$@"; do
  n=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  env $envs NMMA_HIP_LIB=$PWD/build_dbg/lib_$n.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err
  python - <<PY
import json
try:
    d=json.load(open("/tmp/b.json")); print("$spec", "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,2), "frac", round(d["roofline"]["frac"],3), "block", d["config"]["launch"]["block"])
except Exception as e:
    print("$spec failed", e, open("/tmp/b.err").read()[-400:])
PY
done
