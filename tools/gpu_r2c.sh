#!/bin/bash
o=gpurun_out/r2c; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gputests.log 2>&1; tail -3 $o/gputests.log
python tools/timeline_iw.py 4096 > $o/timeline_iw.log 2>&1; cat $o/timeline_iw.log; NMMA_HIP_LIB=$PWD/build_dbg/lib_STAMPS.so python tools/timeline_iw.py 4096 2>&1 | tail -2
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; python - <<PY
import json; d=json.load(open("$o/bench.json")); print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "ms_per_step", d["ms_per_step"])
PY
