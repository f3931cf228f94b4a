#!/bin/bash
# Runs ON THE GPU BOX: the GPU test suite and a short bench (kernel time, roofline fraction) -- the quick check after a kernel change
o=gpurun_out/check; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gputests.log 2>&1; tail -15 $o/gputests.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; python - <<PY
import json; d=json.load(open("$o/bench.json")); print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "ms_per_step", d["ms_per_step"])
PY
