#!/bin/bash
# round-2 GPU check: parity tests, timeline, quick bench, instruction-mix counters of em_logl
o=gpurun_out/r2b; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gputests.log 2>&1; tail -5 $o/gputests.log
python tools/timeline.py 4096 > $o/timeline.log 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; cat $o/bench.json
bash tools/pmc_quick.sh r2b > $o/pmc.log 2>&1; cat $o/pmc.log | tail -14
