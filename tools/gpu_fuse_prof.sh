#!/bin/bash
# Runs ON THE GPU BOX: kernel times of the sampler queue (fused MCMC step vs likelihood + walk_step) by rocprofv3
export TMPDIR=/tmp
rm -rf gpurun_out/prof_fuse; mkdir -p gpurun_out/prof_fuse
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fuse/a -- python3 bench.py --steps 5 --warmup 2 --repeats 3 --cpu-seconds 0.5 --sustained-seconds 0 > gpurun_out/prof_fuse/a.log 2>&1
f=$(find gpurun_out/prof_fuse/a -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    n = r["Name"]
    if "em_logl" in n or "walk" in n:
        print(n[:90].ljust(92), r["Calls"], r["AverageNs"], r["MinNs"])
PY
