#!/bin/bash
# Runs ON THE GPU BOX: the fused MCMC step's kernel time for measurement builds (build_dbg/lib_<name>.so)
export TMPDIR=/tmp
for n in "$@"; do
  export NMMA_HIP_LIB=$PWD/build_dbg/lib_$n.so
  rm -rf gpurun_out/prof_fuse_$n; mkdir -p gpurun_out/prof_fuse_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fuse_$n -- python3 bench.py --steps 5 --warmup 2 --repeats 3 --cpu-seconds 0.5 --sustained-seconds 0 > gpurun_out/prof_fuse_$n.log 2>&1
  f=$(find gpurun_out/prof_fuse_$n -name "*kernel_stats.csv" | head -1)
  python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    n = r["Name"]
    if "em_logl" in n and n.split(">")[0].rstrip().endswith(", 8"):
        print("$n".ljust(10), n[:50], r["Calls"], "avg", r["AverageNs"], "min", r["MinNs"])
PY
done
