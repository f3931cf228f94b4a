#!/bin/bash
# Runs ON THE GPU BOX: kernel times of config 3's tail (tools/perf_models.py) for measurement builds (build_dbg/lib_<name>.so)
export TMPDIR=/tmp
for n in "$@"; do
  export NMMA_HIP_LIB=$PWD/build_dbg/lib_$n.so
  rm -rf gpurun_out/prof_lc_$n; mkdir -p gpurun_out/prof_lc_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lc_$n -- python3 tools/perf_models.py > gpurun_out/prof_lc_$n.log 2>&1
  f=$(find gpurun_out/prof_lc_$n -name "*kernel_stats.csv" | head -1)
  python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    n = r["Name"]
    if "lc_loglike<32" in n:
        print("$n".ljust(10), n[:50], r["Calls"], "avg", r["AverageNs"], "min", r["MinNs"])
PY
done
