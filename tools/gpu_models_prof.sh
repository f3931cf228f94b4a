#!/bin/bash
# Runs ON THE GPU BOX: the models tests, then kernel times of the config-3 tail (rocprofv3 --kernel-trace --stats of tools/perf_models.py)
python -m pytest tests/test_gpu_models.py -q 2>&1 | grep -v amdgpu.ids | tail -5
export TMPDIR=/tmp
rm -rf gpurun_out/prof_models
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_models -- python3 tools/perf_models.py > gpurun_out/prof_models.log 2>&1
grep -v amdgpu gpurun_out/prof_models.log | grep -A3 "config 3"
f=$(find gpurun_out/prof_models -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    n = r["Name"]
    if any(k in n for k in ("lc_loglike", "lc_stack")):
        print(n[:60].ljust(62), r["Calls"], r["AverageNs"], r["MinNs"])
PY
