#!/bin/bash
# Runs ON THE GPU BOX: config 3's shape with the sub-models on their own grids (tools/perf_owngrids.py) -- wall times, then the same
# under rocprofv3 --kernel-trace --stats; and config 3 itself (tools/perf_models.py) for the FASTM 7 kernel's time after the change.
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_own
python3 tools/perf_owngrids.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_own/wall.log
NMMA_PERF_WARM=1500 NMMA_PERF_N=200 python3 tools/perf_owngrids.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_own/wall_steady.log
rm -rf gpurun_out/r06_own/prof gpurun_out/r06_own/prof_models
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_own/prof -- python3 tools/perf_owngrids.py > gpurun_out/r06_own/prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_own/prof_models -- python3 tools/perf_models.py > gpurun_out/r06_own/prof_models.log 2>&1
cat gpurun_out/r06_own/wall.log gpurun_out/r06_own/wall_steady.log
for d in prof prof_models; do
f=$(find gpurun_out/r06_own/$d -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06_own/${d}_kernel_stats.csv
python3 - <<PY
import csv
print("$d")
for r in csv.DictReader(open("$f")):
    n = r["Name"]
    if any(k in n for k in ("em_logl", "lc_loglike", "lc_regrid", "em_fused", "stack2_redo")):
        print("  ", n[:70].ljust(72), r["Calls"], r["AverageNs"], r["MinNs"])
PY
done
# the same two tools in the steady state (1500 untimed calls per timing, 200 timed ones), under rocprofv3: the headline's protocol
export NMMA_PERF_WARM=1500 NMMA_PERF_N=200
for t in owngrids models; do
rm -rf gpurun_out/r06_own/steady_$t
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_own/steady_$t -- python3 tools/perf_$t.py > gpurun_out/r06_own/steady_$t.log 2>&1
f=$(find gpurun_out/r06_own/steady_$t -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06_own/steady_${t}_kernel_stats.csv
echo "steady $t"; grep -v "amdgpu.ids\|rocprofv3\|HSA version\|Opened result" gpurun_out/r06_own/steady_$t.log | head -20
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    n = r["Name"]
    if any(k in n for k in ("em_logl", "lc_loglike", "lc_regrid", "em_fused", "stack2_redo")):
        print("  ", n[:70].ljust(72), r["Calls"], r["AverageNs"], r["MinNs"])
PY
done
