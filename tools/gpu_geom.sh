#!/bin/bash
# geometry sweep of em_logl: MFMA waves x likelihood waves (16-sample tiles), plus the 32-sample-tile variants
for cfg in "8 4" "8 8" "4 8" "4 4"; do
  set -- $cfg
  export NMMA_EM_MFMA_WAVES=$1 NMMA_EM_VALU_WAVES=$2
  echo "=== mfma_waves=$1 valu_waves=$2"
  timeout -s KILL 100 python tools/timeline.py 4096 2048 2>&1 | grep -E "A\(|B\(|prologue" | awk '{printf "%s ", $0} END {print ""}' | sed 's/  */ /g'
  timeout -s KILL 100 python tools/perf_probe.py 4096 2>&1 | grep "round 1 tile 1" | cut -c1-60
  timeout -s KILL 100 python tools/perf_probe.py 65536 2>&1 | grep "round 1 tile" | cut -c1-60
  timeout -s KILL 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
  timeout -s KILL 200 python tools/debug_w8.py 2>&1 | grep -v "amdgpu.ids\|mismatches 0"
done
exit 0
