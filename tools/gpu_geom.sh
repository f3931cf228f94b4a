#!/bin/bash
# geometry sweep of em_logl at R = 1: MFMA waves x likelihood waves x lanes-per-sample
for cfg in "4 8 0" "4 8 1" "8 4 1" "8 4 0" "4 4 1" "8 8 1"; do
  set -- $cfg
  export NMMA_EM_MFMA_WAVES=$1 NMMA_EM_VALU_WAVES=$2
  if [ "$3" = "1" ]; then export NMMA_EM_G16=1; else unset NMMA_EM_G16; fi
  echo "=== mfma_waves=$1 valu_waves=$2 g16=$3"
  timeout 100 python tools/timeline.py 4096 2048 2>&1 | grep -E "A\(|B\(|prologue" | awk '{printf "%s ", $0} END {print ""}' | sed 's/  */ /g'
  timeout 100 python tools/perf_probe.py 4096 2>&1 | grep "round 1 tile 1" | cut -c1-60
  timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
