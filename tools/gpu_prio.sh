#!/bin/bash
# priority / geometry sweep of em_logl (timeline of workgroup 0 + throughput probe)
for cfg in "3,0" "2,0" "1,0" "0,0" "0,1" "1,2" "3,1"; do
  for w in 4 8; do
    echo "=== prio(valu,mfma)=$cfg mfma_waves=$w"
    NMMA_EM_PRIO=$cfg NMMA_EM_MFMA_WAVES=$w timeout -s KILL 100 python tools/timeline.py 4096 2048 2>&1 | grep -E "A\(|B\(|prologue" | awk '{printf "%s ", $0} END {print ""}' | sed 's/  */ /g'
    NMMA_EM_PRIO=$cfg NMMA_EM_MFMA_WAVES=$w timeout -s KILL 100 python tools/perf_probe.py 4096 2>&1 | grep "round 1 tile 1" | cut -c1-60
  done
done
