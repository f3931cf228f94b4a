#!/bin/bash
# priority sweep of em_logl's two roles (throughput probe at 4096 and 65536 rows)
for cfg in "3,0" "2,0" "1,0" "0,0" "3,1" "2,1" "1,1" "0,1"; do
  echo "=== prio(likelihood,mfma)=$cfg"
  for i in 1 2; do NMMA_EM_PRIO=$cfg timeout -s KILL 100 python tools/perf_probe.py 4096 1 2>&1 | grep "round 1 tile 1" | cut -c1-50; done
  NMMA_EM_PRIO=$cfg timeout -s KILL 100 python tools/perf_probe.py 65536 2 2>&1 | grep "round 1 tile 2" | cut -c1-50
done
