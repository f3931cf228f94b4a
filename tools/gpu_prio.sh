#!/bin/bash
# Runs ON THE GPU BOX: priority sweep of em_logl's two roles (NMMA_EM_PRIO="likelihood,mfma", read at create) at 4096 and 65536 rows
for cfg in "3,0" "2,0" "1,0" "0,0" "3,1" "2,1" "1,1" "0,1"; do
  echo "=== prio(likelihood,mfma)=$cfg"
  for b in 4096 4096 65536; do NMMA_EM_PRIO=$cfg timeout -s KILL 100 python tools/perf_case.py c2_default $b 2>&1 | grep "us per launch" | cut -c1-80; done
done
