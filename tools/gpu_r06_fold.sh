#!/bin/bash
# Round 6, on the GPU box: kernel times of the flavours that run on a two-stage sample grid (stage-1 lerp folded into the rows):
# rocprofv3 --kernel-trace --stats per case -> gpurun_out/r06_fold/<case>/ ; the HIP-event line of tools/perf_case.py beside it
o=gpurun_out/r06_fold
rm -rf $o; mkdir -p $o
export TMPDIR=/tmp
for spec in "at2017gfo 4096" "c2_dt05_limit 4096" "c2_dt05 4096" "log_grid 4096" "c2_default 4096" "syserr_param 4096" "c4_dt05 8192" "c4_shape 8192"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/$1 -- python3 tools/perf_case.py $1 $2 > $o/$1.log 2>&1
  tail -1 $o/$1.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/models -- python3 tools/perf_models.py > $o/models.log 2>&1
grep -i "config 3\|stack2\|one launch" $o/models.log | head
for d in $o/*/; do f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && echo "== $d" && head -4 $f | cut -c1-160; done
