#!/bin/bash
# Runs ON THE GPU BOX: config 4's shape with a sampled em_syserr (dense task, SYS) -- the Chebyshev table of sum ln sigma_tot (round 6)
# against NMMA_EM_NO_LNSIG_TAB=1 (a logarithm and a reciprocal square root per datum and sample), under rocprofv3 --kernel-trace --stats;
# plus the parity tests that cover the flavour.
export TMPDIR=/tmp
o=gpurun_out/r06_c4sys
rm -rf $o; mkdir -p $o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -k "c4 or dense or syserr" 2>&1 | tail -2
for v in tab notab; do
  if [ $v = notab ]; then export NMMA_EM_NO_LNSIG_TAB=1; fi
  for rows in 8192 65536; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $o/${v}_$rows -- python3 tools/perf_case.py c4_syserr $rows > $o/${v}_$rows.log 2>&1
    f=$(find $o/${v}_$rows -name "*kernel_stats.csv" | head -1); cp $f $o/${v}_${rows}_kernel_stats.csv
    echo "$v $rows: $(grep em_logl $f | cut -d'"' -f3 | cut -d, -f2-4) | $(grep 'us per launch' $o/${v}_$rows.log | cut -d, -f1)"
  done
done
unset NMMA_EM_NO_LNSIG_TAB
rocprofv3 --kernel-trace --stats --output-format csv -d $o/plain_8192 -- python3 tools/perf_case.py c4_shape 8192 > $o/plain_8192.log 2>&1
f=$(find $o/plain_8192 -name "*kernel_stats.csv" | head -1); echo "c4_shape (constant systematics) 8192: $(grep em_logl $f | cut -d'"' -f3 | cut -d, -f2-4)"
