#!/bin/bash
# Runs ON THE GPU BOX: the combined model's one-launch form -- tests, timings, rocprofv3 kernel stats -> gpurun_out/stack2/
o=gpurun_out/stack2
rm -rf $o; mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests/test_gpu_stack2.py tests/test_gpu_models.py tests/test_gpu_configs.py -x -q -s -k "stack2 or combined or config3 or one_launch or union or handles" > $o/tests.log 2>&1
tail -5 $o/tests.log
python3 tools/perf_models.py > $o/perf_models.log 2>&1
grep "config 3" -A3 $o/perf_models.log
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_models -- python3 tools/perf_models.py > $o/stats_models.log 2>&1
f=$(find $o/stats_models -name "*kernel_stats.csv" | head -1)
cp $f $o/models_kernel_stats.csv
head -12 $o/models_kernel_stats.csv | cut -c1-200
