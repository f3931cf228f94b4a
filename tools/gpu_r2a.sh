mkdir -p gpurun_out/r2a
./tools/ubench/valu_mix > gpurun_out/r2a/valu_mix.log 2>&1
python tools/timeline.py 4096 > gpurun_out/r2a/timeline.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/gputests.log 2>&1
tail -3 gpurun_out/r2a/gputests.log
cat gpurun_out/r2a/valu_mix.log
