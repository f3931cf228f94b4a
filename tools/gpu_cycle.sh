#!/bin/bash
# One GPU round trip used while tuning em_logl: parity tests, in-kernel timeline, throughput probes.
# Usage (on the GPU box, from the repo root): tools/gpu_cycle.sh [tag]
tag=${1:-cur}
o=gpurun_out
mkdir -p $o
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $o/pytest_gpu.log
NMMA_EM_MFMA_WAVES=8 timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $o/pytest_gpu8.log
timeout -s KILL 100 python tools/timeline.py 4096 2048 > $o/tl4_$tag.log 2>&1
NMMA_EM_MFMA_WAVES=8 timeout -s KILL 100 python tools/timeline.py 4096 2048 > $o/tl8_$tag.log 2>&1
timeout -s KILL 200 python tools/perf_probe.py 4096 > $o/probe_4096_$tag.log 2>&1
NMMA_EM_MFMA_WAVES=8 timeout -s KILL 200 python tools/perf_probe.py 4096 > $o/probe_4096_w8_$tag.log 2>&1
timeout -s KILL 200 python tools/perf_probe.py 65536 > $o/probe_65536_$tag.log 2>&1
cat $o/pytest_gpu.log $o/pytest_gpu8.log
echo "--- timeline 4 MFMA waves"; grep -v amdgpu.ids $o/tl4_$tag.log
echo "--- timeline 8 MFMA waves"; grep -v amdgpu.ids $o/tl8_$tag.log
for f in probe_4096_$tag probe_4096_w8_$tag probe_65536_$tag; do echo "--- $f"; tail -n 4 $o/$f.log | cut -c1-230; done
