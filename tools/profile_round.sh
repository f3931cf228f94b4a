#!/bin/bash
# Runs ON THE GPU BOX (gpurun): bench line + rocprofv3 kernel stats + PMC passes for em_logl.
# Usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>_*)
tag=${1:-r01}
o=gpurun_out
mkdir -p $o
export TMPDIR=/tmp
python3 bench.py --steps 200 --warmup 20 > $o/${tag}_bench_line.json 2> $o/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $o/${tag}_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_pmc_$c -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/${tag}_pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/${tag}_pmc_sq -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/${tag}_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $o/${tag}_pmc_sq2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/${tag}_pmc_sq2.log 2>&1
find $o/${tag}_stats $o/${tag}_pmc_* -name "*.csv" | head -40
cat $o/${tag}_bench_line.json
