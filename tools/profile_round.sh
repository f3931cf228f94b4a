#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the evidence behind the bench line, written under gpurun_out/<tag>/ (wiped first) and
# condensed by tools/summarize_profiles.py into the tracked profiles/<tag>_*:
#   bench line (with cpu baselines)          -> <tag>/bench_line.json
#   rocprofv3 --kernel-trace --stats         -> <tag>/stats_<what>/   for: bench (config 2, lean task), c2_dt05_limit @ 4096
#                                               (general lean task; dt05ext: the same on the extended task), c4_shape @ 8192 (config 4 shape), models (Me2017, combined,
#                                               em_fused outputs), bench with the opt-in in-wave kernel, gw (inner products)
#   rocprofv3 --pmc (separate passes)        -> <tag>/pmc_<set>/      for the bench command
# Under rocprofv3 the program goes directly after `--` (no env / bash -c hops).
tag=${1:-r03}
o=gpurun_out/$tag
rm -rf $o; mkdir -p $o
export TMPDIR=/tmp
python3 bench.py --steps 200 --warmup 20 > $o/bench_line.json 2> $o/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_line_driver_flags.json 2> $o/bench_driver.err     # what the driver runs
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_bench -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --sustained-seconds 0 > $o/stats_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_dt05 -- python3 tools/perf_case.py c2_dt05_limit 4096 > $o/stats_dt05.log 2>&1
export NMMA_EM_NO_LEAN_LIM=1       # the same case on the extended task (em_logl<.., 2>), which had the finite limits before
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_dt05ext -- python3 tools/perf_case.py c2_dt05_limit 4096 > $o/stats_dt05ext.log 2>&1
unset NMMA_EM_NO_LEAN_LIM
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_c4 -- python3 tools/perf_case.py c4_shape 8192 > $o/stats_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_models -- python3 tools/perf_models.py > $o/stats_models.log 2>&1
if [ -z "$SKIP_GW" ]; then       # (round 4: the GW leg is frozen -- SKIP_GW=1 leaves its profiles of round 3 in place)
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_gw -- python3 tools/perf_gw.py 2048 > $o/stats_gw.log 2>&1
# the GW leg from parameters at config 5's shape (16 384 samples x 259 585 bins x 3 detectors), with and without phase marginalisation
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_gw_fused -- python3 tools/perf_gw_fused.py --batch 16384 --reps 5 > $o/stats_gw_fused.log 2>&1
python3 tools/perf_gw_fused.py --batch 16384 --reps 5 --pm > $o/gw_fused_pm.log 2>&1
python3 tools/perf_gw_fused.py --batch 2048 --reps 5 > $o/gw_fused_2048.log 2>&1
# the marginalised forms of the GW likelihood at config 5's shape: distance + phase, time, all three
python3 tools/perf_gw_fused.py --batch 16384 --reps 3 --dm --pm > $o/gw_fused_dm_pm.log 2>&1
python3 tools/perf_gw_fused.py --batch 16384 --reps 3 --tm > $o/gw_fused_tm.log 2>&1
python3 tools/perf_gw_fused.py --batch 16384 --reps 1 --tm --dm --pm > $o/gw_fused_tm_dm_pm.log 2>&1
bash tools/pmc_gw.sh $tag > $o/pmc_gw.log 2>&1
fi
bash tools/pmc_models.sh > $o/pmc_models.log 2>&1
bash tools/pmc_c4.sh $tag > $o/pmc_c4.log 2>&1
# the real AT2017gfo photometry (9 filters, CLI grid, sampled em_syserr)
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_at2017gfo -- python3 tools/perf_case.py at2017gfo 4096 > $o/stats_at2017gfo.log 2>&1
# small batches: kernel time against the batch size with and without the band split; bench lines at 512 rows and at one row
python3 tools/perf_small_batch.py > $o/small_batch.log 2>&1
python3 bench.py --steps 200 --warmup 20 --batch 512 --no-cpu-baseline --sustained-seconds 0 > $o/bench_line_b512.json 2> $o/bench_b512.err
python3 bench.py --steps 200 --warmup 20 --batch 1 --no-cpu-baseline --sustained-seconds 0 > $o/bench_line_b1.json 2> $o/bench_b1.err
# groups of three / two bands per workgroup
python3 bench.py --steps 200 --warmup 20 --batch 1024 --no-cpu-baseline --sustained-seconds 0 > $o/bench_line_b1024.json 2> $o/bench_b1024.err
python3 bench.py --steps 200 --warmup 20 --batch 2048 --no-cpu-baseline --sustained-seconds 0 > $o/bench_line_b2048.json 2> $o/bench_b2048.err
python3 tools/perf_table.py > $o/perf_table.log 2>&1
# the lock-step ensemble walk on the device against the host walk (MCMC steps of 4096 chains)
for n in 1024 2048 4096 16384; do python3 tools/perf_device_walk.py $n 400; done > $o/device_walk.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_device_walk -- python3 tools/perf_device_walk.py 4096 400 > $o/stats_device_walk.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/pmc_$c -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $o/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/pmc_sq -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $o/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $o/pmc_sq2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $o/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $o/pmc_sq3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $o/pmc_sq3.log 2>&1
find $o -name "*kernel_stats.csv" | head
cat $o/bench_line.json
# the sampler queue (4096 records x 100 steps): kernel times of the fused MCMC step and of the queue's other launches
bash tools/gpu_fuse_prof.sh > $o/stats_queue.log 2>&1
cp $(find gpurun_out/prof_fuse/a -name "*kernel_stats.csv" | head -1) $o/queue_kernel_stats.csv 2>/dev/null
