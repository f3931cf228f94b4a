#!/bin/bash
# Round 6, on the GPU box: the flavours real data takes IN THE STEADY STATE -- the committed <tag>_*_kernel_stats.csv of these cases are
# 35-launch profiles of a fresh process (ramping clocks: config 2's own kernel reads 30.8 us there, 27.2 us in bench.py's steady state),
# the headline's is not.  Here every case gets 2000 untimed launches first (tools/perf_case.py, NMMA_PERF_WARM) and 400 timed ones, under
# rocprofv3 --kernel-trace --stats; the average of the LAST 400 launches of the trace is printed next to the HIP-event time.
o=gpurun_out/r06_steady
rm -rf $o; mkdir -p $o
export TMPDIR=/tmp NMMA_PERF_WARM=2000 NMMA_PERF_N=400
for spec in "c2_default 4096" "at2017gfo 4096" "c2_dt05_limit 4096" "c2_dt05 4096" "log_grid 4096" "syserr_param 4096" "averaging 4096" "c4_shape 8192" "c4_dt05 8192" "c4_shape 65536"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/$1_$2 -- python3 tools/perf_case.py $1 $2 > $o/$1_$2.log 2>&1
  python3 - "$o/$1_$2" "$1" "$2" "$(grep "us per launch" $o/$1_$2.log | tail -1 | cut -d, -f1)" <<'PY'
import csv, glob, sys
d, name, rows, ev = sys.argv[1:5]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
t = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "em_logl" in r["Kernel_Name"]]
t.sort()
last = [x[1] for x in t[-400:]]
print(f"{name:14s} {rows:>6s} rows: last 400 of {len(t)} launches {sum(last) / len(last) / 1e3:8.2f} us (rocprofv3 trace) | {ev}")
PY
done | tee $o/steady_state.log
