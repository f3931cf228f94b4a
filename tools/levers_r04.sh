#!/bin/bash
# Runs ON THE GPU BOX.  VERDICT r03 item 5: the four levers on the headline kernel em_logl<1,1,8,8,1>, each as a measurement build of
# the headline-only library (tools/build_variant.sh; NMMA_HIP_LIB selects it) against the same source without the flag ("base"):
#   nochains        (a) ceiling of a per-sample-scalars pre-pass: the prologue's four chains replaced by constants (-DNMMA_DBG_NOCHAINS)
#   nodiv           (b) ceiling of DMA-staged prologue tables: reciprocal grid spacings without their division (-DNMMA_DBG_NODIV)
#   nochains_nodiv  (a) + (b)
#   sleep15, wake15, wake40   (c) polls: s_sleep 15 without / with s_wakeup on every signal; s_sleep 40 with wake-ups
#   preload         (d) MFMA role reads theta AFTER issuing the record ring's first loads (-DNMMA_DBG_PRELOAD_FIRST)
# Per variant: kernel time in the bench loop (HIP events, 200 steps), MCMC-step time of the device walk (theta produced on the
# device by the previous launch), and the SQ counters of 10 launches at 4096 rows (two rocprofv3 --pmc passes).
export TMPDIR=/tmp
o=gpurun_out/levers_r04
rm -rf $o; mkdir -p $o
for n in base nochains nodiv nochains_nodiv sleep15 wake15 wake40 preload base; do
  export NMMA_HIP_LIB=$PWD/build_dbg/lib_$n.so
  python3 bench.py --steps 200 --warmup 20 --repeats 9 --no-cpu-baseline > $o/b_$n.json 2> $o/b_$n.err
  python3 tools/perf_device_walk.py 4096 400 2>/dev/null | head -1 > $o/w_$n.log
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $o/p1_$n -- python3 tools/run_mode.py loglike 4096 12 > $o/p1_$n.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $o/p2_$n -- python3 tools/run_mode.py loglike 4096 12 > $o/p2_$n.log 2>&1
  python3 - <<PY
import csv, glob, json, collections
n = "$n"
try:
    d = json.load(open("$o/b_%s.json" % n)); k = d["roofline"]["kernel_ms"] * 1e3; step = d["ms_per_step"] * 1e3
except Exception as e:
    k = step = float("nan")
walk = open("$o/w_%s.log" % n).read().strip().split(" us per")[0].split()[-1] if open("$o/w_%s.log" % n).read().strip() else "nan"
acc = collections.defaultdict(list)
for sub in ("p1", "p2"):
    for f in glob.glob("$o/%s_%s/**/*counter_collection.csv" % (sub, n), recursive=True):
        for row in csv.DictReader(open(f)):
            if "em_logl" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {c: sum(v[2:]) / max(1, len(v[2:])) for c, v in acc.items()}
g = lambda c: m.get(c, float("nan"))
print(f"{n:15s} kernel {k:6.2f} us  step {step:6.2f} us  walk-step {walk:>6s} us | VALU {g('SQ_INSTS_VALU')/1e6:6.3f} M  MFMA {g('SQ_INSTS_MFMA')/1e6:6.3f} M  "
      f"VALU/MFMA {g('SQ_INSTS_VALU')/max(1.0, g('SQ_INSTS_MFMA')):5.2f}  SALU {g('SQ_INSTS_SALU')/1e6:6.3f} M  LDS {g('SQ_INSTS_LDS')/1e6:6.3f} M | "
      f"MFMA busy {g('SQ_VALU_MFMA_BUSY_CYCLES')/1024/1e3:6.2f} k  wave-cycles/SIMD {g('SQ_WAVE_CYCLES')/1024/1e3:7.2f} k  wait {g('SQ_WAIT_INST_ANY')/1024/1e3:7.2f} k")
PY
done
