#!/usr/bin/env python
"""Condense gpurun_out/r06_steady/ (tools/gpu_r06_steady.sh) into profiles/r06_steady_state.log and one
profiles/r06_steady_<case>_<rows>_kernel_stats.csv per case: average of the last 400 launches of the rocprofv3 kernel trace, the
HIP-event time tools/perf_case.py printed for the same 400, and the CSV's own average over all 2400 launches."""
import csv
import glob
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r06_steady")
dst = os.path.join(ROOT, "profiles")
lines = ["# tools/gpu_r06_steady.sh on MI355X: 2000 untimed launches, then 400 timed ones, under rocprofv3 --kernel-trace --stats",
         "# case, rows: average of the LAST 400 launches of the rocprofv3 kernel trace | tools/perf_case.py's HIP-event time of the same 400 | "
         "all 2400 launches (the CSV's average, clock ramp included)"]
for d in sorted(glob.glob(os.path.join(src, "*_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not traces or not stats:
        continue
    trace, stat = max(traces, key=os.path.getmtime), max(stats, key=os.path.getmtime)
    t = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
               for r in csv.DictReader(open(trace)) if "em_logl" in r["Kernel_Name"])
    last = [x[1] for x in t[-400:]]
    kern = re.search(r"em_logl<[^>]*>", t[-1][2]).group(0)
    allavg = [float(r["AverageNs"]) for r in csv.DictReader(open(stat)) if "em_logl" in r["Name"]][0]
    log = open(os.path.join(src, name + ".log")).read()
    ev = re.findall(r"([\d.]+) us per launch", log)
    lines.append(f"{name:22s} {kern:28s} last 400: {sum(last) / len(last) / 1e3:8.2f} us | HIP events {float(ev[-1]):7.1f} us | all {len(t)}: {allavg / 1e3:8.2f} us")
    shutil.copy(stat, os.path.join(dst, f"r06_steady_{name}_kernel_stats.csv"))
open(os.path.join(dst, "r06_steady_state.log"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
