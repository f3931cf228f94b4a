#!/usr/bin/env python
"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh, which wipes the directory first) into the tracked
summaries under profiles/: one kernel-stats CSV per profiled command, PMC means per dispatch of the likelihood kernel,
HBM traffic per launch, the bench line."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


for what in ("bench", "dt05", "dt05ext", "c4", "models", "gw", "gw_fused", "at2017gfo", "device_walk"):
    f = newest(os.path.join(src, f"stats_{what}", "**", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_{what}_kernel_stats.csv"))
        rows = list(csv.DictReader(open(f)))
        top = sorted(rows, key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))[:4]
        for r in top:
            print(f"{what:7s} {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.2f} us")

pmc = {}
for sub in ("FETCH_SIZE", "WRITE_SIZE", "sq", "sq2", "sq3"):
    f = newest(os.path.join(src, f"pmc_{sub}", "**", "*counter_collection.csv"))
    if not f:
        continue
    acc = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if "em_logl" in r.get("Kernel_Name", "")]
    # the bench also times single-point host calls: keep the full-batch launches (largest grid) only
    full = max((int(r["Grid_Size"]) for r in rows), default=0)
    for row in rows:
        if int(row["Grid_Size"]) == full:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v = v[5:] if len(v) > 10 else v          # drop the warm-up launches
        pmc[k] = {"n_dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
pmc["_note"] = ("rocprofv3 --pmc (separate passes, --kernel-trace only) on `python3 bench.py --steps 20 --warmup 5 "
                "--no-cpu-baseline`; likelihood kernel em_logl, B=4096, per dispatch over the whole GPU (1024 SIMDs)")
json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc_em_logl_summary.json"), "w"), indent=1)

if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # FETCH_SIZE / WRITE_SIZE are in KiB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts
    # 64-B units for 128-B requests -> doubled
    fetch = pmc["FETCH_SIZE"]["mean"] * 1024 * 2
    write = pmc["WRITE_SIZE"]["mean"] * 1024
    traffic = {
        "bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 20 "
                  "--warmup 5`, likelihood kernel, B=4096; FETCH_SIZE in KiB doubled per MI355X_MICROARCH.md (HBM section)",
        # theta in + logL out + every weight/table byte once
        "algorithmic_bytes": 1088432,
    }
    json.dump(traffic, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)

f = os.path.join(src, "queue_kernel_stats.csv")
if os.path.exists(f):
    shutil.copy(f, os.path.join(dst, f"{tag}_queue_kernel_stats.csv"))
line = os.path.join(src, "bench_line.json")
if os.path.exists(line):
    shutil.copy(line, os.path.join(dst, f"{tag}_bench_line.json"))
for name in ("bench_line_b512.json", "bench_line_b1.json", "bench_line_b1024.json", "bench_line_b2048.json", "bench_line_driver_flags.json"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}"))
for name in ("stats_models.log", "stats_dt05.log", "stats_dt05ext.log", "stats_c4.log", "stats_gw.log", "stats_gw_fused.log", "gw_fused_pm.log",
             "gw_fused_2048.log", "gw_fused_dm_pm.log", "gw_fused_tm.log", "gw_fused_tm_dm_pm.log", "pmc_gw.log", "pmc_c4.log", "pmc_models.log", "stats_queue.log", "stats_at2017gfo.log", "small_batch.log", "perf_table.log", "device_walk.log"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        # (keep the result lines, not rocprofv3's chatter)
        keep = [ln for ln in open(f, errors="replace") if not ln.startswith(("W2026", "E2026", "I2026")) and "amdgpu.ids" not in ln]
        open(os.path.join(dst, f"{tag}_{name}"), "w").writelines(keep)
print(json.dumps({k: v["mean"] for k, v in pmc.items() if isinstance(v, dict)}, indent=1))

# One table of what the documents quote (DESIGN.md / README.md cite the averages of the committed CSVs): every em_* / stack2 / walk kernel
# of every profiled command with its call count, average, minimum -- profiles/<tag>_summary.md, generated, never edited.
rows = []
for what in ("bench", "dt05", "dt05ext", "c4", "models", "at2017gfo", "device_walk", "queue"):
    f = os.path.join(dst, f"{tag}_{what}_kernel_stats.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if not any(k in name for k in ("em_logl", "em_fused", "em_lc_loglike", "stack2", "lc_stack", "me2017", "walk_", "lc_regrid")):
            continue
        short = name.replace("void ", "").replace("nmma::", "")
        short = short[:short.index("(")] if "(" in short else short
        rows.append((what, short, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
if rows:
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as fh:
        fh.write(f"# {tag}: kernel times of the committed rocprofv3 CSVs (generated by tools/summarize_profiles.py)\n\n"
                 "`what` = the profiled command of tools/profile_round.sh (bench: bench.py --steps 200, steady state; the others: 35-call runs of a "
                 "fresh process, i.e. at ramping clocks -- config 2's own kernel reads ~30.8 us there).\n\n"
                 "| what | kernel | calls | average us | min us |\n|---|---|---|---|---|\n")
        for what, short, calls, avg, mn in rows:
            fh.write(f"| {what} | `{short}` | {calls} | {avg:.2f} | {mn:.2f} |\n")
    print("wrote", f"{tag}_summary.md", len(rows), "rows")
