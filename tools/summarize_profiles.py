#!/usr/bin/env python
"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>_*) into the tracked
summaries under profiles/: kernel stats CSV, PMC means, HBM traffic per launch, bench line."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))

pmc = {}
for sub in ("FETCH_SIZE", "WRITE_SIZE", "sq", "sq2"):
    for f in glob.glob(os.path.join(src, f"{tag}_pmc_{sub}", "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "em_logl" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            v = v[5:] if len(v) > 10 else v          # drop the warm-up launches
            pmc[k] = {"n_dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
pmc["_note"] = ("rocprofv3 --pmc (separate passes, --kernel-trace only) on `python3 bench.py --steps 20 --warmup 5 "
                "--no-cpu-baseline`; kernel em_logl, B=4096, per dispatch over the whole GPU (1024 SIMDs)")
json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc_em_logl_summary.json"), "w"), indent=1)

if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # FETCH_SIZE / WRITE_SIZE are in KiB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts
    # 64-B units for 128-B requests -> doubled
    fetch = pmc["FETCH_SIZE"]["mean"] * 1024 * 2
    write = pmc["WRITE_SIZE"]["mean"] * 1024
    traffic = {
        "bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 20 "
                  "--warmup 5`, kernel em_logl, B=4096; FETCH_SIZE in KiB doubled per MI355X_MICROARCH.md (HBM section)",
        # theta in + logL out + every weight/table byte once
        "algorithmic_bytes": 1088432,
    }
    json.dump(traffic, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)

line = os.path.join(src, f"{tag}_bench_line.json")
if os.path.exists(line):
    shutil.copy(line, os.path.join(dst, f"{tag}_bench_line.json"))
print(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv")).read()[:400] if stats else "no stats")
print(json.dumps({k: v["mean"] for k, v in pmc.items() if isinstance(v, dict)}, indent=1))
