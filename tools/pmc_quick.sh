#!/bin/bash
# Runs ON THE GPU BOX: SQ instruction-mix counters of em_logl for the default geometry (or env overrides).
o=gpurun_out/pmc_quick_${1:-x}
export TMPDIR=/tmp
rm -rf $o; mkdir -p $o
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $o/a -- python3 tools/run_mode.py loglike 4096 12 > $o/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/b -- python3 tools/run_mode.py loglike 4096 12 > $o/b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $o/c -- python3 tools/run_mode.py loglike 4096 12 > $o/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in "abc":
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "em_logl" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            v = v[2:] if len(v) > 4 else v
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):14.1f}  per SIMD={sum(v)/len(v)/1024:10.1f}")
PY
