#!/bin/bash
# Runs ON THE GPU BOX: SQ instruction-mix / busy counters of gw_logl_kernel at a reduced config-5 shape (B = 2048).
o=gpurun_out/${1:-x}/pmc_gw
export TMPDIR=/tmp
rm -rf $o; mkdir -p $o
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --kernel-trace --output-format csv -d $o/a -- python3 tools/perf_gw_fused.py --batch 2048 --reps 2 > $o/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $o/b -- python3 tools/perf_gw_fused.py --batch 2048 --reps 2 > $o/b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $o/c -- python3 tools/perf_gw_fused.py --batch 2048 --reps 2 > $o/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in "abc":
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "gw_logl" in row.get("Kernel_Name", "") and "finish" not in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}  per bin-sample={sum(v)/len(v)*64/(2048*259585):10.3f} (x64 lanes)")
PY
