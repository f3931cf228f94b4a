#!/bin/bash
# Runs ON THE GPU BOX: SQ instruction-mix / busy counters of em_logl on BASELINE config 4's shape (12 filters x 200 epochs, NP = 6)
# at 8192 rows -- per dispatch, divided by the 1024 SIMDs of the chip.
o=gpurun_out/${1:-x}/pmc_c4
export TMPDIR=/tmp
rm -rf $o; mkdir -p $o
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/a -- python3 tools/perf_case.py c4_shape 8192 > $o/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $o/b -- python3 tools/perf_case.py c4_shape 8192 > $o/b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $o/c -- python3 tools/perf_case.py c4_shape 8192 > $o/c.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $o/d -- python3 tools/perf_case.py c4_shape 8192 > $o/d.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in "abcd":
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "em_logl" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            v = v[3:] if len(v) > 6 else v
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}  per SIMD={sum(v)/len(v)/1024:12.1f}")
PY
