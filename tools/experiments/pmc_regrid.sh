#!/bin/bash
# Runs ON THE GPU BOX: SQ counters + HBM traffic of lc_regrid_lds_kernel for tools/perf_owngrids.py (config 3's shape on own grids)
o=gpurun_out/pmc_regrid
export TMPDIR=/tmp
rm -rf $o; mkdir -p $o
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $o/a -- python3 tools/perf_owngrids.py > $o/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $o/b -- python3 tools/perf_owngrids.py > $o/b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $o/c -- python3 tools/perf_owngrids.py > $o/c.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $o/d -- python3 tools/perf_owngrids.py > $o/d.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/e -- python3 tools/perf_owngrids.py > $o/e.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/f -- python3 tools/perf_owngrids.py > $o/f.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in "abcdef":
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if "regrid" in name:
                acc[(name.split("(")[0][-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k[0]:30s} {k[1]:24s} n={len(v):3d} mean={sum(v)/len(v):14.1f}  per SIMD {sum(v)/len(v)/1024:10.1f}")
PY
