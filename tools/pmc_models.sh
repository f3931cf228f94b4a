#!/bin/bash
# Runs ON THE GPU BOX: SQ instruction-mix counters of the likelihood-from-curves kernels (em_lc_loglike<G, NM, SD>) and lc_stack_kernel
# for tools/perf_models.py (config 3's shape at 8192 rows is the largest launch: the means below keep launches of >= 500 workgroups).
o=gpurun_out/pmc_models
export TMPDIR=/tmp
rm -rf $o; mkdir -p $o
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $o/a -- python3 tools/perf_models.py > $o/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_FLAT --kernel-trace --output-format csv -d $o/b -- python3 tools/perf_models.py > $o/b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $o/c -- python3 tools/perf_models.py > $o/c.log 2>&1
# (round 5: config 3 in one launch -- em_logl<.., 7>: its matrix-pipe share and its HBM traffic, FETCH_SIZE / WRITE_SIZE in passes of their own)
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/d -- python3 tools/perf_models.py > $o/d.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/e -- python3 tools/perf_models.py > $o/e.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/f -- python3 tools/perf_models.py > $o/f.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in "abcdef":
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if ("lc_loglike" in name or "lc_stack" in name or ("em_logl" in name and ", 7, 0>" in name) or "stack2_redo" in name or
                    ("em_fused<3" in name)) and int(row.get("Grid_Size", 0)) >= 250 * 256:
                acc[(name.split("(")[0][-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k[0]:42s} {k[1]:24s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
PY
