#!/bin/bash
# Round 6, on the GPU box: where config 4's vector instructions go -- SQ counters of em_logl<2,2,8,8,6> at 8192 rows for the library and for
# development variants of its unit (build_dbg/lib_<name>.so: tasks removed = everything but the tasks; longer poll sleeps)
export TMPDIR=/tmp
for lib in "" build_dbg/lib_novalu.so build_dbg/lib_sleep32.so; do
  o=gpurun_out/r06_c4pmc/$(basename ${lib:-default} .so)
  rm -rf $o; mkdir -p $o
  export NMMA_HIP_LIB=$lib
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $o/a -- python3 tools/perf_case.py c4_shape 8192 > $o/a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $o/b -- python3 tools/perf_case.py c4_shape 8192 > $o/b.log 2>&1
  echo "== ${lib:-default}: $(tail -1 $o/a.log | cut -c1-60)"
  python3 - <<PY
import csv, glob, collections
for sub in "ab":
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "em_logl" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            v = v[3:] if len(v) > 6 else v
            print(f"   {k:24s} per SIMD={sum(v)/len(v)/1024:12.1f}")
PY
done
