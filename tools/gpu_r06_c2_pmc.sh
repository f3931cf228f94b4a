#!/bin/bash
# Round 6, on the GPU box: SQ counters of the headline kernel (config 2, 4096 rows) for the library and for a build of its unit whose tasks
# only wait and signal (tools/build_unit_variant.sh em_logl_f1 novalu1 -DNMMA_DBG_NOVALU): what the likelihood tasks cost beside the surrogate
export TMPDIR=/tmp
for lib in "" build_dbg/lib_sleep64.so build_dbg/lib_nv_norelu.so build_dbg/lib_nv_norelu_s64.so; do
  o=gpurun_out/r06_c2pmc/$(basename ${lib:-default} .so); rm -rf $o; mkdir -p $o
  export NMMA_HIP_LIB=$lib
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $o/a -- python3 tools/perf_case.py c2_default 4096 > $o/a.log 2>&1
  echo "== ${lib:-default}: $(grep 'us per launch' $o/a.log | cut -c1-50)"
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$o/a/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "em_logl" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v = v[3:] if len(v) > 6 else v
        print(f"   {k:24s} per SIMD={sum(v)/len(v)/1024:12.1f}")
PY
done
