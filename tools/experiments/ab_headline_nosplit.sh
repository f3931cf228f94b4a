#!/bin/bash
# Runs ON THE GPU BOX: the headline kernel (config 2, 4096 rows, steady state: 2000 warm-up launches) on the library against
# build_dbg/lib_f1nosplit.so (tools/build_unit_variant.sh em_logl_f1 f1nosplit -DNMMA_DBG_NO_SPLIT: without the band-split forms of the epilogue)
export NMMA_PERF_WARM=2000 NMMA_PERF_N=400
for i in 1 2 3; do
for lib in "" build_dbg/lib_f1nosplit.so; do
echo "lib '$lib': $(NMMA_HIP_LIB=$lib NMMA_EM_SPLIT=0 python3 tools/perf_case.py c2_default 4096 2>&1 | grep 'us per launch' | cut -c1-45)"
done; done
