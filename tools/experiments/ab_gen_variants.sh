#!/bin/bash
# Runs ON THE GPU BOX: the general lean task (em_logl<.., 5>) on the library against build_dbg/lib_f5plain.so (tools/build_unit_variant.sh em_logl_f5
# f5plain -DNMMA_DBG_GEN_PLAIN_ONLY: its constant-systematics, equally-spaced variants alone in the kernel) -- the CLI grid with finite limits and the
# averaged bands, steady state
export NMMA_PERF_WARM=2000 NMMA_PERF_N=400
for i in 1 2 3; do
for lib in "" build_dbg/lib_f5plain.so; do
echo "lib '$lib': $(NMMA_HIP_LIB=$lib python3 tools/perf_case.py c2_dt05_limit 4096 2>&1 | grep 'us per launch' | cut -c1-48) | $(NMMA_HIP_LIB=$lib python3 tools/perf_case.py averaging 4096 2>&1 | grep 'us per launch' | cut -c1-45)"
done; done
