#!/bin/bash
# Runs ON THE GPU BOX: what the sampled-systematic variants' code costs the constant-systematics variants inside one kernel --
# config 3 (em_logl<.., 7>; tools/perf_models.py) on the library against build_dbg/lib_f7nosys.so, the CLI grid with extinction (em_logl<.., 3>
# with constant systematics: extinction_p92) against build_dbg/lib_f3nosys.so (tools/build_unit_variant.sh <unit> <name> -DNMMA_DBG_NO_SYS_VARIANTS)
for i in 1 2 3; do
for lib in "" build_dbg/lib_f7nosys.so; do
echo "lib '$lib': $(NMMA_HIP_LIB=$lib python3 tools/perf_models.py 2>&1 | grep 'em_logl<.., 7> alone' | cut -c1-80)"
done
for lib in "" build_dbg/lib_f3nosys.so; do
echo "lib '$lib': $(NMMA_HIP_LIB=$lib python3 tools/perf_case.py extinction_p92 4096 2>&1 | grep 'us per launch' | cut -c1-50)"
done; done
