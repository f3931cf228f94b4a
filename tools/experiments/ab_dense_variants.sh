#!/bin/bash
# Runs ON THE GPU BOX: config 4's shape on the library against build_dbg/lib_plainonly.so (tools/build_unit_variant.sh em_logl_f6 plainonly
# -DNMMA_DBG_DENSE_PLAIN_ONLY: the constant-systematics, equally-spaced variant of the dense task alone in its kernel), alternating
for i in 1 2 3; do
for lib in "" build_dbg/lib_plainonly.so; do
echo "lib '$lib': $(NMMA_HIP_LIB=$lib python3 tools/perf_case.py c4_shape 8192 2>&1 | grep 'us per launch' | cut -c1-40) | $(NMMA_HIP_LIB=$lib python3 tools/perf_case.py c4_shape 65536 2>&1 | grep 'us per launch' | cut -c1-42)"
done; done
