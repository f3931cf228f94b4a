#!/bin/bash
# Runs ON THE GPU BOX: kernel time of nmma_lc_regrid at config 3's shape on own grids (rocprofv3 average over tools/perf_owngrids.py's calls)
export TMPDIR=/tmp
rm -rf /tmp/pv; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python3 tools/perf_owngrids.py > /tmp/pv.log 2>&1
f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1); echo "regrid: $(grep -i regrid $f | cut -d, -f2-4)"; grep "max rel" /tmp/pv.log
