#!/bin/bash
# Runs ON THE GPU BOX: kernel time of nmma_lc_regrid at config 3's shape on own grids (rocprofv3 average over tools/perf_owngrids.py's calls),
# the packed kernel (several curves per wave) and, with NMMA_REGRID_NO_PACK=1, one curve per wave
export TMPDIR=/tmp
for np in 0 1; do
rm -rf /tmp/pv
if [ $np = 1 ]; then export NMMA_REGRID_NO_PACK=1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python3 tools/perf_owngrids.py > /tmp/pv.log 2>&1
f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1); echo "no_pack=$np: $(grep -i regrid $f | cut -d, -f1-4)"; grep "max rel\|one launch:\|materialising" /tmp/pv.log
done
