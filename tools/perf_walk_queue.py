#!/usr/bin/env python
"""One queue of the nested sampler (GPUPool.map(walker.sample, queue), 100 MCMC steps) for several queue sizes, with the MCMC step
fused into the likelihood launch and as two launches (NMMA_WALK_NO_FUSE=1: the likelihood may then split small batches by band)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from nmma_amd import synthetic as syn  # noqa: E402

case = syn.config2_case()
for n in (256, 512, 1024, 2048, 4096):
    row = []
    for nofuse in (False, True):
        if nofuse:
            os.environ["NMMA_WALK_NO_FUSE"] = "1"
        else:
            os.environ.pop("NMMA_WALK_NO_FUSE", None)
        r = bench.device_walk_queue(case, syn, 1.0, n=n, walks=100, repeats=5)
        row.append((r["map_ms"], r["device_ms"]))
    print(f"queue of {n:5d} records x 100 steps: fused {row[0][0]:6.3f} ms wall ({row[0][1]:6.3f} device), two launches {row[1][0]:6.3f} ms wall ({row[1][1]:6.3f} device)")
