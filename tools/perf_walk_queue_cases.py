#!/usr/bin/env python
"""Engine-level queue (nmma_em_walk_queue) of 4096 chains x 100 steps for golden-case configurations: the fused MCMC step against
two launches per step.  Usage: perf_walk_queue_cases.py case[,case...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import sampler as smp  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import UniformPrior, engine_from_case  # noqa: E402

rng = np.random.default_rng(5)
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2_default", "c2_dt05_limit", "averaging", "log_grid", "syserr_param", "c4_shape", "c4_syserr"]):
    case = (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
    eng = engine_from_case(case)
    names = case["names"]
    th = case["theta"]
    lo, hi = th.min(axis=0) - 1e-3, th.max(axis=0) + 1e-3
    pri = {k: UniformPrior(float(a), float(b)) for k, a, b in zip(names, lo, hi)}
    pt = smp.BatchedPriorTransform(pri, names)
    w = smp.EnsembleWalkSampler(ndim=len(names), walks=100)
    table = smp.device_prior_table(pri, names, w.periodic, w.reflective)
    for n in (1024, 4096):
        live = rng.uniform(0.2, 0.8, (n, len(names)))
        u0 = live.copy()
        bound = np.full(n, np.quantile(eng.loglike(np.ascontiguousarray(pt(live))), 0.2))
        keys = rng.integers(1, 2 ** 62, n).astype(np.uint64)
        out = []
        for nofuse in (False, True):
            if nofuse:
                os.environ["NMMA_WALK_NO_FUSE"] = "1"
            else:
                os.environ.pop("NMMA_WALK_NO_FUSE", None)
            eng.walk_queue(table, live, u0, bound, keys, 100)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                eng.walk_queue(table, live, u0, bound, keys, 100)
                ts.append(time.perf_counter() - t0)
            out.append((1e3 * float(np.median(ts)), eng.last_walk_gpu_ms))
        os.environ.pop("NMMA_WALK_NO_FUSE", None)
        print(f"{name:16s} {n:5d} chains x 100 steps: fused {out[0][0]:6.3f} ms ({out[0][1]:6.3f} device), two launches {out[1][0]:6.3f} ms ({out[1][1]:6.3f} device)", flush=True)
    eng.close()
