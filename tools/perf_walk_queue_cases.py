#!/usr/bin/env python
"""Engine-level queue (nmma_em_walk_queue) of 1024 / 4096 / 8192 / 16384 chains (NMMA_PERF_CHAINS) x 100 steps for golden-case configurations: the fused MCMC step
against two launches per step; with ``+con`` after a case name also under a Constraint program (its interpreter in the fused step).
Usage: perf_walk_queue_cases.py case[,case...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import sampler as smp  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import UniformPrior, engine_from_case  # noqa: E402

rng = np.random.default_rng(5)
def _per_filter_syserr_case():
    """One sampled systematic PER observed filter: 6 + 9 = 15 sampled dimensions (tests/test_gpu_walk_queue.py)."""
    from nmma_amd import synthetic as syn
    filters = syn.AT2017GFO_FILTERS
    sys_names = [f"em_syserr_{k}" for k in range(len(filters))]
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind"] + sys_names
    c = cases._base(seed=5234, names=names, batch=48)
    c["systematics"] = dict(mode="mixed", names=dict(zip(filters, sys_names)), nodes={})
    return c


DEFAULT = ["c2_default", "c2_default+con", "c2_dt05_limit", "averaging", "log_grid", "syserr_param", "syserr_param+con", "c4_shape", "c4_syserr",
           "syserr_per_filter"]
for spec in (sys.argv[1].split(",") if len(sys.argv) > 1 else DEFAULT):
    name, _, flag = spec.partition("+")
    case = _per_filter_syserr_case() if name == "syserr_per_filter" else (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
    eng = engine_from_case(case)
    if os.environ.get("NMMA_PERF_LANES"):          # (16: the fused step with 16 lanes per chain whatever the number of sampled dimensions)
        eng.set_option("walk_lanes", int(os.environ["NMMA_PERF_LANES"]))
    names = case["names"]
    th = case["theta"]
    lo, hi = th.min(axis=0) - 1e-3, th.max(axis=0) + 1e-3
    pri = {k: UniformPrior(float(a), float(b)) for k, a, b in zip(names, lo, hi)}
    pt = smp.BatchedPriorTransform(pri, names)
    w = smp.EnsembleWalkSampler(ndim=len(names), walks=100)
    table = smp.device_prior_table(pri, names, w.periodic, w.reflective)
    con = None
    if flag == "con":       # 10 ** log10_mej_dyn + 10 ** log10_mej_wind < bound, timeshift > bound: two pow, an add, three checks
        from nmma_amd import _lib as L
        from nmma_amd.core.constraints import ConstraintProgram
        c4, c5 = names.index("log10_mej_dyn"), names.index("log10_mej_wind")
        mid = float(np.median(10 ** th[:, c4] + 10 ** th[:, c5]))
        con = ConstraintProgram([(L.CON_PUSH_CONST, 0, 10.0), (L.CON_PUSH_COL, c4, 0.0), (L.CON_POW, 0, 0.0), (L.CON_PUSH_CONST, 0, 10.0),
                                 (L.CON_PUSH_COL, c5, 0.0), (L.CON_POW, 0, 0.0), (L.CON_ADD, 0, 0.0), (L.CON_CHECK_LT, 0, 1.5 * mid),
                                 (L.CON_PUSH_COL, 3, 0.0), (L.CON_CHECK_GT, 0, float(np.quantile(th[:, 3], 0.02))), (L.CON_CHECK_LT, 0, 1e300)],
                                len(names), 0)
    for n in [int(x) for x in os.environ.get("NMMA_PERF_CHAINS", "1024,4096,8192,16384").split(",")]:
        live = rng.uniform(0.2, 0.8, (n, len(names)))
        u0 = live.copy()
        bound = np.full(n, np.quantile(eng.loglike(np.ascontiguousarray(pt(live))), 0.2))
        keys = rng.integers(1, 2 ** 62, n).astype(np.uint64)
        out = []
        for fuse in (2, 0):          # (2: the fused step wherever an instantiation exists, also where the library prefers two launches)
            eng.set_option("walk_fuse", fuse)
            eng.walk_queue(table, live, u0, bound, keys, 100, constraints=con)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                eng.walk_queue(table, live, u0, bound, keys, 100, constraints=con)
                ts.append(time.perf_counter() - t0)
            out.append((1e3 * float(np.median(ts)), eng.last_walk_gpu_ms))
        eng.set_option("walk_fuse", 1)
        print(f"{spec:20s} {n:5d} chains x 100 steps: fused {out[0][0]:6.3f} ms ({out[0][1]:6.3f} device), two launches {out[1][0]:6.3f} ms ({out[1][1]:6.3f} device)", flush=True)
    eng.close()
