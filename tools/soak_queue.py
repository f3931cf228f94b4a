#!/usr/bin/env python
"""Soak test of the sampler queue (nmma_em_walk_queue): random queue sizes, walk lengths (shared and per chain) and live sets; the
default form (MCMC step fused into the likelihood launch, split by band for small queues) must give the bits of the two-launch step
loop every time, and the hand-off watchdog must stay clean."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import sampler as smp  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import UniformPrior, engine_from_case  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 60
names_cases = sys.argv[2].split(",") if len(sys.argv) > 2 else ["c2_default", "syserr_param", "log_grid"]
rng = np.random.default_rng(5)
for name in names_cases:
    case = (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
    eng = engine_from_case(case)
    names = case["names"]
    th = case["theta"]
    lo, hi = th.min(axis=0) - 1e-3, th.max(axis=0) + 1e-3
    pri = {k: UniformPrior(float(a), float(b)) for k, a, b in zip(names, lo, hi)}
    pt = smp.BatchedPriorTransform(pri, names)
    w = smp.EnsembleWalkSampler(ndim=len(names), periodic=[1], reflective=[2], walks=7)
    table = smp.device_prior_table(pri, names, w.periodic, w.reflective)
    bad, t0, steps_total = 0, time.time(), 0
    for it in range(n_iter):
        n = int(rng.choice([1, 3, 16, 17, 100, 255, 256, 257, 1000, 2048, 2049, 3000, 4096]))
        n_live = int(rng.choice([3, 50, 500]))
        live = rng.uniform(0.2, 0.8, (n_live, len(names)))
        u0 = live[rng.integers(0, n_live, n)].copy()
        bound = np.full(n, np.quantile(eng.loglike(np.ascontiguousarray(pt(live))), rng.uniform(0.05, 0.9)))
        keys = rng.integers(1, 2 ** 62, n).astype(np.uint64)
        walks = int(rng.integers(1, 40))
        steps = walks if rng.uniform() < 0.5 else rng.integers(1, walks + 1, n).astype(np.int32)
        fused = eng.walk_queue(table, live, u0, bound, keys, steps)
        eng.set_option("walk_fuse", 0)
        try:
            two = eng.walk_queue(table, live, u0, bound, keys, steps)
        finally:
            eng.set_option("walk_fuse", 1)
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(fused, two))
        steps_total += n * walks
        if not same:
            bad += 1
            print(f"{name}: MISMATCH iter {it} n={n} n_live={n_live} walks={walks}", flush=True)
    eng.check()
    print(f"{name}: {n_iter} queues ({steps_total} chain steps), {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
    eng.close()
