#!/usr/bin/env python
"""Soak test of em_logl's hand-off protocol: many launches over varying batch sizes, every result compared
bit-for-bit with the first evaluation of the same rows; the watchdog must stay clean."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["c2_default", "fast_many_filters", "syserr_param"]
rng = np.random.default_rng(11)
for name in names:
    case = (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
    eng = engine_from_case(case)
    _, theta = syn.draw_theta(99, 20000, case["names"])
    th = torch.as_tensor(theta, device="cuda:0")
    ref = eng.loglike(th).cpu().numpy()
    eng.check()
    bad = 0
    t0 = time.time()
    for it in range(n_iter):
        B = int(rng.choice([1, 7, 16, 17, 33, 100, 1000, 4096, 4097, 8000, 20000]))
        lo = int(rng.integers(0, 20000 - B + 1))
        got = eng.loglike(th[lo:lo + B]).cpu().numpy()
        if not np.array_equal(got, ref[lo:lo + B]):
            bad += 1
            idx = np.nonzero(got != ref[lo:lo + B])[0]
            print(f"{name}: MISMATCH iter {it} B={B} lo={lo} rows {idx[:8]}", flush=True)
    eng.check()
    print(f"{name}: {n_iter} launches, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
    eng.close()
