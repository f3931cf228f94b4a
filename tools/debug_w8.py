#!/usr/bin/env python
"""Debug helper: batch-size independence / determinism of em_logl (run with NMMA_EM_MFMA_WAVES=8 or 4)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
case = cases.case_c2_default()
_, theta = syn.draw_theta(4242, 4096, case["names"])
eng = engine_from_case(case)
th = torch.as_tensor(theta, device="cuda:0")
a = eng.loglike(th).cpu().numpy()
for n in (1000, 1000, 1008, 16, 4096, 4096):
    d = eng.loglike(th[:n]).cpu().numpy()
    bad = np.nonzero(d != a[:n])[0]
    print(n, "mismatches", bad.size, bad[:10], (d[bad[:5]] - a[bad[:5]]) if bad.size else "")
eng.close()
