#!/usr/bin/env python
"""Run one entry point N times (for rocprofv3).  Usage: run_mode.py {loglike|coeff|lc} [B] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import engine_from_case  # noqa: E402

mode = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
case = cases.case_c2_default()
eng = engine_from_case(case)
th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
fn = {"loglike": eng.loglike, "coeff": eng.coefficients, "lc": eng.lightcurves}[mode]
for _ in range(N):
    fn(th)
torch.cuda.synchronize()
eng.close()
