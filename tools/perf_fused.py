#!/usr/bin/env python
"""em_fused outputs of BASELINE config 2 at B rows: coefficients / detector-frame light curves (HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import engine_from_case  # noqa: E402
from tools.perf_table import timed  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name in ("c2_default", "c2_dt05"):
    case = cases.CASES[name]()
    eng = engine_from_case(case)
    th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
    print(f"{name} B={B}: coefficients {timed(lambda: eng.coefficients(th)):7.1f} us   light curves {timed(lambda: eng.lightcurves(th)):7.1f} us   "
          f"model curves {timed(lambda: eng.model_lightcurves(th)):7.1f} us")
    eng.close()
