#!/usr/bin/env python
"""Averaged bands: the lean task (em_logl<.., 5>) against the generic item phase (NMMA_EM_NO_LEAN_AVG=1), HIP events.
Usage (GPU box): python tools/perf_avg.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from tests.helpers import engine_from_case  # noqa: E402
from tests.test_gpu_parity import _averaging_variant  # noqa: E402
from tools.perf_table import timed  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for variant in ("plain", "cli_grid", "log_grid", "em_syserr", "time_nodes", "extinction", "p92", "many_points"):
    case = _averaging_variant(variant)
    th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
    out = torch.empty(B, dtype=torch.float64, device="cuda:0")
    res = []
    for env in (None, "1"):
        if env:
            os.environ["NMMA_EM_NO_LEAN_AVG"] = env
        else:
            os.environ.pop("NMMA_EM_NO_LEAN_AVG", None)
        eng = engine_from_case(case)
        us = timed(lambda: eng.loglike(th, out=out))
        eng.check()
        res.append((us, eng.last_launch_geometry()["block"]))
        eng.close()
    os.environ.pop("NMMA_EM_NO_LEAN_AVG", None)
    print(f"averaging/{variant:12s} B={B}: lean {res[0][0]:7.1f} us (block {res[0][1]})   generic {res[1][0]:7.1f} us (block {res[1][1]})")
