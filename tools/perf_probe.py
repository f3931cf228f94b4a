#!/usr/bin/env python
"""Within-process timing of the three entry points per launch geometry (torch events on
the launch stream).  coefficients() = phase A only (MLP), lightcurves() = A + reconstruction,
loglike() = everything + combine.  Usage: python tools/perf_probe.py [B] [tiles...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import engine_from_case  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tiles = sys.argv[2:] or ["1", "2"]
case = cases.case_c2_default()
names, theta = syn.draw_theta(7, B, case["names"])
th = torch.as_tensor(theta, device="cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


for rnd in range(2):
    for tile in tiles:
        os.environ["NMMA_EM_TILE"] = tile
        eng = engine_from_case(case)
        out = torch.empty(B, dtype=torch.float64, device="cuda:0")
        t_l = timeit(lambda: eng.loglike(th, out=out))
        g = eng.last_launch_geometry()
        t_c = timeit(lambda: eng.coefficients(th))
        t_lc = timeit(lambda: eng.lightcurves(th))
        th_host = np.ascontiguousarray(theta)
        import time as _t
        eng.loglike(th_host)
        t0 = _t.perf_counter()
        for _ in range(20):
            eng.loglike(th_host)
        t_h = (_t.perf_counter() - t0) / 20 * 1e6
        fl = eng.flops_per_eval * B
        print(f"round {rnd} tile {tile:4s} B={B}: loglike {t_l:7.1f} us  coeff(phaseA) {t_c:7.1f} us  lightcurves {t_lc:7.1f} us"
              f"  | grid {g['grid_x']}x{g['grid_y']}x{g['block']} lds {g['lds_bytes']}"
              f"  | loglike {fl / t_l / 1e6:6.1f} TF/s  phaseA {fl / t_c / 1e6:6.1f} TF/s"
              f"  | host-buffer call {t_h:7.1f} us = {B / t_h:6.2f} Mevals/s PCIe-inclusive")
        eng.close()
