#!/usr/bin/env python
"""Time nmma_em_loglike for any named case.  Usage: perf_case.py <case> [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
case = (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
eng = engine_from_case(case)
th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
out = torch.empty(B, dtype=torch.float64, device="cuda:0")
# (NMMA_PERF_WARM=<n>: n untimed launches first -- ~1600 bring the clocks to the steady state bench.py measures in; default: the 5 of a
#  fresh process, the protocol of the committed *_kernel_stats.csv files)
for _ in range(int(os.environ.get("NMMA_PERF_WARM", "5"))):
    eng.loglike(th, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(os.environ.get("NMMA_PERF_N", "30"))
e0.record()
for _ in range(n):
    eng.loglike(th, out=out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
eng.check()
print(f"{name} B={B}: {us:.1f} us per launch, {B / us:.2f} Mevals/s, {eng.flops_per_eval * B / us / 1e6:.1f} TF/s, geometry {eng.last_launch_geometry()}")
eng.close()
