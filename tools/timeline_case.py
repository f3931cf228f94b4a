#!/usr/bin/env python
"""In-kernel timeline of workgroup 0 for any named case (cycles relative to the first stamp; nmma_em_debug_timeline): the MFMA role's
per-item intervals, the likelihood role's prologue and the first task of every item.  Usage: timeline_case.py <case> [B]"""
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
for _ in range(3):
    eng.loglike(th)
st = eng.debug_timeline(th)
t0 = min(x for x in st if x > 1000)
W = len(case["observed_filters"])
print(f"{name} B={B}: geometry {eng.last_launch_geometry()}")
print("MFMA role (wave 0): item  start  end  dur   [shader clocks, 100 MHz timer x ...: see clock64]")
for k in range(min(W, 32)):
    print(f"   A({k:2d}) {st[2*k]-t0:8d} {st[2*k+1]-t0:8d}  {st[2*k+1]-st[2*k]:7d}")
print("likelihood role: prologue", st[64]-t0, st[65]-t0, st[65]-st[64])
for k in range(1, min(W, 15) + 1):
    print(f"   Q({k-1:2d}) {st[64+2*k]-t0:8d} {st[64+2*k+1]-t0:8d}  {st[64+2*k+1]-st[64+2*k]:7d}   (stage Q of the item's first task)")
print("tasks: index wave claim -> done (cycles rel. to t0)")
for i in range(24):
    if st[16 + i] > 0:
        print(f"   task {i:2d} wave {st[40+i]:2d}  claim {st[16+i]-t0:7d}  done {st[104+i]-t0:7d}  dur {st[104+i]-st[16+i]:6d}")
eng.close()
