#!/usr/bin/env python
"""Print the in-kernel timeline of workgroup 0 (cycles relative to the first stamp)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NH = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
case = cases._base(n_hidden=NH)
eng = engine_from_case(case)
th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
for _ in range(3):
    eng.loglike(th)
st = eng.debug_timeline(th)
t0 = min(x for x in st if x > 1000)
W = 6
print("MFMA role: item  start  end  dur")
for k in range(W):
    print(f"   A({k}) {st[2*k]-t0:8d} {st[2*k+1]-t0:8d}  {st[2*k+1]-st[2*k]:7d}")
print("likelihood role: prologue", st[64]-t0, st[65]-t0, st[65]-st[64])
for k in range(1, W + 1):
    print(f"   Q({k-1}) {st[64+2*k]-t0:8d} {st[64+2*k+1]-t0:8d}  {st[64+2*k+1]-st[64+2*k]:7d}   (stage Q of the item's first task)")
eng.close()

print("last item, task c=0: P", st[97]-st[96], "| wait", st[98]-st[97], "| coef", st[99]-st[98], "| rows+FMA+term", st[100]-st[99], "| group_sum", st[101]-st[100])

print("tasks: index wave claim -> done (cycles rel. to t0)")
for i in range(24):
    if st[16 + i] > 0:
        print(f"   task {i:2d} (item {i//4}) wave {st[40+i]:2d}  claim {st[16+i]-t0:7d}  done {st[104+i]-t0:7d}  dur {st[104+i]-st[16+i]:6d}")

if any(st[128 + 8 * i] > 0 for i in range(24)):       # a -DNMMA_DBG_TASKSTAMPS build: stage stamps per (item, chunk)
    print("stage stamps (item, chunk): wave | entry | +staging issued | +stage P | +rows landed (c = 0) | rows seen | coefficients seen | end")
    for i in range(24):
        b = 128 + 8 * i
        if st[b] <= 0:
            continue
        rel = lambda j: (st[b + j] - t0) if st[b + j] > 0 else -1
        print(f"   ({i // 4}, {i % 4}) wave {st[b + 7]:2d}  entry {rel(0):6d}  stg {rel(1):6d}  P {rel(2):6d}  landed {rel(3):6d}  rows {rel(4):6d}  coef {rel(5):6d}  end {rel(6):6d}")
