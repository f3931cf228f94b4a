#!/usr/bin/env python
"""In-kernel timeline of em_logl_iw, workgroup 0: per wave the shader-clock stamps
0 entry | 1 after the barrier | 2 prologue done | 3+2k stream k starts | 4+2k stream k ends | 15 before the store."""
import os, sys
os.environ.setdefault("NMMA_EM_IW", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
case = cases._base()
eng = engine_from_case(case)
th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
for _ in range(3):
    eng.loglike(th)
raw = eng.debug_timeline(th)
st = raw[:128].reshape(8, 16)
t0 = st[:, 0].min()
print("wave  entry barrier prolog |" + "".join(f"  s{k}:start   end" for k in range(6)) + " |  final")
for w in range(8):
    r = st[w] - t0
    print(f"{w:4d} {r[0]:6d} {r[1]:7d} {r[2]:6d} |" + "".join(f" {r[3+2*k]:8d} {r[4+2*k]:6d}" for k in range(6)) + f" | {r[15]:6d}")
eng.close()

for w, base in ((0, 128), (4, 160)):
    ts = raw[base:base + 17]
    if ts.any():
        d = np.diff(ts.astype(np.int64)) & 0xffffffff
        print(f"wave {w}, stream 3: ticks per record step:", " ".join(str(int(x)) for x in d), "| total", int(d.sum()))
