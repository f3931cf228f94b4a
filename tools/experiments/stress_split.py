"""Stress the band split's in-kernel combine (release / acquire across workgroups on any XCD, per-tile arrival counters that re-arm
themselves): thousands of launches of varying size against the one-workgroup-per-tile results, bit for bit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmma_amd import synthetic as syn
from nmma_amd.engine import EMEngine
case = syn.config2_case()
th = torch.as_tensor(syn.draw_theta(7, 1024, case["names"])[1], device="cuda:0")
os.environ["NMMA_EM_SPLIT"] = "0"
ref = EMEngine.from_case(case).loglike(th).cpu().numpy()
os.environ["NMMA_EM_SPLIT"] = "1"
eng = EMEngine.from_case(case)
rng = np.random.default_rng(0)
bad = 0
outs = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3000):
    n = int(rng.integers(1, 1025))
    lo = int(rng.integers(0, 1025 - n))
    outs.append((lo, n, eng.loglike(th[lo:lo + n])))
    if len(outs) == 64:
        for lo, n, o in outs:
            bad += int(not np.array_equal(o.cpu().numpy(), ref[lo:lo + n]))
        outs = []
print("launches with a wrong bit:", bad)
