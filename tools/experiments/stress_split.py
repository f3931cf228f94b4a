"""Stress the band split's in-kernel combine (release / acquire across workgroups on any XCD, per-tile arrival counters that re-arm
themselves): thousands of launches of varying size against the one-workgroup-per-tile results, bit for bit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmma_amd import synthetic as syn
from nmma_amd.engine import EMEngine
case = syn.config2_case()
NMAX = 2304        # (auto mode: groups of one band up to 672 rows, of two up to 1360, of three up to 2048, none beyond)
th = torch.as_tensor(syn.draw_theta(7, NMAX, case["names"])[1], device="cuda:0")
os.environ["NMMA_EM_SPLIT"] = "0"
ref = EMEngine.from_case(case).loglike(th).cpu().numpy()
mode = sys.argv[2] if len(sys.argv) > 2 else "auto"      # "1": one band per workgroup whatever the size; "auto": the library's choice
if mode == "1":
    os.environ["NMMA_EM_SPLIT"] = "1"
else:
    os.environ.pop("NMMA_EM_SPLIT")
eng = EMEngine.from_case(case)
rng = np.random.default_rng(0)
bad = 0
outs = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3000):
    n = int(rng.integers(1, NMAX + 1))
    lo = int(rng.integers(0, NMAX + 1 - n))
    outs.append((lo, n, eng.loglike(th[lo:lo + n])))
    if len(outs) == 64:
        for lo, n, o in outs:
            bad += int(not np.array_equal(o.cpu().numpy(), ref[lo:lo + n]))
        outs = []
print(f"mode {mode}: launches with a wrong bit:", bad)
