import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
case = cases._base()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
os.environ["NMMA_EM_SPLIT"] = "0"
e0 = engine_from_case(case)
ref = e0.loglike(th).cpu().numpy()
os.environ["NMMA_EM_SPLIT"] = "1"
e1 = engine_from_case(case)
for i in range(3):
    got = e1.loglike(th).cpu().numpy()
    print(i, "equal", np.array_equal(got, ref), "ratio", (got / ref)[:8])
