import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests import cases
from tests.helpers import engine_from_case
from nmma_amd import synthetic as syn
for ring in ("4", "3", "2"):
    os.environ["NMMA_EM_RING"] = ring
    case = cases.CASES["c4_syserr"]()
    eng = engine_from_case(case)
    _, th = syn.draw_theta(5, 4096, case["names"])
    t = torch.as_tensor(th, device="cuda:0")
    out = torch.empty(4096, dtype=torch.float64, device="cuda:0")
    for _ in range(50): eng.loglike(t, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): eng.loglike(t, out=out)
    e1.record(); torch.cuda.synchronize()
    print("ring", ring, "%.2f us per launch" % (e0.elapsed_time(e1) * 1e3 / 200), eng.last_launch_geometry())
    eng.close()
