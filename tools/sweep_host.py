import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
case = cases.case_c2_default()
eng = engine_from_case(case)
_, theta = syn.draw_theta(5, 8192, case["names"])
for B in (1, 16, 17, 32, 64, 65, 128, 256, 1024, 4096, 8192, 64, 1):
    x = np.ascontiguousarray(theta[:B])
    for _ in range(5): eng.loglike(x)
    t0 = time.perf_counter()
    for _ in range(100): eng.loglike(x)
    dt = (time.perf_counter() - t0) / 100
    xt = torch.as_tensor(x, device="cuda:0"); out = torch.empty(B, dtype=torch.float64, device="cuda:0")
    for _ in range(5): eng.loglike(xt, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): eng.loglike(xt, out=out)
    torch.cuda.synchronize(); dd = (time.perf_counter() - t0) / 100
    print(f"B={B:5d}: host call {dt*1e6:7.1f} us   device call {dd*1e6:7.1f} us")
