import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
case = cases.case_c4_shape()
case = cases._base(seed=7234, model="Bu2022Ye", filters=[f"band{i:02d}" for i in range(12)], counts=200, batch=16,
                   names=case["names"] + ["em_syserr"], upper_limit_filter="band03")
case["systematics"] = dict(mode="param", name="em_syserr")
th = torch.as_tensor(syn.draw_theta(7, 8192, case["names"])[1], device="cuda:0")
out = torch.empty(8192, dtype=torch.float64, device="cuda:0")
for env in ({}, {"NMMA_EM_NO_DENSE": "1"}, {"NMMA_EM_NO_DENSE": "1", "NMMA_EM_NO_ITEM_DAT": "1"}):
    for k in ("NMMA_EM_NO_DENSE", "NMMA_EM_NO_ITEM_DAT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = engine_from_case(case)
    for _ in range(5): eng.loglike(th, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): eng.loglike(th, out=out)
    e1.record(); torch.cuda.synchronize()
    print(env, f"{e0.elapsed_time(e1) / 30 * 1e3:.1f} us", eng.last_launch_geometry(), flush=True)
    eng.close()
