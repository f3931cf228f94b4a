"""How much does the depth of the LDS item ring matter for a config-4-like shape?  (12 filters x N epochs, NP = 6, 8192 rows;
N = 100 leaves room for 2-3 ring slots, N = 200 -- BASELINE config 4 -- only for one.)  NMMA_EM_RING=<n> bounds the depth."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
counts = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
filters = [f"band{i:02d}" for i in range(12)]
names = ["luminosity_distance", "inclination_EM", "timeshift", "log10_mej_dyn", "vej_dyn", "Yedyn", "log10_mej_wind", "vej_wind"]
case = cases._base(seed=7234, model="Bu2022Ye", filters=filters, counts=counts, batch=16, names=names, upper_limit_filter="band03")
for ring in ("1", "2", "3"):
    os.environ["NMMA_EM_RING"] = ring
    eng = engine_from_case(case)
    th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
    out = torch.empty(B, dtype=torch.float64, device="cuda:0")
    for _ in range(5): eng.loglike(th, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): eng.loglike(th, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"counts {counts} ring<= {ring}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us  {eng.last_launch_geometry()}", flush=True)
    eng.close()
