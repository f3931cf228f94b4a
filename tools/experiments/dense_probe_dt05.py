"""Dense lean task on the documented CLI grid (--tmin .1 --tmax 20 --dt .5: 41 sample nodes) against the two-stage row form."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import engine_from_case
counts = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; shape = sys.argv[3] if len(sys.argv) > 3 else "c2"
if shape == "c4":
    filters = [f"band{i:02d}" for i in range(12)]
    names = ["luminosity_distance", "inclination_EM", "timeshift", "log10_mej_dyn", "vej_dyn", "Yedyn", "log10_mej_wind", "vej_wind"]
    case = cases._base(seed=7234, model="Bu2022Ye", filters=filters, counts=counts, batch=16, names=names, upper_limit_filter="band03")
else:
    case = cases._base(seed=5234, counts=counts, batch=16)
case["sample_times"] = np.geomspace(0.2, 20.0, 150) if os.environ.get("PROBE_LOG_GRID") else np.arange(0.1, 20.5, 0.5)
for mode in ("1", ""):
    if mode: os.environ["NMMA_EM_NO_DENSE"] = mode
    else: os.environ.pop("NMMA_EM_NO_DENSE", None)
    eng = engine_from_case(case)
    th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
    out = torch.empty(B, dtype=torch.float64, device="cuda:0")
    for _ in range(5): eng.loglike(th, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): eng.loglike(th, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{shape} dt05 counts {counts} B {B} {'row form' if mode else 'default '}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us  {eng.last_launch_geometry()}", flush=True)
    eng.close()
