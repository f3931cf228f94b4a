import numpy as np, torch, sys, os
sys.path.insert(0, '.')
from tests import cases
from tests.helpers import engine_from_case
from oracle import nmma_oracle as orc
case = cases.case_c4_shape()
th = torch.as_tensor(case["theta"], device="cuda:0")
p = orc.model_parameter_conversion(dict(zip(case["names"], case["theta"].T)), case["model_parameters"])
plist = np.stack([np.broadcast_to(p[k], (len(case["theta"]),)) for k in case["model_parameters"]], 1)
def run(tag):
    eng = engine_from_case(case)
    c = eng.coefficients(th).cpu().numpy()
    out = []
    for k, f in enumerate(case["model_filters"][:3]):
        t = case["svd"][f]
        x = (plist - t["param_mins"]) / (t["param_maxs"] - t["param_mins"])
        ideal = orc.mlp_forward(x, t["W1"], t["b1"], t["W2"], t["b2"], "f64acc")
        out.append(np.abs(c[:, k] - ideal).max(axis=1))
    print(tag, "per-sample max abs diff, filter0:", np.array2string(out[0], precision=2), "filter1:", np.array2string(out[1][:4], precision=2))
    eng.close()
for tile in ("1,8", "1,4", "2,4", "4,4", "2,8"):
    os.environ["NMMA_EM_TILE"] = tile
    run(tile)
