#!/usr/bin/env python
"""Where the log-likelihood error of the `bulla_svd` golden case (a surrogate with real conditioning) comes from: per row, the HIP
path against the reference golden (numpy fp32 MLP stand-in) and against the oracle with fp64-ACCUMULATED fp32 operands (the
order-independent "ideal fp32" value both fp32 implementations approximate), with the coefficient and light-curve errors of the
worst rows.  GPU box: python tools/experiments/bulla_diag.py [case]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nmma_oracle as orc  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import engine_from_case  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bulla_svd"
case = cases.CASES[name]()
gold = cases.load_golden(name)["logl"]
eng = engine_from_case(case)
th = torch.as_tensor(case["theta"], device="cuda:0")
got = eng.loglike(th).cpu().numpy()
print("launch", eng.last_launch_geometry())
ideal = orc.log_likelihood_batch(orc.likelihood_from_case(case, use_scipy=False, mlp_mode="f64acc"), case["names"], case["theta"], case.get("fixed"))
den = np.maximum(1.0, np.abs(ideal))
e_gold, e_ideal, e_ref = np.abs(got - gold) / den, np.abs(got - ideal) / den, np.abs(gold - ideal) / den
print(f"{name}: HIP vs golden(numpy f32) max {e_gold.max():.3e} | HIP vs ideal max {e_ideal.max():.3e} | golden vs ideal max {e_ref.max():.3e}")
c = eng.coefficients(th).cpu().numpy()
olik = orc.likelihood_from_case(case, use_scipy=False)
p = olik.model.parameter_conversion(dict(zip(case["names"], case["theta"].T)))
plist = np.stack([np.broadcast_to(p[k], (len(case["theta"]),)) for k in case["model_parameters"]], 1)
ce_h, ce_n = [], []
for k, f in enumerate(case["model_filters"]):
    t = case["svd"][f]
    x = (plist - t["param_mins"]) / (t["param_maxs"] - t["param_mins"])
    idl = orc.mlp_forward(x, t["W1"], t["b1"], t["W2"], t["b2"], "f64acc")
    n32 = orc.mlp_forward(x, t["W1"], t["b1"], t["W2"], t["b2"], "f32")
    ce_h.append(np.abs(c[:, k] - idl).max(axis=1))
    ce_n.append(np.abs(n32 - idl).max(axis=1))
ce_h, ce_n = np.array(ce_h), np.array(ce_n)        # [filters, rows]
print(f"coefficients: HIP vs ideal max {ce_h.max():.3e} (median row max {np.median(ce_h.max(0)):.3e}); numpy f32 vs ideal max {ce_n.max():.3e} "
      f"(median {np.median(ce_n.max(0)):.3e})")
order = np.argsort(-e_gold)[:8]
for r in order:
    print(f"row {r:3d} logL {ideal[r]:14.4f}  HIP-gold {e_gold[r]:.2e} HIP-ideal {e_ideal[r]:.2e} gold-ideal {e_ref[r]:.2e}  "
          f"c err HIP {ce_h[:, r].max():.2e} numpy {ce_n[:, r].max():.2e}  theta {np.array2string(case['theta'][r], precision=3)}")
parts = getattr(eng, "loglike_parts", None)
if parts is not None:
    try:
        pr = eng.loglike_parts(th)
        pr = [x.cpu().numpy() for x in pr] if isinstance(pr, (tuple, list)) else pr.cpu().numpy()
        print("parts available:", type(pr), getattr(pr, "shape", None))
    except Exception as e:  # noqa: BLE001
        print("parts:", e)
eng.close()
