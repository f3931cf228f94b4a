#!/usr/bin/env python
"""Config 3's shape with two NULL filters (a radio and an X-ray band the kilonova lists without a network for them,
lightcurve_generation.py:168-169; the drivers' shared grid): the one-launch form with the null filters as model filters of the engine
against the materialising path (the surrogate's curves of its 9 filters, regrid onto the 11, flux sum + likelihood from curves).
Wall times per call (torch events)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from nmma_amd.engine import EMEngine  # noqa: E402
from tests import cases_combined  # noqa: E402
from tools.perf_owngrids import timeit  # noqa: E402  (runs that tool's measurements first when imported: kept in one profile)

c = cases_combined.case_combined_nullfilters()
B = int(os.environ.get("NMMA_PERF_C3_ROWS", "8192"))
_, th6 = syn.draw_theta(777, B, cases_combined.NAMES[:6])
rng = np.random.default_rng(778)
theta = np.concatenate([th6, rng.uniform(-17.5, -14.0, (B, 1)), rng.uniform(0.8, 1.6, (B, 1))], axis=1)
t = torch.as_tensor(theta, device="cuda:0")
st, F, A = c["sample_times"], c["filters"], c["all_filters"]
with np.errstate(divide="ignore"):
    base = theta[:, 6:7] + 2.5 * theta[:, 7:8] * np.log10(st)[None, :]
ext = torch.as_tensor(np.stack([np.where(st >= 0.3, base + 0.15 * k, np.inf) for k in range(len(A))], axis=1), device="cuda:0")

one = EMEngine(c["svd"], A, c["model_parameters"], c["names"], sample_times=st, cosmo_grid=c["cosmo_grid"], data=c["data"],
               observed_filters=A, stack_operands=1, null_filters=cases_combined.NULL_FILTERS)
kn = EMEngine(c["svd"], F, c["model_parameters"], c["names"], sample_times=st, cosmo_grid=c["cosmo_grid"])
tail = EMEngine(None, A, [], c["names"], sample_times=st, cosmo_grid=c["cosmo_grid"], data=c["data"], observed_filters=A, model_kind="external")
plan = [[k] for k in range(len(F))] + [[], []]


def materialising():
    return tail.loglike_lc_sets(t, [tail.regrid(kn.model_lightcurves(t), st, plan), ext])


a, b = one.loglike_stack2(t, ext), materialising()
rel = ((a - b).abs() / b.abs().clamp(min=1.0)).max().item()
print(f"config 3's shape + 2 null filters (11 model filters, {len(st)} nodes) B={B}: max rel diff one launch vs materialising {rel:.2e}")
print(f"   one launch: em_logl<.., 7> + stack2_redo {timeit(lambda: one.loglike_stack2(t, ext)):8.1f} us per call")
print(f"   materialising path: em_fused<MODE_LC_ABS> (9 filters) + regrid onto 11 + em_lc_loglike {timeit(materialising):8.1f} us per call")
for e in (one, kn, tail):
    e.close()
