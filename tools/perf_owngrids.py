#!/usr/bin/env python
"""Config 3's shape with the sub-models on their OWN time grids (kilonova on the CLI grid of 41 nodes, second transient on 36
log-spaced nodes: a union grid of 76 nodes): the one-launch form on the union grid (engine argument base_times: regrid of the operand
+ em_logl<.., 8> + the re-evaluation launch) against the materialising path (em_fused<MODE_LC_ABS> + two regrids + em_lc_loglike),
and against the same photometry with both sub-models on ONE grid.  Wall times per call (torch events)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from nmma_amd.engine import EMEngine  # noqa: E402
from tests import cases_combined  # noqa: E402


def timeit(fn, n=int(os.environ.get("NMMA_PERF_N", "20"))):
    for _ in range(int(os.environ.get("NMMA_PERF_WARM", "3"))):      # (NMMA_PERF_WARM=1500: the steady state bench.py measures in)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


c3 = cases_combined.case_combined()
B = int(os.environ.get("NMMA_PERF_C3_ROWS", "8192"))
_, th6 = syn.draw_theta(777, B, cases_combined.NAMES[:6])
rng = np.random.default_rng(778)
theta = np.concatenate([th6, rng.uniform(-17.5, -14.0, (B, 1)), rng.uniform(0.8, 1.6, (B, 1))], axis=1)
t = torch.as_tensor(theta, device="cuda:0")
st, gt = c3["sample_times"], np.geomspace(0.25, 30.0, 36)
union = np.array(sorted(set(st.tolist()) | set(gt.tolist())))
F = c3["filters"]
ident = [[k] for k in range(len(F))]
with np.errstate(divide="ignore"):
    base = theta[:, 6:7] + 2.5 * theta[:, 7:8] * np.log10(gt)[None, :]
ext = torch.as_tensor(np.stack([np.where(gt >= 0.3, base + 0.15 * k, np.inf) for k in range(len(F))], axis=1), device="cuda:0")

one = EMEngine(c3["svd"], F, c3["model_parameters"], c3["names"], sample_times=union, base_times=st, cosmo_grid=c3["cosmo_grid"],
               data=c3["data"], observed_filters=F, stack_operands=1)
kn = EMEngine(c3["svd"], F, c3["model_parameters"], c3["names"], sample_times=st, cosmo_grid=c3["cosmo_grid"])
tail = EMEngine(None, F, [], c3["names"], sample_times=union, cosmo_grid=c3["cosmo_grid"], data=c3["data"], observed_filters=F, model_kind="external")


def one_launch():
    return one.loglike_stack2(t, one.regrid(ext, gt, ident), completed=True)


def materialising():
    return tail.loglike_lc_sets(t, [tail.regrid(kn.model_lightcurves(t), st, ident), tail.regrid(ext, gt, ident)])


a, b = one_launch(), materialising()
rel = ((a - b).abs() / b.abs().clamp(min=1.0)).max().item()
print(f"config 3's shape on own grids (union grid of {len(union)} nodes) B={B}: max rel diff one launch vs materialising {rel:.2e}; "
      f"{int((a == a.min()).sum())} rows on the floor")
print(f"   one launch: regrid of the operand + em_logl<.., 8> + stack2_redo {timeit(one_launch):8.1f} us per call")
lc2 = one.regrid(ext, gt, ident)
print(f"      the likelihood call alone (operand already on the union grid) {timeit(lambda: one.loglike_stack2(t, lc2, completed=True)):8.1f} us; "
      f"regrid of the operand {timeit(lambda: one.regrid(ext, gt, ident)):8.1f} us")
print(f"   materialising path: em_fused<MODE_LC_ABS> + two regrids + em_lc_loglike {timeit(materialising):8.1f} us per call")
for e in (one, kn, tail):
    e.close()
