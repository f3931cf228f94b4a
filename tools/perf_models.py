#!/usr/bin/env python
"""Exercise the kernels beyond em_logl under a profiler (config 3 also in ONE launch: nmma_em_loglike_stack2): me2017_lc + em_lc_loglike (config 1, B = 128 and 8192),
lc_stack_kernel + em_fused<MODE_LC_ABS> + em_lc_loglike (config 3 shape, B = 8192), em_fused coefficient / light-curve
outputs (B = 4096).  Prints wall times per call (torch events)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from nmma_amd.engine import EMEngine  # noqa: E402
from tests import cases, cases_combined, cases_me2017  # noqa: E402
from tests.helpers import engine_from_case  # noqa: E402


def timeit(fn, n=int(os.environ.get("NMMA_PERF_N", "20"))):
    for _ in range(int(os.environ.get("NMMA_PERF_WARM", "3"))):      # (NMMA_PERF_WARM=2000: the steady state bench.py measures in)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# ---- config 1: Me2017
c1 = cases_me2017.case_me2017()
from nmma_amd.em.model import BUILTIN_FILTER_LAMBDAS, C_SI  # noqa: E402
nu0 = {f: C_SI / BUILTIN_FILTER_LAMBDAS[f] for f in c1["filters"]}
eng1 = EMEngine(None, c1["filters"], ["log10_mej", "log10_vej", "beta", "log10_kappa_r"], c1["names"],
                sample_times=c1["sample_times"], cosmo_grid=c1["cosmo_grid"], data=c1["data"],
                observed_filters=c1["filters"], model_kind="me2017", filter_nu0=nu0)
for B in (128, 8192):
    rng = np.random.default_rng(B)
    th = np.stack([rng.uniform(1.0, 200.0, B), rng.uniform(1.0, 5.0, B), rng.uniform(-1.0, 2.0, B),
                   rng.uniform(-2.0, 1.0, B), rng.uniform(-2.0, -0.5, B), rng.uniform(-3.0, -0.5, B)], axis=1)
    t = torch.as_tensor(th, device="cuda:0")
    print(f"config 1 (Me2017) B={B}: loglike {timeit(lambda: eng1.loglike(t)):8.1f} us per call")
eng1.close()

# ---- config 3 shape: SVD curves + external curves -> stack -> likelihood from curves
c3 = cases_combined.case_combined()
B = int(os.environ.get("NMMA_PERF_C3_ROWS", "8192"))        # (65536: eight resident rounds instead of one)
_, th6 = syn.draw_theta(777, B, cases_combined.NAMES[:6])
rng = np.random.default_rng(778)
theta = np.concatenate([th6, rng.uniform(-17.5, -14.0, (B, 1)), rng.uniform(0.8, 1.6, (B, 1))], axis=1)
kn = EMEngine(c3["svd"], c3["filters"], c3["model_parameters"], c3["names"], sample_times=c3["sample_times"],
              cosmo_grid=c3["cosmo_grid"])
tail = EMEngine(None, c3["filters"], [], c3["names"], sample_times=c3["sample_times"], cosmo_grid=c3["cosmo_grid"],
                data=c3["data"], observed_filters=c3["filters"], model_kind="external")
t = torch.as_tensor(theta, device="cuda:0")
ext = torch.full((B, len(c3["filters"]), len(c3["sample_times"])), -15.0, dtype=torch.float64, device="cuda:0")


def combined():                    # what CombinedLightCurveModelContainer's likelihood runs: the stack formed on chip
    lc = kn.model_lightcurves(t)
    return tail.loglike_lc_sets(t, [lc, ext])


def combined_materialised():       # round 3's form: lc_stack_kernel writes the stacked set, the likelihood reads it back
    lc = kn.model_lightcurves(t)
    return tail.loglike_lc(t, tail.stack([lc, ext]))


# the one-launch form (round 5): the surrogate engine carries the photometry and takes the second transient's curves as an operand
one = EMEngine(c3["svd"], c3["filters"], c3["model_parameters"], c3["names"], sample_times=c3["sample_times"], cosmo_grid=c3["cosmo_grid"],
               data=c3["data"], observed_filters=c3["filters"], stack_operands=1)
assert one.loglike_stack2(t, ext) is not None
ref = combined()
got = one.loglike_stack2(t, ext)
rel = ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
print(f"config 3 shape B={B}: em_logl<.., 7> + the re-evaluation launch (stack2_redo, nothing flagged) {timeit(lambda: one.loglike_stack2(t, ext)):8.1f} us per call; "
      f"max rel diff to the materialising path {rel:.2e}")
print(f"   with the gap-free promise (NMMA_STACK2_GAP_FREE: no re-evaluation launch) {timeit(lambda: one.loglike_stack2(t, ext, gap_free=True)):8.1f} us per call")
print(f"   the same handle's single-model likelihood (the kilonova alone, no operand: what the flux sum adds) {timeit(lambda: one.loglike(t)):8.1f} us per call")
one.set_option("stack2_fixup", 0)             # (measurement only: the kernel alone, without the re-evaluation launch)
print(f"   em_logl<.., 7> alone (no re-evaluation launch) {timeit(lambda: one.loglike_stack2(t, ext)):8.1f} us per call")
one.close()

lc_fixed = kn.model_lightcurves(t)
stacked = tail.stack([lc_fixed, ext])
print(f"config 3 shape B={B}: SVD curves + fused stack / likelihood {timeit(combined):8.1f} us per call "
      f"(materialised stack: {timeit(combined_materialised):8.1f} us)")
print(f"   tail alone: fused stack + likelihood {timeit(lambda: tail.loglike_lc_sets(t, [lc_fixed, ext])):8.1f} us; "
      f"lc_stack {timeit(lambda: tail.stack([lc_fixed, ext])):8.1f} us + likelihood from curves {timeit(lambda: tail.loglike_lc(t, stacked)):8.1f} us")
for grp in ("16", "32", "64"):                 # every group size of the same kernels, like for like (default: 32 lanes per sample)
    tail.set_option("lc_group", int(grp))
    print(f"   lc_group={grp}: fused {timeit(lambda: tail.loglike_lc_sets(t, [lc_fixed, ext])):8.1f} us; "
          f"likelihood from curves {timeit(lambda: tail.loglike_lc(t, stacked)):8.1f} us")
tail.set_option("lc_group", 0)
assert torch.equal(tail.loglike_lc_sets(t, [lc_fixed, ext]), tail.loglike_lc(t, stacked))
kn.close(); tail.close()

# ---- auxiliary outputs of the SVD model
c2 = cases.case_c2_default()
eng = engine_from_case(c2)
t = torch.as_tensor(syn.draw_theta(7, 4096, c2["names"])[1], device="cuda:0")
print(f"config 2 B=4096: coefficients {timeit(lambda: eng.coefficients(t)):8.1f} us, light curves {timeit(lambda: eng.lightcurves(t)):8.1f} us, "
      f"per-filter parts {timeit(lambda: eng.loglike_parts(t)):8.1f} us")
eng.close()
