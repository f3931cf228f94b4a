#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE'S OWN SOURCE (imported from
/root/reference under oracle/ref_harness.py) on the seeded cases of tests/cases.py.

Runs only in the build container (the reference does not travel to the GPU box).
Output: tests/golden/<case>.npz holding expected outputs only (inputs are re-created
from seeds; a weights digest guards the generator):

  logl[B]                      EMTransientLikelihood.log_likelihood per theta row
  digest                       tests.cases.weights_digest(svd)
  s<k>_obs_times, s<k>_app_<i> gen_detector_lc for the first K rows (filter index i)
  s<k>_est_<j>                 update_lightcurve_reference per observed filter j
  s<k>_c_<i>                   SVD coefficients per model filter (fp32 numpy stand-in
                               for the Keras call, lightcurve_generation.py:198)

Usage: python tools/make_golden.py [case ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from oracle import nmma_oracle as orc  # noqa: E402
from tests import cases  # noqa: E402

N_STAGE_ROWS = 4


class _KerasStandIn:
    """Callable with the interface the reference uses: ``model(x2d).numpy()``
    (nmma/em/lightcurve_generation.py:198).  fp32 numpy Dense-relu-Dense."""

    def __init__(self, filt):
        self.f = filt
        self.last = None

    def __call__(self, x2d):
        out = orc.mlp_forward(x2d, self.f["W1"], self.f["b1"], self.f["W2"], self.f["b2"], "f32")
        self.last = out.copy()

        class _T:
            def numpy(_s):
                return out
        return _T()


def build_reference_likelihood(case):
    ref = ref_harness.reference_modules()
    svd_ref = {}
    for f, t in case["svd"].items():
        d = {k: t[k] for k in ("param_mins", "param_maxs", "mins", "maxs", "tt", "n_coeff")}
        d["VA"] = t["VA"]
        d["model"] = _KerasStandIn(t)
        svd_ref[f] = d

    all_names = list(dict.fromkeys(list(case["model_filters"]) + list(case["observed_filters"])))
    ref.utils.get_all_bandpass_metadata = lambda: [{"name": n} for n in all_names
                                                   if n not in ("w", "o", "c", "V", "I", "F606W", "F814W")]

    m = object.__new__(ref.model.SVDLightCurveModel)
    m.model = case["model"]
    m.model_parameters = list(case["model_parameters"])
    m.filters = list(case["model_filters"])
    m.svd_mag_model = svd_ref
    m.mag_ncoeff = None
    m.lbol_ncoeff = None
    m.good_parameters = True
    m.default_filts = list(case["model_filters"])
    m.lambdas = np.ones(len(m.filters))
    m.nu_0s = np.ones(len(m.filters))
    m.model_times = (case["sample_times"] if case["sample_times"] is not None
                     else m.setup_model_times())
    grid = case["cosmo_grid"]
    if grid is not None:
        m.redshift_func = lambda p: np.interp(p["luminosity_distance"], grid[0], grid[1])
    else:
        m.redshift_func = lambda p: 0.0
    m.check_vs_priors = lambda priors: None

    times, mags, sigmas = case["data"]
    priors = ref.base.PriorDict({n: object() for n in case["names"]})
    sysr = case["systematics_ref"]
    handler = ref.systematics.FilterSystematicsHandler(
        list(case["observed_filters"]), systematics_file=sysr["systematics_file"],
        error_budget=sysr["error_budget"], light_curve_times=times)
    lik = ref.em_likelihood.EMTransientLikelihood(
        m, (times, mags, sigmas, 0.0), handler, priors,
        filters=list(case["observed_filters"]), detection_limit=case["detection_limit"])
    return ref, lik, m


def run_case(name):
    case = cases.CASES[name]()
    ref, lik, m = build_reference_likelihood(case)
    names, theta = case["names"], case["theta"]
    out = {"digest": np.float64(cases.weights_digest(case["svd"]))}
    logl = np.empty(len(theta))
    fixed = case.get("fixed") or {}
    for i, row in enumerate(theta):
        p = dict(zip(names, (float(v) for v in row)), **fixed)
        logl[i] = lik.log_likelihood(p)
        if i < N_STAGE_ROWS:
            pc = lik.parameter_conversion(dict(zip(names, (float(v) for v in row)), **fixed))
            obs_times, lc = m.gen_detector_lc(pc)
            out[f"s{i}_obs_times"] = np.asarray(obs_times, float)
            for k, f in enumerate(case["model_filters"]):
                out[f"s{i}_app_{k}"] = np.asarray(lc[f], float)
                out[f"s{i}_c_{k}"] = m.svd_mag_model[f]["model"].last[0].copy()
            if lik.sub_model.sanity_check(lc):
                est = lik.sub_model.update_lightcurve_reference(obs_times, lc)
                for j, f in enumerate(case["observed_filters"]):
                    out[f"s{i}_est_{j}"] = np.asarray(est[f], float)
    out["logl"] = logl
    # cross-check with the restatement right away
    olik = build_oracle_likelihood(case)
    ol = orc.log_likelihood_batch(olik, names, theta, fixed)
    rel = np.max(np.abs(ol - logl) / np.maximum(1.0, np.abs(logl)))
    n_floor = int(np.sum(logl == orc.LOGL_FLOOR))
    print(f"{name:20s} B={len(theta):4d} floor={n_floor:3d} "
          f"logL[min,max]=({logl[logl > orc.LOGL_FLOOR].min() if n_floor < len(logl) else float('nan'):.3f},"
          f"{logl.max():.3f})  oracle-vs-reference max rel diff = {rel:.3e}")
    os.makedirs(cases.GOLDEN_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(cases.GOLDEN_DIR, f"{name}.npz"), **out)
    return rel


def build_oracle_likelihood(case, use_scipy=True, mlp_mode="f32"):
    return orc.likelihood_from_case(case, use_scipy=use_scipy, mlp_mode=mlp_mode)


if __name__ == "__main__":
    todo = sys.argv[1:] or list(cases.CASES)
    worst = 0.0
    for n in todo:
        worst = max(worst, run_case(n))
    print("worst rel diff:", worst)
