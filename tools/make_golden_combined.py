#!/usr/bin/env python
"""Golden vectors for the combined-model path (BASELINE config 3 shape) from the REFERENCE'S OWN
CombinedLightCurveModelContainer (nmma/em/model.py:1342-1510) under oracle/ref_harness.py.  The GRB
sub-model is a power-law stand-in subclassing the reference's LightCurveModelContainer (afterglowpy
is third-party and absent).  Output: tests/golden/combined.npz, combined_union.npz, combined_syserr.npz, combined_loggrid.npz, combined_owngrids.npz, combined_nullfilters.npz, combined_limit.npz, combined_nodes.npz
(`python tools/make_golden_combined.py combined_owngrids` / `combined_nullfilters` / `combined_limit` / `combined_nodes` writes that one alone)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import nmma_oracle as orc  # noqa: E402
from oracle import ref_harness  # noqa: E402
from tests import cases_combined  # noqa: E402
from tools.make_golden import _KerasStandIn  # noqa: E402


def build_reference(case):
    """(likelihood, combined model) built from the reference's own classes for a tests.cases_combined case
    (shared grid / filters, or -- keys grb_filters, grb_times, observed_filters -- the general union case)."""
    ref = ref_harness.reference_modules()
    # (key all_filters: both sub-models LIST these filters, the surrogate has a network for case["filters"] only -- null output for the rest)
    listed = list(case.get("all_filters", case["filters"]))
    grb_filters = list(case.get("grb_filters", listed))
    grb_times = case.get("grb_times", case["sample_times"])
    obs_filters = list(case.get("observed_filters", listed))
    ref.utils.get_all_bandpass_metadata = lambda: [{"name": n} for n in set(listed) | set(grb_filters)
                                                   if n not in ("w", "o", "c", "V", "I", "F606W", "F814W")]
    ref.utils.M4OPT_INSTALLED = False
    grid = case["cosmo_grid"]
    zfun = lambda p: np.interp(p["luminosity_distance"], grid[0], grid[1])

    kn = object.__new__(ref.model.SVDLightCurveModel)
    svd_ref = {}
    for f, t in case["svd"].items():
        d = {k: t[k] for k in ("param_mins", "param_maxs", "mins", "maxs", "tt", "n_coeff", "VA")}
        d["model"] = _KerasStandIn(t)
        svd_ref[f] = d
    kn.model, kn.model_parameters, kn.filters = case["model"], list(case["model_parameters"]), list(listed)
    kn.svd_mag_model, kn.mag_ncoeff, kn.lbol_ncoeff, kn.good_parameters = svd_ref, None, None, True
    kn.default_filts, kn.lambdas, kn.nu_0s = list(listed), np.ones(len(listed)), np.ones(len(listed))
    kn.model_times, kn.redshift_func = case["sample_times"], zfun
    kn.check_vs_priors = lambda priors: None

    helper = orc.OraclePowerLawModel(grb_filters, grb_times, hole=case.get("grb_hole"))

    class PowerLawGRB(ref.model.LightCurveModelContainer):
        def __init__(self):
            self.model, self.model_parameters = "PLGRB", ["grb_mag0", "grb_slope"]
            self.filters, self.default_filts = list(grb_filters), list(grb_filters)
            self.lambdas = self.nu_0s = np.ones(len(grb_filters))
            self.good_parameters, self.model_times, self.redshift_func = True, np.asarray(grb_times, float), zfun

        def check_vs_priors(self, priors):
            pass

        def generate_lightcurve(self, sample_times, parameters):
            self.em_parameter_setup(parameters)
            return helper.abs_lightcurves(parameters, sample_times)

    comb = ref.model.CombinedLightCurveModelContainer([kn, PowerLawGRB()])
    times, mags, sigmas = case["data"]
    priors = ref.base.PriorDict({n: object() for n in case["names"]})
    sys_ref = case.get("systematics_ref") or dict(error_budget=1.0, systematics_file=None)
    handler = ref.systematics.FilterSystematicsHandler(obs_filters, systematics_file=sys_ref["systematics_file"],
                                                       error_budget=sys_ref["error_budget"], light_curve_times=times)
    lik = ref.em_likelihood.EMTransientLikelihood(comb, (times, mags, sigmas, 0.0), handler, priors,
                                                  filters=obs_filters, detection_limit=case.get("detection_limit", np.inf))
    return lik, comb


def run(case, oracle_builder, name, n_stage=3):
    lik, comb = build_reference(case)
    names, theta = case["names"], case["theta"]
    logl = np.array([lik.log_likelihood(dict(zip(names, (float(v) for v in row)))) for row in theta])
    olik, _ = oracle_builder(case)
    ol = orc.log_likelihood_batch(olik, names, theta)
    floor = logl == orc.LOGL_FLOOR
    assert np.array_equal(ol == orc.LOGL_FLOOR, floor)
    rel = np.max(np.abs(ol[~floor] - logl[~floor]) / np.maximum(1, np.abs(logl[~floor])))
    print(f"{name}: B={len(theta)} floor={floor.sum()} logL range ({logl[~floor].min():.2f}, {logl[~floor].max():.2f})"
          f" oracle-vs-reference max rel diff {rel:.3e}")
    out = {"logl": logl}
    for i in range(n_stage):
        p = lik.parameter_conversion(dict(zip(names, (float(v) for v in theta[i]))))
        tobs, lc = comb.gen_detector_lc(p)
        out[f"s{i}_obs_times"] = np.asarray(tobs, float)
        for k, f in enumerate(comb.all_filters if hasattr(comb, "all_filters") else case["filters"]):
            out[f"s{i}_app_{f}"] = np.asarray(lc[f], float)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"{name}.npz"), **out)


def main():
    if sys.argv[1:] == ["combined_owngrids"]:
        run(cases_combined.case_combined_owngrids(), cases_combined.oracle_likelihood_owngrids, "combined_owngrids")
        return
    if sys.argv[1:] == ["combined_nodes"]:
        run(cases_combined.case_combined_nodes(), cases_combined.oracle_likelihood, "combined_nodes")
        return
    if sys.argv[1:] == ["combined_limit"]:
        run(cases_combined.case_combined_limit(), cases_combined.oracle_likelihood, "combined_limit")
        return
    if sys.argv[1:] == ["combined_nullfilters"]:
        run(cases_combined.case_combined_nullfilters(), cases_combined.oracle_likelihood_nullfilters, "combined_nullfilters")
        return
    case = cases_combined.case_combined()
    # (the shared-grid golden keeps its original key layout: s<i>_app_<filter index>)
    lik, comb = build_reference(case)
    names, theta = case["names"], case["theta"]
    logl = np.array([lik.log_likelihood(dict(zip(names, (float(v) for v in row)))) for row in theta])
    olik, _ = cases_combined.oracle_likelihood(case)
    ol = orc.log_likelihood_batch(olik, names, theta)
    floor = logl == orc.LOGL_FLOOR
    assert np.array_equal(ol == orc.LOGL_FLOOR, floor)
    rel = np.max(np.abs(ol[~floor] - logl[~floor]) / np.maximum(1, np.abs(logl[~floor])))
    print(f"combined: B={len(theta)} floor={floor.sum()} oracle-vs-reference max rel diff {rel:.3e}")
    out = {"logl": logl}
    for i in range(3):
        p = lik.parameter_conversion(dict(zip(names, (float(v) for v in theta[i]))))
        tobs, lc = comb.gen_detector_lc(p)
        out[f"s{i}_obs_times"] = np.asarray(tobs, float)
        for k, f in enumerate(case["filters"]):
            out[f"s{i}_app_{k}"] = np.asarray(lc[f], float)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "combined.npz"), **out)
    run(cases_combined.case_combined_union(), cases_combined.oracle_likelihood_union, "combined_union")
    run(cases_combined.case_combined_syserr(), cases_combined.oracle_likelihood, "combined_syserr")
    run(cases_combined.case_combined_loggrid(), cases_combined.oracle_likelihood, "combined_loggrid")
    run(cases_combined.case_combined_owngrids(), cases_combined.oracle_likelihood_owngrids, "combined_owngrids")
    run(cases_combined.case_combined_nullfilters(), cases_combined.oracle_likelihood_nullfilters, "combined_nullfilters")
    run(cases_combined.case_combined_limit(), cases_combined.oracle_likelihood, "combined_limit")
    run(cases_combined.case_combined_nodes(), cases_combined.oracle_likelihood, "combined_nodes")


if __name__ == "__main__":
    main()
