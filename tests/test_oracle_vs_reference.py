"""Re-run the reference's own source (when /root/reference is present, i.e. in the
build container) and check it still reproduces EVERY committed golden vector: all SVD cases
(every row), the Me2017 model (config 1) and the combined model (config 3 shape)."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/nmma"),
                                reason="reference tree not available (GPU box)")


def _rows(lik, names, theta, fixed=None):
    return np.array([lik.log_likelihood(dict(zip(names, (float(v) for v in row)), **(fixed or {}))) for row in theta])


@pytest.mark.parametrize("name", list(cases.CASES))
def test_reference_reproduces_golden(name):
    from tools.make_golden import build_reference_likelihood
    case = cases.CASES[name]()
    gold = cases.load_golden(name)
    _, lik, _ = build_reference_likelihood(case)
    got = _rows(lik, case["names"], case["theta"], case.get("fixed"))
    assert len(got) == len(gold["logl"])
    np.testing.assert_allclose(got, gold["logl"], rtol=1e-13)


def test_reference_reproduces_me2017_golden():
    from tests import cases_me2017
    from tools.make_golden_me2017 import build_reference
    case = cases_me2017.case_me2017()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("me2017")["logl"], rtol=1e-13)


def test_reference_reproduces_combined_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined")["logl"], rtol=1e-13)


@pytest.mark.parametrize("name", ["combined_syserr", "combined_loggrid"])
def test_reference_reproduces_combined_extras_golden(name):
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = getattr(cases_combined, "case_" + name)()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden(name)["logl"], rtol=1e-13)


def test_reference_reproduces_combined_union_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined_union()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined_union")["logl"], rtol=1e-13)


def test_reference_reproduces_combined_owngrids_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined_owngrids()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined_owngrids")["logl"], rtol=1e-13)


def test_reference_reproduces_combined_nullfilters_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined_nullfilters()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined_nullfilters")["logl"], rtol=1e-13)


def test_reference_floors_a_single_model_that_lists_a_filter_without_a_network():
    """calc_svd_lc's null output (+inf on every node) fails sanity_check for every sample: what the plugin answers for such a model."""
    from nmma_amd import synthetic as syn
    from oracle import ref_harness
    from tools.make_golden import _KerasStandIn
    case = syn.config2_case()
    allf = list(case["model_filters"]) + ["X-ray-1keV"]
    ref = ref_harness.reference_modules()
    ref.utils.get_all_bandpass_metadata = lambda: [{"name": n} for n in allf]
    ref.utils.M4OPT_INSTALLED = False
    grid = case["cosmo_grid"]
    m = object.__new__(ref.model.SVDLightCurveModel)
    svd_ref = {}
    for f, t in case["svd"].items():
        d = {k: t[k] for k in ("param_mins", "param_maxs", "mins", "maxs", "tt", "n_coeff", "VA")}
        d["model"] = _KerasStandIn(t)
        svd_ref[f] = d
    m.model, m.model_parameters, m.filters = case["model"], list(case["model_parameters"]), list(allf)
    m.svd_mag_model, m.mag_ncoeff, m.lbol_ncoeff, m.good_parameters = svd_ref, None, None, True
    m.default_filts, m.lambdas, m.nu_0s = list(allf), np.ones(len(allf)), np.ones(len(allf))
    m.model_times = case["sample_times"] if case["sample_times"] is not None else next(iter(svd_ref.values()))["tt"]
    m.redshift_func = lambda p: np.interp(p["luminosity_distance"], grid[0], grid[1])
    m.check_vs_priors = lambda priors: None
    times, mags, sigmas = case["data"]
    priors = ref.base.PriorDict({n: object() for n in case["names"]})
    handler = ref.systematics.FilterSystematicsHandler(case["observed_filters"], systematics_file=None, error_budget=1.0, light_curve_times=times)
    lik = ref.em_likelihood.EMTransientLikelihood(m, (times, mags, sigmas, 0.0), handler, priors, filters=case["observed_filters"], detection_limit=np.inf)
    got = _rows(lik, case["names"], case["theta"][:6])
    assert np.all(got == -1.7976931348623157e308)


def test_reference_reproduces_combined_limit_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined_limit()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined_limit")["logl"], rtol=1e-13)


def test_reference_reproduces_combined_nodes_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined_nodes()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined_nodes")["logl"], rtol=1e-13)
