"""CPU: the Me2017 (config 1) and combined-model (config 3 shape) oracles against the golden
vectors produced by the reference's own source."""
import numpy as np
import pytest

from oracle import nmma_oracle as orc
from tests import cases, cases_combined, cases_me2017


def test_me2017_oracle_matches_golden():
    case = cases_me2017.case_me2017()
    gold = cases.load_golden("me2017")
    lik = cases_me2017.oracle_likelihood(case)
    got = orc.log_likelihood_batch(lik, case["names"], case["theta"][:32])
    np.testing.assert_allclose(got, gold["logl"][:32], rtol=1e-12)
    p = lik.model.parameter_conversion(dict(zip(case["names"], (float(v) for v in case["theta"][0]))))
    tobs, lc = lik.model.gen_detector_lc(p)
    np.testing.assert_allclose(tobs, gold["s0_obs_times"], rtol=1e-15)
    for k, f in enumerate(case["filters"]):
        want = gold[f"s0_app_{k}"]
        assert np.array_equal(np.isfinite(lc[f]), np.isfinite(want))
        np.testing.assert_allclose(lc[f][np.isfinite(want)], want[np.isfinite(want)], rtol=1e-13)


def test_combined_oracle_matches_golden():
    case = cases_combined.case_combined()
    gold = cases.load_golden("combined")
    lik, _ = cases_combined.oracle_likelihood(case)
    got = orc.log_likelihood_batch(lik, case["names"], case["theta"][:16])
    np.testing.assert_allclose(got, gold["logl"][:16], rtol=1e-12)


@pytest.mark.parametrize("name", ["combined_syserr", "combined_loggrid"])
def test_combined_extras_oracle_matches_golden(name):
    """Shared-grid combinations with a sampled systematic / on the log-spaced grid, the second transient's curves with an interior hole
    for part of the rows (filled by the reference's autocomplete_data): the oracle against the reference-made golden, stage by stage."""
    case = getattr(cases_combined, "case_" + name)()
    gold = cases.load_golden(name)
    lik, _ = cases_combined.oracle_likelihood(case)
    rows = np.r_[0:10, np.nonzero(case["theta"][:, 7] > case["grb_hole"][2])[0][:6]]      # (some rows without, some with holes)
    got = orc.log_likelihood_batch(lik, case["names"], case["theta"][rows])
    np.testing.assert_allclose(got, gold["logl"][rows], rtol=1e-12)
    for i in range(3):
        p = lik.model.parameter_conversion(dict(zip(case["names"], (float(v) for v in case["theta"][i]))))
        tobs, lc = lik.model.gen_detector_lc(p)
        np.testing.assert_allclose(tobs, gold[f"s{i}_obs_times"], rtol=1e-15)
        for f in case["filters"]:
            want = gold[f"s{i}_app_{f}"]
            fin = np.isfinite(want)
            assert np.array_equal(np.isfinite(lc[f]), fin)
            np.testing.assert_allclose(lc[f][fin], want[fin], rtol=1e-13)


def test_combined_union_oracle_matches_golden():
    case = cases_combined.case_combined_union()
    gold = cases.load_golden("combined_union")
    lik, _ = cases_combined.oracle_likelihood_union(case)
    got = orc.log_likelihood_batch(lik, case["names"], case["theta"][:16])
    np.testing.assert_allclose(got, gold["logl"][:16], rtol=1e-12)
    p = lik.model.parameter_conversion(dict(zip(case["names"], (float(v) for v in case["theta"][0]))))
    tobs, lc = lik.model.gen_detector_lc(p)
    np.testing.assert_allclose(tobs, gold["s0_obs_times"], rtol=1e-15)
    for f in ("g", "z", "J", "w"):
        want = gold[f"s0_app_{f}"]
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(lc[f]), fin)
        np.testing.assert_allclose(lc[f][fin], want[fin], rtol=1e-13)


def test_combined_owngrids_oracle_matches_golden():
    """Sub-models on their own time grids, all filters listed by the surrogate (the one-launch form's round-6 case)."""
    case = cases_combined.case_combined_owngrids()
    gold = cases.load_golden("combined_owngrids")
    lik, _ = cases_combined.oracle_likelihood_owngrids(case)
    got = orc.log_likelihood_batch(lik, case["names"], case["theta"][:20])
    assert np.array_equal(got == orc.LOGL_FLOOR, gold["logl"][:20] == orc.LOGL_FLOOR)
    fin = got != orc.LOGL_FLOOR
    np.testing.assert_allclose(got[fin], gold["logl"][:20][fin], rtol=1e-12)
    p = lik.model.parameter_conversion(dict(zip(case["names"], (float(v) for v in case["theta"][0]))))
    tobs, lc = lik.model.gen_detector_lc(p)
    np.testing.assert_allclose(tobs, gold["s0_obs_times"], rtol=1e-15)
    for f in ("ps1::g", "2massh", "sdssu"):
        want = gold[f"s0_app_{f}"]
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(lc[f]), fin)
        np.testing.assert_allclose(lc[f][fin], want[fin], rtol=1e-13)


def test_combined_nullfilters_oracle_matches_golden():
    """Filters the surrogate lists without having a network for them (calc_svd_lc's null output): the band is the afterglow's alone."""
    case = cases_combined.case_combined_nullfilters()
    gold = cases.load_golden("combined_nullfilters")
    lik, _ = cases_combined.oracle_likelihood_nullfilters(case)
    got = orc.log_likelihood_batch(lik, case["names"], case["theta"][:16])
    np.testing.assert_allclose(got, gold["logl"][:16], rtol=1e-12)


def test_combined_limit_oracle_matches_golden():
    case = cases_combined.case_combined_limit()
    lik, _ = cases_combined.oracle_likelihood(case)
    got = orc.log_likelihood_batch(lik, case["names"], case["theta"][:16])
    np.testing.assert_allclose(got, cases.load_golden("combined_limit")["logl"][:16], rtol=1e-12)


def test_combined_nodes_oracle_matches_golden():
    case = cases_combined.case_combined_nodes()
    lik, _ = cases_combined.oracle_likelihood(case)
    got = orc.log_likelihood_batch(lik, case["names"], case["theta"][:16])
    np.testing.assert_allclose(got, cases.load_golden("combined_nodes")["logl"][:16], rtol=1e-12)
