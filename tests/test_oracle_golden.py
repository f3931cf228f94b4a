"""The CPU oracle (oracle/nmma_oracle.py) against the golden vectors produced by the
reference's own source (tools/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import nmma_oracle as orc
from tests import cases
from tools.make_golden import build_oracle_likelihood, N_STAGE_ROWS

ALL = list(cases.CASES)


@pytest.fixture(scope="module", params=ALL)
def loaded(request):
    case = cases.CASES[request.param]()
    gold = cases.load_golden(request.param)
    return request.param, case, gold


def test_generator_is_stable(loaded):
    _, case, gold = loaded
    assert cases.weights_digest(case["svd"]) == pytest.approx(float(gold["digest"]), rel=1e-13)


@pytest.mark.parametrize("use_scipy", [True, False])
def test_logl_matches_reference(loaded, use_scipy):
    name, case, gold = loaded
    lik = build_oracle_likelihood(case, use_scipy=use_scipy)
    n = len(case["theta"]) if use_scipy else min(len(case["theta"]), 24)
    got = orc.log_likelihood_batch(lik, case["names"], case["theta"][:n], case.get("fixed"))
    want = gold["logl"][:n]
    floor = want == orc.LOGL_FLOOR
    assert np.array_equal(got == orc.LOGL_FLOOR, floor)
    np.testing.assert_allclose(got[~floor], want[~floor], rtol=1e-12, atol=0)


def test_stages_match_reference(loaded):
    name, case, gold = loaded
    lik = build_oracle_likelihood(case)
    for i in range(min(N_STAGE_ROWS, len(case["theta"]))):
        p = dict(zip(case["names"], (float(v) for v in case["theta"][i])), **(case.get("fixed") or {}))
        p = lik.model.parameter_conversion(p)
        obs_times, lc = lik.model.gen_detector_lc(p)
        np.testing.assert_allclose(obs_times, gold[f"s{i}_obs_times"], rtol=1e-15)
        for k, f in enumerate(case["model_filters"]):
            np.testing.assert_allclose(lc[f], gold[f"s{i}_app_{k}"], rtol=1e-14)
        if f"s{i}_est_0" in gold:
            est = lik.expected_mags(obs_times, lc)
            for j, f in enumerate(case["observed_filters"]):
                np.testing.assert_allclose(est[f], gold[f"s{i}_est_{j}"], rtol=1e-14)


def test_floor_value():
    assert orc.LOGL_FLOOR == -1.7976931348623157e308


def test_closed_form_truncnorm_matches_scipy():
    from scipy import stats
    rng = np.random.default_rng(0)
    m = rng.normal(18, 1, 200)
    loc = rng.normal(18, 1, 200)
    sc = rng.uniform(0.05, 2.0, 200)
    for lim in (np.inf, 19.0, 17.5, 30.0):
        want = stats.truncnorm.logpdf(m, -np.inf, (lim - loc) / sc, loc=loc, scale=sc)
        got = orc.truncated_gaussian_logpdf(m, loc, sc, lim)
        fin = np.isfinite(want)
        assert np.array_equal(np.isneginf(want), np.isneginf(got))
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-13)
    # infinite model magnitude -> NaN (both limit kinds)
    for lim in (np.inf, 20.0):
        assert np.isnan(stats.truncnorm.logpdf(18.0, -np.inf, (lim - np.inf) / 1.0, loc=np.inf, scale=1.0))
        assert np.isnan(orc.truncated_gaussian_logpdf(np.array([18.0]), np.array([np.inf]), np.array([1.0]), lim))[0]


def test_f64acc_mlp_close_to_f32():
    case = cases.case_c2_default()
    f = case["svd"]["ps1::g"]
    x = np.random.default_rng(1).random((32, 4))
    a = orc.mlp_forward(x, f["W1"], f["b1"], f["W2"], f["b2"], "f32")
    b = orc.mlp_forward(x, f["W1"], f["b1"], f["W2"], f["b2"], "f64acc")
    assert np.max(np.abs(a - b)) < 2e-5
