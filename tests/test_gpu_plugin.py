"""GPU tests of the reference-shaped plugin classes (the drop-in boundary): the same
golden vectors, driven the way the reference's tests / samplers drive the reference."""
import pickle

import numpy as np
import pytest

from tests import cases
from tests.helpers import plugin_from_case, rel_err

pytestmark = pytest.mark.gpu
FLOOR = -1.7976931348623157e308


@pytest.mark.parametrize("name", ["c2_default", "c2_dt05_limit", "syserr_time_nodes", "averaging", "edges"])
def test_likelihood_plugin_matches_golden(name):
    case = cases.CASES[name]()
    gold = cases.load_golden(name)["logl"]
    model, handler, lik = plugin_from_case(case)
    # batched entry point
    got = lik.log_likelihood_batch(case["theta"], case["names"])
    floor = gold == FLOOR
    assert np.array_equal(got == FLOOR, floor)
    assert rel_err(got[~floor], gold[~floor]).max() <= 1e-6
    # per-sample reference API: dict in, float out
    for i in range(6):
        p = dict(zip(case["names"], (float(v) for v in case["theta"][i])))
        v = lik.log_likelihood(p)
        assert isinstance(v, float)
        if floor[i]:
            assert v == FLOOR
        else:
            assert v == pytest.approx(gold[i], rel=1e-6)
    assert lik.noise_log_likelihood() == 0.0
    assert lik.log_likelihood_ratio(dict(zip(case["names"], case["theta"][0]))) == pytest.approx(
        lik.log_likelihood(dict(zip(case["names"], case["theta"][0]))))
    assert "EMTransientLikelihood" in repr(lik)


def test_gen_detector_lc_matches_golden():
    case = cases.case_c2_dt05_limit()
    gold = cases.load_golden("c2_dt05_limit")
    model, _, lik = plugin_from_case(case)
    for i in range(3):
        p = lik.parameter_conversion(dict(zip(case["names"], (float(v) for v in case["theta"][i]))))
        assert "KNtheta" in p
        tobs, lc = model.gen_detector_lc(p)
        np.testing.assert_allclose(tobs, gold[f"s{i}_obs_times"], rtol=1e-15)
        for k, f in enumerate(case["model_filters"]):
            want = gold[f"s{i}_app_{k}"]
            fin = np.isfinite(want)
            assert np.array_equal(np.isfinite(lc[f]), fin)
            np.testing.assert_allclose(lc[f][fin], want[fin], rtol=0, atol=2e-5)
    # batched call: arrays in, [B, NS] out
    p = lik.parameter_conversion({n: case["theta"][:5, j] for j, n in enumerate(case["names"])})
    tobs, lc = model.gen_detector_lc(p)
    assert tobs.shape == (5, len(case["sample_times"])) and lc[case["model_filters"][0]].shape == tobs.shape


def test_pickle_drops_gpu_handle_and_rebuilds():
    case = cases.case_small_hidden()
    _, _, lik = plugin_from_case(case)
    a = lik.log_likelihood_batch(case["theta"], case["names"])
    assert lik.sub_model._engine is not None
    clone = pickle.loads(pickle.dumps(lik))
    assert clone.sub_model._engine is None
    b = clone.log_likelihood_batch(case["theta"], case["names"])
    assert np.array_equal(a, b)


def test_gpu_pool_batches_a_map():
    from nmma_amd.pool import GPUPool
    case = cases.case_c2_default()
    gold = cases.load_golden("c2_default")["logl"]
    _, _, lik = plugin_from_case(case)
    with GPUPool(lik, queue_size=64, names=case["names"]) as pool:
        assert pool.is_master() and pool.size == 64
        res = pool.map(pool.log_likelihood, list(case["theta"]))
        assert pool.n_batches == 1 and pool.n_evals == len(case["theta"])
        assert rel_err(np.array(res), gold).max() <= 1e-6
        assert pool.map(lambda x: x + 1, [1, 2]) == [2, 3]


def test_sharded_evaluator_single_rank():
    import torch
    from nmma_amd.parallel import ShardedEvaluator
    case = cases.case_c2_default()
    _, _, lik = plugin_from_case(case)
    ev = ShardedEvaluator(lambda th: lik.log_likelihood_batch(th, case["names"]))
    th = torch.as_tensor(case["theta"], device="cuda:0")
    out = ev.evaluate(th).cpu().numpy()
    assert rel_err(out, cases.load_golden("c2_default")["logl"]).max() <= 1e-6


def test_legacy_alias():
    from nmma_amd.em.em_likelihood import EMTransientLikelihood, OpticalLightCurve
    assert OpticalLightCurve is EMTransientLikelihood


def test_default_extinction_law_through_the_plugin():
    """A sampled ``Ebv`` with the model's default ``extinction_law`` ("P92_SMC_host", model.py:198-201): the plugin
    passes the filter frequencies (from ``filter_lambdas``) and the law to the engine; results match the oracle's
    restatement of get_extinction_mags.  Without wavelengths for its filters the engine refuses a sampled Ebv."""
    from nmma_amd import _lib as L
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    from oracle import nmma_oracle as orc
    from tests.helpers import SimplePrior, oracle_from_case
    case = cases.SHAPE_CASES["extinction_p92"]()
    lam_m = {f: 2.99792458e8 / nu for f, nu in case["filter_nu0"].items()}
    priors = {n: SimplePrior(0.0, 1.0) for n in case["names"]}
    times, mags, sigmas = case["data"]

    def build(filter_lambdas):
        model = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=case["model_filters"],
                                   model_parameters=case["model_parameters"], sample_times=case["sample_times"],
                                   cosmo_grid=case["cosmo_grid"], filter_lambdas=filter_lambdas)
        handler = FilterSystematicsHandler(case["observed_filters"], error_budget=1.0, light_curve_times=times)
        return EMTransientLikelihood(model, (times, mags, sigmas, 0.0), handler, priors,
                                     filters=case["observed_filters"], detection_limit=case["detection_limit"])

    lik = build(lam_m)
    assert lik.sub_model.light_curve_model.extinction_law == "P92_SMC_host"
    got = lik.log_likelihood_batch(case["theta"], case["names"])
    want = orc.log_likelihood_batch(oracle_from_case(case, use_scipy=False), case["names"], case["theta"])
    floor = want == FLOOR
    assert np.array_equal(got == FLOOR, floor) and (~floor).sum() > 20
    assert rel_err(got[~floor], want[~floor]).max() <= 1e-6
    with pytest.raises(L.NMMAHipError, match="Ebv"):
        build(None).log_likelihood_batch(case["theta"], case["names"])
