"""GPU tests of the reference-shaped plugin classes (the drop-in boundary): the same
golden vectors, driven the way the reference's tests / samplers drive the reference."""
import pickle

import numpy as np
import pytest

from nmma_amd import synthetic as syn
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


def test_at2017gfo_real_photometry_through_build_em_likelihood():
    """The data set the metric is named after: the reference's example_files/lightcurves/AT2017gfo.dat (fixture
    tests/golden/at2017gfo_photometry.npz: 9 filters, 141 rows, 3 upper limits) from raw MJD photometry through the drivers'
    own order of calls -- cut_data_to_time_range, then build_em_likelihood (trigger-relative times, model-window check,
    systematics handler with the sampled em_syserr, likelihood) -- against the golden the reference produced from the same
    arrays; and the uncut data set is refused like the reference refuses it."""
    from nmma_amd.em import utils as em_utils
    from nmma_amd.em.em_likelihood import build_em_likelihood
    from nmma_amd.em.model import SVDLightCurveModel
    from tests.helpers import SimplePrior
    case = cases.case_at2017gfo()
    gold = cases.load_golden("at2017gfo")["logl"]
    priors = {n: SimplePrior(0.0, 1.0) for n in case["names"]}
    for key, (lo, hi) in case["prior_bounds"].items():
        priors[key] = SimplePrior(lo, hi)
    model = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=case["model_filters"],
                               model_parameters=case["model_parameters"], sample_times=case["sample_times"],
                               cosmo_grid=case["cosmo_grid"])
    raw = em_utils.cut_data_to_time_range(cases.at2017gfo_raw_photometry(), None, case["trigger_time"], *case["data_window"])
    lik = build_em_likelihood(model, raw, case["trigger_time"], priors, error_budget=None)
    assert sorted(lik.sub_model.observed_filters) == sorted(case["observed_filters"]) and len(case["observed_filters"]) == 9
    n_ul = sum(int(np.sum(~np.isfinite(lik.sub_model.light_curve_uncertainties[f]))) for f in case["observed_filters"])
    assert n_ul == 3
    got = lik.log_likelihood_batch(case["theta"], case["names"])
    assert np.all(gold > FLOOR) and rel_err(got, gold).max() <= 1e-6
    p = dict(zip(case["names"], (float(v) for v in case["theta"][3])))
    assert lik.log_likelihood(p) == pytest.approx(gold[3], rel=1e-6)
    with pytest.raises(ValueError, match="Last data point"):
        build_em_likelihood(model, cases.at2017gfo_raw_photometry(), case["trigger_time"], priors, error_budget=None)


def test_sampled_hubble_constant_through_the_plugin():
    """priors/Bu2019lm_Hubble.prior: H0 sampled next to d_L.  check_vs_priors tabulates one 256-node z(d_L) grid for the native
    cosmology's own H0 and the device scales every sample's distance by H0 / H0_ref; the oracle root-finds every sample's
    redshift in its own cloned cosmology (radiation density included, which does NOT scale with H0: the stated <= 3e-6
    deviation on z shows up as ~1e-8 on log L)."""
    from nmma_amd import synthetic as syn
    from nmma_amd.core.conversion import native_cosmology
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    from oracle import nmma_oracle as orc
    from tests.helpers import SimplePrior
    case = cases.case_hubble_sampled()
    priors = {n: SimplePrior(0.0, 1.0) for n in case["names"]}
    priors["luminosity_distance"], priors["Hubble_constant"] = SimplePrior(1.0, 200.0), SimplePrior(50.0, 90.0)
    model = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=case["model_filters"],
                               model_parameters=case["model_parameters"], sample_times=case["sample_times"])
    model.check_vs_priors(priors)
    assert len(model.cosmo_grid[0]) == 256 and model.hubble_reference == pytest.approx(67.66)
    times, mags, sigmas = case["data"]
    handler = FilterSystematicsHandler(case["observed_filters"], systematics_file=None, error_budget=1.0, light_curve_times=times)
    lik = EMTransientLikelihood(model, (times, mags, sigmas, 0.0), handler, priors, filters=case["observed_filters"])
    got = lik.log_likelihood_batch(case["theta"], case["names"])
    base = native_cosmology(None)
    # the device's definition of "the sample's cosmology": the reference cosmology's expansion history E(z) with the sample's H0,
    # i.e. z(d_L; H0) = z_ref(d_L H0 / H0_ref) -- exact for matter + Lambda
    case["z_of_dl"] = lambda d, h: base.z_at_luminosity_distance(float(d) * float(h) / base.H0)
    want = orc.log_likelihood_batch(orc.likelihood_from_case(case, use_scipy=False), case["names"], case["theta"])
    assert np.all(want > FLOOR)
    err = rel_err(got, want)
    # ... against astropy-style clone(H0=...), whose photon / neutrino densities scale as 1 / h^2: z moves by
    # <= 3e-6 relative, log L -- which changes by ~1e4 per unit redshift -- by a few 1e-5 relative
    case["z_of_dl"] = lambda d, h: base.clone(H0=float(h)).z_at_luminosity_distance(float(d))
    clone = orc.log_likelihood_batch(orc.likelihood_from_case(case, use_scipy=False), case["names"], case["theta"])
    print(f"sampled H0: max rel err {err.max():.2e}; against per-sample clones with rescaled radiation {rel_err(got, clone).max():.2e}")
    assert err.max() <= 1e-6 and rel_err(got, clone).max() <= 2e-4
    # per-sample reference API: the conversion chain adds the redshift on the host, the device uses it as given
    p = dict(zip(case["names"], (float(v) for v in case["theta"][5])))
    assert lik.log_likelihood(p) == pytest.approx(want[5], rel=1e-6)
    with pytest.raises(Exception, match="Omega_matter"):
        lik.sub_model.engine(case["names"] + ["Omega_matter"])


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


def test_lockstep_walker_drives_the_gpu_likelihood():
    """The sampler seam end to end on the GPU: a queue of nlive ensemble-walk chains handed to ``GPUPool.map`` the way
    dynesty does it -> one batched prior transform and ONE kernel launch per MCMC step; every accepted point's logL equals the
    per-sample evaluation, and chains driven one at a time give the same walk."""
    import collections
    import time
    from nmma_amd.pool import GPUPool
    from nmma_amd.sampler import BatchedPriorTransform, LockstepEnsembleWalk
    from nmma_amd import synthetic as syn
    Args = collections.namedtuple("Args", "u loglstar rseed prior_transform loglikelihood kwargs")

    class Uni:
        def __init__(self, lo, hi):
            self.minimum, self.maximum = lo, hi

        def rescale(self, u):
            return self.minimum + (self.maximum - self.minimum) * np.asarray(u)

    case = cases.case_c2_default()
    _, _, lik = plugin_from_case(case)
    names = list(case["names"])
    # prior box = the range the synthetic thetas were drawn from
    draw = syn.draw_theta(11, 4000, names)[1]
    priors = {n: Uni(float(draw[:, i].min()), float(draw[:, i].max())) for i, n in enumerate(names)}
    pt = BatchedPriorTransform(priors, names)
    pool = GPUPool(lik, queue_size=1024, names=names, prior_transform_many=pt)
    rng = np.random.default_rng(5)
    live_u = rng.random((1024, len(names)))
    live_logl = lik.log_likelihood_batch(pt(live_u), names)
    loglstar = float(np.quantile(live_logl[live_logl > FLOOR], 0.3))
    queue = [Args(u=live_u[i], loglstar=loglstar, rseed=900 + i, prior_transform=pt, loglikelihood=pool.log_likelihood,
                  kwargs={"live_u": live_u}) for i in range(1024)]
    walker = LockstepEnsembleWalk(len(names), walks=25, maxmcmc=100)
    pool.n_batches = pool.n_evals = 0
    t0 = time.perf_counter()
    res = pool.map(walker, queue)
    dt = time.perf_counter() - t0
    assert len(res) == 1024
    # one launch per MCMC step (not per point): at most maxmcmc + 1 launches for 1024 chains
    assert pool.n_batches <= 101 and pool.n_evals >= 1024, (pool.n_evals, pool.n_batches)
    print(f"lock-step walk: {pool.n_evals} evaluations in {pool.n_batches} launches, {dt * 1e3:.1f} ms "
          f"({pool.n_evals / dt:.3g} evals/s through the unmodified pool.map call pattern)")
    # every returned point carries the likelihood of its own parameters and obeys the acceptance rule
    chk = lik.log_likelihood_batch(np.stack([r[1] for r in res]), names)
    assert np.array_equal(chk, np.array([r[2] for r in res]))
    moved = [r for r in res if r[4]["accept"] > 0]
    assert len(moved) >= 16 and all(r[2] > loglstar for r in moved)
    assert all(np.array_equal(r[1], pt(r[0])) for r in res[:50])
    # the per-chain coroutine form (own random stream per chain) equals driving a chain alone, one point per call
    short = LockstepEnsembleWalk(len(names), walks=4, maxmcmc=20)
    ref = short.run_many_chains(queue[:64], pool.log_likelihood_many, pt)
    for i in (0, 17, 63):
        u, v, logl, ncall, blob = short(queue[i])
        assert np.array_equal(u, ref[i][0]) and logl == ref[i][2] and ncall == ref[i][3] and blob == ref[i][4]


def test_model_from_the_references_file_layout_runs_on_the_gpu(tmp_path):
    """``{model}.joblib`` + ``{model}_tf/{filter}.h5`` (nmma/em/model.py:593-696) -> ``SVDLightCurveModel(svd_path=...)`` -> light curves on
    the device: the same bits as the model built from the tensors directly.  The ``.h5`` is one of the reference's trained networks
    (tests/golden/bu2019nsbh_tf_h5/ztfr.h5), read without h5py (em/hdf5_lite.py)."""
    import os
    import shutil
    import joblib
    import torch
    from nmma_amd.em.model import SVDLightCurveModel
    here = os.path.dirname(os.path.abspath(__file__))
    mp, svd = syn.make_svd_model(8634, ["ztfr"], model="Bu2019nsbh")
    with np.load(os.path.join(here, "golden", "bu2019nsbh_tf_weights.npz")) as z:
        for k in ("W1", "b1", "W2", "b2"):
            svd["ztfr"][k] = np.ascontiguousarray(z[f"ztfr/{k}"], dtype=np.float32)
    nt = len(svd["ztfr"]["tt"])
    va_full = np.zeros((nt, nt))
    va_full[:, :10] = svd["ztfr"]["VA"]
    meta = {"ztfr": dict(param_mins=svd["ztfr"]["param_mins"], param_maxs=svd["ztfr"]["param_maxs"], mins=svd["ztfr"]["mins"],
                         maxs=svd["ztfr"]["maxs"], tt=svd["ztfr"]["tt"], n_coeff=10, VA=va_full)}
    root = tmp_path / "models"
    (root / "Bu2019nsbh_tf").mkdir(parents=True)
    joblib.dump(meta, str(root / "Bu2019nsbh.joblib"), compress=9)
    shutil.copy(os.path.join(here, "golden", "bu2019nsbh_tf_h5", "ztfr.h5"), str(root / "Bu2019nsbh_tf" / "ztfr.h5"))
    grid = syn.flat_lcdm_grid(1.0, 200.0)
    from_files = SVDLightCurveModel("Bu2019nsbh_tf", svd_path=str(root), interpolation_type="tensorflow", cosmo_grid=grid)
    direct = SVDLightCurveModel("Bu2019nsbh", svd_mag_model=svd, filters=["ztfr"], model_parameters=mp, cosmo_grid=grid)
    names = ["luminosity_distance", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind"]
    _, theta = syn.draw_theta(8636, 32, names)
    th = torch.as_tensor(theta, device="cuda:0")
    a = from_files.lightcurves_abs(th, names).cpu().numpy()
    b = direct.lightcurves_abs(th, names).cpu().numpy()
    assert a.shape == (32, 1, nt) and np.array_equal(a, b) and np.all(np.isfinite(a))
