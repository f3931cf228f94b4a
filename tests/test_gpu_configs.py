"""Every BASELINE.json configuration on the HIP path at its full single-GPU batch: size-independent properties
(determinism, permutation equivariance, independence of the batch composition) plus an oracle spot check of
64 rows.  Config 1 (Me2017, 128 samples) and config 2 (4096) are in test_gpu_models.py / test_gpu_parity.py."""
import numpy as np
import pytest

from tests import cases, cases_combined
from tests.helpers import SimplePrior, engine_from_case, oracle_from_case, plugin_from_case, rel_err

pytestmark = pytest.mark.gpu
FLOOR = -1.7976931348623157e308
LOGL_RTOL = 1e-6


def _properties(fn, theta, n_sub):
    """fn(theta[B, D] numpy) -> logL[B] numpy.  Returns the full-batch result after the property checks."""
    a = fn(theta)
    assert np.array_equal(a, fn(theta))                                   # deterministic
    perm = np.random.default_rng(0).permutation(len(theta))
    assert np.array_equal(fn(theta[perm]), a[perm])                       # row-wise independent
    assert np.array_equal(fn(theta[:n_sub]), a[:n_sub])                   # independent of the batch size / tiling
    return a


def test_config3_combined_kn_plus_grb_at_8192():
    """Bu2019lm + GRB afterglow (power-law stand-in for the third-party afterglowpy curves), 9 filters, B = 8192."""
    import torch
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import CombinedLightCurveModelContainer, ExternalLightCurveModel, SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    from nmma_amd import synthetic as syn
    from oracle import nmma_oracle as orc
    case = cases_combined.case_combined()
    B = 8192
    _, th6 = syn.draw_theta(777, B, cases_combined.NAMES[:6])
    rng = np.random.default_rng(778)
    theta = np.concatenate([th6, rng.uniform(-17.5, -14.0, (B, 1)), rng.uniform(0.8, 1.6, (B, 1))], axis=1)
    kn = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=case["filters"],
                            model_parameters=case["model_parameters"], sample_times=case["sample_times"],
                            cosmo_grid=case["cosmo_grid"])
    grb = ExternalLightCurveModel("PLGRB", case["filters"], case["sample_times"])
    comb = CombinedLightCurveModelContainer([kn, grb], cosmo_grid=case["cosmo_grid"])
    times, mags, sigmas = case["data"]
    priors = {n: SimplePrior(0.0, 1.0) for n in case["names"]}
    handler = FilterSystematicsHandler(case["filters"], error_budget=1.0, light_curve_times=times)
    lik = EMTransientLikelihood(comb, (times, mags, sigmas, 0.0), handler, priors, filters=case["filters"])
    st = case["sample_times"]
    i0, i1 = case["names"].index("grb_mag0"), case["names"].index("grb_slope")

    def ext_curves(th):        # the stand-in's source-frame curves, vectorised (oracle: OraclePowerLawModel.abs_lightcurves)
        with np.errstate(divide="ignore"):
            base = th[:, i0, None] + 2.5 * th[:, i1, None] * np.log10(st)[None, :]
        lc = base[:, None, :] + 0.15 * np.arange(len(case["filters"]))[None, :, None]
        return np.where(st[None, None, :] >= 0.3, lc, np.inf)

    def fn(th):
        return lik.log_likelihood_batch(th, case["names"], external_lc={"PLGRB": torch.as_tensor(ext_curves(th))})

    a = _properties(fn, theta, 1000)
    assert np.all(a <= 0) and np.mean(a > FLOOR) > 0.9
    # two sub-models on one grid: ONE launch (em_logl<.., 7>, nmma_em_loglike_stack2), the likelihood-from-curves engine never built
    assert lik.sub_model._engine2 is not None and lik.sub_model._engine is None
    olik, _ = cases_combined.oracle_likelihood(case, use_scipy=False)
    rows = np.linspace(0, B - 1, 64).astype(int)
    want = orc.log_likelihood_batch(olik, case["names"], theta[rows])
    floor = want == FLOOR
    assert np.array_equal(a[rows] == FLOOR, floor)
    assert rel_err(a[rows][~floor], want[~floor]).max() <= LOGL_RTOL


@pytest.mark.parametrize("batch", [8192, 65536])
def test_config4_shape_full_batch(batch):
    """Bu2022Ye shape (NP = 6), 12 filters x 200 epochs, at B = 8192 and at the whole 65 536 batch of config 4."""
    import torch
    from nmma_amd import synthetic as syn
    from oracle import nmma_oracle as orc
    case = cases.case_c4_shape()
    _, theta = syn.draw_theta(4244, batch, case["names"])
    eng = engine_from_case(case)

    def fn(th):
        out = eng.loglike(torch.as_tensor(th, device="cuda:0")).cpu().numpy()
        eng.check()
        return out

    a = _properties(fn, theta, 3000)
    assert np.all(np.isfinite(a)) and np.all(a <= 0)
    olik = oracle_from_case(case, use_scipy=False)
    rows = np.linspace(0, batch - 1, 64).astype(int)
    want = orc.log_likelihood_batch(olik, case["names"], theta[rows])
    assert rel_err(a[rows], want).max() <= LOGL_RTOL
    eng.close()


def test_config5_joint_em_leg_at_16384():
    """Joint GW + EM (config 5): the EM leg on the GPU at B = 16 384, the GW log-likelihood supplied as a device tensor
    (its arithmetic is third-party bilby / lalsimulation), summed and floored by MultiMessengerLikelihood
    (joint_likelihood.py:62-67).  Checked against the oracle's EM values + the same GW numbers."""
    import torch
    from nmma_amd import synthetic as syn
    from nmma_amd.joint.joint_likelihood import ExternalLogLikelihood, MultiMessengerLikelihood
    from oracle import nmma_oracle as orc
    case = cases.case_c2_default()
    B = 16384
    _, theta = syn.draw_theta(5151, B, case["names"])
    theta[5, 0] = np.nan                                     # a broken sample: EM floor
    _, _, em = plugin_from_case(case)
    rng = np.random.default_rng(5)
    gw = -0.5 * rng.chisquare(4, B) - 30.0
    gw[7] = -np.inf                                          # GW messenger failure -> joint floor
    gw_lh = ExternalLogLikelihood("gw", func=None)
    mm = MultiMessengerLikelihood([gw_lh, em], em.priors)
    th_dev = torch.as_tensor(theta, device="cuda:0")
    got = mm.log_likelihood_batch(th_dev, case["names"], external_logl={"gw": torch.as_tensor(gw, device="cuda:0")})
    assert got.is_cuda and got.shape == (B,)
    got = got.cpu().numpy()
    got_np = mm.log_likelihood_batch(theta, case["names"], external_logl={"gw": gw})
    assert np.array_equal(got, got_np)
    assert got[5] == FLOOR and got[7] == FLOOR
    olik = oracle_from_case(case, use_scipy=False)
    rows = np.concatenate([np.linspace(0, B - 1, 60).astype(int), [5, 7, 8, 9]])
    em_want = orc.log_likelihood_batch(olik, case["names"], theta[rows])
    want = em_want + np.where(np.isfinite(gw[rows]), gw[rows], FLOOR)
    want = np.where(np.isfinite(want) & (want > FLOOR), want, FLOOR)      # joint_likelihood.py:64-67
    floor = want == FLOOR
    assert np.array_equal(got[rows] == FLOOR, floor)
    assert rel_err(got[rows][~floor], want[~floor]).max() <= LOGL_RTOL
    # the per-sample reference API gives the same numbers (GW supplied by a function)
    gw_lh.func = lambda p: float(gw[p["_row"]])
    for r in (0, 11, 4097):
        p = dict(zip(case["names"], (float(v) for v in theta[r])), _row=r)
        assert mm.log_likelihood(p) == pytest.approx(got[r], rel=1e-12)


def test_config5_joint_gw_plus_em_from_parameters_at_16384():
    """BASELINE config 5 from PARAMETERS: IMRPhenomD_NRTidalv2 GW log-likelihood (3 detectors, 128 s at 4096 Hz: 259 585 bins in
    band) + Bu2019lm EM log-likelihood, one joint theta[16 384, 17] whose luminosity_distance and theta_jn feed both messengers,
    summed and floored by MultiMessengerLikelihood (joint/joint_likelihood.py:62-67).  Properties at the full batch + an oracle spot
    check of both legs (the GW oracle restates third-party algorithms: parity unpinned against bilby / lalsimulation)."""
    import time
    import torch
    from nmma_amd import synthetic as syn
    from nmma_amd.gw import GravitationalWaveTransientLikelihood, WaveformGenerator
    from nmma_amd.joint.joint_likelihood import MultiMessengerLikelihood
    from oracle import nmma_oracle as orc
    from tests.gw_helpers import make_case, oracle_loglike_ratio
    from oracle import gw_waveform_oracle as gwo
    em_names = ["luminosity_distance", "KNphi", "theta_jn", "timeshift", "log10_mej_dyn", "log10_mej_wind"]
    case = cases._base(seed=1234, names=em_names)            # config 2's model and photometry, inclination from theta_jn
    _, _, em = plugin_from_case(case)
    gwc = make_case(seed=55, duration=128.0, sampling_frequency=4096.0, minimum_frequency=23.0)
    B = 16384
    gw_names = [n for n in syn.GW_NAMES if n != "phase"]     # phase marginalised, as NMMA's GW170817 runs do
    names = gw_names + [n for n in em_names if n not in gw_names]
    _, th_gw = syn.draw_gw_theta(77, B, centre=gwc["injection"], names=gw_names, width=0.3)
    _, th_em = syn.draw_theta(78, B, em_names)
    theta = np.concatenate([th_gw, th_em[:, [em_names.index(n) for n in names[len(gw_names):]]]], axis=1)
    theta[5, names.index("timeshift")] = np.nan              # EM failure -> joint floor
    theta[7, names.index("chirp_mass")] = np.nan             # GW failure -> joint floor
    theta[9, names.index("lambda_1")] = -5.0                 # unphysical -> GW floor -> joint floor
    priors = {n: SimplePrior(0.0, 1.0) for n in names}
    wg = WaveformGenerator(128.0, 4096.0, waveform_arguments=gwc["waveform_arguments"])
    gw = GravitationalWaveTransientLikelihood(priors, gwc["ifos"], wg, phase_marginalization=True)
    mm = MultiMessengerLikelihood([gw, em], priors)

    def fn(th):
        return mm.log_likelihood_batch(torch.as_tensor(th, device="cuda:0"), names).cpu().numpy()

    fn(theta[:64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = _properties(fn, theta, 1000)
    print(f"config 5 joint GW+EM from parameters: 4 evaluations of <= {B} rows in {time.perf_counter() - t0:.2f} s")
    assert a[5] == FLOOR and a[7] == FLOOR and a[9] == FLOOR and (a > FLOOR).sum() == B - 3
    rows = np.concatenate([np.linspace(0, B - 1, 20).astype(int), [5, 7, 9, 10]])
    olik = oracle_from_case(case, use_scipy=False)
    em_cols = [names.index(n) for n in em_names]
    em_want = orc.log_likelihood_batch(olik, em_names, theta[np.ix_(rows, em_cols)])
    good = np.array([r not in (7, 9) for r in rows])
    gw_want = np.full(len(rows), FLOOR)
    gw_want[good] = oracle_loglike_ratio(gwc, gw_names, theta[np.ix_(rows[good], np.arange(len(gw_names)))], phase_marginalization=True) \
        + gwo.noise_log_likelihood(gwc["oracle_ifos"])
    want = em_want + gw_want
    want = np.where(np.isfinite(want) & (want > FLOOR), want, FLOOR)      # joint_likelihood.py:64-67
    floor = want == FLOOR
    assert np.array_equal(a[rows] == FLOOR, floor)
    err = rel_err(a[rows][~floor], want[~floor])
    print(f"  oracle spot check of {int((~floor).sum())} rows: max rel err {err.max():.3e}")
    assert err.max() <= LOGL_RTOL
    # the per-sample reference API (one dict per call) gives the same numbers
    for r in (0, 4097):
        p = dict(zip(names, (float(v) for v in theta[r])))
        assert mm.log_likelihood(p) == pytest.approx(a[r], rel=1e-9)


def test_constraints_on_the_batch_path_match_per_sample_calls():
    """Constraint priors (core/base.py:67-68): the batched entry point floors exactly the rows the per-sample
    log_likelihood floors."""
    import torch
    from nmma_amd.core.base import Constraint
    case = cases.case_c2_default()
    model, handler, lik = plugin_from_case(case)
    pri = dict(lik.priors)
    # a derived quantity produced by the model's conversion (KNtheta from inclination_EM) and a sampled one
    pri["KNtheta"] = Constraint(minimum=10.0, maximum=60.0, name="KNtheta")
    pri["timeshift"] = SimplePrior(0.0, 1.0)
    lik.priors = pri
    lik.constraints["log10_mej_dyn"] = Constraint(minimum=-2.8, maximum=-1.2, name="log10_mej_dyn")
    theta = case["theta"]
    got = lik.log_likelihood_batch(theta, case["names"])
    got_t = lik.log_likelihood_batch(torch.as_tensor(theta, device="cuda:0"), case["names"]).cpu().numpy()
    assert np.array_equal(got, got_t)
    single = np.array([lik.log_likelihood(dict(zip(case["names"], (float(v) for v in row)))) for row in theta])
    assert np.array_equal(got == FLOOR, single == FLOOR)
    assert 0 < np.sum(got == FLOOR) < len(theta)
    fin = got > FLOOR
    assert rel_err(got[fin], single[fin]).max() <= 1e-12
