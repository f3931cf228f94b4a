"""The scalar fp64 numerics the HIP kernels inline (nmma_amd/csrc/em_math.h), compiled for
the host and checked against numpy / scipy -- the semantics the reference relies on."""
import ctypes as C

import numpy as np
import pytest
from scipy import special, stats

from tests.hostcheck import build as hc_build


@pytest.fixture(scope="module")
def hc():
    lib = C.CDLL(hc_build.build())
    d, i, pd = C.c_double, C.c_int, C.POINTER(C.c_double)
    sig = {"hc_interp_np": [d, pd, pd, i, d, d], "hc_lerp_np": [d] * 5, "hc_ndtr": [d], "hc_log_ndtr": [d],
           "hc_log_gauss_mass_neginf": [d], "hc_log_gauss_mass_tab": [d], "hc_upper_limit_term_tab": [d] * 3, "hc_detection_term_tab": [d] * 5, "hc_detection_term": [d] * 5, "hc_upper_limit_term": [d] * 3,
           "hc_apply_slot": [i, i, d, pd], "hc_distance_modulus": [d], "hc_redshift_correction": [d],
           "hc_extinction_mag": [i, d, d, d]}
    for name, args in sig.items():
        getattr(lib, name).restype = d
        getattr(lib, name).argtypes = args
    return lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_interp_matches_numpy(hc):
    rng = np.random.default_rng(0)
    xp = np.sort(rng.uniform(0, 20, 37))
    fp = rng.normal(size=37)
    xs = np.concatenate([rng.uniform(-2, 22, 400), xp, [xp[0], xp[-1], np.nextafter(xp[-1], 30)]])
    for left, right in ((np.inf, np.inf), (fp[0], fp[-1]), (-1.5, 2.5)):
        want = np.interp(xs, xp, fp, left=left, right=right)
        got = np.array([hc.hc_interp_np(x, _p(xp), _p(fp), len(xp), left, right) for x in xs])
        assert np.array_equal(got, want)          # bit-exact, including the exact-node branches
    assert np.isnan(hc.hc_interp_np(np.nan, _p(xp), _p(fp), len(xp), 0.0, 0.0))


def test_lerp_nan_fallbacks_match_numpy(hc):
    for (x0, x1, y0, y1, x) in ((0, 1, np.inf, 3.0, 0.25), (0, 1, 2.0, np.inf, 0.75), (0, 1, np.inf, np.inf, 0.5),
                                (0, 1, 1.0, 2.0, 0.3)):
        want = np.interp(x, [x0, x1], [y0, y1])
        got = hc.hc_lerp_np(x, x0, x1, y0, y1)
        assert (np.isnan(want) and np.isnan(got)) or got == want


def test_log_ndtr_and_ndtr_match_scipy(hc):
    xs = np.concatenate([np.linspace(-38, 8, 1201), [-1.0, -1.0000001, 0.0, 1e-300, 37.0]])
    got = np.array([hc.hc_log_ndtr(x) for x in xs])
    # host stand-in for erfcx (exp(t^2)*erfc(t)) limits this to ~1e-11; the device uses ocml erfcx
    np.testing.assert_allclose(got, special.log_ndtr(xs), rtol=5e-11, atol=0)
    got = np.array([hc.hc_ndtr(x) for x in xs])
    np.testing.assert_allclose(got, special.ndtr(xs), rtol=2e-13, atol=1e-300)


def test_truncation_mass_from_the_table_matches_scipy(hc):
    """``log_gauss_mass_tab`` -- the polynomial table the general lean task reads for detections under a finite limit -- against
    scipy's ``_log_gauss_mass(-inf, b)`` = ``log_ndtr(b)`` (3e-15 relative; 6e-16 against 30-digit arithmetic) on the table's range [-9.5, 8.5), 0 beyond 8.5 (where scipy's value is
    below half an ulp of every term it is subtracted from), scipy's own formula below -9.5; and the detection term built on it against
    ``truncnorm.logpdf`` like the exact one."""
    b = np.concatenate([np.linspace(-9.5, 8.5, 7001)[:-1], np.arange(-9.5, 8.5, 1.0), np.nextafter(np.arange(-8.5, 8.5, 1.0), -10)])
    got = np.array([hc.hc_log_gauss_mass_tab(x) for x in b])
    want = special.log_ndtr(b)
    assert (np.abs(got - want) / np.maximum(1.0, np.abs(want))).max() <= 3e-15          # (scipy's own log_ndtr is a few ulp off around b = -0.84)
    import mpmath as mp
    mp.mp.dps = 30
    exact = np.array([float(mp.log(mp.ncdf(float(x)))) for x in b[::7]])
    assert (np.abs(got[::7] - exact) / np.maximum(1.0, np.abs(exact))).max() <= 6e-16
    for x in (8.5, 9.0, 37.0, np.inf):
        assert hc.hc_log_gauss_mass_tab(x) == 0.0 and abs(special.log_ndtr(x)) < 0.5 * np.spacing(0.9189385332046727)
    low = np.array([-9.5000001, -10.5, -15.0, -20.0, -37.0])
    np.testing.assert_allclose([hc.hc_log_gauss_mass_tab(x) for x in low], special.log_ndtr(low), rtol=5e-11)
    assert np.isnan(hc.hc_log_gauss_mass_tab(np.nan))
    rng = np.random.default_rng(11)
    m, est, sig = rng.uniform(17, 22, 400), rng.uniform(17, 22, 400), rng.uniform(0.05, 1.5, 400)
    for lim in (20.5, 22.5, 30.0):
        got = np.array([hc.hc_detection_term_tab(a, b_, c, np.log(c), lim) for a, b_, c in zip(m, est, sig)])
        exact = np.array([hc.hc_detection_term(a, b_, c, np.log(c), lim) for a, b_, c in zip(m, est, sig)])
        with np.errstate(divide="ignore"):
            want = stats.truncnorm.logpdf(m, -np.inf, (lim - est) / sig, loc=est, scale=sig)
        fin = np.isfinite(want)
        assert np.array_equal(np.isneginf(got), np.isneginf(want))
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(got[fin], exact[fin], rtol=0, atol=2e-15 * np.abs(exact[fin]).max())
    assert np.isnan(hc.hc_detection_term_tab(18.0, np.inf, 0.3, np.log(0.3), 21.0))
    # an upper limit: norm.logsf(m, est, sigma_sys) (em_likelihood.py:246-249)
    m, est, sig = rng.uniform(17, 22, 600), rng.uniform(14, 25, 600), rng.uniform(0.1, 2.0, 600)
    got = np.array([hc.hc_upper_limit_term_tab(a, b_, c) for a, b_, c in zip(m, est, sig)])
    want = stats.norm.logsf(m, est, sig)
    np.testing.assert_allclose(got, want, rtol=3e-15, atol=3e-15)
    assert hc.hc_upper_limit_term_tab(18.0, -np.inf, 1.0) == -np.inf and hc.hc_upper_limit_term_tab(18.0, np.inf, 1.0) == 0.0
    assert np.isnan(hc.hc_upper_limit_term_tab(18.0, 19.0, 0.0)) and np.isnan(hc.hc_upper_limit_term_tab(np.nan, 19.0, 1.0))


def test_detection_term_matches_truncnorm(hc):
    rng = np.random.default_rng(1)
    m, est = rng.normal(18, 1.5, 500), rng.normal(18, 1.5, 500)
    sig = rng.uniform(0.05, 2.0, 500)
    for lim in (np.inf, 30.0, 19.0, 17.0):
        want = stats.truncnorm.logpdf(m, -np.inf, (lim - est) / sig, loc=est, scale=sig)
        got = np.array([hc.hc_detection_term(a, b, c, np.log(c), lim) for a, b, c in zip(m, est, sig)])
        assert np.array_equal(np.isneginf(got), np.isneginf(want))
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-11)
    # infinite / NaN model magnitude -> NaN (the reference then returns the floor)
    for lim in (np.inf, 20.0):
        for est_bad in (np.inf, np.nan):
            assert np.isnan(hc.hc_detection_term(18.0, est_bad, 0.3, np.log(0.3), lim))
            with np.errstate(invalid="ignore"):
                assert np.isnan(stats.truncnorm.logpdf(18.0, -np.inf, (lim - est_bad) / 0.3, loc=est_bad, scale=0.3))


def test_upper_limit_term_matches_norm_logsf(hc):
    rng = np.random.default_rng(2)
    m, est = rng.normal(18, 2, 300), rng.normal(18, 2, 300)
    sig = rng.uniform(0.1, 2.0, 300)
    want = stats.norm.logsf(m, est, sig)
    got = np.array([hc.hc_upper_limit_term(a, b, c) for a, b, c in zip(m, est, sig)])
    np.testing.assert_allclose(got, want, rtol=1e-11)
    assert hc.hc_upper_limit_term(18.0, np.inf, 1.0) == 0.0 == stats.norm.logsf(18.0, np.inf, 1.0)
    assert np.isnan(hc.hc_upper_limit_term(18.0, np.nan, 1.0))


def test_slots_and_distance(hc):
    row = np.array([0.7, 40.0, -2.0, 0.3])
    assert hc.hc_apply_slot(0, 1, 0.0, _p(row)) == 0.7 * 180.0 / np.pi
    assert hc.hc_apply_slot(0, 2, 0.0, _p(row)) == 0.7 / 180.0 * np.pi
    assert hc.hc_apply_slot(1, 3, 0.0, _p(row)) == pytest.approx(np.log10(40.0), rel=1e-15)
    assert hc.hc_apply_slot(2, 4, 0.0, _p(row)) == pytest.approx(10 ** -2.0, rel=1e-15)
    assert hc.hc_apply_slot(-1, 0, 3.5, _p(row)) == 3.5
    tj = 2.8
    assert hc.hc_apply_slot(0, 5, 0.0, _p(np.array([tj]))) == min(tj, np.pi - tj) * 180.0 / np.pi
    assert hc.hc_apply_slot(3, 6, 0.0, _p(row)) == pytest.approx(np.arccos(0.3) * 180 / np.pi, rel=1e-15)
    assert hc.hc_distance_modulus(40.0) == pytest.approx(5.0 * (5 + np.log10(40.0)), rel=1e-15)
    assert hc.hc_redshift_correction(0.01) == pytest.approx(-2.5 * np.log10(1.01), rel=1e-14)


def test_p92_smc_extinction_matches_oracle(hc):
    """em_math.h:p92_smc_ext_mag against the oracle's restatement of extinctionFactorP92SMC + get_extinction_mags
    (utils.py:373-428, model.py:323-342): inside the curve's range, at both edges, beyond the 2e16 Hz cut-off,
    below the far-infrared end, and at Ebv = 0."""
    from oracle import nmma_oracle as orc
    lam_um = np.concatenate([np.geomspace(0.016, 900.0, 60), [0.0149, 0.0151, 999.0, 1001.0, 0.01, 5e3]])
    nu = 2.99792458e14 / lam_um
    for z in (0.0, 0.0098, 0.21):
        for ebv in (0.05, 0.4, 1.7):
            want = orc.extinction_mags_p92_smc(nu, z, ebv)
            got = np.array([hc.hc_extinction_mag(1, x, 1 + z, ebv) for x in nu])
            inside = np.isfinite(want) & (want != 0)
            assert inside.sum() >= 55
            np.testing.assert_allclose(got, want, rtol=2e-14, atol=2e-15)   # (10^x then log10: absolute, not relative)
        assert all(hc.hc_extinction_mag(1, x, 1 + z, 0.0) == 0.0 for x in nu[:5])
    # known shape of the SMC curve: ~1 near 0.55 um (the fit is normalised in B, converted with the MW A_B/A_V), steep UV rise
    assert orc.p92_axav(0.55) == pytest.approx(1.0, abs=0.06)
    assert 4.0 < orc.p92_axav(0.15) < 7.0
    # the linear law is coeff * Ebv
    assert hc.hc_extinction_mag(0, 3.1, 1.3, 0.25) == 3.1 * 0.25
