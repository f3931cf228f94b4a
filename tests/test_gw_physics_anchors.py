"""PROPERTY checks of the gravitational-wave leg -- NOT parity.

The waveform of SURVEY section 8 row f4 (IMRPhenomD_NRTidalv2 behind bilby.gw.likelihood.GravitationalWaveTransient,
nmma/gw/gw_likelihood.py:185-203) cannot be pinned in this image: lalsimulation and bilby are absent, and both the oracle
(oracle/gw_waveform_oracle.py) and the kernel's math (nmma_amd/csrc/gw_math.h) carry fit tables typed from the publications.  The
checks below hold whatever those tables contain -- they anchor the parts of the leg that closed-form physics fixes:

* the low-velocity limit of the phase is the TaylorF2 series, whose leading coefficients are textbook closed forms
  (3 / (128 eta) v^-5 [1 + (3715/756 + 55 eta / 9) v^2 - 16 pi v^3 + ...], v = (pi M f)^(1/3));
* amplitude and phase are C^1 across the whole band (the three IMRPhenomD regions are joined with continuous first derivative by
  construction; a mistyped connection coefficient shows up as a kink);
* the strain scales as 1 / d_L (so <h|h> ~ 1 / d_L^2) and the phase does not depend on d_L;
* detector response: F+^2 + Fx^2 = 1 for a source on the detector's normal whatever the polarisation angle, the (F+, Fx) pair
  rotates by 2 psi, <F+^2> = <Fx^2> = 1/5 over the sky, arrival-time delays are bounded by the Earth's light-crossing time and
  equal -|r| / c towards the detector's own direction.

They run the kernel's own scalar math compiled for the host (tests/hostcheck), on the CPU."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import gw_waveform_oracle as gwo          # detector constants and GMST only (inputs of the projection)
from tests.hostcheck import build as hc_build

NAMES = ["mass_1", "mass_2", "chi_1", "chi_2", "lambda_1", "lambda_2", "luminosity_distance", "theta_jn", "phase", "ra", "dec",
         "psi", "geocent_time"]
T_C = 1187008882.43
MTSUN = 4.925490947641267e-06
BASE = dict(mass_1=1.4, mass_2=1.3, chi_1=0.0, chi_2=0.0, lambda_1=0.0, lambda_2=0.0, luminosity_distance=100.0, theta_jn=0.0,
            phase=0.0, ra=1.0, dec=0.2, psi=0.3, geocent_time=T_C)


@pytest.fixture(scope="module")
def hc():
    lib = C.CDLL(hc_build.build_gw())
    pd = C.POINTER(C.c_double)
    lib.hc_gw_source_doubles.restype = C.c_int
    lib.hc_gw_setup.argtypes = [pd, C.c_double, C.c_int, pd, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, pd]
    lib.hc_gw_eval.argtypes = [pd, pd, C.c_int, pd, pd]
    lib.hc_gw_projection.argtypes = [pd, C.c_int, pd, pd, pd]
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _setup(hc, p, ifos=("H1",), tidal=False, start=T_C - 30.0, geometry=None):
    geo = geometry or {n: gwo.detector_geometry(n) for n in ifos}
    det = np.concatenate([np.concatenate([np.asarray(geo[n][1], float).ravel(), np.asarray(geo[n][0], float)]) for n in ifos])
    x = np.array([p[k] for k in NAMES], float)
    out = np.zeros(hc.hc_gw_source_doubles())
    gref = gwo.greenwich_mean_sidereal_time(T_C)
    rate = (gwo.greenwich_mean_sidereal_time(T_C + 64.0) - gwo.greenwich_mean_sidereal_time(T_C - 64.0)) / 128.0
    hc.hc_gw_setup(_ptr(x), 20.0, int(tidal), _ptr(det), len(ifos), start, T_C, gref, rate, _ptr(out))
    assert out[0] == 1.0
    return out


def _eval(hc, S, f):
    f = np.ascontiguousarray(f, dtype=float)
    amp, ph = np.empty_like(f), np.empty_like(f)
    hc.hc_gw_eval(_ptr(S), _ptr(f), len(f), _ptr(amp), _ptr(ph))
    return amp, np.pi * ph                 # h = amp exp(-i Psi)


@pytest.mark.parametrize("m1, m2", [(0.12, 0.10), (0.30, 0.15), (0.05, 0.05)])
def test_newtonian_limit_of_the_phase_is_taylorf2(hc, m1, m2):
    """Second frequency derivative of the phase (it kills the time and phase offsets the model fixes by its own conventions) for
    sub-solar masses at 20-40 Hz, where v < 0.05: the closed-form TaylorF2 0PN term to a few per cent (the 1PN correction),
    0PN + 1PN + 1.5PN to 1e-4."""
    p = dict(BASE, mass_1=m1, mass_2=m2)
    S = _setup(hc, p)
    M = (m1 + m2) * MTSUN
    eta = m1 * m2 / (m1 + m2) ** 2
    h = 0.25
    for f0 in (20.0, 30.0, 40.0):
        f = f0 + h * np.arange(-2, 3)
        _, psi = _eval(hc, S, f)
        d2 = (-psi[0] + 16 * psi[1] - 30 * psi[2] + 16 * psi[3] - psi[4]) / (12 * h * h)          # five-point stencil, O(h^4)
        v = (math.pi * M * f0) ** (1.0 / 3.0)
        # Psi = 3/(128 eta) sum_k a_k v^(k-5), v ~ f^(1/3): the term k contributes ((k-5)/3)((k-8)/3) a_k v^(k-5) / f^2
        pref = 3.0 / (128.0 * eta) / (f0 * f0)
        t0 = pref * (40.0 / 9.0) * v ** -5
        t2 = pref * 2.0 * (3715.0 / 756.0 + 55.0 * eta / 9.0) * v ** -3
        t3 = pref * (10.0 / 9.0) * (-16.0 * math.pi) * v ** -2
        assert d2 > 0 and abs(d2 / t0 - 1.0) < 12.0 * v * v              # leading order: the chirp, 3/(128 eta) (pi M f)^(-5/3)
        assert abs(d2 / (t0 + t2 + t3) - 1.0) < 40.0 * v ** 4 + 2e-6    # next: 2PN, O(v^4) with a coefficient of order 15


@pytest.mark.parametrize("name, p, tidal", [
    ("bbh", dict(BASE, mass_1=36.0, mass_2=29.0, chi_1=0.3, chi_2=-0.2, luminosity_distance=410.0), False),
    ("heavy_bbh", dict(BASE, mass_1=80.0, mass_2=60.0, chi_1=0.7, chi_2=0.5), False),
    ("unequal", dict(BASE, mass_1=20.0, mass_2=4.0, chi_1=-0.5, chi_2=0.1), False),
    ("bns", dict(BASE, mass_1=1.46, mass_2=1.27, chi_1=0.02, chi_2=-0.01, lambda_1=400.0, lambda_2=600.0), True),
])
def test_amplitude_and_phase_are_c1_across_the_band(hc, name, p, tidal):
    """No kink anywhere between 20 Hz and the point where the amplitude has dropped by 1e-6 (or the 0.2 / M cut): on a uniform
    grid a jump J of the first derivative makes ONE second difference of size J h, against h^2 y'' elsewhere.  The joins of the
    IMRPhenomD regions (Mf = 0.014 and the amplitude peak for the amplitude, Mf = 0.018 and f_RD / 2 for the phase) lie inside
    the scanned band for the black-hole binaries."""
    S = _setup(hc, p, tidal=tidal)
    M = (p["mass_1"] + p["mass_2"]) * MTSUN
    f_hi = min(2048.0, 0.2 / M * 0.999)
    h = 1.0 / 64.0
    f = np.arange(20.0, f_hi, h)
    amp, psi = _eval(hc, S, f)
    live = amp > amp.max() * 1e-6
    last = np.nonzero(live)[0][-1]
    f, amp, psi = f[:last], amp[:last], psi[:last]
    if not tidal:
        assert f[0] < 0.014 / M < f[-1] and f[0] < 0.018 / M < f[-1]          # the joins are in the band
    for y, what in ((np.log(amp), "ln amplitude"), (psi, "phase")):
        d1 = np.diff(y) / h
        jump = np.abs(np.diff(d1))                   # |y'(f + h) - y'(f)| = |second difference| / h
        scale = np.maximum(np.abs(d1[:-1]), np.abs(d1[1:]))
        # smooth: y'' h relative to y' -- curvature scales are >= a few Hz, so this stays below a few 1e-3; a kink of 1e-2 y' shows
        # (the derivative of the phase changes sign where the model puts its time reference: the scale is floored at 5 % of its maximum)
        worst = np.max(jump / np.maximum(scale, 5e-2 * np.max(scale)))
        # and the jumps themselves vary smoothly: no isolated spike three times its neighbours' level
        med = np.maximum(np.convolve(jump, np.ones(9) / 9.0, mode="same"), 1e-12 * np.max(scale))
        spike = np.max((jump / med)[8:-8])
        assert worst < 1e-2 and spike < 3.0, (name, what, worst, spike)


def test_strain_scales_with_inverse_distance(hc):
    """h ~ 1 / d_L at every frequency, phase independent of d_L: <h|h> ~ 1 / d_L^2, <d|h> ~ 1 / d_L (what the distance
    marginalisation of the likelihood relies on)."""
    f = np.geomspace(20.0, 1800.0, 500)
    p = dict(BASE, mass_1=1.46, mass_2=1.27, lambda_1=400.0, lambda_2=600.0, chi_1=0.02)
    a1, p1 = _eval(hc, _setup(hc, dict(p, luminosity_distance=40.0), tidal=True), f)
    a2, p2 = _eval(hc, _setup(hc, dict(p, luminosity_distance=137.5), tidal=True), f)
    live = a1 > 0
    assert live.sum() > 400 and np.array_equal(a2 > 0, live)
    assert np.max(np.abs(a1[live] * 40.0 / (a2[live] * 137.5) - 1.0)) < 1e-13
    assert np.array_equal(p1, p2)
    df = np.gradient(f)
    hh1, hh2 = np.sum(a1 ** 2 * df), np.sum(a2 ** 2 * df)
    assert hh1 / hh2 == pytest.approx((137.5 / 40.0) ** 2, rel=1e-12)


def _normal_and_arms(tensor):
    """The detector normal (null direction of D = (x x - y y) / 2) from the tensor itself."""
    w, v = np.linalg.eigh(np.asarray(tensor, float).reshape(3, 3))
    order = np.argsort(w)
    assert w[order[0]] == pytest.approx(-0.5, abs=5e-3) and w[order[2]] == pytest.approx(0.5, abs=5e-3) and abs(w[order[1]]) < 5e-3
    return v[:, order[1]]


@pytest.mark.parametrize("ifo", ["H1", "L1", "V1"])
def test_antenna_pattern_identities(hc, ifo):
    vertex, tensor = gwo.detector_geometry(ifo)
    n = _normal_and_arms(tensor)
    r = np.asarray(vertex, float)
    if np.dot(n, r) < 0:
        n = -n
    # the detector's normal is the local vertical: within a fraction of a degree of the geocentric direction of the vertex
    assert np.degrees(np.arccos(np.dot(n, r) / np.linalg.norm(r))) < 0.5
    gmst = gwo.greenwich_mean_sidereal_time(T_C)
    ra, dec = (math.atan2(n[1], n[0]) + gmst) % (2 * math.pi), math.asin(n[2])
    k_re, k_im, dt = np.zeros(1), np.zeros(1), np.zeros(1)
    resp = []
    for psi in np.linspace(0.0, math.pi, 13):
        S = _setup(hc, dict(BASE, ra=ra, dec=dec, psi=psi, theta_jn=0.0), ifos=(ifo,))       # face-on: K = F+ - i Fx
        hc.hc_gw_projection(_ptr(S), 1, _ptr(k_re), _ptr(k_im), _ptr(dt))
        resp.append((k_re[0], -k_im[0]))
        # a source on the detector's normal: full response whatever the polarisation angle
        assert k_re[0] ** 2 + k_im[0] ** 2 == pytest.approx(1.0, abs=2e-4)
        # ... and it arrives |r| / c earlier than at the geocentre
        assert dt[0] - 30.0 == pytest.approx(-np.dot(n, r) / 299792458.0, abs=2e-7)
    resp = np.array(resp)
    # (F+, Fx) at psi + pi/4 is (-Fx, F+) at psi up to the sign convention of the rotation: the pair turns by 2 psi
    fp0, fc0 = resp[0]
    for psi, (fp, fc) in zip(np.linspace(0.0, math.pi, 13), resp):
        c, s = math.cos(2 * psi), math.sin(2 * psi)
        assert min(abs(fp - (c * fp0 + s * fc0)) + abs(fc - (-s * fp0 + c * fc0)),
                   abs(fp - (c * fp0 - s * fc0)) + abs(fc - (s * fp0 + c * fc0))) < 1e-9
    # edge-on sources carry no cross polarisation: K is real
    S = _setup(hc, dict(BASE, ra=2.0, dec=-0.3, psi=0.4, theta_jn=math.pi / 2), ifos=(ifo,))
    hc.hc_gw_projection(_ptr(S), 1, _ptr(k_re), _ptr(k_im), _ptr(dt))
    assert abs(k_im[0]) < 1e-15 and abs(k_re[0]) <= 0.5 + 1e-12          # |F+| (1 + 0) / 2


def test_sky_average_of_the_antenna_patterns_is_one_fifth(hc):
    """<F+^2> = <Fx^2> = 1/5 over the sphere and the polarisation angle (an L-shaped interferometer), and delays stay within the
    Earth's light-crossing time."""
    rng = np.random.default_rng(5)
    n = 4000
    ra, dec, psi = rng.uniform(0, 2 * math.pi, n), np.arcsin(rng.uniform(-1, 1, n)), rng.uniform(0, math.pi, n)
    k_re, k_im, dt = np.zeros(1), np.zeros(1), np.zeros(1)
    fp2 = fc2 = 0.0
    for i in range(n):
        S = _setup(hc, dict(BASE, ra=ra[i], dec=dec[i], psi=psi[i], theta_jn=0.0), ifos=("L1",))
        hc.hc_gw_projection(_ptr(S), 1, _ptr(k_re), _ptr(k_im), _ptr(dt))
        fp2 += k_re[0] ** 2
        fc2 += k_im[0] ** 2
        assert abs(dt[0] - 30.0) < 0.02128                # R_earth / c
    # variance of F^2 over the sky is ~0.04: the mean of 4000 draws is good to ~0.003 (1 sigma)
    assert fp2 / n == pytest.approx(0.2, abs=0.012) and fc2 / n == pytest.approx(0.2, abs=0.012)


def test_the_kink_detector_sees_a_per_mille_jump_of_the_derivative():
    """Self-check of the C^1 scan above on a synthetic chirp-like curve: a 1e-3 relative jump of the first derivative trips the
    spike criterion, the smooth curve does not."""
    h = 1.0 / 64.0
    f = np.arange(20.0, 400.0, h)
    y = 1e3 * f ** (-5.0 / 3.0)

    def spike(y):
        d1 = np.diff(y) / h
        jump = np.abs(np.diff(d1))
        scale = np.maximum(np.abs(d1[:-1]), np.abs(d1[1:]))
        med = np.maximum(np.convolve(jump, np.ones(9) / 9.0, mode="same"), 1e-12 * np.max(scale))
        return np.max((jump / med)[8:-8])
    assert spike(y) < 1.1
    k = len(f) // 2
    kinked = y.copy()
    kinked[k:] += 1e-3 * np.gradient(y, h)[k] * (f[k:] - f[k])
    assert spike(kinked) > 3.0
