"""The scalar math of the gravitational-wave leg (nmma_amd/csrc/gw_math.h, compiled for the host by tests/hostcheck) against the
numpy oracle (oracle/gw_waveform_oracle.py) -- waveform set-up, per-bin amplitude and phase in all three IMRPhenomD regions,
NRTidalv2 terms, detector projection -- and ln I0 against scipy.  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import gw_waveform_oracle as gwo
from tests.hostcheck import build as hc_build


@pytest.fixture(scope="module")
def hc():
    lib = C.CDLL(hc_build.build_gw())
    pd = C.POINTER(C.c_double)
    lib.hc_gw_source_doubles.restype = C.c_int
    lib.hc_gw_setup.argtypes = [pd, C.c_double, C.c_int, pd, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, pd]
    lib.hc_gw_eval.argtypes = [pd, pd, C.c_int, pd, pd]
    lib.hc_gw_projection.argtypes = [pd, C.c_int, pd, pd, pd]
    lib.hc_gw_ln_i0.restype = C.c_double
    lib.hc_gw_ln_i0.argtypes = [C.c_double]
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


NAMES = ["mass_1", "mass_2", "chi_1", "chi_2", "lambda_1", "lambda_2", "luminosity_distance", "theta_jn", "phase", "ra", "dec",
         "psi", "geocent_time"]
T_C = 1187008882.43
SOURCES = {
    "bns": dict(mass_1=1.46, mass_2=1.27, chi_1=0.02, chi_2=-0.01, lambda_1=400.0, lambda_2=600.0, luminosity_distance=40.0,
                theta_jn=2.6, phase=1.3, ra=3.446, dec=-0.408, psi=0.7, geocent_time=T_C),
    "bns_swapped": dict(mass_1=1.2, mass_2=1.6, chi_1=-0.04, chi_2=0.03, lambda_1=900.0, lambda_2=150.0, luminosity_distance=75.0,
                        theta_jn=0.4, phase=5.0, ra=1.0, dec=0.3, psi=2.9, geocent_time=T_C + 0.05),
    "nsbh_like": dict(mass_1=5.0, mass_2=1.4, chi_1=0.4, chi_2=0.0, lambda_1=0.0, lambda_2=500.0, luminosity_distance=200.0,
                      theta_jn=1.1, phase=0.2, ra=5.1, dec=-1.1, psi=0.1, geocent_time=T_C - 0.08),
    "bbh": dict(mass_1=36.0, mass_2=29.0, chi_1=0.3, chi_2=-0.2, lambda_1=0.0, lambda_2=0.0, luminosity_distance=410.0,
                theta_jn=2.0, phase=3.0, ra=2.2, dec=-1.2, psi=1.6, geocent_time=T_C),
    "heavy_bbh": dict(mass_1=80.0, mass_2=60.0, chi_1=0.7, chi_2=0.5, lambda_1=0.0, lambda_2=0.0, luminosity_distance=1000.0,
                      theta_jn=0.8, phase=0.0, ra=0.2, dec=0.9, psi=0.3, geocent_time=T_C),
}


def _setup(hc, p, f_ref=20.0, ifos=("H1", "L1", "V1"), start=T_C - 30.0, tidal=True):
    det = np.concatenate([np.concatenate([gwo.detector_geometry(n)[1].ravel(), gwo.detector_geometry(n)[0]]) for n in ifos])
    x = np.array([p[k] for k in NAMES], float)
    out = np.zeros(hc.hc_gw_source_doubles())
    gref = gwo.greenwich_mean_sidereal_time(T_C)
    rate = (gwo.greenwich_mean_sidereal_time(T_C + 64.0) - gwo.greenwich_mean_sidereal_time(T_C - 64.0)) / 128.0
    hc.hc_gw_setup(_ptr(x), f_ref, int(tidal), _ptr(det), len(ifos), start, T_C, gref, rate, _ptr(out))
    return out


@pytest.mark.parametrize("name", list(SOURCES))
def test_waveform_amplitude_and_phase(name, hc):
    p = SOURCES[name]
    tidal = "bbh" not in name            # the black-hole binaries run plain IMRPhenomD
    S = _setup(hc, p, tidal=tidal)
    assert S[0] == 1.0
    f = np.arange(160, 2048 * 8 + 1) / 8.0          # 20 ... 2048 Hz, df = 1/8
    amp, ph = np.empty_like(f), np.empty_like(f)
    hc.hc_gw_eval(_ptr(S), _ptr(f), len(f), _ptr(amp), _ptr(ph))
    src = gwo.PhenomDNRTidalv2(p["mass_1"], p["mass_2"], p["chi_1"], p["chi_2"], p["lambda_1"], p["lambda_2"], tidal=tidal)
    h = src.h22(f, p["luminosity_distance"], p["phase"], 20.0)
    want_amp = np.abs(h)
    live = want_amp > 0
    assert np.array_equal(amp > 0, live)
    assert np.max(np.abs(amp[live] / want_amp[live] - 1.0)) < 2e-11
    got = amp * np.exp(-1j * np.pi * ph)
    # phase differences are compared on the unit circle (absolute phases reach 1e5 rad: 1e-9 rad is 1e-14 relative)
    err = np.abs(got[live] / h[live] - 1.0)
    assert err.max() < 5e-9, err.max()


def test_regions_are_all_exercised(hc):
    hit = set()
    f = np.arange(160, 2048 * 8 + 1) / 8.0
    for name, p in SOURCES.items():
        src = gwo.PhenomDNRTidalv2(p["mass_1"], p["mass_2"], p["chi_1"], p["chi_2"], p["lambda_1"], p["lambda_2"])
        src.kappa2T = src.kappa2T if "bbh" not in name else 0.0
        Mf = f * src.M_sec
        if np.any(Mf < gwo.AMP_FJOIN_INS): hit.add("amp_ins")
        if np.any((Mf >= gwo.AMP_FJOIN_INS) & (Mf < src.fmax)): hit.add("amp_int")
        if np.any((Mf >= src.fmax) & (Mf <= gwo.F_CUT)): hit.add("amp_mrd")
        if np.any(Mf > gwo.F_CUT): hit.add("cut")
        if np.any((Mf >= gwo.PHI_FJOIN_INS) & (Mf < 0.5 * src.fRD)): hit.add("phi_int")
        if np.any(Mf >= 0.5 * src.fRD): hit.add("phi_mrd")
        if src.kappa2T > 0 and np.any((f > src.f_merger_hz) & (f < 1.2 * src.f_merger_hz)): hit.add("taper")
        if src.kappa2T > 0 and np.any(f >= 1.2 * src.f_merger_hz): hit.add("tapered_out")
    assert hit == {"amp_ins", "amp_int", "amp_mrd", "cut", "phi_int", "phi_mrd", "taper", "tapered_out"}, hit


@pytest.mark.parametrize("name", list(SOURCES))
def test_detector_projection(name, hc):
    p = SOURCES[name]
    ifos = ("H1", "L1", "V1")
    start = T_C - 30.0
    S = _setup(hc, p, ifos=ifos, start=start)
    k_re, k_im, dt = np.zeros(3), np.zeros(3), np.zeros(3)
    hc.hc_gw_projection(_ptr(S), 3, _ptr(k_re), _ptr(k_im), _ptr(dt))
    ci = math.cos(p["theta_jn"])
    for i, n in enumerate(ifos):
        vertex, tensor = gwo.detector_geometry(n)
        fp, fc = gwo.antenna_response(tensor, p["ra"], p["dec"], p["geocent_time"], p["psi"])
        delay = gwo.time_delay_from_geocenter(vertex, p["ra"], p["dec"], p["geocent_time"])
        assert abs(k_re[i] - fp * 0.5 * (1 + ci * ci)) < 1e-11
        assert abs(k_im[i] + fc * ci) < 1e-11
        assert abs(dt[i] - (p["geocent_time"] - start + delay)) < 1e-9      # (GPS seconds ~1e9: 1 ulp is 2e-7 s on the input itself)
        assert abs(delay) < 0.0213                                          # Earth radius / c


def test_invalid_inputs_are_flagged(hc):
    for bad in (dict(mass_1=float("nan")), dict(luminosity_distance=-1.0), dict(chi_1=1.5), dict(lambda_2=-3.0), dict(mass_2=0.0),
                dict(ra=float("inf"))):
        p = dict(SOURCES["bns"], **bad)
        assert _setup(hc, p)[0] == 0.0


def test_ln_bessel_i0_matches_scipy(hc):
    from scipy.special import ive
    x = np.concatenate([np.linspace(0, 30, 601), np.geomspace(1e-8, 1e6, 400), [14.999, 15.0, 15.001]])
    got = np.array([hc.hc_gw_ln_i0(float(v)) for v in x])
    want = np.log(ive(0, x)) + x
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 5e-15
