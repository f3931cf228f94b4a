"""Seeded synthetic inputs shaped like the reference's production runs.

Shapes follow SURVEY.md section 8d / BASELINE.md section 3:

* SVD surrogate per filter: ``Dense(NP -> 2048, relu) -> Dense(2048 -> 10)`` fp32
  (architecture nmma/em/training.py:353-364; weight statistics measured on the
  reference's own trained nets ``nmma/tests/data/Bu2019nsbh_tf/*.h5``), an
  orthonormal SVD basis ``VA[NT, NC]`` fp64, 211-point training grid
  ``arange(0, 21.1, 0.1)`` (doc/training.md:49).
* Photometry: AT2017gfo filter set and per-filter epoch counts taken from
  ``example_files/lightcurves/AT2017gfo.dat`` (counts only, no data copied),
  sigma ~ U(0.01, 0.2) mag, one upper limit (sigma = inf).
* theta: uniform / sine draws from the ``priors/Bu2019lm.prior`` box.

This module only *creates inputs* (it contains a small numpy forward of the
surrogate, used solely to place synthetic data points on a fiducial light curve).
It is not the oracle and not the product compute path.
"""
from __future__ import annotations

import numpy as np

#: model_parameters of the SVD families used by BASELINE.json's configs
#: (names: nmma/em/model.py:29-87)
MODEL_PARAMETERS = {
    "Bu2019lm": ["log10_mej_dyn", "log10_mej_wind", "KNphi", "KNtheta"],
    "Bu2019nsbh": ["log10_mej_dyn", "log10_mej_wind", "KNtheta"],
    "Bu2022Ye": ["log10_mej_dyn", "vej_dyn", "Yedyn", "log10_mej_wind", "vej_wind", "KNtheta"],
}

#: surrogate input box (the prior box the grids were simulated on)
PARAM_BOX = {
    "log10_mej_dyn": (-3.0, -1.0), "log10_mej_wind": (-3.0, -0.5), "KNphi": (15.0, 75.0),
    "KNtheta": (0.0, 90.0), "vej_dyn": (0.12, 0.25), "Yedyn": (0.15, 0.3),
    "vej_wind": (0.03, 0.15),
}

AT2017GFO_FILTERS = ["ps1::g", "ps1::r", "ps1::i", "ps1::z", "ps1::y", "2massj"]
AT2017GFO_COUNTS = {"ps1::g": 13, "ps1::r": 19, "ps1::i": 20, "ps1::z": 18, "ps1::y": 15,
                    "2massj": 14, "2massh": 17, "2massks": 23, "sdssu": 2}


def make_svd_model(seed, filters, model="Bu2019lm", n_hidden=2048, n_coeff=10, tt=None):
    """Random surrogate with the statistics of a trained one.  Returns
    ``(model_parameters, svd_model)`` with ``svd_model[filt]`` holding
    W1,b1,W2,b2 (f32), VA[NT,NC],mins,maxs,tt,param_mins,param_maxs (f64), n_coeff."""
    rng = np.random.default_rng(seed)
    names = MODEL_PARAMETERS[model]
    n_p = len(names)
    if tt is None:
        tt = np.arange(0.0, 21.1, 0.1)
    n_t = len(tt)
    pmin = np.array([PARAM_BOX[n][0] for n in names])
    pmax = np.array([PARAM_BOX[n][1] for n in names])
    svd = {}
    for f in filters:
        q, _ = np.linalg.qr(rng.standard_normal((n_t, n_t)))
        # leading coefficient large and negative like the trained nets (range [-13, -1])
        b2 = (0.15 * rng.standard_normal(n_coeff)).astype(np.float32)
        svd[f] = dict(
            W1=(0.78 * rng.standard_normal((n_p, n_hidden))).astype(np.float32),
            b1=(0.12 * rng.standard_normal(n_hidden)).astype(np.float32),
            W2=(0.03 * rng.standard_normal((n_hidden, n_coeff))).astype(np.float32),
            b2=b2,
            VA=np.ascontiguousarray(q[:, :n_coeff]),
            mins=-18.0 + 0.1 * rng.random(n_t),
            maxs=-8.0 + 0.1 * rng.random(n_t),
            tt=np.array(tt, dtype=float),
            param_mins=pmin.copy(), param_maxs=pmax.copy(), n_coeff=n_coeff,
        )
    return names, svd


def flat_lcdm_grid(d_min, d_max, n=50, H0=67.66, Om0=0.30966):
    """(dist_grid, z_grid) with the layout of nmma/core/conversion.py:49-55
    (50 geometric redshift nodes spanning [z(d_min), z(d_max)]).  Flat LCDM
    (matter + Lambda) by trapezoid quadrature -- an INPUT to the path, not a
    parity claim about astropy's Planck18."""
    c_kms = 299792.458

    def d_lum(z):
        z = np.atleast_1d(z)
        out = np.empty_like(z)
        for i, zi in enumerate(z):
            zz = np.linspace(0.0, zi, 2049)
            ez = np.sqrt(Om0 * (1 + zz) ** 3 + (1 - Om0))
            out[i] = (1 + zi) * c_kms / H0 * np.trapezoid(1.0 / ez, zz)
        return out

    def z_at(d):
        lo, hi = 0.0, 10.0
        for _ in range(80):
            mid = 0.5 * (lo + hi)
            if d_lum(mid)[0] < d:
                lo = mid
            else:
                hi = mid
        return 0.5 * (lo + hi)

    z_grid = np.geomspace(z_at(d_min), z_at(d_max), n)
    return d_lum(z_grid), z_grid


def _forward_abs_mag(svd_filt, plist):
    """numpy forward of one filter (data synthesis only)."""
    x = ((np.asarray(plist, float) - svd_filt["param_mins"]) /
         (svd_filt["param_maxs"] - svd_filt["param_mins"])).astype(np.float32)
    h = np.maximum(x @ svd_filt["W1"] + svd_filt["b1"], 0)
    c = (h @ svd_filt["W2"] + svd_filt["b2"]).astype(np.float64)
    return (svd_filt["VA"][:, :svd_filt["n_coeff"]] @ c) * (svd_filt["maxs"] - svd_filt["mins"]) + svd_filt["mins"]


def make_photometry(seed, svd, model_parameters, filters=None, counts=None, fiducial=None,
                    t_range=(0.5, 14.0), n_upper_limits=1, upper_limit_filter="ps1::i",
                    cosmo_grid=None):
    """AT2017gfo-shaped photometry placed on the fiducial light curve.
    Returns ``(times, mags, sigmas)`` dicts keyed by filter (days since trigger)."""
    rng = np.random.default_rng(seed)
    filters = list(filters or AT2017GFO_FILTERS)
    counts = counts or AT2017GFO_COUNTS
    if fiducial is None:
        fiducial = dict(log10_mej_dyn=-2.2, log10_mej_wind=-1.3, KNphi=30.0, KNtheta=25.0,
                        vej_dyn=0.2, Yedyn=0.2, vej_wind=0.08,
                        luminosity_distance=40.0, timeshift=0.0)
    plist = [fiducial[k] for k in model_parameters]
    d_l = fiducial.get("luminosity_distance", 40.0)
    z = float(np.interp(d_l, *cosmo_grid)) if cosmo_grid is not None else 0.0
    distmod = 5.0 * (5 + np.log10(d_l)) - 2.5 * np.log10(1 + z)
    times, mags, sigmas = {}, {}, {}
    for f in filters:
        n = counts[f] if isinstance(counts, dict) else int(counts)
        t = np.sort(rng.uniform(t_range[0], t_range[1], n))
        sig = rng.uniform(0.01, 0.2, n)
        tt = svd[f]["tt"]
        m_true = np.interp((t - fiducial.get("timeshift", 0.0)) / (1 + z), tt,
                           _forward_abs_mag(svd[f], plist)) + distmod
        m = m_true + sig * rng.standard_normal(n)
        if f == upper_limit_filter and n_upper_limits:
            idx = rng.choice(n, size=n_upper_limits, replace=False)
            sig[idx] = np.inf
            m[idx] = m_true[idx] - 0.5       # limit brighter than the model: mild penalty
        times[f], mags[f], sigmas[f] = t, m, sig
    return times, mags, sigmas


def draw_theta(seed, batch, names=None):
    """Prior draws (``priors/Bu2019lm.prior`` box): returns ``(names, theta[B, D])``."""
    rng = np.random.default_rng(seed)
    names = list(names or ["luminosity_distance", "KNphi", "inclination_EM", "timeshift",
                           "log10_mej_dyn", "log10_mej_wind"])
    cols = []
    for n in names:
        if n == "luminosity_distance":
            cols.append(rng.uniform(1.0, 200.0, batch))
        elif n == "inclination_EM":                       # bilby Sine prior on [0, pi/2]
            cols.append(np.arccos(1.0 - rng.uniform(0.0, 1.0, batch)))
        elif n == "timeshift":
            cols.append(rng.uniform(-2.0, 0.1, batch))
        elif n.startswith("em_syserr"):
            cols.append(rng.uniform(0.1, 2.0, batch))
        elif n == "Ebv":
            cols.append(rng.uniform(0.0, 0.5, batch))
        elif n == "Hubble_constant":                      # priors/Bu2019lm_Hubble.prior
            cols.append(rng.uniform(50.0, 90.0, batch))
        elif n == "theta_jn":                             # isotropic viewing angle on [0, pi]: folded by the conversion
            cols.append(np.arccos(rng.uniform(-1.0, 1.0, batch)))
        elif n == "cos_theta_jn":
            cols.append(rng.uniform(-1.0, 1.0, batch))
        elif n in ("mej_dyn", "mej_wind"):                # linear masses: the model's log10_ alias applies
            lo, hi = PARAM_BOX["log10_" + n]
            cols.append(10.0 ** rng.uniform(lo, hi, batch))
        elif n in PARAM_BOX:
            lo, hi = PARAM_BOX[n]
            cols.append(rng.uniform(lo, hi, batch))
        else:
            raise KeyError(n)
    return names, np.stack(cols, axis=1)


def make_case(seed=1234, model="Bu2019lm", filters=None, counts=None, batch=64, n_hidden=2048,
          sample_times=None, names=None, upper_limit_filter="ps1::i", t_range=(0.5, 14.0), n_coeff=10, tt=None):
    """One seeded likelihood configuration as a plain dict of inputs (model tensors, photometry, systematics spec, theta):
    the defaults are BASELINE config 2 (Bu2019lm-shaped surrogate, 6 AT2017gfo filters with 13/19/20/18/15/14 epochs, one
    upper limit, sigma_sys = 1 mag, the SVD training grid as sample_times)."""
    filters = list(filters or AT2017GFO_FILTERS)
    mp, svd = make_svd_model(seed, filters, model=model, n_hidden=n_hidden, n_coeff=n_coeff, tt=tt)
    grid = flat_lcdm_grid(1.0, 200.0)
    data = make_photometry(seed + 1, svd, mp, filters=filters, counts=counts,
                               cosmo_grid=grid, upper_limit_filter=upper_limit_filter,
                               t_range=t_range)
    names, theta = draw_theta(seed + 2, batch, names)
    return dict(model=model, model_parameters=mp, svd=svd, model_filters=filters,
                sample_times=sample_times, cosmo_grid=grid, data=data,
                observed_filters=filters, detection_limit=np.inf,
                systematics=dict(mode="budget", values={f: 1.0 for f in filters}),
                systematics_ref=dict(error_budget=1.0, systematics_file=None),
                names=names, theta=theta)




def config2_case():
    """BASELINE config 2: the headline workload of bench.py."""
    return make_case()


# ---------------------------------------------------------------------------------------------------------------------
# Gravitational-wave leg (BASELINE config 5: "GW170817 + AT2017gfo synthetic")
# ---------------------------------------------------------------------------------------------------------------------
#: GW170817-like source (detector-frame masses; GPS time of the event) used as the centre of the synthetic config-5 runs
GW170817_LIKE = dict(chirp_mass=1.1977, mass_ratio=0.87, chi_1=0.02, chi_2=-0.01, lambda_1=400.0, lambda_2=600.0,
                     luminosity_distance=40.0, theta_jn=2.6, phase=1.3, ra=3.44616, dec=-0.408084, psi=0.7,
                     geocent_time=1187008882.43)
GW_NAMES = ["chirp_mass", "mass_ratio", "chi_1", "chi_2", "lambda_1", "lambda_2", "luminosity_distance", "theta_jn", "phase",
            "ra", "dec", "psi", "geocent_time"]


def analytic_psd(frequency, scale=1.0):
    """Smooth advanced-detector-like one-sided PSD [1/Hz] (Sathyaprakash & Schutz 2009, eq. 3.8 shape): an INPUT of the
    synthetic runs, not a claim about any real detector."""
    f = np.maximum(np.asarray(frequency, float), 1.0)
    x = f / 215.0
    return scale * 1e-49 * (x ** -4.14 - 5.0 / (x * x) + 111.0 * (1.0 - x * x + 0.5 * x ** 4) / (1.0 + 0.5 * x * x))


def make_gw_noise(seed, duration, sampling_frequency, ifo_names=("H1", "L1", "V1")):
    """Coloured Gaussian noise per detector on ``f_k = k / duration``: returns ``(frequency_array, {name: (noise, psd)})``
    with ``<|n_k|^2> = S_k T / 2`` (bilby's convention for frequency-domain strain)."""
    rng = np.random.default_rng(seed)
    n = int(round(duration * sampling_frequency)) // 2 + 1
    freq = np.arange(n) / float(duration)
    out = {}
    for i, name in enumerate(ifo_names):
        psd = analytic_psd(freq, scale=2.5 if name == "V1" else 1.0 + 0.1 * i)
        sigma = np.sqrt(psd * duration / 4.0)
        noise = sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        noise[0] = 0.0
        out[name] = (noise, psd)
    return freq, out


def draw_gw_theta(seed, batch, centre=None, names=None, width=1.0):
    """Parameter vectors scattered around ``centre`` the way a late-stage live-point set is (every row a physically valid
    binary): returns ``(names, theta[B, D])``."""
    rng = np.random.default_rng(seed)
    c = dict(centre or GW170817_LIKE)
    names = list(names or GW_NAMES)
    spread = dict(chirp_mass=2e-4, mass_ratio=0.08, chi_1=0.03, chi_2=0.03, lambda_1=300.0, lambda_2=400.0,
                  luminosity_distance=12.0, theta_jn=0.35, cos_theta_jn=0.2, phase=np.pi, ra=0.08, dec=0.08, psi=np.pi / 2,
                  geocent_time=2e-3, mass_1=0.05, mass_2=0.05)
    cols = []
    for n in names:
        v = c[n] + width * spread[n] * rng.uniform(-1.0, 1.0, batch)
        if n == "mass_ratio":
            v = np.clip(v, 0.4, 1.0)
        elif n in ("lambda_1", "lambda_2"):
            v = np.clip(v, 0.0, 5000.0)
        elif n == "theta_jn":
            v = np.clip(v, 0.0, np.pi)
        elif n == "cos_theta_jn":
            v = np.clip(v, -1.0, 1.0)
        elif n == "luminosity_distance":
            v = np.clip(v, 5.0, None)
        cols.append(v)
    return names, np.stack(cols, axis=1)


def make_gw_interferometers(seed, duration, sampling_frequency, ifo_names=("H1", "L1", "V1"), injection=None,
                            minimum_frequency=20.0, reference_frequency=20.0, approximant="IMRPhenomD_NRTidalv2",
                            post_trigger=2.0, device=0):
    """Synthetic detector data for the GW leg WITHOUT the oracle: coloured noise plus an injection whose strain the device
    path itself generates (a first engine over signal-free data evaluates ``nmma_gw_strain`` for the injected parameters).
    Needs a HIP device.  Returns ``(interferometers, waveform_arguments, injection)``."""
    from .gw import GWEngine, Interferometer
    inj = dict(injection or GW170817_LIKE)
    start = inj["geocent_time"] + post_trigger - duration
    freq, noise = make_gw_noise(seed, duration, sampling_frequency, ifo_names)
    wa = dict(waveform_approximant=approximant, reference_frequency=reference_frequency, minimum_frequency=minimum_frequency)
    blank = [Interferometer(n, np.zeros(len(freq), complex), noise[n][1], duration, start, minimum_frequency=minimum_frequency,
                            sampling_frequency=sampling_frequency) for n in ifo_names]
    names = list(inj)
    eng = GWEngine(blank, names, waveform_arguments=wa, device=device)
    h = eng.strain(np.array([[inj[k] for k in names]])).cpu().numpy()[0]
    eng.close()
    ifos = [Interferometer(n, noise[n][0] + h[i], noise[n][1], duration, start, minimum_frequency=minimum_frequency,
                           sampling_frequency=sampling_frequency) for i, n in enumerate(ifo_names)]
    return ifos, wa, inj
