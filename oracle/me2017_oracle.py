"""CPU oracle for the Me2017 analytic kilonova light curve (BASELINE config 1).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (same rules as oracle/nmma_oracle.py).

Restates ``eff_metzger_lc`` (nmma/em/lightcurve_generation.py:566-652),
``mag_dict_for_blackbody`` / ``bb_flux_from_inv_temp`` (:43-58), ``flux_to_ABmag``
(nmma/em/utils.py:793-811) and ``SimpleKilonovaLightCurveModel.generate_lightcurve``
(nmma/em/model.py:1321-1337).  Pinned against the reference's own source run under
oracle/ref_harness.py (tools/make_golden_me2017.py).

Physical constants: the CODATA-2018 values astropy >= 4 defines, which the reference
reads through ``nmma/core/constants.py:15-35`` (astropy is absent here).
"""
from __future__ import annotations

import numpy as np

from .nmma_oracle import autocomplete_data, distance_modulus_nmma, redshift_from_parameters

MSUN_CGS = 1.988409870698051e33
C_CGS = 2.99792458e10
C_SI = 2.99792458e8
H_CGS = 6.62607015e-27
KB_CGS = 1.380649e-16
SIGSB_CGS = 5.6703744191844314e-05
PC_CGS = 3.085677581491367e18
D_CGS = 10 * PC_CGS
SECONDS_A_DAY = 86400.0

#: wavelengths (m) of the built-in filter names, nmma/em/utils.py:714-719
BUILTIN_LAMBDAS = dict(zip(["u", "g", "r", "i", "z", "y", "J", "H", "K"],
                           1e-10 * np.array([3561.8, 4866.46, 6214.6, 7687.0, 7127.0, 7544.6,
                                             8679.5, 9633.3, 12350.0])))


def flux_to_ABmag(flux, residual_magnitude=-48.6):
    """utils.py:793-811 (cgs units)."""
    good = np.argwhere(flux > 0)
    if len(good) < 2:
        return np.full_like(flux, np.nan)
    m = np.full_like(flux, np.inf)
    m[good] = -2.5 * np.log10(flux[good]) + residual_magnitude
    return m


def bb_flux_from_inv_temp(nu, inv_temp, r_photo):
    """lightcurve_generation.py:43-46."""
    exponent = np.clip(H_CGS * nu * inv_temp / KB_CGS, None, 700)
    return 2.0 * H_CGS / C_CGS ** 2 * nu ** 3 / np.expm1(exponent) * r_photo * r_photo / D_CGS ** 2


def eff_metzger_lc(sample_times, param_dict, nu_host, filters):
    """lightcurve_generation.py:566-652."""
    m0 = 10 ** param_dict["log10_mej"] * MSUN_CGS
    v0 = 10 ** param_dict["log10_vej"] * C_CGS
    beta = param_dict["beta"]
    kappa_r = 10 ** param_dict["log10_kappa_r"]
    t = sample_times * SECONDS_A_DAY
    tprec = len(t)
    mn, ye = 1e-8, 0.1
    xn0max = 1 - 2 * ye
    mprec = 300
    m = np.geomspace(1e-8, m0 / MSUN_CGS, mprec)
    vm = v0 * np.power(m * MSUN_CGS / m0, -1.0 / beta)
    vm[vm > C_CGS] = C_CGS
    tsf = 2 * 0.17 * sample_times ** 0.74
    eth = 0.36 * (np.exp(-0.56 * sample_times) + np.log(1.0 + tsf) / tsf)
    xn0 = xn0max * 2 * np.arctan(mn / m) / np.pi
    xr = 1.0 - xn0
    xn = xn0[:, None] * np.exp(-t[None, :] / 900.0)
    edot = 3.2e14 * xn + 2.1e10 * eth[None, :] * ((t[None, :] / SECONDS_A_DAY) ** (-1.3))
    kappa = 0.4 * (1.0 - xn - xr[:, None]) + kappa_r * xr[:, None]
    ene = np.zeros(mprec - 1)
    lum = np.zeros((mprec - 1, tprec))
    r_photo = np.zeros(tprec)
    dt = t[1:] - t[:-1]
    dm = m[1:] - m[:-1]
    for j in range(tprec - 1):
        tdiff = 0.08 * kappa[:-1, j] * m[:-1] * MSUN_CGS * 3 / (vm[:-1] * C_CGS * t[j] * beta)
        tau = m[:-1] * MSUN_CGS * kappa[:-1, j] / (4 * np.pi * (t[j] * vm[:-1]) ** 2)
        lum_j = ene / (tdiff + t[j] * (vm[:-1] / C_CGS))
        lum[:, j] = lum_j * dm * MSUN_CGS
        ene += dt[j] * (edot[:-1, j] - (ene / t[j]) - lum_j)
        r_photo[j] = vm[np.argmin(np.abs(tau - 1))] * t[j]
    ltot = np.abs(np.sum(lum, axis=0) / 1e20 / 1e20)
    with np.errstate(invalid="ignore", divide="ignore"):
        tobs = 1e10 * (ltot / (4 * np.pi * r_photo ** 2 * SIGSB_CGS)) ** 0.25
        tobs = autocomplete_data(sample_times, sample_times, tobs)
        tobs[tobs <= 0.0] = np.nan
        inv_t = 1.0 / tobs
        inv_t[~np.isfinite(inv_t)] = np.inf
        return {f: flux_to_ABmag(bb_flux_from_inv_temp(nu_host[i], inv_t, r_photo)) for i, f in enumerate(filters)}


class OracleMe2017Model:
    """SimpleKilonovaLightCurveModel("Me2017") + LightCurveModelContainer.gen_detector_lc
    (nmma/em/model.py:1280-1337, :352-404)."""

    model_parameters = ["log10_mej", "log10_vej", "beta", "log10_kappa_r"]

    def __init__(self, filters, sample_times, cosmo_grid=None):
        self.filters = list(filters)
        self.model_times = np.asarray(sample_times, float)
        self.cosmo_grid = cosmo_grid
        self.nu_0s = C_SI / np.array([BUILTIN_LAMBDAS[f] for f in self.filters])
        self.good_parameters = True

    def parameter_conversion(self, parameters):
        from .nmma_oracle import model_parameter_conversion
        return model_parameter_conversion(parameters, self.model_parameters)

    def gen_detector_lc(self, parameters, sample_times=None):
        st = self.model_times if sample_times is None else sample_times
        d_l = parameters.get("luminosity_distance", 1e-5)
        distmod = distance_modulus_nmma(d_l)
        timeshift = parameters.get("timeshift", 0.0)
        z = redshift_from_parameters(parameters, self.cosmo_grid)
        pdict = {k: parameters[k] for k in self.model_parameters}
        with np.errstate(invalid="ignore", divide="ignore"):
            lc = eff_metzger_lc(st, pdict, self.nu_0s * (1 + z), self.filters)
        obs_times = st * (1 + z) + timeshift
        rc = -2.5 * np.log10(1 + z)
        out = {}
        for f, mags in lc.items():
            if np.isfinite(mags).sum() >= 2:
                out[f] = mags + distmod + rc
            else:
                out[f] = np.full_like(obs_times, np.inf)
        return obs_times, out
