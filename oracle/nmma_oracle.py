"""CPU oracle for the NMMA EM light-curve likelihood hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; ``nmma_amd/`` must
never do so (the product path fails loudly when its HIP extension is missing).

This is a from-scratch numpy/scipy *restatement* of the reference algorithm
(`/root/reference/nmma`, v1.0.1).  Every function cites the reference lines it
follows.  It is pinned against the reference's own source run under
``oracle/ref_harness.py`` -- see ``tools/make_golden.py`` (fixtures in
``tests/golden/``) and ``tests/test_oracle_vs_reference.py``.

Parity status
-------------
* pinned from the SVD coefficients ``c`` onward (reference source executed here);
* pinned for ``scipy.stats.truncnorm/norm`` (scipy 1.15.3 installed = pinned range);
* the Keras fp32 forward itself is UNPINNED (keras/tensorflow absent from this
  image; the reference's only KAT for it, ``tests/joint_analysis_pipeline.py:117``,
  is skipped upstream and its data blob is missing): we restate the same
  ``Dense(relu) -> Dense`` graph in fp32 numpy.

Layout conventions shared with the HIP path (plain numpy, no classes required):

``svd_model``  dict  filter -> dict(W1[NP,NH] f32, b1[NH] f32, W2[NH,NC] f32,
               b2[NC] f32, VA[NT,>=NC] f64, mins[NT], maxs[NT], tt[NT],
               param_mins[NP], param_maxs[NP], n_coeff)
``data``       (times{f}, mags{f}, sigmas{f})  detector-frame days since trigger
"""
from __future__ import annotations

import math

import numpy as np
from scipy import special as _sp
from scipy import stats as _st

#: ``np.nan_to_num(-np.inf)`` -- the reference's universal failure value
#: (nmma/core/base.py:82, :181; nmma/em/em_likelihood.py:194, :220, :348)
LOGL_FLOOR = float(np.nan_to_num(-np.inf))

_NORM_PDF_LOGC = math.log(math.sqrt(2.0 * math.pi))  # scipy _continuous_distns._norm_pdf_logC


# ---------------------------------------------------------------------------
# a2: parameter conversion
# ---------------------------------------------------------------------------
def observation_angle_conversion(parameters):
    """nmma/core/conversion.py:119-126.  Mutates and returns ``parameters``."""
    theta_jn = parameters.get("theta_jn", np.arccos(parameters.get("cos_theta_jn", 1.0)))
    theta_jn = np.minimum(theta_jn, np.pi - theta_jn)
    if "KNtheta" not in parameters:
        parameters["KNtheta"] = parameters.get("inclination_EM", theta_jn) * 180.0 / np.pi
    if "inclination_EM" not in parameters:
        parameters["inclination_EM"] = parameters["KNtheta"] / 180.0 * np.pi
    return parameters


def model_parameter_conversion(parameters, model_parameters):
    """nmma/em/model.py:272-286 (LightCurveModelContainer.parameter_conversion).

    Note the reference uses ``key.lstrip("log10_")`` (character-set strip, not a
    prefix removal); restated literally.
    """
    new = observation_angle_conversion(parameters)
    for key in model_parameters:
        if key not in new:
            if key.lstrip("log10_") in new.keys():
                new[key] = np.log10(new[key.lstrip("log10_")])
            elif "log10_" + key in new.keys():
                new[key] = 10 ** new["log10_" + key]
    return new


# ---------------------------------------------------------------------------
# a3: per-sample distance / redshift quantities
# ---------------------------------------------------------------------------
def distance_modulus_nmma(d_lum=1e-5):
    """nmma/core/conversion.py:30-34."""
    return 5.0 * (5 + np.log10(d_lum))


def redshift_from_parameters(parameters, cosmo_grid=None, z_of_dl=None):
    """nmma/core/conversion.py:57-64 + nmma/em/model.py:255-267.

    ``cosmo_grid = (dist_grid[50], z_grid[50])`` is an *input* (astropy's Planck18
    root-find builds it once in the reference, nmma/core/conversion.py:49-55; astropy
    is not installed here -- SURVEY.md section 8c).  A FIXED distance gives a constant
    grid (get_cosmo_grids(d, d)), i.e. np.interp returns z(d_L).  Without a grid the
    reference root-finds z for the sample's distance (get_redshift ->
    luminosity_distance_to_redshift, conversion.py:36-47): ``z_of_dl`` stands in for
    that call (astropy-free callable); zeros only when no distance is present at all.
    """
    if "redshift" in parameters:
        return parameters["redshift"]
    if "luminosity_distance" in parameters and "Hubble_constant" in parameters:
        # core/base.py:161-164 -> cosmology_to_distance (core/conversion.py:57-101): the redshift of the sample's own cosmology,
        # root-found per sample; ``z_of_dl(d_L, H0)`` stands in for the astropy call
        if z_of_dl is None:
            raise ValueError("oracle: a sampled Hubble_constant needs z_of_dl(d_L, H0)")
        return z_of_dl(parameters["luminosity_distance"], parameters["Hubble_constant"])
    if "luminosity_distance" in parameters:
        if cosmo_grid is not None:
            dist_grid, z_grid = cosmo_grid
            return np.interp(parameters["luminosity_distance"], dist_grid, z_grid)
        if z_of_dl is None:
            raise ValueError("oracle: a luminosity_distance needs cosmo_grid or z_of_dl (the reference never uses z = 0 here)")
        return z_of_dl(parameters["luminosity_distance"])
    return 0.0


# ---------------------------------------------------------------------------
# a7: autocomplete_data
# ---------------------------------------------------------------------------
def autocomplete_data(interp_points, ref_points, ref_data, extrapolate="linear", ref_value=np.inf):
    """nmma/em/utils.py:626-677 (spline branch omitted: not on the hot path)."""
    ref_data = np.asarray(ref_data, dtype=float)
    ok = np.isfinite(ref_data)
    if ok.sum() < 2:
        return np.full_like(interp_points, ref_value)
    xr = np.asarray(ref_points)[ok]
    yr = ref_data[ok]
    x = np.atleast_1d(interp_points)
    if isinstance(extrapolate, (float, int)):
        return np.interp(x, xr, yr, left=extrapolate, right=extrapolate)
    if extrapolate == "linear":
        y = np.interp(x, xr, yr)
        lo = x < xr[0]
        hi = x > xr[-1]
        y[lo] = yr[0] + (yr[1] - yr[0]) / (xr[1] - xr[0]) * (x[lo] - xr[0])
        y[hi] = yr[-1] + (yr[-1] - yr[-2]) / (xr[-1] - xr[-2]) * (x[hi] - xr[-1])
        return y
    if extrapolate == "constant":
        return np.interp(x, xr, yr, left=yr[0], right=yr[-1])
    raise ValueError(f"Unknown extrapolation method: {extrapolate}.")


# ---------------------------------------------------------------------------
# a5: surrogate MLP + SVD reconstruction
# ---------------------------------------------------------------------------
def mlp_forward(x64, W1, b1, W2, b2, mode="f32"):
    """Keras ``Dense(NH, relu) -> Dropout(inactive) -> Dense(NC)`` in fp32.

    Architecture: nmma/em/training.py:353-364; call site
    nmma/em/lightcurve_generation.py:198 (``model(np.atleast_2d(x)).numpy()``).
    Keras casts the float64 input to the layer dtype (float32).

    mode="f32"    : numpy fp32 matmuls (BLAS order; closest to what Eigen does)
    mode="f64acc" : fp64 accumulation of the same fp32 operands, rounded to fp32
                    at each layer output -- the order-independent "ideal fp32" value
    """
    x32 = np.atleast_2d(np.asarray(x64, dtype=np.float64)).astype(np.float32)
    if mode == "f32":
        h = np.maximum(x32 @ W1 + b1, np.float32(0))
        return (h @ W2 + b2).astype(np.float32)
    if mode == "f64acc":
        h = np.maximum(x32.astype(np.float64) @ W1.astype(np.float64) + b1.astype(np.float64), 0.0)
        h = h.astype(np.float32).astype(np.float64)
        return (h @ W2.astype(np.float64) + b2.astype(np.float64)).astype(np.float32)
    raise ValueError(mode)


def eval_svd_model(svd_filt, ass_ncoeff, param_list, mlp_mode="f32"):
    """nmma/em/lightcurve_generation.py:180-217 (Keras branch)."""
    n_coeff = min(ass_ncoeff, svd_filt["n_coeff"]) if ass_ncoeff else svd_filt["n_coeff"]
    x = (np.array(param_list, dtype=float) - svd_filt["param_mins"]) / (
        svd_filt["param_maxs"] - svd_filt["param_mins"])
    c = mlp_forward(x, svd_filt["W1"], svd_filt["b1"], svd_filt["W2"], svd_filt["b2"],
                    mode=mlp_mode).T.flatten()
    mag = np.dot(svd_filt["VA"][:, :n_coeff], c)  # c (fp32) promotes to fp64
    mag *= svd_filt["maxs"] - svd_filt["mins"]
    mag += svd_filt["mins"]
    return svd_filt["tt"], mag


def calc_svd_lc(sample_times, param_list, svd_model, mag_ncoeff=None, filters=None, mlp_mode="f32"):
    """nmma/em/lightcurve_generation.py:147-178."""
    if filters is None:
        filters = list(svd_model.keys())
    out = {f: np.full_like(sample_times, np.inf) for f in filters if f not in svd_model}
    for f in filters:
        if f in out:
            continue
        tt, mag = eval_svd_model(svd_model[f], mag_ncoeff, param_list, mlp_mode)
        out[f] = autocomplete_data(sample_times, tt, mag, extrapolate=np.inf)
    return out


# ---------------------------------------------------------------------------
# f3: host-galaxy extinction, Pei (1992) SMC curve
# ---------------------------------------------------------------------------
#: constants of nmma/core/constants.py used by extinctionFactorP92SMC
C_CGS = 29979245800.0
#: dust_extinction.shapes.P92: x_range in 1/micron and the B -> V amplitude conversion
P92_X_RANGE = (1.0 / 1e3, 1.0 / 1e-3)
P92_ABAV = 1.0 / 3.08 + 1.0
#: (amplitude / AbAv, lambda [um], b, n) of the six terms with the SMC values of nmma/em/utils.py:398-421
P92_SMC_TERMS = (
    (185.0, 0.042, 90.0, 2.0),      # BKG
    (27, 0.08, 5.5, 4.0),           # FUV
    (0.005, 0.22, -1.95, 2.0),      # NUV
    (0.010, 9.7, -1.95, 2.0),       # SIL1
    (0.012, 18.0, -1.80, 2.0),      # SIL2
    (0.030, 25.0, 0.0, 2.0),        # FIR
)


def p92_axav(lam_um):
    """A(lambda)/A(V) of dust_extinction.shapes.P92.evaluate with the SMC parameters (the
    third-party package is absent from this image: its published formula -- Pei 1992, eq. 20 -- is
    ``sum_i a_i / ((lam/lam_i)^n_i + (lam/lam_i)^-n_i + b_i)`` summed in the order BKG, FUV, NUV,
    SIL1, SIL2, FIR; PARITY UNPINNED against the package itself)."""
    lam_um = np.asarray(lam_um, dtype=float)
    total = None
    for amp, cen, b, n in P92_SMC_TERMS:
        l_norm = lam_um / cen
        term = (amp * P92_ABAV) / (np.power(l_norm, n) + np.power(l_norm, -1 * n) + b)
        total = term if total is None else total + term
    return total


def extinction_factor_p92_smc(nu, Ebv, z, cutoff_hi=2e16):
    """nmma/em/utils.py:373-428: multiplicative flux factor per observer-frame frequency."""
    nu = np.asarray(nu, dtype=float)
    nu_lo = P92_X_RANGE[0] * 1e4 * C_CGS
    nu_hi = min(cutoff_hi, P92_X_RANGE[1] * 1e4 * C_CGS)
    nu_host = nu * (1 + z)
    opt = (nu_host >= nu_lo) & (nu_host <= nu_hi)
    lam_host_cm = C_CGS / nu_host[opt]
    # the package converts the wavelength to wavenumbers in 1/um and back
    x = 1.0 / (lam_host_cm * 1e4)
    ax_o_av = p92_axav(1.0 / x)
    av = 2.93 * Ebv
    ext = np.ones(nu.shape)
    ext[opt] = np.power(10.0, -0.4 * ax_o_av * av)
    return ext


def extinction_mags_p92_smc(nu_0s, redshift, Ebv):
    """get_extinction_mags (nmma/em/model.py:323-342) for extinction_law = "P92_SMC_host"."""
    nu_0s = np.asarray(nu_0s, dtype=float)
    if Ebv == 0.0:
        return np.zeros_like(nu_0s)
    with np.errstate(divide="ignore"):
        return -2.5 * np.log10(extinction_factor_p92_smc(nu_0s, Ebv, redshift))


# ---------------------------------------------------------------------------
# a4 + a8: SVD light-curve model in the detector frame
# ---------------------------------------------------------------------------
class OracleSVDModel:
    """Restates LightCurveModelContainer + SVDLightCurveModel
    (nmma/em/model.py:175-408, :535-731) for an in-memory ``svd_model``.

    ``ext_mag_func(redshift, Ebv) -> array[len(filters)]`` replaces
    ``get_extinction_mags`` (:323-342) whose dust law lives in the absent
    third-party ``dust_extinction``; default: zeros (the reference's value for
    ``Ebv == 0``, :328-330).
    """

    def __init__(self, model_parameters, svd_model, filters=None, sample_times=None,
                 mag_ncoeff=None, cosmo_grid=None, ext_mag_func=None, mlp_mode="f32", z_of_dl=None):
        self.z_of_dl = z_of_dl          # (d_L, H0) -> z for a sampled Hubble constant (stands in for the astropy root-find)
        self.model_parameters = list(model_parameters)
        self.svd_mag_model = svd_model
        self.filters = list(filters) if filters is not None else list(svd_model.keys())
        self.mag_ncoeff = mag_ncoeff
        self.cosmo_grid = cosmo_grid
        self.ext_mag_func = ext_mag_func
        self.mlp_mode = mlp_mode
        # model.py:655-660: default model_times = training grid of the first filter
        self.model_times = (np.asarray(sample_times, dtype=float) if sample_times is not None
                            else next(iter(svd_model.values()))["tt"])
        self.good_parameters = True

    def parameter_conversion(self, parameters):
        return model_parameter_conversion(parameters, self.model_parameters)

    def em_parameter_setup(self, parameters):
        """model.py:288-303 + SVD combine_lc_params :701-705."""
        self.Ebv = parameters.get("Ebv", 0.0)
        self.luminosity_distance = parameters.get("luminosity_distance", 1e-5)
        self.distmod = distance_modulus_nmma(self.luminosity_distance)
        self.timeshift = parameters.get("timeshift", 0.0)
        self.redshift = redshift_from_parameters(parameters, self.cosmo_grid, self.z_of_dl)
        return [parameters[k] for k in self.model_parameters]

    def generate_lightcurve(self, sample_times, parameters):
        """model.py:707-728."""
        plist = self.em_parameter_setup(parameters)
        return calc_svd_lc(sample_times, plist, self.svd_mag_model, self.mag_ncoeff,
                           self.filters, self.mlp_mode)

    def gen_detector_lc(self, parameters, sample_times=None):
        """model.py:352-404 (gen_detector_lc + combine_detector_data)."""
        if sample_times is None:
            sample_times = self.model_times
        model_lc = self.generate_lightcurve(sample_times, parameters)
        obs_times = sample_times * (1 + self.redshift) + self.timeshift
        if self.ext_mag_func is not None and self.Ebv != 0.0:
            ext = self.ext_mag_func(self.redshift, self.Ebv)
            for e, f in zip(ext, self.filters):
                if f in model_lc:
                    model_lc[f] = model_lc[f] + e
        rc = -2.5 * np.log10(1 + self.redshift)
        lc = {}
        for f, mags in model_lc.items():
            if np.isfinite(mags).sum() >= 2:
                lc[f] = mags + self.distmod + rc
            else:
                lc[f] = np.full_like(obs_times, np.inf)
        return obs_times, lc


# ---------------------------------------------------------------------------
# a9: filter-name maps (static tables; nmma/em/utils.py:478-584)
# ---------------------------------------------------------------------------
_RENAMES = {"B": "g", "R": "z", "F160W": "H", "U": "u", "UVW2": "u", "UVW1": "u", "UVM2": "u"}
_AVERAGES = {"w": ["g", "r", "i"], "o": ["r", "i"], "c": ["g", "r"], "V": ["g", "r"],
             "F606W": ["g", "r"], "I": ["z", "y"], "F814W": ["z", "y"]}


def get_filter_name_mapping(observed_filters, known_filters=()):
    """utils.py:478-546.  ``known_filters``: names that map to themselves
    (the reference's hard-coded list plus every sncosmo bandpass name)."""
    if isinstance(observed_filters, str):
        observed_filters = [observed_filters]
    direct, averaging = {}, {}
    for f in observed_filters:
        if f in _RENAMES:
            direct[f] = _RENAMES[f]
        elif f in known_filters or f.startswith("radio") or f.startswith("X-ray"):
            direct[f] = f
        elif f in _AVERAGES:
            averaging[f] = list(_AVERAGES[f])
        else:
            raise ValueError(f"Unknown filter: {f}. Cannot be processed")
    return direct, averaging


def average_mags(mag, filt):
    """utils.py:566-584."""
    names = _AVERAGES[filt]
    if len(names) == 3:
        return (mag[names[0]] + mag[names[1]] + mag[names[2]]) / 3.0
    return (mag[names[0]] + mag[names[1]]) / 2.0


# ---------------------------------------------------------------------------
# a10: systematics
# ---------------------------------------------------------------------------
class OracleSystematics:
    """The four evaluators of FilterSystematicsHandler (nmma/em/systematics.py:279-296)
    plus the constant error budget (:51, :203-210).

    spec: dict with key ``mode`` in
      "budget"  : {"values": {filt: float}}               (from_budget)
      "param"   : {"name": str}                            (from_param)
      "single"  : {"names": {filt: str}}                   (from_single_params)
      "interp"  : {"nodes": {filt: ([names], time_nodes)}} (from_interpolated_params)
      "mixed"   : "names" and "nodes"                      (from_parameters)
    """

    def __init__(self, spec, filters, light_curve_times):
        self.spec, self.filters, self.t = spec, list(filters), light_curve_times

    def __call__(self, parameters):
        s, m = self.spec, self.spec["mode"]
        if m == "budget":
            return {f: np.full_like(self.t[f], s["values"][f]) for f in self.filters}
        if m == "param":
            return {f: np.full_like(self.t[f], parameters[s["name"]]) for f in self.filters}
        out = {}
        if m in ("single", "mixed"):
            for f, nm in s["names"].items():
                out[f] = np.full_like(self.t[f], parameters[nm])
        if m in ("interp", "mixed"):
            for f, (names, nodes) in s["nodes"].items():
                out[f] = autocomplete_data(self.t[f], nodes, [parameters[p] for p in names],
                                           extrapolate="constant")
        return out


# ---------------------------------------------------------------------------
# a11: likelihood terms
# ---------------------------------------------------------------------------
def _log_gauss_mass_neg_inf(b):
    """scipy.stats._continuous_distns._log_gauss_mass(a=-inf, b) (scipy 1.15):
    b <= 0 -> log_ndtr(b) ; b > 0 -> log1p(-ndtr(-b)).  NaN b -> NaN."""
    b = np.asarray(b, dtype=float)
    out = np.full(b.shape, np.nan)
    left = b <= 0
    out[left] = _sp.log_ndtr(b[left])
    cen = b > 0
    out[cen] = np.log1p(-_sp.ndtr(-b[cen]))
    return out


def truncated_gaussian_logpdf(m_det, loc, scale, upper_lim):
    """Closed form of ``truncnorm.logpdf(m, a=-inf, b=(lim-loc)/scale, loc, scale)``
    (nmma/em/em_likelihood.py:252-256) following scipy's rv_continuous.logpdf:
    invalid args (b NaN or b <= a) -> NaN; x outside [a, b] -> -inf; x NaN -> NaN.
    """
    with np.errstate(invalid="ignore", divide="ignore"):
        b = (upper_lim - loc) / scale
        x = (m_det - loc) / scale
        val = (-(x ** 2) / 2.0 - _NORM_PDF_LOGC) - _log_gauss_mass_neg_inf(b) - np.log(scale)
        bad_args = ~((b > -np.inf) & (scale > 0))         # _argcheck: a < b, scale > 0
        val = np.where(x > b, -np.inf, val)                # outside the support
        val = np.where(np.isnan(x), np.nan, val)
        val = np.where(bad_args, np.nan, val)
    return val


def chisquare_gaussianlog_from_lc_data(est_mag, data_mag, data_sigma, upperlim_sigma, lim=np.inf,
                                       use_scipy=True):
    """nmma/em/em_likelihood.py:224-250."""
    fin = np.isfinite(data_sigma)
    inf = ~fin
    if fin.sum() >= 1:
        if use_scipy:
            b = (lim - est_mag[fin]) / data_sigma[fin]
            terms = _st.truncnorm.logpdf(data_mag[fin], -np.inf, b, loc=est_mag[fin],
                                         scale=data_sigma[fin])
        else:
            terms = truncated_gaussian_logpdf(data_mag[fin], est_mag[fin], data_sigma[fin], lim)
        minus_chisquare = np.sum(terms)
        if np.isnan(minus_chisquare):
            return False, -np.inf
    else:
        minus_chisquare = 0.0
    logsf = np.zeros(2)
    if inf.sum() > 0:
        if use_scipy:
            logsf = _st.norm.logsf(data_mag[inf], est_mag[inf], upperlim_sigma[inf])
        else:
            with np.errstate(invalid="ignore", divide="ignore"):
                logsf = _sp.log_ndtr(-(data_mag[inf] - est_mag[inf]) / upperlim_sigma[inf])
    return minus_chisquare, np.sum(logsf)


# ---------------------------------------------------------------------------
# a1 + a9 + a11: the whole likelihood for one parameter vector
# ---------------------------------------------------------------------------
class OracleLikelihood:
    """EMTransientLikelihood / MultiFilterTransient restated
    (nmma/core/base.py:77-82, :178-182; nmma/em/em_likelihood.py:186-204, :266-352).

    ``use_scipy=True`` evaluates ``scipy.stats.truncnorm/norm`` exactly as the
    reference does (the calling pattern timed as the CPU baseline); ``False`` uses
    the closed forms above (checked equal in tests).
    """

    def __init__(self, model, data, systematics_spec, filters, detection_limit=np.inf,
                 known_filters=None, use_scipy=True):
        self.model = model
        self.times, self.mags, self.sigmas = data
        self.filters = list(filters)
        known = set(known_filters) if known_filters is not None else set(model.filters) | set(filters) - set(_AVERAGES)
        self.direct, self.averaging = get_filter_name_mapping(self.filters, known)
        if isinstance(detection_limit, dict):
            self.detection_limit = {f: float(detection_limit.get(f, np.inf)) for f in self.filters}
        elif isinstance(detection_limit, (list, tuple)):
            self.detection_limit = {f: float(v) for f, v in zip(self.filters, detection_limit)}
        else:
            self.detection_limit = {f: float(detection_limit) for f in self.filters}
        self.systematics = OracleSystematics(systematics_spec, self.filters, self.times)
        self.use_scipy = use_scipy

    # em_likelihood.py:313-335
    def expected_mags(self, obs_times, lc):
        out = {}
        for f in self.filters:
            if f in self.direct:
                out[f] = autocomplete_data(self.times[f], obs_times, lc[self.direct[f]],
                                           extrapolate=np.inf)
            else:
                # NB reference quirk: helper filters are looked up in the map built from
                # the *observed* filters only (em_likelihood.py:330), so every helper
                # band must itself be an observed filter (KeyError otherwise).
                helper = {h: autocomplete_data(self.times[f], obs_times, lc[self.direct[h]],
                                               extrapolate=np.inf)
                          for h in self.averaging[f]}
                out[f] = average_mags(helper, f)
        return out

    def sub_log_likelihood(self, parameters, return_parts=False):
        obs_times, lc = self.model.gen_detector_lc(parameters)
        # em_likelihood.py:305-311
        if (not lc) or any(np.isinf(m).all() for m in lc.values()):
            return LOGL_FLOOR
        est = self.expected_mags(obs_times, lc)
        err = self.systematics(parameters)
        chi_tot, gp_tot, parts = 0.0, 0.0, {}
        for f, e in err.items():                                   # em_likelihood.py:337-352
            sig = np.sqrt(self.sigmas[f] ** 2 + e ** 2)
            chi, gp = chisquare_gaussianlog_from_lc_data(est[f], self.mags[f], sig, e,
                                                         lim=self.detection_limit[f],
                                                         use_scipy=self.use_scipy)
            if chi is False:
                return LOGL_FLOOR
            chi_tot += chi
            gp_tot += gp
            parts[f] = (chi, gp)
        total = chi_tot + gp_tot
        if return_parts:
            return total, parts, est
        return total

    def log_likelihood(self, parameters):
        """core/base.py:77-82 + :178-182 (no Constraint priors: product == 1)."""
        p = self.model.parameter_conversion(dict(parameters))
        if not self.model.good_parameters:
            return LOGL_FLOOR
        v = self.sub_log_likelihood(p)
        if not np.isfinite(v):
            return LOGL_FLOOR
        return float(v)


# ---------------------------------------------------------------------------
# Vectorised variant (NOT what the reference does; used to check big batches fast)
# ---------------------------------------------------------------------------
def log_likelihood_batch(lik: OracleLikelihood, names, theta, fixed=None):
    """Loop of ``log_likelihood`` over the rows of ``theta[B, D]`` (columns = names); ``fixed`` holds the
    delta-function parameters every sample dict also carries (bilby hands them to the likelihood)."""
    out = np.empty(len(theta))
    for i, row in enumerate(theta):
        p = dict(zip(names, (float(v) for v in row)))
        if fixed:
            p.update(fixed)
        out[i] = lik.log_likelihood(p)
    return out


# ---------------------------------------------------------------------------
# a12: combined models (flux addition)
# ---------------------------------------------------------------------------
class OracleCombinedModel:
    """CombinedLightCurveModelContainer.gen_detector_lc + stack_magnitudes
    (nmma/em/model.py:1362-1374, :1411-1459, :1486-1510): union of the sub-models' filters and time grids,
    per-model re-interpolation with +inf outside, per-filter lookup (direct / renamed name, else the mean of the
    helper bands of an averaged filter, else the model does not contribute)."""

    def __init__(self, models):
        self.lc_models = list(models)
        self.filters = []
        for m in self.lc_models:
            for f in m.filters:
                if f not in self.filters:
                    self.filters.append(f)
        self.model_times = np.array(sorted(set().union(*[np.asarray(m.model_times, float).tolist() for m in self.lc_models])))
        # model.py:1370 -- compatible_filters: direct map of the union filters (averaged names are absent from it)
        self.compatible = {f: _RENAMES.get(f, f) for f in self.filters if f not in _AVERAGES}
        self.good_parameters = True

    def parameter_conversion(self, parameters):
        for m in self.lc_models:
            parameters = m.parameter_conversion(parameters)
        return parameters

    def gen_detector_lc(self, parameters, sample_times=None):
        from scipy.special import logsumexp
        per_model, times = [], []
        for m in self.lc_models:
            t, lc = m.gen_detector_lc(parameters, sample_times)
            if not lc:
                return t, lc
            per_model.append(lc)
            times.append(t)
        connected = np.array(sorted(set().union(*[np.asarray(t).tolist() for t in times]))) if sample_times is None else times[-1]
        joint = [{f: autocomplete_data(connected, t, v, extrapolate=np.inf) for f, v in lc.items()}
                 for t, lc in zip(times, per_model)]
        ln10 = np.log(10)
        out = {}
        for f in self.filters:                      # stack_magnitudes, model.py:1490-1510
            terms = []
            for mag in joint:
                try:
                    mag_f = mag[self.compatible[f]]
                except KeyError:
                    if f in _AVERAGES and all(g in mag for g in _AVERAGES[f]):
                        mag_f = np.sum([mag[g] for g in _AVERAGES[f]], axis=0) / float(len(_AVERAGES[f]))
                    else:
                        continue
                terms.append(-2.0 / 5.0 * ln10 * np.array(mag_f))
            out[f] = (-5.0 / 2.0 * logsumexp(terms, axis=0) / ln10) if terms else np.full_like(connected, np.inf)
        return connected, out


class OraclePowerLawModel:
    """Synthetic stand-in for a GRB afterglow (the real one is third-party afterglowpy):
    abs mag = grb_mag0 + 2.5 * grb_slope * log10(t / 1 day), defined for t >= t_start.
    ``hole = (first, last, slope_threshold)``: parameter vectors with ``grb_slope`` above the threshold leave the nodes
    first..last (indices of the requested times) without a value (NaN) -- what the combined model's
    ``autocomplete_data`` (nmma/em/model.py:1440-1448) fills from the finite neighbours."""

    model_parameters = ["grb_mag0", "grb_slope"]

    def __init__(self, filters, sample_times, cosmo_grid=None, t_start=0.3, colour=0.15, hole=None):
        self.filters, self.model_times = list(filters), np.asarray(sample_times, float)
        self.cosmo_grid, self.t_start, self.colour, self.hole = cosmo_grid, t_start, colour, hole
        self.good_parameters = True

    def parameter_conversion(self, parameters):
        return parameters

    def abs_lightcurves(self, parameters, sample_times):
        with np.errstate(divide="ignore"):
            base = parameters["grb_mag0"] + 2.5 * parameters["grb_slope"] * np.log10(sample_times)
        lc = {}
        for k, f in enumerate(self.filters):
            v = base + self.colour * k
            v = np.where(sample_times >= self.t_start, v, np.inf)
            if self.hole is not None and parameters["grb_slope"] > self.hole[2]:
                v = v.copy()
                v[self.hole[0] + (k % 2): self.hole[1] + 1] = np.nan        # (the hole starts one node later in every other filter)
            lc[f] = v
        return lc

    def gen_detector_lc(self, parameters, sample_times=None):
        st = self.model_times if sample_times is None else sample_times
        lc = self.abs_lightcurves(parameters, st)
        d_l = parameters.get("luminosity_distance", 1e-5)
        z = redshift_from_parameters(parameters, self.cosmo_grid)
        obs = st * (1 + z) + parameters.get("timeshift", 0.0)
        rc = -2.5 * np.log10(1 + z)
        out = {f: (v + distance_modulus_nmma(d_l) + rc) if np.isfinite(v).sum() >= 2 else np.full_like(obs, np.inf)
               for f, v in lc.items()}
        return obs, out


def likelihood_from_case(case, use_scipy=True, mlp_mode="f32"):
    """The oracle likelihood of a case dict (nmma_amd.synthetic.make_case / tests.cases): checker side of the parity tests,
    of smoke() and of bench.py's cpu_baseline leg."""
    ext = None
    if case.get("ebv_coeff") is not None:           # A_f = k_f * Ebv (stands in for the absent dust_extinction law)
        coeff = np.array([case["ebv_coeff"][f] for f in case["model_filters"]], float)
        ext = lambda redshift, ebv: coeff * ebv      # noqa: E731
    elif case.get("filter_nu0") is not None:        # the default law, restated (oracle/nmma_oracle.py: P92 SMC, host frame)
        nu0 = np.array([case["filter_nu0"][f] for f in case["model_filters"]], float)
        ext = lambda redshift, ebv: extinction_mags_p92_smc(nu0, redshift, ebv)      # noqa: E731
    model = OracleSVDModel(case["model_parameters"], case["svd"], filters=case["model_filters"],
                               sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"],
                               ext_mag_func=ext, mlp_mode=mlp_mode, z_of_dl=case.get("z_of_dl"))
    return OracleLikelihood(model, case["data"], case["systematics"], case["observed_filters"],
                                detection_limit=case["detection_limit"], use_scipy=use_scipy)
