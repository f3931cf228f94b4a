"""Light-curve model plugin API (``nmma/em/model.py:175-408``) and the SVD surrogate
model (:535-731) evaluated on the GPU.

``SVDLightCurveModel`` keeps the reference's constructor and attributes
(``model, model_parameters, filters, model_times, good_parameters, svd_mag_model``) and
its methods (``check_vs_priors, parameter_conversion, gen_detector_lc,
generate_lightcurve, em_parameter_setup``); the arithmetic of ``calc_svd_lc`` /
``eval_svd_model`` / ``combine_detector_data`` runs in the HIP kernels through
:class:`nmma_amd.engine.EMEngine`.
"""
from __future__ import annotations

import os

import numpy as np

from ..core.conversion import get_cosmo_grids, observation_angle_conversion
from . import io as em_io

#: surrogate inputs per model family, order matters (nmma/em/model.py:29-125)
model_parameters_dict = {
    "Bu2019nsbh": ["log10_mej_dyn", "log10_mej_wind", "KNtheta"],
    "Bu2019lm": ["log10_mej_dyn", "log10_mej_wind", "KNphi", "KNtheta"],
    "Bu2019lm_sparse": ["log10_mej_dyn", "log10_mej_wind"],
    "Ka2017": ["log10_mej", "log10_vej", "log10_Xlan"],
    "Bu2022mv": ["log10_mej_dyn", "vej_dyn", "log10_mej_wind", "vej_wind", "KNtheta"],
    "Bu2022Ye": ["log10_mej_dyn", "vej_dyn", "Yedyn", "log10_mej_wind", "vej_wind", "KNtheta"],
    "Bu2023Ye": ["log10_mej_dyn", "vej_dyn", "Yedyn", "log10_mej_wind", "vej_wind", "Yewind", "KNtheta"],
    "LANL2022": ["log10_mej_dyn", "vej_dyn", "log10_mej_wind", "vej_wind", "KNtheta"],
    "AnBa2022_sparse": ["mrp", "xmix"],
    "AnBa2022_log": ["log10_mtot", "log10_mni", "vej", "log10_mrp", "xmix"],
    "AnBa2022_linear": ["mtot", "mni", "vej", "mrp", "xmix"],
}
for _n in ("LANLTP1", "LANLTP2", "LANLTS1", "LANLTS2"):
    model_parameters_dict[_n] = list(model_parameters_dict["LANL2022"])


class LightCurveModelContainer:
    """Parent class of light-curve models (model.py:175-408): the host-side protocol."""

    extinction_law = "P92_SMC_host"

    def __init__(self, model, filters=None, model_parameters=None, sample_times=None):
        if model_parameters is None:
            assert model in model_parameters_dict, f"{model} unknown, please pass model_parameters"
            self.model_parameters = list(model_parameters_dict[model])
        else:
            self.model_parameters = list(model_parameters)
        self.model = model
        if isinstance(filters, str):
            filters = filters.split(",")
        self.filters = filters
        self.good_parameters = True
        self.cosmo_grid = None
        self.model_times = np.asarray(sample_times, float) if sample_times is not None else self.setup_model_times()

    def __repr__(self):
        return self.__class__.__name__ + f"(model={self.model})"

    def setup_model_times(self, tmin=0.01, tmax=14.0, nsteps=150):
        return np.geomspace(tmin, tmax, nsteps)

    def check_vs_priors(self, priors):
        """model.py:247-267: warn about missing parameters; build the z(d_L) grid once."""
        for key in self.model_parameters:
            if key not in priors and key != "KNtheta":
                print(f"Parameter {key} not found in priors, might fail.")
        if "redshift" not in priors and "luminosity_distance" in priors:
            pr = priors["luminosity_distance"]
            lo, hi = getattr(pr, "minimum", None), getattr(pr, "maximum", None)
            if lo is not None and hi is not None and self.cosmo_grid is None and np.isfinite([lo, hi]).all():
                cosmology = getattr(pr, "cosmology", None)
                h0 = priors["Hubble_constant"] if "Hubble_constant" in priors else None
                h_lo, h_hi = getattr(h0, "minimum", None), getattr(h0, "maximum", None)
                if h0 is not None and h_lo is not None and h_hi is not None and h_hi > h_lo:
                    # a sampled Hubble constant: one finer grid for the reference H0, wide enough for d_L * H0 / H0_ref over the
                    # whole prior box (the device scales the distance per sample; 256 nodes = what the kernel stages in LDS)
                    from ..core.conversion import native_cosmology
                    self.hubble_reference = float(native_cosmology(cosmology).H0)
                    self.cosmo_grid = get_cosmo_grids(lo * h_lo / self.hubble_reference, hi * h_hi / self.hubble_reference,
                                                      cosmology, n=256)
                else:
                    # hi == lo (a DeltaFunction): a constant grid, np.interp then returns z(d_L) -- the engine
                    # applies that redshift as a constant (never z = 0 for a fixed distance)
                    self.cosmo_grid = get_cosmo_grids(lo, hi, cosmology)

    def sanity_checks(self, parameters):
        self.good_parameters = True

    def parameter_conversion(self, parameters):
        """model.py:272-286."""
        new = observation_angle_conversion(parameters)
        for key in self.model_parameters:
            if key not in new:
                if key.lstrip("log10_") in new.keys():
                    new[key] = np.log10(new[key.lstrip("log10_")])
                elif "log10_" + key in new.keys():
                    new[key] = 10 ** new["log10_" + key]
        self.sanity_checks(new)
        return new

    @property
    def citation(self):
        return {self.model: []}


class SVDLightCurveModel(LightCurveModelContainer):
    """SVD surrogate light-curve model on the GPU (reference: model.py:535-731).

    ``svd_path`` may be (i) a flat ``.npz`` written by :func:`nmma_amd.em.io.save_svd_model`,
    (ii) a directory holding ``{model}.npz``, or (iii) the reference's own layout
    (``{model}.joblib`` + per-filter ``.keras/.h5``), converted on the fly when
    joblib + keras/h5py are importable.  ``svd_mag_model`` (a dict of tensors) may also
    be passed directly.
    """

    def __init__(self, model, svd_path=None, svd_mag_ncoeff=None, svd_lbol_ncoeff=None,
                 interpolation_type="keras", model_parameters=None, filters=None, sample_times=None,
                 local_only=True, svd_mag_model=None, cosmo_grid=None, ebv_coeff=None, device=0,
                 extinction_law=None, filter_lambdas=None, **em_model_kwargs):
        comps = model.split("_")
        if "tf" in comps:
            comps.remove("tf")
        core = "_".join(comps)
        self.mag_ncoeff = svd_mag_ncoeff
        self.lbol_ncoeff = svd_lbol_ncoeff
        self.interpolation_type = interpolation_type
        if interpolation_type not in ("keras", "tensorflow", "torch", "jax"):
            raise ValueError("nmma_amd evaluates the neural-network surrogates only "
                             "(--interpolation-type keras/tensorflow); GP surrogates are out of scope")
        self.svd_path = svd_path
        file_params = None
        if svd_mag_model is None:
            svd_mag_model, file_params = self._load(core, svd_path, filters, interpolation_type)
        self.svd_mag_model = {k.replace("_", ":") if "::" not in k and "__" in k else k: v
                              for k, v in svd_mag_model.items()}
        self.svd_lbol_model = None
        if model_parameters is None and file_params is not None and core not in model_parameters_dict:
            model_parameters = file_params
        super().__init__(core, filters, model_parameters, sample_times)
        if self.filters is None:
            self.filters = list(self.svd_mag_model.keys())
        missing = [f for f in self.filters if f not in self.svd_mag_model]
        if missing:
            print(f"Warning: no surrogate for filters {missing}; they evaluate to +inf "
                  "(nmma/em/lightcurve_generation.py:168-169)")
        self.cosmo_grid = cosmo_grid
        self.ebv_coeff = ebv_coeff
        if extinction_law is not None:
            self.extinction_law = extinction_law          # class default: "P92_SMC_host" (model.py:198-201)
        # effective wavelengths [m] -> filter frequencies (model.py:223-224); sncosmo bandpass names have none
        # built in and need filter_lambdas
        lam = dict(BUILTIN_FILTER_LAMBDAS)
        lam.update(filter_lambdas or {})
        self.filter_nu0 = ({f: C_SI / lam[f] for f in self.filters} if all(f in lam for f in self.filters) else None)
        self.device = device
        self._lc_engine, self._lc_names = None, None

    @staticmethod
    def _load(core, svd_path, filters, interpolation_type):
        if svd_path is None:
            raise ValueError("svd_path (or svd_mag_model) is required: model download is out of scope")
        if os.path.isfile(svd_path):
            return em_io.load_svd_model(svd_path)
        flat = os.path.join(svd_path, f"{core}.npz")
        if os.path.isfile(flat):
            return em_io.load_svd_model(flat)
        if os.path.isfile(os.path.join(svd_path, f"{core}.joblib")):
            itype = "tensorflow" if os.path.isdir(os.path.join(svd_path, f"{core}_tf")) else interpolation_type
            return em_io.convert_reference_model(svd_path, core, flat, filters, itype), None
        raise ValueError(f"Model file not found under {svd_path}")

    def setup_model_times(self):
        return next(iter(self.svd_mag_model.values()))["tt"]

    def __repr__(self):
        return super().__repr__() + f"(model={self.model}, svd_path={self.svd_path})"

    # ---- engine plumbing ------------------------------------------------------------
    @property
    def gpu_filters(self):
        """model filters that have a surrogate (the others are all-inf in the reference)."""
        return [f for f in self.filters if f in self.svd_mag_model]

    def engine_kwargs(self):
        kw = dict(svd_model=self.svd_mag_model, model_filters=self.gpu_filters,
                  model_parameters=self.model_parameters,
                  sample_times=None if self._default_times() else self.model_times,
                  cosmo_grid=self.cosmo_grid, ebv_coeff=self.ebv_coeff, device=self.device,
                  n_coeff=self.mag_ncoeff, hubble_reference=getattr(self, "hubble_reference", None))
        # extinction (get_extinction_mags, model.py:323-342): caller-supplied linear coefficients win; else the
        # default host-frame SMC law is evaluated natively when the filter frequencies are known
        if self.ebv_coeff is None and self.extinction_law == "P92_SMC_host" and self.filter_nu0 is not None:
            kw.update(extinction_law="P92_SMC_host", filter_nu0=self.filter_nu0)
        return kw

    def lightcurves_abs(self, theta, names):
        """Source-frame absolute magnitudes [B, M, NS] (calc_svd_lc for every row of theta)."""
        from ..engine import EMEngine
        names = list(names)
        if self._lc_engine is None or names != self._lc_names:
            self._lc_engine = EMEngine(parameter_names=names, **self.engine_kwargs())
            self._lc_names = names
        return self._lc_engine.model_lightcurves(theta)

    def _default_times(self):
        tt = next(iter(self.svd_mag_model.values()))["tt"]
        return len(self.model_times) == len(tt) and np.array_equal(self.model_times, tt)

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_lc_engine"], state["_lc_names"] = None, None      # GPU handles never travel
        return state

    # ---- reference API: per-sample light curves ---------------------------------------
    def gen_detector_lc(self, parameters=None, sample_times=None):
        """(obs_times[NS], {filt: mag_app[NS]}) -- model.py:352-404.  ``parameters`` may hold
        scalars (one light curve) or equal-length arrays (a batch: mags are [B, NS])."""
        from ..engine import EMEngine
        import torch
        if sample_times is not None and not np.array_equal(sample_times, self.model_times):
            self.model_times = np.asarray(sample_times, float)
            self._lc_engine = None
        parameters = _with_host_redshift(parameters, self.cosmo_grid)
        names = sorted(k for k, v in parameters.items() if np.ndim(v) <= 1 and _is_number(v))
        if self._lc_engine is None or names != self._lc_names:
            if self._lc_engine is not None:
                self._lc_engine.close()
            self._lc_engine = EMEngine(parameter_names=names, **self.engine_kwargs())
            self._lc_names = names
        cols = [np.atleast_1d(np.asarray(parameters[k], float)) for k in names]
        n = max(len(c) for c in cols)
        theta = np.stack([np.broadcast_to(c, (n,)) for c in cols], axis=1)
        tobs, mag = self._lc_engine.lightcurves(torch.as_tensor(theta))
        tobs, mag = tobs.cpu().numpy(), mag.cpu().numpy()
        scalar = all(np.ndim(parameters[k]) == 0 for k in names)
        lc = {}
        gi = {f: i for i, f in enumerate(self.gpu_filters)}
        for f in self.filters:
            if f in gi:
                lc[f] = mag[0, gi[f]] if scalar else mag[:, gi[f]]
            else:
                lc[f] = np.full(tobs.shape[1] if scalar else tobs.shape, np.inf)
        return (tobs[0] if scalar else tobs), lc

    def generate_lightcurve(self, sample_times, parameters, filters="all"):
        """Absolute-magnitude light curves (model.py:707-728): the detector-frame result with
        distance / redshift / timeshift neutralised."""
        p = dict(parameters)
        p.update(timeshift=0.0, Ebv=0.0)
        p.pop("redshift", None)
        p.pop("luminosity_distance", None)        # absent: 10 pc and z = 0 (model.py:291-293, conversion.py:57-64)
        grid, self.cosmo_grid = self.cosmo_grid, None
        try:
            self._lc_engine = None
            _, lc = self.gen_detector_lc(p, sample_times)
        finally:
            self.cosmo_grid = grid
            self._lc_engine = None
        if filters not in ("all", None):
            lc = {f: lc[f] for f in filters}
        return lc


def _with_host_redshift(parameters, cosmo_grid):
    """Without a z(d_L) grid the reference root-finds the redshift of every sample's distance
    (get_redshift, conversion.py:36-47, :57-64); do the same on the host and hand it over as a
    ``redshift`` column."""
    if "redshift" in parameters or "luminosity_distance" not in parameters or cosmo_grid is not None:
        return parameters
    from ..core.conversion import native_cosmology
    cosmo = native_cosmology()
    d = np.asarray(parameters["luminosity_distance"], float)
    z = np.array([cosmo.z_at_luminosity_distance(x) for x in np.atleast_1d(d)])
    out = dict(parameters)
    out["redshift"] = float(z[0]) if d.ndim == 0 else z
    return out


def _is_number(v):
    try:
        np.asarray(v, dtype=float)
        return True
    except (TypeError, ValueError):
        return False


#: model names the reference's factory resolves to classes that are not surrogates (model.py:1572-1579); ``Me2017`` is built here
HOST_ONLY_MODELS = ("TrPi2018", "Piro2021", "PL_BB_fixedT", "Sr2023", "Arnett")


def create_light_curve_model_from_args(em_transient, args, filters=None, sample_times=None, host_models=None):
    """Factory with the reference's shape (model.py:1617-1668, :1591-1614): ``em_transient`` is a model name or a comma-separated
    list / list of names; EVERY sub-model gets the same ``filters`` and ``sample_times = setup_sample_times(args)`` (which is what
    makes a two-model combination eligible for the one-launch likelihood); more than one name gives a
    :class:`CombinedLightCurveModelContainer`.  Names: ``Me2017`` -> the analytic kilonova on the device; a key of ``host_models``
    -> an :class:`ExternalLightCurveModel` around that object's ``generate_lightcurve`` (a model whose arithmetic is third-party
    code and stays on the host: the reference's ``GRBLightCurveModel`` for ``TrPi2018``, its supernova / shock-cooling / host-galaxy
    models); anything else is an SVD surrogate, as in the reference's own fall-through (model.py:1585-1587)."""
    from . import utils
    names = em_transient.split(",") if isinstance(em_transient, str) else list(em_transient)
    names = [n.strip() for n in names]
    if filters is None:
        filters = utils.set_filters(args)          # (model.py:1618-1619: --filters / --em-detectors)
    if sample_times is None:
        sample_times = utils.setup_sample_times(args)
    host_models = host_models or {}
    # names the reference maps to models whose arithmetic is third-party or not built here (model.py:1572-1586: afterglowpy, the
    # shock-cooling / host-galaxy / bolometric / power-law black-body models): they need the host object -- falling through to the SVD
    # surrogate would fail later with an unrelated missing-file error
    missing = [n for n in names if n in HOST_ONLY_MODELS and n not in host_models]
    if missing:
        raise ValueError(f"model(s) {missing} are evaluated on the host: pass host_models={{name: model}} -- an object with the reference's "
                         "generate_lightcurve(sample_times, parameters) and model_parameters -- the likelihood then takes their curves as an operand")
    law = getattr(args, "em_extinction_law", None)
    models = []
    for name in names:
        if name in host_models:
            host = host_models[name]
            if sample_times is None:
                raise ValueError(f"host model {name!r}: sample_times are needed (--em-tmin / --em-tmax)")
            m = ExternalLightCurveModel(name, filters, sample_times, model_parameters=getattr(host, "model_parameters", ()),
                                        generate_lightcurve=host.generate_lightcurve, gap_free=bool(getattr(host, "gap_free", False)))
        elif name == "Me2017":
            m = SimpleKilonovaLightCurveModel(name, filters=filters, sample_times=sample_times)
        else:
            m = SVDLightCurveModel(name, svd_path=getattr(args, "svd_path", None), svd_mag_ncoeff=getattr(args, "svd_mag_ncoeff", None),
                                   interpolation_type=getattr(args, "interpolation_type", "keras"), filters=filters,
                                   sample_times=sample_times, local_only=True)
        if law:
            m.extinction_law = law
        models.append(m)
    return models[0] if len(models) == 1 else CombinedLightCurveModelContainer(models)


#: effective wavelengths (m) of the built-in filter names (nmma/em/utils.py:680-721)
BUILTIN_FILTER_LAMBDAS = dict(zip(
    ["u", "g", "r", "i", "z", "y", "J", "H", "K"],
    1e-10 * np.array([3561.8, 4866.46, 6214.6, 7687.0, 7127.0, 7544.6, 8679.5, 9633.3, 12350.0])))
BUILTIN_FILTER_LAMBDAS.update(dict(zip(["U", "B", "V", "R", "I"],
                                       1e-10 * np.array([3605.07, 4413.08, 5512.12, 6585.91, 8059.88]))))
C_SI = 2.99792458e8


class _TensorModelMixin:
    """Shared plumbing of models evaluated directly on ``sample_times`` (no SVD surrogate)."""

    gpu_model_kind = "external"

    @property
    def gpu_filters(self):
        return list(self.filters)

    def engine_kwargs(self):
        return dict(svd_model=None, model_filters=self.gpu_filters, model_parameters=self.model_parameters,
                    sample_times=self.model_times, cosmo_grid=self.cosmo_grid, device=self.device,
                    model_kind=self.gpu_model_kind, filter_nu0=getattr(self, "filter_nu0", None),
                    ebv_coeff=getattr(self, "ebv_coeff", None))

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_lc_engine"], state["_lc_names"] = None, None
        return state

    def _model_engine(self, names):
        from ..engine import EMEngine
        names = list(names)
        if getattr(self, "_lc_engine", None) is None or names != self._lc_names:
            self._lc_engine = EMEngine(parameter_names=names, **self.engine_kwargs())
            self._lc_names = names
        return self._lc_engine

    def lightcurves_abs(self, theta, names):
        """Source-frame absolute magnitudes [B, M, NS] (torch CUDA tensor) for rows of theta."""
        return self._model_engine(names).model_lightcurves(theta)


class SimpleKilonovaLightCurveModel(_TensorModelMixin, LightCurveModelContainer):
    """Analytic kilonova models on the GPU (reference: model.py:1280-1337).  ``Me2017``
    (lightcurve_generation.py:566-652) is implemented; the explicit-Euler layer model runs
    one wave per parameter vector (``me2017_lc``)."""

    gpu_model_kind = "me2017"

    def __init__(self, model="Me2017", filters=None, sample_times=None, filter_lambdas=None, cosmo_grid=None,
                 device=0, **em_model_kwargs):
        if model != "Me2017":
            raise ValueError("nmma_amd implements the Me2017 analytic model only")
        mp = ["log10_mej", "log10_vej", "beta", "log10_kappa_r"]
        super().__init__(model, filters, mp, sample_times)
        if np.any(np.asarray(self.model_times) == 0):
            raise ValueError("For Me2017, start later than t=0")
        lam = dict(BUILTIN_FILTER_LAMBDAS)
        lam.update(filter_lambdas or {})
        missing = [f for f in self.filters if f not in lam]
        if missing:
            raise ValueError(f"effective wavelengths needed for filters {missing} (pass filter_lambdas)")
        self.lambdas = np.array([lam[f] for f in self.filters])
        self.nu_0s = C_SI / self.lambdas
        self.filter_nu0 = dict(zip(self.filters, self.nu_0s))
        self.cosmo_grid, self.device = cosmo_grid, device
        self._lc_engine, self._lc_names = None, None

    def gen_detector_lc(self, parameters=None, sample_times=None):
        import torch
        names = sorted(k for k, v in parameters.items() if np.ndim(v) <= 1 and _is_number(v))
        eng = self._model_engine(names)
        cols = [np.atleast_1d(np.asarray(parameters[k], float)) for k in names]
        n = max(len(c) for c in cols)
        theta = np.stack([np.broadcast_to(c, (n,)) for c in cols], axis=1)
        tobs, mag = eng.lightcurves(torch.as_tensor(theta))
        tobs, mag = tobs.cpu().numpy(), mag.cpu().numpy()
        scalar = all(np.ndim(parameters[k]) == 0 for k in names)
        lc = {f: (mag[0, i] if scalar else mag[:, i]) for i, f in enumerate(self.filters)}
        return (tobs[0] if scalar else tobs), lc


class ExternalLightCurveModel(_TensorModelMixin, LightCurveModelContainer):
    """A model whose source-frame light curves are supplied by the caller as a tensor
    [B, M, NS] (e.g. GRB afterglows computed with afterglowpy, which is third-party C and
    out of scope here).  Stands in for ``GRBLightCurveModel`` (model.py:891-1011) inside
    :class:`CombinedLightCurveModelContainer`."""

    gpu_model_kind = "external"

    def __init__(self, model, filters, sample_times, model_parameters=(), cosmo_grid=None, device=0, gap_free=False,
                 generate_lightcurve=None, generate_lightcurve_batch=None):
        """``gap_free``: the supplied curves never hold a non-finite node strictly inside the time grid (afterglowpy's are finite, or
        the row is reported as failed: lightcurve_generation.py:259-283) -- a combined model's one-launch likelihood then skips its
        re-evaluation launch; a curve that breaks the promise makes the next likelihood call raise.
        ``generate_lightcurve``: a host callable with the reference's model signature, ``(sample_times, parameters: dict) ->
        {filter: abs_mag[NS]}`` or something falsy when the model has no light curve for these parameters (model.py:378-379, :1423-1426)
        -- e.g. the bound ``generate_lightcurve`` of the reference's own ``GRBLightCurveModel``.  With it the likelihood needs no
        ``external_lc``: ``log_likelihood(parameters)``, ``log_likelihood_batch(theta)`` and a ``GPUPool`` call it once per row on the
        host, then evaluate the whole batch in one launch.
        ``generate_lightcurve_batch``: the vectorised form for host models that have one -- ``(sample_times, parameters: dict of
        arrays [B]) -> {filter: abs_mag[B, NS]}`` or ``({...}, ok[B])`` -- called once per batch instead of once per row."""
        super().__init__(model, filters, list(model_parameters), sample_times)
        self.gap_free = bool(gap_free)
        self.cosmo_grid, self.device = cosmo_grid, device
        self._lc_engine, self._lc_names = None, None
        self.generator = generate_lightcurve
        self.batch_generator = generate_lightcurve_batch
        self.batch_gap_free = False       # set per batch by CombinedLightCurveModelContainer.host_operands
        self.batch_checked = False        # ... which has then LOOKED at the batch's curves: its finding stands in for `gap_free`

    def generate_lightcurve(self, sample_times, parameters):
        """model.py:405-408: the model's source-frame light curve for one parameter dict (the supplied callable's)."""
        if self.generator is None:
            raise RuntimeError(f"external model {self.model!r} has no `generate_lightcurve` callable")
        return self.generator(sample_times, parameters)

    def generate_batch(self, theta, names, fixed=None, conversion=None):
        """``(lc[B, M, NS] float64, ok[B] bool)`` from one host call of the generator per row: every row's dict holds the sampled
        columns, the fixed parameters and what ``conversion`` (the likelihood's parameter conversion chain) derives from them; a
        falsy result marks the row as failed (the combined model then returns no light curve: model.py:1423-1426 -> floor); a filter
        the generator does not return stays +inf (no flux)."""
        theta = np.asarray(theta, dtype=float)
        st = np.asarray(self.model_times, float)
        lc = np.full((len(theta), len(self.filters), len(st)), np.inf)
        ok = np.ones(len(theta), dtype=bool)
        if self.batch_generator is not None:
            p = {k: np.full(len(theta), v) for k, v in (fixed or {}).items()}
            p.update({n: theta[:, j].copy() for j, n in enumerate(names)})
            if conversion is not None:
                p = conversion(p)
            res = self.batch_generator(st, p)
            if isinstance(res, tuple):
                res, good = res
                ok &= np.asarray(good, dtype=bool)
            for k, f in enumerate(self.filters):
                if f in res:
                    lc[:, k] = np.asarray(res[f], dtype=float)
            lc[~ok] = np.inf
            return lc, ok
        for i, row in enumerate(theta):
            p = dict(fixed or {})
            p.update(zip(names, (float(v) for v in row)))
            if conversion is not None:
                p = conversion(p)
            res = self.generate_lightcurve(st, p)
            if not res:
                ok[i] = False
                continue
            for k, f in enumerate(self.filters):
                if f in res:
                    lc[i, k] = np.asarray(res[f], dtype=float)
        return lc, ok

    def lightcurves_abs(self, theta, names):
        raise RuntimeError(f"light curves of external model {self.model!r} must be passed in `external_lc` "
                           "(or give the model a `generate_lightcurve` callable)")


class CombinedLightCurveModelContainer(_TensorModelMixin):
    """Flux sum of several models (reference: model.py:1342-1510).  Sub-models may bring their own ``sample_times``
    and filter lists: the combination lives on the sorted union of the time grids (:1372-1374) and on the union of the
    filters (:1368); every sub-model's curves are interpolated onto the union grid with +inf outside their finite
    nodes (:1440-1448) and looked up per filter as ``stack_magnitudes`` does (:1490-1503: the filter itself or its
    renamed equivalent, else the mean of the helper bands of an averaged filter, else no contribution)."""

    gpu_model_kind = "external"

    def __init__(self, models, cosmo_grid=None, device=0):
        from . import utils
        self.lc_models = list(models)
        self.model = [m.model for m in self.lc_models]
        first = self.lc_models[0]
        self.filters = []
        for m in self.lc_models:
            for f in m.filters:
                if f not in self.filters:
                    self.filters.append(f)
        self.all_filters = list(self.filters)
        self.model_times = np.array(sorted(set().union(*[np.asarray(m.model_times, float).tolist() for m in self.lc_models])))
        self.model_parameters = []
        self.cosmo_grid = cosmo_grid if cosmo_grid is not None else getattr(first, "cosmo_grid", None)
        self.device = device
        self._lc_engine, self._lc_names = None, None
        # per sub-model: None when it already lives on the union grid with the union filters, else the source indices of
        # every union filter in the sub-model's own filter list
        averaging = utils.FILTER_AVERAGES
        direct = {}
        for f in self.filters:
            if f in utils.FILTER_RENAMES:
                direct[f] = utils.FILTER_RENAMES[f]
            elif f not in averaging:
                direct[f] = f
        self._regrid = []
        for m in self.lc_models:
            mf = list(m.gpu_filters) if hasattr(m, "gpu_filters") else list(m.filters)
            same = mf == self.filters and np.array_equal(np.asarray(m.model_times, float), self.model_times)
            if same:
                self._regrid.append(None)
                continue
            srcs = []
            for f in self.filters:
                if f in direct and direct[f] in mf:
                    srcs.append([mf.index(direct[f])])
                elif f in averaging and all(g in mf for g in averaging[f]):
                    srcs.append([mf.index(g) for g in averaging[f]])
                else:
                    srcs.append([])
            self._regrid.append(srcs)

    def __repr__(self):
        return "Combination of " + " and ".join(repr(m) for m in self.lc_models)

    def check_vs_priors(self, priors):
        for m in self.lc_models:
            m.check_vs_priors(priors)
        for m in self.lc_models:
            if getattr(m, "cosmo_grid", None) is not None and self.cosmo_grid is None:
                self.cosmo_grid = m.cosmo_grid

    @property
    def good_parameters(self):
        return all(m.good_parameters for m in self.lc_models)

    @good_parameters.setter
    def good_parameters(self, value):
        for m in self.lc_models:
            m.good_parameters = value

    def parameter_conversion(self, parameters):
        for m in self.lc_models:
            parameters = m.parameter_conversion(parameters)
        return parameters

    def host_operands(self, theta, names, fixed=None, have=None):
        """``external_lc`` entries -- ``{model name: (lc[B, M_k, NS_k], ok[B])}`` -- of every external sub-model that brings its own
        ``generate_lightcurve`` callable and is not in ``have`` already: one host call per row, with the combination's parameter
        conversion chain applied first (model.py:1405-1408)."""
        out = dict(have or {})
        for m in self.lc_models:
            if not isinstance(m, ExternalLightCurveModel):
                continue
            m.batch_gap_free = m.batch_checked = False
            if m.model not in out and (m.generator is not None or m.batch_generator is not None):
                th = theta.detach().cpu().numpy() if hasattr(theta, "detach") else np.asarray(theta)
                lc, ok = m.generate_batch(th, names, fixed, self.parameter_conversion)
                # (the curves are in host memory anyway: whether any delivered row has a non-finite node strictly inside the grid is
                #  one pass over them -- without one, the one-launch likelihood needs no re-evaluation launch for this batch)
                # (and it REPLACES a constructor promise `gap_free=True` for this batch: a filter the generator does not return is +inf
                #  on every node -- a legal "no flux" outcome that a blanket promise would turn into a poisoned handle)
                m.batch_gap_free = bool(np.isfinite(lc[ok][:, :, 1:-1]).all())
                m.batch_checked = True
                out[m.model] = (lc, ok)
        return out

    def stack2_plan(self):
        """``(surrogate_model, other_model)`` when the combination can be evaluated in ONE launch (``EMEngine.loglike_stack2``): two
        sub-models, one of them an SVD surrogate that lists every filter of the combination (in the combination's order) -- its
        curves then never leave the chip.  Both on the combination's grid and filters is what the reference's drivers build (every
        sub-model with the same ``filters`` and ``sample_times``, model.py:1591-1614).  Round 6: the sub-models may bring their OWN
        time grids (model.py:1372-1374, :1440-1448) -- the surrogate's move onto the union grid is a static linear map folded into
        the kernel's basis rows (``base_times`` of the engine), the other sub-model's curves take ``EMEngine.regrid`` (any filter
        list: missing filters carry no flux, averaged bands their helper bands' mean) before they become the operand.  Else None
        (a filter the surrogate does not list: ``stacked_sets`` + ``EMEngine.loglike_lc_sets``)."""
        if len(self.lc_models) != 2:
            return None
        for i, m in enumerate(self.lc_models):
            if not isinstance(m, SVDLightCurveModel):
                continue
            # (the filters the model LISTS: one without a surrogate is its null output -- +inf on every node, "radio and X-ray filters
            #  when using with GRB data", lightcurve_generation.py:168-169 -- and rides along as a null filter of the engine)
            if list(m.filters) == self.filters and len(m.gpu_filters) > 0:
                return m, self.lc_models[1 - i]
        return None

    def stack2_union(self):
        """``(surrogate's own sample_times or None, the other sub-model's regrid plan or None)`` of ``stack2_plan``: what differs from
        the combination's grid / filters.  ``(None, None)``: both sub-models already live there."""
        kn, other = self.stack2_plan()
        i = self.lc_models.index(kn)
        same_times = np.array_equal(np.asarray(kn.model_times, float), self.model_times)
        return (None if same_times else np.asarray(kn.model_times, float)), self._regrid[1 - i]

    def stack2_engine_kwargs(self):
        """Engine arguments of the one-launch form: the surrogate's tensors with the COMBINATION's grid, cosmology and extinction
        settings (the ones the likelihood-from-curves engine of the materialising path gets from ``engine_kwargs``)."""
        kn, _ = self.stack2_plan()
        own = self.engine_kwargs()
        kw = kn.engine_kwargs()
        kw.pop("extinction_law", None)
        kw.update(sample_times=self.model_times, cosmo_grid=own["cosmo_grid"], device=own["device"], ebv_coeff=own["ebv_coeff"],
                  filter_nu0=own["filter_nu0"], stack_operands=1)
        base, _ = self.stack2_union()
        if base is not None:        # the surrogate on its own grid, the task's rows on the union grid
            kw["base_times"] = base
        # every filter the combination lists is a model filter of the engine; those without a surrogate as null filters
        kw["model_filters"] = list(self.filters)
        null = [f for f in self.filters if f not in kn.svd_mag_model]
        if null:
            kw["null_filters"] = null
            if kw.get("filter_nu0") is not None and any(f not in kw["filter_nu0"] for f in null):
                kw["filter_nu0"] = dict(kw["filter_nu0"], **{f: 1.0 for f in null if f not in kw["filter_nu0"]})
            if kw.get("ebv_coeff") is not None:
                kw["ebv_coeff"] = dict(kw["ebv_coeff"], **{f: 0.0 for f in null if f not in kw["ebv_coeff"]})
        return kw

    def second_operand(self, theta, names, external_lc=None, stack_engine=None):
        """The other sub-model's source-frame set ``[B, M, NS]`` for ``EMEngine.loglike_stack2`` and the rows for which it
        delivered no light curve (bool CUDA tensor or None) -- see ``stacked_sets`` for ``external_lc``.  A sub-model on its own
        grid / filter list is moved onto the combination's by ``stack_engine.regrid`` (then pass ``completed=True`` to
        ``loglike_stack2``: ``stack2_union()[1] is not None``)."""
        import torch
        _, m = self.stack2_plan()
        _, plan = self.stack2_union()
        failed = None
        if isinstance(m, ExternalLightCurveModel):
            val = (external_lc or {})[m.model]
            if isinstance(val, (tuple, list)):
                val, ok = val
                bad = ~(ok.to(torch.bool) if isinstance(ok, torch.Tensor) else torch.as_tensor(np.asarray(ok, dtype=bool)))
                failed = bad.to(f"cuda:{self.device}")
            lc = torch.as_tensor(val).to(f"cuda:{self.device}")
        else:
            lc = m.lightcurves_abs(theta, names)
        if plan is not None:
            if stack_engine is None:
                raise RuntimeError("second_operand: this combination's second sub-model lives on its own grid / filters: pass stack_engine")
            lc = stack_engine.regrid(lc, np.asarray(m.model_times, float), plan)
        return lc, failed

    def stacked_sets(self, theta, names, external_lc=None, stack_engine=None):
        """The sub-models' source-frame sets ``[B, M, NS]`` on the union grid / filters (the operands of ``stack_magnitudes``) and
        the rows for which a sub-model delivered no light curve (bool CUDA tensor [B], or None).  ``external_lc`` maps the name of
        each :class:`ExternalLightCurveModel` to its tensor ``[B, M_k, NS_k]`` (on that model's own filters and times), or to a
        pair ``(tensor, ok[B])``: rows with ``ok == False`` are the sub-model's "no light curve for these parameters"
        (an empty dict in the reference, model.py:1423-1426) and floor the sample."""
        import torch
        external_lc = external_lc or {}
        eng = stack_engine or self._model_engine(names)
        sets, failed = [], None
        for m, plan in zip(self.lc_models, self._regrid):
            if isinstance(m, ExternalLightCurveModel):
                val = external_lc[m.model]
                if isinstance(val, (tuple, list)):
                    val, ok = val
                    # (the mask may be a CUDA tensor, the natural companion of a CUDA light-curve tensor)
                    bad = ~(ok.to(torch.bool) if isinstance(ok, torch.Tensor) else torch.as_tensor(np.asarray(ok, dtype=bool)))
                    bad = bad.to(f"cuda:{self.device}")
                    failed = bad if failed is None else (failed | bad)
                lc = torch.as_tensor(val).to(f"cuda:{self.device}")
            else:
                lc = m.lightcurves_abs(theta, names)
            if plan is not None:
                lc = eng.regrid(lc, np.asarray(m.model_times, float), plan)
            sets.append(lc)
        return sets, failed

    def stacked_lightcurves_abs(self, theta, names, external_lc=None, stack_engine=None):
        """[B, M, NS] flux-summed source-frame curves on the union grid / filters (NaN rows where a sub-model failed): the
        materialised form, for plots and stage-level tests -- the likelihood stacks on chip (``EMEngine.loglike_lc_sets``)."""
        eng = stack_engine or self._model_engine(names)
        sets, failed = self.stacked_sets(theta, names, external_lc, eng)
        out = eng.stack(sets)
        if failed is not None and bool(failed.any()):
            out[failed.to(out.device)] = float("nan")
        return out
