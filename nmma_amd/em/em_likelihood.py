"""EM likelihood plugin (``nmma/em/em_likelihood.py``): ``EMTransientLikelihood``
wrapping ``MultiFilterTransient`` -- same constructors, attributes and per-sample
``log_likelihood(parameters)`` as the reference, plus the batched entry point
``log_likelihood_batch(theta)`` that evaluates a whole live-point set in one launch.

``OpticalLightCurve`` (the <=0.2.x name) is exported as an alias.
"""
from __future__ import annotations

import numpy as np

from ..core.base import LOGL_FLOOR, NMMALikelihood, fixed_value, is_constraint
from ..core.conversion import convert_mtot_mni
from . import utils


class MultiFilterTransient:
    """em_likelihood.py:266-354 (+ BasicEMTransient :136-263) on the GPU."""

    def __init__(self, filters, light_curve_model, light_curve_data, systematics_handler, priors,
                 detection_limit, verbose):
        self.observed_filters = list(filters)
        # (stand-in for the bandpass registry the reference consults: every model / observed filter name EXCEPT the averaged
        #  ones -- w, o, c, V, I ... are never registry names, so they keep their helper-band mean even when a model lists them)
        known = {f for f in (getattr(light_curve_model, "filters", None) or []) if f not in utils.FILTER_AVERAGES} | \
            {f for f in self.observed_filters if f not in utils.FILTER_AVERAGES}
        self.model_filter_mapping, self.obs_average_mapping = utils.get_filter_name_mapping(
            self.observed_filters, known)
        self.light_curve_model = light_curve_model
        self.light_curve_model.check_vs_priors(priors)
        (self.light_curve_times, self.light_curves,
         self.light_curve_uncertainties, self.trigger_time) = light_curve_data
        systematics_handler.reset(self.light_curve_model.model_times, priors)
        self.systematics_handler = systematics_handler
        self.verbose = verbose
        self.set_detection_limit(detection_limit)
        self.priors = priors
        self._engine, self._names = None, None
        self._engine2, self._names2, self._stack2_off = None, None, False      # a combined model's one-launch engine

    def set_detection_limit(self, detection_limit):
        self.detection_limit = utils.set_filter_associated_dict(detection_limit, self.observed_filters)
        self._engine = None
        self._engine2 = None

    def __repr__(self):
        return f"{self.__class__.__name__} (light_curve_model={self.light_curve_model})"

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None          # rebuilt lazily per process (core/mpi_setup.py:614-636)
        state["_engine2"] = None
        return state

    # ---- theta layout -------------------------------------------------------------------
    def sampling_layout(self):
        """(names, fixed): sampled columns of theta and fixed parameter values, from the priors."""
        names, fixed = [], {}
        for key, prior in self.priors.items():
            if is_constraint(prior):
                continue
            val = fixed_value(prior)
            if val is None:
                names.append(key)
            else:
                fixed[key] = val
        return names, fixed

    def engine(self, names=None):
        if names is None:
            names = self._names or self.sampling_layout()[0]
        names = list(names)
        self._check_names(names)
        if self._engine is None or names != self._names:
            if self._engine is not None:
                self._engine.close()
            model = self.light_curve_model
            self._engine = self._build_engine(names, model.gpu_filters, model.engine_kwargs())
            self._names = names
        return self._engine

    @staticmethod
    def _check_names(names):
        if "Omega_matter" in names:
            # the device scales distances with a sampled H0 (one grid serves every H0 in a flat universe); a sampled matter
            # density changes the shape of z(d_L) and would need a grid per sample
            from .. import _lib as L
            raise L.NMMAHipError("a sampled Omega_matter needs a z(d_L) relation per sample, which the device path does not "
                                 "tabulate: sample 'redshift' instead, or fix Omega_matter")

    def _build_engine(self, names, gpu_filters, engine_kwargs):
        from ..engine import EMEngine
        _, fixed = self.sampling_layout()
        fixed = {k: v for k, v in fixed.items() if k not in names}
        obs = [f for f in self.observed_filters if len(self.light_curve_times[f]) > 0]
        sources = utils.resolve_sources(obs, gpu_filters, known_filters=set(self.model_filter_mapping.values()))
        return EMEngine(parameter_names=names, fixed=fixed,
                        data=(self.light_curve_times, self.light_curves, self.light_curve_uncertainties),
                        observed_filters=obs, sources=sources, detection_limit=self.detection_limit,
                        systematics=self.systematics_handler.kernel_spec(), **engine_kwargs)

    def stack2_engine(self, names=None):
        """The one-launch engine of a combined model (``CombinedLightCurveModelContainer.stack2_plan``): the surrogate's engine
        with the likelihood's photometry, laid out to take the other sub-model's curves as an operand."""
        names = list(names if names is not None else (self._names2 or self._names or self.sampling_layout()[0]))
        self._check_names(names)
        if self._engine2 is None or names != self._names2:
            if self._engine2 is not None:
                self._engine2.close()
            model = self.light_curve_model
            kn, _ = model.stack2_plan()
            kw = model.stack2_engine_kwargs()
            self._engine2 = self._build_engine(names, kw["model_filters"], kw)
            self._names2 = names
        return self._engine2

    # ---- evaluation -------------------------------------------------------------------
    def _null_model_filters(self):
        """Filters a single surrogate model lists without having a network for them (not for combinations)."""
        model = self.light_curve_model
        if hasattr(model, "stacked_lightcurves_abs") or not hasattr(model, "svd_mag_model"):
            return []
        return [f for f in (model.filters or []) if f not in model.svd_mag_model]

    def log_likelihood(self, parameters):
        """One parameter dict -> float (em_likelihood.py:186-204)."""
        if self._null_model_filters():
            return -np.inf
        names = self._names or self.sampling_layout()[0]
        names = [n for n in names if n in parameters] if self._names is None else names
        if hasattr(self.light_curve_model, "stacked_lightcurves_abs"):      # a combined model: the batched path with one row
            val = float(self.log_likelihood_batch(np.array([[float(parameters[n]) for n in names]]), names)[0])
            return -np.inf if val == LOGL_FLOOR else val
        eng = self.engine(names)
        theta = np.array([[float(parameters[n]) for n in eng.parameter_names]])
        val = float(eng.loglike(theta)[0])
        if self.verbose:
            print(parameters, val)
        return -np.inf if val == LOGL_FLOOR else val

    def log_likelihood_batch(self, theta, names=None, external_lc=None):
        """theta[B, D] (numpy or torch CUDA tensor; columns = ``names`` or the sampled prior
        keys) -> logL[B] with the reference's floor already applied.  For combined models
        ``external_lc`` maps external sub-model names to their light-curve tensors."""
        model = self.light_curve_model
        null = self._null_model_filters()
        if null:
            # the model LISTS filters its surrogate has nothing for: calc_svd_lc answers +inf on every node (lightcurve_generation.py:168-169)
            # and sanity_check fails for every sample (em_likelihood.py:305-311) -- the reference's floor, without a launch.  (Inside
            # a combination the same filters are fine: the other sub-model's flux is the band's curve.)
            try:
                import torch
                if isinstance(theta, torch.Tensor):
                    return torch.full((theta.shape[0],), LOGL_FLOOR, dtype=torch.float64, device=theta.device)
            except ImportError:
                pass
            return np.full(len(np.asarray(theta)), LOGL_FLOOR)
        if hasattr(model, "stacked_lightcurves_abs"):       # CombinedLightCurveModelContainer
            import torch
            th = torch.as_tensor(np.asarray(theta)) if not isinstance(theta, torch.Tensor) else theta
            out = None
            if hasattr(model, "host_operands"):             # sub-models whose curves a host callable computes (e.g. afterglowpy)
                lay_names, fixed = self.sampling_layout()
                use = list(names) if names is not None else (self._names2 or self._names or lay_names)
                external_lc = model.host_operands(th, use, {k: v for k, v in fixed.items() if k not in use}, external_lc)
            if not self._stack2_off and model.stack2_plan() is not None:
                # two sub-models on one grid (the reference drivers' case): ONE launch, the surrogate's curves never leave the chip
                eng2 = self.stack2_engine(names)
                th = th.to(f"cuda:{eng2.device}", dtype=torch.float64)
                lc2, failed = model.second_operand(th, eng2.parameter_names, external_lc, stack_engine=eng2)
                other = model.stack2_plan()[1]
                base, plan2 = model.stack2_union()
                regridded = plan2 is not None      # (own grid / filters: the operand went through regrid -- no interior gaps left)
                # (a "no interior gaps" promise says nothing about nodes that NEITHER sub-model covers on a union grid: those rows need
                #  the re-evaluation launch, so the promise is not forwarded there)
                promise = bool(getattr(other, "batch_gap_free", False) if getattr(other, "batch_checked", False) else getattr(other, "gap_free", False))
                out = eng2.loglike_stack2(th, lc2, failed, completed=regridded, gap_free=promise and base is None and not regridded)
                if out is None:                             # (the handle has no one-launch form: decided once per likelihood)
                    self._stack2_off = True
                    self._engine2.close()
                    self._engine2 = None
            if out is None:
                eng = self.engine(names)
                th = th.to(f"cuda:{eng.device}", dtype=torch.float64)
                # (the flux sum of the sub-models is formed on chip while the likelihood kernel stages a sample's curves)
                sets, failed = model.stacked_sets(th, eng.parameter_names, external_lc, stack_engine=eng)
                out = eng.loglike_lc_sets(th, sets, failed)
            return out if isinstance(theta, torch.Tensor) else out.cpu().numpy()
        return self.engine(names).loglike(theta)

    def final_diagnostics(self, bestfit_params, args, result=None):
        return self.light_curve_model.gen_detector_lc(dict(bestfit_params))


class EMTransientLikelihood(NMMALikelihood):
    """A generic EM transient likelihood object (em_likelihood.py:42-133)."""

    def __init__(self, light_curve_model, light_curve_data, systematics_handler, priors, filters=None,
                 detection_limit=np.inf, verbose=False, **kwargs):
        if not filters:
            filters = list(light_curve_data[0].keys())
        sub_model = MultiFilterTransient(filters, light_curve_model, light_curve_data, systematics_handler,
                                         priors, detection_limit, verbose)
        super().__init__(sub_model, priors, **kwargs)

    def setup_submodel_conversion(self):
        """em_likelihood.py:91-100: the AnBa2022 grids add their derived masses (which priors may
        constrain) ahead of the model's own conversion."""
        lc_model = self.sub_model.light_curve_model
        model_names = lc_model.model if isinstance(lc_model.model, (list, tuple)) else [lc_model.model]
        if any(name in ("AnBa2022_linear", "AnBa2022_log") for name in model_names):
            self.conv_functions.append(convert_mtot_mni)
        self.conv_functions.append(lc_model.parameter_conversion)

    def sanity_checks(self):
        return self.sub_model.light_curve_model.good_parameters

    def __repr__(self):
        return f"{self.__class__.__name__} based on {self.sub_model.__repr__()}"

    def log_likelihood(self, parameters=None):
        if parameters is None:
            parameters = self.parameters
        return float(super().log_likelihood(dict(parameters)))

    def log_likelihood_batch(self, theta, names=None, external_lc=None):
        """Batched ``log_likelihood``: every row of ``theta`` is one parameter vector.
        Conversions (KNtheta <- inclination_EM, log10 aliases), z(d_L), distance modulus,
        systematics and the floor are all applied on the device."""
        out = self.sub_model.log_likelihood_batch(theta, names, external_lc)
        if self.constraints:
            out = self._apply_constraints_batch(out, theta, names)
        return out

    def _apply_constraints_batch(self, out, theta, names):
        """core/base.py:67-68, :77-82 for a batch: on the device for a CUDA ``theta`` (the conversion chain traced into a
        constraint program, ``core/constraints.py``), numpy on the converted columns for host arrays."""
        cols = list(names) if names is not None else self.sub_model.engine().parameter_names
        _, fixed = self.sub_model.sampling_layout()
        return self.apply_constraints_batch(out, theta, cols, fixed)

    def parameter_names(self):
        return self.sub_model.sampling_layout()[0]

    def posterior_conversion(self, posterior_samples):
        """em_likelihood.py:122-131."""
        if "log10_mej_dyn" in posterior_samples and "log10_mej_wind" in posterior_samples:
            posterior_samples["log10_mej"] = np.log10(10 ** posterior_samples["log10_mej_wind"]
                                                      + 10 ** posterior_samples["log10_mej_dyn"])
        if "thetaCore" in posterior_samples:            # afterglow jet wing: angle <-> ratio to the core
            if "thetaWing" in posterior_samples:
                posterior_samples["alphaWing"] = posterior_samples["thetaWing"] / posterior_samples["thetaCore"]
            elif "alphaWing" in posterior_samples:
                posterior_samples["thetaWing"] = posterior_samples["alphaWing"] * posterior_samples["thetaCore"]
        return posterior_samples

    def final_diagnostics(self, bestfit_params, args, result=None):
        return self.sub_model.final_diagnostics(bestfit_params, args, result)


def build_em_likelihood(light_curve_model, light_curve_data, trigger_time, priors, filters=None,
                        systematics_file=None, error_budget=None, injection=None, detection_limit=np.inf,
                        verbose=False):
    """Raw photometry -> ready likelihood, in the order the reference's drivers use (em/analysis.py:135-170,
    em_likelihood.py:12-40): times relative to the trigger, the model-window consistency check (which
    trims injected data and rejects real data the model cannot cover), the systematics handler on the
    final epochs, then the likelihood.

    ``light_curve_data``: ``{filter: {"time": mjd, "mag": ..., "mag_error": ...}}``."""
    from .systematics import FilterSystematicsHandler
    data = utils.setup_filtered_lc_data(light_curve_data, trigger_time)
    data = utils.check_model_time_consistency(data, light_curve_model, priors, injection)
    filters = list(filters) if filters else list(data[0].keys())
    handler = FilterSystematicsHandler(filters, systematics_file, error_budget, data[0])
    return EMTransientLikelihood(light_curve_model, data, handler, priors, filters=filters,
                                 detection_limit=detection_limit, verbose=verbose)


#: legacy (<= 0.2.x) name used by BASELINE.json's north_star
OpticalLightCurve = EMTransientLikelihood
