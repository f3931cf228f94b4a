"""Joint multi-messenger likelihood (``nmma/joint/joint_likelihood.py:12-88``).

``MultiMessengerLikelihood`` sums the log-likelihoods of its messengers per sample and floors a
non-finite total (``joint_likelihood.py:62-67``).  The EM messenger is evaluated on the GPU; any other
messenger (gravitational waves, EOS, population) whose arithmetic lives in third-party code enters
either as an ordinary likelihood object (``sub_log_likelihood(parameters) -> float``, evaluated per
sample on the host as the reference does) or -- on the batched path -- as a tensor of per-sample
log-likelihoods the caller already has on the device (:class:`ExternalLogLikelihood`).
"""
from __future__ import annotations

import numpy as np

from ..core.base import LOGL_FLOOR, NMMALikelihoodMixin, _BilbyLikelihood


class ExternalLogLikelihood:
    """A messenger whose per-sample log-likelihood is computed elsewhere (e.g. bilby's
    ``GravitationalWaveTransient`` for BASELINE config 5) and handed over as data.

    Per-sample API: ``func(parameters) -> float``.  Batched API: the values for a batch are passed to
    ``MultiMessengerLikelihood.log_likelihood_batch(..., external_logl={name: tensor})``.
    """

    def __init__(self, name, func=None, noise_log_likelihood=0.0):
        self.name, self.func, self._noise = name, func, float(noise_log_likelihood)

    def __repr__(self):
        return f"ExternalLogLikelihood({self.name})"

    def parameter_conversion(self, parameters):
        return parameters

    def posterior_conversion(self, posterior):
        return posterior

    def sanity_checks(self):
        return True

    def noise_log_likelihood(self):
        return self._noise

    def sub_log_likelihood(self, parameters):
        if self.func is None:
            raise RuntimeError(f"messenger {self.name!r} has no per-sample function: use log_likelihood_batch "
                               "with external_logl")
        val = self.func(parameters)
        return val if np.isfinite(val) else np.nan_to_num(-np.inf)

    def final_diagnostics(self, *a, **k):
        return None


class MultiMessengerLikelihood(NMMALikelihoodMixin, _BilbyLikelihood):
    """joint_likelihood.py:12-88 (``bilby.core.likelihood.JointLikelihood`` surface: ``likelihoods``)."""

    def __init__(self, messenger_likelihoods, priors, conversion_instructions=None):
        super().__init__()
        self.likelihoods = list(messenger_likelihoods)
        self.priors = priors
        self.conversion_instructions = conversion_instructions
        self._noise_logl = float(sum(lh.noise_log_likelihood() for lh in self.likelihoods))
        self.multi_conversion = None
        self.setup_parameter_conversion()

    def setup_parameter_conversion(self):
        """joint_likelihood.py:42-58: without instructions every messenger's own conversion in turn; with a
        ``conversion_instructions`` dict a ``MultimessengerConversion`` chain in the reference's order (cosmo, gw, eos, ejecta,
        em, custom), the messengers' conversions filled in by their kind."""
        if self.conversion_instructions is None:
            return
        from ..core.conversion import MultimessengerConversion
        from ..em.em_likelihood import EMTransientLikelihood
        instructions = dict(self.conversion_instructions)
        for lh in self.likelihoods:
            if isinstance(lh, EMTransientLikelihood):
                instructions["em"] = lh.parameter_conversion
            elif type(lh).__name__ == "GravitationalWaveTransientLikelihood":
                instructions["gw"] = lh.parameter_conversion
            elif type(lh).__name__ == "EquationofStateLikelihood":
                instructions["eos"] = lh.parameter_conversion
        self.multi_conversion = MultimessengerConversion.from_dict(instructions)

    def __repr__(self):
        reprs = [repr(lh) for lh in self.likelihoods]
        if len(reprs) == 1:
            return f"{self.__class__.__name__} with {reprs[0]}"
        return f"{self.__class__.__name__} with {', '.join(reprs[:-1])} and {reprs[-1]}"

    # ---- reference surface --------------------------------------------------------------
    def noise_log_likelihood(self):
        return self._noise_logl

    def sanity_checks(self):
        return bool(np.prod([lh.sanity_checks() for lh in self.likelihoods]))

    def parameter_conversion(self, parameters):
        """joint_likelihood.py:72-78: the ``MultimessengerConversion`` chain when ``conversion_instructions`` were given (:72-73),
        else basic_parameter_conversion -- every messenger's conversion in turn."""
        if self.multi_conversion is not None:
            return self.multi_conversion.convert_to_multimessenger_parameters(parameters)
        for lh in self.likelihoods:
            parameters = lh.parameter_conversion(parameters)
        return parameters

    def _batched_conversion(self, columns):
        """``parameter_conversion`` for columns (a batch of samples, or the symbolic columns of the constraint tracer): the
        ``MultimessengerConversion`` chain without its scalar unwrapping."""
        if self.multi_conversion is not None:
            return self.multi_conversion.convert_to_multimessenger_parameters(columns, batched=True)
        for lh in self.likelihoods:
            columns = lh.parameter_conversion(columns)
        return columns

    def conversion_chain(self):
        return [self._batched_conversion]

    def posterior_conversion(self, posterior_samples):
        for lh in self.likelihoods:
            posterior_samples = lh.posterior_conversion(posterior_samples)
        return posterior_samples

    def sub_log_likelihood(self, parameters):
        logl = sum(lh.sub_log_likelihood(parameters) for lh in self.likelihoods)
        return logl if np.isfinite(logl) else np.nan_to_num(-np.inf)

    def final_diagnostics(self, bestfit_params, args, result=None):
        return [lh.final_diagnostics(bestfit_params, args, result) for lh in self.likelihoods]

    # ---- batched path ---------------------------------------------------------------------
    def log_likelihood_batch(self, theta, names=None, external_logl=None, external_lc=None):
        """``log_likelihood`` for every row of ``theta`` (core/base.py:77-82 with joint_likelihood.py:62-67 inside): each
        messenger's per-row log-likelihood, summed in messenger order and floored by one HIP kernel
        (``nmma_logl_sum_floor``), then the JOINT likelihood's own Constraint priors on the converted columns.

        Messengers with ``log_likelihood_batch`` (EM, GW) run on the GPU; an :class:`ExternalLogLikelihood` takes its values from
        ``external_logl[name]`` (torch tensor on the device, or numpy).  torch in -> torch out, numpy in -> numpy out."""
        import ctypes as C
        import torch
        from .. import _lib as L
        external_logl = external_logl or {}
        as_torch = isinstance(theta, torch.Tensor)
        parts, dev = [], None
        for lh in self.likelihoods:
            if isinstance(lh, ExternalLogLikelihood):
                if lh.name not in external_logl:
                    raise L.NMMAHipError(f"external_logl has no entry for messenger {lh.name!r}")
                part = external_logl[lh.name]
            elif hasattr(lh, "log_likelihood_batch"):
                kw = {"external_lc": external_lc} if external_lc is not None else {}
                part = lh.log_likelihood_batch(theta, names, **kw)
            else:
                raise L.NMMAHipError(f"messenger {lh!r} has no batched evaluation; wrap its values in ExternalLogLikelihood")
            part = part if isinstance(part, torch.Tensor) else torch.as_tensor(np.asarray(part, dtype=np.float64))
            if dev is None and part.is_cuda:
                dev = part.device
            parts.append(part)
        if dev is None:
            dev = torch.device("cuda:0")
        parts = [p.to(dev, dtype=torch.float64).contiguous() for p in parts]
        n = parts[0].shape[0]
        if any(p.shape != (n,) for p in parts):
            raise L.NMMAHipError(f"messengers returned different batch sizes: {[tuple(p.shape) for p in parts]}")
        total = torch.empty(n, dtype=torch.float64, device=dev)
        if n:
            ptrs = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
            L.check(L.load_library().nmma_logl_sum_floor(ptrs, len(parts), n, C.c_void_p(total.data_ptr()), dev.index or 0,
                                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                    "nmma_logl_sum_floor")
        if self.constraints:
            total = self._apply_joint_constraints(total, theta, names)
        return total if as_torch else total.cpu().numpy()

    def _apply_joint_constraints(self, total, theta, names):
        """The joint prior's Constraint entries on the batch: on the device when ``theta`` is a CUDA tensor and every constrained
        quantity is traceable arithmetic of the sampled columns (mass_1 / mass_2 from chirp mass and mass ratio, ...), else the
        numpy mask on the converted columns."""
        import torch
        from ..core.base import fixed_value, is_constraint
        if isinstance(theta, torch.Tensor) and theta.is_cuda:
            cols = list(names) if names is not None else [k for k, p in self.priors.items() if fixed_value(p) is None and not is_constraint(p)]
            fixed = {k: fixed_value(p) for k, p in self.priors.items() if fixed_value(p) is not None and k not in cols}
            prog = self.device_constraints(cols, fixed, theta.device.index or 0)
            if prog is not None:
                th = theta if (theta.dtype == torch.float64 and theta.is_contiguous()) else theta.to(torch.float64).contiguous()
                return prog.floor(th, total)
        return self.floor_constrained_rows(total, self._columns(theta, names))

    def _columns(self, theta, names):
        """Sampled columns + fixed priors as a dict of arrays: what ``parameter_conversion`` and the constraints work on."""
        import torch
        from ..core.base import fixed_value, is_constraint
        from .. import _lib as L
        if names is None:
            names = [k for k, p in self.priors.items() if fixed_value(p) is None and not is_constraint(p)]
        host = theta.detach().cpu().numpy() if isinstance(theta, torch.Tensor) else np.asarray(theta, dtype=float)
        cols = {n: host[:, i] for i, n in enumerate(names)}
        for key, prior in self.priors.items():
            val = fixed_value(prior)
            if val is not None and key not in cols:
                cols[key] = np.full(len(host), val)
        return cols

    def floor_constrained_rows(self, logl, columns):
        """As the mixin's, with a clear error when a constrained key is neither a column nor derived by a messenger's
        conversion (the reference would raise the same KeyError per sample, core/base.py:67-68)."""
        from .. import _lib as L
        converted = self._batched_conversion(dict(columns))
        missing = [k for k in self.constraints if k not in converted]
        if missing:
            raise L.NMMAHipError(f"Constraint priors on {missing} cannot be evaluated on the batched path: no messenger's "
                                 "parameter_conversion derives them from the sampled columns")
        from ..core.base import floor_rows
        return floor_rows(logl, self.evaluate_constraints(converted))
