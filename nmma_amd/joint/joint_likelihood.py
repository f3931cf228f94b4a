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
        """basic_parameter_conversion (joint_likelihood.py:75-78): every messenger's conversion in turn.
        (The MultimessengerConversion object of :58 needs the EOS / GW converters, which are out of scope.)"""
        for lh in self.likelihoods:
            parameters = lh.parameter_conversion(parameters)
        return parameters

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
        """Sum over messengers for every row of ``theta`` with the reference's floor.

        EM messengers (anything with ``log_likelihood_batch``) run on the GPU; an
        :class:`ExternalLogLikelihood` takes its values from ``external_logl[name]`` (torch tensor on the
        device, or numpy).  torch in -> torch out, numpy in -> numpy out."""
        import torch
        from .. import _lib as L
        external_logl = external_logl or {}
        as_torch = isinstance(theta, torch.Tensor)
        total, dev = None, None
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
            total = part if total is None else total.to(part.device if part.is_cuda else total.device) + \
                part.to(total.device if total.is_cuda else part.device)
        # a messenger's own floor (-1.797e308) plus anything stays below every finite log-likelihood and is re-floored
        # here exactly like the reference's `if np.isfinite(logl)` (the sum of two floors overflows to -inf)
        floor = torch.full_like(total, LOGL_FLOOR)
        total = torch.where(torch.isfinite(total) & (total > LOGL_FLOOR), total, floor)
        return total if as_torch else total.cpu().numpy()
