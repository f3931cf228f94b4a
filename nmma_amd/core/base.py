"""Likelihood plugin boundary: the classes samplers call (``nmma/core/base.py:37-185``).

``NMMALikelihood`` subclasses ``bilby.core.likelihood.Likelihood`` when bilby is
importable (drop-in for ``bilby.run_sampler`` / parallel-bilby); otherwise a duck-typed
base with the same surface (``parameters``, ``log_likelihood_ratio``, ``meta_data``).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - bilby is not installed in the build image
    from bilby.core.likelihood import Likelihood as _BilbyLikelihood
    from bilby.core.prior import Constraint as _BilbyConstraint
    HAVE_BILBY = True
except Exception:  # noqa: BLE001
    HAVE_BILBY = False

    class _BilbyLikelihood:
        """Surface of bilby.core.likelihood.Likelihood used by NMMA's drivers."""

        def __init__(self, parameters=None):
            self.parameters = parameters if parameters is not None else {}
            self._meta_data = None
            self._marginalized_parameters = []

        def log_likelihood(self, parameters=None):
            return np.nan

        def noise_log_likelihood(self):
            return np.nan

        def log_likelihood_ratio(self, parameters=None):
            return self.log_likelihood(parameters) - self.noise_log_likelihood()

        @property
        def meta_data(self):
            return getattr(self, "_meta_data", None)

        @meta_data.setter
        def meta_data(self, meta_data):
            self._meta_data = meta_data

        @property
        def marginalized_parameters(self):
            return self._marginalized_parameters

    class _BilbyConstraint:
        def __init__(self, minimum, maximum, name=None, **kw):
            self.minimum, self.maximum, self.name = minimum, maximum, name

        def prob(self, val):
            return (val > self.minimum) & (val < self.maximum)

Likelihood = _BilbyLikelihood
Constraint = _BilbyConstraint

#: np.nan_to_num(-np.inf): the reference's universal failure value (core/base.py:82, :181)
LOGL_FLOOR = float(np.nan_to_num(-np.inf))


def is_constraint(prior):
    return isinstance(prior, Constraint) or type(prior).__name__ == "Constraint"


def fixed_value(prior):
    """Value of a delta-function / plain-number prior, else None."""
    if isinstance(prior, (int, float, np.floating, np.integer)):
        return float(prior)
    if type(prior).__name__ == "DeltaFunction" or (hasattr(prior, "peak") and not hasattr(prior, "sample_chain")):
        peak = getattr(prior, "peak", None)
        if peak is not None:
            return float(peak)
    return None


class NMMALikelihoodMixin:
    """core/base.py:37-131."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    @property
    def priors(self):
        return self._priors

    @priors.setter
    def priors(self, value):
        self.constraints = value
        sampling_keys = [k for k in value.keys() if k not in self.constraints]
        self.check_parameter_equivalencies(sampling_keys)
        self._priors = value

    @property
    def constraints(self):
        return self._constraints

    @constraints.setter
    def constraints(self, value):
        if is_constraint(value):
            constr = {value.name: value}
        elif hasattr(value, "items"):
            constr = {k: v for k, v in value.items() if is_constraint(v)}
        else:
            constr = {}
        self._constraints = constr

    def evaluate_constraints(self, out_sample):
        return np.prod([con.prob(out_sample[k]) for k, con in self.constraints.items()])

    def identity_conversion(self, parameters):
        return parameters

    def __call__(self, parameters):
        return np.exp(self.log_likelihood(parameters))

    def log_likelihood(self, parameters):
        parameters = self.parameter_conversion(parameters)
        if self.evaluate_constraints(parameters) and self.sanity_checks():
            return self.sub_log_likelihood(parameters)
        return np.nan_to_num(-np.inf)

    def sanity_checks(self):
        return True

    def check_parameter_equivalencies(self, parameter_names):
        """core/base.py:112-130."""
        for group in [["inclination_EM", "KNtheta", "theta_jn", "cos_theta_jn", "thetaObs"]]:
            inter = set(parameter_names).intersection(group)
            if len(inter) > 1:
                raise ValueError(f"Multiple equivalent parameters found: {inter}. Please only provide one of these.")
        for group in [["redshift", "luminosity_distance", "Hubble_constant"],
                      ["mass_1", "mass_1_source", "chirp_mass", "mass_ratio", "eta", "mass_2", "mass_2_source"]]:
            inter = set(parameter_names).intersection(group)
            if len(inter) > 2:
                raise ValueError(f"Mutually dependent parameters found: {inter}. Please only provide up to two of these.")


class NMMALikelihood(NMMALikelihoodMixin, Likelihood):
    """core/base.py:133-185."""

    def __init__(self, sub_model, priors, **kwargs):
        super().__init__()
        self.sub_model = sub_model
        try:
            self._noise_logl = self.sub_model.noise_log_likelihood()
        except AttributeError:
            self._noise_logl = 0.0
        self.conv_functions = []
        self.priors = priors
        self.setup_submodel_conversion()

    def __repr__(self):
        return self.__class__.__name__ + " with " + self.sub_model.__repr__()

    def setup_submodel_conversion(self):
        pass

    def parameter_conversion(self, parameters):
        for conv in reversed(self.conv_functions):
            parameters = conv(parameters)
        return parameters

    def posterior_conversion(self, parameters):
        return self.parameter_conversion(parameters)

    def sub_log_likelihood(self, parameters):
        logl = self.sub_model.log_likelihood(parameters)
        if not np.isfinite(logl):
            return np.nan_to_num(-np.inf)
        return logl

    def noise_log_likelihood(self):
        return self._noise_logl
