"""Shared test helpers: build the HIP engine / the oracle from a tests.cases case."""
import numpy as np


def engine_from_case(case, device=0):
    from nmma_amd.engine import EMEngine
    return EMEngine.from_case(case, device=device)


def oracle_from_case(case, **kw):
    from oracle.nmma_oracle import likelihood_from_case
    return likelihood_from_case(case, **kw)


def rel_err(got, want):
    got, want = np.asarray(got, float), np.asarray(want, float)
    return np.abs(got - want) / np.maximum(1.0, np.abs(want))


class SimplePrior:
    """Minimal stand-in for a bilby prior (bilby is not installed in this image)."""

    def __init__(self, minimum=None, maximum=None, peak=None):
        self.minimum, self.maximum = minimum, maximum
        if peak is not None:
            self.peak = peak


class UniformPrior(SimplePrior):
    """bilby.core.prior.Uniform(minimum, maximum): ``prob`` as bilby defines it."""

    def prob(self, val):
        val = np.asarray(val, dtype=np.float64)
        return np.where((val >= self.minimum) & (val <= self.maximum), 1.0 / (self.maximum - self.minimum), 0.0)

    def rescale(self, val):
        return self.minimum + val * (self.maximum - self.minimum)


class PowerLawPrior(SimplePrior):
    """bilby.core.prior.PowerLaw(alpha, minimum, maximum): p(x) ~ x^alpha (alpha = 2: uniform in volume, bilby's usual
    luminosity-distance prior) -- ``prob`` and ``rescale`` as bilby defines them (bilby/core/prior/analytical.py)."""

    def __init__(self, alpha, minimum, maximum):
        super().__init__(minimum, maximum)
        self.alpha = alpha

    def prob(self, val):
        val = np.asarray(val, dtype=np.float64)
        a1 = 1 + self.alpha
        inside = (val >= self.minimum) & (val <= self.maximum)
        return np.where(inside, val ** self.alpha * a1 / (self.maximum ** a1 - self.minimum ** a1), 0.0)

    def rescale(self, val):
        a1 = 1 + self.alpha
        return (self.minimum ** a1 + val * (self.maximum ** a1 - self.minimum ** a1)) ** (1.0 / a1)


def plugin_from_case(case, device=0, verbose=False):
    """The reference-shaped objects (model, systematics handler, likelihood) for a case."""
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    priors = {n: SimplePrior(0.0, 1.0) for n in case["names"]}
    model = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=case["model_filters"],
                               model_parameters=case["model_parameters"], sample_times=case["sample_times"],
                               cosmo_grid=case["cosmo_grid"], device=device)
    times, mags, sigmas = case["data"]
    kw = case["systematics_ref"]
    handler = FilterSystematicsHandler(case["observed_filters"], systematics_file=kw["systematics_file"],
                                       error_budget=kw["error_budget"], light_curve_times=times)
    lik = EMTransientLikelihood(model, (times, mags, sigmas, 0.0), handler, priors,
                                filters=case["observed_filters"], detection_limit=case["detection_limit"],
                                verbose=verbose)
    return model, handler, lik
