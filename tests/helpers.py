"""Shared test helpers: build the HIP engine / the oracle from a tests.cases case."""
import numpy as np


def engine_from_case(case, device=0):
    from nmma_amd.engine import EMEngine
    from nmma_amd.em.utils import resolve_sources, FILTER_AVERAGES
    obs = list(case["observed_filters"])
    lim = case["detection_limit"]
    if not isinstance(lim, dict):
        lim = {f: float(lim) for f in obs}
    return EMEngine(case["svd"], case["model_filters"], case["model_parameters"], case["names"],
                    fixed=case.get("fixed"), sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"],
                    data=case["data"], observed_filters=obs,
                    sources=resolve_sources(obs, case["model_filters"],
                                            known_filters=[f for f in obs if f not in FILTER_AVERAGES]),
                    detection_limit=lim, systematics=case["systematics"], ebv_coeff=case.get("ebv_coeff"),
                    filter_nu0=case.get("filter_nu0"),
                    extinction_law="P92_SMC_host" if case.get("filter_nu0") is not None else None,
                    device=device)


def oracle_from_case(case, **kw):
    from tools.make_golden import build_oracle_likelihood
    return build_oracle_likelihood(case, **kw)


def rel_err(got, want):
    got, want = np.asarray(got, float), np.asarray(want, float)
    return np.abs(got - want) / np.maximum(1.0, np.abs(want))


class SimplePrior:
    """Minimal stand-in for a bilby prior (bilby is not installed in this image)."""

    def __init__(self, minimum=None, maximum=None, peak=None):
        self.minimum, self.maximum = minimum, maximum
        if peak is not None:
            self.peak = peak


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
