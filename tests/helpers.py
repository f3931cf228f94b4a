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
                    sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"],
                    data=case["data"], observed_filters=obs,
                    sources=resolve_sources(obs, case["model_filters"],
                                            known_filters=[f for f in obs if f not in FILTER_AVERAGES]),
                    detection_limit=lim, systematics=case["systematics"], device=device)


def oracle_from_case(case, **kw):
    from tools.make_golden import build_oracle_likelihood
    return build_oracle_likelihood(case, **kw)


def rel_err(got, want):
    got, want = np.asarray(got, float), np.asarray(want, float)
    return np.abs(got - want) / np.maximum(1.0, np.abs(want))
