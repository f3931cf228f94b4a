"""BASELINE config 1: Me2017 analytic kilonova, 3 filters (g, r, i), 50 epochs per filter,
128 prior samples (priors/Me2017.prior box), sample_times = arange(0.1, 14.5, 0.5)."""
import numpy as np

from nmma_amd import synthetic as syn

FILTERS = ["g", "r", "i"]
NAMES = ["luminosity_distance", "beta", "log10_kappa_r", "timeshift", "log10_vej", "log10_mej"]


def case_me2017(seed=4321, batch=128, epochs=50):
    rng = np.random.default_rng(seed)
    grid = syn.flat_lcdm_grid(1.0, 200.0)
    sample_times = np.arange(0.1, 14.5, 0.5)
    theta = np.stack([rng.uniform(1.0, 200.0, batch), rng.uniform(1.0, 5.0, batch), rng.uniform(-1.0, 2.0, batch),
                      rng.uniform(-2.0, 1.0, batch), rng.uniform(-2.0, -0.5, batch), rng.uniform(-3.0, -0.5, batch)],
                     axis=1)
    times, mags, sigmas = {}, {}, {}
    for k, f in enumerate(FILTERS):
        t = np.sort(rng.uniform(1.5, 11.0, epochs))
        sig = rng.uniform(0.02, 0.2, epochs)
        times[f] = t
        mags[f] = 19.0 + 0.35 * t + 0.3 * k + sig * rng.standard_normal(epochs)
        sigmas[f] = sig
    sigmas["r"][7] = np.inf
    return dict(filters=FILTERS, sample_times=sample_times, cosmo_grid=grid, data=(times, mags, sigmas),
                names=NAMES, theta=theta)


def oracle_likelihood(case, use_scipy=True):
    from oracle import me2017_oracle as meo
    from oracle import nmma_oracle as orc
    model = meo.OracleMe2017Model(case["filters"], case["sample_times"], cosmo_grid=case["cosmo_grid"])
    return orc.OracleLikelihood(model, case["data"], dict(mode="budget", values={f: 1.0 for f in case["filters"]}),
                                case["filters"], detection_limit=np.inf, known_filters=case["filters"],
                                use_scipy=use_scipy)
