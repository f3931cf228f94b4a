#!/usr/bin/env python
"""Golden vectors for BASELINE config 1 (Me2017 analytic kilonova, 3 filters x 50 epochs,
128 prior samples) from the REFERENCE'S OWN SOURCE under oracle/ref_harness.py.
Output: tests/golden/me2017.npz (logl[128]; light curves of the first 4 rows)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import me2017_oracle as meo  # noqa: E402
from oracle import nmma_oracle as orc  # noqa: E402
from oracle import ref_harness  # noqa: E402
from tests import cases_me2017  # noqa: E402


def build_reference(case):
    ref = ref_harness.reference_modules()
    ref.utils.get_all_bandpass_metadata = lambda: []
    ref.utils.M4OPT_INSTALLED = False        # the mocked import 'succeeds'; the real package is absent
    m = ref.model.SimpleKilonovaLightCurveModel("Me2017", filters=list(case["filters"]),
                                                sample_times=case["sample_times"])
    grid = case["cosmo_grid"]
    m.redshift_func = lambda p: np.interp(p["luminosity_distance"], grid[0], grid[1])
    m.check_vs_priors = lambda priors: None
    times, mags, sigmas = case["data"]
    priors = ref.base.PriorDict({n: object() for n in case["names"]})
    handler = ref.systematics.FilterSystematicsHandler(list(case["filters"]), error_budget=1.0,
                                                       light_curve_times=times)
    lik = ref.em_likelihood.EMTransientLikelihood(m, (times, mags, sigmas, 0.0), handler, priors,
                                                  filters=list(case["filters"]), detection_limit=np.inf)
    return lik, m


def main():
    case = cases_me2017.case_me2017()
    lik, m = build_reference(case)
    names, theta = case["names"], case["theta"]
    out = {}
    logl = np.empty(len(theta))
    for i, row in enumerate(theta):
        p = dict(zip(names, (float(v) for v in row)))
        logl[i] = lik.log_likelihood(p)
        if i < 4:
            tobs, lc = m.gen_detector_lc(lik.parameter_conversion(dict(p)))
            out[f"s{i}_obs_times"] = np.asarray(tobs, float)
            for k, f in enumerate(case["filters"]):
                out[f"s{i}_app_{k}"] = np.asarray(lc[f], float)
    out["logl"] = logl
    olik = cases_me2017.oracle_likelihood(case)
    ol = orc.log_likelihood_batch(olik, names, theta)
    floor = logl == orc.LOGL_FLOOR
    assert np.array_equal(ol == orc.LOGL_FLOOR, floor)
    rel = np.max(np.abs(ol[~floor] - logl[~floor]) / np.maximum(1, np.abs(logl[~floor])))
    print(f"me2017: B={len(theta)} floor={floor.sum()} logL range ({logl[~floor].min():.2f}, {logl[~floor].max():.2f}) "
          f"oracle-vs-reference max rel diff {rel:.3e}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "me2017.npz"), **out)


if __name__ == "__main__":
    main()
