#!/usr/bin/env python
"""A surrogate with REAL conditioning for the golden set: the SVD basis the reference's own training code builds from the 28
POSSIS light curves its tests hold (``nmma/tests/data/bulla/*.dat``: a 4 x 7 grid in ejecta masses, 9 filters, 99 epochs).

Runs only in the build container (imports the reference from /root/reference under oracle/ref_harness.py).  Steps:

1. ``nmma.em.io.read_photometry_files`` + ``nmma.em.model_parameters.Bu2019lm_sparse`` read the grid (the reference's own
   readers; training.py's test does the same, nmma/tests/training.py:43-48);
2. ``BaseTrainingModel.interpolate_data`` and ``BaseTrainingModel.generate_svd_model`` (nmma/em/training.py:163-265, called
   unbound on an attribute-only instance: the constructor wants keras and a model directory) give, per filter, ``VA``,
   ``mins``, ``maxs``, ``param_mins``, ``param_maxs`` and the training targets ``cAmat`` on the documented training grid
   ``--tmin 0 --tmax 21 --dt 0.1`` (doc/training.md:49);
3. the fp32 network ``Dense(2 -> 2048, relu) -> Dense(2048 -> 10)`` (architecture: training.py:353-364) is NOT trained with
   keras (absent): layer 1 is a seeded random-feature layer whose kinks cross the unit square, layer 2 is the ridge
   least-squares solution onto ``cAmat`` -- so the net really predicts the SVD coefficients of the grid (checked below).

Output: tests/golden/bulla_svd_model.npz -- ARRAYS ONLY (tensors of the model; no reference source, no raw data file).
``tests/cases.py:case_bulla_svd`` builds the parity case from it and ``tools/make_golden.py bulla_svd`` then runs the
reference's likelihood on it like on every other case.
"""
import glob
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402

N_COEFF, N_HIDDEN, SEED = 10, 2048, 20260401
FILTERS = ["sdssu", "ztfg", "ztfr", "ztfi", "ps1::z", "ps1::y", "2massj", "2massh", "2massks"]


def reference_svd_model():
    ref_harness.reference_modules()
    io = importlib.import_module("nmma.em.io")
    mp = importlib.import_module("nmma.em.model_parameters")
    tr = importlib.import_module("nmma.em.training")
    files = sorted(glob.glob(os.path.join(ref_harness.REFERENCE_ROOT, "nmma/tests/data/bulla/*.dat")))
    assert len(files) == 28, files
    data, parameters = mp.Bu2019lm_sparse(io.read_photometry_files(files))
    t = object.__new__(tr.BaseTrainingModel)
    t.data, t.model_parameters, t.filters = data, parameters, list(FILTERS)
    t.sample_times = np.arange(0.0, 21.1, 0.1)
    t.n_coeff, t.data_type, t.time_scale_factor = N_COEFF, "photometry", 1.0
    t.univariate_spline, t.univariate_spline_s = False, 2
    tr.BaseTrainingModel.interpolate_data(t)
    return parameters, tr.BaseTrainingModel.generate_svd_model(t)


def fit_network(x, camat, rng):
    """Random-feature layer 1 (every hidden unit's kink passes through a random point of the unit square), ridge
    least squares for layer 2, everything rounded to fp32 as the saved keras weights are."""
    n_p = x.shape[1]
    w1 = (1.5 * rng.standard_normal((n_p, N_HIDDEN))).astype(np.float32)
    x0 = rng.uniform(-0.2, 1.2, (N_HIDDEN, n_p))
    b1 = (-np.einsum("hp,ph->h", x0, w1.astype(np.float64))).astype(np.float32)
    h = np.maximum(x.astype(np.float32) @ w1 + b1, 0).astype(np.float64)
    a = np.hstack([h, np.ones((len(x), 1))])
    lam = 1e-6
    # min-norm ridge solution in the dual form (28 training points, 2049 features)
    sol = a.T @ np.linalg.solve(a @ a.T + lam * np.eye(len(x)), camat.T)
    return w1, b1, sol[:-1].astype(np.float32), sol[-1].astype(np.float32)


def main():
    parameters, svd = reference_svd_model()
    rng = np.random.default_rng(SEED)
    out = {"filters": np.array(FILTERS), "model_parameters": np.array(parameters)}
    worst = 0.0
    for f in FILTERS:
        m = svd[f]
        x = np.asarray(m["param_array_postprocess"], float)
        w1, b1, w2, b2 = fit_network(x, m["cAmat"], rng)
        pred = (np.maximum(x.astype(np.float32) @ w1 + b1, 0) @ w2 + b2).astype(np.float64)       # [28, NC]
        err = np.max(np.abs(pred - m["cAmat"].T))
        # how well the curves of the grid come back (mag): reconstruction from the fitted coefficients vs the data
        va = m["VA"][:, :N_COEFF]
        span = m["maxs"] - m["mins"]
        worst = max(worst, err)
        sv = np.linalg.svd(va * span[:, None], compute_uv=False)
        print(f"{f:8s} coefficient fit max|err| {err:.2e}; cAmat range [{m['cAmat'].min():+.2f}, {m['cAmat'].max():+.2f}]; "
              f"span [{span.min():.2f}, {span.max():.2f}] mag; cond(VA*span) {sv[0] / sv[-1]:.1f}; |W2| max {np.abs(w2).max():.3f}")
        for k, v in (("W1", w1), ("b1", b1), ("W2", w2), ("b2", b2), ("VA", np.ascontiguousarray(va)), ("mins", m["mins"]),
                     ("maxs", m["maxs"]), ("tt", m["tt"]), ("param_mins", m["param_mins"]), ("param_maxs", m["param_maxs"])):
            out[f"{f}/{k}"] = np.asarray(v)
    path = os.path.join(ROOT, "tests", "golden", "bulla_svd_model.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; worst coefficient error", worst)


if __name__ == "__main__":
    main()
