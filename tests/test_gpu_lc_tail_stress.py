"""The generic likelihood-from-curves kernel (``em_lc_loglike``) on shapes that take its less-travelled code paths, against the CPU
oracle (``oracle/nmma_oracle.py``: OracleLikelihood over a model that returns the SAME externally supplied curves):

* more sampled columns than a sample's spare LDS holds (the scalar chains then read theta from memory),
* more upper limits than the queue of the general term holds (the overflow is evaluated in place),
* more photometry than is staged in LDS (a wave per sample, data read from memory),
* a sampled ``em_syserr`` (the general term reads theta), two-node grids, a single filter, ragged batch sizes,
* one, two and three sets of curves summed on chip, with non-finite nodes.
"""
import os

import numpy as np
import pytest

from oracle import nmma_oracle as orc

pytestmark = pytest.mark.gpu
FLOOR = -1.7976931348623157e308


class _TableModel:
    """Oracle-side model whose source-frame curves are given per row (the reference's container would get them from afterglowpy or a
    second surrogate): ``parameters["_row"]`` selects the row; the detector-frame transform is gen_detector_lc's (model.py:352-404)."""

    model_parameters = []

    def __init__(self, filters, sample_times, cosmo_grid, stacked):
        self.filters, self.model_times, self.cosmo_grid, self.stacked = list(filters), np.asarray(sample_times, float), cosmo_grid, stacked
        self.good_parameters = True

    def parameter_conversion(self, parameters):
        return parameters

    def gen_detector_lc(self, parameters, sample_times=None):
        st = self.model_times
        row = self.stacked[int(parameters["_row"])]
        z = orc.redshift_from_parameters(parameters, self.cosmo_grid)
        obs = st * (1 + z) + parameters.get("timeshift", 0.0)
        rc = -2.5 * np.log10(1 + z)
        dm = orc.distance_modulus_nmma(parameters.get("luminosity_distance", 1e-5))
        out = {}
        for k, f in enumerate(self.filters):
            v = row[k]
            out[f] = (v + dm + rc) if np.isfinite(v).sum() >= 2 else np.full_like(obs, np.inf)
        return obs, out


def _stack(sets, st):
    """stack_magnitudes with its per-model gap filling (model.py:1440-1448, :1486-1510), in numpy."""
    from scipy.special import logsumexp
    ln10 = np.log(10.0)
    filled = []
    for s in sets:
        f = np.full_like(s, np.inf)
        for b in range(s.shape[0]):
            for m in range(s.shape[1]):
                fin = np.isfinite(s[b, m])
                if fin.any():
                    f[b, m] = np.interp(st, st[fin], s[b, m][fin], left=np.inf, right=np.inf)
        filled.append(f)
    with np.errstate(invalid="ignore", divide="ignore"):
        return -2.5 * logsumexp([-0.4 * ln10 * f for f in filled], axis=0) / ln10


def _cosmo():
    d = np.linspace(1.0, 400.0, 60)
    return d, d * 2.3e-4 * (1.0 + 1e-4 * d)


SHAPES = {
    # name: (filters, points per filter, NS, extra sampled columns, upper limits per filter, systematics)
    "more_columns_than_spare_lds": (1, 1, 2, 9, 0, "budget"),
    "upper_limits_overflow_the_queue": (3, 40, 3, 0, 40, "budget"),
    "photometry_not_staged": (2, 600, 25, 0, 7, "budget"),
    "sampled_syserr": (4, 17, 41, 1, 3, "param"),
    "ordinary": (5, 23, 30, 0, 2, "budget"),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_lc_tail_against_the_oracle(shape):
    import torch
    from nmma_amd.engine import EMEngine
    nfilt, npts, NS, extra, n_ul, sysmode = SHAPES[shape]
    rng = np.random.default_rng(sum(map(ord, shape)))
    filters = [f"band{k}" for k in range(nfilt)]
    st = np.linspace(0.2, 14.0, NS)
    names = ["luminosity_distance", "timeshift"] + [f"spare_{i}" for i in range(extra)]
    if sysmode == "param":
        names[-1] = "em_syserr"
    times, mags, sigmas = {}, {}, {}
    for f in filters:
        t = np.sort(rng.uniform(0.9, 13.0, npts))            # inside the model window of every row but the shifted ones below
        times[f] = t
        mags[f] = rng.uniform(17.0, 22.0, npts)
        s = rng.uniform(0.05, 0.3, npts)
        s[rng.choice(npts, size=min(n_ul, npts), replace=False)] = np.inf
        sigmas[f] = s
    systematics = dict(mode="budget", values={f: 0.7 for f in filters}) if sysmode == "budget" else dict(mode="param", name="em_syserr")
    cosmo = _cosmo()
    eng = EMEngine(None, filters, [], names, sample_times=st, cosmo_grid=cosmo, data=(times, mags, sigmas), observed_filters=filters,
                   systematics=systematics, detection_limit={f: np.inf for f in filters}, model_kind="external")
    for B in (1, 7, 64, 131):
        theta = np.column_stack([rng.uniform(20.0, 300.0, B), rng.uniform(-0.3, 0.4, B)] + [rng.uniform(0.1, 1.0, B) for _ in range(extra)])
        theta[5::11, 1] = 3.0                               # the window starts after the first epochs: NaN terms -> floor
        for n_sets in (1, 2, 3):
            sets = []
            for k in range(n_sets):
                a = rng.uniform(-17.0, -11.0, (B, nfilt, NS)) + 1.5 * k
                if NS > 4:
                    hole = rng.uniform(size=a.shape) < 0.04
                    a[hole] = rng.choice([np.inf, np.nan], size=int(hole.sum()))
                sets.append(a)
            stacked = _stack(sets, st) if n_sets > 1 else np.where(np.isnan(sets[0]), np.nan, sets[0])
            model = _TableModel(filters, st, cosmo, stacked)
            lik = orc.OracleLikelihood(model, (times, mags, sigmas), systematics, filters, detection_limit=np.inf, use_scipy=False)
            want = np.empty(B)
            for b in range(B):
                p = dict(zip(names, theta[b]))
                p["_row"] = b
                want[b] = lik.log_likelihood(p)
            th = torch.as_tensor(theta, device="cuda:0")
            dev_sets = [torch.as_tensor(s, device="cuda:0") for s in sets]
            for grp in (None, "16", "32"):
                eng.set_option("lc_group", int(grp or 0))
                try:
                    got = (eng.loglike_lc_sets(th, dev_sets) if n_sets > 1 else eng.loglike_lc(th, dev_sets[0])).cpu().numpy()
                finally:
                    eng.set_option("lc_group", 0)
                floor = ~np.isfinite(want) | (want <= FLOOR)
                assert np.array_equal(got == FLOOR, floor), (shape, B, n_sets, grp)
                ok = ~floor
                assert B < 7 or ok.sum() >= B // 2, (shape, B, n_sets, "the case is meant to have finite rows")
                if ok.any():
                    err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
                    assert err.max() < 1e-11, (shape, B, n_sets, grp, err.max())
    eng.close()

