"""The lock-step ensemble walk on the device (nmma_walk_propose / nmma_walk_accept / nmma_walk_rescale) against the host walk of
nmma_amd/sampler.py: same counter-hash random numbers, same proposal, the analytic bilby priors' transforms, and the whole walk
through the config-2 EM likelihood."""
import ctypes as C
import time

import numpy as np
import pytest

from nmma_amd import sampler as smp
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import PowerLawPrior, SimplePrior, UniformPrior, plugin_from_case

pytestmark = pytest.mark.gpu


class Sine(SimplePrior):
    def rescale(self, val):
        norm = 1 / (np.cos(self.minimum) - np.cos(self.maximum))
        return np.arccos(np.cos(self.minimum) - val / norm)


class Cosine(SimplePrior):
    def rescale(self, val):
        norm = 1 / (np.sin(self.maximum) - np.sin(self.minimum))
        return np.arcsin(val / norm + np.sin(self.minimum))


class LogUniform(SimplePrior):
    def rescale(self, val):
        return self.minimum * np.exp(val * np.log(self.maximum / self.minimum))


class Gaussian(SimplePrior):
    def __init__(self, mu, sigma):
        super().__init__(-np.inf, np.inf)
        self.mu, self.sigma = mu, sigma

    def rescale(self, val):
        from scipy.special import erfinv
        return self.mu + erfinv(2 * val - 1) * 2 ** 0.5 * self.sigma


class DeltaFunction(SimplePrior):
    def __init__(self, peak):
        super().__init__(peak=peak)

    def rescale(self, val):
        return self.peak * np.ones_like(val)


def _priors():
    pri = {"a": UniformPrior(-2.0, 5.0), "b": Sine(0.0, np.pi), "c": Cosine(-np.pi / 2, np.pi / 2), "d": PowerLawPrior(2.0, 10.0, 300.0),
           "e": LogUniform(1e-3, 1e2), "f": Gaussian(1.5, 0.3), "g": DeltaFunction(0.7)}
    return pri, list(pri)


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def test_prior_transform_matches_bilbys_formulas(torch_cuda):
    torch = torch_cuda
    from nmma_amd import _lib as L
    lib = L.load_library()
    pri, keys = _priors()
    table = smp.device_prior_table(pri, keys)
    assert table is not None
    u = np.random.default_rng(1).uniform(1e-6, 1 - 1e-6, (5000, len(keys)))
    ud = torch.as_tensor(u, device="cuda:0")
    out = torch.empty_like(ud)
    L.check(lib.nmma_walk_rescale(table, len(keys), C.c_void_p(ud.data_ptr()), len(u), C.c_void_p(out.data_ptr()), 0, None), "rescale")
    got = out.cpu().numpy()
    want = np.stack([np.asarray(pri[k].rescale(u[:, i]), dtype=float) for i, k in enumerate(keys)], axis=1)
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 1e-12
    assert smp.device_prior_table({"x": object()}, ["x"]) is None           # no device formula: the caller stays on the host


def test_one_step_is_the_host_walks_step(torch_cuda):
    """Same counter-hash uniforms, same differential-evolution proposal, same boundary handling, same inside-the-cube mask."""
    torch = torch_cuda
    from nmma_amd import _lib as L
    lib = L.load_library()
    rng = np.random.default_rng(5)
    n, ndim, n_live = 3000, 6, 500
    live, u = rng.uniform(size=(n_live, ndim)), rng.uniform(size=(n, ndim))
    pri = {f"p{i}": UniformPrior(0.0, 1.0) for i in range(ndim)}
    w = smp.EnsembleWalkSampler(ndim=ndim, periodic=[1], reflective=[3], walks=10)
    table = smp.device_prior_table(pri, list(pri), w.periodic, w.reflective)
    keys = rng.integers(1, 2 ** 62, n).astype(np.uint64)
    for step in (1, 7):
        r = smp.counter_uniforms(keys, np.full(n, step, dtype=np.uint64))
        want, inside = w._propose(u, live, r)
        t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), device="cuda:0")
        ud, ld, kd = t(u), t(live), t(keys.view(np.int64))
        prop, theta, ins = torch.empty_like(ud), torch.empty_like(ud), torch.empty(n, dtype=torch.int32, device="cuda:0")
        p = lambda x: C.c_void_p(x.data_ptr())
        L.check(lib.nmma_walk_propose(table, ndim, p(ld), n_live, p(ud), p(ud), p(kd), n, step, p(prop), p(theta), p(ins), 0, None), "propose")
        assert np.array_equal(ins.cpu().numpy().astype(bool), inside)
        assert np.max(np.abs(prop.cpu().numpy() - want)) < 1e-13
        assert 0.2 < inside.mean() < 1.0


def test_whole_walk_through_the_em_likelihood(torch_cuda):
    """4096 chains x 25 steps of the config-2 likelihood on the device against the host walk of the same queue: every returned
    point is what the likelihood says it is, beats its bound, and the two walks accept the same fraction of steps."""
    torch = torch_cuda
    case = syn.config2_case()
    _, _, lik = plugin_from_case(case)
    names = case["names"]
    lo, hi = syn.draw_theta(3, 20000, names)[1].min(axis=0), syn.draw_theta(3, 20000, names)[1].max(axis=0)
    pri = {k: UniformPrior(float(a), float(b)) for k, a, b in zip(names, lo, hi)}
    pt = smp.BatchedPriorTransform(pri, names)
    rng = np.random.default_rng(11)
    n, walks = 4096, 25
    live = rng.uniform(0.3, 0.7, (n, len(names)))
    l_live = lik.log_likelihood_batch(torch.as_tensor(pt(live), device="cuda:0"), names).cpu().numpy()
    bound = np.quantile(l_live, 0.2)
    kw = dict(live=live, walks=walks)
    args = [smp.SamplerArgument(live[i].copy(), bound, 1000 + i, pt, None, kw) for i in range(n)]
    w = smp.EnsembleWalkSampler(ndim=len(names), walks=walks, naccept=10)
    dev_ll = lambda th: lik.log_likelihood_batch(th, names)
    host_ll = lambda th: lik.log_likelihood_batch(torch.as_tensor(np.ascontiguousarray(th), device="cuda:0"), names).cpu().numpy()
    w.run_many_device(args[:64], dev_ll, pri, names)                   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = w.run_many_device(args, dev_ll, pri, names)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = w.run_many(args, host_ll, pt)
    t_host = time.perf_counter() - t0
    # the same walk with the engine called directly (no plugin wrapper around the launch)
    eng = lik.sub_model.engine(names)
    buf = torch.empty(n, dtype=torch.float64, device="cuda:0")
    raw_ll = lambda th: eng.loglike(th, out=buf)
    w.run_many_device(args[:64], raw_ll, pri, names)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    raw = w.run_many_device(args, raw_ll, pri, names)
    t_raw = time.perf_counter() - t0
    assert np.array_equal(np.array([g[2] for g in raw]), np.array([g[2] for g in got]))
    # the steps themselves (run_many_device adds the queue's bookkeeping: 4096 SamplerReturn records are ~5 ms of Python)
    table = smp.device_prior_table(pri, names)
    keys = np.array([smp.chain_key(a.rseed) for a in args], dtype=np.uint64)
    smp.device_walk(table, live, live, np.full(n, bound), keys, 10, raw_ll)
    t0 = time.perf_counter()
    smp.device_walk(table, live, live, np.full(n, bound), keys, 400, raw_ll)
    t_steps = (time.perf_counter() - t0) / 400
    print(f"device_walk alone: {t_steps * 1e6:.1f} us per step of {n} chains = {n / t_steps / 1e6:.1f} M likelihood evaluations/s")
    print(f"device walk {t_dev / walks * 1e6:.0f} us per step through the plugin, {t_raw / walks * 1e6:.0f} us with the engine called directly; "
          f"host walk {t_host / walks * 1e6:.0f} us per step ({n} chains, {n * walks / t_raw / 1e6:.1f} M evals/s)")
    u = np.stack([g[0] for g in got]); v = np.stack([g[1] for g in got]); logl = np.array([g[2] for g in got])
    assert np.all((u >= 0) & (u <= 1)) and np.allclose(v, pt(u), rtol=1e-12, atol=1e-12)
    again = host_ll(v)
    assert np.array_equal(again, logl) and np.all(logl > bound)
    acc_dev = np.mean([g[4]["accept"] for g in got]) / walks
    acc_host = np.mean([g[4]["accept"] for g in ref]) / walks
    assert abs(acc_dev - acc_host) < 0.01 and 0.05 < acc_dev < 0.95
    # (no assertion on the timings: a garbage collection of the Python process -- ~75 ms with torch imported -- can land in either loop)


def test_pool_map_takes_the_device_walk(torch_cuda):
    """``GPUPool.map(walker.sample, queue)`` -- the call an unmodified dynesty makes -- runs the walk on the device when the pool was
    given the sampled priors, and on the host when a prior has no device formula; both return dynesty's records."""
    torch = torch_cuda
    from nmma_amd.pool import GPUPool
    case = syn.config2_case()
    _, _, lik = plugin_from_case(case)
    names = case["names"]
    th = syn.draw_theta(3, 20000, names)[1]
    pri = {k: UniformPrior(float(a), float(b)) for k, a, b in zip(names, th.min(axis=0), th.max(axis=0))}
    pt = smp.BatchedPriorTransform(pri, names)
    rng = np.random.default_rng(12)
    n, walks = 600, 12
    live = rng.uniform(0.3, 0.7, (n, len(names)))
    kw = dict(live=live, walks=walks)
    args = [smp.SamplerArgument(live[i].copy(), -1e6, 50 + i, pt, None, kw) for i in range(n)]
    w = smp.EnsembleWalkSampler(ndim=len(names), walks=walks, naccept=5)
    pool = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt, priors=pri)
    got = pool.map(w.sample, args)
    assert pool.n_batches == walks and len(got) == n and all(len(g) == 5 for g in got)
    host_pool = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt)
    ref = host_pool.map(w.sample, args)
    assert host_pool.n_batches >= walks
    # with a bound far below every likelihood value each inside-the-cube proposal is accepted: the two walks make the SAME moves
    # (same random numbers, same proposals) up to the last bits of log / the prior transform
    du = np.max(np.abs(np.stack([g[0] for g in got]) - np.stack([r[0] for r in ref])))
    assert du < 1e-9
    assert [g[4]["accept"] for g in got] == [r[4]["accept"] for r in ref]

    # chains queued with different walk lengths share the launches: each stops after its own
    args_mixed = [smp.SamplerArgument(live[i].copy(), -1e6, 50 + i, pt, None, dict(live=live, walks=4 + (i % 9))) for i in range(n)]
    got_m, ref_m = pool.map(w.sample, args_mixed), host_pool.map(w.sample, args_mixed)
    assert [g[4]["walks"] for g in got_m] == [4 + (i % 9) for i in range(n)]
    assert [g[4]["accept"] for g in got_m] == [r[4]["accept"] for r in ref_m]
    assert np.max(np.abs(np.stack([g[0] for g in got_m]) - np.stack([r[0] for r in ref_m]))) < 1e-9

    class Odd(UniformPrior):            # a prior the device has no formula for: the pool falls back to the host walk
        pass
    Odd.__name__ = "Tabulated"
    pri2 = dict(pri); pri2[names[0]] = Odd(pri[names[0]].minimum, pri[names[0]].maximum)
    pool2 = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt, priors=pri2)
    again = pool2.map(w.sample, args)
    assert [g[4]["accept"] for g in again] == [r[4]["accept"] for r in ref]


def test_acceptance_tracking_rwalk_on_the_device(torch_cuda):
    """``sample="rwalk"``: the per-chain autocorrelation estimate and stop rule evaluated in the accept kernel give the host walk's
    chains -- same accept / reject counts, same number of likelihood calls per chain, same estimate left behind for the next queue
    (first queue: no previous estimate; second queue: smoothed with the first's)."""
    torch = torch_cuda
    ndim, n, n_live = 5, 1500, 1500
    pri = {f"p{i}": UniformPrior(-1.0 - i, 2.0 + i) for i in range(ndim)}
    names = list(pri)
    pt = smp.BatchedPriorTransform(pri, names)
    lo = np.array([pri[k].minimum for k in names]); width = np.array([pri[k].maximum - pri[k].minimum for k in names])
    host_ll = lambda th: -np.sum(((np.asarray(th) - lo) / width - 0.5) ** 2, axis=1) * 40.0
    lo_d, w_d = torch.as_tensor(lo, device="cuda:0"), torch.as_tensor(width, device="cuda:0")
    dev_ll = lambda th: -torch.sum(((th - lo_d) / w_d - 0.5) ** 2, dim=1) * 40.0
    rng = np.random.default_rng(21)
    live = 0.5 + 0.12 * rng.standard_normal((n_live, ndim))
    live = np.clip(live, 0.01, 0.99)
    bound = np.quantile(host_ll(pt(live)), 0.3)
    kw = dict(live=live, nlive=n_live)
    args = [smp.SamplerArgument(live[i].copy(), bound, 7000 + i, pt, None, kw) for i in range(n)]
    cls = smp.AcceptanceTrackingRWalk
    results = {}
    for where in ("host", "device"):
        cls.old_act = None
        w = cls(ndim=ndim, periodic=[0], reflective=[2], maxmcmc=400, nact=6)
        out = []
        for _ in range(2):
            if where == "host":
                out.append(w.run_many(args, host_ll, pt))
            else:
                out.append(w.run_many_device(args, dev_ll, pri, names))
            out.append(cls.old_act)
        results[where] = out
    cls.old_act = None
    for q in (0, 2):
        got, ref = results["device"][q], results["host"][q]
        same = [g[4] == r[4] and g[3] == r[3] for g, r in zip(got, ref)]
        assert np.mean(same) > 0.995                     # (a proposal within an ulp of the bound may fall either way)
        ok = np.nonzero(same)[0]
        du = np.max(np.abs(np.stack([got[i][0] for i in ok]) - np.stack([ref[i][0] for i in ok])))
        assert du < 1e-9
        assert results["device"][q + 1] == pytest.approx(results["host"][q + 1], rel=1e-3)
        logl = np.array([g[2] for g in got])
        moved = np.array([g[4]["accept"] > 0 for g in got])       # (a chain that never moved returns a fresh prior draw)
        assert moved.mean() > 0.9 and np.all(logl[moved] > bound)
        assert np.allclose(logl, host_ll(np.stack([g[1] for g in got])), rtol=1e-12)
    lengths = np.array([g[3] for g in results["device"][0]])
    assert lengths.min() < lengths.max()                 # chains stop at their own times
    assert results["device"][1] != results["device"][3]  # and the second queue was smoothed with the first's estimate

    # through the pool: the unmodified pool.map(sample, queue) takes the device walk for this walker too
    from nmma_amd.pool import GPUPool

    class _Lik:
        def log_likelihood_batch(self, th, names=None):
            return dev_ll(th)
    pool = GPUPool(_Lik(), queue_size=n, names=names, prior_transform_many=pt, priors=pri)
    cls.old_act = None
    w = cls(ndim=ndim, periodic=[0], reflective=[2], maxmcmc=400, nact=6)
    via_pool = pool.map(w.sample, args)
    cls.old_act = None
    assert [g[4] for g in via_pool] == [g[4] for g in results["device"][0]]
