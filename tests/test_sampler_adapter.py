"""The sampler seam on the CPU.  A fake nested-sampling driver assembles the ``sample=`` object exactly as the reference does
(nmma/core/mpi_setup.py:202-245: by name from ``dy_utils`` with ``internal_kwargs``), gets its argument records from
``prepare_sampler``, maps ``sample`` over them through ``pool.map`` (:282-285, :339) and feeds ``tune`` -- with
``dy_utils = nmma_amd.sampler`` and a :class:`nmma_amd.pool.GPUPool` the queue becomes one likelihood batch per MCMC step."""
import collections
import types

import numpy as np
import pytest
from scipy import stats

import nmma_amd.sampler as dy_utils          # <- the one edit in mpi_setup.py (there: bilby's dynesty utilities)
from nmma_amd.pool import GPUPool
from nmma_amd.sampler import BatchedPriorTransform, LockstepEnsembleWalk, counter_uniforms

Args = collections.namedtuple("Args", "u loglstar rseed prior_transform loglikelihood kwargs")


class _Uniform:
    boundary = None

    def __init__(self, lo, hi, boundary=None):
        self.minimum, self.maximum, self.boundary = lo, hi, boundary

    def rescale(self, u):
        return self.minimum + (self.maximum - self.minimum) * np.asarray(u)


class _FakeLikelihood:
    """Stands in for EMTransientLikelihood: a Gaussian bump, counting launches."""

    def __init__(self):
        self.batches, self.evals, self.largest = 0, 0, 0

    def log_likelihood_batch(self, theta, names=None):
        theta = np.asarray(theta)
        self.batches += 1
        self.evals += len(theta)
        self.largest = max(self.largest, len(theta))
        return -0.5 * np.sum(((theta - 0.3) / 0.2) ** 2, axis=1)


def _reference_sampler_object(sample, priors, keys, nact=2, naccept=10, maxmcmc=200, walks=20):
    """mpi_setup.py:184-245 (_init_sampler_kwargs), line for line where it concerns the walker, with dy_utils swapped."""
    periodic, reflective = [], []
    for ii, key in enumerate(keys):
        if priors[key].boundary == "periodic":
            periodic.append(ii)
        elif priors[key].boundary == "reflective":
            reflective.append(ii)
    periodic = periodic or None
    reflective = reflective or None
    kwargs = dict(sample=sample, walks=walks, bound="live", ndim=len(keys), periodic=periodic, reflective=reflective)
    internal_kwargs = dict(ndim=kwargs["ndim"], nonbounded=None, periodic=kwargs["periodic"], reflective=kwargs["reflective"],
                           maxmcmc=maxmcmc)
    if kwargs["sample"] == "act-walk":
        internal_kwargs["nact"] = nact
        sample_meth = dy_utils.ACTTrackingEnsembleWalk(**internal_kwargs)
        _ = f"thinning by {sample_meth.thin} with maximum length {sample_meth.thin * sample_meth.maxmcmc}"
    elif kwargs["sample"] == "acceptance-walk":
        internal_kwargs["naccept"] = naccept
        internal_kwargs["walks"] = kwargs["walks"]
        sample_meth = dy_utils.EnsembleWalkSampler(**internal_kwargs)
        _ = f"an average of {sample_meth.naccept} accepted steps up to chain length {sample_meth.maxmcmc}"
    elif kwargs["sample"] == "rwalk":
        internal_kwargs["nact"] = nact
        sample_meth = dy_utils.AcceptanceTrackingRWalk(**internal_kwargs)
        _ = f"An average of {2 * sample_meth.nact} steps will be accepted up to chain length {sample_meth.maxmcmc}"
    kwargs["sample"], kwargs["bound"] = sample_meth, "none"
    return kwargs


def _fake_nested_iteration(sampler_obj, pool, live_u, live_logl, pt, n_evolve, seed0):
    """What dynesty does per batch: worst points -> prepare_sampler -> mapper(sample, records) -> tune."""
    loglstar = float(np.quantile(live_logl, 0.2))
    ns = types.SimpleNamespace(live_u=live_u, nlive=len(live_u))
    args = sampler_obj.prepare_sampler(loglstar=loglstar, points=live_u[:n_evolve], axes=None,
                                       seeds=[seed0 + i for i in range(n_evolve)], prior_transform=pt,
                                       loglikelihood=pool.log_likelihood, nested_sampler=ns)
    res = pool.map(sampler_obj.sample, args)
    for r in res:
        sampler_obj.tune(r.tuning_info, update=True)
    return loglstar, args, res


@pytest.mark.parametrize("sample", ["acceptance-walk", "rwalk", "act-walk"])
def test_reference_walker_objects_are_replaced_by_name(sample):
    ndim, nlive = 4, 600
    keys = [f"p{i}" for i in range(ndim)]
    priors = {k: _Uniform(-1.0, 1.0, boundary=b) for k, b in zip(keys, ("periodic", "reflective", None, None))}
    pt = BatchedPriorTransform(priors, keys)
    lik = _FakeLikelihood()
    pool = GPUPool(lik, queue_size=nlive, prior_transform_many=pt)
    kw = _reference_sampler_object(sample, priors, keys)
    walker = kw["sample"]
    assert kw["bound"] == "none" and list(walker.periodic) == [0] and list(walker.reflective) == [1] and walker.maxmcmc == 200
    rng = np.random.default_rng(3)
    live_u = rng.random((nlive, ndim))
    live_logl = lik.log_likelihood_batch(pt(live_u))
    lik.batches = lik.evals = lik.largest = 0
    loglstar, args, res = _fake_nested_iteration(walker, pool, live_u, live_logl, pt, n_evolve=nlive, seed0=1000)
    assert len(res) == nlive and lik.largest >= nlive // 2            # the whole queue in one launch per MCMC step
    assert lik.batches <= walker.thin * walker.maxmcmc + 2 if sample == "act-walk" else lik.batches <= 2 * walker.maxmcmc
    for a, r in zip(args, res):
        u, v, logl, ncall, blob = r                                   # dynesty 2 unpacking ...
        assert r.u is u and r.ncalls == ncall and r.tuning_info is blob      # ... and dynesty 3 field names
        assert np.all((u >= 0) & (u <= 1)) and ncall >= 1
        assert np.array_equal(v, pt(u)) and logl == pool.log_likelihood(v)
        assert logl > loglstar or blob["accept"] == 0                 # (a chain that never accepts returns a prior draw)
    acc = np.array([r[4]["accept"] for r in res])
    assert (acc > 0).mean() > 0.9
    # a chain driven alone through the per-point protocol is the same chain, bit for bit
    for i in (0, 7, 311):
        solo = walker.sample(args[i]) if sample != "rwalk" else None
        if solo is not None:
            assert np.array_equal(solo[0], res[i][0]) and solo[2] == res[i][2] and solo[3] == res[i][3] and solo[4] == res[i][4]


def test_naccept_adaptation_steers_the_walk_length():
    """EnsembleWalkSampler.tune: the walk length settles where a chain accepts ``naccept`` steps on average."""
    ndim, nlive = 3, 400
    keys = [f"p{i}" for i in range(ndim)]
    priors = {k: _Uniform(-1.0, 1.0) for k in keys}
    pt = BatchedPriorTransform(priors, keys)
    lik = _FakeLikelihood()
    pool = GPUPool(lik, queue_size=nlive, prior_transform_many=pt)
    walker = _reference_sampler_object("acceptance-walk", priors, keys, naccept=12, walks=5, maxmcmc=500)["sample"]
    rng = np.random.default_rng(5)
    live_u = rng.random((nlive, ndim))
    live_logl = lik.log_likelihood_batch(pt(live_u))
    history = []
    for it in range(6):
        _, _, res = _fake_nested_iteration(walker, pool, live_u, live_logl, pt, n_evolve=nlive, seed0=10_000 * it)
        history.append((walker.walks, np.mean([r[4]["accept"] for r in res])))
    assert history[0][1] < 6                       # 5 steps cannot accept 12 times
    assert 9 <= history[-1][1] <= 16, history      # settled around naccept = 12
    assert 2 <= walker.walks <= walker.maxmcmc and walker.sampler_kwargs["walks"] == walker.walks


def test_chains_are_bitwise_reproducible_in_any_queue():
    """Counter-based randomness keyed by (chain seed, step): a chain's result does not depend on the queue it runs in."""
    ndim, n = 4, 300
    priors = {f"p{i}": _Uniform(-1.0, 1.0) for i in range(ndim)}
    pt = BatchedPriorTransform(priors, list(priors))
    lik = _FakeLikelihood()
    pool = GPUPool(lik, queue_size=n, prior_transform_many=pt)
    rng = np.random.default_rng(3)
    live = rng.random((n, ndim))
    logl = lik.log_likelihood_batch(pt(live))
    queue = [Args(u=live[i], loglstar=float(np.quantile(logl, 0.2)), rseed=1000 + i, prior_transform=pt,
                  loglikelihood=pool.log_likelihood, kwargs={"live": live}) for i in range(n)]
    walker = dy_utils.EnsembleWalkSampler(ndim=ndim, walks=15, maxmcmc=100, periodic=[0], reflective=[1])
    full = pool.map(walker, queue)
    perm = rng.permutation(n)
    shuffled = pool.map(walker, [queue[i] for i in perm])
    part = pool.map(walker.sample, queue[40:57])
    for q, i in enumerate(perm):
        assert np.array_equal(shuffled[q][0], full[i][0]) and shuffled[q][2:] == full[i][2:]
    for q, i in enumerate(range(40, 57)):
        assert np.array_equal(part[q][0], full[i][0]) and part[q][2:] == full[i][2:]
    one = walker(queue[5])
    assert np.array_equal(one[0], full[5][0]) and one[2:] == full[5][2:]
    # the draws themselves: a pure function of (seed, step, k), uniform on (0, 1)
    r = counter_uniforms(np.arange(20000), np.full(20000, 3))
    assert np.array_equal(r[:10], counter_uniforms(np.arange(10), np.full(10, 3)))
    assert r.min() > 0 and r.max() < 1 and stats.kstest(r.ravel(), "uniform").pvalue > 1e-3
    assert abs(np.corrcoef(r[:, 0], r[:, 1])[0, 1]) < 0.03


def _independent_chain(u, live, loglstar, walks, rng, loglike, periodic, reflective):
    """The same rules written the plain way -- one chain, numpy's own generator, a Python loop."""
    ndim = len(u)
    accept = ncall = 0
    logl = None
    for _ in range(walks):
        i, j = rng.choice(len(live), 2, replace=False)
        gamma = 1.0 if rng.random() < 0.5 else 2.38 / np.sqrt(2 * ndim) * rng.gamma(4, 0.25)
        prop = u + gamma * (live[j] - live[i])
        prop[periodic] = np.mod(prop[periodic], 1.0)
        q = np.mod(prop[reflective], 2.0)
        prop[reflective] = np.where(q > 1, 2 - q, q)
        if prop.min() < 0 or prop.max() > 1:
            continue
        ncall += 1
        lp = loglike(prop)
        if lp > loglstar:
            u, logl, accept = prop, lp, accept + 1
    return logl, ncall, accept


def test_lockstep_walk_samples_the_same_distribution_as_plain_chains():
    """10 000 chains: accepted-point log-likelihoods, evaluation counts and acceptance counts of the lock-step walker against an
    independent plain implementation of the same rules (two-sample Kolmogorov-Smirnov)."""
    ndim, n_live, n_chain, walks = 3, 500, 10_000, 12
    rng = np.random.default_rng(11)
    live = rng.random((n_live, ndim))

    def loglike(u):
        return float(-0.5 * np.sum(((2 * u - 1 - 0.3) / 0.2) ** 2))

    live_logl = np.array([loglike(x) for x in live])
    loglstar = float(np.quantile(live_logl, 0.5))
    starts = live[rng.integers(0, n_live, n_chain)]
    walker = dy_utils.EnsembleWalkSampler(ndim=ndim, walks=walks, maxmcmc=100, periodic=[0], reflective=[1])
    queue = [Args(u=starts[i], loglstar=loglstar, rseed=77_000 + i, prior_transform=lambda u: u, loglikelihood=loglike,
                  kwargs={"live": live}) for i in range(n_chain)]
    res = walker.run_many(queue, lambda th: -0.5 * np.sum(((2 * th - 1 - 0.3) / 0.2) ** 2, axis=1), lambda u: u)
    got_acc = np.array([r[4]["accept"] for r in res])
    got_logl = np.array([r[2] for r in res])[got_acc > 0]
    got_ncall = np.array([r[3] for r in res])[got_acc > 0]
    ref = [_independent_chain(starts[i].copy(), live, loglstar, walks, rng, loglike, [0], [1]) for i in range(n_chain)]
    ref_acc = np.array([r[2] for r in ref])
    ref_logl = np.array([r[0] for r in ref if r[2] > 0])
    ref_ncall = np.array([r[1] for r in ref if r[2] > 0])
    assert stats.ks_2samp(got_logl, ref_logl).pvalue > 1e-3
    assert stats.ks_2samp(got_ncall, ref_ncall).pvalue > 1e-3
    assert stats.ks_2samp(got_acc, ref_acc).pvalue > 1e-3
    assert abs(got_acc.mean() - ref_acc.mean()) < 0.05 * ref_acc.mean()


def test_round2_walker_name_still_works():
    ndim, n = 4, 200
    priors = {f"p{i}": _Uniform(-1.0, 1.0) for i in range(ndim)}
    pt = BatchedPriorTransform(priors, list(priors))
    lik = _FakeLikelihood()
    pool = GPUPool(lik, queue_size=n, prior_transform_many=pt)
    rng = np.random.default_rng(3)
    live = rng.random((n, ndim))
    logl = lik.log_likelihood_batch(pt(live))
    queue = [Args(u=live[i], loglstar=float(np.quantile(logl, 0.2)), rseed=1000 + i, prior_transform=pt,
                  loglikelihood=pool.log_likelihood, kwargs={"live_u": live}) for i in range(n)]
    walker = LockstepEnsembleWalk(ndim, walks=20, maxmcmc=200, periodic=[0], reflective=[1])
    res = pool.map(walker, queue)
    ref = walker.run_many_chains(queue[:20], pool.log_likelihood_many, pt)
    for i in range(20):
        assert np.array_equal(res[i][0], ref[i][0]) and res[i][2:] == ref[i][2:]
    assert all(r[4]["accept"] + r[4]["reject"] >= walker.walks for r in res)


def test_batched_prior_transform_matches_per_point():
    priors = {"a": _Uniform(0.0, 2.0), "b": _Uniform(-5.0, 5.0)}
    pt = BatchedPriorTransform(priors, ["a", "b"])
    u = np.random.default_rng(0).random((64, 2))
    many = pt(u)
    for i in range(64):
        assert np.array_equal(pt(u[i]), many[i])
    assert many[:, 0].min() >= 0 and many[:, 1].min() >= -5


def test_pool_maps_its_own_likelihood_as_one_batch():
    lik = _FakeLikelihood()
    pool = GPUPool(lik, queue_size=2048)
    thetas = list(np.random.default_rng(1).random((2048, 3)))
    out = pool.map(pool.log_likelihood, thetas)
    assert lik.batches == 1 and len(out) == 2048
    assert pool.map(len, [[1, 2], [3]]) == [2, 1]      # any other function: plain map
