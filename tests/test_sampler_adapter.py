"""The sampler seam on the CPU: a fake nested-sampling driver calls ``pool.map`` exactly as dynesty does
(``mapper(sample_object, queue_of_argument_records)``, nmma/core/mpi_setup.py:282-285) and the lock-step walker
turns the queue into one likelihood batch per MCMC step."""
import collections

import numpy as np

from nmma_amd.pool import GPUPool
from nmma_amd.sampler import BatchedPriorTransform, LockstepEnsembleWalk

Args = collections.namedtuple("Args", "u loglstar rseed prior_transform loglikelihood kwargs")


class _Uniform:
    def __init__(self, lo, hi):
        self.minimum, self.maximum = lo, hi

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


def _queue(n, ndim, rng, pt, ll):
    live = rng.random((n, ndim))
    logl = np.array([ll(pt(u)) for u in live])
    loglstar = np.quantile(logl, 0.2)
    return [Args(u=live[i], loglstar=loglstar, rseed=1000 + i, prior_transform=pt, loglikelihood=ll,
                 kwargs={"live_u": live}) for i in range(n)]


def test_lockstep_queue_is_one_launch_per_step_and_matches_per_point_chains():
    ndim, n = 4, 1500
    priors = {f"p{i}": _Uniform(-1.0, 1.0) for i in range(ndim)}
    pt = BatchedPriorTransform(priors, list(priors))
    lik = _FakeLikelihood()
    pool = GPUPool(lik, queue_size=n, prior_transform_many=pt)
    walker = LockstepEnsembleWalk(ndim, walks=20, maxmcmc=200, periodic=[0], reflective=[1])
    queue = _queue(n, ndim, np.random.default_rng(3), pt, pool.log_likelihood)
    # (a) coroutine lock-step: every chain keeps its own random stream -> identical to driving the chains one at a time
    lik.batches = lik.evals = lik.largest = 0
    res = walker.run_many_chains(queue, pool.log_likelihood_many, pt)
    assert len(res) == n
    assert lik.largest >= 1000                         # whole queue in one launch
    assert lik.evals / lik.batches >= 500              # stragglers (chains without an acceptance yet) thin the last batches
    assert lik.batches <= 201
    for i in (0, 7, 311, n - 1):
        u, v, logl, ncall, blob = walker(queue[i])
        ru, rv, rl, rn, rb = res[i]
        assert np.array_equal(u, ru) and np.array_equal(v, rv) and logl == rl and ncall == rn and blob == rb
        assert rl > queue[i].loglstar or rb["accept"] == 0
        assert np.all((ru >= 0) & (ru <= 1))
    # (b) the array form the pool uses: same rules, one random stream, no Python work per chain
    lik.batches = lik.evals = lik.largest = 0
    res = pool.map(walker, queue)                      # <- the unmodified call pattern of the sampler
    assert len(res) == n and lik.largest >= 500 and lik.batches <= 201     # (out-of-cube proposals are rejected without an evaluation)
    acc = np.array([r[4]["accept"] for r in res])
    ref_acc = np.array([r[4]["accept"] for r in walker.run_many_chains(queue, pool.log_likelihood_many, pt)])
    assert abs(acc.mean() - ref_acc.mean()) < 0.15 * ref_acc.mean()          # same acceptance statistics
    for (u, v, logl, ncall, blob), a in zip(res, queue):
        assert np.all((u >= 0) & (u <= 1)) and ncall >= 1
        assert logl > a.loglstar or blob["accept"] == 0
        assert np.array_equal(v, pt(u)) and logl == pool.log_likelihood(v)
        assert blob["accept"] + blob["reject"] >= walker.walks


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
