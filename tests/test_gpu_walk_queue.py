"""The sampler seam end to end (nmma/core/mpi_setup.py:202-245, :282-303, :339): a whole queue of the nested
sampler as ONE library call (``nmma_em_walk_queue``), its array-backed argument / result sequences, Constraint priors on the
device, and the priors the device transform knows."""
import ctypes as C

import numpy as np
import pytest

from nmma_amd import sampler as smp
from nmma_amd import synthetic as syn
from nmma_amd.core.base import Constraint
from tests.helpers import SimplePrior, UniformPrior, plugin_from_case

pytestmark = pytest.mark.gpu
FLOOR = -1.7976931348623157e308


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def config2(torch_cuda):
    case = syn.config2_case()
    _, _, lik = plugin_from_case(case)
    names = case["names"]
    th = syn.draw_theta(3, 20000, names)[1]
    pri = {k: UniformPrior(float(a), float(b)) for k, a, b in zip(names, th.min(axis=0), th.max(axis=0))}
    return case, lik, names, pri


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_queue_call_is_bit_identical_to_the_step_loop(torch_cuda, config2):
    """``EMEngine.walk_queue`` (upload, step loop, fresh draws, download inside the library) against ``device_walk`` +
    ``_device_fresh_draws`` (the same kernels driven from Python): the same bits in u, v, logL and all four counters -- equal and
    per-chain walk lengths, chains that never move (bound = +inf: they come back as fresh prior draws)."""
    torch = torch_cuda
    case, lik, names, pri = config2
    eng = lik.sub_model.engine(names)
    pt = smp.BatchedPriorTransform(pri, names)
    rng = np.random.default_rng(31)
    n, n_live = 1500, 900
    live = rng.uniform(0.3, 0.7, (n_live, len(names)))
    u0 = live[rng.integers(0, n_live, n)].copy()
    l_live = eng.loglike(np.ascontiguousarray(pt(live)))
    bound = np.full(n, np.quantile(l_live, 0.3))
    bound[::50] = np.inf                                   # these chains can never accept
    keys = rng.integers(1, 2 ** 62, n).astype(np.uint64)
    table = smp.device_prior_table(pri, names)
    w = smp.EnsembleWalkSampler(ndim=len(names), walks=9)
    buf = torch.empty(n, dtype=torch.float64, device="cuda:0")
    dev_ll = lambda t: eng.loglike(t, out=buf)
    for steps in (9, (3 + np.arange(n) % 8).astype(np.int32)):
        u, v, logl, counts = smp.device_walk(table, live, u0, bound, keys, steps, dev_ll)
        w._device_fresh_draws(np.nonzero(counts[:, 0] == 0)[0], table, keys, u, v, logl, counts, lambda t: eng.loglike(t), 0)
        qu, qv, ql, qc = eng.walk_queue(table, live, u0, bound, keys, steps)
        assert _same(qu, u) and _same(qv, v) and _same(ql, logl) and np.array_equal(qc, counts)
        assert eng.last_walk_gpu_ms > 0
        stuck = counts[:, 0] == 0
        assert stuck[::50].all() and np.all(counts[stuck, 3] >= 1) and np.all(np.isfinite(ql))
        moved = ~stuck
        assert np.all(ql[moved] > bound[moved]) and _same(eng.loglike(np.ascontiguousarray(qv)), ql)


def _per_filter_syserr_case():
    """One sampled systematic PER observed filter (a YAML systematics file without time nodes): 6 + 9 = 15 sampled dimensions on the
    lean task with sampled systematics (FASTM 3)."""
    from nmma_amd import synthetic as syn
    from tests import cases
    filters = syn.AT2017GFO_FILTERS
    sys_names = [f"em_syserr_{k}" for k in range(len(filters))]
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind"] + sys_names
    c = cases._base(seed=5234, names=names, batch=48)
    c["systematics"] = dict(mode="mixed", names=dict(zip(filters, sys_names)), nodes={})
    return c


@pytest.mark.parametrize("name", ["c2_default", "syserr_param", "fast_np6", "log_grid", "c2_dt05_limit", "averaging", "c4_shape", "c4_syserr", "syserr_per_filter",
                                  "extinction_p92"])        # (Pei-1992 extinction: in the step since the law is a series in the kernel, no pre-pass launch)
def test_fused_mcmc_step_is_the_two_launch_step(torch_cuda, name):
    """The queue's one-launch MCMC step (accept + next proposal in the likelihood kernel's epilogue: ``nmma_em_loglike_walk``,
    ``em_logl<..., WALKF>``) against the likelihood launch + ``walk_step_kernel`` (option ``walk_fuse`` = 0) and against the
    Python-driven step loop: the same bits -- for the constant-systematics flavour (config 2), the sampled ``em_syserr`` flavour
    (7 sampled dimensions) and a six-input surrogate (KP = 2) -- with periodic / reflective dimensions and ragged queue sizes."""
    import os
    from tests import cases
    from tests.helpers import engine_from_case
    torch = torch_cuda
    case = _per_filter_syserr_case() if name == "syserr_per_filter" else (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
    eng = engine_from_case(case)
    names = case["names"]
    assert len(names) <= 16        # (more than 8 sampled dimensions: 16 lanes per chain in the fused step -- c4_syserr, syserr_per_filter)
    th = case["theta"]
    lo, hi = th.min(axis=0) - 1e-3, th.max(axis=0) + 1e-3
    pri = {k: UniformPrior(float(a), float(b)) for k, a, b in zip(names, lo, hi)}
    pt = smp.BatchedPriorTransform(pri, names)
    w = smp.EnsembleWalkSampler(ndim=len(names), periodic=[1], reflective=[2], walks=7)
    table = smp.device_prior_table(pri, names, w.periodic, w.reflective)
    rng = np.random.default_rng(71)
    for n in (1, 37, 1000, 4096, 4500):          # (beyond 4096 chains: the step on 32-sample tiles)
        n_live = 300
        live = rng.uniform(0.2, 0.8, (n_live, len(names)))
        u0 = live[rng.integers(0, n_live, n)].copy()
        bound = np.full(n, np.quantile(eng.loglike(np.ascontiguousarray(pt(live))), 0.3))
        keys = rng.integers(1, 2 ** 62, n).astype(np.uint64)
        steps = 7 if n != 37 else (2 + np.arange(n) % 6).astype(np.int32)
        fused = eng.walk_queue(table, live, u0, bound, keys, steps)      # (small queues: the launch is split by band AND carries the step)
        eng.set_option("walk_split", 0)          # every workgroup walks all bands and steps its own tile
        try:
            unsplit = eng.walk_queue(table, live, u0, bound, keys, steps)
        finally:
            eng.set_option("walk_split", 1)
        for a, b in zip(fused, unsplit):
            assert _same(a, b), (name, n, "split / unsplit")
        eng.set_option("walk_fuse", 0)
        try:
            two = eng.walk_queue(table, live, u0, bound, keys, steps)
        finally:
            eng.set_option("walk_fuse", 1)
        for a, b in zip(fused, two):
            assert _same(a, b), (name, n)
        buf = torch.empty(n, dtype=torch.float64, device="cuda:0")
        u, v, logl, counts = smp.device_walk(table, live, u0, bound, keys, steps, lambda t: eng.loglike(t, out=buf))
        w._device_fresh_draws(np.nonzero(counts[:, 0] == 0)[0], table, keys, u, v, logl, counts, lambda t: eng.loglike(t), 0)
        assert _same(fused[0], u) and _same(fused[2], logl) and np.array_equal(fused[3], counts)
    eng.close()


@pytest.mark.parametrize("name", ["c2_default", "syserr_param", "log_grid", "syserr_per_filter"])
def test_fused_mcmc_step_with_constraints_and_long_queues(torch_cuda, name):
    """The one-launch MCMC step with the chains' Constraint program evaluated in it (``em_logl<..., WALKF | 64>``: the interpreter's
    stack in LDS) and for queues of more than 4096 chains (the step in the 32-sample-tile kernels, four / eight rounds of chains per tile):
    the same bits as the likelihood launch + ``walk_step_kernel`` -- 6, 7 and 15 sampled dimensions (8 / 16 lanes per chain)."""
    from nmma_amd.core.constraints import ConstraintProgram
    from nmma_amd import _lib as L
    from tests import cases
    from tests.helpers import engine_from_case
    case = _per_filter_syserr_case() if name == "syserr_per_filter" else (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
    eng = engine_from_case(case)
    names = case["names"]
    th = case["theta"]
    lo, hi = th.min(axis=0) - 1e-3, th.max(axis=0) + 1e-3
    pri = {k: UniformPrior(float(a), float(b)) for k, a, b in zip(names, lo, hi)}
    pt = smp.BatchedPriorTransform(pri, names)
    w = smp.EnsembleWalkSampler(ndim=len(names), periodic=[1], reflective=[2], walks=6)
    table = smp.device_prior_table(pri, names, w.periodic, w.reflective)
    # a program with arithmetic, a transcendental and both kinds of check: 10 ** col4 + 10 ** col5 < bound, col3 > lower bound
    c4, c5 = names.index("log10_mej_dyn"), names.index("log10_mej_wind")
    mid = float(np.median(10 ** th[:, c4] + 10 ** th[:, c5]))
    ops = [(L.CON_PUSH_CONST, 0, 10.0), (L.CON_PUSH_COL, c4, 0.0), (L.CON_POW, 0, 0.0), (L.CON_PUSH_CONST, 0, 10.0), (L.CON_PUSH_COL, c5, 0.0),
           (L.CON_POW, 0, 0.0), (L.CON_ADD, 0, 0.0), (L.CON_CHECK_LT, 0, 1.3 * mid),
           (L.CON_PUSH_COL, 3, 0.0), (L.CON_CHECK_GT, 0, float(np.quantile(th[:, 3], 0.05))), (L.CON_CHECK_LT, 0, 1e300)]
    prog = ConstraintProgram(ops, len(names), 0)
    rng = np.random.default_rng(72)
    for n, con in ((37, prog), (4096, prog), (5000, None), (8192 + 77, prog)):
        n_live = 300
        live = rng.uniform(0.2, 0.8, (n_live, len(names)))
        u0 = live[rng.integers(0, n_live, n)].copy()
        bound = np.full(n, np.quantile(eng.loglike(np.ascontiguousarray(pt(live))), 0.3))
        keys = rng.integers(1, 2 ** 62, n).astype(np.uint64)
        steps = 6 if n != 37 else (2 + np.arange(n) % 5).astype(np.int32)
        fused = eng.walk_queue(table, live, u0, bound, keys, steps, constraints=con)
        eng.set_option("walk_fuse", 0)
        try:
            two = eng.walk_queue(table, live, u0, bound, keys, steps, constraints=con)
        finally:
            eng.set_option("walk_fuse", 1)
        for a, b in zip(fused, two):
            assert _same(a, b), (name, n)
        if con is not None:          # the constraint bites: some proposals were floored, i.e. evaluated and rejected
            free = eng.walk_queue(table, live, u0, bound, keys, steps)
            assert not _same(free[0], fused[0])
    prog.close()
    eng.close()


def test_pool_map_on_the_argument_batch_returns_array_backed_records(torch_cuda, config2):
    """``prepare_sampler`` hands the queue over as ONE set of arrays and ``GPUPool.map(walker.sample, queue)`` answers with an
    array-backed sequence: the records (u, v, logl, ncall, blob) are those of the per-record list path, bit for bit."""
    from nmma_amd.pool import GPUPool
    case, lik, names, pri = config2
    pt = smp.BatchedPriorTransform(pri, names)
    rng = np.random.default_rng(32)
    n, walks = 700, 11
    live = rng.uniform(0.3, 0.7, (n, len(names)))
    w = smp.EnsembleWalkSampler(ndim=len(names), walks=walks, naccept=5)

    class _NS:
        live_u = live
    seeds = rng.integers(1, 2 ** 62, n)
    bound = float(np.quantile(lik.log_likelihood_batch(pt(live), names), 0.25))
    batch = w.prepare_sampler(loglstar=bound, points=live.copy(), axes=None, seeds=seeds, prior_transform=pt, loglikelihood=None, nested_sampler=_NS)
    assert isinstance(batch, smp.SamplerArgumentBatch) and len(batch) == n and batch[3].rseed == seeds[3] and batch[-1].loglstar == bound
    pool = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt, priors=pri)
    got = pool.map(w.sample, batch)
    assert isinstance(got, smp.WalkResults) and len(got) == n
    ref = pool.map(w.sample, list(batch))                 # per-record arguments, as dynesty 2 builds them
    recs, refs = list(got), list(ref)
    assert all(len(r) == 5 for r in recs)
    for k in (0, 1):
        assert _same(np.stack([r[k] for r in recs]), np.stack([r[k] for r in refs]))
    assert [r[2] for r in recs] == [r[2] for r in refs] and [r[3] for r in recs] == [r[3] for r in refs]
    assert [r[4] for r in recs] == [r[4] for r in refs]
    assert _same(got[5][0], recs[5][0]) and _same(got[5][1], recs[5][1]) and got[5][2:] == recs[5][2:] and got[-1][2:] == recs[-1][2:]
    assert got[5].tuning_info["walks"] == walks and got[5].ncalls == recs[5][3]
    w.tune(got[0].tuning_info)
    # the library's queue against the Python-driven step loop through the plugin's batched call
    slow = w.run_many_device(batch, lambda t: lik.log_likelihood_batch(t, names), pri, names)
    assert _same(slow.u, got.u) and _same(slow.logl, got.logl) and np.array_equal(slow.accept, got.accept)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0, 0, 0, 0, 0]])
def test_queue_sharded_over_engines_is_the_single_device_queue(torch_cuda, config2, devices):
    """``GPUPool(devices=[...])`` shards a queue over one engine per entry -- contiguous balanced shards, the live set replicated,
    every shard's ``nmma_em_walk_queue_begin`` issued before the first ``..._end`` (core/mpi_setup.py:651-667, :679-683: the
    reference spreads a queue's chains over its ranks).  Tested with several engines on the ONE device a test box has: ragged
    splits, equal and per-chain walk lengths, a constrained prior set -- records identical to the single-device queue, bit for
    bit, in queue order; ``log_likelihood`` batches take the same shards."""
    from nmma_amd.pool import GPUPool
    case, lik, names, pri = config2
    pt = smp.BatchedPriorTransform(pri, names)
    rng = np.random.default_rng(40 + len(devices))
    for n, walks in ((1, 5), (7, 6), (701, 9), (4096, 4)):
        live = rng.uniform(0.3, 0.7, (max(n, 50), len(names)))
        w = smp.EnsembleWalkSampler(ndim=len(names), walks=walks, naccept=5)

        class _NS:
            live_u = live
        seeds = rng.integers(1, 2 ** 62, n)
        bound = float(np.quantile(lik.log_likelihood_batch(pt(live), names), 0.25))
        batch = w.prepare_sampler(loglstar=bound, points=live[:n].copy(), axes=None, seeds=seeds, prior_transform=pt, loglikelihood=None,
                                  nested_sampler=_NS)
        one = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt, priors=pri)
        many = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt, priors=pri, devices=devices)
        ref, got = one.map(w.sample, batch), many.map(w.sample, batch)
        assert len(many._shard_engines) == len(devices) and many._shard_engines[0][0] is lik.sub_model.engine(names)
        assert _same(got.u, ref.u) and _same(got.v, ref.v) and _same(got.logl, ref.logl)
        assert np.array_equal(got.ncall, ref.ncall) and np.array_equal(got.accept, ref.accept)
        theta = pt(rng.uniform(0.2, 0.8, (n, len(names))))
        assert np.array_equal(many.log_likelihood_many(theta), one.log_likelihood_many(theta))
        many.close()
    # per-chain walk lengths through the walker's own entry point, a list of (engine, program) shards
    n = 333
    eng = lik.sub_model.engine(names)
    extra = [lik.sub_model._build_engine(names, lik.sub_model.light_curve_model.gpu_filters, lik.sub_model.light_curve_model.engine_kwargs())
             for _ in devices[1:]]
    table = smp.device_prior_table(pri, names)
    live = rng.uniform(0.3, 0.7, (400, len(names)))
    u0 = live[rng.integers(0, 400, n)].copy()
    bound = np.full(n, np.quantile(eng.loglike(np.ascontiguousarray(pt(live))), 0.3))
    keys = rng.integers(1, 2 ** 62, n).astype(np.uint64)
    steps = (2 + np.arange(n) % 7).astype(np.int32)
    ref = eng.walk_queue(table, live, u0, bound, keys, steps)
    from nmma_amd.parallel import shard_bounds
    shards = [eng] + extra
    toks = []
    for r, e in enumerate(shards):
        lo, hi = shard_bounds(n, len(shards), r)
        toks.append(e.walk_queue_begin(table, live, u0[lo:hi], bound[lo:hi], keys[lo:hi], np.ascontiguousarray(steps[lo:hi])))
    parts = [e.walk_queue_end(t) for e, t in zip(shards, toks)]
    for i in range(4):
        assert _same(np.concatenate([p[i] for p in parts]), ref[i])
    # one queue in flight per engine: a second begin without end is refused
    tok = eng.walk_queue_begin(table, live, u0, bound, keys, 3)
    with pytest.raises(Exception, match="not collected"):
        eng.walk_queue_begin(table, live, u0, bound, keys, 3)
    eng.walk_queue_end(tok)
    for e in extra:
        e.close()


def test_constrained_queue_sharded_over_engines(torch_cuda, config2):
    """A constrained prior set: every shard gets the constraint program of its device; rows that violate it count as evaluated."""
    from nmma_amd.pool import GPUPool
    case, _, names, pri = config2
    _, _, lik = plugin_from_case(case)
    lik.constraints["log10_mej_dyn"] = Constraint(minimum=-2.6, maximum=-1.3, name="log10_mej_dyn")
    pt = smp.BatchedPriorTransform(pri, names)
    rng = np.random.default_rng(77)
    n, walks = 500, 8
    live = rng.uniform(0.3, 0.7, (n, len(names)))
    w = smp.EnsembleWalkSampler(ndim=len(names), walks=walks, naccept=5)

    class _NS:
        live_u = live
    seeds = rng.integers(1, 2 ** 62, n)
    bound = float(np.quantile(lik.log_likelihood_batch(pt(live), names), 0.25))
    batch = w.prepare_sampler(loglstar=bound, points=live.copy(), axes=None, seeds=seeds, prior_transform=pt, loglikelihood=None, nested_sampler=_NS)
    one = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt, priors=pri)
    many = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt, priors=pri, devices=[0, 0, 0])
    ref, got = one.map(w.sample, batch), many.map(w.sample, batch)
    assert all(prog is not None for _, prog in many._shard_engines)
    assert _same(got.u, ref.u) and _same(got.logl, ref.logl) and np.array_equal(got.ncall, ref.ncall)
    many.close()


def test_new_prior_transforms_match_bilbys_formulas(torch_cuda):
    """TruncatedGaussian (priors/Sr2023.prior), LogNormal, HalfGaussian: ``rescale`` as bilby/core/prior/analytical.py defines them."""
    torch = torch_cuda
    from scipy.special import erf, erfinv
    from nmma_amd import _lib as L

    class TruncatedNormal(SimplePrior):
        def __init__(self, mu, sigma, minimum, maximum):
            super().__init__(minimum, maximum)
            self.mu, self.sigma = mu, sigma

        def rescale(self, val):
            norm = (erf((self.maximum - self.mu) / 2 ** 0.5 / self.sigma) - erf((self.minimum - self.mu) / 2 ** 0.5 / self.sigma)) / 2
            return erfinv(2 * val * norm + erf((self.minimum - self.mu) / 2 ** 0.5 / self.sigma)) * 2 ** 0.5 * self.sigma + self.mu

    class LogNormal(SimplePrior):
        def __init__(self, mu, sigma):
            super().__init__(0.0, np.inf)
            self.mu, self.sigma = mu, sigma

        def rescale(self, val):
            return np.exp(self.mu + np.sqrt(2 * self.sigma ** 2) * erfinv(2 * val - 1))

    class HalfGaussian(SimplePrior):
        def __init__(self, sigma):
            super().__init__(0.0, np.inf)
            self.sigma = sigma

        def rescale(self, val):
            return erfinv(val) * 2 ** 0.5 * self.sigma

    pri = {"a": TruncatedNormal(0.9, 0.3, 0.0, 10.0), "b": TruncatedNormal(0.0, 2.0, -1.0, 0.5), "c": LogNormal(0.2, 0.7), "d": HalfGaussian(1.3)}
    keys = list(pri)
    table = smp.device_prior_table(pri, keys)
    assert table is not None
    u = np.random.default_rng(2).uniform(1e-6, 1 - 1e-6, (4000, len(keys)))
    ud = torch.as_tensor(u, device="cuda:0")
    out = torch.empty_like(ud)
    L.check(L.load_library().nmma_walk_rescale(table, len(keys), C.c_void_p(ud.data_ptr()), len(u), C.c_void_p(out.data_ptr()), 0, None), "rescale")
    want = np.stack([np.asarray(pri[k].rescale(u[:, i]), dtype=float) for i, k in enumerate(keys)], axis=1)
    got = out.cpu().numpy()
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 1e-11
    assert np.all((got[:, 0] > 0.0) & (got[:, 0] < 10.0)) and np.all((got[:, 1] > -1.0) & (got[:, 1] < 0.5))


def test_constraint_kernel_is_the_interpreter(torch_cuda):
    """``nmma_con_floor`` against ``evaluate_program`` (numpy) for the traced GW mass constraints and the EM conversions, NaN rows
    included; a bad program is refused by ``nmma_con_create``."""
    torch = torch_cuda
    from nmma_amd import _lib as L
    from nmma_amd.core import conversion as cv
    from nmma_amd.core.constraints import ConstraintProgram, evaluate_program, trace_constraints
    rng = np.random.default_rng(41)
    names = ["chirp_mass", "mass_ratio", "lambda_1", "theta_jn"]
    cons = {"mass_1": Constraint(1.001398, 1.9), "mass_2": Constraint(1.1, 4.31), "lambda_1": Constraint(0.0, np.inf),
            "symmetric_mass_ratio": Constraint(0.2, 0.2499), "theta_jn": Constraint(0.1, 3.0)}
    prog = trace_constraints(cons, names, {}, [cv.bns_source_frame])
    assert prog is not None
    theta = np.column_stack([rng.uniform(1.1, 1.3, 5000), rng.uniform(0.3, 1.0, 5000), rng.uniform(-100, 3000, 5000), rng.uniform(0, np.pi, 5000)])
    theta[7, 1] = np.nan
    theta[11, 0] = np.inf
    want = evaluate_program(prog, theta)
    p = ConstraintProgram(prog, len(names))
    th = torch.as_tensor(theta, device="cuda:0")
    logl = torch.arange(len(theta), dtype=torch.float64, device="cuda:0")
    p.floor(th, logl)
    got = logl.cpu().numpy()
    assert np.array_equal(got == FLOOR, ~want) and np.array_equal(got[want], np.arange(len(theta), dtype=float)[want])
    assert 0.05 < want.mean() < 0.95 and not want[7] and not want[11]
    # a wider theta (ld > columns of the program) is fine; a narrower one is refused
    wide = torch.as_tensor(np.column_stack([theta, theta[:, :2]]), device="cuda:0")
    l2 = torch.zeros(len(theta), dtype=torch.float64, device="cuda:0")
    p.floor(wide, l2)
    assert np.array_equal(l2.cpu().numpy() == FLOOR, ~want)
    with pytest.raises(L.NMMAHipError):
        p.floor(th[:, :3].contiguous(), l2)
    p.close()
    for bad in ([(L.CON_ADD, -1, 0.0)], [(L.CON_PUSH_COL, 9, 0.0), (L.CON_CHECK_GT, -1, 0.0), (L.CON_CHECK_LT, -1, 1.0)],
                [(L.CON_PUSH_COL, 0, 0.0)], [(99, 0, 0.0)]):
        with pytest.raises(L.NMMAHipError):
            ConstraintProgram(bad, 4)


def test_constraints_of_a_cuda_batch_never_touch_the_host(torch_cuda, config2):
    """core/base.py:51-82 on the batched path: with a CUDA ``theta`` the Constraint priors are evaluated by a kernel -- no ``.cpu()``
    of theta, no blocking copy (torch's sync debug mode raises on any) -- and floor exactly the rows the per-sample
    ``log_likelihood`` floors."""
    torch = torch_cuda
    case, _, names, _ = config2
    _, _, lik = plugin_from_case(case)
    pri = dict(lik.priors)
    pri["KNtheta"] = Constraint(minimum=10.0, maximum=60.0, name="KNtheta")          # derived by the model's conversion
    lik.priors = pri
    lik.constraints["log10_mej_dyn"] = Constraint(minimum=-2.8, maximum=-1.2, name="log10_mej_dyn")     # (edited in place, as callers do)
    theta = case["theta"]
    th = torch.as_tensor(theta, device="cuda:0")
    warm = lik.log_likelihood_batch(th, names)               # builds the engine and lowers the constraint set
    assert lik.device_constraints(names, {}, 0) is not None
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = lik.log_likelihood_batch(th, names)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    got = out.cpu().numpy()
    assert np.array_equal(got, warm.cpu().numpy())
    single = np.array([lik.log_likelihood(dict(zip(names, (float(v) for v in row)))) for row in theta])
    assert np.array_equal(got == FLOOR, single == FLOOR) and 0 < np.sum(got == FLOOR) < len(theta)
    assert np.array_equal(lik.log_likelihood_batch(theta, names) == FLOOR, single == FLOOR)       # host arrays: the numpy mask
    # an edit of the set is seen (the program is cached by content)
    lik.constraints["timeshift"] = Constraint(minimum=-1.0, maximum=0.0, name="timeshift")
    again = lik.log_likelihood_batch(th, names).cpu().numpy()
    ts = theta[:, names.index("timeshift")]
    assert np.array_equal(again == FLOOR, (got == FLOOR) | ~((ts > -1.0) & (ts < 0.0)))


def test_device_walk_honours_a_constrained_prior_set(torch_cuda, config2):
    """A queue over a likelihood WITH Constraint priors: the library's queue (constraint program fused into the accept step) and the
    Python-driven step loop around the plugin's batched call (which floors on the device) return the same chains; every accepted
    point satisfies the constraints."""
    from nmma_amd.pool import GPUPool
    case, _, names, pri = config2
    _, _, lik = plugin_from_case(case)
    p2 = dict(lik.priors)
    p2["KNtheta"] = Constraint(minimum=15.0, maximum=70.0, name="KNtheta")
    lik.priors = p2
    lik.constraints["log10_mej_wind"] = Constraint(minimum=-2.5, maximum=-0.8, name="log10_mej_wind")      # (a sampled column)
    assert "log10_mej_wind" in lik.constraints
    pt = smp.BatchedPriorTransform(pri, names)
    rng = np.random.default_rng(33)
    n, walks = 1200, 15
    live = rng.uniform(0.25, 0.75, (n, len(names)))
    w = smp.EnsembleWalkSampler(ndim=len(names), walks=walks, naccept=5)

    class _NS:
        live_u = live
    batch = w.prepare_sampler(loglstar=-1e5, points=live.copy(), axes=None, seeds=np.arange(5000, 5000 + n), prior_transform=pt,
                              loglikelihood=None, nested_sampler=_NS)
    pool = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt, priors=pri)
    eng, prog = pool._walk_engine()
    assert eng is not None and prog is not None
    got = pool.map(w.sample, batch)
    slow = w.run_many_device(batch, lambda t: lik.log_likelihood_batch(t, names), pri, names)
    assert _same(slow.u, got.u) and _same(slow.v, got.v) and _same(slow.logl, got.logl) and np.array_equal(slow.accept, got.accept)
    moved = got.accept > 0
    kn = got.v[:, names.index("inclination_EM")] * 180.0 / np.pi
    mw = got.v[:, names.index("log10_mej_wind")]
    ok = (kn > 15.0) & (kn < 70.0) & (mw > -2.5) & (mw < -0.8)
    assert moved.mean() > 0.8 and np.all(ok[moved])
    # a chain that never moved returns a fresh prior draw, which may violate the constraints: its logL is then the floor
    assert np.all(got.logl[~ok] == FLOOR)
    # the walk with the constraints makes fewer moves than the same walk without them
    _, _, free = plugin_from_case(case)
    got_free = GPUPool(free, queue_size=n, names=names, prior_transform_many=pt, priors=pri).map(w.sample, batch)
    assert got.accept.sum() < got_free.accept.sum()
