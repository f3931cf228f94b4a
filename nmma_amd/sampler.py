"""Sampler-side batching: drop-in replacements for the MCMC walk objects NMMA hands to dynesty, written so that a whole
queue of chains advances in lock-step -- ONE batched prior transform and ONE likelihood launch per MCMC step (SURVEY.md
section 8f-1).

What they replace.  ``nmma/core/mpi_setup.py:202-245`` builds the ``sample=`` object of the nested sampler from bilby's
dynesty utilities, by name and with these keyword arguments::

    internal_kwargs = dict(ndim=..., nonbounded=None, periodic=..., reflective=..., maxmcmc=maxmcmc)
    "acceptance-walk": dy_utils.EnsembleWalkSampler(**internal_kwargs, naccept=naccept, walks=walks)        # :222-224
    "act-walk":        dy_utils.ACTTrackingEnsembleWalk(**internal_kwargs, nact=nact)                        # :209-211
    "rwalk":           dy_utils.AcceptanceTrackingRWalk(**internal_kwargs, nact=nact)                        # :235-237

and reads ``.naccept``, ``.maxmcmc``, ``.thin``, ``.nact`` back for its log lines (:215-219, :227-231, :240-244).  dynesty then
calls ``prepare_sampler(...)`` to get one argument record per live point to evolve, maps ``sample`` over the records through
``pool.map`` (:282-285, :339) and feeds the returned tuning information to ``tune``.  The three classes below have the same
names, constructor keywords, attributes and methods, so ``import nmma_amd.sampler as dy_utils`` is the whole edit
(``tests/test_sampler_adapter.py`` drives exactly that with a fake nested sampler; dynesty / bilby are absent from the image, so
the protocol is restated from bilby's public ``dynesty_utils`` / ``dynesty3_utils`` -- stated, not pinned).

What is different inside.  Every chain is a *coroutine* that yields the unit-cube point it wants evaluated.  Called on one
record (``sample(args)``: the per-point protocol) a chain is driven alone, as in bilby.  Given to
:meth:`nmma_amd.pool.GPUPool.map` with the whole queue the chains run as rows of numpy arrays (:meth:`run_many`).  All
randomness is COUNTER-BASED: the numbers chain ``c`` draws at its step ``n`` are a hash of ``(seed_c, n, k)`` (SplitMix64),
so a chain sees the same random numbers -- and produces bit-identical results -- whether it runs alone, in a queue of 10 or in a
queue of 4096, in any order.

Moves and rules (bilby ``dynesty_utils``): differential-evolution proposal ``u' = u + gamma (a - b)`` with ``a, b`` two other live
points and ``gamma = 2.38 / sqrt(2 ndim) x Gamma(4, 1/4)`` or 1 with probability 1/2; periodic / reflective wrapping, proposals
outside the unit cube are rejected without an evaluation; accepted when ``logL > loglstar``; a chain that never accepts returns
a fresh prior draw.  ``EnsembleWalkSampler`` walks ``int(walks)`` steps and ``tune`` steers ``walks`` towards ``naccept`` accepted
steps per chain; ``AcceptanceTrackingRWalk`` runs until ``nact`` autocorrelation times, estimated from the acceptance ratio
(``estimate_nmcmc``), have passed; ``ACTTrackingEnsembleWalk`` estimates the autocorrelation time from the chain itself.
"""
from __future__ import annotations

from collections.abc import Sequence

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_K_STEP = np.uint64(0xD1342543DE82EF95)
_K_DRAW = np.uint64(0xA0761D6478BD642F)
N_DRAWS = 8         # uniforms a chain consumes per step


def _mix64(x):
    """SplitMix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    x = (x ^ (x >> np.uint64(30))) * _M1
    x = (x ^ (x >> np.uint64(27))) * _M2
    return x ^ (x >> np.uint64(31))


def counter_uniforms(seeds, steps, n_draws=N_DRAWS):
    """``[len(seeds), n_draws]`` uniforms in (0, 1): draw ``k`` of step ``steps[c]`` of the chain with key ``seeds[c]``.
    A pure function of (seed, step, k): no state, any evaluation order."""
    with np.errstate(over="ignore"):
        s = np.asarray(seeds, dtype=np.uint64)[:, None]
        n = np.asarray(steps, dtype=np.uint64)[:, None]
        k = np.arange(n_draws, dtype=np.uint64)[None, :]
        x = _mix64(_mix64(s * _GOLDEN + _GOLDEN) ^ (n * _K_STEP + k * _K_DRAW + _GOLDEN))
        x = _mix64(x)
    return ((x >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def chain_key(rseed):
    """64-bit key of a chain from dynesty's ``rseed`` (an integer, or a numpy Generator / SeedSequence of older versions)."""
    if isinstance(rseed, (int, np.integer)):
        return int(rseed) & 0xFFFFFFFFFFFFFFFF
    if isinstance(rseed, np.random.Generator):
        return int(rseed.integers(0, 2 ** 63 - 1))
    if isinstance(rseed, np.random.SeedSequence):
        return int(rseed.generate_state(1, dtype=np.uint64)[0])
    return int(np.random.default_rng(rseed).integers(0, 2 ** 63 - 1))


def chain_keys(rseeds):
    """``chain_key`` for a whole queue: one vectorised pass when the seeds are integers (dynesty 3 hands out an integer array)."""
    a = rseeds if isinstance(rseeds, np.ndarray) else None
    if a is None:
        try:
            a = np.asarray(rseeds)
        except (TypeError, ValueError):
            a = None
    if a is not None and a.dtype.kind in "iu" and a.ndim == 1:
        return a.astype(np.uint64, copy=False) if a.dtype.kind == "u" else a.astype(np.int64).view(np.uint64)
    return np.array([chain_key(r) for r in rseeds], dtype=np.uint64)


class BatchedPriorTransform:
    """``prior_transform`` for arrays of unit-cube points: ``u[B, D] -> theta[B, D]`` with one vectorised ``rescale`` per
    parameter (bilby priors' ``rescale`` are numpy ufunc-style; ``priors.rescale(keys, u)`` of ``core/mpi_setup.py:679-683`` is
    the per-point form, and a 1-D ``u`` is answered in that form)."""

    def __init__(self, priors, keys):
        self.keys = list(keys)
        self.priors = [priors[k] for k in self.keys]

    def __call__(self, u):
        u = np.asarray(u, dtype=float)
        single = u.ndim == 1
        u2 = np.atleast_2d(u)
        out = np.empty_like(u2)
        for i, pr in enumerate(self.priors):
            out[:, i] = pr.rescale(u2[:, i])
        return out[0] if single else out


class SamplerArgument:
    """The record dynesty hands to ``sample``: start point, likelihood bound, seed, the two callables and the walker's kwargs
    (``live`` = the unit-cube live points the differential-evolution move draws from)."""
    __slots__ = ("u", "loglstar", "axes", "scale", "rseed", "prior_transform", "loglikelihood", "kwargs")

    def __init__(self, u, loglstar, rseed, prior_transform, loglikelihood, kwargs, axes=None, scale=1.0):
        self.u, self.loglstar, self.rseed = u, loglstar, rseed
        self.prior_transform, self.loglikelihood, self.kwargs = prior_transform, loglikelihood, kwargs
        self.axes, self.scale = axes, scale


class SamplerReturn(tuple):
    """``(u, v, logl, ncall, blob)`` -- the tuple of dynesty 2's walkers -- with dynesty 3's field names as attributes."""
    __slots__ = ()

    def __new__(cls, u, v, logl, ncall, blob):
        return tuple.__new__(cls, (u, v, logl, ncall, blob))

    u = property(lambda self: self[0])
    v = property(lambda self: self[1])
    logl = property(lambda self: self[2])
    ncalls = property(lambda self: self[3])
    tuning_info = property(lambda self: self[4])


class SamplerArgumentBatch(Sequence):
    """What ``prepare_sampler`` returns: the queue's argument records as ONE set of arrays (start points ``u[n, D]``, seeds
    ``[n]``) plus what the records share (bound, callables, walker kwargs).  It is a sequence of :class:`SamplerArgument` -- a
    driver that indexes or iterates it gets the per-record objects -- but ``GPUPool.map`` and ``run_many_device`` read the arrays
    directly: building, then unpacking, 4096 Python records costs more than the queue's whole device time."""

    def __init__(self, u, loglstar, rseeds, prior_transform, loglikelihood, kwargs, axes=None, scale=1.0):
        self.u = np.ascontiguousarray(u, dtype=float)
        self.loglstar, self.rseeds = loglstar, rseeds
        self.prior_transform, self.loglikelihood, self.kwargs = prior_transform, loglikelihood, kwargs
        self.axes, self.scale = axes, scale

    def __len__(self):
        return len(self.u)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        star = self.loglstar[i] if np.ndim(self.loglstar) else self.loglstar
        axes = self.axes[i] if isinstance(self.axes, (list, tuple)) or np.ndim(self.axes) > 2 else self.axes
        return SamplerArgument(u=self.u[i], loglstar=star, rseed=self.rseeds[i], prior_transform=self.prior_transform,
                               loglikelihood=self.loglikelihood, kwargs=self.kwargs, axes=axes, scale=self.scale)

    def keys(self):
        return chain_keys(self.rseeds)

    def bounds(self):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(self.loglstar, dtype=float), (len(self),)))


class WalkResults(Sequence):
    """The queue's results as arrays; item ``q`` is the :class:`SamplerReturn` ``(u, v, logl, ncall, blob)`` dynesty expects, built
    when it is asked for.  ``list(results)`` materialises all records (what dynesty's ``self.queue = list(mapper(...))`` does)."""

    def __init__(self, u, v, logl, ncall, accept, reject, scale, walks=None):
        self.u, self.v, self.logl, self.ncall, self.accept, self.reject = u, v, logl, ncall, accept, reject
        self.scale, self.walks = scale, walks

    def __len__(self):
        return len(self.logl)

    def _blob(self, q):
        blob = {"accept": int(self.accept[q]), "reject": int(self.reject[q]),
                "scale": float(self.scale[q]) if np.ndim(self.scale) else self.scale}
        if self.walks is not None:
            blob["walks"] = int(self.walks[q]) if np.ndim(self.walks) else int(self.walks)
        return blob

    def __getitem__(self, q):
        if isinstance(q, slice):
            return [self[j] for j in range(*q.indices(len(self)))]
        if q < 0:
            q += len(self)
        if not 0 <= q < len(self):
            raise IndexError(q)
        return SamplerReturn(self.u[q], self.v[q], float(self.logl[q]), int(self.ncall[q]), self._blob(q))

    def __iter__(self):
        # (plain Python numbers from tolist(), row views from iterating the arrays, the records built by tuple.__new__ directly in one
        #  comprehension: 4096 records cost ~2 ms this way -- 30 % less than a generator calling SamplerReturn(...) per record, a
        #  third of what numpy scalars cost)
        acc, rej, ncall, ll = (np.asarray(a).tolist() for a in (self.accept, self.reject, self.ncall, self.logl))
        scale = np.broadcast_to(self.scale, (len(self),)).tolist()
        new, cls = tuple.__new__, SamplerReturn
        if self.walks is None:
            return iter([new(cls, (u, v, l, nc, {"accept": a, "reject": r, "scale": s}))
                         for u, v, l, nc, a, r, s in zip(self.u, self.v, ll, ncall, acc, rej, scale)])
        walks = np.broadcast_to(self.walks, (len(self),)).tolist()
        return iter([new(cls, (u, v, l, nc, {"accept": a, "reject": r, "scale": s, "walks": w}))
                     for u, v, l, nc, a, r, s, w in zip(self.u, self.v, ll, ncall, acc, rej, scale, walks)])

    def tuning_summary(self):
        """What ``tune`` needs from the whole queue without materialising a record: mean accepted steps and the walk length."""
        return {"accept": float(np.mean(self.accept)), "walks": None if self.walks is None else float(np.mean(self.walks))}


def estimate_nmcmc(accept_ratio, safety=5, tau=None, maxmcmc=5000, old_act=None):
    """bilby ``dynesty_utils.estimate_nmcmc``: chain length from the acceptance ratio -- autocorrelation time ``2 / a - 1`` of a
    Metropolis chain, smoothed over ``tau`` calls with the previous estimate."""
    if tau is None:
        tau = maxmcmc / safety
    if accept_ratio == 0.0:
        n_exact = np.inf if old_act is None else (1 + 1 / tau) * old_act
    else:
        n_exact = safety * (2 / accept_ratio - 1)
        if old_act is not None:
            n_exact = (1 - 1 / tau) * old_act + n_exact / tau
    return max(safety, float(min(n_exact, maxmcmc)))


def _live_points(args):
    kw = args.kwargs
    for key in ("live", "live_u"):
        if kw.get(key) is not None:
            return np.asarray(kw[key], dtype=float)
    raise KeyError("the sampler argument carries no live points (kwargs['live'])")


class _LockstepWalk:
    """Shared machinery: the move, the boundary rules, the per-point protocol, the lock-step driver.  Subclasses define
    when a chain stops (``_init_state``, ``_continues``)."""

    def __init__(self, ndim=None, nonbounded=None, periodic=None, reflective=None, maxmcmc=5000, **kwargs):
        self.ndim = None if ndim is None else int(ndim)
        self.nonbounded = nonbounded
        self.periodic = np.asarray(periodic if periodic is not None else [], dtype=int)
        self.reflective = np.asarray(reflective if reflective is not None else [], dtype=int)
        self.maxmcmc = int(maxmcmc)
        self.scale = 1.0
        self.nlive = None
        self.sampler_kwargs = dict(ndim=self.ndim, periodic=periodic, reflective=reflective, maxmcmc=self.maxmcmc)
        self.n_batches = self.n_evals = 0

    # ---- dynesty 3 protocol -------------------------------------------------------------------------------------------
    def prepare_sampler(self, loglstar=None, points=None, axes=None, seeds=None, prior_transform=None, loglikelihood=None,
                        nested_sampler=None):
        """One :class:`SamplerArgument` per live point to evolve (dynesty: ``InternalSampler.prepare_sampler``)."""
        live = np.asarray(getattr(nested_sampler, "live_u"), dtype=float)
        self.nlive = len(live)
        kwargs = dict(self.sampler_kwargs, live=live, walks=getattr(self, "walks", None), nlive=self.nlive)
        return SamplerArgumentBatch(points, loglstar, seeds, prior_transform, loglikelihood, kwargs, axes=axes, scale=self.scale)

    def tune(self, tuning_info, update=True):
        return None

    def sample(self, args):
        """One chain, driven alone (what ``pool.map`` does when nothing batches): identical, bit for bit, to the same chain
        inside :meth:`run_many`."""
        return self.run_many([args], lambda v: np.array([args.loglikelihood(v[0])], dtype=float),
                             lambda u: np.asarray(args.prior_transform(u[0]), dtype=float)[None, :])[0]

    __call__ = sample           # dynesty 2: the ``sample=`` object is called on the argument record

    # ---- the move -------------------------------------------------------------------------------------------------------
    def _propose(self, u, live, r):
        """Differential evolution for the rows of ``u`` with uniforms ``r[:, 0:7]``; returns (proposal, inside-the-cube mask)."""
        n_live, ndim = live.shape
        i = np.minimum((r[:, 0] * n_live).astype(int), n_live - 1)
        j = (i + 1 + np.minimum((r[:, 1] * (n_live - 1)).astype(int), n_live - 2)) % n_live        # a different live point
        gamma = np.where(r[:, 2] < 0.5, 1.0,
                         2.38 / np.sqrt(2.0 * ndim) * (-0.25 * np.log(r[:, 3] * r[:, 4] * r[:, 5] * r[:, 6])))   # Gamma(4, 1/4)
        prop = u + gamma[:, None] * (live[j] - live[i])
        if self.periodic.size:
            prop[:, self.periodic] = np.mod(prop[:, self.periodic], 1.0)
        if self.reflective.size:
            q = np.mod(prop[:, self.reflective], 2.0)
            prop[:, self.reflective] = np.where(q > 1.0, 2.0 - q, q)
        return prop, np.all((prop >= 0.0) & (prop <= 1.0), axis=1)

    # ---- the whole queue in lock-step -----------------------------------------------------------------------------------
    def run_many(self, args_list, loglike_many, prior_transform_many=None):
        """Evolve every chain of the queue: per MCMC step one ``prior_transform_many(u[m, D])`` and one
        ``loglike_many(theta[m, D]) -> logL[m]`` for the ``m`` chains still running whose proposal fell inside the unit cube."""
        n = len(args_list)
        if n == 0:
            return []
        first = args_list[0]
        live = _live_points(first)
        shared = all(a.kwargs is first.kwargs or a.kwargs.get("live", a.kwargs.get("live_u")) is
                     first.kwargs.get("live", first.kwargs.get("live_u")) for a in args_list[1:])
        if not shared:          # per-chain ensembles: evolve the groups that do share one
            out = [None] * n
            groups = {}
            for q, a in enumerate(args_list):
                groups.setdefault(id(a.kwargs.get("live", a.kwargs.get("live_u"))), []).append(q)
            for idx in groups.values():
                for q, r in zip(idx, self.run_many([args_list[q] for q in idx], loglike_many, prior_transform_many)):
                    out[q] = r
            return out
        pt_many = prior_transform_many if prior_transform_many is not None else (
            lambda uu: np.stack([np.asarray(first.prior_transform(x), dtype=float) for x in uu]))
        keys = np.array([chain_key(a.rseed) for a in args_list], dtype=np.uint64)
        u = np.stack([np.asarray(a.u, dtype=float) for a in args_list])
        loglstar = np.array([a.loglstar for a in args_list], dtype=float)
        v, logl = np.full_like(u, np.nan), np.full(n, np.nan)
        st = dict(accept=np.zeros(n, dtype=int), reject=np.zeros(n, dtype=int), nfail=np.zeros(n, dtype=int),
                  ncall=np.zeros(n, dtype=int), step=np.zeros(n, dtype=int))
        self._init_state(st, args_list, u)
        self.n_batches = self.n_evals = 0
        active = self._continues(st)
        while active.any():
            idx = np.nonzero(active)[0]
            st["step"][idx] += 1
            r = counter_uniforms(keys[idx], st["step"][idx])
            prop, inside = self._propose(u[idx], live, r)
            st["nfail"][idx[~inside]] += 1
            ev = idx[inside]
            if ev.size:
                v_prop = np.asarray(pt_many(prop[inside]), dtype=float)
                l_prop = np.asarray(loglike_many(v_prop), dtype=float)
                self.n_batches += 1
                self.n_evals += ev.size
                st["ncall"][ev] += 1
                ok = l_prop > loglstar[ev]
                good = ev[ok]
                u[good], v[good], logl[good] = prop[inside][ok], v_prop[ok], l_prop[ok]
                st["accept"][good] += 1
                st["reject"][ev[~ok]] += 1
            self._after_step(st, idx, u)
            active = self._continues(st)
        # a chain that never moved returns a fresh draw from the prior (bilby: "Unable to find a new point using walk")
        stuck = np.nonzero(self._stuck(st))[0]
        if stuck.size:
            u[stuck] = counter_uniforms(keys[stuck], np.zeros(stuck.size, dtype=np.uint64), max(N_DRAWS, u.shape[1]))[:, :u.shape[1]]
            v[stuck] = np.asarray(pt_many(u[stuck]), dtype=float)
            logl[stuck] = np.asarray(loglike_many(v[stuck]), dtype=float)
            self.n_batches += 1
            self.n_evals += stuck.size
            st["ncall"][stuck] += 1
        self._finish(st)
        return [SamplerReturn(u[q], v[q], float(logl[q]), int(st["ncall"][q]), self._blob(st, q, args_list[q])) for q in range(n)]

    def _device_fresh_draws(self, stuck, table, rseeds, u, v, logl, counts, loglike_device, device):
        """A chain of a device walk that never moved returns a fresh draw from the prior, as on the host (``run_many``)."""
        if not stuck.size:
            return
        import torch
        from . import _lib as L
        lib = L.load_library()
        fresh = counter_uniforms(rseeds[stuck], np.zeros(stuck.size, dtype=np.uint64), max(N_DRAWS, u.shape[1]))[:, :u.shape[1]]
        fu = torch.as_tensor(np.ascontiguousarray(fresh), device=f"cuda:{int(device)}")
        fv = torch.empty_like(fu)
        L.check(lib.nmma_walk_rescale(table, u.shape[1], C_void(fu.data_ptr()), stuck.size, C_void(fv.data_ptr()), int(device),
                                      C_void(torch.cuda.current_stream(fu.device).cuda_stream)), "nmma_walk_rescale")
        u[stuck], v[stuck] = fresh, fv.cpu().numpy()
        logl[stuck] = loglike_device(fv).cpu().numpy()
        counts[stuck, 3] += 1
        self.n_batches += 1
        self.n_evals += int(stuck.size)

    # ---- hooks ----------------------------------------------------------------------------------------------------------
    def _init_state(self, st, args_list, u):
        pass

    def _after_step(self, st, idx, u):
        pass

    def _stuck(self, st):
        return st["accept"] == 0

    def _finish(self, st):
        pass

    def _blob(self, st, q, args):
        return {"accept": int(st["accept"][q]), "reject": int(st["reject"][q] + st["nfail"][q]), "scale": getattr(args, "scale", 1.0)}


# -----------------------------------------------------------------------------------------------------------------------
# The walk on the device: likelihood -> accept + next proposal as two launches per MCMC step, no host round trip
# -----------------------------------------------------------------------------------------------------------------------
#: bilby/core/prior/analytical.py class names with a device formula.  EXACT names only (a trailing "Prior" of a stand-in class is
#: dropped): a suffix match would hand ``TruncatedNormal`` or ``LogNormal`` the plain Gaussian's transform and
#: ``SymmetricLogUniform`` the log-uniform's -- different distributions (round-3 advisor finding).
_PRIOR_KINDS = {"Uniform": "uniform", "Sine": "sine", "Cosine": "cosine", "PowerLaw": "powerlaw", "LogUniform": "loguniform",
                "Gaussian": "gaussian", "Normal": "gaussian", "DeltaFunction": "delta",
                "TruncatedGaussian": "truncgaussian", "TruncatedNormal": "truncgaussian",
                "LogNormal": "lognormal", "LogGaussian": "lognormal", "HalfGaussian": "halfgaussian", "HalfNormal": "halfgaussian"}


def device_prior_kind(prior):
    """The device formula of a bilby prior, by its EXACT class name (or None: the host transform is used)."""
    name = type(prior).__name__
    if name.startswith("Conditional") or hasattr(prior, "condition_func") or hasattr(prior, "required_variables"):
        return None             # (bilby's conditional priors depend on other parameters: host transform)
    kind = _PRIOR_KINDS.get(name)
    if kind is None and name.endswith("Prior"):
        kind = _PRIOR_KINDS.get(name[:-5])
    return kind


def device_prior_table(priors, keys, periodic=(), reflective=()):
    """The analytic bilby priors of ``keys`` as the device's table (``nmma_walk_prior``: kind, boundary, a, b, alpha, c), recognised
    by class name (bilby/core/prior/analytical.py: Uniform, Sine, Cosine, PowerLaw, LogUniform, Gaussian / Normal,
    TruncatedGaussian / TruncatedNormal, LogNormal / LogGaussian, HalfGaussian / HalfNormal, DeltaFunction).  Returns ``None``
    when a prior has no device formula -- the caller then stays on the host path."""
    from math import erf, sqrt
    from . import _lib as L
    keys = list(keys)
    if not 1 <= len(keys) <= L.WALK_MAX_DIM:
        return None
    table = (L.WalkPrior * len(keys))()
    periodic, reflective = set(int(i) for i in periodic), set(int(i) for i in reflective)
    for d, key in enumerate(keys):
        pr = priors[key]
        kind = device_prior_kind(pr)
        try:
            if kind == "uniform":
                table[d].kind, table[d].a, table[d].b = L.PRIOR_UNIFORM, float(pr.minimum), float(pr.maximum)
            elif kind == "sine":
                table[d].kind, table[d].a, table[d].b = L.PRIOR_SINE, float(pr.minimum), float(pr.maximum)
            elif kind == "cosine":
                table[d].kind, table[d].a, table[d].b = L.PRIOR_COSINE, float(pr.minimum), float(pr.maximum)
            elif kind == "powerlaw":
                table[d].kind, table[d].a, table[d].b, table[d].alpha = L.PRIOR_POWERLAW, float(pr.minimum), float(pr.maximum), float(pr.alpha)
            elif kind == "loguniform":
                table[d].kind, table[d].a, table[d].b, table[d].alpha = L.PRIOR_POWERLAW, float(pr.minimum), float(pr.maximum), -1.0
            elif kind == "gaussian":
                # (a Gaussian that ALSO carries finite bounds is not bilby's Gaussian: leave it to the host)
                lo, hi = getattr(pr, "minimum", None), getattr(pr, "maximum", None)
                if (lo is not None and np.isfinite(lo)) or (hi is not None and np.isfinite(hi)):
                    return None
                table[d].kind, table[d].a, table[d].b = L.PRIOR_GAUSSIAN, float(pr.mu), float(pr.sigma)
            elif kind == "truncgaussian":
                mu, sigma, lo, hi = float(pr.mu), float(pr.sigma), float(pr.minimum), float(pr.maximum)
                e_lo, e_hi = erf((lo - mu) / (sqrt(2.0) * sigma)), erf((hi - mu) / (sqrt(2.0) * sigma))
                table[d].kind, table[d].a, table[d].b = L.PRIOR_TRUNC_GAUSSIAN, mu, sigma
                table[d].alpha, table[d].c = (e_hi - e_lo) / 2.0, e_lo         # bilby: TruncatedGaussian.normalisation, rescale
            elif kind == "lognormal":
                table[d].kind, table[d].a, table[d].b = L.PRIOR_LOGNORMAL, float(pr.mu), float(pr.sigma)
            elif kind == "halfgaussian":
                table[d].kind, table[d].b = L.PRIOR_HALF_GAUSSIAN, float(pr.sigma)
            elif kind == "delta":
                table[d].kind, table[d].a = L.PRIOR_DELTA, float(pr.peak)
            else:
                return None
        except (AttributeError, TypeError, ValueError):
            return None
        table[d].boundary = L.BOUNDARY_PERIODIC if d in periodic else (L.BOUNDARY_REFLECTIVE if d in reflective else L.BOUNDARY_NONE)
    return table


def device_walk(table, live, u0, loglstar, keys, n_steps, loglike_device, device=0, first_step=1):
    """``n_steps`` lock-step MCMC steps of ``len(u0)`` chains on the GPU (an int, or one walk length per chain: the launches run to
    the longest and a chain stops taking part after its own).  ``loglike_device(theta[n, D] CUDA tensor) -> logL[n]``
    (e.g. ``lambda t: likelihood.log_likelihood_batch(t, names)``).  Returns host arrays (u, v, logl, counts[n, 4] = accept,
    reject, outside-the-cube, likelihood calls); ``logl`` is NaN for a chain that never moved, like the host walk's."""
    import torch
    from . import _lib as L
    lib = L.load_library()
    dev = torch.device(f"cuda:{int(device)}")
    f64 = dict(dtype=torch.float64, device=dev)
    live_d = torch.as_tensor(np.ascontiguousarray(live, dtype=np.float64), device=dev)
    u = torch.as_tensor(np.ascontiguousarray(u0, dtype=np.float64), device=dev)
    n, ndim = u.shape
    star = torch.as_tensor(np.ascontiguousarray(loglstar, dtype=np.float64), device=dev)
    key = torch.as_tensor(np.ascontiguousarray(keys, dtype=np.uint64).view(np.int64), device=dev)
    v, prop, theta = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    logl = torch.full((n,), float("nan"), **f64)
    inside = torch.empty(n, dtype=torch.int32, device=dev)
    counts = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    stream = C_void(torch.cuda.current_stream(dev).cuda_stream)
    per_chain = np.ndim(n_steps) > 0
    lengths = torch.as_tensor(np.ascontiguousarray(n_steps, dtype=np.int32), device=dev) if per_chain else None
    n_steps = int(np.max(n_steps)) if per_chain else int(n_steps)
    ptr = lambda t: C_void(t.data_ptr())
    L.check(lib.nmma_walk_rescale(table, ndim, ptr(u), n, ptr(v), int(device), stream), "nmma_walk_rescale")
    # (the buffers do not move: their addresses are taken once -- a step then costs three ctypes calls and the likelihood's wrapper)
    p_live, p_u, p_v, p_key, p_prop, p_theta, p_in = ptr(live_d), ptr(u), ptr(v), ptr(key), ptr(prop), ptr(theta), ptr(inside)
    p_star, p_logl, p_cnt, n_live, dev_i = ptr(star), ptr(logl), ptr(counts), live_d.shape[0], int(device)
    p_len = ptr(lengths) if per_chain else None
    first, n_steps = int(first_step), int(n_steps)
    if n_steps < 1:
        return u.cpu().numpy(), v.cpu().numpy(), logl.cpu().numpy(), counts.cpu().numpy()
    # two launches per step: the likelihood, and the accept of step k fused with the proposal of step k + 1 (nmma_walk_step)
    L.check(lib.nmma_walk_propose(table, ndim, p_live, n_live, p_u, p_v, p_key, n, first, p_prop, p_theta, p_in, dev_i, stream), "nmma_walk_propose")
    step = lib.nmma_walk_step
    for k in range(1, n_steps + 1):
        l_prop = loglike_device(theta)
        if l_prop.dtype != torch.float64 or not l_prop.is_contiguous():
            l_prop = l_prop.to(torch.float64).contiguous()
        if k < n_steps:
            if step(table, ndim, p_live, n_live, p_key, n, p_prop, p_theta, p_in, ptr(l_prop), p_star, p_u, p_v, p_logl, p_cnt, p_len, k, first, dev_i, stream):
                L.check(1, "nmma_walk_step")
        else:
            L.check(lib.nmma_walk_accept(ndim, n, p_prop, p_theta, p_in, ptr(l_prop), p_star, p_u, p_v, p_logl, p_cnt, p_len, k, dev_i, stream),
                    "nmma_walk_accept")
    return u.cpu().numpy(), v.cpu().numpy(), logl.cpu().numpy(), counts.cpu().numpy()


def device_rwalk(table, live, u0, loglstar, keys, nact, maxmcmc, tau, old_act, loglike_device, device=0, poll=8):
    """The acceptance-tracking walk (``AcceptanceTrackingRWalk``) of ``len(u0)`` chains on the GPU: each chain runs until ``nact``
    of its autocorrelation estimates have passed (``nmma_walk_accept_rwalk``); the host only reads the number of chains still
    running every ``poll`` steps (a finished chain's remaining launches are no-ops for it).  Returns host arrays (u, v, logl,
    counts[n, 4], act[n], steps taken)."""
    import torch
    from . import _lib as L
    lib = L.load_library()
    dev = torch.device(f"cuda:{int(device)}")
    live_d = torch.as_tensor(np.ascontiguousarray(live, dtype=np.float64), device=dev)
    u = torch.as_tensor(np.ascontiguousarray(u0, dtype=np.float64), device=dev)
    n, ndim = u.shape
    star = torch.as_tensor(np.ascontiguousarray(loglstar, dtype=np.float64), device=dev)
    key = torch.as_tensor(np.ascontiguousarray(keys, dtype=np.uint64).view(np.int64), device=dev)
    v, prop, theta = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    logl = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    act = torch.full((n,), float("inf"), dtype=torch.float64, device=dev)
    inside = torch.empty(n, dtype=torch.int32, device=dev)
    active = torch.ones(n, dtype=torch.int32, device=dev)
    counts = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    stream = C_void(torch.cuda.current_stream(dev).cuda_stream)
    ptr = lambda t: C_void(t.data_ptr())
    L.check(lib.nmma_walk_rescale(table, ndim, ptr(u), n, ptr(v), int(device), stream), "nmma_walk_rescale")
    p_live, p_u, p_v, p_key, p_prop, p_theta, p_in = ptr(live_d), ptr(u), ptr(v), ptr(key), ptr(prop), ptr(theta), ptr(inside)
    p_star, p_logl, p_cnt, p_act, p_on, n_live, dev_i = ptr(star), ptr(logl), ptr(counts), ptr(act), ptr(active), live_d.shape[0], int(device)
    old = -1.0 if old_act is None else float(old_act)
    s = 0
    # two launches per step: the likelihood, and the accept of step s fused with the proposal of step s + 1 (nmma_walk_step_rwalk)
    L.check(lib.nmma_walk_propose(table, ndim, p_live, n_live, p_u, p_v, p_key, n, 1, p_prop, p_theta, p_in, dev_i, stream), "nmma_walk_propose")
    while True:
        s += 1
        l_prop = loglike_device(theta)
        if l_prop.dtype != torch.float64 or not l_prop.is_contiguous():
            l_prop = l_prop.to(torch.float64).contiguous()
        L.check(lib.nmma_walk_step_rwalk(table, ndim, p_live, n_live, p_key, n, p_prop, p_theta, p_in, ptr(l_prop), p_star, p_u, p_v, p_logl, p_cnt,
                                         p_act, p_on, s, float(nact), int(maxmcmc), float(tau), old, dev_i, stream), "nmma_walk_step_rwalk")
        if s % int(poll) == 0 and int(active.sum().item()) == 0:
            break
    return u.cpu().numpy(), v.cpu().numpy(), logl.cpu().numpy(), counts.cpu().numpy(), act.cpu().numpy(), s


def C_void(x):
    import ctypes
    return ctypes.c_void_p(int(x))


class EnsembleWalkSampler(_LockstepWalk):
    """``sample="acceptance-walk"`` (mpi_setup.py:221-232): ``int(walks)`` differential-evolution steps per chain; ``tune`` moves
    ``walks`` so that a chain accepts ``naccept`` steps on average (bilby: ``EnsembleWalkSampler.tune``), capped at ``maxmcmc``."""

    def __init__(self, ndim=None, nonbounded=None, periodic=None, reflective=None, maxmcmc=5000, naccept=60, walks=100, **kwargs):
        super().__init__(ndim=ndim, nonbounded=nonbounded, periodic=periodic, reflective=reflective, maxmcmc=maxmcmc, **kwargs)
        self.naccept = naccept
        self.walks = max(2, walks)
        self.sampler_kwargs["walks"] = self.walks

    def _init_state(self, st, args_list, u):
        st["walks"] = np.array([int(a.kwargs.get("walks") or self.walks) for a in args_list])

    def _continues(self, st):
        return st["step"] < st["walks"]

    def _blob(self, st, q, args):
        # (bilby counts every step that did not accept as a rejection, out-of-cube proposals included)
        return {"accept": int(st["accept"][q]), "reject": int(st["walks"][q] - st["accept"][q]), "scale": getattr(args, "scale", 1.0),
                "walks": int(st["walks"][q])}

    def run_many_device(self, args_list, loglike_device, priors, keys, device=0, loglike_many=None, prior_transform_many=None,
                        engine=None, constraints=None):
        """``run_many`` with the whole walk on the GPU: ``int(walks)`` steps of propose -> likelihood -> accept with no host
        round trip.  With ``engine`` (the ``EMEngine`` behind the likelihood; ``GPUPool`` passes it) the queue is ONE library call
        (``EMEngine.walk_queue`` -> ``nmma_em_walk_queue``: packed upload, the step loop, fresh draws, packed download);
        otherwise ``device_walk`` drives the same kernels from Python around ``loglike_device(theta)``.  ``priors`` / ``keys``:
        the sampled priors in column order, needed as a device table (``device_prior_table``); a prior without a device formula
        or chains with their own ensembles send the queue to the host walk ``run_many(args_list, loglike_many,
        prior_transform_many)``.  ``constraints``: the likelihood's lowered Constraint priors (``ConstraintProgram``) for the
        ``engine`` path -- a likelihood called through ``loglike_device`` applies its own.  The random numbers and the proposal
        are the host walk's; only the libm of ``log`` / the prior transform differs in the last bits.  Returns a
        :class:`WalkResults` (a sequence of the records dynesty expects, array-backed)."""
        n = len(args_list)
        if n == 0:
            return []
        batch = args_list if isinstance(args_list, SamplerArgumentBatch) else None
        first = args_list[0]
        live = _live_points(first)
        if batch is not None:
            shared, walks = True, [int(batch.kwargs.get("walks") or self.walks)]
        else:
            shared = all(_live_points(a) is live for a in args_list[1:])
            walks = [int(a.kwargs.get("walks") or self.walks) for a in args_list]
        table = device_prior_table(priors, keys, self.periodic, self.reflective)
        if table is None or not shared or np.asarray(live).shape[0] < 3:
            if loglike_many is None:
                raise ValueError("this queue needs the host walk: pass loglike_many (and prior_transform_many)")
            return self.run_many(args_list, loglike_many, prior_transform_many)
        if batch is not None:
            rseeds, u0, loglstar, scales = batch.keys(), batch.u, batch.bounds(), batch.scale
        else:
            rseeds = chain_keys([a.rseed for a in args_list])
            u0 = np.stack([np.asarray(a.u, dtype=float) for a in args_list])
            loglstar = np.array([a.loglstar for a in args_list], dtype=float)
            scales = np.array([getattr(a, "scale", 1.0) for a in args_list], dtype=float)
        same = len(set(walks)) == 1
        steps = walks[0] if same else np.array(walks, dtype=np.int32)
        if engine is not None:
            # ``engine``: the EMEngine behind the likelihood, or a list of (engine, constraint program) pairs, one per device -- the
            # queue is then SHARDED over them (the reference spreads a queue's chains over its MPI ranks, core/mpi_setup.py:651-667,
            # :679-683): contiguous balanced shards, the live points replicated, every shard's call begun before the first is collected;
            # a chain's path depends on its key only, so the result is the single-device queue's, bit for bit
            # or a ``parallel.ShardedQueue`` (one process per GPU: this rank walks its shard, ONE all-gather of the packed records)
            from .parallel import ShardedQueue, shard_bounds
            # (... or the pool's master-side wrapper of one: anything with ``run(live, u0, loglstar, keys, walks, table=)`` that is not an engine)
            rank_queue = isinstance(engine, ShardedQueue) or (hasattr(engine, "run") and not hasattr(engine, "walk_queue_begin"))
            shards = [] if rank_queue else list(engine) if isinstance(engine, (list, tuple)) else [(engine, constraints)]
            if rank_queue:
                u, v, logl, counts = engine.run(live, u0, loglstar, rseeds, steps, table=table)
            elif len(shards) == 1:
                u, v, logl, counts = shards[0][0].walk_queue(table, live, u0, loglstar, rseeds, steps, constraints=shards[0][1])
            else:
                u0 = np.ascontiguousarray(u0, dtype=float)
                star = np.ascontiguousarray(np.broadcast_to(np.asarray(loglstar, dtype=float), (n,)))
                keys_all = np.ascontiguousarray(rseeds, dtype=np.uint64)
                begun, parts, failure = [], [], None
                try:
                    for r, (eng_r, con_r) in enumerate(shards):
                        lo, hi = shard_bounds(n, len(shards), r)
                        if hi > lo:
                            st = steps if same else np.ascontiguousarray(steps[lo:hi])
                            begun.append((eng_r, eng_r.walk_queue_begin(table, live, u0[lo:hi], star[lo:hi], keys_all[lo:hi], st, constraints=con_r)))
                except Exception as exc:      # noqa: BLE001  (a shard that could not begin: the begun ones are still collected below)
                    failure = exc
                # every begun shard is collected whatever happened to the others: a workspace left "pending" would refuse every
                # later queue on its engine
                for eng_r, tok in begun:
                    try:
                        parts.append(eng_r.walk_queue_end(tok))
                    except Exception as exc:      # noqa: BLE001
                        failure = failure or exc
                if failure is not None:
                    raise failure
                u, v, logl, counts = (np.concatenate([p[i] for p in parts]) for i in range(4))
            self.n_batches, self.n_evals = max(walks) + int(np.any(counts[:, 0] == 0)), int(counts[:, 3].sum())
        else:
            u, v, logl, counts = device_walk(table, live, u0, loglstar, rseeds, steps, loglike_device, device=device)
            self.n_batches, self.n_evals = max(walks), int(counts[:, 3].sum())
            self._device_fresh_draws(np.nonzero(counts[:, 0] == 0)[0], table, rseeds, u, v, logl, counts, loglike_device, device)
        wl = walks[0] if same else np.array(walks)
        return WalkResults(u, v, logl, counts[:, 3], counts[:, 0], wl - counts[:, 0], scales, walks=wl)

    def tune(self, tuning_info, update=True):
        """Steer the walk length towards ``naccept`` accepted steps; ``delay`` averages over about a tenth of the live points.
        The acceptance probability is taken over the walk length the reporting chain actually used (``tuning_info["walks"]``):
        a lock-step queue returns hundreds of chains that all ran with the length of the moment they were queued -- bilby's
        ``accept / self.walks`` is the same number when chains return one at a time."""
        if not update:
            return
        nlive = self.nlive if self.nlive is not None else 0
        accept_prob = max(0.5, tuning_info["accept"]) / tuning_info.get("walks", self.walks)
        delay = max(nlive // 10 - 1, 0)
        self.walks = min((self.walks * delay + self.naccept / accept_prob) / (delay + 1), self.maxmcmc)
        self.sampler_kwargs["walks"] = self.walks


class AcceptanceTrackingRWalk(_LockstepWalk):
    """``sample="rwalk"`` (mpi_setup.py:234-245): the chain runs until ``nact`` autocorrelation times have passed, the
    autocorrelation time being estimated from the running acceptance ratio (``estimate_nmcmc`` with ``safety=1``, smoothed over
    ``nlive`` calls with the estimate the previous chains left behind in ``old_act``), at most ``maxmcmc`` evaluations."""
    old_act = None

    def __init__(self, ndim=None, nonbounded=None, periodic=None, reflective=None, maxmcmc=5000, nact=40, **kwargs):
        super().__init__(ndim=ndim, nonbounded=nonbounded, periodic=periodic, reflective=reflective, maxmcmc=maxmcmc, **kwargs)
        self.nact = nact
        self.thin = 1

    def _init_state(self, st, args_list, u):
        st["act"] = np.full(len(args_list), np.inf)
        st["tau"] = np.array([a.kwargs.get("nlive") or a.kwargs.get("walks") or 100 for a in args_list], dtype=float)
        st["old_act"] = type(self).old_act        # every chain of the queue starts from the same previous estimate

    def _continues(self, st):
        done = st["accept"] + st["reject"]
        return (st["step"] < self.nact * st["act"]) & (done <= self.maxmcmc)

    def _after_step(self, st, idx, u):
        a, r, f = st["accept"][idx], st["reject"][idx], st["nfail"][idx]
        upd = (a + r) > self.nact
        for q in np.nonzero(upd)[0]:
            st["act"][idx[q]] = estimate_nmcmc(a[q] / (a[q] + r[q] + f[q]), safety=1, tau=st["tau"][idx[q]], maxmcmc=self.maxmcmc,
                                               old_act=st["old_act"])

    def _stuck(self, st):
        return ~(np.isfinite(st["act"]) & (st["accept"] > 0))

    def run_many_device(self, args_list, loglike_device, priors, keys, device=0, loglike_many=None, prior_transform_many=None):
        """``run_many`` with the walk on the GPU (``device_rwalk``): the per-chain autocorrelation estimate and stop rule are
        evaluated in the accept kernel.  Queues the device cannot take (a prior without a device formula, per-chain ensembles or
        smoothing lengths) go to the host walk."""
        n = len(args_list)
        if n == 0:
            return []
        live = _live_points(args_list[0])
        shared = all(_live_points(a) is live for a in args_list[1:])
        taus = {float(a.kwargs.get("nlive") or a.kwargs.get("walks") or 100) for a in args_list}
        table = device_prior_table(priors, keys, self.periodic, self.reflective)
        if table is None or not shared or len(taus) != 1 or np.asarray(live).shape[0] < 3:
            if loglike_many is None:
                raise ValueError("this queue needs the host walk: pass loglike_many (and prior_transform_many)")
            return self.run_many(args_list, loglike_many, prior_transform_many)
        if isinstance(args_list, SamplerArgumentBatch):
            rseeds, u0, loglstar = args_list.keys(), args_list.u, args_list.bounds()
        else:
            rseeds = chain_keys([a.rseed for a in args_list])
            u0 = np.stack([np.asarray(a.u, dtype=float) for a in args_list])
            loglstar = np.array([a.loglstar for a in args_list], dtype=float)
        u, v, logl, counts, act, steps = device_rwalk(table, live, u0, loglstar, rseeds, self.nact, self.maxmcmc, taus.pop(), type(self).old_act,
                                                      loglike_device, device=device)
        self.n_batches, self.n_evals = steps, int(counts[:, 3].sum())
        st = dict(act=act, accept=counts[:, 0])
        self._device_fresh_draws(np.nonzero(self._stuck(st))[0], table, rseeds, u, v, logl, counts, loglike_device, device)
        self._finish(st)
        scales = args_list.scale if isinstance(args_list, SamplerArgumentBatch) else np.array([getattr(a, "scale", 1.0) for a in args_list])
        return WalkResults(u, v, logl, counts[:, 3], counts[:, 0], counts[:, 1] + counts[:, 2], scales)

    def _finish(self, st):
        fin = st["act"][np.isfinite(st["act"])]
        if fin.size:
            type(self).old_act = float(np.median(fin))      # (a queue leaves ONE estimate behind: the median of its chains')


def integrated_autocorrelation_time(x, c=5.0):
    """Integrated autocorrelation time of the columns of ``x[n, D]`` (FFT autocorrelation, Sokal's automatic window
    ``M >= c tau``); the largest over the columns."""
    x = np.asarray(x, dtype=float)
    n = x.shape[0]
    if n < 8:
        return np.inf
    x = x - x.mean(axis=0)
    size = 1 << (2 * n - 1).bit_length()
    f = np.fft.rfft(x, n=size, axis=0)
    acf = np.fft.irfft(f * np.conj(f), n=size, axis=0)[:n]
    var = acf[0]
    keep = var > 0
    if not np.any(keep):
        return np.inf
    rho = acf[:, keep] / var[keep]
    taus = 2.0 * np.cumsum(rho, axis=0) - 1.0
    m = np.arange(n)[:, None]
    ok = m >= c * taus
    first = np.where(ok.any(axis=0), ok.argmax(axis=0), n - 1)
    return float(np.max(taus[first, np.arange(taus.shape[1])]))


class ACTTrackingEnsembleWalk(_LockstepWalk):
    """``sample="act-walk"`` (mpi_setup.py:209-220): the chain length follows the autocorrelation time MEASURED on the chain:
    every ``check_interval`` steps the integrated autocorrelation time of the positions visited so far is re-estimated, and the
    chain stops once it is ``nact`` of them long (at most ``thin x maxmcmc`` steps).  bilby's class additionally caches the
    thinned remainder of a long chain for later calls; in lock-step every record of the queue gets a chain of its own."""
    check_interval = 50

    def __init__(self, ndim=None, nonbounded=None, periodic=None, reflective=None, maxmcmc=5000, nact=2, **kwargs):
        super().__init__(ndim=ndim, nonbounded=nonbounded, periodic=periodic, reflective=reflective, maxmcmc=maxmcmc, **kwargs)
        self.nact = nact
        self.thin = nact
        self.act = 1.0

    def _init_state(self, st, args_list, u):
        st["act"] = np.full(len(args_list), np.inf)
        st["hist"] = [[] for _ in args_list]

    def _continues(self, st):
        return (st["step"] < self.nact * st["act"]) & (st["step"] < self.thin * self.maxmcmc)

    def _after_step(self, st, idx, u):
        for q in idx:
            st["hist"][q].append(u[q].copy())
            if st["step"][q] % self.check_interval == 0 and st["accept"][q] > 0:
                st["act"][q] = max(1.0, integrated_autocorrelation_time(np.asarray(st["hist"][q])))

    def _finish(self, st):
        fin = st["act"][np.isfinite(st["act"])]
        if fin.size:
            self.act = float(np.median(fin))


class LockstepEnsembleWalk(EnsembleWalkSampler):
    """Round-2 name of the fixed-length ensemble walk (``walks`` steps, continuing up to ``maxmcmc`` until the chain has accepted
    once); kept for callers that construct it positionally."""

    def __init__(self, ndim, walks=100, maxmcmc=5000, periodic=None, reflective=None):
        super().__init__(ndim=ndim, periodic=periodic, reflective=reflective, maxmcmc=maxmcmc, walks=walks)

    def _continues(self, st):
        return (st["step"] < st["walks"]) | ((st["accept"] == 0) & (st["step"] < self.maxmcmc))

    def _blob(self, st, q, args):
        return {"accept": int(st["accept"][q]), "reject": int(st["reject"][q] + st["nfail"][q]), "scale": 1.0}

    def run_many_chains(self, args_list, loglike_many, prior_transform_many=None):
        """Every chain driven on its own (queue of one): the reference the lock-step form is bit-identical to."""
        return [self.run_many([a], loglike_many, prior_transform_many)[0] for a in args_list]
