"""Sampler-side batching: the pieces that let NMMA's nested-sampling drivers feed the GPU whole batches
instead of one parameter vector per call (``nmma/core/mpi_setup.py:209-242``,
``:282-303``, ``:339``; ``nmma/core/base.py:316-329``).

The reference evolves each live point with an MCMC chain (bilby's ``EnsembleWalkSampler`` /
``ACTTrackingEnsembleWalk`` passed to dynesty as ``sample=``); dynesty hands ``queue_size`` such chains to
``pool.map``.  Every chain calls the likelihood once per step -- so a GPU sees one point at a time.
:class:`LockstepEnsembleWalk` is the same kind of object (callable on one argument record, usable as
``sample=``), written as a generator that *yields* the point it wants evaluated.  Driven one chain at a
time it behaves like the reference's walker; given to :meth:`nmma_amd.pool.GPUPool.map` with a whole queue
it advances all chains in lock-step: one batched prior transform and ONE likelihood launch per MCMC step.

Argument record (the fields of dynesty's ``SamplerArgument`` that an ensemble walk uses): ``u`` (start, unit
cube), ``loglstar``, ``rseed`` (int or numpy Generator), ``prior_transform``, ``loglikelihood`` and
``kwargs["live_u"]`` (the current live points, unit cube, the proposal ensemble).  Return value per chain:
``(u, v, logl, ncall, blob)`` with ``blob = {"accept": ..., "reject": ..., "scale": ...}``.
dynesty / bilby are not installed in the build image: the protocol is exercised by a fake driver in
``tests/test_sampler_adapter.py``; INTEGRATION.md shows the wiring into ``mpi_setup.py``.
"""
from __future__ import annotations

import numpy as np


class BatchedPriorTransform:
    """``prior_transform`` for arrays of unit-cube points: ``u[B, D] -> theta[B, D]`` with one vectorised
    ``rescale`` per parameter (bilby priors' ``rescale`` are numpy ufunc-style; ``priors.rescale(keys, u)``
    of ``core/mpi_setup.py:682-683`` is the per-point form)."""

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


def _generator(rseed):
    return rseed if isinstance(rseed, np.random.Generator) else np.random.default_rng(rseed)


class LockstepEnsembleWalk:
    """Differential-evolution ensemble random walk (the move of bilby's ``EnsembleWalkSampler``:
    ``u' = u + gamma (a - b)`` with ``a, b`` two other live points, ``gamma = 2.38 / sqrt(2 ndim)`` scaled by a
    log-normal factor, or 1 with probability 0.5 to jump between modes), accepted when the new point is inside
    the unit cube (after periodic / reflective wrapping) and ``logL > loglstar``.  A chain runs ``walks``
    steps and continues (up to ``maxmcmc``) until it has accepted at least once."""

    def __init__(self, ndim, walks=100, maxmcmc=5000, periodic=None, reflective=None):
        self.ndim, self.walks, self.maxmcmc = int(ndim), int(walks), int(maxmcmc)
        self.periodic = np.asarray(periodic if periodic is not None else [], dtype=int)
        self.reflective = np.asarray(reflective if reflective is not None else [], dtype=int)

    # ---- one chain as a coroutine: yields unit-cube proposals, receives (v, logl) -------------------
    def _wrap(self, u):
        u = u.copy()
        if self.periodic.size:
            u[self.periodic] = np.mod(u[self.periodic], 1.0)
        if self.reflective.size:
            r = np.mod(u[self.reflective], 2.0)
            u[self.reflective] = np.where(r > 1.0, 2.0 - r, r)
        return u

    def chain(self, args):
        rng = _generator(args.rseed)
        live = np.asarray(args.kwargs["live_u"], dtype=float)
        n_live = len(live)
        u = np.asarray(args.u, dtype=float).copy()
        v, logl = None, None
        accept = reject = ncall = 0
        gamma0 = 2.38 / np.sqrt(2.0 * self.ndim)
        step = 0
        while step < self.walks or (accept == 0 and step < self.maxmcmc):
            step += 1
            i, j = rng.choice(n_live, size=2, replace=False)
            gamma = 1.0 if rng.random() < 0.5 else gamma0 * np.exp(0.5 * rng.standard_normal())
            prop = self._wrap(u + gamma * (live[i] - live[j]))
            if np.any(prop < 0.0) or np.any(prop > 1.0):
                reject += 1
                continue
            v_prop, logl_prop = yield prop
            ncall += 1
            if logl_prop > args.loglstar:
                u, v, logl = prop, v_prop, logl_prop
                accept += 1
            else:
                reject += 1
        if v is None:                      # never moved: the start point itself, evaluated through the same channel
            v, logl = yield u              # (batched with the other chains' requests in lock-step mode)
            ncall += 1
        return u, v, logl, ncall, {"accept": accept, "reject": reject, "scale": 1.0}

    # ---- per-point protocol (what dynesty calls through pool.map when nothing batches) -------------
    def __call__(self, args):
        gen = self.chain(args)
        try:
            prop = next(gen)
            while True:
                v = args.prior_transform(prop)
                prop = gen.send((v, args.loglikelihood(v)))
        except StopIteration as stop:
            return stop.value

    # ---- the whole queue in lock-step -----------------------------------------------------------------
    def run_many_chains(self, args_list, loglike_many, prior_transform_many=None):
        """Lock-step over the per-chain coroutines: per MCMC step ONE ``prior_transform_many(u[B, D])`` and ONE
        ``loglike_many(theta[B, D]) -> logL[B]``; every chain keeps its own random stream, so the results equal those of
        ``__call__`` chain by chain, bit for bit.  The Python work per chain and step (~10 us) bounds it to ~1e5 proposals/s:
        the reference for :meth:`run_many`, not the fast path."""
        gens = [self.chain(a) for a in args_list]
        results = [None] * len(gens)
        pending = {}
        for idx, g in enumerate(gens):
            try:
                pending[idx] = next(g)
            except StopIteration as stop:
                results[idx] = stop.value
        self.n_batches, self.n_evals = 0, 0
        while pending:
            order = list(pending)
            u = np.stack([pending[i] for i in order])
            if prior_transform_many is not None:
                v = np.asarray(prior_transform_many(u))
            else:
                v = np.stack([args_list[i].prior_transform(u[q]) for q, i in enumerate(order)])
            logl = np.asarray(loglike_many(v), dtype=float)
            self.n_batches += 1
            self.n_evals += len(order)
            for q, i in enumerate(order):
                try:
                    pending[i] = gens[i].send((v[q], float(logl[q])))
                except StopIteration as stop:
                    results[i] = stop.value
                    del pending[i]
        return results

    def run_many(self, args_list, loglike_many, prior_transform_many=None):
        """The same walk for the whole queue with the chains as rows of numpy arrays: no Python work per chain, one random
        stream for all chains (seeded from every chain's ``rseed``).  Same move, same acceptance rule, same stopping rule and
        the same return tuples as :meth:`chain`; only the random numbers a given chain sees differ from its solo run."""
        n = len(args_list)
        if n == 0:
            return []
        first = args_list[0]
        live = np.asarray(first.kwargs["live_u"], dtype=float)
        if any(a.kwargs["live_u"] is not first.kwargs["live_u"] for a in args_list[1:]):
            return self.run_many_chains(args_list, loglike_many, prior_transform_many)     # (per-chain ensembles: no common array)
        n_live, ndim = live.shape
        seeds = [a.rseed for a in args_list]
        if all(isinstance(x, (int, np.integer)) for x in seeds):
            rng = np.random.default_rng(np.random.SeedSequence([int(x) & 0xFFFFFFFF for x in seeds]))
        else:
            rng = _generator(seeds[0])
        pt_many = prior_transform_many if prior_transform_many is not None else (
            lambda uu: np.stack([first.prior_transform(x) for x in uu]))
        u = np.stack([np.asarray(a.u, dtype=float) for a in args_list])
        loglstar = np.array([a.loglstar for a in args_list], dtype=float)
        v = np.full_like(u, np.nan)
        logl = np.full(n, np.nan)
        accept = np.zeros(n, dtype=int)
        reject = np.zeros(n, dtype=int)
        ncall = np.zeros(n, dtype=int)
        step = np.zeros(n, dtype=int)
        gamma0 = 2.38 / np.sqrt(2.0 * self.ndim)
        self.n_batches, self.n_evals = 0, 0
        active = np.ones(n, dtype=bool) if (self.walks > 0 or self.maxmcmc > 0) else np.zeros(n, dtype=bool)
        while active.any():
            idx = np.nonzero(active)[0]
            m = idx.size
            step[idx] += 1
            i = rng.integers(n_live, size=m)
            j = (i + 1 + rng.integers(n_live - 1, size=m)) % n_live            # two different live points
            gamma = np.where(rng.random(m) < 0.5, 1.0, gamma0 * np.exp(0.5 * rng.standard_normal(m)))
            prop = u[idx] + gamma[:, None] * (live[i] - live[j])
            if self.periodic.size:
                prop[:, self.periodic] = np.mod(prop[:, self.periodic], 1.0)
            if self.reflective.size:
                r = np.mod(prop[:, self.reflective], 2.0)
                prop[:, self.reflective] = np.where(r > 1.0, 2.0 - r, r)
            inside = np.all((prop >= 0.0) & (prop <= 1.0), axis=1)
            reject[idx[~inside]] += 1
            ev = idx[inside]
            if ev.size:
                v_prop = np.asarray(pt_many(prop[inside]))
                l_prop = np.asarray(loglike_many(v_prop), dtype=float)
                self.n_batches += 1
                self.n_evals += ev.size
                ncall[ev] += 1
                ok = l_prop > loglstar[ev]
                good = ev[ok]
                u[good] = prop[inside][ok]
                v[good] = v_prop[ok]
                logl[good] = l_prop[ok]
                accept[good] += 1
                reject[ev[~ok]] += 1
            active = (step < self.walks) | ((accept == 0) & (step < self.maxmcmc))
        still = np.nonzero(accept == 0)[0]          # never moved: the start points themselves, one more batch
        if still.size:
            v[still] = np.asarray(pt_many(u[still]))
            logl[still] = np.asarray(loglike_many(v[still]), dtype=float)
            self.n_batches += 1
            self.n_evals += still.size
            ncall[still] += 1
        return [(u[q], v[q], float(logl[q]), int(ncall[q]), {"accept": int(accept[q]), "reject": int(reject[q]), "scale": 1.0})
                for q in range(n)]
