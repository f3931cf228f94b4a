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
        if v is None:                      # never moved: the start point itself (its logL is the caller's)
            v = args.prior_transform(u)
            logl = args.loglikelihood(v)
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
    def run_many(self, args_list, loglike_many, prior_transform_many=None):
        """Advance every chain of ``args_list`` together: per MCMC step ONE ``prior_transform_many(u[B, D])`` and
        ONE ``loglike_many(theta[B, D]) -> logL[B]``.  Chains that finish drop out of the batch."""
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
