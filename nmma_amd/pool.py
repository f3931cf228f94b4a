"""schwimmbad-style pool whose ``map`` evaluates a whole list of parameter vectors in ONE
kernel launch -- the batching seam of the reference's samplers
(``nmma/core/mpi_setup.py:282-285, :298-303, :339, :651-654``: ``sampler.pool`` /
``queue_size`` / ``mapper = pool.map``; pool API = ``map``, ``size``, ``is_master()``,
context manager).

Usage with dynesty / parallel-bilby style drivers::

    pool = GPUPool(likelihood, queue_size=4096)
    sampler = dynesty.NestedSampler(pool.log_likelihood, prior_transform, ndim,
                                    pool=pool, queue_size=pool.size, use_pool={"loglikelihood": True})

``pool.map(pool.log_likelihood, thetas)`` recognises its own callable and sends the list
to the GPU as one batch; ``pool.map(walker, sampler_arguments)`` with a lock-step walker
(:class:`nmma_amd.sampler.LockstepEnsembleWalk`, the ``sample=`` object) advances all chains of the queue
together, one launch per MCMC step -- on the GPU end to end (proposal, prior transform, likelihood, accept) when the pool knows
the sampled ``priors`` and they are analytic bilby priors; any other function is mapped serially on the host.
"""
from __future__ import annotations

import numpy as np


class GPUPool:
    def __init__(self, likelihood, queue_size=4096, names=None, prior_transform_many=None, priors=None, device=0, device_walk=True,
                 devices=None, group=None, master_worker=False):
        """``devices``: several HIP devices driven from this one process (e.g. ``range(8)``) -- a queue of the device walk and a
        batch of ``log_likelihood`` calls are then SHARDED over them, contiguous balanced shards with the live set replicated, as
        the reference's task farm spreads a queue's chains over its MPI ranks (core/mpi_setup.py:651-667, :679-683); results come
        back in queue order and are the single-device results bit for bit.  Default: the likelihood's own device only.
        ``group``: one process per GPU instead (``torchrun``; ``True`` = the default ``torch.distributed`` group, or a group object):
        EVERY rank runs the same sampler on the same seeds and calls ``map`` with the same queue; this rank walks / evaluates its
        contiguous shard on its own device and the shards are exchanged with ONE all-gather per queue / batch (RCCL over xGMI:
        ``parallel.ShardedQueue`` -- records packed on the device, one download -- and ``parallel.ShardedEvaluator``), so every rank
        continues with the full result, bit-identical to the single-device run.
        ``master_worker=True`` (with ``group``): the reference's own structure instead of SPMD (core/mpi_setup.py:651-667 -- ``with POOL()
        as pool: if pool.is_master(): run the sampler``): only rank 0 runs the sampler; the other ranks block in ``wait()`` (entered by
        ``with pool:`` / ``pool.wait()``) and serve -- rank 0's ``map`` broadcasts the queue (one object + one byte broadcast), every rank
        walks its shard, ONE all-gather brings the records back -- until rank 0 closes the pool.  Nothing a worker does can diverge from
        the master (wall-clock checkpoints, ``max_run_time``): the sampler exists once."""
        self.likelihood = likelihood
        self.devices = [int(d) for d in devices] if devices is not None else None
        if group is not None and devices is not None:
            raise ValueError("GPUPool: `devices` (one process, several GPUs) and `group` (one process per GPU) exclude each other")
        self.group = group
        self.master_worker = bool(master_worker) and group is not None
        self._closed = False
        self._rank_queue = None           # parallel.ShardedQueue over `group`, built on first use
        self._rank_eval = None            # parallel.ShardedEvaluator over `group`
        self._shard_engines = None        # [(engine, constraint program)] per entry of `devices`, built on first use
        self._evaluator = None
        self.size = int(queue_size)
        self.names = names
        self.prior_transform_many = prior_transform_many     # e.g. nmma_amd.sampler.BatchedPriorTransform
        # with the sampled priors (dict name -> bilby prior, `names` = the column order) a fixed-length ensemble walk runs ENTIRELY
        # on the GPU -- likelihood -> accept + next proposal, two launches per MCMC step (nmma_amd.sampler.device_walk) -- when every
        # prior has a device formula; otherwise, and for the ACT-tracking walkers, the host walk below
        self.priors = priors
        self.device = int(device)
        self.device_walk = bool(device_walk)
        self.n_batches = 0
        self.n_evals = 0

    # ---- schwimmbad surface
    def _dist(self):
        import torch.distributed as dist
        return dist

    def _grp(self):
        return None if self.group is True else self.group

    def is_master(self):
        if not self.master_worker:
            return True
        return self._dist().get_rank(self._grp()) == 0

    def is_worker(self):
        return not self.is_master()

    def _rank_device(self):
        eng = self._walk_engine()[0] if self.names is not None else None
        return f"cuda:{eng.device if eng is not None else self.device}"

    def _queue_for(self, eng, prog):
        q = self._rank_queue
        if q is None or q.engine is not eng or q.constraints is not prog:
            from .parallel import ShardedQueue
            q = self._rank_queue = ShardedQueue(engine=eng, constraints=prog, group=self._grp())
        return q

    def _evaluate_sharded(self, theta):
        """This rank's row shard of ``theta`` on its device, then ONE all-gather of the log L shards (parallel.ShardedEvaluator)."""
        import torch
        if self._rank_eval is None:
            from .parallel import ShardedEvaluator
            self._rank_eval = ShardedEvaluator(lambda th: self.likelihood.log_likelihood_batch(th, self.names), group=self._grp())
        res = self._rank_eval.evaluate(torch.as_tensor(theta).to(self._rank_device()))      # (the shard is evaluated and exchanged on the device)
        return res.cpu().numpy() if hasattr(res, "cpu") else np.asarray(res)

    def wait(self, callback=None):
        """schwimmbad's worker loop (master / worker form): a rank other than 0 serves rank 0's ``map`` calls -- receive the queue, walk
        / evaluate this rank's shard, take part in the all-gather -- until rank 0 closes the pool.  Rank 0, and every rank of the SPMD
        form, returns at once."""
        if self.is_master():
            return None
        from . import _lib as L
        from .parallel import recv_command
        dist, grp = self._dist(), self._grp()
        while True:
            header, arr = recv_command(dist, grp, self._rank_device())
            op = header["op"]
            if op == "close":
                self._closed = True
                return None
            if op == "loglike":
                self._evaluate_sharded(arr["theta"])
            elif op == "walk":
                eng, prog = self._walk_engine()
                if eng is None:
                    raise RuntimeError("GPUPool worker: the master sent a device queue but this rank's likelihood has no engine form")
                table = (L.WalkPrior * len(self.names)).from_buffer_copy(arr["table"].tobytes())
                walks = arr["walks"] if header["per_chain"] else int(arr["walks"][0])
                self._queue_for(eng, prog).run(arr["live"], arr["u0"], arr["loglstar"], arr["keys"], walks, table=table)
            else:
                raise RuntimeError(f"GPUPool worker: unknown command {op!r}")

    def close(self):
        if self.master_worker and not self._closed and self.is_master():
            from .parallel import send_command
            send_command(self._dist(), self._grp(), {"op": "close"}, {}, self._rank_device())
        self._closed = True
        self._drop_shards()
        self._rank_queue = self._rank_eval = None
        return None

    def __enter__(self):
        if self.master_worker and not self.is_master():
            self.wait()           # (a worker serves here until the master leaves ITS with-block; then it falls through an empty body)
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- the batched callable
    def log_likelihood(self, theta):
        """One vector -> float (what the sampler thinks it maps)."""
        return float(self.log_likelihood_many([theta])[0])

    def log_likelihood_many(self, thetas):
        if isinstance(thetas, np.ndarray) and thetas.ndim == 2:
            theta = np.ascontiguousarray(thetas, dtype=float)          # (the lock-step walker hands over arrays)
        else:
            theta = np.ascontiguousarray(np.stack([np.asarray(t, dtype=float) for t in thetas]))
        shards = self._walk_engines() if self.devices is not None and len(self.devices) > 1 else None
        if self.group is not None:
            # one process per GPU: this rank's row shard, then ONE all-gather of the logL shards (parallel.ShardedEvaluator); in the
            # master / worker form rank 0 first hands the batch to the ranks that wait
            if self.master_worker:
                from .parallel import send_command
                send_command(self._dist(), self._grp(), {"op": "loglike"}, {"theta": theta}, self._rank_device())
            out = self._evaluate_sharded(theta)
        elif shards and all(prog is None for _, prog in shards):
            # several devices: row shards launched asynchronously on every device, gathered on the first (no collective library)
            if self._evaluator is None:
                from .parallel import MultiDeviceEvaluator
                self._evaluator = MultiDeviceEvaluator(None, [e.device for e, _ in shards], engines=[e for e, _ in shards])
            out = self._evaluator.evaluate(theta).cpu().numpy()
        else:
            out = self.likelihood.log_likelihood_batch(theta, self.names)
        self.n_batches += 1
        self.n_evals += len(theta)
        return np.asarray(out)

    def _log_likelihood_device(self, theta):
        """theta[n, D] CUDA tensor -> logL[n] CUDA tensor (the device walk's likelihood step)."""
        return self.likelihood.log_likelihood_batch(theta, self.names)

    def _walk_engine(self):
        """(engine, constraint program) when the likelihood is ONE EM surrogate likelihood -- its queue then runs as one library
        call (``EMEngine.walk_queue``); ``(None, None)`` for everything else (joint likelihoods, combined models, constraint sets
        without a device form): the walk is then driven step by step around ``log_likelihood_batch``."""
        lik = self.likelihood
        sub = getattr(lik, "sub_model", None)
        if sub is None or not hasattr(sub, "engine") or hasattr(getattr(sub, "light_curve_model", None), "stacked_lightcurves_abs"):
            return None, None
        from .em.em_likelihood import EMTransientLikelihood
        if getattr(type(lik), "log_likelihood_batch", None) is not EMTransientLikelihood.log_likelihood_batch:
            return None, None                     # (another likelihood class, or a subclass with its own batched evaluation)
        names = list(self.names)
        eng = sub.engine(names)
        prog = None
        if getattr(lik, "constraints", None):
            _, fixed = sub.sampling_layout()
            prog = lik.device_constraints(names, {k: v for k, v in fixed.items() if k not in names}, eng.device)
            if prog is None:
                return None, None
        return eng, prog

    def _drop_shards(self):
        if self._shard_engines:
            for e, _ in self._shard_engines[1:]:        # (the first one belongs to the likelihood)
                e.close()
        self._shard_engines = None
        self._evaluator = None

    def _walk_engines(self):
        """One (engine, constraint program) per entry of ``devices`` (the likelihood's own engine first; further engines are copies
        of it on the other devices -- or on the same one, which is how the sharding is tested on a single GPU); None when the
        likelihood has no single-engine form."""
        eng, prog = self._walk_engine()
        if self._shard_engines is not None:
            # the likelihood rebuilds its engine when the sampled names or the detection limit change: copies made from the old one
            # would keep its data -- results that differ between devices -- so they go with it
            if eng is self._shard_engines[0][0] and prog is self._shard_engines[0][1]:
                return self._shard_engines
            self._drop_shards()
        if eng is None:
            return None
        lik, sub = self.likelihood, self.likelihood.sub_model
        names = list(self.names)
        shards = [(eng, prog)]
        for d in (self.devices or [eng.device])[1:]:
            model = sub.light_curve_model
            e = sub._build_engine(names, model.gpu_filters, dict(model.engine_kwargs(), device=int(d)))
            p = None
            if getattr(lik, "constraints", None):
                _, fixed = sub.sampling_layout()
                p = lik.device_constraints(names, {k: v for k, v in fixed.items() if k not in names}, int(d))
                if p is None:
                    e.close()
                    return None
            shards.append((e, p))
        self._shard_engines = shards
        return shards

    def map(self, func, iterable, callback=None):
        from .sampler import SamplerArgumentBatch
        items = iterable if isinstance(iterable, (list, SamplerArgumentBatch)) else list(iterable)
        if not len(items):
            return []
        if getattr(func, "__self__", None) is self and getattr(func, "__func__", None) is GPUPool.log_likelihood:
            res = list(self.log_likelihood_many(items))
        elif hasattr(func, "run_many") or hasattr(getattr(func, "__self__", None), "run_many"):
            # a lock-step walker of nmma_amd.sampler -- the `sample=` object itself (dynesty 2) or its bound `sample` method
            # (dynesty 3): the queue of chains advances together, one likelihood launch per MCMC step
            walker = func if hasattr(func, "run_many") else func.__self__
            if self.device_walk and self.priors is not None and self.names is not None and hasattr(walker, "run_many_device"):
                kw = {}
                if "engine" in walker.run_many_device.__code__.co_varnames:
                    kw["engine"], kw["constraints"] = self._walk_engine()
                    if kw["engine"] is not None and self.devices is not None and len(self.devices) > 1:
                        shards = self._walk_engines()
                        if shards is not None:
                            kw["engine"], kw["constraints"] = shards, None
                    elif kw["engine"] is not None and self.group is not None:
                        q = self._queue_for(kw["engine"], kw["constraints"])
                        kw["engine"], kw["constraints"] = (_MasterQueue(self, q) if self.master_worker else q), None
                res = walker.run_many_device(items, self._log_likelihood_device, self.priors, self.names, device=self.device,
                                             loglike_many=self.log_likelihood_many, prior_transform_many=self.prior_transform_many, **kw)
                self.n_batches += getattr(walker, "n_batches", 0)
                self.n_evals += getattr(walker, "n_evals", 0)
            else:
                res = walker.run_many(items, self.log_likelihood_many, self.prior_transform_many)
        else:
            res = [func(it) for it in items]
        if callback is not None:
            for r in res:
                callback(r)
        return res


class _MasterQueue:
    """Rank 0's side of a queue in the master / worker form: hand the queue to the ranks that wait (``GPUPool.wait``), then walk this
    rank's shard and take part in the all-gather like everybody else (``parallel.ShardedQueue.run``)."""

    def __init__(self, pool, queue):
        self.pool, self.queue = pool, queue
        self.engine, self.constraints = queue.engine, queue.constraints

    def run(self, live, u0, loglstar, keys, walks, table=None):
        from .parallel import send_command
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        n = len(u0)
        per_chain = np.ndim(walks) > 0
        arrays = {"live": np.ascontiguousarray(live, dtype=np.float64), "u0": u0,
                  "loglstar": np.ascontiguousarray(np.broadcast_to(np.asarray(loglstar, dtype=np.float64), (n,))),
                  "keys": np.ascontiguousarray(keys, dtype=np.uint64),
                  "walks": np.ascontiguousarray(walks, dtype=np.int32) if per_chain else np.array([int(walks)], dtype=np.int32),
                  "table": np.frombuffer(bytes(table), dtype=np.uint8).copy()}
        pool = self.pool
        send_command(pool._dist(), pool._grp(), {"op": "walk", "per_chain": bool(per_chain)}, arrays, pool._rank_device())
        return self.queue.run(arrays["live"], u0, arrays["loglstar"], arrays["keys"], walks, table=table)
