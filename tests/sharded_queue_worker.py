"""Worker of tests/test_gpu_sharded_queue.py: ONE rank of a sampler queue sharded over ranks (``parallel.ShardedQueue`` with a real
``EMEngine``).  Started by ``torch.distributed.run`` (several ranks sharing device 0, backend gloo) or directly (one rank, backend
``nccl`` = RCCL).  Every rank also walks the WHOLE queue on its own engine -- the single-device reference -- and compares: the
sharded records must be the single-device records bit for bit.  Prints ``OK <rank> <cases>`` on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    backend = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "RANK" not in os.environ:            # a process group of this one rank
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    from nmma_amd import sampler as smp
    from nmma_amd import synthetic as syn
    from nmma_amd.core.base import Constraint
    from nmma_amd.parallel import ShardedQueue
    from nmma_amd.pool import GPUPool
    from tests.helpers import UniformPrior, plugin_from_case

    def same(a, b):
        return np.array_equal(a, b, equal_nan=True)

    case = syn.config2_case()
    _, _, lik = plugin_from_case(case)
    names = case["names"]
    th = syn.draw_theta(3, 20000, names)[1]
    pri = {k: UniformPrior(float(a), float(b)) for k, a, b in zip(names, th.min(axis=0), th.max(axis=0))}
    pt = smp.BatchedPriorTransform(pri, names)
    eng = lik.sub_model.engine(names)
    w = smp.EnsembleWalkSampler(ndim=len(names), periodic=[1], reflective=[2], walks=9)
    table = smp.device_prior_table(pri, names, w.periodic, w.reflective)
    rng = np.random.default_rng(2024)            # (the same stream on every rank)
    n_live = 700
    live = rng.uniform(0.3, 0.7, (n_live, len(names)))
    l_live = eng.loglike(np.ascontiguousarray(pt(live)))
    queue = ShardedQueue(engine=eng, table=table)
    assert queue.world == world and queue.rank == rank
    done = 0
    # ragged splits (1001 = 3 x 333 + 2, 5 chains over up to 3 ranks, fewer chains than ranks), one round of 16-sample tiles exactly,
    # beyond it (32-sample tiles), equal and per-chain walk lengths, chains that can never accept (fresh prior draws)
    for n in (1001, 5, 1, 4096, 4500, 0):
        u0 = live[rng.integers(0, n_live, n)].copy() if n else np.empty((0, len(names)))
        bound = np.full(n, np.quantile(l_live, 0.3))
        bound[::50] = np.inf
        keys = rng.integers(1, 2 ** 62, n).astype(np.uint64)
        for steps in (7, (2 + np.arange(n) % 6).astype(np.int32)):
            if n == 0 and np.ndim(steps):
                continue
            got = queue.run(live, u0, bound, keys, steps)
            if n:
                want = eng.walk_queue(table, live, u0, bound, keys, steps)
            else:
                want = (np.empty((0, len(names))), np.empty((0, len(names))), np.empty(0), np.empty((0, 4), dtype=np.int32))
            for a, b in zip(got, want):
                assert a.shape == b.shape and a.dtype == b.dtype and same(a, b), (rank, n, np.ndim(steps))
            if n:
                assert np.all(got[3][::50, 0] == 0)          # never moved: came back as fresh draws
                assert queue.last_gpu_ms >= 0 and queue.last_collect_ms > 0
            done += 1

    # the same through the pool a driver holds: GPUPool(group=True).map(walker.sample, queue) on every rank
    pool_one = GPUPool(lik, queue_size=512, names=names, prior_transform_many=pt, priors=pri)
    pool_all = GPUPool(lik, queue_size=512, names=names, prior_transform_many=pt, priors=pri, group=True)
    nq = 333
    seeds = np.arange(77, 77 + nq)

    class _NS:
        live_u = live

    def args():
        return w.prepare_sampler(loglstar=float(np.quantile(l_live, 0.3)), points=live[:nq].copy(), axes=None, seeds=seeds, prior_transform=pt,
                                 loglikelihood=None, nested_sampler=_NS)
    one, many = pool_one.map(w.sample, args()), pool_all.map(w.sample, args())
    assert len(one) == len(many) == nq
    for a, b in zip(one, many):
        assert same(a[0], b[0]) and same(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]
    # ... and a batch of log_likelihood calls: row shards + ONE all-gather
    thetas = np.ascontiguousarray(pt(rng.uniform(0.2, 0.8, (257, len(names)))))
    assert same(pool_one.log_likelihood_many(thetas), pool_all.log_likelihood_many(thetas))
    assert pool_one.log_likelihood(thetas[0]) == pool_all.log_likelihood(thetas[0])        # (one row: every rank but the first holds an empty shard)
    done += 2

    # a constrained prior set: the likelihood's lowered Constraint program travels with every rank's shard
    _, _, lik_c = plugin_from_case(case)
    lik_c.constraints["log10_mej_dyn"] = Constraint(minimum=-2.6, maximum=-1.3, name="log10_mej_dyn")
    one_c = GPUPool(lik_c, queue_size=512, names=names, prior_transform_many=pt, priors=pri)
    all_c = GPUPool(lik_c, queue_size=512, names=names, prior_transform_many=pt, priors=pri, group=True)
    ref_c, got_c = one_c.map(w.sample, args()), all_c.map(w.sample, args())
    assert all_c._rank_queue is not None and all_c._rank_queue.constraints is not None
    assert same(got_c.u, ref_c.u) and same(got_c.logl, ref_c.logl) and np.array_equal(got_c.ncall, ref_c.ncall)
    assert not same(got_c.u, one.u)              # (the constraint bites)
    done += 1
    # a whole nested-sampling run, SPMD: every rank runs the same sampler on the same seed, the pool shards every queue over the ranks
    # -- the run has to be THE single-device run (same iterations, evidence, weights) on every rank
    from tests.test_gpu_nested_sampling import nested_sampling
    host_ll = lambda v: lik.log_likelihood_batch(np.ascontiguousarray(v), names)
    runs = []
    for pool_r in (GPUPool(lik, queue_size=384, names=names, prior_transform_many=pt, priors=pri),
                   GPUPool(lik, queue_size=384, names=names, prior_transform_many=pt, priors=pri, group=True)):
        walker = smp.EnsembleWalkSampler(ndim=len(names), naccept=10, walks=20, maxmcmc=500)
        runs.append(nested_sampling(pool_r, walker, pt, host_ll, len(names), 300, 384, seed=13, dlogz=0.5, max_iter=20000))
    a, b = runs
    assert a["niter"] == b["niter"] and a["ncall"] == b["ncall"] and a["logz"] == b["logz"] and a["niter"] > 300 * 5
    assert np.array_equal(a["weights"], b["weights"]) and np.array_equal(a["samples"], b["samples"])
    z = torch.tensor([b["logz"], float(b["niter"])], dtype=torch.float64)
    zs = [torch.empty_like(z) for _ in range(world)]
    if backend == "nccl":
        z, zs = z.cuda(), [t.cuda() for t in zs]
    dist.all_gather(zs, z)
    assert all(torch.equal(t, z) for t in zs)        # the same run on every rank
    done += 1
    # the reference's own structure (core/mpi_setup.py:651-667): ONLY rank 0 runs the sampler, the other ranks wait in the pool and
    # serve its map calls -- `with POOL() as pool: if pool.is_master(): ...` -- and rank 0's run is again THE single-device run
    dist.barrier()
    walker_mw = smp.EnsembleWalkSampler(ndim=len(names), naccept=10, walks=20, maxmcmc=500)
    result_mw = None
    with GPUPool(lik, queue_size=384, names=names, prior_transform_many=pt, priors=pri, group=True, master_worker=True) as pool_mw:
        assert pool_mw.is_master() == (rank == 0) and pool_mw.is_worker() == (rank != 0)
        if pool_mw.is_master():
            # (a likelihood batch first: row shards + one all-gather, served by the waiting ranks)
            assert same(pool_mw.log_likelihood_many(thetas), pool_one.log_likelihood_many(thetas))
            assert pool_mw.log_likelihood(thetas[1]) == pool_one.log_likelihood(thetas[1])
            result_mw = nested_sampling(pool_mw, walker_mw, pt, host_ll, len(names), 300, 384, seed=13, dlogz=0.5, max_iter=20000)
    if rank == 0:
        assert result_mw["niter"] == a["niter"] and result_mw["ncall"] == a["ncall"] and result_mw["logz"] == a["logz"]
        assert np.array_equal(result_mw["weights"], a["weights"]) and np.array_equal(result_mw["samples"], a["samples"])
    else:
        assert result_mw is None and pool_mw._closed
    done += 1
    dist.barrier()
    print(f"OK {rank} {done}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
