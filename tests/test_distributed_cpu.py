"""world_size-2 gloo test of the N>1 path (batch sharding + one all-gather), on CPU."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["NMMA_ROOT"])
from nmma_amd.parallel import ShardedEvaluator, shard_bounds
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = int(os.environ["NMMA_N"])
theta = torch.arange(n * 3, dtype=torch.float64).reshape(n, 3)
calls = []
def local_fn(shard):
    calls.append(shard.shape[0])
    return (shard ** 2).sum(dim=1) * -1.0          # stands in for the per-rank GPU evaluation
out = ShardedEvaluator(local_fn).evaluate(theta)
want = (theta ** 2).sum(dim=1) * -1.0
assert torch.equal(out, want), (rank, out, want)
lo, hi = shard_bounds(n, world, rank)
assert calls == [hi - lo]
if rank == 0:
    print("OK", n, world)
dist.destroy_process_group()
'''


def _run(n, world=2):
    env = dict(os.environ, NMMA_ROOT=ROOT, NMMA_N=str(n), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + (os.getpid() + n) % 400), "-c", WORKER]
    # torch.distributed.run has no -c: write the worker to a temp file instead
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as fh:
        fh.write(WORKER)
        path = fh.name
    cmd = cmd[:-2] + [path]
    try:
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    finally:
        os.unlink(path)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert f"OK {n} {world}" in proc.stdout


def test_sharded_all_gather_even():
    _run(64)


def test_sharded_all_gather_ragged():
    _run(33)


def test_sharded_all_gather_ragged_three_ranks():
    """10 rows over 3 ranks: shards of 4, 3 and 3 rows, the short ones padded to the 4-row slot of the all-gather."""
    _run(10, world=3)


def test_sharded_all_gather_fewer_rows_than_ranks():
    """2 rows over 3 ranks: the last rank owns an empty shard and still takes part in the collective."""
    _run(2, world=3)


def test_shard_bounds_partition_every_batch():
    from nmma_amd.parallel import shard_bounds
    for n in (0, 1, 2, 7, 33, 4096, 4097):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


# ---------------------------------------------------------------------------------------------------------------------------------
# The sampler queue over ranks / devices (SURVEY section 8 f1 + e; core/mpi_setup.py:651-667, :679-683): split, walk, concatenate.
# ---------------------------------------------------------------------------------------------------------------------------------
QUEUE_WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["NMMA_ROOT"])
from nmma_amd.parallel import ShardedQueue, shard_bounds
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, D = int(os.environ["NMMA_N"]), 5
rng = np.random.default_rng(1)
live = rng.uniform(size=(40, D)); u0 = rng.uniform(size=(n, D)); star = rng.normal(size=n)
keys = rng.integers(1, 2 ** 62, n).astype(np.uint64); walks = (3 + np.arange(n) % 5).astype(np.int32)
def walk(live, u0, star, keys, walks):            # stands in for engine.walk_queue: a function of each chain's own record only
    k = (keys % np.uint64(1000)).astype(float)[:, None]
    u = u0 + 1e-3 * k + live[0]
    walks = np.broadcast_to(np.asarray(walks), (len(u0),))
    counts = np.stack([walks, walks + 1, (keys % np.uint64(7)).astype(np.int32), walks * 2], axis=1).astype(np.int32)
    return u, 2.0 * u, star - k[:, 0], counts
calls = []
def local_fn(*a):
    calls.append(len(a[1]))
    return walk(*a)
got = ShardedQueue(local_fn).run(live, u0, star, keys, walks)
want = walk(live, u0, star, keys, walks)
for g, w in zip(got, want):
    assert g.dtype == w.dtype and np.array_equal(g, w), (rank, g[:3], w[:3])
lo, hi = shard_bounds(n, world, rank)
assert calls == ([hi - lo] if hi > lo else [])
same = ShardedQueue(local_fn).run(live, u0, star, keys, 4)           # one walk length for all chains
assert np.array_equal(same[3], walk(live, u0, star, keys, np.full(n, 4, dtype=np.int32))[3])
if rank == 0:
    print("OK", n, world)
dist.destroy_process_group()
'''


def _run_worker(src, n, world):
    import tempfile
    env = dict(os.environ, NMMA_ROOT=ROOT, NMMA_N=str(n), MASTER_ADDR="127.0.0.1")
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as fh:
        fh.write(src)
        path = fh.name
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + (os.getpid() + 7 * n + world) % 400), path]
    try:
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    finally:
        os.unlink(path)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert f"OK {n} {world}" in proc.stdout


def test_sharded_queue_two_ranks_ragged():
    _run_worker(QUEUE_WORKER, 33, 2)


def test_sharded_queue_three_ranks_fewer_chains_than_ranks():
    _run_worker(QUEUE_WORKER, 2, 3)


def test_queue_sharded_over_engines_in_one_process():
    """``run_many_device`` with a LIST of (engine, constraint program) shards -- what ``GPUPool(devices=[...])`` passes: contiguous
    balanced shards, live points replicated, every begin before the first end, records concatenated in queue order.  Engines are
    stand-ins (the GPU suite runs the same path on real ones: tests/test_gpu_walk_queue.py)."""
    from nmma_amd import sampler as smp
    from tests.helpers import UniformPrior
    names = ["a", "b", "c"]
    pri = {k: UniformPrior(0.0, 1.0) for k in names}
    log = []

    class FakeEngine:
        def __init__(self, tag):
            self.tag = tag

        def walk_queue_begin(self, table, live, u0, loglstar, keys, walks, constraints=None):
            log.append(("begin", self.tag, len(u0)))
            k = (np.asarray(keys) % np.uint64(1000)).astype(float)
            wl = np.broadcast_to(np.asarray(walks), (len(u0),))
            counts = np.stack([wl, wl * 0, wl * 0, wl + 1], axis=1).astype(np.int32)
            assert constraints == f"prog{self.tag}"
            return (np.asarray(u0) + k[:, None], np.asarray(u0) * 2.0, np.asarray(loglstar) + k, counts)

        def walk_queue_end(self, token):
            log.append(("end", self.tag))
            return token

        def walk_queue(self, *a, **kw):
            return self.walk_queue_end(self.walk_queue_begin(*a, **kw))

    rng = np.random.default_rng(2)
    for n, n_eng in ((10, 3), (2, 3), (4096, 8), (7, 1)):
        del log[:]
        live = rng.uniform(size=(max(n, 50), 3))
        w = smp.EnsembleWalkSampler(ndim=3, walks=6, naccept=3)

        class _NS:
            live_u = live
        seeds = rng.integers(1, 2 ** 62, n)
        batch = w.prepare_sampler(loglstar=-3.0, points=live[:n].copy(), axes=None, seeds=seeds, prior_transform=lambda u: u, loglikelihood=None,
                                  nested_sampler=_NS)
        shards = [(FakeEngine(i), f"prog{i}") for i in range(n_eng)]
        got = w.run_many_device(batch, None, pri, names, engine=shards)
        ref = w.run_many_device(batch, None, pri, names, engine=FakeEngine(0), constraints="prog0")
        assert np.array_equal(got.u, ref.u) and np.array_equal(got.logl, ref.logl) and np.array_equal(got.ncall, ref.ncall)
        if n_eng > 1:
            from nmma_amd.parallel import shard_bounds
            sizes = [hi - lo for lo, hi in (shard_bounds(n, n_eng, r) for r in range(n_eng)) if hi > lo]
            begins = [e for e in log if e[0] == "begin"][:len(sizes)]
            assert [b[2] for b in begins] == sizes
            first_end = next(i for i, e in enumerate(log) if e[0] == "end")
            assert all(e[0] == "begin" for e in log[:first_end]) and first_end == len(sizes)      # every shard begun before any is collected


def test_record_packing_round_trips_bits():
    """``pack_records`` / ``unpack_records`` (the layout of ``nmma_walk_queue::records_dev``): NaN payloads, infinities and the full
    int32 range of the counters survive a byte-moving collective unchanged."""
    from nmma_amd.parallel import pack_records, unpack_records
    rng = np.random.default_rng(5)
    n, d = 37, 6
    u, v = rng.uniform(size=(n, d)), rng.normal(size=(n, d)) * 1e300
    logl = rng.normal(size=n)
    logl[3], logl[4] = np.nan, -np.inf
    counts = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(n, 4)).astype(np.int32)
    rows = pack_records(u, v, logl, counts)
    assert rows.shape == (n, 2 * d + 3) and rows.dtype == np.float64
    moved = np.frombuffer(rows.tobytes(), dtype=np.float64).reshape(rows.shape)       # (what a collective does: bytes)
    gu, gv, gl, gc = unpack_records(moved, d)
    assert np.array_equal(gu, u) and np.array_equal(gv, v) and np.array_equal(gl, logl, equal_nan=True) and np.array_equal(gc, counts)
    assert gc.dtype == np.int32
    eu, ev, el, ec = unpack_records(np.empty((0, 2 * d + 3)), d)
    assert eu.shape == (0, d) and el.shape == (0,) and ec.shape == (0, 4)


def test_sharded_queue_takes_either_a_function_or_an_engine():
    import pytest
    from nmma_amd.parallel import ShardedQueue
    with pytest.raises(ValueError):
        ShardedQueue()
    with pytest.raises(ValueError):
        ShardedQueue(local_fn=lambda *a: None, engine=object())


COMMAND_WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["NMMA_ROOT"])
from nmma_amd.parallel import send_command, recv_command
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(3)                        # (the same stream on every rank: the workers know what to expect)
want = {"live": rng.uniform(size=(37, 6)), "keys": rng.integers(1, 2 ** 62, 11).astype(np.uint64),
        "walks": np.array([7], dtype=np.int32), "table": np.frombuffer(rng.bytes(40 * 6), dtype=np.uint8).copy(), "empty": np.empty((0, 6))}
want["live"][3, 2] = np.nan
for round_ in range(3):
    if rank == 0:
        header, got = send_command(dist, None, {"op": "walk", "per_chain": False, "round": round_}, want)
    else:
        header, got = recv_command(dist, None)
    assert header == {"op": "walk", "per_chain": False, "round": round_}, header
    assert list(got) == list(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape and np.array_equal(got[k], want[k], equal_nan=True), k
header, got = send_command(dist, None, {"op": "close"}, {}) if rank == 0 else recv_command(dist, None)
assert header == {"op": "close"} and got == {}
dist.barrier()
if rank == 0:
    print("OK", 0, world)
dist.destroy_process_group()
'''


def test_master_worker_commands_round_trip_over_gloo():
    """``send_command`` / ``recv_command`` (the master / worker form of ``GPUPool``: rank 0 hands a queue to the ranks that wait): header
    and arrays -- NaNs, uint64 keys, raw table bytes, an empty array -- arrive bit for bit on every other rank, for world sizes 2 and 3."""
    import tempfile
    for world in (2, 3):
        with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as fh:
            fh.write(COMMAND_WORKER)
            path = fh.name
        env = dict(os.environ, NMMA_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(29900 + (os.getpid() + world) % 90), path]
        try:
            proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        finally:
            os.unlink(path)
        assert proc.returncode == 0, proc.stdout + proc.stderr
        assert f"OK 0 {world}" in proc.stdout
