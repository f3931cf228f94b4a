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
