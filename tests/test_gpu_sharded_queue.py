"""A sampler queue sharded over RANKS with real engines (the reference spreads a queue's chains over its MPI
ranks, nmma/core/mpi_setup.py:651-667, :679-683): ``parallel.ShardedQueue(engine=...)`` -- every rank walks its contiguous shard with
its own ``EMEngine``, the library packs the shard's records on the device (``nmma_walk_queue::records_dev``), ONE all-gather exchanges
them, one download -- must reproduce the single-device queue's records bit for bit.  On the one GPU of a test box: a process group of
ONE rank over ``nccl`` (= RCCL: the collective an 8-GPU run issues, on device buffers) and two / three ranks sharing the GPU over
``gloo``.  The checks themselves live in tests/sharded_queue_worker.py (every rank asserts)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "sharded_queue_worker.py")


def _env():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def test_sharded_queue_with_a_real_engine_over_rccl_world_size_one():
    proc = subprocess.run([sys.executable, WORKER, "nccl"], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    assert "OK 0 " in proc.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_queue_ranks_sharing_the_gpu_reproduce_the_single_device_queue(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), WORKER, "gloo"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=_env(), cwd=ROOT)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    for r in range(world):
        assert f"OK {r} " in proc.stdout, proc.stdout[-2000:]


def test_bench_two_ranks_prints_the_sharded_queue_block():
    """``bench.py --gpus 2`` (ranks sharing the GPU, gloo): the N > 1 line carries ``device_walk_queue_sharded`` -- weak and strong."""
    import json
    env = dict(_env(), NMMA_BENCH_SHARE_GPU="1")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--repeats", "3",
                           "--clock-warmup-steps", "64", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    blk = json.loads(lines[0])["device_walk_queue_sharded"]
    assert "error" not in blk, blk
    assert blk["weak"]["chains"] == 8192 and blk["weak"]["chains_per_rank"] == 4096 and blk["strong"]["chains"] == 4096
    for mode in ("weak", "strong"):
        assert blk[mode]["queue_ms"] > 0 and blk[mode]["evals_per_s"] > 0 and blk[mode]["walks"] == 100
    assert blk["allgather_us_per_call"] > 0 and "TEST MODE" in blk["exchange"]


def test_bench_one_rank_over_rccl_prints_the_sharded_queue_block():
    import json
    env = dict(_env(), NMMA_BENCH_FORCE_DIST="1")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--repeats", "3",
                           "--clock-warmup-steps", "64", "--no-cpu-baseline", "--sustained-seconds", "0"], capture_output=True, text=True, timeout=900,
                          env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    blk = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])["device_walk_queue_sharded"]
    assert "error" not in blk, blk
    assert blk["weak"]["chains"] == 4096 and blk["exchange"].startswith("RCCL")
