"""The N > 1 code of bench.py and the single-process multi-device evaluator, exercised on the ONE GPU a test box has:
bench.py launches its own ranks (torch.distributed.run as a child process) which share device 0 and exchange logL over gloo
(NMMA_BENCH_SHARE_GPU=1; the line says so in config.exchange).  What this covers: launcher, sharding (even and ragged), both
scaling modes in one invocation, the JSON contract.  What it cannot cover: RCCL over xGMI (no scaling number is claimed here)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, **env_extra):
    env = dict(os.environ, NMMA_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    env.update(env_extra)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", *flags],
                          capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_both_scaling_modes(scaling):
    line = _bench("--gpus", "2", "--scaling", scaling)
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 2 and line["scaling"] == scaling
    assert line["metric"].startswith("log-likelihood evals/sec") and line["unit"] == "evals/s" and line["higher_is_better"] is True
    want = {"weak": (4096, 8192), "strong": (2048, 4096)}
    assert (line["config"]["batch_per_gpu"], line["config"]["global_batch"]) == want[scaling]
    assert "TEST MODE" in line["config"]["exchange"]
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert line["value"] == pytest.approx(line["config"]["global_batch"] / (line["ms_per_step"] * 1e-3), rel=1e-9)
    other = line["other_scaling"]
    o = "strong" if scaling == "weak" else "weak"
    assert other["scaling"] == o and (other["batch_per_gpu"], other["global_batch"]) == want[o]
    assert np.isfinite(other["value"]) and other["value"] > 0
    assert line["roofline"]["kernel"] == "em_logl" and 0 < line["roofline"]["frac"] < 1


def test_bench_ragged_strong_shards():
    """4097 rows over 2 ranks: 2049 + 2048, the short shard padded to the 2049-row all-gather slot."""
    line = _bench("--gpus", "2", "--scaling", "strong", "--batch", "4097")
    assert line["config"]["global_batch"] == 4097 and line["config"]["batch_per_gpu"] == 2049
    assert np.isfinite(line["value"]) and line["value"] > 0


@pytest.mark.parametrize("batch", [512, 1])
def test_bench_small_batches(batch):
    line = _bench("--batch", str(batch))
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == batch and "other_scaling" not in line
    assert np.isfinite(line["value"]) and line["value"] > 0


@pytest.mark.parametrize("blocking", ["0", "1", "stream"])
def test_bench_rccl_exchange_with_one_rank(blocking):
    """The RCCL path of bench.py on the one GPU of the box: process group `nccl` (= RCCL) of world size 1, the all-gather of logL
    after every evaluation -- on a second stream, overlapping the next kernel (the form an 8-GPU run takes), and blocking on the
    compute stream -- plus the barrier / max-over-ranks timing.  No scaling number; the code an N-GPU run executes has run."""
    # ("stream": the two-stream pipeline issued step by step; "0": the same K steps of a region replayed as one captured HIP graph)
    line = _bench("--gpus", "1", "--repeats", "3", NMMA_BENCH_SHARE_GPU="0", NMMA_BENCH_FORCE_DIST="1", NMMA_BENCH_BLOCKING="1" if blocking == "1" else "0",
                  NMMA_BENCH_NO_GRAPH="1" if blocking == "stream" else "0",
                  HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["warmup"] == 2 and line["repeats"] == 3
    ex = line["config"]["exchange"]
    assert ex.startswith("RCCL all_gather of logL per step") and ("pipelined" in ex) == (blocking != "1")
    assert ("captured HIP graph" in ex) == (blocking == "0"), ex
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert line["value"] == pytest.approx(4096 / (line["ms_per_step"] * 1e-3), rel=1e-9)
    assert line["roofline"]["kernel"] == "em_logl" and 0 < line["roofline"]["frac"] < 1


def test_bench_line_reports_the_median_region():
    line = _bench("--repeats", "7")
    assert line["repeats"] == 7 and line["spread_pct"] >= 0
    assert line["ms_per_step_best"] <= line["ms_per_step"] <= line["ms_per_step_worst"]


def test_multi_device_evaluator_in_one_process():
    """Single process, several engines (here: three on device 0) -- rows split, launched on per-engine streams, gathered:
    the same bits as one engine evaluating the whole batch."""
    import torch
    from nmma_amd import synthetic as syn
    from nmma_amd.engine import EMEngine
    from nmma_amd.parallel import MultiDeviceEvaluator
    case = syn.config2_case()
    _, theta = syn.draw_theta(99, 1000, case["names"])
    one = EMEngine.from_case(case)
    want = one.loglike(torch.as_tensor(theta, device="cuda:0")).cpu().numpy()
    ev = MultiDeviceEvaluator(lambda d: EMEngine.from_case(case, device=d), [0, 0, 0])
    for src in (theta, torch.as_tensor(theta, device="cuda:0")):
        got = ev.evaluate(src)
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(ev.evaluate(theta[:2]).cpu().numpy(), want[:2])      # fewer rows than engines
    ev.close(); one.close()
