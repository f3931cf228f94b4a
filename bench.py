#!/usr/bin/env python
"""Benchmark of the MI355X EM light-curve log-likelihood path.

Metric (BASELINE.json): log-likelihood evals/sec (Bu2019lm, AT2017gfo filters).
A "step" = one pass of the hot path over one batch of 4096 live points per GPU
(BASELINE config 2: Bu2019lm-shaped SVD surrogate NP=4, 2048 hidden, 10 coefficients,
211-point grid; 6 AT2017gfo filters with 13/19/20/18/15/14 epochs, one upper limit,
sigma_sys = 1 mag): theta[B, 6] resident in HBM -> logL[B] in HBM.  With N > 1 every rank
evaluates its own 4096-point shard (no data-path collective inside the evaluation) and the
shards' logL are exchanged with ONE RCCL all-gather per step (weak scaling).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` prices the log-likelihood kernel (em_logl: surrogate
MLP on the f32 MFMA pipe, SVD reconstruction, interpolation and likelihood terms in one launch)
against the dense fp32 MFMA peak with the ALGORITHMIC flop count of SURVEY.md section 8d
(sum over filters of 2*NP*NH + 2*NH*NC + 2*NC*NT = 369 384 flop/eval); its duration is
measured with HIP events on the launch stream inside the timed region.  `cpu_baseline`
times the CPU oracle (a port of the reference's per-sample Python calling pattern,
scipy.stats included) on one host core over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, dense f32 MFMA
BATCH_PER_GPU = 4096
N_THETA_SETS = 8


def _measured_traffic():
    """HBM bytes per launch from the committed PMC passes (profiles/r01_hbm_traffic.json)."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh)["bytes_per_launch"]
    except Exception:
        return None


TRAFFIC = _measured_traffic()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="live points per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("NMMA_BENCH_FORCE_DIST") == "1"   # (1-rank exercise of the exchange path)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    from nmma_amd import synthetic as syn
    from tests import cases
    from tests.helpers import engine_from_case

    case = cases.case_c2_default()          # model + photometry of BASELINE config 2
    eng = engine_from_case(case, device=local_rank)
    B = args.batch
    dev = torch.device(f"cuda:{local_rank}")
    thetas = [torch.as_tensor(syn.draw_theta(1000 + 97 * rank + i, B, case["names"])[1], device=dev)
              for i in range(N_THETA_SETS)]
    out = torch.empty(B, dtype=torch.float64, device=dev)
    gathered = torch.empty(world * B, dtype=torch.float64, device=dev) if use_dist else None

    def step(i):
        eng.loglike(thetas[i % N_THETA_SETS], out=out)
        if use_dist:
            # blocking form on purpose: an overlapped variant (second stream + events, or async_op) costs ~60 us of
            # host work per step in torch.distributed -- more than the 32 us kernel it would hide the collective behind
            dist.all_gather_into_tensor(gathered, out)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    # HIP events on the launch stream inside the timed region: with one GPU every second group of 8
    # back-to-back launches is bracketed by one event pair (a pair around a single ~35 us launch over-reads
    # by the dispatch latency behind the start event); with a collective between launches, single launches.
    os.environ["NMMA_PROFILE_GROUP"] = "8" if world == 1 else "1"
    os.environ["NMMA_PROFILE_STRIDE"] = "2" if world == 1 else "4"
    eng.profile_begin(args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the numbers we just timed are real likelihood values
    last = out.cpu().numpy()
    assert np.all(np.isfinite(last)) and np.all(last < 0)

    if rank == 0:
        geom = eng.last_launch_geometry()
        evals = world * B * args.steps
        fused_ms = prof["fused_ms_total"] / max(1, prof["n_launches"])
        achieved = eng.flops_per_eval * B / (fused_ms * 1e-3) / 1e12 if fused_ms > 0 else None
        line = {
            "metric": "log-likelihood evals/sec (Bu2019lm, AT2017gfo filters)",
            "value": evals / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 MLP + f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: Bu2019lm SVD surrogate (NP=4, NH=2048, NC=10, NT=211), "
                                   "AT2017gfo 6-filter synthetic photometry (99 epochs, 1 upper limit), "
                                   f"batch={B} live points per GPU, sigma_sys=1, detection_limit=inf",
                       "batch_per_gpu": B, "global_batch": world * B,
                       "exchange": "RCCL all_gather of logL per step" if world > 1 else "none",
                       "launch": geom},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": (achieved / PEAK_FP32_MFMA_TFLOPS) if achieved else None,
                         "traffic": TRAFFIC, "kernel": "em_logl", "kernel_ms": fused_ms,
                         "kernel_launches_timed": prof["n_launches"], "flops_per_eval": eng.flops_per_eval,
                         "traffic_unit": "bytes/launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, "
                                         "rocprofv3 --pmc, profiles/r01_*)"},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(case, args.cpu_seconds)
            line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(case, budget_s):
    """The CPU oracle in the reference's calling pattern (one parameter vector per call,
    Python loop over filters, numpy.interp, scipy.stats.truncnorm/norm) on ONE core."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")     # the reference forces this (joint/main.py:2)
    import numpy as np
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:
        limiter = None
    from nmma_amd import synthetic as syn
    from tests.helpers import oracle_from_case
    lik = oracle_from_case(case, use_scipy=True)
    names, theta = syn.draw_theta(555, 4096, case["names"])
    rows = [dict(zip(names, (float(v) for v in r))) for r in theta]
    for r in rows[:20]:
        lik.log_likelihood(r)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        lik.log_likelihood(rows[n % len(rows)])
        n += 1
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    return {"value": n / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"{n} evaluations cycling through the 4096 live points of the same workload, one parameter vector per call "
                      f"({dt:.1f} s, numpy fp32 MLP + scipy.stats as in the reference)"}


if __name__ == "__main__":
    main()
