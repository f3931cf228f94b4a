#!/usr/bin/env python
"""Benchmark of the MI355X EM light-curve log-likelihood path.

Metric (BASELINE.json): log-likelihood evals/sec (Bu2019lm, AT2017gfo filters).
A "step" = one pass of the hot path over one batch of 4096 live points per GPU
(BASELINE config 2: Bu2019lm-shaped SVD surrogate NP=4, 2048 hidden, 10 coefficients,
211-point grid; 6 AT2017gfo filters with 13/19/20/18/15/14 epochs, one upper limit,
sigma_sys = 1 mag): theta[B, 6] resident in HBM -> logL[B] in HBM.  With N > 1 every rank
evaluates its own 4096-point shard (no data-path collective inside the evaluation) and the
shards' logL are exchanged with ONE RCCL all-gather per step (weak scaling).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Without a launcher (RANK unset) and N > 1 the script starts ``torch.distributed.run`` itself as a CHILD process --
before anything here touches the GPU -- and relays the child's JSON line.  ``--scaling weak`` (default) keeps 4096
live points per GPU; ``--scaling strong`` splits ONE 4096-point batch over the ranks (north_star's 8-GPU target).

Before the W warmup steps ``--clock-warmup-steps`` (1536) untimed steps let the GPU's clocks settle (a fresh process runs its first
~900 launches at ramping clocks).  The K-step region (W untimed steps first, each region bracketed by barrier + synchronize) is timed ``--repeats`` times and
`value` / `ms_per_step` are the MEDIAN region's (`spread_pct` = the 10-90 percentile spread over the repeats, relative to it).
With N > 1 over RCCL the K steps of a timed region -- evaluation on one stream, the all-gather of the previous step's log L on a second --
are replayed as ONE captured HIP graph (issued step by step the pipeline's event calls cost the host more than the kernel they overlap with;
NMMA_BENCH_NO_GRAPH=1 keeps the stream form, NMMA_BENCH_BLOCKING=1 one stream).
With N > 1 BOTH scaling modes are measured in the one invocation: the line's top level is the mode ``--scaling`` names, the
other one sits under ``other_scaling``.  Rank 0 prints ONE JSON line.  `roofline` prices the log-likelihood kernel (em_logl: surrogate
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


def _profiled_traffic():
    """HBM bytes per launch from the newest committed PMC passes (profiles/rNN_hbm_traffic.json); NOT measured in
    this run -- the line says so in roofline.traffic_source."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    for path in reversed(paths):
        try:
            with open(path) as fh:
                return json.load(fh)["bytes_per_launch"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


TRAFFIC, TRAFFIC_SOURCE = _profiled_traffic()


def _relaunch_under_torchrun(args):
    """--gpus N > 1 without a launcher: run N ranks as a child of this (GPU-untouched) process, relay its output."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=25, help="the K-step region is timed this many times; value = the median region")
    ap.add_argument("--clock-warmup-steps", type=int, default=1536,
                    help="untimed steps BEFORE the W warmup steps: the GPU's clocks need ~30 ms of continuous load (~900 launches) to settle")
    ap.add_argument("--sustained-seconds", type=float, default=5.0,
                    help="length of the `sustained` context leg (back-to-back steps on the same handle; 0 = skip)")
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="live points per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch live points per GPU; strong: one --batch-point batch split over the ranks")
    ap.add_argument("--cpu-worker", type=float, default=None, help=argparse.SUPPRESS)   # child of the all-cores baseline
    args = ap.parse_args()
    if args.cpu_worker is not None:
        return _cpu_worker(args.cpu_worker)
    if args.gpus > 1 and "RANK" not in os.environ:
        _relaunch_under_torchrun(args)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    # test mode (1-GPU boxes): all ranks on device 0, logL exchanged over gloo -- exercises the launcher, the sharding and
    # the JSON plumbing of the N > 1 path; the line says so in config.exchange
    share_gpu = os.environ.get("NMMA_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("NMMA_BENCH_FORCE_DIST") == "1"   # (1-rank exercise of the exchange path)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "RANK" not in os.environ:        # NMMA_BENCH_FORCE_DIST=1 without a launcher: a process group of this one rank
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    from nmma_amd import synthetic as syn
    from nmma_amd.engine import EMEngine

    case = syn.config2_case()               # model + photometry of BASELINE config 2
    if use_dist and not share_gpu and os.environ.get("NMMA_BENCH_BLOCKING") != "1":
        # the collective of one step overlaps the kernel of the next: leave LDS on every CU for RCCL's kernels (a ring of 2 item
        # slots instead of 4 costs the likelihood kernel 1.2 % and frees 57 KiB per CU)
        os.environ.setdefault("NMMA_EM_RING", "2")
    eng = EMEngine.from_case(case, device=local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    import gc

    def measure(scaling, cold=False):
        """W untimed + K timed steps of one scaling mode: (elapsed max over ranks, rows of this rank, global batch, pipelined, prof).
        cold: FIRST time one contract-shaped region -- W warm-up steps, K timed steps, nothing before them -- as the process's first
        load on the GPU (`value_cold`: what the driver's flags measure without the clock warm-up)."""
        if scaling == "strong":
            from nmma_amd.parallel import shard_bounds
            lo, hi = shard_bounds(args.batch, world, rank)
            B = hi - lo
        else:
            B = args.batch
        global_batch = args.batch if scaling == "strong" else world * args.batch
        thetas = [torch.as_tensor(syn.draw_theta(1000 + 97 * rank + i, B, case["names"])[1], device=dev)
                  for i in range(N_THETA_SETS)]
        out = torch.empty(B, dtype=torch.float64, device=dev)
        slot = -(-global_batch // world)         # equal-sized all-gather slots (ragged strong-scaling shards are padded)
        send = torch.zeros(slot, dtype=torch.float64, device=dev) if use_dist else None
        gathered = torch.empty(world * slot, dtype=torch.float64, device=dev) if use_dist else None

        # N > 1: the all-gather of step i runs on a second stream while the kernel of step i + 1 runs on the first (two logL
        # buffers; events order "kernel i -> gather i" and "gather i -> kernel i + 2").  Host cost per step is the same as the
        # blocking form (~38 us, tools/overlap_probe.py), but the GPU no longer serialises kernel + collective.
        # NMMA_BENCH_BLOCKING=1 keeps everything on one stream.
        # (decided from the GLOBAL split, so that every rank takes the same form: the forms issue different numbers of collectives)
        pipelined = use_dist and not share_gpu and global_batch % world == 0 and os.environ.get("NMMA_BENCH_BLOCKING") != "1"
        prev_stream = torch.cuda.current_stream()
        if pipelined:
            s_eval, s_coll = torch.cuda.Stream(), torch.cuda.Stream()
            outs = [out, torch.empty_like(out)]
            gathers = [gathered, torch.empty_like(gathered)]
            ev_eval = [torch.cuda.Event() for _ in range(2)]
            ev_coll = [torch.cuda.Event() for _ in range(2)]
            torch.cuda.synchronize()
            torch.cuda.set_stream(s_coll)            # torch.distributed enqueues on the current stream

        def step(i):
            if pipelined:
                b = i & 1
                if i >= 2:
                    s_eval.wait_event(ev_coll[b])    # the gather that read outs[b] two steps ago
                eng.loglike(thetas[i % N_THETA_SETS], out=outs[b], stream=s_eval)
                ev_eval[b].record(s_eval)
                s_coll.wait_event(ev_eval[b])
                dist.all_gather_into_tensor(gathers[b], outs[b])
                ev_coll[b].record(s_coll)
                return
            eng.loglike(thetas[i % N_THETA_SETS], out=out)
            if use_dist:
                if share_gpu:
                    send[:B] = out
                    parts = [torch.empty(slot, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(parts, send.cpu())
                elif slot == B:
                    dist.all_gather_into_tensor(gathered, out)
                else:
                    send[:B] = out
                    dist.all_gather_into_tensor(gathered, send)

        # A full collection of Python's garbage collector takes ~75 ms with torch imported (2 400 launches' worth) and fires
        # whenever enough container objects have been allocated: none inside the timed region
        gc.collect()
        gc.disable()
        # Clock warm-up (untimed, before the W warmup steps of the contract): a freshly started process runs its first ~900 launches
        # -- ~28 ms of continuous load -- at ramping clocks (the kernel takes 31.3 us at launch 0 and 27.6 us from launch ~900 on,
        # rocprofv3 trace in profiles/r04_bench_kernel_stats.csv); a sampler lives in the steady state, the driver's 20-step region
        # would not.
        done = 0                # steps issued so far (the pipelined form alternates its two buffers on this count)
        cold_elapsed = None
        if cold:
            for _ in range(args.warmup):
                step(done)
                done += 1
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(done, done + args.steps):
                step(i)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
                torch.cuda.synchronize()
            cold_elapsed = time.perf_counter() - t0
            done += args.steps
            if use_dist:
                t = torch.tensor([cold_elapsed], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                cold_elapsed = float(t.item())
        # N > 1, pipelined: the K steps of a timed region as ONE captured HIP graph.  Issued step by step the two-stream pipeline costs
        # the HOST 38 us per step -- four event record / wait calls at ~5.5 us each on top of the kernel launch (5.4 us) and the
        # collective's call (10.5 us): tools/experiments/step_cost_probe.py -- i.e. more than the 27.6 us kernel it is meant to overlap
        # with; in a graph the same dependencies are edges and the host issues one launch per region.  The graph holds exactly
        # args.steps steps (kernel on s_eval -> all-gather on s_coll, two log L buffers); capture failures fall back to the stream form.
        graph, graph_note, graph_last = None, None, None
        if pipelined and os.environ.get("NMMA_BENCH_NO_GRAPH") != "1":
            try:
                # (two stream-ordered steps first: whatever the library allocates on the first launch of a batch size -- the band split's
                #  workspace -- must exist before a capture, which forbids allocations)
                for _ in range(2):
                    step(done)
                    done += 1
                torch.cuda.synchronize()
                if use_dist:
                    dist.barrier()
                g = torch.cuda.CUDAGraph()
                # (events of their own: an event last recorded inside a capture must not be waited on by the stream-ordered steps)
                # (NB log L buffers: evaluation i + NB waits for the collective of step i.  With one rank over RCCL 2, 4 and 8 buffers give the
                #  same 35 us per step at K = 200 -- the collective's kernel beside the likelihood's costs the latter 4 us and the pair ~3 us
                #  of serialisation whatever the depth -- so two, as in the stream form; NMMA_BENCH_GRAPH_BUFFERS for an 8-GPU ring that
                #  needs more slack)
                NB = max(2, int(os.environ.get("NMMA_BENCH_GRAPH_BUFFERS", "2")))
                gev_eval = [torch.cuda.Event() for _ in range(NB)]
                gev_coll = [torch.cuda.Event() for _ in range(NB)]
                g_outs = [torch.empty_like(out) for _ in range(NB)]
                g_gathers = [torch.empty_like(gathered) for _ in range(NB)]
                with torch.cuda.graph(g, stream=s_eval):
                    for i in range(args.steps):
                        b = i % NB
                        if i >= NB:
                            s_eval.wait_event(gev_coll[b])
                        eng.loglike(thetas[i % N_THETA_SETS], out=g_outs[b], stream=s_eval)
                        gev_eval[b].record(s_eval)
                        s_coll.wait_event(gev_eval[b])
                        with torch.cuda.stream(s_coll):
                            dist.all_gather_into_tensor(g_gathers[b], g_outs[b])
                        gev_coll[b].record(s_coll)
                    for b in range(min(NB, args.steps)):      # join: the capture ends on s_eval with every collective behind it
                        s_eval.wait_event(gev_coll[b])
                graph_last = g_outs[(args.steps - 1) % NB]
                torch.cuda.synchronize()
                graph = g
            except Exception as exc:      # noqa: BLE001
                graph, graph_note = None, f"{type(exc).__name__}: {exc}"[:200]
                torch.cuda.synchronize()
            # every rank must issue the same collectives in the same order: graph mode only if EVERY rank captured its graph
            agree = torch.tensor([1 if graph is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 0 and graph is not None:
                graph, graph_note = None, "another rank could not capture its graph"

        def region():
            """K steps: one graph launch, or K stream-ordered steps."""
            nonlocal done
            if graph is not None:
                graph.replay()
            else:
                for i in range(done, done + args.steps):
                    step(i)
            done += args.steps

        n_before_clock = done
        if graph is not None:
            for _ in range(-(-args.clock_warmup_steps // args.steps)):      # (the same count on every rank)
                region()
        else:
            for _ in range(args.clock_warmup_steps):        # (a fixed count: every rank issues the same collectives)
                step(done)
                done += 1
        torch.cuda.synchronize()
        n_clock = done - n_before_clock
        for _ in range(args.warmup):
            step(done)
            done += 1
        torch.cuda.synchronize()
        # HIP events on the launch stream inside the timed region: with one GPU every second group of 8
        # back-to-back launches is bracketed by one event pair (a pair around a single ~35 us launch over-reads
        # by the dispatch latency behind the start event); with a collective between launches, single launches.
        os.environ["NMMA_PROFILE_GROUP"] = os.environ.get("NMMA_BENCH_EVENT_GROUP", "8" if not use_dist else "1")
        os.environ["NMMA_PROFILE_STRIDE"] = os.environ.get("NMMA_BENCH_EVENT_STRIDE", "2" if not use_dist else "4")
        # The K-step region is timed R times (each one bracketed by barrier + synchronize on both sides); `value` comes from
        # the MEDIAN region: at the driver's flags the region is well under a millisecond and a single one carries the noise
        # of whatever the box did in that millisecond.
        times, prof = [], dict(fused_ms_total=0.0, n_launches=0)
        for _ in range(args.repeats):
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            if graph is None:
                eng.profile_begin(args.steps)
            t0 = time.perf_counter()
            region()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
                torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            if graph is None:
                p = eng.profile_end()
                prof["fused_ms_total"] += p["fused_ms_total"]
                prof["n_launches"] += p["n_launches"]
        if graph is not None:
            # (HIP events cannot bracket launches inside a captured graph: the kernel time of `roofline` comes from the same K steps
            #  issued stream by stream right after the timed regions -- same kernels, same collectives beside them; the line says so)
            for _ in range(min(args.repeats, 5)):
                if use_dist:
                    dist.barrier()
                torch.cuda.synchronize()
                eng.profile_begin(args.steps)
                for i in range(args.steps):
                    step(i)
                torch.cuda.synchronize()
                p = eng.profile_end()
                prof["fused_ms_total"] += p["fused_ms_total"]
                prof["n_launches"] += p["n_launches"]
        gc.enable()
        if pipelined:
            torch.cuda.set_stream(prev_stream)
        if use_dist:        # every region's time is the MAX over ranks
            t = torch.tensor(times, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            times = t.cpu().tolist()
        elapsed = float(np.median(times))
        spread = 100.0 * (float(np.percentile(times, 90)) - float(np.percentile(times, 10))) / elapsed if len(times) > 1 else 0.0
        n_done = args.steps if graph is not None else done      # (the profiling leg's steps count from 0)
        # sanity: the numbers we just timed are real likelihood values
        last = (outs[(n_done - 1) & 1] if pipelined else out).cpu().numpy()
        if graph is not None:
            glast = graph_last.cpu().numpy()
            assert np.all(np.isfinite(glast)) and np.all(glast < 0)
        assert np.all(np.isfinite(last)) and np.all(last < 0)
        return dict(elapsed=elapsed, B=B, global_batch=global_batch, pipelined=pipelined, graph=graph is not None, graph_note=graph_note, prof=prof,
                    geom=eng.last_launch_geometry(), n_clock=n_clock,
                    spread_pct=spread, best=min(times), worst=max(times), cold_elapsed=cold_elapsed, thetas=thetas, out=out)

    def exchange_label(m):
        if not use_dist:
            return "none"
        if share_gpu:
            return "gloo all_gather (TEST MODE: ranks share one GPU)"
        return ("RCCL all_gather of logL per step" + (", pipelined with the next evaluation" if m["pipelined"] else "")
                + (", the K steps of a region replayed as ONE captured HIP graph (kernel time of `roofline`: the same steps issued stream by stream)"
                   if m.get("graph") else (f" (graph capture failed: {m['graph_note']})" if m.get("graph_note") else "")))

    main_mode = measure(args.scaling, cold=True)
    # with more than one GPU the other scaling mode is measured in the same invocation and reported under "other_scaling"
    other = measure("strong" if args.scaling == "weak" else "weak") if world > 1 else None

    # the sampler queue sharded over the ranks (parallel.ShardedQueue: what the reference's task farm does with a queue's chains,
    # core/mpi_setup.py:651-667, :679-683) -- every rank takes part, rank 0 reports; context, never `value`
    sharded = None
    if use_dist:
        try:
            sharded = device_walk_queue_sharded(eng, case, syn, dist, world, rank, dev, share_gpu)
        except Exception as exc:      # noqa: BLE001  (a failure here must not cost the contract line; every rank fails alike or the barrier below reports)
            sharded = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        m = main_mode
        B, global_batch, prof = m["B"], m["global_batch"], m["prof"]
        evals = global_batch * args.steps
        fused_ms = prof["fused_ms_total"] / max(1, prof["n_launches"])
        achieved = eng.flops_per_eval * B / (fused_ms * 1e-3) / 1e12 if fused_ms > 0 else None
        line = {
            "metric": "log-likelihood evals/sec (Bu2019lm, AT2017gfo filters)",
            "value": evals / m["elapsed"], "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * m["elapsed"] / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32 MLP + f64",
            "data": "synthetic", "repeats": args.repeats, "spread_pct": m["spread_pct"], "clock_warmup_launches": m["n_clock"],
            # the contract-shaped single region (W warm-up steps, K timed steps) run as the process's FIRST load, before the clock warm-up
            "value_cold": evals / m["cold_elapsed"], "ms_per_step_cold": 1e3 * m["cold_elapsed"] / args.steps,
            "value_is": f"steady state: median of {args.repeats} timed K-step regions after {m['n_clock']} untimed clock-warm-up steps + W; "
                        "value_cold is the single W + K region of a fresh process; sustained.evals_per_s is >= 5 s of back-to-back steps",
            "ms_per_step_best": 1e3 * m["best"] / args.steps, "ms_per_step_worst": 1e3 * m["worst"] / args.steps,
            "config": {"workload": "BASELINE config 2: Bu2019lm SVD surrogate (NP=4, NH=2048, NC=10, NT=211), "
                                   "AT2017gfo 6-filter synthetic photometry (99 epochs, 1 upper limit), "
                                   f"batch={B} live points per GPU, sigma_sys=1, detection_limit=inf",
                       "batch_per_gpu": B, "global_batch": global_batch, "exchange": exchange_label(m), "launch": m["geom"]},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": (achieved / PEAK_FP32_MFMA_TFLOPS) if achieved else None,
                         "traffic": TRAFFIC, "traffic_source": (f"{TRAFFIC_SOURCE}: rocprofv3 --pmc passes of this command, "
                                                                "not collected in this run") if TRAFFIC_SOURCE else None,
                         "kernel": "em_logl", "kernel_ms": fused_ms,
                         "kernel_launches_timed": prof["n_launches"], "flops_per_eval": eng.flops_per_eval,
                         "traffic_unit": "bytes/launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)"},
        }
        if other is not None:
            o_scaling = "strong" if args.scaling == "weak" else "weak"
            o_ms = other["prof"]["fused_ms_total"] / max(1, other["prof"]["n_launches"])
            line["other_scaling"] = {"scaling": o_scaling, "value": other["global_batch"] * args.steps / other["elapsed"],
                                     "unit": "evals/s", "ms_per_step": 1e3 * other["elapsed"] / args.steps,
                                     "batch_per_gpu": other["B"], "global_batch": other["global_batch"],
                                     "exchange": exchange_label(other), "kernel_ms": o_ms}
        if sharded is not None:
            line["device_walk_queue_sharded"] = sharded
        if world == 1 and args.sustained_seconds > 0:
            try:
                line["sustained"] = sustained_leg(eng, m["thetas"], m["out"], B, args.sustained_seconds, line["value"], fused_ms)
            except Exception as exc:      # noqa: BLE001
                line["sustained"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_cpu_baseline and world == 1:      # (the profiled command lines pass --no-cpu-baseline: device launches only)
            # (context blocks: a failure in one of them must not cost the contract line)
            def context(key, fn):
                try:
                    line[key] = fn()
                except Exception as exc:      # noqa: BLE001
                    line[key] = {"error": f"{type(exc).__name__}: {exc}"}
            context("host_call_ms", lambda: host_call_ms(eng, case, syn))

            def walk_block():
                ms = device_walk_step_ms(eng, case, syn)
                return {"chains": 4096, "ms_per_mcmc_step": ms, "evals_per_s": 4096 / (ms * 1e-3),
                        "what": "lock-step ensemble walk on the device: likelihood -> accept + next proposal, two launches per step"}
            context("device_walk", walk_block)
            context("device_walk_queue", lambda: device_walk_queue(case, syn, line["device_walk"]["evals_per_s"]))
            line["cpu_baseline"] = cpu_baseline(case, args.cpu_seconds)
            line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
            context("cpu_baseline_all_cores", lambda: cpu_baseline_all_cores(args.cpu_seconds, line["cpu_baseline"]["value"]))
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def sustained_leg(eng, thetas, out, batch, seconds, value, kernel_ms_region, chunk=8192):
    """>= `seconds` of back-to-back steps on the same handle -- what a sampler's likelihood sees after the first second.  The timed
    regions of `value` are ~0.6 ms each, 30 ms after the clocks settled; power and thermal management act on seconds.  Chunks of
    `chunk` steps; per chunk the wall time and the kernel time from HIP events on the launch stream (every 16th group of 8
    launches).  Context, never `value`; the line's `roofline` is restated on this leg's kernel time as `roofline_frac`."""
    import gc
    import numpy as np
    import torch
    saved_env = {k: os.environ.get(k) for k in ("NMMA_PROFILE_GROUP", "NMMA_PROFILE_STRIDE")}
    os.environ["NMMA_PROFILE_GROUP"], os.environ["NMMA_PROFILE_STRIDE"] = "8", "16"
    n_sets = len(thetas)
    gc.collect()
    gc.disable()
    try:
        torch.cuda.synchronize()
        chunks, done = [], 0
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < seconds:
            eng.profile_begin(chunk)
            t0 = time.perf_counter()
            for i in range(done, done + chunk):
                eng.loglike(thetas[i % n_sets], out=out)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            p = eng.profile_end()
            chunks.append((t0 - t_start, t1 - t0, 1e3 * p["fused_ms_total"] / max(1, p["n_launches"])))
            done += chunk
        total = time.perf_counter() - t_start
    finally:
        gc.enable()
        for k, v in saved_env.items():      # (the legs that follow -- host calls, CPU baseline children -- must not inherit the profiling knobs)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    start, wall, kern_us = (np.array(v) for v in zip(*chunks))
    first, last = start < 1.0, start >= start[-1] - 1.0
    k_first, k_last = float(kern_us[first].mean()), float(kern_us[last].mean())
    rate = done * batch / total
    k_all = float(kern_us.mean())
    return {"seconds": total, "steps": done, "evals_per_s": rate, "ratio_to_value": rate / value,
            "ms_per_step": 1e3 * total / done, "kernel_us_first_second": k_first, "kernel_us_last_second": k_last,
            "kernel_us_mean": k_all, "kernel_us_min_chunk": float(kern_us.min()), "kernel_us_max_chunk": float(kern_us.max()),
            "drift_pct": 100.0 * (k_last - k_first) / k_first,
            "evals_per_s_first_second": float(chunk * batch * first.sum() / wall[first].sum()),
            "evals_per_s_last_second": float(chunk * batch * last.sum() / wall[last].sum()),
            "roofline_frac": eng.flops_per_eval * batch / (k_all * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "kernel_us_timed_regions": 1e3 * kernel_ms_region,
            "what": f"back-to-back steps of the same workload for >= {seconds:g} s in chunks of {chunk} (synchronised per chunk); kernel time from HIP "
                    "events on the launch stream; drift_pct = last second's kernel time against the first second's"}


def host_call_ms(eng, case, syn):
    """PCIe-inclusive cost of the reference-shaped call: numpy theta in host memory -> logL in host memory
    (nmma_em_loglike_host), for the full batch and for the single point an unmodified sampler sends.  Never `value`."""
    import gc
    import numpy as np
    res = {}
    gc.collect()
    for b in (4096, 1):
        th = np.ascontiguousarray(syn.draw_theta(4321, b, case["names"])[1])
        for _ in range(5):
            eng.loglike(th)
        # median over chunks: a fresh process shows an occasional ~50 ms runtime hiccup between host calls
        per_chunk, chunks = (25, 5) if b > 1 else (100, 5)
        vals = []
        for _ in range(chunks):
            t0 = time.perf_counter()
            for _ in range(per_chunk):
                eng.loglike(th)
            vals.append(1e3 * (time.perf_counter() - t0) / per_chunk)
        res[f"batch_{b}"] = float(np.median(vals))
    return res


def device_walk_step_ms(eng, case, syn, n=4096, steps=400):
    """One MCMC step of the lock-step ensemble walk on the device -- proposal + prior transform, the likelihood launch, accept:
    two launches (nmma_amd.sampler.device_walk) -- for `n` chains.  Context for the sampler seam; never `value`."""
    import numpy as np
    import torch
    from nmma_amd import sampler as smp

    class Uniform:          # (the analytic prior the device table recognises by name)
        def __init__(self, lo, hi):
            self.minimum, self.maximum = float(lo), float(hi)

    names = case["names"]
    th = syn.draw_theta(3, 20000, names)[1]
    lo, hi = th.min(axis=0), th.max(axis=0)
    table = smp.device_prior_table({k: Uniform(a, b) for k, a, b in zip(names, lo, hi)}, names)
    live = np.random.default_rng(11).uniform(0.3, 0.7, (n, len(names)))
    # (the bound a nested sampler would hold at this moment: a low quantile of the live points' likelihoods -- tools/perf_device_walk.py)
    bound = np.full(n, np.quantile(eng.loglike(np.ascontiguousarray(lo + live * (hi - lo))), 0.2))
    keys = np.arange(1000, 1000 + n, dtype=np.uint64)
    buf = torch.empty(n, dtype=torch.float64, device=f"cuda:{eng.device}")
    ll = lambda t: eng.loglike(t, out=buf)
    smp.device_walk(table, live, live, bound, keys, 20, ll, device=eng.device)
    t0 = time.perf_counter()
    smp.device_walk(table, live, live, bound, keys, steps, ll, device=eng.device)
    return 1e3 * (time.perf_counter() - t0) / steps


def device_walk_queue_sharded(eng, case, syn, dist, world, rank, dev, share_gpu, per_rank=4096, walks=100, repeats=7):
    """ONE queue of the nested sampler sharded over the torchrun ranks -- ``parallel.ShardedQueue`` with this rank's ``EMEngine``:
    every rank walks its contiguous shard of the chains on its GPU (one library call: upload, `walks` fused likelihood + MCMC-step
    launches, fresh draws), the library packs the shard's records on the device, ONE all-gather (RCCL over xGMI; gloo through the host
    in the share-GPU test mode) exchanges them and one download brings all records to every rank.  Weak: `per_rank` chains per rank;
    strong: `per_rank` chains in total.  Times are wall per queue, MAX over ranks, median of `repeats` (barrier + synchronize on both
    sides).  The reference spreads a queue's chains over its MPI ranks the same way (core/mpi_setup.py:651-667, :679-683)."""
    import numpy as np
    import torch
    from nmma_amd import sampler as smp
    from nmma_amd.parallel import ShardedQueue, shard_bounds

    class Uniform:          # (the analytic prior the device table recognises by name)
        def __init__(self, lo, hi):
            self.minimum, self.maximum = float(lo), float(hi)

    names = case["names"]
    th = syn.draw_theta(3, 20000, names)[1]
    lo, hi = th.min(axis=0), th.max(axis=0)
    table = smp.device_prior_table({k: Uniform(a, b) for k, a, b in zip(names, lo, hi)}, names)
    ndim = len(names)
    queue = ShardedQueue(engine=eng, table=table)

    def timed(fn):
        vals = []
        for _ in range(repeats):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            vals.append(time.perf_counter() - t0)
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        if share_gpu:
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(np.median(t.cpu().numpy()))

    out = {}
    for mode, n in (("weak", per_rank * world), ("strong", per_rank)):
        live = np.random.default_rng(11).uniform(0.3, 0.7, (n, ndim))            # (the same on every rank: seeded)
        # (the bound a nested sampler would hold at this moment: a low quantile of the live points' likelihoods -- tools/perf_device_walk.py)
        bound = float(np.quantile(eng.loglike(np.ascontiguousarray(lo + live * (hi - lo))), 0.2))
        keys = np.arange(1000, 1000 + n, dtype=np.uint64)
        res = queue.run(live, live, bound, keys, walks)                         # (warm-up: workspace, buffers, communicator)
        a, b = shard_bounds(n, world, rank)
        t_queue = timed(lambda: queue.run(live, live, bound, keys, walks))
        walk_ms, exch_ms = queue.last_gpu_ms, queue.last_collect_ms
        # the rank's shard alone, records downloaded instead of exchanged (what one GPU does with its share of the queue)
        t_local = timed(lambda: eng.walk_queue(table, live, live[a:b], bound, keys[a:b], walks)) if b > a else 0.0
        u, v, logl, counts = res
        assert u.shape == (n, ndim) and np.all(np.isfinite(logl)) and int(counts[:, 3].sum()) > 0
        out[mode] = {"chains": n, "chains_per_rank": b - a, "walks": walks, "queue_ms": 1e3 * t_queue, "evals_per_s": n * walks / t_queue,
                     "rank0_walk_gpu_ms": walk_ms, "rank0_wait_exchange_download_ms": exch_ms, "rank_shard_alone_ms": 1e3 * t_local,
                     "record_bytes_per_rank": (-(-n // world)) * (2 * ndim + 3) * 8}
    # the collective by itself: the weak-mode send buffer, all-gathered back to back
    slot, width = per_rank, 2 * ndim + 3
    if share_gpu:
        send, recv = torch.zeros((slot, width), dtype=torch.float64), torch.empty((world * slot, width), dtype=torch.float64)
    else:
        send, recv = torch.zeros((slot, width), dtype=torch.float64, device=dev), torch.empty((world * slot, width), dtype=torch.float64, device=dev)
    for _ in range(5):
        dist.all_gather_into_tensor(recv, send)
    calls = 50

    def gathers():
        for _ in range(calls):
            dist.all_gather_into_tensor(recv, send)
    out["allgather_us_per_call"] = 1e6 * timed(gathers) / calls
    out["allgather_bytes_per_rank"] = slot * width * 8
    out["exchange"] = "gloo all_gather through the host (TEST MODE: ranks share one GPU)" if share_gpu else "RCCL all_gather of the device-packed records, one download"
    out["what"] = ("parallel.ShardedQueue over the ranks, one EMEngine per rank: wall per queue (max over ranks, median), weak = "
                   f"{per_rank} chains per rank, strong = {per_rank} chains in total, {walks} MCMC steps per chain; rank_shard_alone_ms = the "
                   "same shard walked and downloaded without the exchange")
    return out


def device_walk_queue(case, syn, inner_rate, n=4096, walks=100, repeats=7):
    """The sampler seam END TO END: wall time of ``GPUPool.map(walker.sample, queue)`` -- the call the nested sampler of
    nmma/core/mpi_setup.py:282-303 makes per queue -- for a queue of `n` records x `walks` MCMC steps of the default
    "acceptance-walk", through the reference-shaped plugin objects (SVDLightCurveModel, EMTransientLikelihood, GPUPool,
    EnsembleWalkSampler.prepare_sampler).  Context for SURVEY section 8 f1; never `value`."""
    import numpy as np
    from nmma_amd import sampler as smp
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    from nmma_amd.pool import GPUPool

    class Uniform:          # (bilby's Uniform as far as the seam reads it; the device table recognises the class by name)
        def __init__(self, lo, hi):
            self.minimum, self.maximum = float(lo), float(hi)

        def rescale(self, val):
            return self.minimum + val * (self.maximum - self.minimum)

    names = case["names"]
    th = syn.draw_theta(3, 20000, names)[1]
    pri = {k: Uniform(a, b) for k, a, b in zip(names, th.min(axis=0), th.max(axis=0))}
    model = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=case["model_filters"], model_parameters=case["model_parameters"],
                               sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    times, mags, sigmas = case["data"]
    handler = FilterSystematicsHandler(case["observed_filters"], systematics_file=None, error_budget=1.0, light_curve_times=times)
    lik = EMTransientLikelihood(model, (times, mags, sigmas, 0.0), handler, dict(pri), filters=case["observed_filters"],
                                detection_limit=case["detection_limit"])
    pt = smp.BatchedPriorTransform(pri, names)
    pool = GPUPool(lik, queue_size=n, names=names, prior_transform_many=pt, priors=pri)
    walker = smp.EnsembleWalkSampler(ndim=len(names), walks=walks, naccept=60)
    live = np.random.default_rng(11).uniform(0.3, 0.7, (n, len(names)))

    class _NS:
        live_u = live
    bound = float(np.quantile(lik.log_likelihood_batch(np.ascontiguousarray(pt(live)), names), 0.2))
    seeds = np.arange(1000, 1000 + n)

    def queue():
        return walker.prepare_sampler(loglstar=bound, points=live, axes=None, seeds=seeds, prior_transform=pt, loglikelihood=None,
                                      nested_sampler=_NS)
    pool.map(walker.sample, queue())
    t_map, t_all, t_list, gpu = [], [], [], []
    eng = lik.sub_model.engine(names)
    for _ in range(repeats):
        t0 = time.perf_counter()
        res = pool.map(walker.sample, queue())
        t1 = time.perf_counter()
        recs = list(res)                     # dynesty: self.queue = list(mapper(...)) -- 4096 Python records
        t2 = time.perf_counter()
        t_map.append(t1 - t0); t_all.append(t2 - t0); gpu.append(eng.last_walk_gpu_ms * 1e-3)
    args_list = list(queue())                # per-record arguments (a driver that builds its own records)
    for _ in range(3):
        t0 = time.perf_counter()
        pool.map(walker.sample, args_list)
        t_list.append(time.perf_counter() - t0)
    assert len(recs) == n and all(np.isfinite(r[2]) for r in recs[:64])
    med = lambda v: float(np.median(v))
    rate = n * walks / med(t_map)
    eng.close()
    return {"records": n, "walks": walks, "map_ms": 1e3 * med(t_map), "device_ms": 1e3 * med(gpu), "map_plus_materialised_records_ms": 1e3 * med(t_all),
            "map_ms_per_record_arguments": 1e3 * med(t_list), "evals_per_s": rate, "fraction_of_inner_loop_rate": rate / inner_rate,
            "what": "wall time of GPUPool.map(walker.sample, queue) for prepare_sampler's array-backed queue: one library call "
                    "(nmma_em_walk_queue: upload, 100 x ONE launch {likelihood + accept + next proposal}, fresh draws, download), array-backed "
                    "records; fraction_of_inner_loop_rate is relative to the two-launch step loop of `device_walk`"}


def _oracle_rows():
    from nmma_amd import synthetic as syn
    from oracle.nmma_oracle import likelihood_from_case
    case = syn.config2_case()
    lik = likelihood_from_case(case, use_scipy=True)
    names, theta = syn.draw_theta(555, 4096, case["names"])
    return lik, [dict(zip(names, (float(v) for v in r))) for r in theta]


def _cpu_worker(budget_s):
    """Child process of cpu_baseline_all_cores: the 1-core loop, prints `evals seconds`."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    lik, rows = _oracle_rows()
    for r in rows[:10]:
        lik.log_likelihood(r)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        lik.log_likelihood(rows[n % len(rows)])
        n += 1
    print(f"{n} {time.perf_counter() - t0}", flush=True)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cgroup_cpu_quota():
    """CPU time the container may use, in cores (cgroup v2 ``cpu.max`` or v1 ``cpu.cfs_quota_us / cpu.cfs_period_us``), or None when
    unlimited / unreadable: ``sched_getaffinity`` reports the host's hardware threads whatever the quota is."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            quota = float(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            period = float(fh.read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def cpu_baseline_all_cores(budget_s, one_core_rate=None):
    """The MPI task farm of the reference (one likelihood per rank, core/mpi_setup.py:651-667) as independent
    single-threaded processes, one per host core -- children that never touch the GPU."""
    import subprocess
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    budget = min(budget_s, 10.0)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(budget)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for _ in range(cores)]
    total, slowest = 0, 0.0
    for pr in procs:
        out, _ = pr.communicate()
        try:
            n, dt = out.split()[-2:]
            total += int(n); slowest = max(slowest, float(dt))
        except (ValueError, IndexError):
            pass
    value = total / slowest if slowest > 0 else None
    # `cores` = the processes started (one per hardware thread this process may run on); a container's CPU quota can be far below
    # that -- then the rate is the quota's, not 256 cores' -- so the quota and the rate in units of the one-core leg are stated too
    return {"value": value, "unit": "evals/s", "cores": cores, "cpu_model": _cpu_model(),
            "cgroup_cpu_quota_cores": _cgroup_cpu_quota(),
            "equivalent_single_cores": (value / one_core_rate) if (value and one_core_rate) else None,
            "kind": "port", "sample": f"{cores} independent single-threaded processes x {budget:.0f} s of the same loop; cores = processes started "
                                      "(sched_getaffinity), cgroup_cpu_quota_cores = the container's CPU-time limit (null: none), "
                                      "equivalent_single_cores = this rate / the one-core leg's"}


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
    from oracle.nmma_oracle import likelihood_from_case
    lik = likelihood_from_case(case, use_scipy=True)
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
    return {"value": n / dt, "unit": "evals/s", "cores": 1, "cpu_model": _cpu_model(), "kind": "port",
            "sample": f"{n} evaluations cycling through the 4096 live points of the same workload, one parameter vector per call "
                      f"({dt:.1f} s, numpy fp32 MLP + scipy.stats as in the reference)"}


if __name__ == "__main__":
    main()
