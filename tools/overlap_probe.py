#!/usr/bin/env python
"""Per-step cost of evaluation + all-gather of logL, blocking (one stream) against pipelined (kernel on one stream, the
collective of the previous step on another).  Run under torch.distributed.run, one rank per GPU (one rank is enough to
see the host-side cost):  python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/overlap_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import engine_from_case  # noqa: E402

rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
case = cases.case_c2_default()
eng = engine_from_case(case, device=local)
B, steps = 4096, 300
dev = torch.device(f"cuda:{local}")
th = torch.as_tensor(syn.draw_theta(7 + rank, B, case["names"])[1], device=dev)
out = [torch.empty(B, dtype=torch.float64, device=dev) for _ in range(2)]
gathered = [torch.empty(world * B, dtype=torch.float64, device=dev) for _ in range(2)]


def blocking(n):
    for i in range(n):
        eng.loglike(th, out=out[0])
        dist.all_gather_into_tensor(gathered[0], out[0])


def pipelined(n):
    s0, s1 = pipelined.streams
    for i in range(n):
        b = i & 1
        if i >= 2:
            s0.wait_event(pipelined.ev_a[b])          # the gather that read out[b] two steps ago
        eng.loglike(th, out=out[b], stream=s0)
        pipelined.ev_k[b].record(s0)
        s1.wait_event(pipelined.ev_k[b])
        dist.all_gather_into_tensor(gathered[b], out[b])      # current stream is s1
        pipelined.ev_a[b].record(s1)


def timed(fn, label):
    fn(20)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6
    if rank == 0:
        print(f"{label:10s} {us:7.1f} us per step ({world} rank(s), {B} rows per rank)")


def kernel_only(n):
    for i in range(n):
        eng.loglike(th, out=out[0])


blocking(200)                      # RCCL sets itself up during the first collectives
torch.cuda.synchronize()
timed(kernel_only, "kernel")
timed(blocking, "blocking")
timed(blocking, "blocking")
ref = gathered[0].clone()
pipelined.streams = (torch.cuda.Stream(), torch.cuda.Stream())
pipelined.ev_k = [torch.cuda.Event() for _ in range(2)]
pipelined.ev_a = [torch.cuda.Event() for _ in range(2)]
torch.cuda.synchronize()
torch.cuda.set_stream(pipelined.streams[1])
timed(pipelined, "pipelined")
timed(pipelined, "pipelined")
torch.cuda.synchronize()
assert torch.equal(gathered[0], ref) and torch.equal(gathered[1], ref), "pipelined result differs"
if rank == 0:
    print("pipelined results identical to blocking")


# Round 6: the same pipeline with the collective replayed from a captured HIP graph (one graph per buffer: torch.distributed's call costs
# ~25 us of host time per step, a graph launch a few) -- the kernel still goes through the C ABI's stream argument.
def graph_pipelined(n):
    s0, s1 = pipelined.streams
    for i in range(n):
        b = i & 1
        if i >= 2:
            s0.wait_event(pipelined.ev_a[b])
        eng.loglike(th, out=out[b], stream=s0)
        pipelined.ev_k[b].record(s0)
        s1.wait_event(pipelined.ev_k[b])
        graph_pipelined.graphs[b].replay()                    # (current stream is s1)
        pipelined.ev_a[b].record(s1)


try:
    gathered[0].zero_(); gathered[1].zero_()
    torch.cuda.synchronize()
    graphs = []
    for b in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=pipelined.streams[1]):
            dist.all_gather_into_tensor(gathered[b], out[b])
        graphs.append(g)
    graph_pipelined.graphs = graphs
    torch.cuda.synchronize()
    torch.cuda.set_stream(pipelined.streams[1])
    timed(graph_pipelined, "graph-coll")
    timed(graph_pipelined, "graph-coll")
    torch.cuda.synchronize()
    assert torch.equal(gathered[0], ref) and torch.equal(gathered[1], ref), "graph-replayed collective differs"
    if rank == 0:
        print("graph-replayed collective: results identical to blocking")
except Exception as exc:      # noqa: BLE001
    if rank == 0:
        print(f"graph capture of the collective failed: {type(exc).__name__}: {exc}")
eng.close()
dist.destroy_process_group()
