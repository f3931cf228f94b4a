#!/usr/bin/env python
"""Where a step of `kernel + all-gather of log L` spends its time with ONE rank over RCCL: host issue time per step (the loop without a
final synchronize) against the GPU's time per step, for the kernel alone, the blocking form, and pipelined forms with 2 and 4 buffers.
Run: python tools/experiments/step_cost_probe.py   (a process group of this one rank)"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmma_amd import synthetic as syn  # noqa: E402
from nmma_amd.engine import EMEngine  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
with socket.socket() as sock:
    sock.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
os.environ.setdefault("NMMA_EM_RING", os.environ.get("PROBE_RING", "2"))
case = syn.config2_case()
eng = EMEngine.from_case(case)
B, steps = 4096, 2000
dev = torch.device("cuda:0")
th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device=dev)
NB = 4
outs = [torch.empty(B, dtype=torch.float64, device=dev) for _ in range(NB)]
gath = [torch.empty(B, dtype=torch.float64, device=dev) for _ in range(NB)]
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
ev_k = [torch.cuda.Event() for _ in range(NB)]
ev_a = [torch.cuda.Event() for _ in range(NB)]


def kernel_only(n):
    for i in range(n):
        eng.loglike(th, out=outs[0], stream=s0)


def blocking(n):
    with torch.cuda.stream(s0):
        for i in range(n):
            eng.loglike(th, out=outs[0], stream=s0)
            dist.all_gather_into_tensor(gath[0], outs[0])


def pipelined(nb):
    def run(n):
        with torch.cuda.stream(s1):
            for i in range(n):
                b = i % nb
                if i >= nb:
                    s0.wait_event(ev_a[b])
                eng.loglike(th, out=outs[b], stream=s0)
                ev_k[b].record(s0)
                s1.wait_event(ev_k[b])
                dist.all_gather_into_tensor(gath[b], outs[b])
                ev_a[b].record(s1)
    return run


def timed(fn, label):
    fn(1600)          # clocks
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:14s} host issue {1e6 * (t1 - t0) / steps:6.1f} us per step, until the GPU is done {1e6 * (t2 - t0) / steps:6.1f} us per step", flush=True)


blocking(100)
torch.cuda.synchronize()
timed(kernel_only, "kernel")
timed(blocking, "blocking")
timed(pipelined(2), "pipelined x2")
timed(pipelined(4), "pipelined x4")
eng.close()
dist.destroy_process_group()
