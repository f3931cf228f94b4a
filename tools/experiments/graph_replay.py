#!/usr/bin/env python
"""Experiment: does replaying N likelihood launches from a HIP graph shorten the gap between dependent launches?
100 back-to-back eng.loglike calls on one stream, eager against torch.cuda.CUDAGraph replay, for several batch sizes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmma_amd import synthetic as syn  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import engine_from_case  # noqa: E402

case = cases.case_c2_default()
eng = engine_from_case(case)
N = 100
for B in (64, 256, 1024, 4096, 8192):
    th = torch.as_tensor(syn.draw_theta(3, B, case["names"])[1], device="cuda:0")
    out = torch.empty(B, dtype=torch.float64, device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(20):
            eng.loglike(th, out=out, stream=s)
        s.synchronize()

        def eager():
            for _ in range(N):
                eng.loglike(th, out=out, stream=s)

        def timed(fn, reps=7):
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record(s)
                fn()
                e1.record(s)
                s.synchronize()
                ts.append((e0.elapsed_time(e1) * 1e3 / N, (time.perf_counter() - t0) * 1e6 / N))
            return np.median(np.array(ts), axis=0)

        te = timed(eager)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s):
                eager()
            tg = timed(g.replay)
            print(f"B={B:5d}: eager {te[0]:6.2f} us/launch device ({te[1]:6.2f} wall), graph replay {tg[0]:6.2f} us/launch device ({tg[1]:6.2f} wall)", flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f"B={B}: capture failed: {exc}", flush=True)
            break
eng.close()
