#!/usr/bin/env python
"""The rows of README's "Measured on MI355X" table in one run (HIP events on the launch stream, 30 launches after 5 of warm-up).
Usage: python tools/perf_table.py > gpurun_out/<tag>/perf_table.log"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from tests import cases  # noqa: E402
from tests.helpers import engine_from_case, plugin_from_case  # noqa: E402


def timed(fn, n=30, warm=5):
    import gc
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    gc.collect()          # (a full collection takes ~75 ms with torch imported and would otherwise fire inside a timed loop now and then)
    gc.disable()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    gc.enable()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    rows = [("c2_default", 4096), ("c2_default", 65536), ("c2_dt05", 4096), ("syserr_param", 4096), ("syserr_time_nodes", 4096),
            ("c2_dt05_limit", 4096), ("extinction_limit", 4096), ("extinction_p92", 4096), ("log_grid", 4096),
            ("averaging", 4096), ("c4_shape", 8192), ("c4_shape", 65536), ("at2017gfo", 4096)]
    for name, B in rows:
        case = (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
        eng = engine_from_case(case)
        th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
        out = torch.empty(B, dtype=torch.float64, device="cuda:0")
        us = timed(lambda: eng.loglike(th, out=out))
        eng.check()
        geo = eng.last_launch_geometry()
        print(f"{name:20s} B={B:6d}: {us:8.1f} us/launch  {B / us:8.2f} Mevals/s  {eng.flops_per_eval * B / us / 1e6:6.1f} TF/s  "
              f"block {geo['block']} tile {geo['tile_samples']}")
        eng.close()

    # host-buffer entry point and the per-sample plugin call
    case = cases.case_c2_default()
    eng = engine_from_case(case)
    th = syn.draw_theta(7, 4096, case["names"])[1]
    def host_median(fn, chunks=9, per_chunk=100):
        """median over chunks: a fresh process shows one or two ~50 ms runtime hiccups in its first second of host calls"""
        vals = []
        for _ in range(chunks):
            t0 = time.perf_counter()
            for _ in range(per_chunk):
                fn()
            vals.append((time.perf_counter() - t0) / per_chunk * 1e6)
        return float(np.median(vals))


    for B in (4096, 1):
        sub = np.ascontiguousarray(th[:B])
        eng.loglike(sub)
        print(f"host numpy in/out    B={B:6d}: {host_median(lambda: eng.loglike(sub)):8.1f} us/call (median of 9 x 100)")
    eng.close()
    _, _, lik = plugin_from_case(case)
    p = dict(zip(case["names"], (float(v) for v in case["theta"][0])))
    lik.log_likelihood(p)
    print(f"plugin log_likelihood(dict)      : {host_median(lambda: lik.log_likelihood(p)):8.1f} us/call (median of 9 x 100)")


if __name__ == "__main__":
    main()
