#!/usr/bin/env python
"""Kernel time of BASELINE config 2 against the batch size, with and without the band split of small batches
(HIP events around groups of back-to-back launches; NMMA_EM_SPLIT=0 | 1 forces one form).  One line per batch size."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn  # noqa: E402
from nmma_amd.engine import EMEngine  # noqa: E402


def timed(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    case = syn.config2_case()
    batches = [int(a) for a in sys.argv[1:]] or [1, 16, 128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096]
    for mode in ("0", "1", "auto"):
        if mode == "auto":
            os.environ.pop("NMMA_EM_SPLIT", None)       # the library's choice: groups of 1 / 2 / 3 bands per workgroup, or none
        else:
            os.environ["NMMA_EM_SPLIT"] = mode
        eng = EMEngine.from_case(case)
        for B in batches:
            th = torch.as_tensor(syn.draw_theta(7, B, case["names"])[1], device="cuda:0")
            out = torch.empty(B, dtype=torch.float64, device="cuda:0")
            us = timed(lambda: eng.loglike(th, out=out))
            g = eng.last_launch_geometry()
            host = np.ascontiguousarray(th.cpu().numpy())
            import time
            eng.loglike(host)
            t0 = time.perf_counter()
            for _ in range(200):
                eng.loglike(host)
            hus = (time.perf_counter() - t0) / 200 * 1e6
            print(f"split={mode:>4} B={B:5d}: {us:7.2f} us/launch (grid {g['grid_x']}x{g['grid_y']}), {B / us:8.3f} Mevals/s; host numpy in/out {hus:7.1f} us/call", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
