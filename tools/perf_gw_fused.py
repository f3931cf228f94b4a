#!/usr/bin/env python
"""Timing of the fused GW kernel (waveform + projection + inner products from parameters) at BASELINE config 5's shape:
128 s at 4096 Hz (262 145 bins, 3 detectors), B = 16 384 -- and smaller shapes.  HIP events on the launch stream bracket
gw_logl_kernel; prints one JSON line per shape.  Usage: python tools/perf_gw_fused.py [--batch B] [--duration T] [--pm]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

#: arithmetic of one (frequency bin, sample) in the inspiral region with tides and 3 detectors, as written in
#: nmma_amd/csrc/gw_math.h:eval_bin + gw_kernels.hip:gw_bin_sample -- +, -, x count 1, an FMA 2, and each division, exp and
#: sincospi ONE (DESIGN section 3.5 has the table)
FLOPS_PER_BIN_SAMPLE = {1: 129, 2: 148, 3: 167, 4: 186}
PEAK_F64_TFLOPS = 78.6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--duration", type=float, default=128.0)
    ap.add_argument("--fs", type=float, default=4096.0)
    ap.add_argument("--ifos", default="H1,L1,V1")
    ap.add_argument("--pm", action="store_true", help="phase marginalisation")
    ap.add_argument("--dm", action="store_true", help="distance marginalisation (bilby's 10^4-node grid, prior uniform in volume)")
    ap.add_argument("--tm", action="store_true", help="time marginalisation (uniform prior of +-0.1 s around the trigger)")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    from nmma_amd import synthetic as syn
    from nmma_amd.gw import GWEngine
    names_ifo = tuple(a.ifos.split(","))
    ifos, wa, inj = syn.make_gw_interferometers(7, a.duration, a.fs, names_ifo)
    names = [n for n in syn.GW_NAMES if not (a.pm and n == "phase") and not (a.dm and n == "luminosity_distance")
             and not (a.tm and n == "geocent_time")]
    _, theta = syn.draw_gw_theta(3, a.batch, centre=inj, names=names, width=0.3)
    kw, fixed = {}, {}
    if a.dm:
        import numpy as np
        grid = np.linspace(10.0, 250.0, 10000)
        kw["distance_marginalization"] = (grid, np.log(3 * grid ** 2 / (250.0 ** 3 - 10.0 ** 3) * (grid[1] - grid[0])))
        fixed["luminosity_distance"] = 100.0
    if a.tm:
        import numpy as np
        n = len(ifos[0].frequency_array) - 1
        t = ifos[0].strain_data.start_time + a.duration / n * np.arange(n)
        with np.errstate(divide="ignore"):
            kw["time_marginalization"] = np.log(np.where(abs(t - inj["geocent_time"]) <= 0.1, 1.0 / 0.2, 0.0) * a.duration / n)
        fixed["geocent_time"] = float(ifos[0].strain_data.start_time)
    eng = GWEngine(ifos, names, fixed=fixed, waveform_arguments=wa, phase_marginalization=a.pm, **kw)
    th = torch.as_tensor(theta, device="cuda:0")
    out = eng.loglike_ratio(th)
    torch.cuda.synchronize()
    eng.profile_begin(a.reps)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        eng.loglike_ratio(th, out=out)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.reps
    ms, n = eng.profile_end()
    ms /= n
    flops = FLOPS_PER_BIN_SAMPLE[len(names_ifo)] * eng.n_bins * a.batch
    o = out.cpu().numpy()
    print(json.dumps(dict(batch=a.batch, n_ifo=len(names_ifo), n_bins=eng.n_bins, phase_marginalization=a.pm,
                          distance_marginalization=a.dm, time_marginalization=a.tm,
                          kernel_ms=ms, call_ms=wall * 1e3, evals_per_s=a.batch / wall,
                          bin_samples_per_s=eng.n_bins * a.batch / (ms * 1e-3),
                          roofline=dict(bound="fp64 vector FMA", achieved=flops / (ms * 1e-3) / 1e12, peak=PEAK_F64_TFLOPS, unit="TFLOP/s",
                                        frac=flops / (ms * 1e-3) / 1e12 / PEAK_F64_TFLOPS,
                                        flops_per_bin_sample=FLOPS_PER_BIN_SAMPLE[len(names_ifo)]),
                          logl_range=[float(o.min()), float(o.max())], floors=int((o < -1e300).sum()))))


if __name__ == "__main__":
    main()
