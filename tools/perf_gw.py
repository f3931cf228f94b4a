#!/usr/bin/env python
"""HBM roofline of the GW inner-product kernel (nmma_gw_loglike_ratio): achieved bytes/s = B * n_ifo * NF * 16 bytes of strain per
launch / kernel time (HIP events on the launch stream); data and weights (24 bytes per bin, shared by all samples) stay in L2.
Usage: python tools/perf_gw.py [B] [duration_s] [sampling_Hz]   (default 2048 samples, 3 detectors, 32 s at 2048 Hz)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd.gw.gw_likelihood import GWStrainLikelihood  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
duration = float(sys.argv[2]) if len(sys.argv) > 2 else 32.0
fs = float(sys.argv[3]) if len(sys.argv) > 3 else 2048.0
n_ifo, nf = 3, int(duration * fs / 2) + 1
rng = np.random.default_rng(0)
freq = np.arange(nf) / duration
psd = np.full((n_ifo, nf), 1e-46)
data = (rng.normal(size=(n_ifo, nf)) + 1j * rng.normal(size=(n_ifo, nf))) * 1e-23
gw = GWStrainLikelihood(data, psd, freq, duration, minimum_frequency=20.0)
strain = torch.view_as_complex(torch.randn(B, n_ifo, nf, 2, dtype=torch.float64, device="cuda:0") * 1e-23)
out = torch.empty(B, dtype=torch.float64, device="cuda:0")
for _ in range(3):
    gw.log_likelihood_ratio_batch(strain, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    gw.log_likelihood_ratio_batch(strain, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
gb = B * n_ifo * nf * 16 / 1e9
print(f"B={B} n_ifo={n_ifo} NF={nf}: {ms * 1e3:.1f} us per launch, {gb:.2f} GB of strain -> {gb / (ms * 1e-3) / 1e3:.2f} TB/s "
      f"({gb / (ms * 1e-3) / 1e3 / 8.0:.2f} of the 8 TB/s HBM peak), {B / (ms * 1e-3):.3g} evals/s")
