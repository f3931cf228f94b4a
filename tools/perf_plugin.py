#!/usr/bin/env python
"""Throughput of the bilby-protocol plugin (EMTransientLikelihood.log_likelihood_batch / log_likelihood)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import synthetic as syn
from tests import cases
from tests.helpers import plugin_from_case
case = cases.case_c2_default()
model, handler, lik = plugin_from_case(case)
names = lik.parameter_names()
_, theta = syn.draw_theta(5, 4096, case["names"])
cols = [case["names"].index(n) for n in names]
th = np.ascontiguousarray(theta[:, cols])
for B in (1, 64, 4096):
    x = th[:B]
    for _ in range(5):
        lik.log_likelihood_batch(x)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        lik.log_likelihood_batch(x)
    dt = (time.perf_counter() - t0) / n
    print(f"log_likelihood_batch numpy B={B}: {dt*1e6:.1f} us per call, {B/dt/1e6:.2f} Mevals/s")
xt = torch.as_tensor(th, device="cuda:0")
for _ in range(5):
    lik.log_likelihood_batch(xt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    lik.log_likelihood_batch(xt)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print(f"log_likelihood_batch cuda tensor B=4096: {dt*1e6:.1f} us per call, {4096/dt/1e6:.2f} Mevals/s")
row = dict(zip(names, (float(v) for v in th[0])))
for _ in range(5):
    lik.log_likelihood(row)
t0 = time.perf_counter()
for _ in range(200):
    lik.log_likelihood(row)
dt = (time.perf_counter() - t0) / 200
print(f"log_likelihood(dict) single point: {dt*1e6:.1f} us per call")
