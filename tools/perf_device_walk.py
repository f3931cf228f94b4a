#!/usr/bin/env python
"""MCMC steps per second of the lock-step ensemble walk on BASELINE config 2: the walk on the device (nmma_em_loglike ->
nmma_walk_step = accept + next proposal, two launches per step) against the host walk of nmma_amd/sampler.py around the same
likelihood launch.  Usage: python tools/perf_device_walk.py [chains] [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import sampler as smp  # noqa: E402
from nmma_amd import synthetic as syn  # noqa: E402
from nmma_amd.engine import EMEngine  # noqa: E402


class Uniform:
    def __init__(self, minimum, maximum):
        self.minimum, self.maximum = float(minimum), float(maximum)

    def rescale(self, val):
        return self.minimum + val * (self.maximum - self.minimum)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    case = syn.config2_case()
    eng = EMEngine.from_case(case)
    names = case["names"]
    th = syn.draw_theta(3, 20000, names)[1]
    pri = {k: Uniform(a, b) for k, a, b in zip(names, th.min(axis=0), th.max(axis=0))}
    pt = smp.BatchedPriorTransform(pri, names)
    table = smp.device_prior_table(pri, names)
    rng = np.random.default_rng(11)
    live = rng.uniform(0.3, 0.7, (n, len(names)))
    l_live = eng.loglike(np.ascontiguousarray(pt(live)))
    bound = np.full(n, np.quantile(l_live, 0.2))
    keys = np.arange(1000, 1000 + n, dtype=np.uint64)
    buf = torch.empty(n, dtype=torch.float64, device="cuda:0")
    dev_ll = lambda t: eng.loglike(t, out=buf)
    smp.device_walk(table, live, live, bound, keys, 20, dev_ll)
    t0 = time.perf_counter()
    u, v, logl, counts = smp.device_walk(table, live, live, bound, keys, steps, dev_ll)
    t_dev = (time.perf_counter() - t0) / steps
    print(f"device walk: {t_dev * 1e6:8.1f} us per MCMC step of {n} chains, {n / t_dev / 1e6:7.1f} M likelihood evaluations/s, "
          f"acceptance {counts[:, 0].mean() / steps:.3f}")
    # the host walk: same chains, same random numbers, numpy bookkeeping around nmma_em_loglike_host
    kw = dict(live=live, walks=min(steps, 50))
    args = [smp.SamplerArgument(live[i].copy(), bound[i], int(keys[i]), pt, None, kw) for i in range(n)]
    w = smp.EnsembleWalkSampler(ndim=len(names), walks=kw["walks"], naccept=10)
    host_ll = lambda x: eng.loglike(np.ascontiguousarray(x))
    w.run_many(args[:64], host_ll, pt)
    t0 = time.perf_counter()
    out = w.run_many(args, host_ll, pt)
    t_host = (time.perf_counter() - t0) / kw["walks"]
    print(f"host walk:   {t_host * 1e6:8.1f} us per MCMC step of {n} chains, {n / t_host / 1e6:7.1f} M likelihood evaluations/s, "
          f"acceptance {np.mean([o[4]['accept'] for o in out]) / kw['walks']:.3f}")
    eng.close()


if __name__ == "__main__":
    main()
