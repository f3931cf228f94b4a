#!/usr/bin/env python
"""Measurement build only (tools/build_variant.sh stamps -DNMMA_DBG_LC_STAMPS; NMMA_HIP_LIB=build_dbg/lib_stamps.so): s_memtime stamps of
one workgroup of em_lc_loglike at its phase boundaries, config 3's shape (8192 rows: every workgroup is resident from the start, so the
last stamp is the kernel's duration -- 35 us -- and the others scale with it)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmma_amd import _lib, synthetic as syn  # noqa: E402
from nmma_amd.engine import EMEngine  # noqa: E402
from tests import cases_combined  # noqa: E402

c3 = cases_combined.case_combined()
B = 8192
_, th6 = syn.draw_theta(777, B, cases_combined.NAMES[:6])
rng = np.random.default_rng(778)
theta = np.concatenate([th6, rng.uniform(-17.5, -14.0, (B, 1)), rng.uniform(0.8, 1.6, (B, 1))], axis=1)
tail = EMEngine(None, c3["filters"], [], c3["names"], sample_times=c3["sample_times"], cosmo_grid=c3["cosmo_grid"],
                data=c3["data"], observed_filters=c3["filters"], model_kind="external")
t = torch.as_tensor(theta, device="cuda:0")
lc = torch.as_tensor(rng.uniform(-17.0, -12.0, (B, 9, 41)), device="cuda:0")
ext = torch.full((B, 9, 41), -15.0, dtype=torch.float64, device="cuda:0")
lib = _lib.load_library()
fn = lib.nmma_dbg_lc_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 64)()
print("stamps: start | curves staged (chains done on one wave) | block barrier | sanity_check | fast lane loop | general terms | end")
for rep in range(4):
    for _ in range(20):
        tail.loglike_lc_sets(t, [lc, ext])
    fn(buf)
    a = np.array(list(buf), dtype=np.int64).reshape(4, 16)[:, :7]
    t0 = a[:, 0].min()
    print("rep", rep)
    for w in range(4):
        print("  wave", w, " ".join(f"{(v - t0) / (a[:, 6].max() - t0):6.3f}" for v in a[w]), "of the workgroup's lifetime")
