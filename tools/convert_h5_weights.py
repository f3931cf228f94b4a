#!/usr/bin/env python
"""Extract the trained Keras networks the reference ships for its tests
(/root/reference/nmma/tests/data/Bu2019nsbh_tf/{ztfr,sdssu,2massks}.h5: Dense(3 -> 2048, relu) -> Dense(2048 -> 10), fp32,
keras 2.15 legacy HDF5) with nmma_amd.em.io._dense_weights_from_h5 and store them as a fixture
(tests/golden/bu2019nsbh_tf_weights.npz).  Needs h5py: run with the image's conda interpreter,

    /opt/conda/bin/python3.9 tools/convert_h5_weights.py

(the only interpreter here that has it).  The fixture is DATA of the reference's own tests, not source."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/nmma/tests/data/Bu2019nsbh_tf"

# load em/io.py alone (the package __init__ chain needs nothing else, but keep the conda interpreter out of torch)
spec = importlib.util.spec_from_file_location("em_io", os.path.join(ROOT, "nmma_amd", "em", "io.py"))
em_io = importlib.util.module_from_spec(spec)
spec.loader.exec_module(em_io)

out = {}
for filt in ("ztfr", "sdssu", "2massks"):
    w1, b1, w2, b2 = em_io._dense_weights_from_h5(os.path.join(SRC, f"{filt}.h5"))
    assert w1.shape == (3, 2048) and b1.shape == (2048,) and w2.shape == (2048, 10) and b2.shape == (10,), (w1.shape, w2.shape)
    assert w1.dtype == np.float32 and w2.dtype == np.float32
    for k, a in (("W1", w1), ("b1", b1), ("W2", w2), ("b2", b2)):
        out[f"{filt}/{k}"] = a
    print(f"{filt}: W1 std {w1.std():.3f}  b1 std {b1.std():.3f}  W2 std {w2.std():.4f}  b2 {np.round(b2[:3], 3)}")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bu2019nsbh_tf_weights.npz"), **out)
print("written", os.path.join(ROOT, "tests", "golden", "bu2019nsbh_tf_weights.npz"))
