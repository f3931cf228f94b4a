#!/opt/conda/bin/python3.9
"""Writes tests/golden/keras3_layout/ztfr.keras: an archive in the layout Keras 3 saves (``.keras`` = zip of config.json, metadata.json
and ``model.weights.h5`` holding ``layers/<name>/vars/{0 = kernel, 1 = bias}`` plus ``optimizer/vars/*``) -- restated from Keras' public
sources, NOT produced by keras (absent from the image; the reference tree holds no .keras file), with the weights of the reference's
trained ztfr network.  Needs h5py: run with /opt/conda/bin/python3.9 in the build container.  The fixture lets the .keras branch of
nmma_amd/em/io.py run without keras; it does not pin the layout."""
import io
import json
import os
import zipfile

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = h5py.File("/root/reference/nmma/tests/data/Bu2019nsbh_tf/ztfr.h5", "r")
w1 = np.array(src["model_weights/dense_50/dense_50/kernel:0"]); b1 = np.array(src["model_weights/dense_50/dense_50/bias:0"])
w2 = np.array(src["model_weights/dense_51/dense_51/kernel:0"]); b2 = np.array(src["model_weights/dense_51/dense_51/bias:0"])
bio = io.BytesIO()
with h5py.File(bio, "w") as f:
    for lname, (k, b) in (("dense", (w1, b1)), ("dense_1", (w2, b2))):
        g = f.create_group(f"layers/{lname}/vars")
        g.create_dataset("0", data=k)
        g.create_dataset("1", data=b)
    f.create_group("layers/dropout/vars")
    o = f.create_group("optimizer/vars")
    for i, a in enumerate((np.int64(7), w1 * 0, b1 * 0, w2 * 0, b2 * 0)):
        o.create_dataset(str(i), data=a)
out = os.path.join(ROOT, "tests", "golden", "keras3_layout", "ztfr.keras")
os.makedirs(os.path.dirname(out), exist_ok=True)
with zipfile.ZipFile(out, "w", zipfile.ZIP_DEFLATED) as z:
    z.writestr("metadata.json", json.dumps({"keras_version": "3.x (layout restated)", "date_saved": "n/a"}))
    z.writestr("config.json", json.dumps({"class_name": "Sequential", "config": {"layers": ["InputLayer", "Dense", "Dropout", "Dense"]}}))
    z.writestr("model.weights.h5", bio.getvalue())
print("wrote", out, os.path.getsize(out))
