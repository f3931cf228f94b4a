"""On-disk formats of the EM path.

* ``save_svd_model`` / ``load_svd_model``: the flat tensor file (one ``.npz``) this
  framework reads -- ``{filter}/W1,b1,W2,b2,VA,mins,maxs,tt,param_mins,param_maxs,n_coeff``.
* ``convert_reference_model``: reads the reference's own layout
  (``{model}.joblib`` + ``{model}[_tf]/{filter}.keras|.h5``, nmma/em/model.py:593-696)
  and writes the flat file: ``.joblib`` through joblib, legacy ``.h5`` networks through h5py or,
  without it, the built-in reader ``em/hdf5_lite.py``; ``.keras`` archives need keras.
* ``load_em_observations`` / ``write_em_observations``: light-curve data files in the reference's formats (nmma/em/io.py:16-184):
  ``time filter mag mag_error`` rows, the forced-photometry CSV, model tables, ``.json`` in standard or model form.
"""
from __future__ import annotations

import os
from datetime import datetime, timezone

import numpy as np

_KEYS = ("W1", "b1", "W2", "b2", "VA", "mins", "maxs", "tt", "param_mins", "param_maxs")


def save_svd_model(path, svd_model, model_parameters=None):
    flat = {}
    for filt, t in svd_model.items():
        n_c = int(t["n_coeff"])
        for k in _KEYS:
            a = np.asarray(t[k])
            if k == "VA":
                a = a[:, :n_c]
            flat[f"{filt}/{k}"] = a
        flat[f"{filt}/n_coeff"] = np.int64(n_c)
    if model_parameters is not None:
        flat["__model_parameters__"] = np.array(list(model_parameters))
    np.savez_compressed(path, **flat)


def load_svd_model(path):
    """Returns ``(svd_model, model_parameters or None)``."""
    svd, params = {}, None
    with np.load(path, allow_pickle=False) as z:
        for name in z.files:
            if name == "__model_parameters__":
                params = [str(x) for x in z[name]]
                continue
            filt, key = name.rsplit("/", 1)
            svd.setdefault(filt, {})[key] = z[name]
    for filt, t in svd.items():
        t["n_coeff"] = int(t["n_coeff"])
        for k in ("W1", "b1", "W2", "b2"):
            t[k] = np.ascontiguousarray(t[k], dtype=np.float32)
        for k in ("VA", "mins", "maxs", "tt", "param_mins", "param_maxs"):
            t[k] = np.ascontiguousarray(t[k], dtype=np.float64)
    return svd, params


def _h5_datasets(path):
    """``{"group/.../name": array}`` of a legacy Keras ``.h5`` file: through h5py when it is importable, else through the built-in
    reader of the HDF5 subset such files use (``em/hdf5_lite.py``; the build image has no h5py)."""
    try:
        import h5py
        if not isinstance(getattr(h5py, "File", None), type):        # (a stand-in module, e.g. the test harness' import stubs)
            raise ImportError("h5py is not the real package")
    except ImportError:
        from . import hdf5_lite
        return hdf5_lite.read_datasets(path)
    out = {}
    with h5py.File(path, "r") as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                out[name] = np.array(obj)
        f.visititems(visit)
    return out


def _dense_weights_from_h5(path):
    """(W1, b1, W2, b2) of a Dense -> (Dropout) -> Dense Keras model saved as legacy HDF5
    (``model_weights/<layer>/<layer>/{kernel,bias}:0``, nmma/em/training.py:353-364; what ``keras.saving.load_model`` reads at
    nmma/em/model.py:635-648).  The two Dense layers are told apart by their shapes -- the first one's output width is the second
    one's input width -- not by the order of the group names (``dense_9`` sorts after ``dense_10``)."""
    layers = {}
    for name, arr in _h5_datasets(path).items():
        parts = name.split("/")
        if parts[0] == "model_weights":          # a full model file (keras model.save): model_weights/<layer>/<layer>/{kernel,bias}:0
            parts = parts[1:]
        if len(parts) < 2 or parts[0] == "optimizer_weights":
            continue                             # (a weights-only file, model.save_weights, has the layer groups at the root)
        leaf = parts[-1].split(":")[0]
        if leaf in ("kernel", "bias"):
            layers.setdefault(parts[0], {})[leaf] = np.asarray(arr)
    dense = [(v["kernel"], v["bias"]) for v in layers.values() if "kernel" in v and "bias" in v]
    if len(dense) != 2:
        raise ValueError(f"{path}: expected two Dense layers, found {len(dense)}")
    if dense[0][0].shape[1] != dense[1][0].shape[0]:
        dense.reverse()
    (w1, b1), (w2, b2) = dense
    if w1.ndim != 2 or w2.ndim != 2 or w1.shape[1] != w2.shape[0] or b1.shape != (w1.shape[1],) or b2.shape != (w2.shape[1],):
        raise ValueError(f"{path}: layer shapes do not chain: {w1.shape}, {b1.shape}, {w2.shape}, {b2.shape}")
    return w1, b1, w2, b2


def _dense_pair(datasets, where):
    """The two Dense layers among ``datasets`` (name -> array): groups that hold exactly one 2-D kernel and one 1-D bias of matching
    width, ordered by shape chaining."""
    groups = {}
    for name, arr in datasets.items():
        parent = name.rsplit("/", 1)[0] if "/" in name else ""
        groups.setdefault(parent, []).append(np.asarray(arr))
    dense = []
    for members in groups.values():
        if len(members) != 2:
            continue
        k = next((a for a in members if a.ndim == 2), None)
        b = next((a for a in members if a.ndim == 1), None)
        if k is not None and b is not None and b.shape[0] == k.shape[1]:
            dense.append((k, b))
    if len(dense) != 2:
        raise ValueError(f"{where}: expected two Dense layers, found {len(dense)}")
    if dense[0][0].shape[1] != dense[1][0].shape[0]:
        dense.reverse()
    (w1, b1), (w2, b2) = dense
    if w1.shape[1] != w2.shape[0]:
        raise ValueError(f"{where}: layer shapes do not chain: {w1.shape}, {w2.shape}")
    return w1, b1, w2, b2


def _dense_weights_from_keras_archive(path):
    """(W1, b1, W2, b2) of a ``.keras`` archive (Keras 3: a zip holding ``config.json`` and ``model.weights.h5``; what
    nmma/em/training.py saves today and nmma/em/model.py:635-643 loads first).  With keras importable the model is loaded as the
    reference does; without it the weights file inside the archive is read directly (h5py or the built-in HDF5 reader) and the two
    Dense layers are picked by their shapes.  The archive layout (``layers/<name>/vars/{0,1}``) is from Keras' public sources, NOT
    pinned here -- keras is absent from the build image and the reference tree holds no ``.keras`` file."""
    try:
        import keras
        if not hasattr(getattr(keras, "saving", None), "load_model") or not isinstance(getattr(keras, "Model", None), type):
            raise ImportError("keras is not the real package")
        net = keras.saving.load_model(path, compile=False)
        dense = [l for l in net.layers if l.get_weights()]
        (w1, b1), (w2, b2) = (l.get_weights() for l in dense)
        return w1, b1, w2, b2
    except ImportError:
        pass
    import zipfile
    with zipfile.ZipFile(path) as z:
        inner = next((n for n in z.namelist() if n.endswith("model.weights.h5")), None)
        if inner is None:
            raise ValueError(f"{path}: no model.weights.h5 in the archive")
        raw = z.read(inner)
    try:
        import h5py
        if not isinstance(getattr(h5py, "File", None), type):
            raise ImportError
        import io as _io
        data = {}
        with h5py.File(_io.BytesIO(raw), "r") as f:
            f.visititems(lambda name, obj: data.__setitem__(name, np.array(obj)) if isinstance(obj, h5py.Dataset) else None)
    except ImportError:
        from . import hdf5_lite
        data = hdf5_lite.read_datasets(data=raw)
    data = {k: v for k, v in data.items() if not k.startswith("optimizer")}
    return _dense_pair(data, path)


def convert_reference_model(svd_path, model, out_path, filters=None, interpolation_type="tensorflow"):
    """Read the reference's model files and write the flat tensor file."""
    import joblib
    core = "_".join(c for c in model.split("_") if c != "tf")
    meta = joblib.load(os.path.join(svd_path, f"{core}.joblib"))
    meta = {k.replace("_", ":"): v for k, v in meta.items()}          # model.py:604-606
    spec = "_tf" if interpolation_type == "tensorflow" else ""
    svd = {}
    for filt in (filters or list(meta)):
        base = os.path.join(svd_path, f"{model}{spec}", filt.replace(":", "_"))
        if os.path.isfile(base + ".keras"):
            w1, b1, w2, b2 = _dense_weights_from_keras_archive(base + ".keras")
        elif os.path.isfile(base + ".h5"):
            w1, b1, w2, b2 = _dense_weights_from_h5(base + ".h5")
        else:
            raise FileNotFoundError(f"no .keras/.h5 network for filter {filt} under {base}")
        m = meta[filt]
        svd[filt] = dict(W1=np.float32(w1), b1=np.float32(b1), W2=np.float32(w2), b2=np.float32(b2),
                         VA=np.asarray(m["VA"], float), mins=np.asarray(m["mins"], float),
                         maxs=np.asarray(m["maxs"], float), tt=np.asarray(m["tt"], float),
                         param_mins=np.asarray(m["param_mins"], float),
                         param_maxs=np.asarray(m["param_maxs"], float), n_coeff=int(m["n_coeff"]))
    save_svd_model(out_path, svd)
    return svd


def _to_mjd(token, time_format=None):
    """One time token -> MJD: an ISO / ISOT date (UTC), else a number in ``time_format`` (``mjd`` -- the default -- ``jd``, ``unix``:
    the leap-second-free ones of the astropy ``Time`` formats the reference's ``--time-format`` takes, io.py:130-136)."""
    try:
        value = float(token)
    except ValueError:
        dt = datetime.fromisoformat(str(token).strip().replace("Z", "+00:00"))
        if dt.tzinfo is None:
            dt = dt.replace(tzinfo=timezone.utc)
        return dt.timestamp() / 86400.0 + 40587.0
    fmt = (time_format or "mjd").lower()
    if fmt == "mjd":
        return value
    if fmt == "jd":
        return value - 2400000.5
    if fmt == "unix":
        return value / 86400.0 + 40587.0
    raise ValueError(f"time format {time_format!r} is not one of mjd, jd, unix")


def _read_observation_rows(path, time_format):
    """``time filter mag mag_error`` rows (io.py:116-144): comment lines and a ``time`` / ``mjd`` header are skipped."""
    rows = {}
    with open(path) as fh:
        for line in fh:
            parts = line.split()
            if not parts or parts[0].startswith(("#", "time", "mjd")):
                continue
            t, filt, mag, err = _to_mjd(parts[0], time_format), parts[1], float(parts[2]), float(parts[3])
            rows.setdefault(filt, []).append((t, mag, err))
    if not rows:
        raise ValueError(f"no photometry rows in {path}")
    return rows


def _read_survey_csv(path):
    """The forced-photometry CSV the reference falls back to (io.py:101-114): columns ``mjd, filter, mag_corr, magerr,
    limiting_mag``; rows without a magnitude are upper limits at ``limiting_mag`` (infinite error)."""
    import csv
    rows = {}
    with open(path, newline="") as fh:
        for rec in csv.DictReader(fh):
            mag = float(rec["mag_corr"]) if rec["mag_corr"].strip() not in ("", "nan", "NaN") else np.nan
            err = float(rec["magerr"]) if rec["magerr"].strip() else np.nan
            if np.isnan(mag):
                mag, err = float(rec["limiting_mag"]), np.inf
            rows.setdefault(rec["filter"], []).append((float(rec["mjd"]), mag, err))
    return rows


def _read_model_table(path):
    """A model light curve as text (io.py:85-97): header ``time f1 f2 ... [f1_error ...]``, one row per epoch."""
    with open(path) as fh:
        lines = [ln for ln in (raw.strip() for raw in fh) if ln]
    header = lines[0].lstrip("#").split()
    table = np.array([[float(v) for v in ln.split()] for ln in lines[1:] if not ln.startswith("#")])
    cols = dict(zip(header, table.T))
    time = cols.pop("time")
    return {f: {"time": time, "mag": v, "mag_error": cols.get(f + "_error", np.zeros_like(time))}
            for f, v in cols.items() if not f.endswith("_error")}


def _decode_json(obj):
    """bilby's JSON encoding of arrays and complex numbers (``{"__array__": true, "content": [...]}``), as the reference's reader
    decodes it (io.py:60-61)."""
    if isinstance(obj, dict):
        if obj.get("__array__"):
            return np.asarray(obj["content"])
        if obj.get("__complex__"):
            return complex(obj["real"], obj["imag"])
    return obj


def _read_json(path):
    import json
    with open(path) as fh:
        data = json.load(fh, object_hook=_decode_json)
    if "time" in data:          # a model light curve: one time axis, one column per filter (io.py:64-73)
        time = data["time"]
        data = {k: {"time": time, "mag": v, "mag_error": data.get(f"{k}_error", np.zeros_like(np.asarray(time, float)))}
                for k, v in data.items() if k != "time" and not k.endswith("_error")}
    return data


def load_em_observations(filename, args=None, format="observations", filters=None):
    """Light-curve data in the reference's standard form ``{filter: {"time": mjd[], "mag": [], "mag_error": []}}`` (io.py:16-55) from

    * a dict (returned as it is) or a Namespace (its ``light_curve_data``),
    * a ``.json`` file in the standard form or in the model form (``time`` + one column per filter [+ ``<filter>_error``]),
    * a text file of ``time filter mag mag_error`` rows (``format="observations"``; ISO times or numbers in ``args.time_format``;
      an infinite error marks an upper limit), or -- when that does not parse -- the forced-photometry CSV
      (``mjd, filter, mag_corr, magerr, limiting_mag``),
    * a text table ``time f1 f2 ...`` (``format="model"``).

    Text observations come back sorted by time within each filter; ``filters`` keeps only the named ones."""
    import argparse
    if isinstance(filename, dict):
        return filename
    if isinstance(filename, argparse.Namespace) or (args is None and hasattr(filename, "light_curve_data")):
        args, filename = filename, filename.light_curve_data
        if isinstance(filename, dict):
            return filename
    if filename is None:
        raise ValueError("No filename provided for lightcurve data.")
    filename = os.fspath(filename)
    if filename.endswith(".json"):
        data = _read_json(filename)
    elif "obs" in format:
        try:
            rows = _read_observation_rows(filename, getattr(args, "time_format", None))
        except Exception:
            rows = _read_survey_csv(filename)
        data = {}
        for filt, r in rows.items():
            a = np.array(sorted(r))
            data[filt] = {"time": a[:, 0], "mag": a[:, 1], "mag_error": a[:, 2]}
    elif "model" in format:
        data = _read_model_table(filename)
    else:
        raise ValueError("Standard format is not supported for reading from csv files. Please use json files instead.")
    return {filt: {k: np.array(v) for k, v in sub.items()} for filt, sub in data.items() if filters is None or filt in filters}


def _isot(mjd):
    dt = datetime.fromtimestamp(round((float(mjd) - 40587.0) * 86400.0 * 1000.0) / 1000.0, tz=timezone.utc)
    return dt.strftime("%Y-%m-%dT%H:%M:%S.") + f"{dt.microsecond // 1000:03d}"


def write_em_observations(filename, data, format="observations"):
    """The reverse (io.py:146-184): ``.json`` in the standard form; ``.txt`` / ``.dat`` as observation rows (ISOT time, filter,
    magnitude and error with three decimals, sorted by time) or as a model table."""
    import json
    directory = os.path.dirname(os.fspath(filename))
    if directory:
        os.makedirs(directory, exist_ok=True)
    filename = os.fspath(filename)
    if filename.endswith(".json"):
        with open(filename, "w") as fh:
            json.dump({f: {k: np.asarray(v).tolist() for k, v in sub.items()} for f, sub in data.items()}, fh, indent=2)
        return
    if not filename.endswith((".txt", ".dat")):
        return
    if format == "observations":
        rows = sorted((float(t), f, float(m), float(e)) for f, sub in data.items()
                      for t, m, e in zip(sub["time"], sub["mag"], sub["mag_error"]))
        with open(filename, "w") as fh:
            fh.write("#time filter mag mag_error\n")
            for t, f, m, e in rows:
                fh.write(f"{_isot(t)} {f} {m:.3f} {e:.3f}\n")
    elif format == "model":
        names = list(data)
        with_err = [f for f in names if not np.all(np.isnan(np.asarray(data[f]["mag_error"], float)))]
        time = np.asarray(data[names[-1]]["time"], float)
        cols = [np.asarray(data[f]["mag"], float) for f in names] + [np.asarray(data[f]["mag_error"], float) for f in with_err]
        with open(filename, "w") as fh:
            fh.write("#time " + " ".join(names + [f + "_error" for f in with_err]) + "\n")
            for i, t in enumerate(time):
                fh.write(f"{t:.5f} " + " ".join(f"{c[i]:.3f}" for c in cols) + "\n")
    else:
        raise ValueError(f"unknown light-curve file format {format!r}")
