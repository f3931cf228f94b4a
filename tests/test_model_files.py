"""Model files in the reference's own layout (nmma/em/model.py:593-696): ``{model}.joblib`` written by
``joblib.dump`` (training.py:save_model) next to ``{model}_tf/{filter}.h5`` written by Keras 2.x -- read here WITHOUT h5py or keras:
``em/hdf5_lite.py`` parses the HDF5 subset such files use.  The fixture ``tests/golden/bu2019nsbh_tf_h5/ztfr.h5`` is one of the three
trained networks the reference ships as test data (nmma/tests/data/Bu2019nsbh_tf, a data file, byte for byte); its weights were read
once with h5py when ``tests/golden/bu2019nsbh_tf_weights.npz`` was made (tools/convert_h5_weights.py), which is what the built-in
reader is compared with.  CPU only."""
import os

import numpy as np
import pytest

from nmma_amd.em import hdf5_lite, io as em_io

HERE = os.path.dirname(os.path.abspath(__file__))
H5 = os.path.join(HERE, "golden", "bu2019nsbh_tf_h5", "ztfr.h5")


def _weights():
    with np.load(os.path.join(HERE, "golden", "bu2019nsbh_tf_weights.npz")) as z:
        return {k: z[f"ztfr/{k}"] for k in ("W1", "b1", "W2", "b2")}


def test_builtin_hdf5_reader_gets_the_trained_weights_bit_for_bit():
    d = hdf5_lite.read_datasets(H5)
    names = sorted(k for k in d if k.startswith("model_weights/"))
    assert names == ["model_weights/dense_50/dense_50/bias:0", "model_weights/dense_50/dense_50/kernel:0",
                     "model_weights/dense_51/dense_51/bias:0", "model_weights/dense_51/dense_51/kernel:0"]
    assert len(d) == 13 and d["optimizer_weights/Adam/iter:0"].dtype == np.int64 and d["optimizer_weights/Adam/iter:0"].shape == ()
    w = _weights()
    w1, b1, w2, b2 = em_io._dense_weights_from_h5(H5)
    for got, key in ((w1, "W1"), (b1, "b1"), (w2, "W2"), (b2, "b2")):
        assert got.dtype == np.float32 and np.array_equal(got, w[key]), key
    assert w1.shape == (3, 2048) and w2.shape == (2048, 10)


def test_reader_refuses_what_it_does_not_parse(tmp_path):
    bad = tmp_path / "not.h5"
    bad.write_bytes(b"PK\x03\x04" + b"\0" * 100)
    with pytest.raises(hdf5_lite.Hdf5LiteError):
        hdf5_lite.read_datasets(str(bad))
    raw = bytearray(open(H5, "rb").read())
    raw[8] = 2                                  # a version-2 superblock
    v2 = tmp_path / "v2.h5"
    v2.write_bytes(bytes(raw))
    with pytest.raises(hdf5_lite.Hdf5LiteError):
        hdf5_lite.read_datasets(str(v2))


def test_reader_survives_damage_cycles_and_foreign_datasets():
    """A truncated file and a link cycle raise ``Hdf5LiteError`` (not struct.error / RecursionError); ONE dataset outside the
    subset is skipped and named in ``skipped`` while the Dense weights stay readable."""
    import struct
    raw = open(H5, "rb").read()
    with pytest.raises(hdf5_lite.Hdf5LiteError):
        hdf5_lite.read_datasets(data=raw[: len(raw) // 3])                      # cut in the middle of the object headers / heaps
    # make one bias dataset's datatype a string class (class 3): that dataset is skipped, everything else is read
    f = hdf5_lite.File(data=raw)
    assert f.skipped == {} and len(f.datasets) == 13
    target = raw.find(np.float32(f.datasets["model_weights/dense_51/dense_51/bias:0"]).tobytes())
    assert target > 0
    # (find the datatype message of some float32 dataset: class 1, 4 bytes -- patch the FIRST one's class nibble to 3)
    sig = bytes([0x11, 0x20, 0x1F, 0x00]) + struct.pack("<I", 4)
    pos = raw.find(sig)
    assert pos > 0
    patched = bytearray(raw)
    patched[pos] = 0x13
    g = hdf5_lite.File(data=bytes(patched))
    assert len(g.skipped) == 1 and len(g.datasets) == 12
    assert "datatype class 3" in next(iter(g.skipped.values()))
    # a cycle: the root group's first symbol-table entry made to point back at the root object header
    root_header = struct.unpack_from("<Q", raw, 24 + 32 + 8)[0]
    snod = raw.find(b"SNOD")
    cyc = bytearray(raw)
    struct.pack_into("<Q", cyc, snod + 8 + 8, root_header)
    h = hdf5_lite.File(data=bytes(cyc))                                          # terminates; the revisited object adds nothing
    assert len(h.datasets) <= 13


def test_weights_only_file_layout_is_accepted(tmp_path):
    """``model.save_weights`` writes the layer groups at the ROOT (no ``model_weights/`` prefix): same reader, same weights."""
    d = hdf5_lite.read_datasets(H5)
    flat = {k[len("model_weights/"):]: v for k, v in d.items() if k.startswith("model_weights/")}
    import nmma_amd.em.io as io_mod
    orig = io_mod._h5_datasets
    io_mod._h5_datasets = lambda path: flat
    try:
        w1, b1, w2, b2 = io_mod._dense_weights_from_h5("weights_only.h5")
    finally:
        io_mod._h5_datasets = orig
    w = _weights()
    assert np.array_equal(w1, w["W1"]) and np.array_equal(b2, w["b2"])


def test_reference_layout_joblib_plus_h5_end_to_end(tmp_path):
    """``{svd_path}/Bu2019nsbh.joblib`` (the metadata dict of training.py:generate_svd_model, keys with ``_`` for ``:`` as the reference
    stores sncosmo names) + ``{svd_path}/Bu2019nsbh_tf/{filter}.h5``: ``SVDLightCurveModel(svd_path=...)`` converts on the fly, writes
    the flat file next to it and comes up with the trained weights and the basis truncated to n_coeff columns."""
    import joblib
    import shutil
    from nmma_amd.em.model import SVDLightCurveModel
    rng = np.random.default_rng(3)
    nt, nc = 60, 10
    tt = np.linspace(0.0, 21.0, nt)
    meta = {}
    for name in ("ztfr", "ps1__y"):             # (the second name exercises the "_" -> ":" mapping of model.py:604-606)
        q, _ = np.linalg.qr(rng.standard_normal((nt, nt)))
        meta[name] = dict(param_array_postprocess=rng.uniform(size=(28, 3)), param_mins=np.array([-3.0, -3.0, 0.0]),
                          param_maxs=np.array([-1.0, -0.5, 90.0]), mins=-18.0 + rng.random(nt), maxs=-8.0 + rng.random(nt), tt=tt,
                          n_coeff=nc, cAmat=rng.standard_normal((nc, 28)), cAstd=np.ones((nc, 28)), VA=q)
    svd_path = tmp_path / "svdmodels"
    (svd_path / "Bu2019nsbh_tf").mkdir(parents=True)
    joblib.dump(meta, str(svd_path / "Bu2019nsbh.joblib"), compress=9)        # training.py: joblib.dump(self.svd_model, self.modelfile, compress=9)
    shutil.copy(H5, str(svd_path / "Bu2019nsbh_tf" / "ztfr.h5"))
    shutil.copy(H5, str(svd_path / "Bu2019nsbh_tf" / "ps1__y.h5"))
    model = SVDLightCurveModel("Bu2019nsbh_tf", svd_path=str(svd_path), interpolation_type="tensorflow")
    assert len(model.filters) == 2 and "ztfr" in model.filters
    w = _weights()
    for filt, t in model.svd_mag_model.items():
        for key in ("W1", "b1", "W2", "b2"):
            assert t[key].dtype == np.float32 and np.array_equal(t[key], w[key])
        src = meta["ztfr" if filt == "ztfr" else "ps1__y"]
        assert np.array_equal(t["VA"][:, :nc], src["VA"][:, :nc])          # (converted on the fly: the full matrix; from the flat file: n_coeff columns)
        assert np.array_equal(t["mins"], src["mins"]) and np.array_equal(t["tt"], tt) and t["n_coeff"] == nc
    assert os.path.isfile(str(svd_path / "Bu2019nsbh.npz"))
    again = SVDLightCurveModel("Bu2019nsbh_tf", svd_path=str(svd_path), interpolation_type="tensorflow")      # now from the flat file
    assert all(np.array_equal(again.svd_mag_model[f]["W2"], model.svd_mag_model[f]["W2"]) for f in model.svd_mag_model)
    assert all(again.svd_mag_model[f]["VA"].shape == (nt, nc) for f in again.svd_mag_model)
    # a missing network is an error, not a silent zero
    os.remove(str(svd_path / "Bu2019nsbh.npz"))
    os.remove(str(svd_path / "Bu2019nsbh_tf" / "ps1__y.h5"))
    with pytest.raises(FileNotFoundError):
        SVDLightCurveModel("Bu2019nsbh_tf", svd_path=str(svd_path), interpolation_type="tensorflow")


def test_keras_archive_branch_runs_without_keras(tmp_path):
    """``{filter}.keras`` (what the reference saves today and loads FIRST, model.py:635-643): the archive's ``model.weights.h5`` is read
    directly and the two Dense layers are picked by their shapes.  The fixture's layout is restated from Keras' public sources
    (tools/make_keras3_fixture.py) -- this test runs the branch, it does not pin Keras' format."""
    import joblib
    import shutil
    from nmma_amd.em.model import SVDLightCurveModel
    arch = os.path.join(HERE, "golden", "keras3_layout", "ztfr.keras")
    w = _weights()
    for got, key in zip(em_io._dense_weights_from_keras_archive(arch), ("W1", "b1", "W2", "b2")):
        assert np.array_equal(got, w[key]), key
    rng = np.random.default_rng(4)
    nt = 30
    q, _ = np.linalg.qr(rng.standard_normal((nt, nt)))
    meta = {"ztfr": dict(param_mins=np.zeros(3), param_maxs=np.ones(3), mins=-18.0 + rng.random(nt), maxs=-8.0 + rng.random(nt),
                         tt=np.linspace(0.0, 20.0, nt), n_coeff=10, VA=q)}
    root = tmp_path / "m"
    (root / "Bu2019nsbh").mkdir(parents=True)
    joblib.dump(meta, str(root / "Bu2019nsbh.joblib"))
    shutil.copy(arch, str(root / "Bu2019nsbh" / "ztfr.keras"))
    shutil.copy(H5, str(root / "Bu2019nsbh" / "ztfr.h5"))          # (present too: the .keras file wins, as in the reference)
    model = SVDLightCurveModel("Bu2019nsbh", svd_path=str(root), interpolation_type="keras")
    assert np.array_equal(model.svd_mag_model["ztfr"]["W1"], w["W1"]) and np.array_equal(model.svd_mag_model["ztfr"]["b2"], w["b2"])
    bad = tmp_path / "bad.keras"
    import zipfile
    with zipfile.ZipFile(str(bad), "w") as z:
        z.writestr("config.json", "{}")
    with pytest.raises(ValueError):
        em_io._dense_weights_from_keras_archive(str(bad))


def test_factory_builds_what_the_reference_drivers_build(tmp_path):
    """``create_light_curve_model_from_args`` (model.py:1617-1668 with :1591-1614): names -> models, every sub-model on the same
    filters and ``setup_sample_times(args)`` -- compared with the reference's own ``setup_sample_times`` for its three argument forms --
    a host model (the GRB afterglow, third-party arithmetic) wrapped around its ``generate_lightcurve``; a surrogate + host-model
    combination built this way is eligible for the one-launch likelihood."""
    import joblib
    import shutil
    import types
    from nmma_amd.em import utils as amd_utils
    from nmma_amd.em.model import (CombinedLightCurveModelContainer, ExternalLightCurveModel, SimpleKilonovaLightCurveModel,
                                   SVDLightCurveModel, create_light_curve_model_from_args)
    forms = [dict(em_tmin=0.1, em_tmax=20.0, em_tstep=0.5, em_timescale="linear", em_nsteps=100),
             dict(em_tmin=0.05, em_tmax=14.0, em_tstep=None, em_timescale="geometric", em_nsteps=36),
             dict(em_tmin=0.0, em_tmax=10.0, em_tstep=None, em_timescale="log", em_nsteps=21),
             dict(em_tmin=None, em_tmax=None, em_tstep=None, em_timescale="linear", em_nsteps=50)]
    if os.path.isdir("/root/reference/nmma"):
        from oracle import ref_harness
        ref = ref_harness.reference_modules()
        for f in forms:
            want, got = ref.utils.setup_sample_times(types.SimpleNamespace(**f)), amd_utils.setup_sample_times(types.SimpleNamespace(**f))
            assert (want is None and got is None) or np.array_equal(want, got)
    assert np.array_equal(amd_utils.setup_sample_times(types.SimpleNamespace(**forms[0])), np.arange(0.1, 20.5, 0.5))
    with pytest.raises(ValueError):
        amd_utils.setup_sample_times(types.SimpleNamespace(em_tmin=1.0, em_tmax=2.0, em_tstep=None, em_timescale="cubic", em_nsteps=5))

    rng = np.random.default_rng(4)
    nt, nc = 60, 10
    meta = {}
    for name in ("ztfr", "ztfg"):
        q, _ = np.linalg.qr(rng.standard_normal((nt, nt)))
        meta[name] = dict(param_array_postprocess=rng.uniform(size=(28, 3)), param_mins=np.array([-3.0, -3.0, 0.0]),
                          param_maxs=np.array([-1.0, -0.5, 90.0]), mins=-18.0 + rng.random(nt), maxs=-8.0 + rng.random(nt),
                          tt=np.linspace(0.0, 21.0, nt), n_coeff=nc, cAmat=rng.standard_normal((nc, 28)), cAstd=np.ones((nc, 28)), VA=q)
    svd_path = tmp_path / "svdmodels"
    (svd_path / "Bu2019nsbh_tf").mkdir(parents=True)
    joblib.dump(meta, str(svd_path / "Bu2019nsbh.joblib"), compress=9)
    for name in meta:
        shutil.copy(H5, str(svd_path / "Bu2019nsbh_tf" / f"{name}.h5"))

    class HostGRB:                       # stands for the reference's GRBLightCurveModel (afterglowpy behind generate_lightcurve)
        model_parameters = ["log10_E0", "thetaCore"]
        gap_free = True

        def generate_lightcurve(self, sample_times, parameters):
            return {f: np.full(len(sample_times), -15.0) for f in ("ztfr", "ztfg")}

    args = types.SimpleNamespace(svd_path=str(svd_path), svd_mag_ncoeff=None, interpolation_type="tensorflow", em_extinction_law=None, **forms[0])
    comb = create_light_curve_model_from_args("Bu2019nsbh_tf, TrPi2018", args, filters=["ztfr", "ztfg"], host_models={"TrPi2018": HostGRB()})
    assert isinstance(comb, CombinedLightCurveModelContainer) and comb.model == ["Bu2019nsbh", "TrPi2018"]        # (the "_tf" suffix is stripped as in model.py:596-599)
    kn, grb = comb.lc_models
    assert isinstance(kn, SVDLightCurveModel) and isinstance(grb, ExternalLightCurveModel) and grb.gap_free
    assert np.array_equal(kn.model_times, np.arange(0.1, 20.5, 0.5)) and np.array_equal(grb.model_times, kn.model_times)
    assert grb.model_parameters == ["log10_E0", "thetaCore"] and list(grb.filters) == ["ztfr", "ztfg"]
    plan = comb.stack2_plan()
    assert plan is not None and plan[0] is kn and plan[1] is grb                 # one launch per batch (nmma_em_loglike_stack2)
    lc, ok = grb.generate_batch(np.zeros((3, 2)), ["log10_E0", "thetaCore"])
    assert lc.shape == (3, 2, 41) and ok.all() and np.all(lc == -15.0)
    single = create_light_curve_model_from_args("Me2017", args, filters=["g", "r", "i"])
    assert isinstance(single, SimpleKilonovaLightCurveModel) and len(single.model_times) == 41
    one = create_light_curve_model_from_args(["Bu2019nsbh_tf"], args, filters=["ztfr"])
    assert isinstance(one, SVDLightCurveModel) and list(one.filters) == ["ztfr"]
    # (advisor, round 5) --filters from the arguments when the caller passes none (model.py:1618-1619) ...
    args_f = types.SimpleNamespace(**dict(vars(args), filters="ztfr, ztfg"))
    from_args = create_light_curve_model_from_args("Bu2019nsbh_tf", args_f)
    assert list(from_args.filters) == ["ztfr", "ztfg"]
    # ... and a name the reference resolves to a host-side model (model.py:1572-1579) needs its host object: a clear error, not a
    # fall-through to a surrogate that fails on a missing file
    for name in ("TrPi2018", "Piro2021", "PL_BB_fixedT", "Sr2023", "Arnett"):
        with pytest.raises(ValueError, match="host_models"):
            create_light_curve_model_from_args(f"Bu2019nsbh_tf,{name}", args, filters=["ztfr"])
