"""CPU tests of the host-side layer: parameter-slot resolution, filter maps, systematics
selection (against the reference's own handler when /root/reference is present), model
file round trip, sharding, and that the C-ABI library loads and exports every symbol."""
import os
import re

import numpy as np
import pytest

from nmma_amd import _lib as L
from nmma_amd.em import io as em_io
from nmma_amd.em import utils
from nmma_amd.em.systematics import FilterSystematicsHandler
from nmma_amd.engine import resolve_model_param_slot
from nmma_amd.parallel import shard_bounds
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nmma_hip.h")).read()
    declared = set(re.findall(r"\b(nmma_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    lib = L.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.nmma_abi_version() == L.ABI_VERSION
    assert b"gfx950" in lib.nmma_build_info()


def test_config_struct_matches_header_field_order():
    hdr = open(os.path.join(ROOT, "include", "nmma_hip.h")).read()
    body = hdr[hdr.index("typedef struct nmma_em_config {"):hdr.index("} nmma_em_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b([A-Za-z_0-9]+)(?:\[[A-Z_]+\])?;", body)
    assert fields == [f[0] for f in L.EmConfig._fields_]


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tests.helpers import engine_from_case
    with pytest.raises(L.NMMAHipError):
        engine_from_case(cases.case_small_hidden())


def test_slot_resolution_follows_reference_conversions():
    names = ["luminosity_distance", "inclination_EM", "log10_mej_dyn", "mej_wind"]
    s = resolve_model_param_slot("KNtheta", names, {})
    assert (s.col, s.op) == (1, L.OP_RAD2DEG)
    s = resolve_model_param_slot("log10_mej_dyn", names, {})
    assert (s.col, s.op) == (2, L.OP_IDENT)
    s = resolve_model_param_slot("log10_mej_wind", names, {})     # log10 of a sampled linear mass
    assert (s.col, s.op) == (3, L.OP_LOG10)
    s = resolve_model_param_slot("mej_dyn", names, {})            # 10 ** sampled log10
    assert (s.col, s.op) == (2, L.OP_POW10)
    s = resolve_model_param_slot("KNphi", names, {"KNphi": 30.0})
    assert s.col == -1 and s.value == 30.0
    s = resolve_model_param_slot("KNtheta", ["theta_jn"], {})
    assert s.op == L.OP_THETAJN2DEG
    s = resolve_model_param_slot("KNtheta", ["x"], {})            # default: face-on
    assert s.col == -1 and s.value == 0.0
    with pytest.raises(KeyError):
        resolve_model_param_slot("KNphi", names, {})


def test_filter_maps():
    direct, avg = utils.get_filter_name_mapping(["g", "B", "w", "ps1::r", "radio-3GHz", "F814W"], ["ps1::r"])
    assert direct == {"g": "g", "B": "g", "ps1::r": "ps1::r", "radio-3GHz": "radio-3GHz"}
    assert avg == {"w": ["g", "r", "i"], "F814W": ["z", "y"]}
    with pytest.raises(ValueError):
        utils.get_filter_name_mapping(["nope"])
    src = utils.resolve_sources(["g", "r", "i", "w"], ["g", "r", "i"])
    assert src["w"] == ["g", "r", "i"] and src["g"] == ["g"]
    with pytest.raises(KeyError):                                  # helper bands must be observed
        utils.resolve_sources(["g", "w"], ["g", "r", "i"])


@pytest.mark.parametrize("name", ["c2_default", "syserr_param", "syserr_time_nodes"])
def test_systematics_selection_matches_case_spec(name):
    case = cases.CASES[name]()
    ref_kw = case["systematics_ref"]
    h = FilterSystematicsHandler(case["observed_filters"], systematics_file=ref_kw["systematics_file"],
                                 error_budget=ref_kw["error_budget"], light_curve_times=case["data"][0])
    h.reset(np.array([0.0, 21.0]), {n: object() for n in case["names"]})
    spec, want = h.kernel_spec(), case["systematics"]
    if want["mode"] == "budget":
        assert spec == want
    elif want["mode"] == "param":
        assert spec == want
    else:
        assert spec["names"] == want["names"]
        assert set(spec["nodes"]) == set(want["nodes"])
        for f, (names, nodes) in want["nodes"].items():
            assert spec["nodes"][f][0] == names
            np.testing.assert_array_equal(spec["nodes"][f][1], nodes)
    # per-sample API agrees with the oracle's evaluator
    from oracle.nmma_oracle import OracleSystematics
    p = dict(zip(case["names"], case["theta"][0]))
    got, exp = h(p), OracleSystematics(want, case["observed_filters"], case["data"][0])(p)
    for f in case["observed_filters"]:
        np.testing.assert_allclose(got[f], exp[f], rtol=1e-15)


@pytest.mark.skipif(not os.path.isdir("/root/reference/nmma"), reason="reference tree not available")
def test_systematics_selection_matches_reference_handler():
    from oracle import ref_harness
    ref = ref_harness.reference_modules()
    case = cases.case_syserr_time_nodes()
    kw = case["systematics_ref"]
    priors = {n: object() for n in case["names"]}
    r = ref.systematics.FilterSystematicsHandler(list(case["observed_filters"]),
                                                 systematics_file=kw["systematics_file"],
                                                 error_budget=kw["error_budget"],
                                                 light_curve_times=case["data"][0])
    r.reset(np.array([0.0, 21.0]), priors)
    h = FilterSystematicsHandler(case["observed_filters"], systematics_file=kw["systematics_file"],
                                 error_budget=kw["error_budget"], light_curve_times=case["data"][0])
    h.reset(np.array([0.0, 21.0]), priors)
    assert h.direct_sys_map == r.direct_sys_map
    assert set(h.interpolate_map) == set(r.interpolate_map)
    p = dict(zip(case["names"], case["theta"][3]))
    a, b = h(p), r(p)
    for f in case["observed_filters"]:
        np.testing.assert_allclose(a[f], b[f], rtol=1e-15)


def test_legacy_yaml_systematics(tmp_path):
    cfg = {"config": {"withTime": {"value": True, "filters": [["g", "r"], "i"], "time_nodes": 3,
                                   "type": "Uniform", "minimum": 0, "maximum": 2},
                      "withoutTime": {"value": False}}}
    names = ["em_syserr_g___r_0", "em_syserr_g___r_1", "em_syserr_g___r_2",
             "em_syserr_i_0", "em_syserr_i_1", "em_syserr_i_2"]
    h = FilterSystematicsHandler(["g", "r", "i"], systematics_file=cfg, light_curve_times=np.linspace(1, 9, 5))
    h.reset(np.array([0.1, 20.0]), {n: 1 for n in names})
    spec = h.kernel_spec()
    assert spec["nodes"]["g"][0] == names[:3] and spec["nodes"]["i"][0] == names[3:]
    np.testing.assert_array_equal(spec["nodes"]["r"][1], np.round(np.linspace(0.1, 20.0, 3), 2))


def test_model_file_round_trip(tmp_path):
    case = cases.case_small_hidden()
    path = tmp_path / "m.npz"
    em_io.save_svd_model(path, case["svd"], case["model_parameters"])
    svd, params = em_io.load_svd_model(path)
    assert params == case["model_parameters"] and set(svd) == set(case["svd"])
    for f in svd:
        for k in ("W1", "b1", "W2", "b2", "VA", "mins", "maxs", "tt"):
            np.testing.assert_array_equal(svd[f][k], np.asarray(case["svd"][f][k])[..., :svd[f][k].shape[-1]]
                                          if k == "VA" else case["svd"][f][k])


def test_photometry_reader(tmp_path):
    p = tmp_path / "lc.dat"
    p.write_text("2017-08-18T00:00:00.000 ps1::g 17.41 0.02\n"
                 "2017-08-19T12:00:00.000 ps1::g 18.41 inf\n"
                 "57984.5 ps1::r 17.9 0.05\n")
    d = em_io.load_em_observations(str(p))
    assert d["ps1::g"]["time"][0] == pytest.approx(57983.0) and np.isinf(d["ps1::g"]["mag_error"][1])
    assert d["ps1::r"]["time"][0] == 57984.5
    times, mags, sig, t0 = utils.setup_filtered_lc_data(d, 57982.5)
    assert times["ps1::g"][0] == pytest.approx(0.5)


@pytest.mark.skipif(not os.path.isfile("/root/reference/example_files/lightcurves/AT2017gfo.dat"),
                    reason="reference example data not available")
def test_reads_reference_example_photometry():
    d = em_io.load_em_observations("/root/reference/example_files/lightcurves/AT2017gfo.dat")
    assert {k: len(v["time"]) for k, v in d.items()}["ps1::g"] == 13
    assert sum(len(v["time"]) for v in d.values()) == 141
    assert sum(np.isinf(v["mag_error"]).sum() for v in d.values()) == 3


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 4096, 4099):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_native_cosmology_is_monotone_and_close_to_hubble_law():
    from nmma_amd.core.conversion import get_cosmo_grids
    dg, zg = get_cosmo_grids(1.0, 200.0)
    assert len(dg) == 50 and np.all(np.diff(dg) > 0) and np.all(np.diff(zg) > 0)
    assert dg[0] == pytest.approx(1.0, rel=1e-9) and dg[-1] == pytest.approx(200.0, rel=1e-9)
    np.testing.assert_allclose(zg, dg * 67.66 / 299792.458, rtol=0.04)
