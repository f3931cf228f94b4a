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


class _AnyPrior(dict):
    """A prior set that contains every name (the documents below only need membership)."""
    def __contains__(self, key):
        return True


SYS_DOCS = {
    "global_scalar": {"prior": "Uniform(0, 2)"},
    "global_nodes": {"time_nodes": 4, "time_range": "lin 2 12", "prior": "Uniform(0, 2)"},
    "global_range_log": {"time_range": "log 0.5 20 5"},
    "global_range_start_end": {"time_range": "1.0 15.0 3"},
    "global_range_spacing_end": {"time_range": "geom 15.0 3"},
    "per_filter_and_rest": {"g": {"time_range": "1 9 3"}, "rest": {"prior": "x"}},
    "group_each_rest": {"opt": {"filters": ["g", "r"], "time_range": "lin 1 10 4"},
                        "nir": {"each": ["i"], "time_nodes": 2, "time_range": "log 19"}, "other": {}},
    "two_gap_fillers": {"a": {"time_range": "0.2 8 3"}, "b": {}},
    "legacy_all": {"config": {"withTime": {"value": True, "filters": [None], "time_nodes": 4},
                              "withoutTime": {"value": False}}},
    "legacy_flat": {"config": {"withTime": {"value": False, "filters": [None], "time_nodes": 4},
                               "withoutTime": {"value": True}}},
}


@pytest.mark.skipif(not os.path.isdir("/root/reference/nmma"), reason="reference tree not available")
@pytest.mark.parametrize("doc", sorted(SYS_DOCS))
def test_systematics_documents_match_reference_handler(doc):
    """Every document form the reference accepts gives the same prior names, node grids and
    per-sample sigma_sys as the reference's own handler."""
    from oracle import ref_harness
    ref = ref_harness.reference_modules()
    filters = ["g", "r", "i", "z"]
    times = {f: np.linspace(0.3 + k, 18.0, 7 + k) for k, f in enumerate(filters)}
    priors = _AnyPrior()
    span = np.array([0.1, 20.0])
    r = ref.systematics.FilterSystematicsHandler(list(filters), systematics_file=SYS_DOCS[doc], light_curve_times=times)
    r.reset(span, priors)
    h = FilterSystematicsHandler(filters, systematics_file=SYS_DOCS[doc], light_curve_times=times)
    h.reset(span, priors)
    rng = np.random.default_rng(5)
    used = set()
    for f, e in h.table.items():
        used.update(e.names)
    for f, n in r.direct_sys_map.items():
        used.add(n)
    for f, (names, _) in r.interpolate_map.items():
        used.update(names)
    used.add(r.base_prior_name)
    p = {n: float(rng.uniform(0.1, 2.0)) for n in sorted(used)}
    a, b = h(p), r(p)
    assert set(a) == set(b) == set(filters)
    for f in filters:
        np.testing.assert_allclose(a[f], b[f], rtol=1e-15, err_msg=f"{doc}/{f}")
    # and the kernel table describes the same thing
    spec = h.kernel_spec()
    if spec["mode"] == "param":
        for f in filters:
            np.testing.assert_array_equal(a[f], np.full_like(times[f], p[spec["name"]]))


def test_bare_time_nodes_span_the_model_range():
    """``time_nodes: n`` without a ``time_range``: the reference's get_time_range leaves the
    spacing unset for this form (UnboundLocalError, systematics.py:141-146); here it means n
    linear nodes over the model's time range, which is what its branch sets up."""
    h = FilterSystematicsHandler(["g", "r"], systematics_file={"time_nodes": 4})
    h.reset(np.array([0.5, 12.5]), _AnyPrior())
    names, nodes = h.interpolate_map["r"]
    assert names == [f"em_syserr_{i}" for i in range(4)]
    np.testing.assert_array_equal(nodes, np.linspace(0.5, 12.5, 4))


def test_filter_addressed_twice_is_rejected():
    doc = {"all": {"filters": ["g", "r"]}, "r": {"time_range": "0.2 8 3"}}
    h = FilterSystematicsHandler(["g", "r"], systematics_file=doc)
    with pytest.raises(KeyError):                       # the reference: set.remove KeyError (:269)
        h.reset(np.array([0.5, 12.5]), _AnyPrior())


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


# ---------------------------------------------------------------------------------------------------
# model-window check, AnBa2022 conversion, posterior conversion, batched constraints
# ---------------------------------------------------------------------------------------------------
class _Bounds:
    def __init__(self, lo, hi):
        self.minimum, self.maximum = lo, hi


class _Grid:
    model_times = np.linspace(0.1, 14.0, 50)


def _window_data(t_first=0.5, t_last=10.0):
    t = {"g": np.array([t_first, 2.0, t_last]), "r": np.array([0.05, 3.0, 30.0])}
    m = {"g": np.array([18.0, 19.0, 20.0]), "r": np.array([21.0, 19.5, 22.0])}
    # the r-band points outside the model are non-detections: they do not count towards the span
    e = {"g": np.array([0.1, 0.1, 0.1]), "r": np.array([np.inf, 0.1, np.inf])}
    return (t, m, e, 57982.5)


@pytest.mark.parametrize("use_reference", [False, True])
def test_check_model_time_consistency(use_reference):
    if use_reference:
        if not os.path.isdir("/root/reference/nmma"):
            pytest.skip("reference tree not available")
        from oracle import ref_harness
        check = ref_harness.reference_modules().utils.check_model_time_consistency
    else:
        check = utils.check_model_time_consistency
    priors = {"redshift": _Bounds(0.0, 0.1), "timeshift": _Bounds(-0.2, 0.3)}
    # window: start (1+0.1)*0.1+0.3 = 0.41, end (1+0)*14-0.2 = 13.8
    out = check(_window_data(), _Grid(), priors)
    np.testing.assert_array_equal(out[0]["g"], [0.5, 2.0, 10.0])
    with pytest.raises(ValueError, match="First data point"):
        check(_window_data(t_first=0.4), _Grid(), priors)
    with pytest.raises(ValueError, match="Last data point"):
        check(_window_data(t_last=13.9), _Grid(), priors)
    # an injection is trimmed to the window instead
    cut = check(_window_data(t_first=0.4, t_last=13.9), _Grid(), priors, injection={"x": 1})
    np.testing.assert_array_equal(cut[0]["g"], [2.0])
    np.testing.assert_array_equal(cut[0]["r"], [3.0])
    np.testing.assert_array_equal(cut[1]["r"], [19.5])
    assert cut[3] == 57982.5


def test_observer_frame_window_from_distance_priors():
    from nmma_amd.core import conversion
    priors = {"luminosity_distance": _Bounds(20.0, 400.0)}
    lo, hi = utils.observer_frame_window(_Grid.model_times, priors)
    z_lo, z_hi = (conversion.luminosity_distance_to_redshift(d) for d in (20.0, 400.0))
    assert lo == pytest.approx((1 + z_hi) * 0.1) and hi == pytest.approx((1 + z_lo) * 14.0)
    # a sampled Hubble constant moves both ends (utils.py:304-311: all minima, then all maxima)
    priors["Hubble_constant"] = _Bounds(60.0, 80.0)
    lo2, hi2 = utils.observer_frame_window(_Grid.model_times, priors)
    z_a = conversion.cosmology_to_distance({"luminosity_distance": 20.0, "Hubble_constant": 60.0})["redshift"]
    z_b = conversion.cosmology_to_distance({"luminosity_distance": 400.0, "Hubble_constant": 80.0})["redshift"]
    assert lo2 == pytest.approx((1 + z_b) * 0.1) and hi2 == pytest.approx((1 + z_a) * 14.0)
    assert z_b > z_hi                                   # larger H0 -> larger z at the same distance


def test_convert_mtot_mni_matches_reference():
    from nmma_amd.core.conversion import convert_mtot_mni
    rng = np.random.default_rng(2)
    cols = {"log10_mtot": rng.uniform(-0.5, 1.0, 9), "log10_mni": rng.uniform(-2.0, -0.5, 9),
            "log10_mrp": rng.uniform(-2, 0, 9), "xmix": rng.uniform(0, 1, 9), "vej": rng.uniform(3, 9, 9)}
    got = convert_mtot_mni({k: v.copy() for k, v in cols.items()})
    np.testing.assert_array_equal(got["mni_c"], 10 ** cols["log10_mni"] / 10 ** cols["log10_mtot"])
    np.testing.assert_array_equal(
        got["mrp_c"], cols["xmix"] * (10 ** cols["log10_mtot"] - 10 ** cols["log10_mni"]) - 10 ** cols["log10_mrp"])
    if os.path.isdir("/root/reference/nmma"):
        from oracle import ref_harness
        ref = ref_harness.reference_modules().conversion.convert_mtot_mni({k: v.copy() for k, v in cols.items()})
        for key in ("mni", "mtot", "mrp", "mni_c", "mrp_c"):
            np.testing.assert_array_equal(got[key], ref[key])


def test_constraint_set_rows_and_scalars():
    from nmma_amd.core.base import LOGL_FLOOR, Constraint, ConstraintSet, floor_rows
    priors = {"mni_c": Constraint(minimum=0.0, maximum=1.0, name="mni_c"), "mrp_c": Constraint(minimum=0.0, maximum=np.inf, name="mrp_c"),
              "vej": _Bounds(1, 2)}
    cons = ConstraintSet.of(priors)
    assert sorted(cons) == ["mni_c", "mrp_c"]
    assert bool(cons.mask({"mni_c": 0.5, "mrp_c": 2.0})) and not bool(cons.mask({"mni_c": 1.5, "mrp_c": 2.0}))
    keep = cons.mask({"mni_c": np.array([0.5, 1.5, 0.2]), "mrp_c": np.array([1.0, 1.0, -1.0])})
    np.testing.assert_array_equal(keep, [True, False, False])
    np.testing.assert_array_equal(floor_rows(np.array([-1.0, -2.0, -3.0]), keep), [-1.0, LOGL_FLOOR, LOGL_FLOOR])
    assert LOGL_FLOOR == float(np.nan_to_num(-np.inf))
    assert not ConstraintSet.of({})                     # empty: nothing to evaluate


def test_anba2022_likelihood_adds_mass_conversion_and_wing_posterior():
    """em_likelihood.py:91-100 and :122-131 on the host side (no engine is built)."""
    from nmma_amd.core.conversion import convert_mtot_mni
    from nmma_amd.em.em_likelihood import EMTransientLikelihood

    class _Model:
        def __init__(self, name):
            self.model = name

        def parameter_conversion(self, p):
            return p

    class _Sub:
        def __init__(self, name):
            self.light_curve_model = _Model(name)

    for name, expect in (("AnBa2022_log", True), (["Bu2019lm", "AnBa2022_linear"], True), ("Bu2019lm", False)):
        lik = EMTransientLikelihood.__new__(EMTransientLikelihood)
        lik.sub_model, lik.conv_functions = _Sub(name), []
        lik.setup_submodel_conversion()
        assert (convert_mtot_mni in lik.conv_functions) is expect
        assert lik.conv_functions[-1] == lik.sub_model.light_curve_model.parameter_conversion
    post = {"thetaWing": np.array([0.2, 0.3]), "thetaCore": np.array([0.1, 0.1]),
            "log10_mej_dyn": np.array([-2.0, -2.0]), "log10_mej_wind": np.array([-2.0, -1.0])}
    out = lik.posterior_conversion(dict(post))
    np.testing.assert_allclose(out["alphaWing"], [2.0, 3.0])
    np.testing.assert_allclose(out["log10_mej"], np.log10([0.02, 0.11]))
    out = lik.posterior_conversion({"alphaWing": np.array([2.0]), "thetaCore": np.array([0.1])})
    np.testing.assert_allclose(out["thetaWing"], [0.2])


def test_hubble_constant_conversion_and_device_refusal():
    """core/base.py:161-164: a sampled H0 adds cosmology_to_distance to the conversions; the device path,
    which tabulates ONE cosmology, refuses it loudly instead of silently using Planck18."""
    from nmma_amd import _lib as L
    from nmma_amd.core.base import NMMALikelihood
    from nmma_amd.core.conversion import cosmology_to_distance
    from nmma_amd.em.em_likelihood import MultiFilterTransient

    class _Sub:
        def log_likelihood(self, p):
            return -0.5 * p["redshift"]

    priors = {"luminosity_distance": _Bounds(10, 100), "Hubble_constant": _Bounds(60, 80)}
    lik = NMMALikelihood(_Sub(), priors)
    lik.setup_parameter_conversion()
    lik.setup_parameter_conversion()                    # idempotent
    assert lik.conv_functions == [cosmology_to_distance]
    a = lik.log_likelihood({"luminosity_distance": 40.0, "Hubble_constant": 60.0})
    b = lik.log_likelihood({"luminosity_distance": 40.0, "Hubble_constant": 80.0})
    assert b < a < 0                                    # larger H0 -> larger z at fixed distance
    assert lik.post_process_bestfit(None, bestfit_params={"luminosity_distance": 40.0, "Hubble_constant": 70.0}) is None
    mft = MultiFilterTransient.__new__(MultiFilterTransient)
    mft._engine = mft._names = None
    # (a sampled Hubble constant runs on the device since round 3: tests/test_gpu_plugin.py; a sampled matter density does not)
    with pytest.raises(L.NMMAHipError, match="Omega_matter"):
        mft.engine(["luminosity_distance", "Hubble_constant", "Omega_matter", "log10_mej"])


def test_multimessenger_conversion_chain_and_joint_setup():
    """``MultimessengerConversion`` (core/conversion.py:768-824): the reference's order (cosmo, gw, eos, ejecta, em, custom), scalar
    unwrapping, ``add_new_keys``, tuple-returning converters, arrays for the batched path; ``MultiMessengerLikelihood`` switches to it
    when ``conversion_instructions`` are given (joint_likelihood.py:42-58, :72-73); the ejecta fits themselves are not built."""
    from nmma_amd.core.conversion import MultimessengerConversion
    from nmma_amd.joint.joint_likelihood import ExternalLogLikelihood, MultiMessengerLikelihood
    order = []

    def step(tag, **add):
        def f(p):
            order.append(tag)
            return dict(p, **{k: (v(p) if callable(v) else v) for k, v in add.items()})
        return f

    chain = MultimessengerConversion.from_dict({
        "custom": step("custom", done=1), "em": lambda p: (step("em", KNtheta=lambda q: q["inc"] * 2)(p), ["KNtheta"]),
        "ejecta": step("ejecta", log10_mej_dyn=-2.0), "eos": step("eos", lambda_1=400.0), "gw": step("gw", chi_eff=0.0)})
    out, added = chain.convert_to_multimessenger_parameters({"inc": np.array([0.25])}, add_new_keys=True)
    assert order == ["gw", "eos", "ejecta", "em", "custom"]
    assert out["KNtheta"] == 0.5 and isinstance(out["inc"], float) and added == ["chi_eff", "lambda_1", "log10_mej_dyn", "KNtheta", "done"]
    batch = chain.convert_to_multimessenger_parameters({"inc": np.linspace(0.0, 1.0, 5)})
    assert np.allclose(batch["KNtheta"], np.linspace(0.0, 2.0, 5))
    with pytest.raises(NotImplementedError):
        MultimessengerConversion.from_dict({"ejecta": True})
    with pytest.raises(NotImplementedError):
        MultimessengerConversion.from_args(None)

    ext = ExternalLogLikelihood("gw", lambda p: -1.0)
    mm = MultiMessengerLikelihood([ext], {}, conversion_instructions={"custom": lambda p: dict(p, total=p["a"] + p["b"])})
    assert mm.multi_conversion is not None and mm.parameter_conversion({"a": 1.0, "b": 2.0})["total"] == 3.0
    plain = MultiMessengerLikelihood([ext], {})
    assert plain.multi_conversion is None and plain.parameter_conversion({"a": 1.0}) == {"a": 1.0}


def test_combined_model_one_launch_plan_is_host_logic():
    """``CombinedLightCurveModelContainer.stack2_plan``: two sub-models on ONE grid and filter list, one of them an SVD surrogate ->
    (surrogate, other) and engine arguments that carry the COMBINATION's grid / cosmology / extinction with ``stack_operands=1``;
    anything else (own grids or filter lists, three sub-models, no surrogate) -> None.  No GPU involved."""
    from nmma_amd import synthetic as syn
    from nmma_amd.em.model import CombinedLightCurveModelContainer, ExternalLightCurveModel, SVDLightCurveModel
    filters = ["ps1::g", "ps1::r", "2massj"]
    mp, svd = syn.make_svd_model(11, filters, model="Bu2019lm")
    st = np.arange(0.1, 20.5, 0.5)
    grid = syn.flat_lcdm_grid(1.0, 200.0)
    kn = SVDLightCurveModel("Bu2019lm", svd_mag_model=svd, filters=filters, model_parameters=mp, sample_times=st, cosmo_grid=grid)
    grb = ExternalLightCurveModel("GRB", filters, st)
    comb = CombinedLightCurveModelContainer([grb, kn], cosmo_grid=grid)          # (order does not matter: the flux sum is symmetric)
    assert comb.stack2_plan() == (kn, grb)
    kw = comb.stack2_engine_kwargs()
    assert kw["stack_operands"] == 1 and kw["svd_model"] is kn.svd_mag_model and np.array_equal(kw["sample_times"], st)
    assert kw["cosmo_grid"] is grid and "extinction_law" not in kw
    assert comb.stack2_union() == (None, None) and "base_times" not in kw
    # own time grids (round 6): still one launch -- the surrogate's own grid becomes base_times, the other sub-model's curves are regridded
    other_grid = ExternalLightCurveModel("GRB", filters, np.geomspace(0.2, 30.0, 30))
    own = CombinedLightCurveModelContainer([kn, other_grid], cosmo_grid=grid)
    assert own.stack2_plan() == (kn, other_grid)
    base, plan2 = own.stack2_union()
    assert np.array_equal(base, st) and plan2 == [[0], [1], [2]]
    kw_own = own.stack2_engine_kwargs()
    assert np.array_equal(kw_own["base_times"], st) and np.array_equal(kw_own["sample_times"], own.model_times) and len(own.model_times) == len(st) + 30
    # a second sub-model that lists fewer filters: no flux there, still one launch; the surrogate already on the combination's grid
    other_filters = ExternalLightCurveModel("GRB", filters[:2], st)
    fewer = CombinedLightCurveModelContainer([kn, other_filters], cosmo_grid=grid)
    assert fewer.stack2_plan() == (kn, other_filters) and fewer.stack2_union()[0] is None and fewer.stack2_union()[1] == [[0], [1], []]
    # a filter the SURROGATE does not list: the materialising path
    more_filters = ExternalLightCurveModel("GRB", filters + ["sdssu"], st)
    assert CombinedLightCurveModelContainer([kn, more_filters], cosmo_grid=grid).stack2_plan() is None
    # ... but a filter it LISTS without having a network for it (calc_svd_lc's null output for radio / X-ray bands,
    # lightcurve_generation.py:168-169) rides along as a null filter of the one-launch engine
    kn_x = SVDLightCurveModel("Bu2019lm", svd_mag_model=svd, filters=filters + ["X-ray-1keV"], model_parameters=mp, sample_times=st, cosmo_grid=grid)
    grb_x = ExternalLightCurveModel("GRB", filters + ["X-ray-1keV"], st)
    joint = CombinedLightCurveModelContainer([kn_x, grb_x], cosmo_grid=grid)
    assert joint.stack2_plan() == (kn_x, grb_x) and joint.stack2_union() == (None, None)
    kw_x = joint.stack2_engine_kwargs()
    assert kw_x["model_filters"] == filters + ["X-ray-1keV"] and kw_x["null_filters"] == ["X-ray-1keV"] and "base_times" not in kw_x
    assert CombinedLightCurveModelContainer([kn, grb, ExternalLightCurveModel("SN", filters, st)], cosmo_grid=grid).stack2_plan() is None
    assert CombinedLightCurveModelContainer([grb, ExternalLightCurveModel("SN", filters, st)], cosmo_grid=grid).stack2_plan() is None


def test_kernel_register_budget():
    """Code-object metadata of the built library: the lean kernels (em_logl<.., 1>, <.., 3>, <.., 4>, <.., 5>, <.., 6>) must not spill -- a change
    that made hipcc spill 2 600 registers in one of them went unnoticed by the parity tests and cost 50 % of its speed --
    and no kernel may use more than a few words of scratch."""
    import re
    import subprocess
    from nmma_amd import _lib
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(readelf) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("llvm-readelf or the built library not available")
    data = open(_lib.LIB_PATH, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
    assert len(starts) >= 2, "no embedded device code object found"
    import tempfile
    notes = ""
    for start in starts[1:]:          # one embedded code object per translation unit (em_logl's instantiations are spread over several)
        with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
            tmp.write(data[start:])
            tmp.flush()
            n = subprocess.run([readelf, "--notes", tmp.name], capture_output=True, text=True).stdout
        if "amdhsa.kernels" in n:
            notes += n
    kernels = {}
    name = None
    for line in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
        for key in ("private_segment_fixed_size", "vgpr_spill_count", "vgpr_count"):
            m = re.search(r"\." + key + r":\s+(\d+)", line)
            if m and name:
                kernels[name][key] = int(m.group(1))
    logl = {k: v for k, v in kernels.items() if "7em_loglI" in k}
    assert len(logl) >= 64, sorted(kernels)
    seen = set()
    for k, v in logl.items():
        # template arguments <R, KP, NMW, NVW, FASTM, WALKF>
        targs = [int(a) for a in re.findall(r"Li(\d+)E", re.search(r"7em_loglI((?:Li\d+E)+)E", k).group(1))]
        assert len(targs) == 6, k
        fastm = targs[4]
        seen.add(fastm)
        if fastm in (1, 3, 4, 6):      # (the fused MCMC step's instantiations keep a few words of scratch for the walk's state, no spills --
            # except the ones that carry the Constraint interpreter, WALKF & 64: two dozen registers around its libm calls, in the epilogue)
            con = (targs[5] & 64) != 0
            assert v["vgpr_spill_count"] <= (32 if con else 0) and (v["private_segment_fixed_size"] == 0 or targs[5] != 0), (k, v)
        if fastm in (7, 8):     # (the combined-model flavours with the finite-limit block of round 6: none / two registers on the unequally spaced grids)
            assert v["vgpr_spill_count"] <= 4 and v["private_segment_fixed_size"] <= 32, (k, v)
        if fastm == 5:      # (the general lean task with its out-of-line limit terms: two registers; with the fused MCMC step up to ten)
            assert v["vgpr_spill_count"] <= (12 if targs[5] else 4), (k, v)
        assert v["private_segment_fixed_size"] <= (160 if (targs[5] & 64) else 64), (k, v)
        assert v["vgpr_count"] <= (128 if fastm else 160), (k, v)
    assert seen >= {0, 1, 2, 3, 4, 5, 6, 7}, seen


def test_external_model_generator_builds_the_operand_row_by_row():
    """``ExternalLightCurveModel(generate_lightcurve=...)``: the reference-shaped host callable (model.py:405-408) is called once per
    row with sampled + fixed + converted parameters; a falsy answer marks the row as failed (model.py:1423-1426), a filter the callable
    does not return carries no flux; entries the caller supplies in ``external_lc`` are left alone."""
    from nmma_amd.em.model import CombinedLightCurveModelContainer, ExternalLightCurveModel
    st = np.linspace(0.5, 10.0, 7)
    seen = []

    def gen(sample_times, p):
        seen.append(dict(p))
        if p["grb_mag0"] > -14.5:
            return {}
        return {"g": p["grb_mag0"] + p["grb_slope"] * np.log10(sample_times), "r": np.full(len(sample_times), p["offset"])}

    ext = ExternalLightCurveModel("PLGRB", ["g", "r", "i"], st, model_parameters=["grb_mag0", "grb_slope"], generate_lightcurve=gen)
    other = ExternalLightCurveModel("OTHER", ["g", "r", "i"], st)
    comb = CombinedLightCurveModelContainer([other, ext])
    theta = np.array([[40.0, 0.3, -16.0, 1.0], [40.0, 0.3, -14.0, 1.2], [80.0, 0.1, -15.0, 0.9]])
    names = ["luminosity_distance", "inclination_EM", "grb_mag0", "grb_slope"]
    given = {"OTHER": np.zeros((3, 3, 7))}
    ops = comb.host_operands(theta, names, {"offset": -12.5}, given)
    assert ops["OTHER"] is given["OTHER"] and set(ops) == {"OTHER", "PLGRB"}
    lc, ok = ops["PLGRB"]
    assert lc.shape == (3, 3, 7) and ok.tolist() == [True, False, True]
    np.testing.assert_array_equal(lc[0, 0], -16.0 + 1.0 * np.log10(st))
    np.testing.assert_array_equal(lc[2, 1], np.full(7, -12.5))
    assert np.all(np.isinf(lc[:, 2])) and np.all(np.isinf(lc[1]))          # a filter not returned / a failed row: no flux
    # the conversion chain ran first: KNtheta from inclination_EM (conversion.py:119-126), the fixed parameter is there
    assert len(seen) == 3 and abs(seen[0]["KNtheta"] - 0.3 * 180 / np.pi) < 1e-12 and seen[2]["offset"] == -12.5
    # the vectorised form: one call per batch, arrays in, [B, NS] arrays (and an ok mask) out; conversions apply to the arrays
    def gen_batch(sample_times, p):
        assert p["grb_mag0"].shape == (3,) and np.allclose(p["KNtheta"], p["inclination_EM"] * 180 / np.pi) and np.all(p["offset"] == -12.5)
        g = p["grb_mag0"][:, None] + p["grb_slope"][:, None] * np.log10(sample_times)[None, :]
        return {"g": g, "r": np.full_like(g, -12.5)}, p["grb_mag0"] <= -14.5
    ext_b = ExternalLightCurveModel("PLGRB", ["g", "r", "i"], st, model_parameters=["grb_mag0", "grb_slope"], generate_lightcurve_batch=gen_batch)
    lc_b, ok_b = CombinedLightCurveModelContainer([other, ext_b]).host_operands(theta, names, {"offset": -12.5}, given)["PLGRB"]
    assert ok_b.tolist() == ok.tolist() and np.array_equal(lc_b, lc)
    # ... and whether the delivered rows are free of interior gaps is noted per batch (the failed row and the edge nodes do not count;
    # a filter the model does not return is all +inf -- interior nodes without a value -- so this model makes no such promise)
    assert not ext_b.batch_gap_free and not other.batch_gap_free
    ext_gr = ExternalLightCurveModel("PLGRB", ["g", "r"], st, model_parameters=["grb_mag0", "grb_slope"], generate_lightcurve_batch=gen_batch)
    CombinedLightCurveModelContainer([ExternalLightCurveModel("OTHER", ["g", "r"], st), ext_gr]).host_operands(theta, names, {"offset": -12.5}, {"OTHER": np.zeros((3, 2, 7))})
    assert ext_gr.batch_gap_free

    def gen_hole(sample_times, p):
        res, good = gen_batch(sample_times, p)
        res["g"][0, 3] = np.nan
        return res, good
    ext_h = ExternalLightCurveModel("PLGRB", ["g", "r"], st, model_parameters=["grb_mag0", "grb_slope"], generate_lightcurve_batch=gen_hole)
    CombinedLightCurveModelContainer([ExternalLightCurveModel("OTHER", ["g", "r"], st), ext_h]).host_operands(theta, names, {"offset": -12.5}, {"OTHER": np.zeros((3, 2, 7))})
    assert not ext_h.batch_gap_free
    # nothing to do when every external sub-model is supplied or has no callable
    assert set(comb.host_operands(theta, names, {}, {"OTHER": given["OTHER"], "PLGRB": (lc, ok)})) == {"OTHER", "PLGRB"}
    with pytest.raises(RuntimeError):
        other.generate_lightcurve(st, {})


def test_filters_and_detection_limits_from_driver_arguments():
    """``set_filters`` / ``create_detection_limit`` (em/utils.py:96-196) against the reference's own functions on the argument forms
    the drivers produce: explicit filter strings with blanks, survey names, Rubin ToO strategies, explicit and survey limits."""
    import types
    from nmma_amd.em import utils as amd
    forms = [dict(filters="ztfg, ztfr,,sdssu "), dict(filters=["g,r", " i "]), dict(filters=None, em_detectors="ztf,lsst"),
             dict(filters=None, em_detectors=["ZTF ", "rubin"], rubin_ToO_type=False), dict(filters=None, em_detectors="rubin", rubin_ToO_type="gold_z"),
             dict(filters=None, em_detectors=None, rubin_ToO_type="silver"), dict(filters=None, em_detectors="lsst,rubin", rubin_ToO_type="platinum"),
             dict(filters=None, em_detectors=None, rubin_ToO_type=False), dict(filters=None, em_detectors="ztf,hst", rubin_ToO_type=False),
             dict(filters=" , ", em_detectors=None)]
    limit_forms = [dict(detection_limit=22.5), dict(detection_limit=[21.0, 22.0, 23.0]), dict(detection_limit={"ztfg": 20.0}),
                   dict(detection_limit=None, em_detectors="ztf,lsst"), dict(detection_limit=None, em_detectors=["rubin"]),
                   dict(detection_limit=None, em_detectors=None, rubin_ToO_type="gold"), dict(detection_limit=None, em_detectors="hst"),
                   dict(detection_limit=None)]
    have_ref = os.path.isdir("/root/reference/nmma")
    if have_ref:
        from oracle import ref_harness
        ref = ref_harness.reference_modules().utils

    def outcome(fn, *a):
        try:
            return fn(*a)
        except (ValueError, NotImplementedError, AssertionError, AttributeError) as exc:
            return type(exc).__name__

    for f in forms:
        ns = types.SimpleNamespace(**{"em_detectors": None, "rubin_ToO_type": False, **f})
        got = outcome(amd.set_filters, ns)
        if have_ref:
            want = outcome(ref.set_filters, types.SimpleNamespace(**vars(ns)))
            # (a ToO strategy without --em-detectors: the reference calls None.copy(); here the strategy's filters come back)
            assert got == want or (want == "AttributeError" and ns.em_detectors is None and ns.rubin_ToO_type), f
    assert amd.set_filters(types.SimpleNamespace(filters="ztfg, ztfr,,sdssu ")) == ["ztfg", "ztfr", "sdssu"]
    assert amd.set_filters(types.SimpleNamespace(filters=None, em_detectors="ztf", rubin_ToO_type="silver_z")) == ["ztfg", "ztfr", "ztfi", "ps1::g", "ps1::z"]
    filt = ["ztfg", "ztfr", "ps1::g"]
    for f in limit_forms:
        ns = types.SimpleNamespace(**{"em_detectors": None, "rubin_ToO_type": None, "detection_limit_fits_file": None, **f})
        got = outcome(amd.create_detection_limit, ns, filt)
        if have_ref:
            assert got == outcome(ref.create_detection_limit, types.SimpleNamespace(**vars(ns)), filt), f
    lim = amd.create_detection_limit(types.SimpleNamespace(detection_limit=None, em_detectors="ztf", rubin_ToO_type=None), filt, 30.0)
    assert lim["ztfg"] == 21.7 and lim["ps1::g"] == 30.0 and lim["ztfi"] == 20.9


def test_single_model_that_lists_a_filter_without_a_network_floors_every_sample():
    """A surrogate model that LISTS a filter it has no network for answers +inf on every node (calc_svd_lc's null output,
    lightcurve_generation.py:168-169) and ``sanity_check`` (em_likelihood.py:305-311) then fails for every sample: the reference's
    likelihood is the floor whatever the parameters (re-checked against the reference itself in tests/test_oracle_vs_reference.py).
    The plugin answers the same without touching the device; inside a combination the same filters are legitimate (null filters of
    the one-launch engine, tests/test_gpu_stack2.py)."""
    from nmma_amd import synthetic as syn
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    from oracle import nmma_oracle as orc
    from tests.helpers import SimplePrior
    case = syn.config2_case()
    allf = list(case["model_filters"]) + ["X-ray-1keV"]
    okn = orc.OracleSVDModel(case["model_parameters"], case["svd"], filters=allf, sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    olik = orc.OracleLikelihood(okn, case["data"], dict(mode="budget", values={f: 1.0 for f in case["observed_filters"]}), case["observed_filters"],
                                detection_limit=np.inf, known_filters=allf, use_scipy=True)
    assert np.all(orc.log_likelihood_batch(olik, case["names"], case["theta"][:8]) == orc.LOGL_FLOOR)
    model = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=allf, model_parameters=case["model_parameters"],
                               sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    times, mags, sigmas = case["data"]
    handler = FilterSystematicsHandler(case["observed_filters"], error_budget=1.0, light_curve_times=times)
    lik = EMTransientLikelihood(model, (times, mags, sigmas, 0.0), handler, {n: SimplePrior(0.0, 1.0) for n in case["names"]},
                                filters=case["observed_filters"])
    got = lik.log_likelihood_batch(case["theta"][:8], case["names"])
    assert got.shape == (8,) and np.all(got == orc.LOGL_FLOOR)
    assert lik.sub_model.log_likelihood(dict(zip(case["names"], map(float, case["theta"][0])))) == -np.inf
