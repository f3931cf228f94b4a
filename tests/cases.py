"""Named, seeded parity cases shared by the golden-vector generator
(``tools/make_golden.py``), the oracle tests and the GPU parity tests.

A case is a plain dict of *inputs* (numpy arrays, lists, floats):

  model_parameters, svd (filter -> tensors), model_filters, sample_times (or None),
  cosmo_grid (dist_grid, z_grid) or None, data (times, mags, sigmas),
  observed_filters, detection_limit, systematics (oracle-style spec, see
  oracle/nmma_oracle.py:OracleSystematics), systematics_ref (kwargs that make the
  *reference's* FilterSystematicsHandler behave the same), names, theta[B, D]
"""
from __future__ import annotations

import os

import numpy as np

from nmma_amd import synthetic as syn

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")


#: the seeded case builder lives with the other synthetic inputs (bench.py and smoke() use it without importing tests/)
_base = syn.make_case


def case_c2_default():
    """BASELINE config 2 shape: Bu2019lm, 6 AT2017gfo filters, sample_times = training grid."""
    return _base()


def case_c2_dt05_limit():
    """Canonical CLI grid ``--em-tmin .1 --em-tmax 20 --em-tstep .5`` (two-stage lerp)
    plus a finite detection limit (truncated Gaussian, -inf when m_obs > lim)."""
    c = _base(seed=2234)
    c["sample_times"] = np.arange(0.1, 20.5, 0.5)
    # a limit just above the faintest datum of most filters; below one datum of 2massj
    lim = {f: float(np.max(c["data"][1][f]) + 0.3) for f in c["observed_filters"]}
    c["detection_limit"] = lim
    return c


def case_c2_dt05():
    """The documented CLI call ``--em-tmin .1 --em-tmax 20 --em-tstep .5`` with the default error budget and no
    detection limit: two-stage interpolation (sample nodes between SVD nodes) on the lean task."""
    c = _base(seed=2236)
    c["sample_times"] = np.arange(0.1, 20.5, 0.5)
    return c


def case_grid_subset():
    """sample_times = every second node of the SVD training grid: every sample node coincides with an SVD node
    (stage-1 offsets are all zero) but the grids differ."""
    c = _base(seed=2237, batch=48)
    tt = next(iter(c["svd"].values()))["tt"]
    c["sample_times"] = np.asarray(tt)[::2].copy()
    return c


def case_limit_violated():
    """One datum fainter than the detection limit -> -inf -> floor for every sample."""
    c = _base(seed=2235, batch=8)
    f = c["observed_filters"][1]
    c["detection_limit"] = {g: np.inf for g in c["observed_filters"]}
    c["detection_limit"][f] = float(np.sort(c["data"][1][f])[-2])
    return c


def case_syserr_param():
    """Sampled global ``em_syserr`` (FilterSystematicsHandler.from_param)."""
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift",
             "log10_mej_dyn", "log10_mej_wind", "em_syserr"]
    c = _base(seed=3234, names=names)
    c["systematics"] = dict(mode="param", name="em_syserr")
    c["systematics_ref"] = dict(error_budget=None, systematics_file=None)
    return c


def case_syserr_time_nodes():
    """Time-dependent systematics: 4 linear time nodes shared by two filter groups and
    a single parameter for the rest (from_parameters = single + interpolated)."""
    filters = syn.AT2017GFO_FILTERS
    g1 = ["ps1::g", "ps1::r"]
    nodes = np.linspace(0.0, 21.0, 4)
    n_a = [f"em_syserr_blue_{i}" for i in range(4)]
    n_b = [f"em_syserr_2massj_{i}" for i in range(4)]
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift",
             "log10_mej_dyn", "log10_mej_wind", "em_syserr_rest"] + n_a + n_b
    c = _base(seed=4234, names=names, batch=48)
    spec_names = {f: "em_syserr_rest" for f in filters if f not in g1 + ["2massj"]}
    spec_nodes = {f: (n_a, nodes) for f in g1}
    spec_nodes["2massj"] = (n_b, nodes)
    c["systematics"] = dict(mode="mixed", names=spec_names, nodes=spec_nodes)
    c["systematics_ref"] = dict(
        error_budget=None,
        systematics_file={
            "blue": {"filters": g1, "time_nodes": 4, "time_range": "lin 0.0 21.0"},
            "2massj": {"time_nodes": 4, "time_range": "lin 0.0 21.0"},
            "rest": {"prior": "unused"},
        })
    return c


def case_syserr_nodes_masked():
    """Non-finite SAMPLED node values of the time-dependent systematics: ``autocomplete_data`` drops them (em/utils.py:634-645 via
    systematics.py:288-291) -- a datum interpolates over the remaining nodes, the end values stay constant, and with fewer than
    two finite nodes the group's sigma_sys is +inf: every one of its data becomes an upper limit (ln 1/2 each)."""
    c = case_syserr_time_nodes()
    names, th = c["names"], c["theta"]
    a = [names.index(f"em_syserr_blue_{i}") for i in range(4)]
    b = [names.index(f"em_syserr_2massj_{i}") for i in range(4)]
    th[1, a[1]] = np.nan                       # a middle node
    th[2, a[0]] = np.inf                       # the first node: constant from the second on
    th[3, b[3]] = np.nan                       # the last node of the other group
    th[4, a[1]] = th[4, a[2]] = np.nan         # two neighbouring middle nodes
    th[5, a[0]] = th[5, a[1]] = th[5, a[2]] = np.nan       # one finite node left: sigma_sys = +inf for the blue bands
    th[6, b[0]] = th[6, b[1]] = th[6, b[2]] = th[6, b[3]] = -np.inf      # none left
    th[7, a[3]] = np.nan; th[7, b[0]] = np.nan
    th[8, names.index("em_syserr_rest")] = np.nan          # a single sampled parameter is NOT masked: NaN sigma -> floor
    return c


def case_averaging(names=None, counts=12, n_new=9, tt=None):
    """Observed filters the model does not provide (``w``, ``o``, ``I``): arithmetic mean
    of mapped model bands (em_likelihood.py:326-333), plus renamed ``B -> g``.
    (The keyword arguments make variants for the lean-task tests; the defaults are the golden case.)"""
    model_filters = ["g", "r", "i", "z", "y"]
    c = _base(seed=5234, filters=model_filters, counts=counts, batch=32, upper_limit_filter="i", names=names, tt=tt)
    times, mags, sigmas = c["data"]
    rng = np.random.default_rng(99)
    for new, src in (("w", ["g", "r", "i"]), ("o", ["r", "i"]), ("I", ["z", "y"]), ("B", ["g"])):
        t = np.sort(rng.uniform(0.6, 13.0, n_new))
        m = np.mean([np.interp(t, times[s], mags[s]) for s in src], axis=0)
        times[new], mags[new], sigmas[new] = t, m + 0.05 * rng.standard_normal(n_new), rng.uniform(0.02, 0.1, n_new)
    observed = ["g", "r", "i", "z", "y", "w", "o", "I", "B"]
    for k in list(times):
        if k not in observed:
            del times[k], mags[k], sigmas[k]
    c["observed_filters"] = observed
    c["systematics"] = dict(mode="budget", values={f: 0.5 for f in observed})
    c["systematics_ref"] = dict(error_budget=0.5, systematics_file=None)
    return c


def case_averaging_nodes_grid():
    """Averaged bands + time-node systematics (shared by the averaged ATLAS / PS1 bands, a group of its own for ``I``, one
    parameter for the rest) + the documented CLI grid: everything the general lean task adds, through the reference's own
    FilterSystematicsHandler and averaging code."""
    nodes = np.linspace(0.0, 21.0, 4)
    n_a = [f"em_syserr_wide_{i}" for i in range(4)]
    n_b = [f"em_syserr_I_{i}" for i in range(4)]
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift",
             "log10_mej_dyn", "log10_mej_wind", "em_syserr_rest"] + n_a + n_b
    c = case_averaging(names=names)
    c["sample_times"] = np.arange(0.1, 20.5, 0.5)
    wide = ["w", "o", "r"]
    spec_nodes = {f: (n_a, nodes) for f in wide}
    spec_nodes["I"] = (n_b, nodes)
    spec_names = {f: "em_syserr_rest" for f in c["observed_filters"] if f not in spec_nodes}
    c["systematics"] = dict(mode="mixed", names=spec_names, nodes=spec_nodes)
    c["systematics_ref"] = dict(
        error_budget=None,
        systematics_file={
            "wide": {"filters": wide, "time_nodes": 4, "time_range": "lin 0.0 21.0"},
            "I": {"time_nodes": 4, "time_range": "lin 0.0 21.0"},
            "rest": {"prior": "unused"},
        })
    return c


def case_edges():
    """timeshift / redshift pushing data out of the model window (m_est = +inf ->
    NaN -> floor), an upper-limit-only filter and a one-point filter."""
    c = _base(seed=6234, batch=40, t_range=(0.3, 20.5))
    times, mags, sigmas = c["data"]
    f_ul = c["observed_filters"][4]
    sigmas[f_ul][:] = np.inf                                  # only upper limits
    f_one = c["observed_filters"][5]
    for d in (times, mags, sigmas):
        d[f_one] = d[f_one][:1].copy()                        # single datum
    th = c["theta"]
    i_ts = c["names"].index("timeshift")
    th[:10, i_ts] = np.linspace(-2.0, 0.9, 10)                # positive shift: early data lost
    th[10:14, i_ts] = 0.3
    return c


def case_c4_shape():
    """BASELINE config 4 shape: Bu2022Ye (NP=6), 12 filters x 200 epochs."""
    filters = [f"band{i:02d}" for i in range(12)]
    names = ["luminosity_distance", "inclination_EM", "timeshift", "log10_mej_dyn", "vej_dyn",
             "Yedyn", "log10_mej_wind", "vej_wind"]
    c = _base(seed=7234, model="Bu2022Ye", filters=filters, counts=200, batch=16, names=names,
              upper_limit_filter="band03")
    return c


def case_c4_syserr():
    """BASELINE config 4's shape with ONE sampled systematic shared by all filters (em_syserr, the default of current NMMA
    priors) and 130 epochs per filter: the dense lean task's sampled-sigma variant on item-staged photometry."""
    filters = [f"band{i:02d}" for i in range(12)]
    names = ["luminosity_distance", "inclination_EM", "timeshift", "log10_mej_dyn", "vej_dyn",
             "Yedyn", "log10_mej_wind", "vej_wind", "em_syserr"]
    c = _base(seed=7334, model="Bu2022Ye", filters=filters, counts=130, batch=16, names=names,
              upper_limit_filter="band07")
    c["systematics"] = dict(mode="param", name="em_syserr")
    c["systematics_ref"] = dict(error_budget=None, systematics_file=None)
    return c


def case_c4_dt05():
    """12 filters x 120 epochs on the documented CLI grid (--tmin .1 --tmax 20 --dt .5: 41 sample nodes, each a stage-1 lerp
    between two SVD nodes): the dense lean task with the lerp folded into its matrix-core operands."""
    filters = [f"band{i:02d}" for i in range(12)]
    names = ["luminosity_distance", "inclination_EM", "timeshift", "log10_mej_dyn", "vej_dyn",
             "Yedyn", "log10_mej_wind", "vej_wind"]
    c = _base(seed=7434, model="Bu2022Ye", filters=filters, counts=120, batch=16, names=names,
              upper_limit_filter="band05", sample_times=np.arange(0.1, 20.5, 0.5))
    return c


def case_fixed_distance():
    """luminosity_distance FIXED by its prior (DeltaFunction): the reference's constant z(d_L) grid still applies
    the redshift of that distance (model.py:255-267 with get_cosmo_grids(d, d)) -- time stretch and K-correction."""
    names = ["KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind"]
    c = _base(seed=8334, batch=32, names=names)
    d = 40.0
    z = float(np.interp(d, *c["cosmo_grid"]))
    c["fixed"] = {"luminosity_distance": d}
    c["cosmo_grid"] = (np.full(50, d), np.full(50, z))          # what get_cosmo_grids(d, d) returns
    return c


def case_conversions():
    """Every device conversion slot: KNtheta from theta_jn (folded to [0, pi/2], conversion.py:119-126) and the
    log10_ alias of a linear mass (model.py:276-281)."""
    names = ["luminosity_distance", "KNphi", "theta_jn", "timeshift", "mej_dyn", "log10_mej_wind"]
    return _base(seed=8434, batch=32, names=names)


def case_conversions_cos():
    """KNtheta from cos_theta_jn; both masses linear."""
    names = ["luminosity_distance", "KNphi", "cos_theta_jn", "timeshift", "mej_dyn", "mej_wind"]
    return _base(seed=8534, batch=32, names=names)


def case_real_nets():
    """The three TRAINED networks the reference ships for its tests (Bu2019nsbh_tf: Dense(3 -> 2048, relu) -> Dense(2048 -> 10),
    fp32; fixture tests/golden/bu2019nsbh_tf_weights.npz written by tools/convert_h5_weights.py) on a synthetic SVD basis
    (the reference tree holds no .joblib metadata for this model).  First case with non-Gaussian, trained weights."""
    filters = ["ztfr", "sdssu", "2massks"]
    names = ["luminosity_distance", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind"]
    mp, svd = syn.make_svd_model(8634, filters, model="Bu2019nsbh")
    with np.load(os.path.join(GOLDEN_DIR, "bu2019nsbh_tf_weights.npz")) as z:
        for f in filters:
            for k in ("W1", "b1", "W2", "b2"):
                assert svd[f][k].shape == z[f"{f}/{k}"].shape
                svd[f][k] = np.ascontiguousarray(z[f"{f}/{k}"], dtype=np.float32)
    grid = syn.flat_lcdm_grid(1.0, 200.0)
    data = syn.make_photometry(8635, svd, mp, filters=filters, counts=dict(ztfr=21, sdssu=9, **{"2massks": 14}),
                               cosmo_grid=grid, upper_limit_filter="ztfr")
    names, theta = syn.draw_theta(8636, 48, names)
    return dict(model="Bu2019nsbh", model_parameters=mp, svd=svd, model_filters=filters, sample_times=None,
                cosmo_grid=grid, data=data, observed_filters=filters, detection_limit=np.inf,
                systematics=dict(mode="budget", values={f: 1.0 for f in filters}),
                systematics_ref=dict(error_budget=1.0, systematics_file=None), names=names, theta=theta)


AT2017GFO_TRIGGER_MJD = 57982.5285236896      # GW170817: 2017-08-17 12:41:04 UTC


def at2017gfo_raw_photometry():
    """The reference's example data set ``example_files/lightcurves/AT2017gfo.dat`` (141 rows, 9 filters, 3 upper limits) as the
    arrays its reader produces -- committed as the fixture tests/golden/at2017gfo_photometry.npz by tools/make_golden_at2017gfo.py
    (data, not source)."""
    with np.load(os.path.join(GOLDEN_DIR, "at2017gfo_photometry.npz")) as z:
        filters = [str(f) for f in z["filters"]]
        return {f: {"time": z[f"{f}/time"].copy(), "mag": z[f"{f}/mag"].copy(), "mag_error": z[f"{f}/mag_error"].copy()}
                for f in filters}


def case_at2017gfo():
    """The REAL AT2017gfo photometry (all 9 filters, 3 upper limits) through the reference's own preparation chain -- time cut
    (utils.py:233-253), times relative to the trigger (:255-286), model-window check (:289-353) -- on the documented CLI grid
    ``--tmin .1 --tmax 20 --dt .5`` with the sampled ``em_syserr`` of current NMMA priors.  The Bu2019lm surrogate itself is
    synthetic (the trained networks are not in the reference tree); rows later than 14.5 d are cut because the model window
    ends at 18.1 d for the earliest allowed trigger time (the reference raises otherwise)."""
    from nmma_amd.em import utils as em_utils
    raw = em_utils.cut_data_to_time_range(at2017gfo_raw_photometry(), None, AT2017GFO_TRIGGER_MJD, tmin=0.0, tmax=14.5)
    filters = list(raw)
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind", "em_syserr"]
    mp, svd = syn.make_svd_model(8734, filters, model="Bu2019lm")
    grid = syn.flat_lcdm_grid(1.0, 200.0)
    times, mags, sigmas, _ = em_utils.setup_filtered_lc_data(raw, AT2017GFO_TRIGGER_MJD)
    names, theta = syn.draw_theta(8736, 64, names)
    # a synthetic surrogate does not know the real magnitudes: shift every filter's span so that the model passes through the data
    # (keeps log L in a numerically meaningful range; the time stamps, error bars and limits stay the real ones)
    for f in filters:
        off = float(np.median(mags[f][np.isfinite(sigmas[f])])) - (5.0 * (5 + np.log10(40.0)) - 13.0)
        svd[f]["mins"] = svd[f]["mins"] + off
        svd[f]["maxs"] = svd[f]["maxs"] + off
    return dict(model="Bu2019lm", model_parameters=mp, svd=svd, model_filters=filters,
                sample_times=np.arange(0.1, 20.5, 0.5), cosmo_grid=grid, data=(times, mags, sigmas), raw_data=raw,
                trigger_time=AT2017GFO_TRIGGER_MJD, data_window=(0.0, 14.5),
                prior_bounds=dict(luminosity_distance=(1.0, 200.0), timeshift=(-2.0, 0.1)),
                observed_filters=filters, detection_limit=np.inf,
                systematics=dict(mode="param", name="em_syserr"),
                systematics_ref=dict(error_budget=None, systematics_file=None), names=names, theta=theta)


def case_unobserved_filter_overflow():
    """A model filter NOBODY observed whose surrogate overflows for part of the prior: the reference evaluates every model
    filter and its sanity_check floors a sample as soon as one of the curves is unusable (em_likelihood.py:305-311) -- here the
    7th filter's leading coefficient is +inf exactly for the samples with a large first surrogate input (one hidden unit
    relu(10 x_0 - 5) times a weight of 3e38 overflows fp32 above x_0 = 0.61), everything observed stays finite."""
    filters = list(syn.AT2017GFO_FILTERS) + ["2massh"]
    c = _base(seed=8934, filters=filters, batch=48)
    obs = filters[:-1]
    c["observed_filters"] = obs
    c["data"] = tuple({f: d[f] for f in obs} for d in c["data"])
    c["systematics"] = dict(mode="budget", values={f: 1.0 for f in obs})
    t = c["svd"]["2massh"]
    t["W1"][:, 0] = 0.0; t["W1"][0, 0] = 10.0
    t["b1"][0] = -5.0
    t["W2"][0, :] = 0.0; t["W2"][0, 0] = 3.0e38
    return c


def case_unobserved_filter_dead():
    """A model filter nobody observed whose basis has a non-finite column: fewer than two finite nodes, the curve is all-inf
    (model.py:381-404) and sanity_check floors EVERY sample."""
    c = case_unobserved_filter_overflow()
    c["svd"] = {f: dict(t) for f, t in c["svd"].items()}
    mp, fresh = syn.make_svd_model(8934, ["2massh"])
    va = fresh["2massh"]["VA"].copy()
    va[:, 3] = np.inf
    c["svd"]["2massh"] = dict(fresh["2massh"], VA=va)
    c["theta"] = c["theta"][:16]
    return c


def case_bulla_svd():
    """A surrogate with REAL conditioning: the SVD basis, ``mins`` / ``maxs`` spans (0.7 ... 23.6 mag across the grid) and
    coefficient ranges the reference's own ``generate_svd_model`` (em/training.py:198-265) builds from the 28 POSSIS curves of
    ``nmma/tests/data/bulla`` (model ``Bu2019lm_sparse``: two ejecta masses, 9 filters, documented training grid), with an fp32
    network fitted to its ``cAmat`` (fixture tests/golden/bulla_svd_model.npz, arrays only, written by
    tools/make_golden_bulla.py) -- every other golden case has a QR-orthonormalised Gaussian basis.  9 filters, the documented
    CLI grid, the sampled ``em_syserr`` of current NMMA priors, one upper limit and one filter with a finite detection limit.
    Rows 0-47 lie inside the training box, the rest extrapolate."""
    with np.load(os.path.join(GOLDEN_DIR, "bulla_svd_model.npz")) as z:
        filters = [str(f) for f in z["filters"]]
        mp = [str(n) for n in z["model_parameters"]]
        svd = {}
        for f in filters:
            t = {k: np.ascontiguousarray(z[f"{f}/{k}"]) for k in ("W1", "b1", "W2", "b2", "VA", "mins", "maxs", "tt",
                                                                   "param_mins", "param_maxs")}
            t["n_coeff"] = int(t["W2"].shape[1])
            svd[f] = t
    grid = syn.flat_lcdm_grid(1.0, 200.0)
    counts = dict(zip(filters, (6, 14, 22, 17, 18, 15, 14, 17, 23)))
    data = syn.make_photometry(8835, svd, mp, filters=filters, counts=counts, cosmo_grid=grid, upper_limit_filter="ztfi")
    names, theta = syn.draw_theta(8836, 64, ["luminosity_distance", "timeshift", "log10_mej_dyn", "log10_mej_wind", "em_syserr"])
    rng = np.random.default_rng(8837)
    pmin, pmax = svd[filters[0]]["param_mins"], svd[filters[0]]["param_maxs"]
    for k, n in enumerate(mp):
        theta[:48, names.index(n)] = rng.uniform(pmin[k], pmax[k], 48)
    lim = {f: np.inf for f in filters}
    finite = np.isfinite(data[2]["ztfr"])
    lim["ztfr"] = float(np.max(data[1]["ztfr"][finite]) + 0.4)
    # Row 34 is the best fit: log L = -7.17 is what is left of ~150 terms of order one.  One ulp(fp32) of the leading SVD
    # coefficient (|c_0| ~ 13: 9.5e-7) times a span of ~20 mag moves a node by ~1e-5 mag and log L by a few 1e-5 -- any two fp32
    # forwards of the network (numpy's BLAS, Keras' Eigen, the MFMA chain) differ by that much; relative to 7.17 it reads 5e-6.
    # The GPU test therefore allows ONE row an absolute 1e-4 instead of the relative 1e-6 (and checks the kernel against the
    # order-independent fp64-accumulated value: test_realistic_basis_hip_is_as_close_to_exact_arithmetic_as_the_reference_stand_in).
    return dict(model="Bu2019lm_sparse", model_parameters=mp, svd=svd, model_filters=filters,
                sample_times=np.arange(0.1, 20.5, 0.5), cosmo_grid=grid, data=data, observed_filters=filters,
                detection_limit=lim, systematics=dict(mode="param", name="em_syserr"),
                systematics_ref=dict(error_budget=None, systematics_file=None), names=names, theta=theta,
                logl_atol=1e-4, logl_atol_rows=1)


def case_small_hidden():
    """Tiny surrogate (NH=64) for fast pure-Python loops."""
    return _base(seed=8234, n_hidden=64, batch=16)


def case_dense_edges(sampled_sys=False):
    """The dense task's bookkeeping (round 6: padded detection records, upper limits behind them, the window test as a mask) pinned by
    the reference itself: five bands of 64 / 47 / 85 / 40 / 33 points on the documented CLI grid, 22 upper limits in one band, a band
    that is all upper limits but two detections, rows whose time shift pushes every epoch out of the model window (detections ->
    floor; upper limits alone would contribute log sf(-inf) = 0), a NaN model parameter."""
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind"] + (["em_syserr"] if sampled_sys else [])
    c = _base(seed=9771, filters=["g", "r", "i", "z", "y"], counts=dict(g=64, r=47, i=85, z=40, y=33), batch=64, names=names, upper_limit_filter="i")
    if sampled_sys:
        c["systematics"] = dict(mode="param", name="em_syserr")
        c["systematics_ref"] = dict(error_budget=None, systematics_file=None)
    times, mags, sig = (dict(d) for d in c["data"])
    sig = {f: np.array(v, dtype=float) for f, v in sig.items()}
    sig["i"][::4] = np.inf
    sig["y"][2:] = np.inf
    c["data"] = (times, mags, sig)
    c["sample_times"] = np.arange(0.1, 20.5, 0.5)
    th = c["theta"]
    ts = names.index("timeshift")
    th[3, ts] = 30.0
    th[4, ts] = -25.0
    th[5, names.index("log10_mej_dyn")] = np.nan
    if sampled_sys:
        th[6, names.index("em_syserr")] = np.nan
    return c


def case_dense_edges_syserr():
    return case_dense_edges(sampled_sys=True)


CASES = {
    "c2_default": case_c2_default,
    "dense_edges": case_dense_edges,
    "dense_edges_syserr": case_dense_edges_syserr,
    "c2_dt05_limit": case_c2_dt05_limit,
    "c2_dt05": case_c2_dt05,
    "grid_subset": case_grid_subset,
    "limit_violated": case_limit_violated,
    "syserr_param": case_syserr_param,
    "syserr_time_nodes": case_syserr_time_nodes,
    "syserr_nodes_masked": case_syserr_nodes_masked,
    "averaging": case_averaging,
    "averaging_nodes_grid": case_averaging_nodes_grid,
    "edges": case_edges,
    "c4_shape": case_c4_shape,
    "c4_syserr": case_c4_syserr,
    "c4_dt05": case_c4_dt05,
    "small_hidden": case_small_hidden,
    "fixed_distance": case_fixed_distance,
    "real_nets": case_real_nets,
    "conversions": case_conversions,
    "conversions_cos": case_conversions_cos,
    "at2017gfo": case_at2017gfo,
    "unobserved_filter_overflow": case_unobserved_filter_overflow,
    "unobserved_filter_dead": case_unobserved_filter_dead,
    "bulla_svd": case_bulla_svd,
}


def load_golden(name):
    """Expected outputs produced by the reference's own source (tools/make_golden.py)."""
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def weights_digest(svd):
    """Order-stable checksum of a synthetic model (guards the seeded generator)."""
    acc = 0.0
    for f in sorted(svd):
        for k in ("W1", "b1", "W2", "b2", "VA", "mins", "maxs"):
            a = np.asarray(svd[f][k], dtype=np.float64)
            a = np.where(np.isfinite(a), a, 0.0)          # (cases with deliberately broken tensors)
            acc += float(np.sum(a * np.cos(np.arange(a.size).reshape(a.shape) % 97)))
    return acc


# ---------------------------------------------------------------------------------------
# Shape cases: they exercise geometry branches of the HIP path (lanes per sample, ring wrap-around, KP = 2).
# ---------------------------------------------------------------------------------------
def case_fast_many_filters():
    """10 filters, 1 ... 128 epochs each: every lanes-per-sample variant of the fast path
    (<= 32, <= 64, <= 128 points) and more work items than ring slots."""
    filters = [f"f{i}" for i in range(10)]
    counts = dict(zip(filters, [13, 40, 70, 100, 16, 33, 64, 128, 1, 20]))
    return _base(seed=9234, filters=filters, counts=counts, batch=48, upper_limit_filter="f3")


def case_fast_np6():
    """Six model parameters (two layer-1 MFMA k-steps) on the fast path."""
    filters = ["u", "g", "r"]
    names = ["luminosity_distance", "inclination_EM", "timeshift", "log10_mej_dyn", "vej_dyn",
             "Yedyn", "log10_mej_wind", "vej_wind"]
    return _base(seed=9334, model="Bu2022Ye", filters=filters, counts=30, batch=40, names=names,
                 upper_limit_filter="g")


def case_many_points():
    """One filter with 150 epochs next to small ones: three data per lane (a second pass over slot pairs)."""
    filters = ["a", "b", "d"]          # (not "c": the reference reads that name as the ATLAS cyan average of g and r)
    counts = dict(a=12, b=150, d=20)
    return _base(seed=9434, filters=filters, counts=counts, batch=24, upper_limit_filter="b")


def case_fast_single_filter():
    """One observed filter: a ring of one slot, a single work item."""
    return _base(seed=9534, filters=["r"], counts=dict(r=17), batch=20, upper_limit_filter="r")


def case_ncoeff7():
    """svd_mag_ncoeff = 7 (the reference's --svd-mag-ncoeff): fewer SVD coefficients than the default 10 -- the generic item
    phase, and a K that is not a multiple of 4 for the fp64-MFMA reconstruction of the light-curve kernels."""
    return _base(seed=9644, batch=40, n_coeff=7, sample_times=np.arange(0.1, 20.5, 0.5))


def case_fast_wide():
    """20 filters x 70 epochs (64 lanes per sample): with 32-sample tiles the task list (640 entries)
    exceeds the LDS task map and takes the scan fallback."""
    filters = [f"w{i:02d}" for i in range(20)]
    return _base(seed=9634, filters=filters, counts=70, batch=40, upper_limit_filter="w07")


def case_extinction_limit():
    """Sampled E(B-V) with per-filter extinction coefficients (A_f = k_f * Ebv, the role of
    get_extinction_mags, em/model.py:323-342) and a finite detection limit on the default grid:
    both handled by the extended fast task."""
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind", "Ebv"]
    c = _base(seed=9734, batch=48, names=names)
    c["ebv_coeff"] = {f: 3.1 - 0.45 * i for i, f in enumerate(c["model_filters"])}
    c["detection_limit"] = 24.5
    return c


def case_extinction_linear():
    """Sampled E(B-V) with per-filter linear coefficients and no detection limit: handled by the lean task."""
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind", "Ebv"]
    c = _base(seed=9735, batch=48, names=names)
    c["ebv_coeff"] = {f: 2.9 - 0.4 * i for i, f in enumerate(c["model_filters"])}
    c["theta"][:4, names.index("Ebv")] = 0.0
    return c


def case_extinction_p92():
    """Sampled E(B-V) under the reference's DEFAULT law: the Pei (1992) SMC curve read at every sample's
    host-frame wavelength (utils.py:373-428, model.py:323-342) -- the extinction magnitude of a filter
    depends on the sample's redshift.  Filter frequencies from UV to the K band; two far outside the curve's
    applicability (> 2e16 Hz cut-off, and a radio band below 3e11 Hz) where the factor is 1."""
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind", "Ebv"]
    c = _base(seed=9744, batch=48, names=names)
    lam_um = [0.155, 0.36, 0.62, 1.25, 2.2]
    nu = [2.99792458e14 / x for x in lam_um] + [3.0e16]
    assert len(nu) >= len(c["model_filters"]) - 1
    c["filter_nu0"] = dict(zip(c["model_filters"], nu))
    c["theta"][:5, names.index("Ebv")] = 0.0             # model.py:328-330: nothing applied at Ebv == 0
    return c


def case_hubble_sampled():
    """A sampled Hubble constant (priors/Bu2019lm_Hubble.prior: H0 in [50, 90], d_L in [1, 200] Mpc): every sample's redshift
    comes from ITS cosmology (core/base.py:161-164 -> conversion.py:57-101, a root-find per sample in the reference).  The
    device reads ONE 256-node grid, tabulated for H0 = 67.66, at d_L * H0 / 67.66; the oracle root-finds per sample in the
    same matter + Lambda cosmology the grid was made with (astropy is absent: the cosmology itself is an input)."""
    names = ["luminosity_distance", "Hubble_constant", "KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind"]
    c = _base(seed=9954, batch=48, names=[n for n in names if n != "Hubble_constant"])
    rng = np.random.default_rng(9955)
    h0 = rng.uniform(50.0, 90.0, 48)
    c["theta"] = np.insert(c["theta"], 1, h0, axis=1)
    c["names"] = names
    h_ref, om0 = 67.66, 0.30966
    c["hubble_reference"] = h_ref
    c["cosmo_grid"] = syn.flat_lcdm_grid(1.0 * 50.0 / h_ref, 200.0 * 90.0 / h_ref, n=256, H0=h_ref, Om0=om0)

    def z_of_dl(d, h):
        lo, hi = 0.0, 1.0
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            zz = np.linspace(0.0, mid, 2049)
            dl = (1 + mid) * 299792.458 / h * np.trapezoid(1.0 / np.sqrt(om0 * (1 + zz) ** 3 + (1 - om0)), zz)
            lo, hi = (mid, hi) if dl < d else (lo, mid)
        return 0.5 * (lo + hi)
    c["z_of_dl"] = z_of_dl
    return c


def case_log_grid():
    """The CLI's default grid when --em-tmin/--em-tmax are given (150 log-spaced sample times,
    em/utils.py:87-88): two-stage interpolation on a non-uniform grid (bisection instead of an index guess)."""
    c = _base(seed=9834, batch=40, sample_times=np.geomspace(0.2, 20.0, 150))
    return c


def case_nonuniform_tt():
    """An SVD model trained on an UNEQUALLY spaced time grid (geometric, as ``--tmin/--tmax`` training without ``--dt`` gives,
    em/utils.py:87-88) evaluated on its own grid (``sample_times=None``): identity stage 1 on a non-uniform grid -- bisection
    for the bracket, tabulated node spacings, and no stage-1 lerp (round-2 advisor finding: the non-uniform lean task read
    stage-1 tables that were not staged for this combination)."""
    return _base(seed=9934, batch=40, tt=np.geomspace(0.1, 21.0, 160))


#: geometry cases: the reference runs all of them unchanged except the two extinction cases (its dust law is third-party),
#: so they are golden cases too (tools/make_golden.py writes tests/golden/<name>.npz for every entry of CASES)
SHAPE_CASES = {
    "fast_many_filters": case_fast_many_filters,
    "fast_np6": case_fast_np6,
    "ncoeff7": case_ncoeff7,
    "many_points": case_many_points,
    "fast_single_filter": case_fast_single_filter,
    "fast_wide": case_fast_wide,
    "extinction_limit": case_extinction_limit,
    "extinction_linear": case_extinction_linear,
    "extinction_p92": case_extinction_p92,
    "log_grid": case_log_grid,
    "nonuniform_tt": case_nonuniform_tt,
    "hubble_sampled": case_hubble_sampled,
}
ORACLE_ONLY_CASES = ("extinction_limit", "extinction_linear", "extinction_p92", "hubble_sampled")
CASES.update({k: v for k, v in SHAPE_CASES.items() if k not in ORACLE_ONLY_CASES})
