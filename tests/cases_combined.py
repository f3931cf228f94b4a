"""BASELINE config 3 shape: SVD kilonova (Bu2019lm-like) + a second transient (GRB afterglow
stand-in: a power-law decay, because the real afterglow model is third-party afterglowpy),
flux-summed by CombinedLightCurveModelContainer on a common sample_times grid; 9 filters."""
import numpy as np

from nmma_amd import synthetic as syn

FILTERS = ["ps1::g", "ps1::r", "ps1::i", "ps1::z", "ps1::y", "2massj", "2massh", "2massks", "sdssu"]
NAMES = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift", "log10_mej_dyn", "log10_mej_wind",
         "grb_mag0", "grb_slope"]


def case_combined(seed=9123, batch=48):
    mp, svd = syn.make_svd_model(seed, FILTERS, model="Bu2019lm")
    grid = syn.flat_lcdm_grid(1.0, 200.0)
    counts = dict(syn.AT2017GFO_COUNTS)
    counts["sdssu"] = 6
    data = syn.make_photometry(seed + 1, svd, mp, filters=FILTERS, counts=counts, cosmo_grid=grid)
    names, theta = syn.draw_theta(seed + 2, batch, NAMES[:6])
    rng = np.random.default_rng(seed + 3)
    theta = np.concatenate([theta, rng.uniform(-17.5, -14.0, (batch, 1)), rng.uniform(0.8, 1.6, (batch, 1))], axis=1)
    sample_times = np.arange(0.1, 20.5, 0.5)
    return dict(model="Bu2019lm", model_parameters=mp, svd=svd, filters=FILTERS, sample_times=sample_times,
                cosmo_grid=grid, data=data, names=NAMES, theta=theta)


def case_combined_syserr(seed=9323, batch=40):
    """The shared-grid combination with a SAMPLED systematic (``em_syserr``, FilterSystematicsHandler.from_param) and a second
    transient whose curves have an interior hole for the steeper half of its slope prior -- the reference fills it from the
    finite neighbours (autocomplete_data, model.py:1440-1448); in the one-launch form those rows take the re-evaluation launch."""
    c = case_combined(seed=seed, batch=batch)
    rng = np.random.default_rng(seed + 5)
    c["names"] = NAMES + ["em_syserr"]
    c["theta"] = np.concatenate([c["theta"], rng.uniform(0.1, 1.5, (batch, 1))], axis=1)
    c["systematics"] = dict(mode="param", name="em_syserr")
    c["systematics_ref"] = dict(error_budget=None, systematics_file=None)
    c["grb_hole"] = (9, 12, 1.2)
    return c


def case_combined_loggrid(seed=9423, batch=40):
    """The shared-grid combination on the CLI's log-spaced grid (unequally spaced sample_times: bracket search instead of the index
    guess), with the same interior holes."""
    c = case_combined(seed=seed, batch=batch)
    c["sample_times"] = np.geomspace(0.1, 20.0, 36)
    c["grb_hole"] = (14, 16, 1.2)
    return c


def oracle_likelihood(case, use_scipy=True):
    from oracle import nmma_oracle as orc
    kn = orc.OracleSVDModel(case["model_parameters"], case["svd"], filters=case["filters"],
                            sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    grb = orc.OraclePowerLawModel(case["filters"], case["sample_times"], cosmo_grid=case["cosmo_grid"], hole=case.get("grb_hole"))
    comb = orc.OracleCombinedModel([kn, grb])
    systematics = case.get("systematics") or dict(mode="budget", values={f: 1.0 for f in case["filters"]})
    return orc.OracleLikelihood(comb, case["data"], systematics,
                                case["filters"], detection_limit=case.get("detection_limit", np.inf), known_filters=case["filters"],
                                use_scipy=use_scipy), grb


# ---------------------------------------------------------------------------------------------------------------
# General combination (model.py:1362-1374, :1434-1448, :1490-1503): the two sub-models bring DIFFERENT time grids and
# filter lists -- union grid, per-model re-interpolation with +inf outside, per-filter lookup with the averaged-band
# fallback ("w" is listed by the second model only: both then contribute the mean of their g, r, i).
# ---------------------------------------------------------------------------------------------------------------
KN_FILTERS_U = ["g", "r", "i", "z", "y"]
GRB_FILTERS_U = ["g", "r", "i", "J", "w"]
OBS_FILTERS_U = ["g", "r", "i", "z", "y", "J", "w"]


def case_combined_union(seed=9223, batch=40):
    mp, svd = syn.make_svd_model(seed, KN_FILTERS_U, model="Bu2019lm")
    grid = syn.flat_lcdm_grid(1.0, 200.0)
    counts = dict(g=13, r=19, i=20, z=18, y=15)
    times, mags, sigmas = syn.make_photometry(seed + 1, svd, mp, filters=KN_FILTERS_U, counts=counts, cosmo_grid=grid,
                                              upper_limit_filter="i")
    rng = np.random.default_rng(seed + 4)
    for f, n in (("J", 9), ("w", 11)):
        t = np.sort(rng.uniform(0.6, 13.0, n))
        base = np.mean([np.interp(t, times[g], mags[g]) for g in ("g", "r", "i")], axis=0)
        times[f], mags[f], sigmas[f] = t, base + (0.4 if f == "J" else 0.0) + 0.05 * rng.standard_normal(n), rng.uniform(0.03, 0.15, n)
    names, theta = syn.draw_theta(seed + 2, batch, NAMES[:6])
    theta = np.concatenate([theta, rng.uniform(-17.5, -14.0, (batch, 1)), rng.uniform(0.8, 1.6, (batch, 1))], axis=1)
    return dict(model="Bu2019lm", model_parameters=mp, svd=svd, filters=KN_FILTERS_U, grb_filters=GRB_FILTERS_U,
                observed_filters=OBS_FILTERS_U, sample_times=np.arange(0.1, 20.5, 0.5),
                grb_times=np.geomspace(0.25, 30.0, 36), cosmo_grid=grid, data=(times, mags, sigmas), names=NAMES, theta=theta)


def oracle_likelihood_union(case, use_scipy=True):
    from oracle import nmma_oracle as orc
    kn = orc.OracleSVDModel(case["model_parameters"], case["svd"], filters=case["filters"],
                            sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    grb = orc.OraclePowerLawModel(case["grb_filters"], case["grb_times"], cosmo_grid=case["cosmo_grid"])
    comb = orc.OracleCombinedModel([kn, grb])
    obs = case["observed_filters"]
    return orc.OracleLikelihood(comb, case["data"], dict(mode="budget", values={f: 1.0 for f in obs}), obs,
                                detection_limit=np.inf, known_filters=[f for f in obs if f != "w"], use_scipy=use_scipy), grb


# ---------------------------------------------------------------------------------------------------------------
# Own time grids (model.py:1372-1374, :1440-1448), filters the surrogate lists completely: the combination the one-launch form
# covers since round 6 (``base_times``).  The kilonova lives on 0.1 .. 14.1 d (its SVD grid ends at 21 d), the second transient on a
# log-spaced grid 0.25 .. 30 d that starts at 0.3 d, lacks one filter (no flux there) and has interior holes for the steeper half
# of its slope prior; the photometry reaches beyond the kilonova's last node (flux sum = second transient alone there) and, for
# early time shifts, beyond every node (floor).
# ---------------------------------------------------------------------------------------------------------------
def case_combined_owngrids(seed=9523, batch=48):
    mp, svd = syn.make_svd_model(seed, FILTERS, model="Bu2019lm")
    grid = syn.flat_lcdm_grid(1.0, 200.0)
    counts = dict(syn.AT2017GFO_COUNTS)
    counts["sdssu"] = 6
    data = syn.make_photometry(seed + 1, svd, mp, filters=FILTERS, counts=counts, cosmo_grid=grid, t_range=(0.3, 19.0))
    # (the filter the second transient lacks is the kilonova's alone: its epochs stay inside the kilonova's window for most time shifts)
    keep = data[0]["2massh"] < 13.3
    for d in data:
        d["2massh"] = d["2massh"][keep]
    names, theta = syn.draw_theta(seed + 2, batch, NAMES[:6])
    rng = np.random.default_rng(seed + 3)
    theta = np.concatenate([theta, rng.uniform(-17.5, -14.0, (batch, 1)), rng.uniform(0.8, 1.6, (batch, 1))], axis=1)
    return dict(model="Bu2019lm", model_parameters=mp, svd=svd, filters=FILTERS, grb_filters=[f for f in FILTERS if f != "2massh"],
                observed_filters=FILTERS, sample_times=np.arange(0.1, 14.6, 0.5), grb_times=np.geomspace(0.25, 30.0, 36),
                grb_hole=(14, 16, 1.2), cosmo_grid=grid, data=data, names=NAMES, theta=theta)


def oracle_likelihood_owngrids(case, use_scipy=True):
    from oracle import nmma_oracle as orc
    kn = orc.OracleSVDModel(case["model_parameters"], case["svd"], filters=case["filters"],
                            sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    grb = orc.OraclePowerLawModel(case["grb_filters"], case["grb_times"], cosmo_grid=case["cosmo_grid"], hole=case.get("grb_hole"))
    comb = orc.OracleCombinedModel([kn, grb])
    obs = case["observed_filters"]
    return orc.OracleLikelihood(comb, case["data"], dict(mode="budget", values={f: 1.0 for f in obs}), obs,
                                detection_limit=np.inf, known_filters=obs, use_scipy=use_scipy), grb


# ---------------------------------------------------------------------------------------------------------------
# Null filters (lightcurve_generation.py:168-169: "null output for other filters, especially radio and X-ray filters when using
# with GRB data"): the drivers' shared grid and filter list (config 3's shape), two of the listed filters without a surrogate --
# the kilonova is +inf there on every node and the band's curve is the afterglow's alone.
# ---------------------------------------------------------------------------------------------------------------
NULL_FILTERS = ["radio-3GHz", "X-ray-1keV"]


def case_combined_nullfilters(seed=9723, batch=40):
    c = case_combined(seed=seed, batch=batch)
    rng = np.random.default_rng(seed + 7)
    times, mags, sigmas = ({k: v for k, v in d.items()} for d in c["data"])
    for k, f in enumerate(NULL_FILTERS):
        n = 7 + 4 * k
        t = np.sort(rng.uniform(0.6, 13.0, n))
        sig = rng.uniform(0.05, 0.2, n)
        m = -16.0 + 2.5 * 1.2 * np.log10(t) + 0.15 * (len(FILTERS) + k) + 5.0 * (5 + np.log10(40.0)) + sig * rng.standard_normal(n)
        times[f], mags[f], sigmas[f] = t, m, sig
    sigmas["X-ray-1keV"][2] = np.inf            # (an upper limit in a band only the afterglow has)
    c["data"] = (times, mags, sigmas)
    c["all_filters"] = FILTERS + NULL_FILTERS
    c["grb_hole"] = (9, 12, 1.2)
    return c


def oracle_likelihood_nullfilters(case, use_scipy=True):
    from oracle import nmma_oracle as orc
    allf = case["all_filters"]
    kn = orc.OracleSVDModel(case["model_parameters"], case["svd"], filters=allf, sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    grb = orc.OraclePowerLawModel(allf, case["sample_times"], cosmo_grid=case["cosmo_grid"], hole=case.get("grb_hole"))
    comb = orc.OracleCombinedModel([kn, grb])
    return orc.OracleLikelihood(comb, case["data"], dict(mode="budget", values={f: 1.0 for f in allf}), allf,
                                detection_limit=np.inf, known_filters=allf, use_scipy=use_scipy), grb


def case_combined_limit(seed=9823, batch=40):
    """The shared-grid combination under finite detection limits (truncated Gaussian per detection, em_likelihood.py:252-256): a limit
    0.3 mag above the faintest datum of every filter -- except one, where a detection is fainter than its limit (-inf for every sample:
    the reference's floor) in the variant ``violated=True``."""
    c = case_combined(seed=seed, batch=batch)
    c["detection_limit"] = {f: float(np.max(c["data"][1][f][np.isfinite(c["data"][2][f])]) + 0.3) for f in c["filters"]}
    c["grb_hole"] = (9, 12, 1.2)
    return c


def case_combined_nodes(seed=9923, batch=40):
    """The shared-grid combination with time-dependent systematics (systematics.py:288-296: four linear time nodes shared by two blue
    bands, four for 2massj, one sampled parameter for the rest), non-finite sampled node values in a few rows (autocomplete_data's
    mask: fewer than two finite nodes turn the group's detections into upper limits) and holes in the afterglow's curves."""
    c = case_combined(seed=seed, batch=batch)
    rng = np.random.default_rng(seed + 9)
    g1 = ["ps1::g", "ps1::r"]
    nodes = np.linspace(0.0, 21.0, 4)
    n_a = [f"em_syserr_blue_{i}" for i in range(4)]
    n_b = [f"em_syserr_2massj_{i}" for i in range(4)]
    extra = ["em_syserr_rest"] + n_a + n_b
    c["names"] = NAMES + extra
    th = np.concatenate([c["theta"], rng.uniform(0.1, 1.5, (batch, len(extra)))], axis=1)
    names = c["names"]
    th[1, names.index(n_a[1])] = np.nan                                            # a middle node
    th[2, names.index(n_a[0])] = np.inf                                            # the first node
    th[3, [names.index(n) for n in n_b[:3]]] = np.nan                              # one finite node left: 2massj's detections become upper limits
    th[4, names.index(n_b[3])] = -np.inf
    c["theta"] = th
    c["systematics"] = dict(mode="mixed", names={f: "em_syserr_rest" for f in FILTERS if f not in g1 + ["2massj"]},
                            nodes={**{f: (n_a, nodes) for f in g1}, "2massj": (n_b, nodes)})
    c["systematics_ref"] = dict(error_budget=None, systematics_file={
        "blue": {"filters": g1, "time_nodes": 4, "time_range": "lin 0.0 21.0"},
        "2massj": {"time_nodes": 4, "time_range": "lin 0.0 21.0"},
        "rest": {"prior": "unused"}})
    c["grb_hole"] = (9, 12, 1.2)
    return c
