"""Host-side helpers of the EM sector (setup-time only; nothing here runs per sample).

Mirrors the pieces of ``nmma/em/utils.py`` the likelihood path needs:
filter-name maps (:478-584), per-filter dict broadcasting (:218-235), photometry
preparation (:255-286) and the model-window consistency check (:289-353).
"""
from __future__ import annotations

import numpy as np

#: names the reference maps to themselves without consulting sncosmo (utils.py:486-510)
BUILTIN_FILTERS = [
    "u", "g", "r", "i", "z", "y", "J", "H", "K", "X-ray-1keV", "X-ray-5keV", "radio-5.5GHz",
    "radio-1.25GHz", "radio-6GHz", "radio-3GHz", "sdss::u", "sdss::g", "sdss::r", "sdss::i",
    "sdss::z", "swope2::y", "swope2::J", "swope2::H",
]
#: hard-coded renames (utils.py:519-529)
FILTER_RENAMES = {"B": "g", "R": "z", "F160W": "H", "U": "u", "UVW2": "u", "UVW1": "u", "UVM2": "u"}
#: observed bands built as the mean of model bands (utils.py:549-563, :566-584)
FILTER_AVERAGES = {"w": ["g", "r", "i"], "o": ["r", "i"], "c": ["g", "r"], "V": ["g", "r"],
                   "F606W": ["g", "r"], "I": ["z", "y"], "F814W": ["z", "y"]}


def get_filter_name_mapping(observed_filters, known_filters=()):
    """``(direct_map, averaging_map)`` as ``nmma.em.utils.get_filter_name_mapping``.

    ``known_filters`` stands in for the sncosmo bandpass registry the reference
    consults (utils.py:511): every name in it maps to itself."""
    if isinstance(observed_filters, str):
        observed_filters = [observed_filters]
    known = set(BUILTIN_FILTERS) | set(known_filters)
    direct, averaging = {}, {}
    for f in observed_filters:
        if f in FILTER_RENAMES:
            direct[f] = FILTER_RENAMES[f]
        elif f in known or f.startswith("radio") or f.startswith("X-ray"):
            direct[f] = f
        elif f in FILTER_AVERAGES:
            averaging[f] = list(FILTER_AVERAGES[f])
        else:
            raise ValueError(f"Unknown filter: {f}. Cannot be processed")
    return direct, averaging


def set_filter_associated_dict(quantity, filters, default_limit=np.inf):
    """utils.py:218-235."""
    if isinstance(quantity, (int, float)):
        return {x: float(quantity) for x in filters}
    if isinstance(quantity, (list, tuple)):
        assert len(quantity) == len(filters), f" {quantity} must match the number of filters: {filters}."
        return {x: float(y) for x, y in zip(filters, quantity)}
    if isinstance(quantity, dict):
        return {filt: float(quantity.get(filt, default_limit)) for filt in filters}
    raise ValueError(f"Could not derive a dict for {quantity} and filters {filters}.")


def setup_filtered_lc_data(light_curve_data, trigger_time):
    """utils.py:255-286: ``{filt: {time, mag, mag_error}}`` -> (times, mags, sigmas, trigger_time)
    with times in days since the trigger."""
    lc_times, lc_mags, lc_unc = {}, {}, {}
    min_time = np.inf
    for filt, sub in light_curve_data.items():
        lc_mags[filt] = np.array(sub["mag"])
        lc_unc[filt] = np.array(sub["mag_error"])
        lc_times[filt] = np.array(sub["time"])
        min_time = np.minimum(min_time, np.min(sub["time"]))
    if min_time < 0:
        raise ValueError(f"trigger_time is {-min_time} days later than earliest data time. "
                         "Please provide a valid trigger time.")
    lc_times = {filt: lc_times[filt] - trigger_time for filt in lc_times}
    return (lc_times, lc_mags, lc_unc, trigger_time)


def cut_data_to_time_range(data, args=None, trigger_time=0.0, tmin=0, tmax=np.inf):
    """utils.py:233-253: keep the rows of every filter whose time since the trigger lies in [data_tmin, data_tmax]
    (taken from ``args`` when it has them), dropping filters left empty.  ``data``: ``{filt: {time, mag, mag_error}}`` in MJD."""
    tmin = getattr(args, "data_tmin", tmin)
    tmax = getattr(args, "data_tmax", tmax)
    for filt in list(data.keys()):
        detector_time = np.asarray(data[filt]["time"]) - trigger_time
        mask = (tmin <= detector_time) & (detector_time <= tmax)
        if not np.any(mask):
            del data[filt]
        else:
            data[filt] = {k: np.asarray(data[filt][k])[mask] for k in ("time", "mag", "mag_error")}
    return data


def _prior_bounds(priors, key, default=None):
    """(minimum, maximum) of a prior; a plain number or delta function counts as both."""
    if key not in priors:
        return default
    prior = priors[key]
    if isinstance(prior, (int, float, np.integer, np.floating)):
        return float(prior), float(prior)
    if hasattr(prior, "minimum") and hasattr(prior, "maximum") and prior.minimum is not None:
        return float(prior.minimum), float(prior.maximum)
    peak = getattr(prior, "peak", None)
    if peak is None:
        raise ValueError(f"prior on {key!r} has neither bounds nor a fixed value")
    return float(peak), float(peak)


def observer_frame_window(model_times, priors):
    """The observer-frame interval every prior draw's model covers: the model grid [t_lo, t_hi] (source
    frame, days) stretched by (1 + z) and shifted by ``timeshift``; taking the latest possible start
    and the earliest possible end (utils.py:301-329)."""
    from ..core import conversion
    if "redshift" in priors:
        z_lo, z_hi = _prior_bounds(priors, "redshift")
    elif "luminosity_distance" in priors:
        d_lo, d_hi = _prior_bounds(priors, "luminosity_distance")
        if "Hubble_constant" in priors:
            # the reference evaluates every prior at its minimum, then at its maximum
            ends = []
            for pick in (0, 1):
                corner = {k: _prior_bounds(priors, k)[pick] for k in ("luminosity_distance", "Hubble_constant",
                                                                      "Omega_matter") if k in priors}
                ends.append(conversion.cosmology_to_distance(corner)["redshift"])
            z_lo, z_hi = ends
        else:
            z_lo = conversion.luminosity_distance_to_redshift(d_lo)
            z_hi = conversion.luminosity_distance_to_redshift(d_hi)
    else:
        z_lo = z_hi = 0.0
    shift_lo, shift_hi = _prior_bounds(priors, "timeshift", (0.0, 0.0))
    t_lo, t_hi = float(model_times[0]), float(model_times[-1])
    return (1.0 + z_hi) * t_lo + shift_hi, (1.0 + z_lo) * t_hi + shift_lo


def check_model_time_consistency(light_curve_data, light_curve_model, priors, injection=None):
    """Refuse (or, for an injection, trim) photometry that falls outside the window the model covers for
    every prior draw (utils.py:289-353).  Detections only: non-finite magnitudes / uncertainties do not
    count towards the data's time span.  Returns the (possibly trimmed) data tuple."""
    lc_times, lc_mags, lc_unc, trigger_time = light_curve_data
    window_start, window_end = observer_frame_window(light_curve_model.model_times, priors)
    if injection is not None:
        for filt in list(lc_times):
            keep = (lc_times[filt] >= window_start) & (lc_times[filt] <= window_end)
            lc_times[filt] = lc_times[filt][keep]
            lc_mags[filt] = lc_mags[filt][keep]
            lc_unc[filt] = lc_unc[filt][keep]
        return (lc_times, lc_mags, lc_unc, trigger_time)
    first, last = np.inf, -np.inf
    for filt, t in lc_times.items():
        detected = np.isfinite(lc_mags[filt]) & np.isfinite(lc_unc[filt])
        if np.any(detected):
            first, last = min(first, t[detected].min()), max(last, t[detected].max())
    if first < window_start:
        raise ValueError(f"First data point is at {first} days, but with your timeshift and redshift settings, "
                         f"the model time in detector frame can start as late as {window_start}.")
    if window_end < last:
        raise ValueError(f"Last data point is at {last} days, but with your timeshift and redshift settings, "
                         f"the model time in detector frame can end as early as {window_end}.")
    return (lc_times, lc_mags, lc_unc, trigger_time)


def resolve_sources(observed_filters, model_filters, known_filters=()):
    """Model bands feeding each observed band (em_likelihood.py:313-335).

    Reference quirk kept: helper bands of an averaged filter are looked up in the map
    built from the *observed* filters (em_likelihood.py:330), so each helper must itself
    be observed; and a mapped band the model does not provide is an error."""
    # (an averaged name -- w, o, c, V, I ... -- is never a registry name: it keeps its helper-band mean even when a
    #  combined model lists a band of that name, utils.py:478-546)
    direct, averaging = get_filter_name_mapping(
        observed_filters, {f for f in set(known_filters) | set(model_filters) if f not in FILTER_AVERAGES})
    sources = {}
    for f in observed_filters:
        if f in direct:
            names = [direct[f]]
        else:
            names = []
            for h in averaging[f]:
                if h not in direct:
                    raise KeyError(h)
                names.append(direct[h])
        for n in names:
            if n not in model_filters:
                raise KeyError(f"model provides no light curve for filter {n!r} (needed by {f!r})")
        sources[f] = names
    return sources


def setup_sample_times(args):
    """The model's sample times from the drivers' arguments (em/utils.py:72-93): None when neither ``em_tmin`` nor ``em_tmax`` is
    given (the model then decides), ``arange`` with a fixed ``em_tstep`` (the legacy form), else ``em_nsteps`` linearly or
    geometrically spaced nodes (``em_timescale``)."""
    tmin, tmax = getattr(args, "em_tmin", None), getattr(args, "em_tmax", None)
    if tmin is None and tmax is None:
        return None
    tstep = getattr(args, "em_tstep", None)
    if tstep:
        return np.arange(tmin, tmax + tstep, tstep)
    scale = getattr(args, "em_timescale", "linear") or "linear"
    nsteps = getattr(args, "em_nsteps", None)
    if "lin" in scale or tmin <= 0.0:
        return np.linspace(tmin, tmax, nsteps)
    if any(k in scale for k in ("log", "geo")):
        return np.geomspace(tmin, tmax, nsteps)
    raise ValueError(f"Unknown time scale {scale}. Please use 'lin(ear)' or 'log(arithmic)' / 'geo(metric)'.")


#: filters of the survey names ``--em-detectors`` takes, the Rubin target-of-opportunity strategies, and the surveys' single-visit
#: depths (em/utils.py:96-196; arXiv:2108.01683 and the Rubin key numbers for LSST / Rubin, the ZTF survey depths)
DETECTOR_FILTERS = {"ztf": ["ztfg", "ztfr", "ztfi"], "lsst": ["lsstg", "lsstr", "lssti", "lsstz", "lssty"],
                    "rubin": ["ps1::g", "ps1::r", "ps1::i", "ps1::z", "ps1::y"]}
RUBIN_TOO_FILTERS = {"platinum": ["ps1::g", "ps1::r", "ps1::i", "ps1::z", "ps1::y"], "gold": ["ps1::g", "ps1::r", "ps1::i"],
                     "gold_z": ["ps1::g", "ps1::r", "ps1::z"], "silver": ["ps1::g", "ps1::i"], "silver_z": ["ps1::g", "ps1::z"]}
DETECTOR_LIMITS = {"lsst": {"lsstu": 23.9, "lsstg": 25.0, "lsstr": 24.7, "lssti": 24.0, "lsstz": 23.3, "lssty": 22.1},
                   "ztf": {"ztfg": 21.7, "ztfr": 21.4, "ztfi": 20.9},
                   "rubin": {"ps1::g": 25.8, "ps1::r": 25.5, "ps1::i": 24.8, "ps1::z": 24.1, "ps1::y": 22.9}}


def _detector_list(args):
    dets = getattr(args, "em_detectors", None) or []
    dets = dets.split(",") if isinstance(dets, str) else list(dets)
    return dets


def set_filters(args):
    """The analysis' filter list from the drivers' arguments (em/utils.py:96-139): ``--filters`` (comma-separated, blanks and empty
    items dropped) wins; else the filters of the ``--em-detectors`` surveys, where LSST takes precedence over a Rubin ToO strategy
    and that over plain ``rubin``; None when nothing is given (the likelihood then takes the filters of the data)."""
    if getattr(args, "filters", None):
        filters = args.filters.split(",") if isinstance(args.filters, str) else args.filters
        filters = [f for item in filters for f in item.replace(" ", "").split(",") if f]
        if not filters:
            raise ValueError("Need at least one valid filter.")
        return filters
    too = getattr(args, "rubin_ToO_type", False)
    if not (getattr(args, "em_detectors", None) or too):
        return None
    dets = [d.strip().lower() for d in _detector_list(args)]
    filters = []
    if "ztf" in dets:
        dets.remove("ztf")
        filters += DETECTOR_FILTERS["ztf"]
    if "lsst" in dets:
        dets.remove("lsst")
        filters += DETECTOR_FILTERS["lsst"]
    elif too:
        filters += RUBIN_TOO_FILTERS.get(too, [])
        if "rubin" in dets:
            dets.remove("rubin")
    elif "rubin" in dets:
        dets.remove("rubin")
        filters += DETECTOR_FILTERS["rubin"]
    if dets:
        raise NotImplementedError(f"{dets} not implemented yet.")
    return filters


def create_detection_limit(args, filters, default_limit=np.inf):
    """The likelihood's ``detection_limit`` dict from the drivers' arguments (em/utils.py:142-196): an explicit
    ``--detection-limit`` (number, list or dict) for the given filters, else ``default_limit`` per filter updated with the depths of
    the ``--em-detectors`` surveys / a Rubin ToO strategy.  (A limit read from an m4opt FITS sky map -- ``--detection-limit-fits-file``
    -- needs astropy and healpy: read it with those and pass the number as ``detection_limit``.)"""
    if getattr(args, "detection_limit", None):
        return set_filter_associated_dict(args.detection_limit, filters, default_limit)
    if getattr(args, "detection_limit_fits_file", None):
        raise NotImplementedError("detection limits from an m4opt FITS sky map need astropy + healpy: pass the limit as detection_limit")
    limits = {f: default_limit for f in filters}
    if getattr(args, "em_detectors", None):
        dets = _detector_list(args)
        for name in ("lsst", "ztf", "rubin"):
            if name in dets:
                dets.remove(name)
                limits.update(DETECTOR_LIMITS[name])
        if dets:
            raise NotImplementedError(f"{dets} not implemented yet.")
    if getattr(args, "rubin_ToO_type", None):
        limits.update(DETECTOR_LIMITS["rubin"])
    return limits
