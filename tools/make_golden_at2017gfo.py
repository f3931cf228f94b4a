#!/usr/bin/env python
"""The real-data golden case: (1) writes the fixture tests/golden/at2017gfo_photometry.npz -- the arrays the reference's
reader produces for its example data set example_files/lightcurves/AT2017gfo.dat (the time column parsed as the reference does,
ISO date -> MJD; em/io.py:116-144) -- and (2) runs the REFERENCE'S OWN preparation chain on them (cut_data_to_time_range,
setup_filtered_lc_data, check_model_time_consistency: em/utils.py:233-353) and checks that this repository's chain produces
identical arrays; then tools/make_golden.py's machinery writes tests/golden/at2017gfo.npz from the reference's likelihood.

Runs only in the build container.  Usage: python tools/make_golden_at2017gfo.py"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from nmma_amd.em import io as em_io  # noqa: E402

SRC = os.path.join(ref_harness.REFERENCE_ROOT, "example_files", "lightcurves", "AT2017gfo.dat")


def write_fixture():
    # (the reference's strict_read_csv keeps file order; astropy's Time(...).mjd of an ISO date is what _to_mjd computes)
    data = {}
    with open(SRC) as fh:
        for line in fh:
            parts = line.split()
            if len(parts) < 4 or line.startswith(("#", "time", "mjd")):
                continue
            d = data.setdefault(parts[1], {"time": [], "mag": [], "mag_error": []})
            d["time"].append(em_io._to_mjd(parts[0])); d["mag"].append(float(parts[2])); d["mag_error"].append(float(parts[3]))
    out = {"filters": np.array(list(data))}
    for f, d in data.items():
        for k, v in d.items():
            out[f"{f}/{k}"] = np.array(v, float)
    from tests import cases
    np.savez_compressed(os.path.join(cases.GOLDEN_DIR, "at2017gfo_photometry.npz"), **out)
    n = sum(len(d["time"]) for d in data.values())
    print(f"fixture: {len(data)} filters, {n} rows, {sum(int(np.sum(~np.isfinite(d['mag_error']))) for d in data.values())} upper limits")
    # this repository's reader returns the same arrays (it sorts by time; the file is time-ordered per filter)
    mine = em_io.load_em_observations(SRC)
    for f, d in data.items():
        for k in d:
            assert np.array_equal(mine[f][k], np.array(d[k], float)), (f, k)


def check_preparation_chain():
    from tests import cases
    ref = ref_harness.reference_modules()
    case = cases.case_at2017gfo()
    grid = case["cosmo_grid"]
    ref.utils.luminosity_distance_to_redshift = lambda d: float(np.interp(d, *grid))     # astropy is a stub in the harness
    raw = cases.at2017gfo_raw_photometry()
    lo, hi = case["data_window"]
    raw = ref.utils.cut_data_to_time_range(raw, types.SimpleNamespace(), case["trigger_time"], tmin=lo, tmax=hi)
    data = ref.utils.setup_filtered_lc_data(raw, case["trigger_time"])
    priors = {k: types.SimpleNamespace(minimum=v[0], maximum=v[1]) for k, v in case["prior_bounds"].items()}
    model = types.SimpleNamespace(model_times=case["sample_times"])
    times, mags, sigmas, trig = ref.utils.check_model_time_consistency(data, model, priors)
    assert list(times) == case["observed_filters"]
    for f in times:
        assert np.array_equal(times[f], case["data"][0][f]) and np.array_equal(mags[f], case["data"][1][f])
        assert np.array_equal(sigmas[f], case["data"][2][f])
    n = sum(len(t) for t in times.values())
    print(f"preparation chain: reference and repository agree on {len(times)} filters, {n} rows after the 14.5 d cut")
    # the uncut data set is refused by both, with the same wording
    full = ref.utils.setup_filtered_lc_data(cases.at2017gfo_raw_photometry(), case["trigger_time"])
    try:
        ref.utils.check_model_time_consistency(full, model, priors)
        raise AssertionError("the reference accepted data beyond the model window")
    except ValueError as exc:
        print("uncut data refused by the reference:", str(exc)[:70], "...")


if __name__ == "__main__":
    write_fixture()
    check_preparation_chain()
    from tools import make_golden
    make_golden.run_case("at2017gfo")
