"""Light-curve data files in the reference's formats (nmma/em/io.py:16-184): every form ``load_em_observations`` accepts and
``write_em_observations`` produces, on hand-written files and round trips, plus the reference's own example file."""
import argparse
import json
import os

import numpy as np
import pytest

from nmma_amd.em.io import load_em_observations, write_em_observations

ROWS = """# time filter mag mag_error
2017-08-18T00:00:00.000 ps1::g 17.41 0.02
2017-08-19T12:00:00.000 ps1::g 18.90 0.05
2017-08-18T06:00:00.000 2massks 17.60 0.10
2017-08-25T00:00:00.000 2massks 21.20 inf
57984.5 ps1::g 19.70 0.08
"""


def test_observation_rows(tmp_path):
    path = tmp_path / "event.dat"
    path.write_text(ROWS)
    data = load_em_observations(str(path))
    assert set(data) == {"ps1::g", "2massks"}
    g = data["ps1::g"]
    np.testing.assert_allclose(g["time"], [57983.0, 57984.5, 57984.5])          # ISO and MJD tokens, sorted by time
    assert sorted(g["mag"].tolist()) == [17.41, 18.90, 19.70] and g["mag"][0] == 17.41
    k = data["2massks"]
    assert np.isinf(k["mag_error"][1]) and k["mag"][1] == 21.20                   # an upper limit
    assert set(load_em_observations(str(path), filters=["2massks"])) == {"2massks"}
    # a Namespace carries the file name and the time format (io.py:40-42, :130-136)
    jd = tmp_path / "jd.dat"
    jd.write_text("time filter mag mag_error\n2457983.5 g 17.0 0.1\n2457984.5 g 18.0 0.1\n")
    got = load_em_observations(argparse.Namespace(light_curve_data=str(jd), time_format="jd"))
    np.testing.assert_allclose(got["g"]["time"], [57983.0, 57984.0])
    assert load_em_observations({"g": {"time": [1.0]}}) == {"g": {"time": [1.0]}}            # a dict is taken as it is
    with pytest.raises(ValueError):
        load_em_observations(None)
    with pytest.raises(ValueError):
        load_em_observations(str(path), format="standard")


def test_forced_photometry_csv(tmp_path):
    """The fallback of the observations reader (io.py:77-82, :101-114): rows without a magnitude are limits at ``limiting_mag``."""
    path = tmp_path / "forced.csv"
    path.write_text("mjd,filter,mag_corr,magerr,limiting_mag\n59000.1,ztfg,19.5,0.1,20.5\n59001.1,ztfg,,,20.9\n59000.6,ztfr,19.1,0.07,20.4\n")
    data = load_em_observations(str(path))
    np.testing.assert_allclose(data["ztfg"]["time"], [59000.1, 59001.1])
    np.testing.assert_allclose(data["ztfg"]["mag"], [19.5, 20.9])
    assert data["ztfg"]["mag_error"][0] == 0.1 and np.isinf(data["ztfg"]["mag_error"][1])
    assert data["ztfr"]["mag"].tolist() == [19.1]


def test_model_tables_and_json(tmp_path):
    table = tmp_path / "model.dat"
    table.write_text("# time g r g_error\n0.5 -15.0 -15.5 0.1\n1.5 -14.0 -14.6 0.2\n")
    data = load_em_observations(str(table), format="model")
    assert set(data) == {"g", "r"}
    np.testing.assert_allclose(data["g"]["mag_error"], [0.1, 0.2])
    np.testing.assert_allclose(data["r"]["mag_error"], [0.0, 0.0])              # no error column: zeros (io.py:95)
    np.testing.assert_allclose(data["r"]["time"], [0.5, 1.5])
    standard = tmp_path / "standard.json"
    standard.write_text(json.dumps({"g": {"time": [1.0, 2.0], "mag": [18.0, 19.0], "mag_error": [0.1, 0.2]}}))
    got = load_em_observations(str(standard))
    assert isinstance(got["g"]["mag"], np.ndarray) and got["g"]["mag"].tolist() == [18.0, 19.0]
    model = tmp_path / "model.json"           # (bilby's array encoding, as the reference's injections are written)
    model.write_text(json.dumps({"time": {"__array__": True, "content": [0.5, 1.5]}, "g": [-15.0, -14.0], "g_error": [0.1, 0.2],
                                 "r": {"__array__": True, "content": [-15.5, -14.6]}}))
    got = load_em_observations(str(model))
    assert set(got) == {"g", "r"} and got["r"]["mag"].tolist() == [-15.5, -14.6] and got["r"]["mag_error"].tolist() == [0.0, 0.0]
    assert got["g"]["mag_error"].tolist() == [0.1, 0.2] and got["g"]["time"].tolist() == [0.5, 1.5]


def test_round_trips(tmp_path):
    rng = np.random.default_rng(5)
    data = {f: {"time": np.sort(rng.uniform(57983.0, 57995.0, n)), "mag": rng.uniform(17.0, 22.0, n), "mag_error": rng.uniform(0.02, 0.3, n)}
            for f, n in (("ps1::g", 7), ("2massh", 4))}
    data["2massh"]["mag_error"][2] = np.inf
    write_em_observations(str(tmp_path / "out" / "lc.json"), data)
    back = load_em_observations(str(tmp_path / "out" / "lc.json"))
    for f in data:
        for k in data[f]:
            np.testing.assert_array_equal(back[f][k], data[f][k])
    write_em_observations(str(tmp_path / "out" / "lc.dat"), data)
    text = (tmp_path / "out" / "lc.dat").read_text().splitlines()
    assert text[0] == "#time filter mag mag_error" and len(text) == 12
    assert [ln.split()[0] for ln in text[1:]] == sorted(ln.split()[0] for ln in text[1:])        # sorted by time (io.py:167)
    back = load_em_observations(str(tmp_path / "out" / "lc.dat"))
    for f in data:
        np.testing.assert_allclose(back[f]["time"], data[f]["time"], atol=1e-8)                  # ISOT with milliseconds
        np.testing.assert_allclose(back[f]["mag"], data[f]["mag"], atol=5.1e-4)                  # three decimals
        assert np.array_equal(np.isinf(back[f]["mag_error"]), np.isinf(data[f]["mag_error"]))
    model = {f: {"time": np.array([0.5, 1.5, 2.5]), "mag": np.array([-15.0, -14.2, -13.1]) + i, "mag_error": np.full(3, np.nan if i else 0.1)}
             for i, f in enumerate(("g", "r"))}
    write_em_observations(str(tmp_path / "model.txt"), model, format="model")
    assert (tmp_path / "model.txt").read_text().splitlines()[0] == "#time g r g_error"
    back = load_em_observations(str(tmp_path / "model.txt"), format="model")
    np.testing.assert_allclose(back["r"]["mag"], model["r"]["mag"])
    np.testing.assert_allclose(back["g"]["mag_error"], [0.1, 0.1, 0.1])


@pytest.mark.skipif(not os.path.isfile("/root/reference/example_files/lightcurves/AT2017gfo.dat"), reason="the reference tree is not here")
def test_the_reference_example_file():
    data = load_em_observations("/root/reference/example_files/lightcurves/AT2017gfo.dat")
    assert sum(len(v["time"]) for v in data.values()) == 141 and len(data) == 9
    assert sum(int(np.isinf(v["mag_error"]).sum()) for v in data.values()) == 3
    assert all(np.all(np.diff(v["time"]) >= 0) for v in data.values())
