"""The legacy systematics file as a source of prior strings (nmma/em/systematics.py:340-513): the reference's own test module
(nmma/tests/systematics.py) restated for ``nmma_amd.em.systematics`` -- same inputs, same expectations -- and, where the reference is
present, every validator and handler run side by side with the reference's functions on accepted and rejected documents."""
import os

import pytest
from yaml import YAMLError

from nmma_amd.em.systematics import (ALLOWED_DISTRIBUTIONS, ALLOWED_FILTERS, ValidationError, get_prior_strings, handle_withoutTime,
                                     handle_withTime, load_yaml, main, validate_distribution, validate_filters, validate_only_one_true)

SAMPLE = """
config:
  withTime:
    value: true
    type: Uniform
    minimum: 0.0
    maximum: 1.0
    time_nodes: 2
    filters:
      - [bessellb, bessellv]
      - ztfr
  withoutTime:
    value: false
    type: Uniform
    minimum: 0.0
    maximum: 1.0
"""
UNIFORM = "em_syserr = Uniform(minimum=0.0, maximum=1.0, name='em_syserr', latex_label='em_syserr', unit=None, boundary=None)"


@pytest.fixture
def sample_yaml_file(tmp_path):
    path = tmp_path / "test_config.yaml"
    path.write_text(SAMPLE)
    return path


def test_one_switch_only(sample_yaml_file):
    validate_only_one_true(load_yaml(sample_yaml_file))
    with pytest.raises(ValidationError, match="Only one configuration key can be set to True at a time"):
        validate_only_one_true({"config": {"withTime": {"value": True}, "withoutTime": {"value": True}}})
    with pytest.raises(ValidationError, match="At least one configuration key must be set to True"):
        validate_only_one_true({"config": {"withTime": {"value": False}, "withoutTime": {"value": False}}})
    with pytest.raises(ValidationError, match="'value' key must be present and be a boolean"):
        validate_only_one_true({"config": {"withTime": {}, "withoutTime": {"value": False}}})
    with pytest.raises(ValidationError, match="Validation error for 'withTime'"):
        validate_only_one_true({"config": {"withTime": {"value": "yes"}, "withoutTime": {"value": False}}})


def test_filter_groups():
    validate_filters([["bessellb", "bessellv"], "ztfr"])
    validate_filters([["bessellb", "bessellv"], None])
    validate_filters([])
    with pytest.raises(ValidationError, match="Invalid filter value 'invalid_filter'"):
        validate_filters([["bessellb", "invalid_filter"], "ztfr"])
    with pytest.raises(ValidationError, match="Invalid filter value 'nope'"):
        validate_filters(["nope"])
    with pytest.raises(ValidationError, match="Duplicate filter value 'bessellb' within the same group"):
        validate_filters([["bessellb", "bessellb"], "ztfr"])
    with pytest.raises(ValidationError, match="Duplicate filter value 'bessellb'. A filter can only be used in one group"):
        validate_filters([["bessellb", "bessellv"], "bessellb"])
    with pytest.raises(ValidationError, match="Duplicate filter value 'ztfr'. A filter can only be used in one group"):
        validate_filters(["ztfr", ["ztfg", "ztfr"]])


def test_distributions():
    assert ALLOWED_DISTRIBUTIONS["Uniform"]
    for wrong in ("nonuniform", "uniform"):          # (case-sensitive class names)
        with pytest.raises(KeyError):
            assert ALLOWED_DISTRIBUTIONS[wrong]
    validate_distribution({"type": "Uniform", "minimum": 0.0, "maximum": 1.0})
    with pytest.raises(ValidationError, match="Invalid distribution 'uniform'"):
        validate_distribution({"type": "uniform", "minimum": 0.0, "maximum": 1.0})
    with pytest.raises(ValidationError, match="Missing required parameters for Uniform distribution: maximum"):
        validate_distribution({"type": "Uniform", "minimum": 0.0})
    with pytest.raises(ValidationError, match="Missing required parameters for Gaussian distribution"):
        validate_distribution({"type": "Gaussian", "mu": 0.0})


def test_handle_withTime():
    values = {"type": "Uniform", "minimum": 0.0, "maximum": 1.0, "time_nodes": 2, "filters": [["bessellb", "bessellv"], "ztfr"]}
    result = handle_withTime(values)
    assert len(result) == 4
    assert "em_syserr_bessellb___bessellv_0" in result[0]
    assert "em_syserr_ztfr_1" in result[3]
    single = handle_withTime({"type": "Uniform", "minimum": 0.0, "maximum": 1.0, "time_nodes": 2, "filters": ["ztfr"]})
    assert len(single) == 2 and all("em_syserr_ztfr" in line for line in single)
    everything = handle_withTime({"type": "Uniform", "minimum": 0.0, "maximum": 1.0, "time_nodes": 1, "filters": [None]})
    assert len(everything) == 1 and "em_syserr_all_0" in everything[0]


def test_handle_withoutTime():
    result = handle_withoutTime({"type": "Uniform", "minimum": 0.0, "maximum": 1.0})
    assert len(result) == 1
    assert UNIFORM in result[0]
    with pytest.warns(UserWarning, match="are not used by Uniform distribution"):
        assert handle_withoutTime({"type": "Uniform", "minimum": 0.0, "maximum": 1.0, "sigma": 3.0}) == [UNIFORM]
    assert handle_withoutTime({"type": "Gaussian", "mu": 0.5, "sigma": 0.25}) == [
        "em_syserr = Gaussian(mu=0.5, sigma=0.25, name='em_syserr', latex_label='em_syserr', unit=None, boundary=None)"]


def test_main(sample_yaml_file, tmp_path):
    result = main(sample_yaml_file)
    assert len(result) == 4 and all("em_syserr" in line for line in result)
    flat = tmp_path / "withoutTime_config.yaml"
    flat.write_text("config:\n  withTime:\n    value: false\n  withoutTime:\n    value: true\n    type: Uniform\n    minimum: 0.0\n    maximum: 1.0\n")
    result = main(flat)
    assert len(result) == 1 and "em_syserr = Uniform" in result[0]
    empty = tmp_path / "empty_config.yaml"
    empty.write_text("config:\n  withTime:\n    value: false\n  withoutTime:\n    value: false\n")
    with pytest.raises(ValidationError, match="At least one configuration key must be set to True"):
        main(empty)


def test_files_that_do_not_parse(tmp_path):
    bad = tmp_path / "invalid_config.yaml"
    bad.write_text("invalid: yaml: content")
    with pytest.raises(YAMLError):
        main(bad)
    worse = tmp_path / "invalid_format.yaml"
    worse.write_text("{ invalid: yaml: content")
    with pytest.raises(YAMLError):
        main(worse)
    with pytest.raises(FileNotFoundError):
        main("non_existent_file.yaml")


@pytest.mark.parametrize("filter_name", ALLOWED_FILTERS)
def test_all_allowed_filters(filter_name):
    result = handle_withTime({"type": "Uniform", "minimum": 0.0, "maximum": 1.0, "time_nodes": 1, "filters": [filter_name]})
    assert len(result) == 1
    assert f"em_syserr_{filter_name}_0" in result[0]


def test_environment_variables_are_expanded(tmp_path, monkeypatch):
    monkeypatch.setenv("NMMA_TEST_MAXIMUM", "2.5")
    path = tmp_path / "env.yaml"
    path.write_text("config:\n  withoutTime:\n    value: true\n    type: Uniform\n    minimum: 0.0\n    maximum: ${NMMA_TEST_MAXIMUM}\n")
    assert "maximum=2.5" in main(path)[0]


@pytest.mark.skipif(not os.path.isdir("/root/reference/nmma"), reason="the reference tree is not here")
def test_side_by_side_with_the_reference(sample_yaml_file):
    """The reference's own module (imported under oracle/ref_harness.py; bilby's prior classes are stand-ins there, so the prior
    STRINGS are compared through their names only) accepts and rejects the same documents with the same messages."""
    from oracle import ref_harness
    ref = ref_harness.reference_modules().systematics
    assert ref.ALLOWED_FILTERS == ALLOWED_FILTERS
    docs = [{"config": {"withTime": {"value": True}, "withoutTime": {"value": True}}},
            {"config": {"withTime": {"value": False}, "withoutTime": {"value": False}}},
            {"config": {"withTime": {}, "withoutTime": {"value": False}}},
            {"config": {"withTime": {"value": True}, "withoutTime": {"value": False}}}]
    for doc in docs:
        outcome = []
        for fn in (ref.validate_only_one_true, validate_only_one_true):
            try:
                fn(doc)
                outcome.append(None)
            except ValueError as exc:
                outcome.append(str(exc))
        assert outcome[0] == outcome[1], doc
    groups = [[["bessellb", "bessellv"], "ztfr"], [["bessellb", "invalid_filter"], "ztfr"], [["bessellb", "bessellb"], "ztfr"],
              [["bessellb", "bessellv"], "bessellb"], [["bessellb", "bessellv"], None], [None, None], [], ["ztfr", "ztfr"], [[None]],
              ["ztfr", ["ztfg", "ztfr"]], ["nope"]]
    for g in groups:
        outcome = []
        for fn in (ref.validate_filters, validate_filters):
            try:
                fn(g)
                outcome.append(None)
            except ValueError as exc:
                outcome.append(str(exc))
        assert outcome[0] == outcome[1], g
