"""Re-run the reference's own source (when /root/reference is present, i.e. in the
build container) and check it still reproduces EVERY committed golden vector: all SVD cases
(every row), the Me2017 model (config 1) and the combined model (config 3 shape)."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/nmma"),
                                reason="reference tree not available (GPU box)")


def _rows(lik, names, theta, fixed=None):
    return np.array([lik.log_likelihood(dict(zip(names, (float(v) for v in row)), **(fixed or {}))) for row in theta])


@pytest.mark.parametrize("name", list(cases.CASES))
def test_reference_reproduces_golden(name):
    from tools.make_golden import build_reference_likelihood
    case = cases.CASES[name]()
    gold = cases.load_golden(name)
    _, lik, _ = build_reference_likelihood(case)
    got = _rows(lik, case["names"], case["theta"], case.get("fixed"))
    assert len(got) == len(gold["logl"])
    np.testing.assert_allclose(got, gold["logl"], rtol=1e-13)


def test_reference_reproduces_me2017_golden():
    from tests import cases_me2017
    from tools.make_golden_me2017 import build_reference
    case = cases_me2017.case_me2017()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("me2017")["logl"], rtol=1e-13)


def test_reference_reproduces_combined_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined")["logl"], rtol=1e-13)


@pytest.mark.parametrize("name", ["combined_syserr", "combined_loggrid"])
def test_reference_reproduces_combined_extras_golden(name):
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = getattr(cases_combined, "case_" + name)()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden(name)["logl"], rtol=1e-13)


def test_reference_reproduces_combined_union_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined_union()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined_union")["logl"], rtol=1e-13)


def test_reference_reproduces_combined_owngrids_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined_owngrids()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined_owngrids")["logl"], rtol=1e-13)


def test_reference_reproduces_combined_nullfilters_golden():
    from tests import cases_combined
    from tools.make_golden_combined import build_reference
    case = cases_combined.case_combined_nullfilters()
    lik, _ = build_reference(case)
    got = _rows(lik, case["names"], case["theta"])
    np.testing.assert_allclose(got, cases.load_golden("combined_nullfilters")["logl"], rtol=1e-13)
