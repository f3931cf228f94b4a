"""Re-run the reference's own source (when /root/reference is present, i.e. in the
build container) and check it still reproduces the committed golden vectors."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/nmma"),
                                reason="reference tree not available (GPU box)")


@pytest.mark.parametrize("name", ["c2_default", "edges", "syserr_time_nodes", "averaging"])
def test_reference_reproduces_golden(name):
    from tools.make_golden import build_reference_likelihood
    case = cases.CASES[name]()
    gold = cases.load_golden(name)
    _, lik, _ = build_reference_likelihood(case)
    n = min(12, len(case["theta"]))
    got = np.array([lik.log_likelihood(dict(zip(case["names"], (float(v) for v in row))))
                    for row in case["theta"][:n]])
    np.testing.assert_allclose(got, gold["logl"][:n], rtol=1e-13)
