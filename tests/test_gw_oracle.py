"""CPU checks of the GW inner-product oracle (oracle/gw_oracle.py, parity unpinned against bilby): the identities the
published formulas imply."""
import numpy as np

from oracle import gw_oracle as gwo
from tests.cases_gw import make_gw_case


def test_inner_product_identities():
    c = make_gw_case(batch=6)
    d, s, m, T = c["data"], c["psd"], c["mask"], c["duration"]
    h = c["strain"]
    # <h|h> is real, positive and quadratic in the amplitude; <d|h> is linear
    for k in range(3):
        hh = gwo.noise_weighted_inner_product(h[1, k][m[k]], h[1, k][m[k]], s[k][m[k]], T)
        assert abs(hh.imag) <= 1e-12 * hh.real and hh.real > 0
        hh2 = gwo.noise_weighted_inner_product(2 * h[1, k][m[k]], 2 * h[1, k][m[k]], s[k][m[k]], T)
        np.testing.assert_allclose(hh2.real, 4 * hh.real, rtol=1e-14)
    # log L ratio of h against data = n + h0:  ratio(h) = Re<d|h> - <h|h>/2, maximal near the injected signal
    ratios = gwo.log_likelihood_ratio_batch(h, d, s, m, T)
    assert np.argmax(ratios) == 0 and ratios[0] > 0
    # zero strain: ratio 0; the full likelihood then equals the noise likelihood
    zero = np.zeros_like(h[0])
    assert gwo.log_likelihood_ratio(zero, d, s, m, T) == 0.0
    assert gwo.noise_log_likelihood(d, s, m, T) < 0
    # Re<d|h> - <h|h>/2 = (<d|d> - <d-h|d-h>)/2: the Gaussian likelihood of the residual
    res = d - h[2]
    lhs = gwo.log_likelihood_ratio(h[2], d, s, m, T)
    rhs = sum(0.5 * (gwo.noise_weighted_inner_product(d[i][m[i]], d[i][m[i]], s[i][m[i]], T).real
                     - gwo.noise_weighted_inner_product(res[i][m[i]], res[i][m[i]], s[i][m[i]], T).real) for i in range(3))
    np.testing.assert_allclose(lhs, rhs, rtol=1e-10)


def test_mask_excludes_out_of_band_bins():
    c = make_gw_case(batch=3)
    base = gwo.log_likelihood_ratio_batch(c["strain"], c["data"], c["psd"], c["mask"], c["duration"])
    h2 = c["strain"].copy()
    h2[:, 2, ~c["mask"][2]] *= 1e6          # bins outside detector 2's own band do not enter its inner products
    np.testing.assert_array_equal(gwo.log_likelihood_ratio_batch(h2, c["data"], c["psd"], c["mask"], c["duration"]), base)
