"""GPU parity of the GW inner-product kernel (nmma_gw_loglike_ratio, through the C ABI) against the oracle's restatement of
bilby's formulas, and config 5's assembly: EM logL on the GPU + GW logL from the kernel, summed by MultiMessengerLikelihood."""
import numpy as np
import pytest

from tests.cases_gw import make_gw_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.mark.parametrize("shape", [dict(), dict(n_ifo=1, batch=5, sampling_frequency=333.0), dict(n_ifo=2, batch=300, duration=2.0)])
def test_inner_products_match_oracle(shape, torch_cuda):
    torch = torch_cuda
    from nmma_amd.gw.gw_likelihood import GWStrainLikelihood
    from oracle import gw_oracle as gwo
    c = make_gw_case(**shape)
    gw = GWStrainLikelihood(c["data"], c["psd"], c["frequency_array"], c["duration"],
                            minimum_frequency=c["minimum_frequency"], maximum_frequency=c["maximum_frequency"])
    np.testing.assert_array_equal(gw.mask, c["mask"])
    want = gwo.log_likelihood_ratio_batch(c["strain"], c["data"], c["psd"], c["mask"], c["duration"])
    got = gw.log_likelihood_ratio_batch(c["strain"]).cpu().numpy()
    # fp64 sums of 1e3-1e5 terms in a different (fixed) order than numpy's pairwise sums, and Re<d|h> - <h|h>/2 formed per bin
    # instead of from two separately rounded totals: 1e-10 relative (measured <= 3e-12)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10 * np.abs(want).max())
    # device-resident input, deterministic, batch-size independent
    s_dev = torch.as_tensor(c["strain"], device="cuda:0")
    again = gw.log_likelihood_ratio_batch(s_dev).cpu().numpy()
    assert np.array_equal(again, got)
    assert np.array_equal(gw.log_likelihood_ratio_batch(s_dev[:3]).cpu().numpy(), got[:3])
    assert gw.noise_log_likelihood() == pytest.approx(gwo.noise_log_likelihood(c["data"], c["psd"], c["mask"], c["duration"]), rel=1e-13)
    full = gw.log_likelihood_batch(s_dev).cpu().numpy()
    np.testing.assert_allclose(full, want + gw.noise_log_likelihood(), rtol=1e-10)


def test_samples_per_workgroup_variants_give_the_same_bits(torch_cuda):
    """The launcher groups 1, 2, 4 or 8 parameter vectors per workgroup depending on the batch size (the shared data / weight
    arrays are read once per group); a sample's summation order does not depend on the grouping: bit-identical rows for
    every batch size, including sizes that are not a multiple of the group."""
    torch = torch_cuda
    from nmma_amd.gw.gw_likelihood import GWStrainLikelihood
    from oracle import gw_oracle as gwo
    c = make_gw_case(n_ifo=2, batch=2500, duration=1.0, sampling_frequency=512.0)
    gw = GWStrainLikelihood(c["data"], c["psd"], c["frequency_array"], c["duration"], minimum_frequency=c["minimum_frequency"])
    s_dev = torch.as_tensor(c["strain"], device="cuda:0")
    full = gw.log_likelihood_ratio_batch(s_dev).cpu().numpy()                       # 2500 rows: 8 per workgroup, ragged tail
    want = gwo.log_likelihood_ratio_batch(c["strain"][:64], c["data"], c["psd"], c["mask"], c["duration"])
    np.testing.assert_allclose(full[:64], want, rtol=1e-10, atol=1e-10 * np.abs(want).max())
    for nrows in (1, 7, 300, 513, 1027, 2049):                                      # 1 / 1 / 1 / 2 / 4 / 8 per workgroup
        part = gw.log_likelihood_ratio_batch(s_dev[:nrows]).cpu().numpy()
        assert np.array_equal(part, full[:nrows]), nrows
        tail = gw.log_likelihood_ratio_batch(s_dev[2500 - nrows:]).cpu().numpy()
        assert np.array_equal(tail, full[2500 - nrows:]), nrows


def test_bad_arguments_are_refused(torch_cuda):
    from nmma_amd import _lib as L
    from nmma_amd.gw.gw_likelihood import GWStrainLikelihood
    c = make_gw_case(batch=2)
    gw = GWStrainLikelihood(c["data"], c["psd"], c["frequency_array"], c["duration"])
    with pytest.raises(L.NMMAHipError):
        gw.log_likelihood_ratio_batch(c["strain"][:, :2])               # wrong number of detectors
    with pytest.raises(L.NMMAHipError):
        gw.log_likelihood_ratio_batch(c["strain"].astype(np.complex64))  # silently reduced precision is not accepted
    bad = c["psd"].copy()
    bad[0, 200] = 0.0
    with pytest.raises(L.NMMAHipError):
        GWStrainLikelihood(c["data"], bad, c["frequency_array"], c["duration"])


def test_config5_gw_plus_em(torch_cuda):
    """BASELINE config 5's assembly on one GPU: joint logL = EM (HIP path) + GW (inner products of supplied strain)."""
    torch = torch_cuda
    from nmma_amd.gw.gw_likelihood import GWStrainLikelihood
    from nmma_amd.joint.joint_likelihood import ExternalLogLikelihood, MultiMessengerLikelihood
    from oracle import gw_oracle as gwo
    from oracle import nmma_oracle as orc
    from tests import cases
    from tests.helpers import oracle_from_case, plugin_from_case, rel_err
    case = cases.case_c2_default()
    n = len(case["theta"])
    c = make_gw_case(batch=n)
    _, _, em = plugin_from_case(case)
    gw = GWStrainLikelihood(c["data"], c["psd"], c["frequency_array"], c["duration"],
                            minimum_frequency=c["minimum_frequency"], maximum_frequency=c["maximum_frequency"])
    joint = MultiMessengerLikelihood([em, ExternalLogLikelihood("gw", noise_log_likelihood=gw.noise_log_likelihood())], em.priors)
    gw_logl = gw.log_likelihood_batch(torch.as_tensor(c["strain"], device="cuda:0"))
    got = joint.log_likelihood_batch(torch.as_tensor(case["theta"], device="cuda:0"), case["names"],
                                     external_logl={"gw": gw_logl}).cpu().numpy()
    em_want = orc.log_likelihood_batch(oracle_from_case(case, use_scipy=False), case["names"], case["theta"])
    gw_want = gwo.log_likelihood_ratio_batch(c["strain"], c["data"], c["psd"], c["mask"], c["duration"]) \
        + gwo.noise_log_likelihood(c["data"], c["psd"], c["mask"], c["duration"])
    floor = em_want == orc.LOGL_FLOOR
    want = np.where(floor, orc.LOGL_FLOOR, em_want + gw_want)
    assert np.array_equal(got == orc.LOGL_FLOOR, floor)
    assert rel_err(got[~floor], want[~floor]).max() <= 1e-6
