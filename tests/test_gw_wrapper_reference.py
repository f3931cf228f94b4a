"""Control flow of the GW messenger class against the REFERENCE'S OWN SOURCE (nmma/gw/gw_likelihood.py:97-247), imported under
the stub harness with a recording stand-in for ``bilby.gw.likelihood.GravitationalWaveTransient`` (bilby's arithmetic is
absent from the image; what can be pinned is everything the reference's wrapper itself does): constructor signature, the
side effects on the waveform generator (:167-168), which kwargs reach the bilby class (:171-183), the choice of the
source-frame conversion (:207-210), the error on an unknown likelihood type (:205) and ``posterior_conversion`` (:212-236).
CPU only; skipped where the reference tree is absent (the GPU box)."""
import importlib
import inspect
import os
import sys
import types

import numpy as np
import pytest

from oracle import ref_harness

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(ref_harness.REFERENCE_ROOT, "nmma")),
                                reason="reference tree not present")


class _Recorder:
    """Stands in for every bilby GW likelihood class: keeps what it was given."""
    calls = []

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.interferometers = kwargs["interferometers"]
        self.waveform_generator = kwargs["waveform_generator"]
        type(self).calls.append((type(self).__name__, kwargs))

    def noise_log_likelihood(self):
        return -123.5


@pytest.fixture(scope="module")
def ref_gw():
    ref_harness.reference_modules()
    lk = importlib.import_module("bilby.gw.likelihood")
    for name in ("GravitationalWaveTransient", "ROQGravitationalWaveTransient", "RelativeBinningGravitationalWaveTransient",
                 "MBGravitationalWaveTransient"):
        setattr(lk, name, type(name, (_Recorder,), {}))
    if "nmma.gw" not in sys.modules:
        m = types.ModuleType("nmma.gw")
        m.__path__ = [os.path.join(ref_harness.REFERENCE_ROOT, "nmma", "gw")]
        sys.modules["nmma.gw"] = m
    return importlib.import_module("nmma.gw.gw_likelihood")


class _Ifo:
    name = "H1"
    time_array = np.array([1000.0, 1000.5])


def _generator(model_name):
    def model(*a, **k):
        return None
    model.__name__ = model_name
    return types.SimpleNamespace(frequency_domain_source_model=model, waveform_arguments={}, parameter_conversion=None, start_time=0.0)


def test_constructor_signature_matches(ref_gw):
    from nmma_amd.gw import GravitationalWaveTransientLikelihood as Mine
    ref_sig = inspect.signature(ref_gw.GravitationalWaveTransientLikelihood.__init__).parameters
    my_sig = inspect.signature(Mine.__init__).parameters
    ref_names = [n for n in ref_sig]
    mine = [n for n in my_sig if n != "device"]
    assert mine == ref_names
    for n in ref_names:
        if ref_sig[n].default is not inspect.Parameter.empty:
            assert my_sig[n].default == ref_sig[n].default, n


@pytest.mark.parametrize("model_name,conv", [("lal_binary_neutron_star", "bns_source_frame"), ("lal_binary_black_hole", "bbh_source_frame")])
def test_reference_wrapper_control_flow(ref_gw, model_name, conv):
    """What the reference's constructor does around the bilby class -- the behaviour nmma_amd.gw mirrors."""
    _Recorder.calls.clear()
    wg = _generator(model_name)
    priors = ref_harness._PriorDict(chirp_mass=ref_harness._Uniform("chirp_mass", 1.0, 2.0))
    lik = ref_gw.GravitationalWaveTransientLikelihood(priors, [_Ifo()], wg, phase_marginalization=True)
    kind, kwargs = _Recorder.calls[-1]
    assert kind == "GravitationalWaveTransient"
    assert kwargs["phase_marginalization"] is True and kwargs["time_marginalization"] is False
    assert kwargs["distance_marginalization"] is False and kwargs["jitter_time"] is True
    assert kwargs["reference_frame"] == "sky" and kwargs["time_reference"] == "geocenter" and kwargs["priors"] is priors
    assert wg.start_time == 1000.0 and wg.parameter_conversion({"x": 1}) == ({"x": 1}, [])
    assert lik.parameter_conversion.__name__ == conv
    assert lik.noise_log_likelihood() == -123.5 and lik.sanity_checks() is True
    with pytest.raises(ValueError):
        ref_gw.GravitationalWaveTransientLikelihood(priors, [_Ifo()], _generator(model_name), gw_likelihood_type="Nonsense")
    # the mirror makes the same choices (no GPU is touched by construction)
    from nmma_amd.gw import GravitationalWaveTransientLikelihood as Mine
    from nmma_amd.gw.detector import Interferometer
    n = 9
    ifo = Interferometer("H1", np.ones(n, complex), np.ones(n), duration=1.0, start_time=1000.0, sampling_frequency=16.0)
    wg2 = _generator(model_name)
    mine = Mine({"chirp_mass": ref_harness._Uniform("chirp_mass", 1.0, 2.0)}, [ifo], wg2, phase_marginalization=True)
    assert wg2.start_time == 1000.0 and wg2.parameter_conversion({"x": 1}) == ({"x": 1}, [])
    assert mine.parameter_conversion.__name__ == conv
    assert mine.sub_model.phase_marginalization is True and mine.sanity_checks() is True
    with pytest.raises(ValueError):
        Mine({}, [ifo], _generator(model_name), gw_likelihood_type="Nonsense")
    # noise term: -<d|d>/2 over the mask [20 Hz, 8 Hz Nyquist] is empty here -> 0; with a lower cut it is the closed form
    ifo2 = Interferometer("H1", np.full(n, 1 + 1j), np.full(n, 2.0), duration=1.0, start_time=1000.0, sampling_frequency=16.0,
                          minimum_frequency=2.0)
    mine2 = Mine({}, [ifo2], _generator(model_name))
    assert mine2.noise_log_likelihood() == pytest.approx(-0.5 * 4.0 * 7 * 2.0 / 2.0)


def test_posterior_conversion_matches_reference(ref_gw):
    from nmma_amd.gw import GravitationalWaveTransientLikelihood as Mine
    from nmma_amd.gw.detector import Interferometer
    rng = np.random.default_rng(3)
    post = dict(mass_ratio=rng.uniform(0.5, 1, 20), chi_1=rng.uniform(-0.05, 0.05, 20), chi_2=rng.uniform(-0.05, 0.05, 20),
                lambda_1=rng.uniform(0, 1000, 20), lambda_2=rng.uniform(0, 2000, 20))
    ref = ref_gw.GravitationalWaveTransientLikelihood(ref_harness._PriorDict(), [_Ifo()], _generator("lal_binary_neutron_star"))
    want = ref.posterior_conversion({k: v.copy() for k, v in post.items()})
    ifo = Interferometer("H1", np.ones(9, complex), np.ones(9), duration=1.0, start_time=0.0, sampling_frequency=16.0)
    got = Mine({}, [ifo], _generator("lal_binary_neutron_star")).posterior_conversion({k: v.copy() for k, v in post.items()})
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), k
    # spin_1z / spin_2z fallbacks and missing ingredients
    alt = dict(mass_ratio=post["mass_ratio"], spin_1z=post["chi_1"], spin_2z=post["chi_2"])
    assert "chi_eff" not in Mine({}, [ifo], _generator("x")).posterior_conversion(dict(mass_ratio=post["mass_ratio"]))
    w2 = ref.posterior_conversion(dict(alt)); g2 = Mine({}, [ifo], _generator("x")).posterior_conversion(dict(alt))
    assert ("chi_eff" in w2) == ("chi_eff" in g2)
