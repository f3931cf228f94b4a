"""Shared set-up of the gravitational-wave tests: synthetic detector data (noise + an injection made by the ORACLE) as
``nmma_amd.gw.Interferometer`` objects, and the oracle-side view of the same data."""
import numpy as np

from nmma_amd import synthetic as syn


def make_case(seed=11, duration=4.0, sampling_frequency=2048.0, ifo_names=("H1", "L1", "V1"), minimum_frequency=20.0,
              f_ref=20.0, injection=None, approximant="IMRPhenomD_NRTidalv2", post_trigger=2.0, snr_scale=1.0):
    """A data segment ending ``post_trigger`` seconds after the injected coalescence."""
    from oracle import gw_waveform_oracle as gwo
    from nmma_amd.gw import Interferometer
    inj = dict(injection or syn.GW170817_LIKE)
    inj["luminosity_distance"] = inj["luminosity_distance"] / snr_scale
    start = inj["geocent_time"] + post_trigger - duration
    freq, noise = syn.make_gw_noise(seed, duration, sampling_frequency, ifo_names)
    tidal = approximant == "IMRPhenomD_NRTidalv2"
    ifos, oracle_ifos = [], []
    for name in ifo_names:
        n, psd = noise[name]
        h = gwo.detector_strain(inj, name, freq, start, f_ref, minimum_frequency, tidal=tidal)
        ifo = Interferometer(name, n + h, psd, duration, start, minimum_frequency=minimum_frequency,
                             sampling_frequency=sampling_frequency)
        ifos.append(ifo)
        oracle_ifos.append(dict(name=name, frequency_array=freq, data=ifo.frequency_domain_strain, psd=psd,
                                mask=ifo.frequency_mask, start_time=start, duration=duration))
    wa = dict(waveform_approximant=approximant, reference_frequency=f_ref, minimum_frequency=minimum_frequency)
    return dict(ifos=ifos, oracle_ifos=oracle_ifos, injection=inj, waveform_arguments=wa, f_ref=f_ref,
                f_min=minimum_frequency, tidal=tidal, start_time=start, duration=duration, frequency_array=freq)


def oracle_loglike_ratio(case, names, theta, fixed=None, phase_marginalization=False, distance_marginalization=None,
                         time_marginalization=None):
    from oracle import gw_waveform_oracle as gwo
    out = np.empty(len(theta))
    for i, row in enumerate(theta):
        p = dict(zip(names, (float(v) for v in row)), **(fixed or {}))
        if "cos_theta_jn" in p and "theta_jn" not in p:
            p["theta_jn"] = float(np.arccos(p["cos_theta_jn"]))
        if phase_marginalization:
            p["phase"] = 0.0
        out[i] = gwo.log_likelihood_ratio(p, case["oracle_ifos"], case["f_ref"], case["f_min"],
                                          phase_marginalization=phase_marginalization, tidal=case["tidal"],
                                          distance_marginalization=distance_marginalization,
                                          time_marginalization=time_marginalization)
    return out
