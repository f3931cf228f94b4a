"""CPU restatement of the gravitational-wave term of the joint likelihood.

TEST INFRASTRUCTURE ONLY: imported by ``tests/`` (and nothing in ``nmma_amd/``).

PARITY UNPINNED.  ``nmma/gw/gw_likelihood.py:97-247`` holds no arithmetic of its own: it wraps
``bilby.gw.likelihood.GravitationalWaveTransient`` (bilby >= 2.7.1, pyproject.toml), which is absent from
the build image, and the reference ships no golden vectors for it (the GW arguments of
``tests/joint_analysis_pipeline.py:75-82`` are commented out).  What follows restates bilby's published
algorithm for the un-marginalised likelihood:

* ``bilby/gw/utils.py: noise_weighted_inner_product(aa, bb, power_spectral_density, duration)``
  ``= 4 / duration * sum(conj(aa) * bb / power_spectral_density)``
* ``bilby/gw/detector/interferometer.py: inner_product / optimal_snr_squared`` apply it on
  ``frequency_mask`` (minimum_frequency <= f <= maximum_frequency) with the detector's PSD array
* ``bilby/gw/likelihood/base.py: calculate_snrs, log_likelihood_ratio``:
  ``sum_ifo ( Re <d|h> - <h|h> / 2 )``; ``noise_log_likelihood``: ``sum_ifo -<d|d> / 2``;
  ``log_likelihood = log_likelihood_ratio + noise_log_likelihood``

The strain per detector (antenna response, time and phase shifts: ``get_detector_response``) and the waveform
itself (lalsimulation) are inputs here, as they are for the HIP kernel.
"""
import numpy as np


def noise_weighted_inner_product(aa, bb, power_spectral_density, duration):
    integrand = np.conj(aa) * bb / power_spectral_density
    return 4 / duration * np.sum(integrand)


def frequency_mask(frequency_array, minimum_frequency, maximum_frequency):
    return (frequency_array >= minimum_frequency) & (frequency_array <= maximum_frequency)


def log_likelihood_ratio(strain, data, psd, mask, duration):
    """``strain[n_ifo][NF]`` (one parameter vector), ``data[n_ifo][NF]``, ``psd[n_ifo][NF]``, ``mask[n_ifo][NF]``."""
    total = 0.0
    for h, d, s, m in zip(strain, data, psd, mask):
        d_inner_h = noise_weighted_inner_product(d[m], h[m], s[m], duration)
        optimal_snr_squared = noise_weighted_inner_product(h[m], h[m], s[m], duration).real
        total += d_inner_h.real - optimal_snr_squared / 2
    return float(total)


def noise_log_likelihood(data, psd, mask, duration):
    total = 0.0
    for d, s, m in zip(data, psd, mask):
        total -= noise_weighted_inner_product(d[m], d[m], s[m], duration).real / 2
    return float(total)


def log_likelihood_ratio_batch(strain, data, psd, mask, duration):
    return np.array([log_likelihood_ratio(h, data, psd, mask, duration) for h in strain])
