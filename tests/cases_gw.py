"""Seeded synthetic GW data for the inner-product tests: three detectors, coloured-noise data, and per-sample detector
strain of a chirp-like signal (amplitude ~ f^(-7/6), quadratic-plus-power-law phase) -- shapes and magnitudes of a
GW170817-like analysis, no waveform model involved (that part is third-party)."""
import numpy as np


def make_gw_case(seed=1717, n_ifo=3, duration=4.0, sampling_frequency=1024.0, batch=24, fmin=(20.0, 20.0, 30.0), fmax=400.0):
    rng = np.random.default_rng(seed)
    nf = int(duration * sampling_frequency / 2) + 1
    freq = np.arange(nf) / duration
    f_safe = np.maximum(freq, 1.0)
    # aLIGO-like PSD shape per detector, ~1e-46 at the bucket
    psd = np.stack([(1e-46 * (1 + 0.2 * k)) * ((f_safe / 150.0) ** -4.1 + 1.0 + (f_safe / 300.0) ** 2.0) for k in range(n_ifo)])
    sigma = np.sqrt(psd * duration / 4.0)
    data = (rng.normal(size=(n_ifo, nf)) + 1j * rng.normal(size=(n_ifo, nf))) * sigma
    mchirp = rng.uniform(1.15, 1.25, batch)
    dist = rng.uniform(20.0, 80.0, batch)
    tc = rng.uniform(-0.01, 0.01, batch)
    phic = rng.uniform(0, 2 * np.pi, batch)
    resp = rng.uniform(-1.0, 1.0, (batch, n_ifo)) + 1j * rng.uniform(-1.0, 1.0, (batch, n_ifo))
    delay = rng.uniform(-0.02, 0.02, (batch, n_ifo))
    amp = 2.5e-21 * (mchirp[:, None] / 1.2) ** (5.0 / 6.0) * (40.0 / dist[:, None]) * (f_safe[None, :] / 100.0) ** (-7.0 / 6.0)
    psi = (2 * np.pi * freq[None, :] * tc[:, None] - phic[:, None]
           + (3.0 / 128.0) * (np.pi * 4.925e-6 * mchirp[:, None] * f_safe[None, :]) ** (-5.0 / 3.0))
    hplus = amp * np.exp(1j * psi)
    strain = resp[:, :, None] * hplus[:, None, :] * np.exp(-2j * np.pi * freq[None, None, :] * delay[:, :, None])
    strain[:, :, 0] = 0.0
    data = data + strain[0]                                     # the first parameter vector is the injected signal
    mask = np.stack([(freq >= fmin[k % len(fmin)]) & (freq <= fmax) for k in range(n_ifo)])
    return dict(frequency_array=freq, duration=duration, data=data, psd=psd, mask=mask, strain=strain,
                minimum_frequency=[fmin[k % len(fmin)] for k in range(n_ifo)], maximum_frequency=fmax)
