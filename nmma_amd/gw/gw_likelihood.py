"""The data-parallel part of the gravitational-wave term (``nmma/gw/gw_likelihood.py:97-247`` ->
``bilby.gw.likelihood.GravitationalWaveTransient``): noise-weighted inner products on the GPU for strain that
has already been projected onto each detector.

Scope: the reference's class holds no arithmetic (bilby does); its waveform generator (lalsimulation) and
detector response stay third-party and on the caller's side.  What this module offers is the reduction that
follows them -- for a whole batch of parameter vectors in one launch -- and its constant noise term, shaped so
that the result plugs into :class:`nmma_amd.joint.MultiMessengerLikelihood` as the GW messenger of BASELINE
config 5::

    gw = GWStrainLikelihood(data, psd, frequency_array, duration, minimum_frequency=20.0)
    joint = MultiMessengerLikelihood([em_likelihood, ExternalLogLikelihood("gw")], priors)
    logl = joint.log_likelihood_batch(theta, names, external_logl={"gw": gw.log_likelihood_batch(strain)})

There is no CPU fallback: a missing library or device raises :class:`nmma_amd._lib.NMMAHipError`.
Parity against bilby is unpinned (absent from the build image); the oracle restates its published formulas.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib as L


class GWStrainLikelihood:
    """``data[n_ifo][NF]`` complex frequency-domain strain of every detector, ``psd[n_ifo][NF]`` its power spectral
    density on the same ``frequency_array[NF]``, ``duration`` of the analysed segment in seconds; the inner products run
    over ``minimum_frequency <= f <= maximum_frequency`` (per detector if sequences are given), as bilby's
    ``Interferometer.frequency_mask`` does."""

    def __init__(self, data, psd, frequency_array, duration, minimum_frequency=20.0, maximum_frequency=np.inf, device=0):
        import torch
        self._lib = L.load_library()
        self.device = int(device)
        data = np.ascontiguousarray(np.atleast_2d(np.asarray(data, dtype=np.complex128)))
        psd = np.ascontiguousarray(np.atleast_2d(np.asarray(psd, dtype=np.float64)))
        freq = np.asarray(frequency_array, dtype=np.float64)
        if data.shape != psd.shape or data.shape[1] != freq.shape[0]:
            raise L.NMMAHipError(f"data {data.shape}, psd {psd.shape} and frequency_array {freq.shape} do not match")
        self.n_ifo, self.n_freq = data.shape
        self.duration = float(duration)
        if not self.duration > 0:
            raise L.NMMAHipError("duration must be positive")
        fmin = np.broadcast_to(np.asarray(minimum_frequency, dtype=float), (self.n_ifo,))
        fmax = np.broadcast_to(np.asarray(maximum_frequency, dtype=float), (self.n_ifo,))
        self.mask = (freq[None, :] >= fmin[:, None]) & (freq[None, :] <= fmax[:, None])
        if np.any(~(psd[self.mask] > 0)):
            raise L.NMMAHipError("the PSD must be positive inside the frequency mask")
        weight = np.zeros_like(psd)
        weight[self.mask] = 1.0 / psd[self.mask]
        dev = torch.device(f"cuda:{self.device}")
        self._data = torch.view_as_real(torch.as_tensor(data)).contiguous().to(dev)
        self._weight = torch.as_tensor(weight).to(dev)
        # -<d|d>/2 per detector (bilby: noise_log_likelihood), a constant of the data
        self._noise = float(-0.5 * 4.0 / self.duration * np.sum((data.real ** 2 + data.imag ** 2) * weight))

    def noise_log_likelihood(self):
        return self._noise

    def log_likelihood_ratio_batch(self, strain, out=None, stream=None):
        """``strain[B][n_ifo][NF]`` complex128 (torch CUDA tensor, or numpy -> copied) -> ``logL ratio[B]`` (torch CUDA
        tensor; asynchronous on ``stream``, default torch's current stream of the device)."""
        import torch
        dev = torch.device(f"cuda:{self.device}")
        if not isinstance(strain, torch.Tensor):
            strain = np.asarray(strain)
            if strain.dtype != np.complex128:
                raise L.NMMAHipError(f"strain must be complex128, got {strain.dtype}")
            strain = torch.as_tensor(np.ascontiguousarray(strain))
        if strain.dtype != torch.complex128:
            raise L.NMMAHipError(f"strain must be complex128, got {strain.dtype}")
        if strain.dim() != 3 or tuple(strain.shape[1:]) != (self.n_ifo, self.n_freq):
            raise L.NMMAHipError(f"strain must be [B, {self.n_ifo}, {self.n_freq}], got {tuple(strain.shape)}")
        strain = strain.to(dev).contiguous()
        n = strain.shape[0]
        if out is None:
            out = torch.empty(n, dtype=torch.float64, device=dev)
        elif (out.dtype != torch.float64 or out.device != dev or out.numel() < n or not out.is_contiguous()):
            raise L.NMMAHipError(f"out must be a contiguous float64 tensor on {dev} with >= {n} elements")
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        real = torch.view_as_real(strain)
        L.check(self._lib.nmma_gw_loglike_ratio(C.c_void_p(real.data_ptr()), C.c_void_p(self._data.data_ptr()),
                                                C.c_void_p(self._weight.data_ptr()), n, self.n_ifo, self.n_freq,
                                                self.duration, C.c_void_p(out.data_ptr()), self.device,
                                                C.c_void_p(s.cuda_stream)), "nmma_gw_loglike_ratio")
        return out[:n]

    def log_likelihood_batch(self, strain, out=None, stream=None):
        """log L = log L ratio + noise log-likelihood (bilby: ``GravitationalWaveTransient.log_likelihood``)."""
        return self.log_likelihood_ratio_batch(strain, out=out, stream=stream) + self._noise
