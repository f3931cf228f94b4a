"""The gravitational-wave messenger (``nmma/gw/gw_likelihood.py:97-247`` -> ``bilby.gw.likelihood.GravitationalWaveTransient``).

Two entry levels, both batched and both GPU-only (a missing library or device raises :class:`nmma_amd._lib.NMMAHipError`):

* :class:`GravitationalWaveTransientLikelihood` -- the reference's class (same constructor and attributes) over
  :class:`GWEngine`: from PARAMETERS to log-likelihood in one fused kernel (frequency-domain IMRPhenomD_NRTidalv2 / IMRPhenomD
  waveform, antenna response and arrival-time shift per detector, noise-weighted inner products, optional phase
  marginalisation); the strain is never written to memory.  BASELINE config 5::

      gw = GravitationalWaveTransientLikelihood(priors, interferometers, waveform_generator, phase_marginalization=True)
      joint = MultiMessengerLikelihood([em_likelihood, gw], priors)
      logl = joint.log_likelihood_batch(theta, names)

* :class:`GWStrainLikelihood` -- the reduction alone, for strain the caller generated with another waveform model and already
  projected onto the detectors (an HBM-bound stream).

bilby and lalsimulation are absent from the build image: parity of this leg is UNPINNED (``oracle/gw_waveform_oracle.py``
restates the published algorithms; the HIP path is tested against that restatement).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib as L
from ..core.base import NMMALikelihood
from .detector import gmst_linearisation


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
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        if s.device.index != self.device:
            raise L.NMMAHipError(f"stream belongs to cuda:{s.device.index}, the likelihood to cuda:{self.device}")
        # staging (host-to-device copy, .contiguous(), the output allocation) runs on torch's current stream: the launch stream has
        # to wait for it when it is a different one
        strain = strain.to(dev).contiguous()
        n = strain.shape[0]
        if out is None:
            out = torch.empty(n, dtype=torch.float64, device=dev)
        elif (out.dtype != torch.float64 or out.device != dev or out.numel() < n or not out.is_contiguous()):
            raise L.NMMAHipError(f"out must be a contiguous float64 tensor on {dev} with >= {n} elements")
        cur = torch.cuda.current_stream(self.device)
        if s.cuda_stream != cur.cuda_stream:
            s.wait_stream(cur)
        real = torch.view_as_real(strain)
        if n:
            L.check(self._lib.nmma_gw_loglike_ratio(C.c_void_p(real.data_ptr()), C.c_void_p(self._data.data_ptr()),
                                                    C.c_void_p(self._weight.data_ptr()), n, self.n_ifo, self.n_freq,
                                                    self.duration, C.c_void_p(out.data_ptr()), self.device,
                                                    C.c_void_p(s.cuda_stream)), "nmma_gw_loglike_ratio")
        return out[:n]

    def log_likelihood_batch(self, strain, out=None, stream=None):
        """log L = log L ratio + noise log-likelihood (bilby: ``GravitationalWaveTransient.log_likelihood``)."""
        return self.log_likelihood_ratio_batch(strain, out=out, stream=stream) + self._noise


# ---------------------------------------------------------------------------------------------------------------------
# The GW term from PARAMETERS: waveform + detector projection + inner products fused on the device
# ---------------------------------------------------------------------------------------------------------------------
#: parameters of bilby's aligned-spin binary-neutron-star source model (bilby/gw/source.py: lal_binary_neutron_star)
GW_PARAMETERS = ("chi_1", "chi_2", "lambda_1", "lambda_2", "luminosity_distance", "theta_jn", "phase", "ra", "dec", "psi",
                 "geocent_time")
SUPPORTED_APPROXIMANTS = {"IMRPhenomD_NRTidalv2": 1, "IMRPhenomD": 0}


class WaveformGenerator:
    """Stand-in for ``bilby.gw.WaveformGenerator`` when bilby is absent: the attributes the likelihood reads
    (``duration``, ``sampling_frequency``, ``start_time``, ``frequency_domain_source_model.__name__``, ``waveform_arguments``)."""

    def __init__(self, duration, sampling_frequency, start_time=0.0, waveform_arguments=None, frequency_domain_source_model=None,
                 parameter_conversion=None):
        self.duration, self.sampling_frequency, self.start_time = float(duration), float(sampling_frequency), float(start_time)
        self.waveform_arguments = dict(waveform_arguments or {})
        if frequency_domain_source_model is None:
            def lal_binary_neutron_star(*_a, **_k):       # name only: the arithmetic runs on the device
                raise NotImplementedError("evaluated on the GPU by GWEngine")
            frequency_domain_source_model = lal_binary_neutron_star
        self.frequency_domain_source_model = frequency_domain_source_model
        self.parameter_conversion = parameter_conversion


class GWEngine:
    """One GPU-resident GW likelihood (a ``nmma_gw_handle``): replaces, for a whole batch of parameter vectors per call,
    ``waveform_generator.frequency_domain_strain`` -> ``Interferometer.get_detector_response`` -> the inner products of
    ``bilby.gw.likelihood.GravitationalWaveTransient.log_likelihood_ratio`` (as wrapped by nmma/gw/gw_likelihood.py:185-203).

    ``parameter_names`` are the columns of ``theta``; anything of the source model that is not a column must be in ``fixed``.
    Masses: ``chirp_mass`` + ``mass_ratio`` or ``mass_1`` + ``mass_2`` (detector frame, solar masses); spins: aligned
    ``chi_1`` / ``chi_2``; inclination: ``theta_jn`` or ``cos_theta_jn``.  No CPU fallback."""

    def __init__(self, interferometers, parameter_names, fixed=None, waveform_arguments=None, phase_marginalization=False,
                 gmst_reference_time=None, device=0, distance_marginalization=None, time_marginalization=None,
                 time_jitter_bounds=None):
        import torch  # noqa: F401  (device buffers / stream)
        self._handle = None
        self._lib = L.load_library()
        self.device = int(device)
        names, fixed = list(parameter_names), dict(fixed or {})
        self.parameter_names = names
        wa = dict(waveform_arguments or {})
        approximant = wa.get("waveform_approximant", "IMRPhenomD_NRTidalv2")
        if approximant not in SUPPORTED_APPROXIMANTS:
            raise L.NMMAHipError(f"waveform_approximant {approximant!r} is not built on the device path "
                                 f"(available: {sorted(SUPPORTED_APPROXIMANTS)})")
        ifos = list(interferometers)
        if not 1 <= len(ifos) <= L.GW_MAX_IFO:
            raise L.NMMAHipError(f"1..{L.GW_MAX_IFO} interferometers, got {len(ifos)}")
        n_freq = len(ifos[0].frequency_array)
        duration, start = float(ifos[0].strain_data.duration), float(ifos[0].strain_data.start_time)
        for ifo in ifos:
            if len(ifo.frequency_array) != n_freq or float(ifo.strain_data.duration) != duration or \
                    float(ifo.strain_data.start_time) != start:
                raise L.NMMAHipError("all interferometers must share one frequency array and one data segment")
        if abs(ifos[0].frequency_array[1] * duration - 1.0) > 1e-9:
            raise L.NMMAHipError("frequency_array must be k / duration")
        self.n_ifo, self.n_freq, self.duration, self.start_time = len(ifos), n_freq, duration, start
        data = np.ascontiguousarray(np.stack([np.asarray(i.frequency_domain_strain, dtype=np.complex128) for i in ifos]))
        psd = np.ascontiguousarray(np.stack([np.asarray(i.power_spectral_density_array, dtype=np.float64) for i in ifos]))
        mask = np.ascontiguousarray(np.stack([np.asarray(i.frequency_mask, dtype=bool) for i in ifos]).astype(np.uint8))
        tensor = np.ascontiguousarray(np.stack([np.asarray(i.detector_tensor, dtype=np.float64).reshape(9) for i in ifos]))
        vertex = np.ascontiguousarray(np.stack([np.asarray(i.vertex, dtype=np.float64).reshape(3) for i in ifos]))

        def slot(key, default=None, op=L.OP_IDENT):
            if key in names:
                return L.Slot.column(names.index(key), op)
            if key in fixed:
                v = float(fixed[key])
                return L.Slot.constant(float(np.arccos(v)) if op == L.OP_ACOS else v)
            if default is None:
                raise L.NMMAHipError(f"GW parameter {key!r} is neither sampled nor fixed (sampled: {names}, fixed: {sorted(fixed)})")
            return L.Slot.constant(default)

        have = set(names) | set(fixed)
        cfg = L.GwConfig()
        if {"chirp_mass", "mass_ratio"} <= have:
            cfg.mass_mode, cfg.mass_a, cfg.mass_b = L.GW_CHIRP_MASS_RATIO, slot("chirp_mass"), slot("mass_ratio")
        elif {"mass_1", "mass_2"} <= have:
            cfg.mass_mode, cfg.mass_a, cfg.mass_b = L.GW_COMPONENT_MASSES, slot("mass_1"), slot("mass_2")
        else:
            raise L.NMMAHipError("the GW leg needs chirp_mass + mass_ratio or mass_1 + mass_2")
        cfg.theta_jn = slot("cos_theta_jn", op=L.OP_ACOS) if ("cos_theta_jn" in have and "theta_jn" not in have) else slot("theta_jn")
        cfg.chi_1, cfg.chi_2 = slot("chi_1", 0.0), slot("chi_2", 0.0)
        cfg.lambda_1, cfg.lambda_2 = slot("lambda_1", 0.0), slot("lambda_2", 0.0)
        cfg.luminosity_distance, cfg.phase = slot("luminosity_distance"), slot("phase", 0.0)
        cfg.ra, cfg.dec, cfg.psi, cfg.geocent_time = slot("ra"), slot("dec"), slot("psi"), slot("geocent_time")
        cfg.abi_version, cfg.device, cfg.n_ifo, cfg.tidal = L.ABI_VERSION, self.device, self.n_ifo, SUPPORTED_APPROXIMANTS[approximant]
        cfg.n_freq, cfg.duration, cfg.start_time = n_freq, duration, start
        cfg.data = np.ascontiguousarray(data.view(np.float64)).ctypes.data_as(L._pd)
        cfg.psd = psd.ctypes.data_as(L._pd)
        cfg.mask = mask.ctypes.data_as(C.POINTER(C.c_uint8))
        cfg.detector_tensor, cfg.vertex = tensor.ctypes.data_as(L._pd), vertex.ctypes.data_as(L._pd)
        t_ref = float(gmst_reference_time if gmst_reference_time is not None else fixed.get("geocent_time", start + duration - 2.0))
        cfg.gmst_ref_time = t_ref
        cfg.gmst_ref, cfg.gmst_rate = gmst_linearisation(t_ref)
        cfg.reference_frequency = float(wa.get("reference_frequency", 50.0))           # bilby's default
        cfg.waveform_minimum_frequency = float(wa.get("minimum_frequency", 20.0))
        cfg.waveform_maximum_frequency = float(wa.get("maximum_frequency", np.inf))
        cfg.phase_marginalization = 1 if phase_marginalization else 0
        cfg.n_dim = len(names)
        dist = None
        if distance_marginalization is not None:
            # (grid[n] in Mpc, ln(prior(d_j) * delta_d)[n]) -- see distance_marginalization_grid
            dgrid = np.ascontiguousarray(distance_marginalization[0], dtype=np.float64)
            dlogw = np.ascontiguousarray(distance_marginalization[1], dtype=np.float64)
            if dgrid.ndim != 1 or dgrid.shape != dlogw.shape or dgrid.size < 2:
                raise L.NMMAHipError("distance_marginalization needs (grid[n], log_weight[n]) with n >= 2")
            if "luminosity_distance" in names:
                raise L.NMMAHipError("luminosity_distance is marginalised: it must not be a sampled column")
            cfg.n_distance = dgrid.size
            cfg.distance_grid, cfg.distance_log_weight = dgrid.ctypes.data_as(L._pd), dlogw.ctypes.data_as(L._pd)
            dist = (dgrid, dlogw)
        tlw = None
        if time_marginalization is not None:
            # ln(prior(t_j) * delta_t) on the n_freq - 1 coalescence times of the segment -- see time_marginalization_weights
            tlw = np.ascontiguousarray(time_marginalization, dtype=np.float64)
            if tlw.shape != (n_freq - 1,):
                raise L.NMMAHipError(f"time_marginalization needs ln(prior x step) on the {n_freq - 1} time shifts of the segment")
            if "geocent_time" in names:
                raise L.NMMAHipError("geocent_time is marginalised: it must not be a sampled column")
            cfg.time_log_weight = tlw.ctypes.data_as(L._pd)
            cfg.time_jitter = L.Slot.constant(0.0)
            if time_jitter_bounds is not None:
                # bilby's jitter_time: the sampled time_jitter moves the waveform and the times the prior is evaluated at
                if "time_jitter" not in names:
                    raise L.NMMAHipError("jitter_time needs the sampled column 'time_jitter' (bilby adds its Uniform(-dt/2, dt/2) prior)")
                cfg.time_jitter = L.Slot.column(names.index("time_jitter"))
                cfg.time_prior_minimum, cfg.time_prior_maximum = float(time_jitter_bounds[0]), float(time_jitter_bounds[1])
        keep = (data, psd, mask, tensor, vertex, dist, tlw)        # alive until create returns (the library copies)
        h = C.c_void_p()
        L.check(self._lib.nmma_gw_create(C.byref(cfg), C.byref(h)), "nmma_gw_create")
        del keep
        self._handle = h
        self.n_bins = int(self._lib.nmma_gw_n_bins(h))
        self.phase_marginalization = bool(phase_marginalization)
        self.distance_marginalization = distance_marginalization is not None
        self.time_marginalization = time_marginalization is not None
        self._noise = float(self._lib.nmma_gw_noise_log_likelihood(h))

    # ---- lifetime
    def close(self):
        if getattr(self, "_handle", None):
            self._lib.nmma_gw_destroy(self._handle)
            self._handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def noise_log_likelihood(self):
        return self._noise

    def _theta(self, theta):
        import torch
        dev = torch.device(f"cuda:{self.device}")
        if not isinstance(theta, torch.Tensor):
            theta = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64))
        if theta.dtype != torch.float64 or theta.dim() != 2 or theta.shape[1] < len(self.parameter_names):
            raise L.NMMAHipError(f"theta must be float64 [B, >= {len(self.parameter_names)}], got {theta.dtype} {tuple(theta.shape)}")
        return theta.to(dev).contiguous(), dev

    def _call(self, fn, what, theta, width, out=None, stream=None):
        import torch
        theta, dev = self._theta(theta)
        n = theta.shape[0]
        shape = (n,) if width == 1 else (n, width)
        if out is None:
            out = torch.empty(shape, dtype=torch.float64, device=dev)
        elif out.dtype != torch.float64 or out.device != dev or tuple(out.shape) != shape or not out.is_contiguous():
            raise L.NMMAHipError(f"out must be a contiguous float64 tensor of shape {shape} on {dev}")
        if n == 0:
            return out
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        if s.device.index != self.device:
            raise L.NMMAHipError(f"stream belongs to cuda:{s.device.index}, the engine to cuda:{self.device}")
        # theta / out were staged on torch's current stream: a different launch stream has to wait for that work
        cur = torch.cuda.current_stream(self.device)
        if s.cuda_stream != cur.cuda_stream:
            s.wait_stream(cur)
        L.check(fn(self._handle, C.c_void_p(theta.data_ptr()), n, theta.stride(0), C.c_void_p(out.data_ptr()),
                   C.c_void_p(s.cuda_stream)), what)
        return out

    def loglike_ratio(self, theta, out=None, stream=None):
        """``theta[B, D]`` -> log-likelihood ratio ``[B]`` (torch CUDA tensor; asynchronous)."""
        return self._call(self._lib.nmma_gw_loglike, "nmma_gw_loglike", theta, 1, out, stream)

    def inner_products(self, theta, out=None, stream=None):
        """``[B, 3]``: Re<d|h>, Im<d|h>, <h|h> summed over the detectors."""
        return self._call(self._lib.nmma_gw_inner_products, "nmma_gw_inner_products", theta, 3, out, stream)

    def strain(self, theta):
        """The projected strain ``[B, n_ifo, n_freq]`` complex128 (tests / plots; B x n_ifo x n_freq x 16 bytes of HBM)."""
        import torch
        theta, dev = self._theta(theta)
        n = theta.shape[0]
        out = torch.empty((n, self.n_ifo, self.n_freq, 2), dtype=torch.float64, device=dev)
        L.check(self._lib.nmma_gw_strain(self._handle, C.c_void_p(theta.data_ptr()), n, theta.stride(0), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "nmma_gw_strain")
        return torch.view_as_complex(out)

    def profile_begin(self, max_launches=64):
        L.check(self._lib.nmma_gw_profile_begin(self._handle, int(max_launches)), "nmma_gw_profile_begin")

    def profile_end(self):
        ms, n = C.c_double(), C.c_int32()
        L.check(self._lib.nmma_gw_profile_end(self._handle, C.byref(ms), C.byref(n)), "nmma_gw_profile_end")
        return ms.value, n.value


def _noise_log_likelihood_host(interferometers):
    """``-sum_ifo <d|d> / 2`` (bilby: GravitationalWaveTransient.noise_log_likelihood) -- a constant of the data, evaluated once
    on the host at construction so that the object answers without touching the GPU."""
    total = 0.0
    for ifo in interferometers:
        m = np.asarray(ifo.frequency_mask, dtype=bool)
        d = np.asarray(ifo.frequency_domain_strain)[m]
        s = np.asarray(ifo.power_spectral_density_array)[m]
        total -= 0.5 * 4.0 / float(ifo.strain_data.duration) * float(np.sum((d.real ** 2 + d.imag ** 2) / s))
    return total


def distance_marginalization_grid(prior, n=10000):
    """What bilby's ``GravitationalWaveTransient._setup_distance_marginalization`` tabulates from the luminosity-distance
    prior: ``linspace(prior.minimum, prior.maximum, n)`` and ``prior.prob`` on it times the step (bilby/gw/likelihood/base.py) --
    returned as (grid, ln(prob x step)) for ``GWEngine(distance_marginalization=...)``, plus bilby's reference distance
    ``prior.rescale(0.5)`` at which the waveform is evaluated (any distance gives the same marginal: the device rescales)."""
    grid = np.linspace(float(prior.minimum), float(prior.maximum), int(n))
    prob = np.asarray(prior.prob(grid), dtype=np.float64)
    with np.errstate(divide="ignore"):
        logw = np.log(prob * (grid[1] - grid[0]))
    ref = float(prior.rescale(0.5)) if hasattr(prior, "rescale") else float(0.5 * (grid[0] + grid[-1]))
    return grid, logw, ref


class _UniformLike:
    """The prior bilby adds for ``time_jitter`` when no bilby is around to supply ``bilby.core.prior.Uniform``: minimum, maximum,
    ``prob`` and ``rescale`` with bilby's meaning (enough for the batched prior transform and the samplers' bookkeeping)."""

    def __init__(self, minimum, maximum, name="time_jitter"):
        self.minimum, self.maximum, self.name, self.latex_label = float(minimum), float(maximum), name, name

    def prob(self, val):
        val = np.asarray(val, dtype=np.float64)
        return np.where((val >= self.minimum) & (val <= self.maximum), 1.0 / (self.maximum - self.minimum), 0.0)

    def rescale(self, val):
        return self.minimum + np.asarray(val) * (self.maximum - self.minimum)


def time_marginalization_weights(prior, start_time, duration, n_freq):
    """What bilby's ``_setup_time_marginalization`` tabulates from the ``geocent_time`` prior (bilby/gw/likelihood/base.py): the
    coalescence times the FFT of the integrand resolves, ``delta_tc = duration / (n_freq - 1)`` (= 2 / sampling_frequency) apart,
    and ``prior.prob`` on them times the step, as ln(prob x step).  Index j is the shift j * delta_tc of a waveform evaluated
    with ``geocent_time = start_time`` (bilby's delta-function replacement of the time prior), so FFT bin j is weighted with the
    prior AT the coalescence time it stands for, ``start_time + j delta_tc``.  Stated deviation (unpinned -- bilby is not in the
    image): bilby's own table is, to the best of two recollections, ``start_time + linspace(0, T, n + 1)[1:]``, i.e. bin j paired
    with the prior one node LATER; for a prior that is flat over its support the two differ only in which edge node carries
    weight."""
    n = int(n_freq) - 1
    delta = float(duration) / n
    times = float(start_time) + delta * np.arange(n)
    prob = np.asarray(prior.prob(times), dtype=np.float64)
    with np.errstate(divide="ignore"):
        return np.log(prob * delta)


class GravitationalWaveTransient:
    """What ``GravitationalWaveTransientLikelihood.sub_model`` exposes of ``bilby.gw.likelihood.GravitationalWaveTransient``:
    the objects it was built from, the marginalisation flags, ``noise_log_likelihood`` and the evaluation -- per sample
    (``log_likelihood(parameters)``, a batch of one) and batched (``log_likelihood_ratio_batch``).  The GPU handle is created
    lazily per process and dropped on pickling, like the EM likelihood's."""

    def __init__(self, interferometers, waveform_generator, priors=None, phase_marginalization=False, device=0,
                 distance_marginalization=False, time_marginalization=False, jitter_time=False):
        self.interferometers, self.waveform_generator, self.priors = list(interferometers), waveform_generator, priors
        self.jitter_time = bool(jitter_time) and bool(time_marginalization)
        self.phase_marginalization, self.time_marginalization = bool(phase_marginalization), bool(time_marginalization)
        self.distance_marginalization = bool(distance_marginalization)
        self._distance, self._time_logw, self._time_bounds = None, None, None
        if self.time_marginalization:
            try:
                ifo = self.interferometers[0]
                prior = priors["geocent_time"]
                self._time_logw = time_marginalization_weights(prior, ifo.strain_data.start_time,
                                                               ifo.strain_data.duration, len(ifo.frequency_array))
                self._time_bounds = (float(prior.minimum), float(prior.maximum)) if self.jitter_time else None
                if self.jitter_time:
                    # the kernel weights a jittered node with the tabulated weight of its nearest in-support node: exact only
                    # when the prior is flat over its support (bilby evaluates prior.prob(times + jitter) per sample)
                    inside = self._time_logw[np.isfinite(self._time_logw)]
                    if inside.size and np.ptp(inside) > 1e-9:
                        # (bilby's default is jitter_time=True: a reference-shaped set-up with, say, a Gaussian time prior must
                        #  keep running -- without the jitter the time grid is fixed, which is bilby's own jitter_time=False)
                        import warnings
                        warnings.warn("jitter_time with a non-uniform geocent_time prior: the device path cannot evaluate the time prior per "
                                      "sample; continuing with jitter_time=False (a fixed time grid)", RuntimeWarning, stacklevel=2)
                        self.jitter_time, self._time_bounds = False, None
                        # (on record: a result file's meta data then says that this run differs from bilby's configuration)
                        self.deviations_from_reference = getattr(self, "deviations_from_reference", []) + [
                            "jitter_time=True requested with a non-uniform geocent_time prior: run with jitter_time=False (no time_jitter prior)"]
                        self.meta_data = dict(getattr(self, "meta_data", None) or {}, deviations_from_reference=list(self.deviations_from_reference))
                if self.jitter_time and "time_jitter" not in priors:
                    # bilby/gw/likelihood/base.py: priors['time_jitter'] = Uniform(-delta_tc / 2, delta_tc / 2)
                    half = 0.5 * float(ifo.strain_data.duration) / (len(ifo.frequency_array) - 1)
                    priors["time_jitter"] = _UniformLike(-half, half)
            except (KeyError, TypeError, AttributeError) as exc:
                raise L.NMMAHipError("time marginalisation needs priors['geocent_time'] with prob()") from exc
        if self.distance_marginalization:
            try:
                prior = priors["luminosity_distance"]
                self._distance = distance_marginalization_grid(prior)
            except (KeyError, TypeError, AttributeError) as exc:
                raise L.NMMAHipError("distance marginalisation needs priors['luminosity_distance'] with minimum, maximum and prob()") from exc
        self.device = int(device)
        self._noise = _noise_log_likelihood_host(self.interferometers)
        self._engine, self._names = None, None

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"], state["_names"] = None, None
        return state

    def fixed_parameters(self, names):
        from ..core.base import fixed_value
        fixed = {}
        for key, prior in (self.priors.items() if hasattr(self.priors, "items") else ()):
            val = fixed_value(prior)
            if val is not None and key not in names:
                fixed[key] = val
        if self.phase_marginalization:
            fixed["phase"] = 0.0        # bilby: the phase prior becomes a delta function at 0
        if self.distance_marginalization:
            fixed["luminosity_distance"] = self._distance[2]      # bilby: ... and the distance prior one at the reference distance
        if self.time_marginalization:
            fixed["geocent_time"] = float(self.interferometers[0].strain_data.start_time)     # bilby: ... the time prior one at the segment start
        return fixed

    def engine(self, names):
        names = [n for n in names]
        if self.phase_marginalization and "phase" in names:
            raise L.NMMAHipError("phase is marginalised: it must not be a sampled column")
        if self.distance_marginalization and "luminosity_distance" in names:
            raise L.NMMAHipError("luminosity_distance is marginalised: it must not be a sampled column")
        if self.time_marginalization and "geocent_time" in names:
            raise L.NMMAHipError("geocent_time is marginalised: it must not be a sampled column")
        if self._engine is None or self._names != names:
            if self._engine is not None:
                self._engine.close()
            self._engine = GWEngine(self.interferometers, names, fixed=self.fixed_parameters(names),
                                    waveform_arguments=self.waveform_generator.waveform_arguments,
                                    phase_marginalization=self.phase_marginalization, device=self.device,
                                    distance_marginalization=self._distance[:2] if self.distance_marginalization else None,
                                    time_marginalization=self._time_logw,
                                    time_jitter_bounds=self._time_bounds if self.time_marginalization else None,
                                    gmst_reference_time=self._gmst_reference_time())
            self._names = names
        return self._engine

    def _gmst_reference_time(self):
        """Where the sidereal time is linearised: with the time marginalised the waveform sits at the segment start while the signal
        sits in the support of the time prior -- take the prior's centre (the antenna pattern is evaluated at the row's geocent_time,
        as in bilby, so this only keeps the linearisation error of the rotation negligible)."""
        if not self.time_marginalization:
            return None
        ifo = self.interferometers[0]
        return float(ifo.strain_data.start_time)

    def noise_log_likelihood(self):
        return self._noise

    def log_likelihood_ratio_batch(self, theta, names, out=None, stream=None):
        return self.engine(names).loglike_ratio(theta, out=out, stream=stream)

    def log_likelihood_ratio(self, parameters):
        fixed = self.fixed_parameters(())
        names = sorted(k for k in parameters if k in _GW_KEYS and k not in fixed)
        row = np.array([[float(parameters[k]) for k in names]])
        return float(self.log_likelihood_ratio_batch(row, names).cpu().numpy()[0])

    def log_likelihood(self, parameters):
        return self.log_likelihood_ratio(parameters) + self._noise


_GW_KEYS = set(GW_PARAMETERS) | {"chirp_mass", "mass_ratio", "mass_1", "mass_2", "cos_theta_jn", "time_jitter"}


class GravitationalWaveTransientLikelihood(NMMALikelihood):
    """``nmma/gw/gw_likelihood.py:97-247``: the GW messenger.  Same constructor, same attributes (``sub_model`` with
    ``interferometers`` / ``waveform_generator``, ``parameter_conversion`` = the neutron-star or black-hole source-frame
    conversion chosen from the source model's name, :207-210), with the arithmetic bilby + lalsimulation do per sample
    moved to the GPU for a whole batch (``log_likelihood_batch``).

    Built on the device path: ``gw_likelihood_type='GravitationalWaveTransient'`` with ``phase_marginalization`` and
    ``distance_marginalization`` on or off (the distance sum evaluated per row instead of bilby's lookup table) or
    ``time_marginalization`` (FFT of the per-bin integrand over the coalescence-time shifts; with ``jitter_time`` the sampled
    ``time_jitter`` column, whose prior is added as bilby does), in any combination, sky reference frame, geocentre time reference, approximants ``IMRPhenomD_NRTidalv2`` / ``IMRPhenomD`` with aligned spins.
    Refused at construction (never approximated): the ROQ / relative-binning / multibanded likelihood classes (they need
    bilby's basis files and fiducial waveforms), other reference frames."""

    def __init__(self, priors, interferometers, waveform_generator, gw_likelihood_type="GravitationalWaveTransient",
                 time_marginalization=False, distance_marginalization=False, phase_marginalization=False,
                 distance_marginalization_lookup_table=None, jitter_time=True, reference_frame="sky",
                 time_reference="geocenter", device=0, **kwargs):
        waveform_generator.parameter_conversion = self.gw_identity_conversion            # :167
        waveform_generator.start_time = interferometers[0].time_array[0]                 # :168
        known = ("GravitationalWaveTransient", "ROQGravitationalWaveTransient", "RelativeBinningGravitationalWaveTransient",
                 "MBGravitationalWaveTransient")
        if gw_likelihood_type not in known:
            raise ValueError("Unknown GW Likelihood class {}")                            # :205 (sic)
        if gw_likelihood_type != "GravitationalWaveTransient":
            raise L.NMMAHipError(f"{gw_likelihood_type} is not built on the device path (it needs bilby's basis / fiducial data)")
        if distance_marginalization and distance_marginalization_lookup_table is not None:
            pass        # (bilby caches its (d_inner_h, h_inner_h) table there; the device evaluates the sum per row and needs none)
        if reference_frame != "sky" or time_reference not in ("geocent", "geocenter"):
            raise L.NMMAHipError("only reference_frame='sky' and time_reference='geocenter' are built on the device path")
        sub_model = GravitationalWaveTransient(interferometers, waveform_generator, priors=priors,
                                               phase_marginalization=phase_marginalization, device=device,
                                               distance_marginalization=distance_marginalization,
                                               time_marginalization=time_marginalization, jitter_time=jitter_time)
        super().__init__(sub_model, priors)
        from ..core import conversion
        name = getattr(waveform_generator.frequency_domain_source_model, "__name__", "")
        self.parameter_conversion = conversion.bns_source_frame if "neutron_star" in name else conversion.bbh_source_frame

    def gw_identity_conversion(self, parameters):
        return parameters, []

    def sanity_checks(self):
        return True

    def final_diagnostics(self, bestfit_params, args, result=None):
        return None

    def noise_log_likelihood(self):
        return self.sub_model.noise_log_likelihood()

    def posterior_conversion(self, posterior_samples):
        """:212-236: chi_eff and the effective tidal deformabilities, where their ingredients are present."""
        from ..core.conversion import tidal_deformabilities_and_mass_ratio_to_eff_tidal_deformabilities as tidal_conversion
        if "chi_eff" not in posterior_samples:
            try:
                q = posterior_samples["mass_ratio"]
                chi_1 = posterior_samples.get("chi_1", posterior_samples["spin_1z"])
                chi_2 = posterior_samples.get("chi_2", posterior_samples["spin_2z"])
                posterior_samples["chi_eff"] = (chi_1 + q * chi_2) / (1 + q)
            except KeyError:
                pass
        if "lambda_tilde" not in posterior_samples:
            try:
                lam_t, dlam_t = tidal_conversion(posterior_samples["lambda_1"], posterior_samples["lambda_2"],
                                                 posterior_samples["mass_ratio"])
                posterior_samples["lambda_tilde"], posterior_samples["delta_lambda_t"] = lam_t, dlam_t
            except KeyError:
                pass
        return posterior_samples

    # ---- batched path
    def log_likelihood_batch(self, theta, names=None, out=None, stream=None):
        """log L for every row of ``theta[B, D]`` (columns ``names``; default: the non-fixed, non-constraint priors in order):
        log-likelihood ratio from the fused kernel + the noise term; rows that violate a Constraint prior or hold
        unphysical parameters get the floor.  Returns a torch CUDA tensor."""
        import torch
        from ..core.base import LOGL_FLOOR
        if names is None:
            from ..core.base import fixed_value, is_constraint
            names = [k for k, p in self.priors.items() if fixed_value(p) is None and not is_constraint(p)]
        names = list(names)
        gw_cols = [i for i, n in enumerate(names) if n in _GW_KEYS]
        gw_names = [names[i] for i in gw_cols]
        th = theta if isinstance(theta, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64))
        if gw_cols != list(range(len(names))):
            th = th[:, gw_cols]
        ratio = self.sub_model.log_likelihood_ratio_batch(th, gw_names, out=out, stream=stream)
        logl = torch.where(ratio > LOGL_FLOOR, ratio + self.sub_model.noise_log_likelihood(), ratio)
        if self.constraints:
            logl = self.apply_constraints_batch(logl, theta, names, self.sub_model.fixed_parameters(names))
        return logl
