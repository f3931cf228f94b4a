"""EMEngine: one GPU-resident EM likelihood (a ``nmma_em_handle``) and its batched calls.

This is the thin layer between the reference-shaped Python classes
(``nmma_amd.em.model.SVDLightCurveModel``, ``nmma_amd.em.em_likelihood.EMTransientLikelihood``)
and the C ABI.  It translates *names* (model parameters, sampled parameters, filters)
into the flat ``nmma_em_config`` the library copies to the device, and moves batches
``theta[B, D]`` through the kernels.  torch is used only to hold device buffers and
to supply the stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

_AVG_DEFAULT_LD = 1e-5   # nmma/em/model.py:291-293: default distance 10 pc = 1e-5 Mpc


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def resolve_model_param_slot(key, names, fixed):
    """Where one surrogate input comes from, following the reference's conversions:
    nmma/core/conversion.py:119-126 (KNtheta <-> inclination_EM / theta_jn) and
    nmma/em/model.py:272-286 (log10_ aliases, including its ``lstrip`` semantics)."""
    names = list(names)
    if key in names:
        return L.Slot.column(names.index(key))
    if key in fixed:
        return L.Slot.constant(fixed[key])
    if key == "KNtheta":
        if "inclination_EM" in names:
            return L.Slot.column(names.index("inclination_EM"), L.OP_RAD2DEG)
        if "inclination_EM" in fixed:
            return L.Slot.constant(fixed["inclination_EM"] * 180.0 / np.pi)
        if "theta_jn" in names:
            return L.Slot.column(names.index("theta_jn"), L.OP_THETAJN2DEG)
        if "cos_theta_jn" in names:
            return L.Slot.column(names.index("cos_theta_jn"), L.OP_COSTHETAJN2DEG)
        tj = fixed.get("theta_jn", np.arccos(fixed.get("cos_theta_jn", 1.0)))
        return L.Slot.constant(min(tj, np.pi - tj) * 180.0 / np.pi)
    if key == "inclination_EM":
        if "KNtheta" in names:
            return L.Slot.column(names.index("KNtheta"), L.OP_DEG2RAD)
        if "KNtheta" in fixed:
            return L.Slot.constant(fixed["KNtheta"] / 180.0 * np.pi)
    stripped = key.lstrip("log10_")          # sic: character-set strip, as the reference does
    if stripped in names:
        return L.Slot.column(names.index(stripped), L.OP_LOG10)
    if stripped in fixed:
        return L.Slot.constant(np.log10(fixed[stripped]))
    if "log10_" + key in names:
        return L.Slot.column(names.index("log10_" + key), L.OP_POW10)
    if "log10_" + key in fixed:
        return L.Slot.constant(10 ** fixed["log10_" + key])
    raise KeyError(f"model parameter {key!r} is neither sampled nor fixed "
                   f"(sampled: {names}, fixed: {sorted(fixed)})")


def _plain_slot(key, names, fixed, default):
    names = list(names)
    if key in names:
        return L.Slot.column(names.index(key))
    return L.Slot.constant(fixed.get(key, default))


class EMEngine:
    """Owns one ``nmma_em_handle``.

    Parameters
    ----------
    svd_model : dict  filter -> dict(W1, b1, W2, b2, VA, mins, maxs, tt, param_mins, param_maxs, n_coeff)
    model_filters : list[str]   model filters to upload (order defines indices)
    model_parameters : list[str]  surrogate inputs, in the model's order
    parameter_names : list[str]   columns of ``theta``
    fixed : dict  fixed (delta-function) parameters
    sample_times : array or None  (None: the SVD training grid, model.py:655-660)
    cosmo_grid : (dist_grid, z_grid) or None
    data : (times, mags, sigmas) dicts keyed by observed filter, or None (model-only engine)
    observed_filters : list[str]
    sources : dict observed filter -> list of model filters (1, or 2-3 to average)
    detection_limit : dict observed filter -> float
    systematics : dict, see ``nmma_amd.em.systematics.FilterSystematicsHandler.kernel_spec``
    ebv_coeff : dict model filter -> A_filter / E(B-V), or None (linear law: ext_mag = coeff * Ebv)
    extinction_law : None / "linear" (``ebv_coeff``) or "P92_SMC_host" (Pei 1992 SMC curve at every sample's
        host-frame wavelength, utils.py:373-428; needs ``filter_nu0``: dict model filter -> Hz)
    """

    def __init__(self, svd_model, model_filters, model_parameters, parameter_names, fixed=None,
                 sample_times=None, cosmo_grid=None, data=None, observed_filters=(), sources=None,
                 detection_limit=None, systematics=None, ebv_coeff=None, device=0, n_coeff=None,
                 model_kind="svd", filter_nu0=None, extinction_law=None, hubble_reference=None, stack_operands=0, base_times=None,
                 null_filters=()):
        self._handle = None
        self._host_out = {}
        lib = L.load_library()
        fixed = dict(fixed or {})
        names = list(parameter_names)
        model_filters = list(model_filters)
        self.parameter_names = names
        self.model_filters = model_filters
        self.observed_filters = list(observed_filters)
        self.device = int(device)

        kinds = {"svd": L.MODEL_SVD, "me2017": L.MODEL_ME2017, "external": L.MODEL_EXTERNAL}
        if model_kind not in kinds:
            raise L.NMMAHipError(f"unknown model_kind {model_kind!r}")
        self.model_kind = model_kind
        cfg = L.EmConfig()
        keep = []
        cfg.abi_version, cfg.device = L.ABI_VERSION, self.device
        cfg.model_kind = kinds[model_kind]
        cfg.n_model_filters = len(model_filters)
        # model filters the surrogate has nothing for (calc_svd_lc's null output for "radio and X-ray filters when using with GRB data",
        # lightcurve_generation.py:168-169): +inf on every node.  Only inside a combination (stack_operands=1); the library ignores
        # their tensors -- zeros of the common shapes stand in
        null_filters = [f for f in model_filters if f in set(null_filters)]
        self.null_filters = null_filters
        if model_kind == "svd":
            if null_filters:
                real = [f for f in model_filters if f not in null_filters]
                if not real:
                    raise L.NMMAHipError("every model filter is a null filter: the surrogate contributes nothing")
                tmpl = svd_model[real[0]]
                blank = dict(tmpl)
                for k in ("W1", "b1", "W2", "b2", "VA"):
                    blank[k] = np.zeros_like(np.asarray(tmpl[k]))
                svd_model = dict(svd_model)
                for f in null_filters:
                    svd_model[f] = blank
            first = svd_model[model_filters[0]]
            n_p = int(np.asarray(first["W1"]).shape[0])
            n_h = int(np.asarray(first["W1"]).shape[1])
            nc_trained = int(first["n_coeff"])
            # lightcurve_generation.py:182-185; with a Keras net the output width is fixed
            n_c = min(int(n_coeff), nc_trained) if n_coeff else nc_trained
            if n_c != int(np.asarray(first["W2"]).shape[1]):
                raise L.NMMAHipError("svd_mag_ncoeff must equal the network's output width "
                                     "(nmma/em/lightcurve_generation.py:182-198)")
            n_t = len(first["tt"])
            if len(model_parameters) != n_p:
                raise L.NMMAHipError("model_parameters do not match the surrogate input width")
            for f in model_filters:
                t = svd_model[f]
                if (np.asarray(t["W1"]).shape != (n_p, n_h) or np.asarray(t["W2"]).shape != (n_h, n_c)
                        or len(t["tt"]) != n_t):
                    raise L.NMMAHipError(f"filter {f}: surrogate shapes differ between filters")
            stack = lambda k, conv: conv(np.stack([np.asarray(svd_model[f][k]) for f in model_filters]))
            W1, b1, W2, b2 = (stack(k, _f32) for k in ("W1", "b1", "W2", "b2"))
            VA = _f64(np.stack([np.asarray(svd_model[f]["VA"])[:, :n_c] for f in model_filters]))
            mins, maxs, tt = (stack(k, _f64) for k in ("mins", "maxs", "tt"))
            pmin, pmax = stack("param_mins", _f64), stack("param_maxs", _f64)
            keep += [W1, b1, W2, b2, VA, mins, maxs, tt, pmin, pmax]
            cfg.n_params, cfg.n_hidden = n_p, n_h
            cfg.n_coeff, cfg.n_tt = n_c, n_t
            cfg.W1, cfg.b1, cfg.W2, cfg.b2 = (_ptr(a, C.c_float) for a in (W1, b1, W2, b2))
            cfg.VA, cfg.mins, cfg.maxs, cfg.tt = (_ptr(a, C.c_double) for a in (VA, mins, maxs, tt))
            cfg.param_mins, cfg.param_maxs = _ptr(pmin, C.c_double), _ptr(pmax, C.c_double)
        else:
            if sample_times is None:
                raise L.NMMAHipError("sample_times are required for non-SVD models")
            n_p, n_h, n_c, n_t = len(model_parameters), 0, 0, 0
            tt = None
            cfg.n_params = n_p
        if filter_nu0 is not None:
            nu0 = _f64([filter_nu0[f] for f in model_filters])
            keep.append(nu0)
            cfg.filter_nu0 = _ptr(nu0, C.c_double)
        self.n_params, self.n_hidden, self.n_coeff, self.n_tt = n_p, n_h, n_c, n_t

        if sample_times is not None:
            st = _f64(sample_times)
            keep.append(st)
            cfg.n_sample_times, cfg.sample_times = len(st), _ptr(st, C.c_double)
            self.sample_times = st
        else:
            cfg.n_sample_times = 0
            self.sample_times = tt[0].copy()

        # redshift: conversion.py:57-64, model.py:255-267
        cfg.n_dim = len(names)
        if "redshift" in names or "redshift" in fixed:
            cfg.redshift_mode = L.Z_SLOT
            cfg.redshift = _plain_slot("redshift", names, fixed, 0.0)
        elif "Hubble_constant" in names and ("luminosity_distance" in names or "luminosity_distance" in fixed):
            # every sample carries its own cosmology (core/base.py:161-164, core/conversion.py:57-101): the grid is tabulated for
            # `hubble_reference` and read at d_L * H0 / hubble_reference (flat universe: d_L scales as c / H0 at fixed z)
            if cosmo_grid is None or not hubble_reference:
                raise L.NMMAHipError("Hubble_constant is sampled: pass the z(d_L) grid of a reference H0 as cosmo_grid together with "
                                     "hubble_reference (SVDLightCurveModel.check_vs_priors builds both from the priors)")
            dg, zg = _f64(cosmo_grid[0]), _f64(cosmo_grid[1])
            keep += [dg, zg]
            cfg.redshift_mode, cfg.n_cosmo = L.Z_GRID, len(dg)
            cfg.dist_grid, cfg.z_grid = _ptr(dg, C.c_double), _ptr(zg, C.c_double)
            cfg.redshift = L.Slot.constant(0.0)
            cfg.hubble_constant = L.Slot.column(names.index("Hubble_constant"))
            cfg.hubble_reference = float(hubble_reference)
        elif "luminosity_distance" in fixed and "luminosity_distance" not in names:
            # a FIXED distance still carries its redshift: the reference's get_cosmo_grids(d, d) is a constant
            # grid, so np.interp returns z(d_L) (model.py:255-267, conversion.py:49-55)
            from .core.conversion import redshift_at_distance
            cfg.redshift_mode = L.Z_SLOT
            cfg.redshift = L.Slot.constant(redshift_at_distance(fixed["luminosity_distance"], cosmo_grid))
        elif "luminosity_distance" in names:
            if cosmo_grid is None:
                raise L.NMMAHipError("luminosity_distance is sampled but no z(d_L) grid was supplied: call "
                                     "check_vs_priors() with a bounded prior, or pass cosmo_grid / a redshift column")
            dg, zg = _f64(cosmo_grid[0]), _f64(cosmo_grid[1])
            keep += [dg, zg]
            cfg.redshift_mode, cfg.n_cosmo = L.Z_GRID, len(dg)
            cfg.dist_grid, cfg.z_grid = _ptr(dg, C.c_double), _ptr(zg, C.c_double)
            cfg.redshift = L.Slot.constant(0.0)
        else:
            cfg.redshift_mode = L.Z_ZERO
            cfg.redshift = L.Slot.constant(0.0)
        for p, key in enumerate(model_parameters):
            cfg.model_param[p] = resolve_model_param_slot(key, names, fixed)
        for p in range(n_p, L.MAX_PARAMS):
            cfg.model_param[p] = L.Slot.constant(0.0)
        cfg.luminosity_distance = _plain_slot("luminosity_distance", names, fixed, _AVG_DEFAULT_LD)
        cfg.timeshift = _plain_slot("timeshift", names, fixed, 0.0)
        cfg.ebv = _plain_slot("Ebv", names, fixed, 0.0)
        laws = {None: L.EXT_LINEAR, "linear": L.EXT_LINEAR, "P92_SMC_host": L.EXT_P92_SMC_HOST}
        if extinction_law not in laws:
            # G23_MW (and anything else z-independent) enters as ebv_coeff: its curve is third-party data
            raise L.NMMAHipError(f"extinction_law {extinction_law!r} is not evaluated natively; pass its "
                                 "A_filter / E(B-V) as ebv_coeff (known: 'P92_SMC_host')")
        cfg.extinction_law = laws[extinction_law]
        if cfg.extinction_law == L.EXT_P92_SMC_HOST and filter_nu0 is None:
            raise L.NMMAHipError("extinction_law 'P92_SMC_host' needs filter_nu0 (observer-frame filter frequencies, Hz)")
        if ebv_coeff is not None:
            ec = _f64([ebv_coeff[f] for f in model_filters])
            keep.append(ec)
            cfg.ebv_coeff = _ptr(ec, C.c_double)
        elif cfg.extinction_law == L.EXT_LINEAR and ("Ebv" in names or fixed.get("Ebv", 0.0) != 0.0):
            raise L.NMMAHipError("Ebv is sampled (or fixed non-zero) but no extinction law can be evaluated: pass "
                                 "ebv_coeff, or extinction_law='P92_SMC_host' with filter_nu0 / filter_lambdas")

        # photometry + systematics
        obs = list(self.observed_filters)
        srcs = dict(sources or {f: [f] for f in obs})
        self.n_real_observed = len(obs)
        if obs and model_kind == "svd":
            # the reference evaluates and sanity-checks EVERY model filter (em_likelihood.py:305-311), observed or not: a model
            # filter no observed band draws on rides along as a band without data (its coefficients are checked, nothing else)
            used = {mf for f in obs for mf in srcs[f]}
            times, mags, sigmas = (dict(d) for d in data)
            for mf in model_filters:
                if mf not in used:
                    ghost = f"__unobserved__{mf}"
                    obs.append(ghost)
                    srcs[ghost] = [mf]
                    times[ghost] = mags[ghost] = sigmas[ghost] = np.zeros(0)
            data = (times, mags, sigmas)
        cfg.n_obs_filters = self._n_obs_uploaded = len(obs)
        if obs:
            times, mags, sigmas = data
            offs = np.zeros(len(obs) + 1, dtype=np.int32)
            for i, f in enumerate(obs):
                if not (len(times[f]) == len(mags[f]) == len(sigmas[f])):
                    raise L.NMMAHipError(f"filter {f}: ragged photometry arrays")
                offs[i + 1] = offs[i] + len(times[f])
            cat = lambda d: _f64(np.concatenate([np.asarray(d[f], dtype=float) for f in obs])
                                 if offs[-1] else np.zeros(0))
            dt, dm, ds = cat(times), cat(mags), cat(sigmas)
            lim = _f64([np.inf if detection_limit is None else detection_limit.get(f, np.inf) for f in obs])
            nsrc = _i32([len(srcs[f]) for f in obs])
            src = np.zeros((len(obs), L.MAX_SOURCES), dtype=np.int32)
            for i, f in enumerate(obs):
                for k, mf in enumerate(srcs[f]):
                    src[i, k] = model_filters.index(mf)
            sysk, sysc, sysn, sysoff, slots, node_t = self._systematics_arrays(systematics, obs, names, fixed)
            keep += [offs, dt, dm, ds, lim, nsrc, src, sysk, sysc, sysn, sysoff, node_t]
            cfg.data_offsets = _ptr(offs, C.c_int32)
            cfg.data_times, cfg.data_mags, cfg.data_sigmas = (_ptr(a, C.c_double) for a in (dt, dm, ds))
            cfg.detection_limit = _ptr(lim, C.c_double)
            cfg.n_sources, cfg.sources = _ptr(nsrc, C.c_int32), _ptr(src, C.c_int32)
            cfg.sys_kind, cfg.sys_const = _ptr(sysk, C.c_int32), _ptr(sysc, C.c_double)
            cfg.sys_n_nodes, cfg.sys_slot_offsets = _ptr(sysn, C.c_int32), _ptr(sysoff, C.c_int32)
            slot_arr = (L.Slot * max(1, len(slots)))(*slots)
            keep.append(slot_arr)
            cfg.sys_slots = C.cast(slot_arr, C.POINTER(L.Slot))
            cfg.sys_node_times = _ptr(node_t, C.c_double)
            self.n_data = int(offs[-1])
        else:
            self.n_data = 0

        # (a combined model whose second transient arrives per call: lay the handle out for the one-launch form, ``loglike_stack2``)
        cfg.stack_operands = int(stack_operands)
        # a combined model on a UNION grid (model.py:1372-1374): `sample_times` is the combination's grid, `base_times` the surrogate's own
        if null_filters:
            nf = np.ascontiguousarray([1 if f in null_filters else 0 for f in model_filters], dtype=np.int32)
            keep.append(nf)
            cfg.null_filters = _ptr(nf, C.c_int32)
        self.base_times = None
        if base_times is not None:
            bt = _f64(base_times)
            keep.append(bt)
            cfg.n_base_times, cfg.base_times = len(bt), _ptr(bt, C.c_double)
            self.base_times = bt
        h = C.c_void_p()
        L.check(lib.nmma_em_create(C.byref(cfg), C.byref(h)), "nmma_em_create")
        self._handle = h
        self._lib = lib
        self.n_sample_times = int(lib.nmma_em_n_sample_times(h))
        self.flops_per_eval = int(lib.nmma_em_flops_per_eval(h))
        del keep

    @staticmethod
    def _systematics_arrays(spec, obs, names, fixed):
        spec = spec or {"mode": "budget", "values": {f: 1.0 for f in obs}}
        kind = np.zeros(len(obs), dtype=np.int32)
        const = np.zeros(len(obs))
        nn = np.zeros(len(obs), dtype=np.int32)
        off = np.zeros(len(obs) + 1, dtype=np.int32)
        slots, node_t = [], []
        mode = spec["mode"]
        for i, f in enumerate(obs):
            off[i] = len(slots)
            if f.startswith("__unobserved__"):          # a band without data: no systematics to evaluate
                kind[i], const[i] = L.SYS_CONST, 1.0
                continue
            if mode == "budget":
                kind[i], const[i] = L.SYS_CONST, float(spec["values"][f])
                continue
            if mode == "param":
                pname, nodes = spec["name"], None
            elif f in spec.get("names", {}):
                pname, nodes = spec["names"][f], None
            elif f in spec.get("nodes", {}):
                pname, nodes = spec["nodes"][f]
            else:
                raise L.NMMAHipError(f"systematics spec has no entry for filter {f}")
            if nodes is None:
                kind[i], nn[i] = L.SYS_PARAM, 1
                slots.append(_plain_slot(pname, names, fixed, np.nan))
                node_t.append(0.0)
            else:
                kind[i], nn[i] = L.SYS_NODES, len(nodes)
                for p, t in zip(pname, nodes):
                    slots.append(_plain_slot(p, names, fixed, np.nan))
                    node_t.append(float(t))
        off[len(obs)] = len(slots)
        return kind, _f64(const), nn, off, slots, _f64(node_t if node_t else [0.0])

    @classmethod
    def from_case(cls, case, device=0, **extra):
        """Engine for a case dict of ``nmma_amd.synthetic.make_case`` (what bench.py, smoke() and the parity tests build);
        ``extra``: further constructor arguments (e.g. ``stack_operands=1``)."""
        from .em.utils import FILTER_AVERAGES, resolve_sources
        obs = list(case["observed_filters"])
        lim = case["detection_limit"]
        if not isinstance(lim, dict):
            lim = {f: float(lim) for f in obs}
        return cls(case["svd"], case["model_filters"], case["model_parameters"], case["names"],
                   fixed=case.get("fixed"), sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"],
                   data=case["data"], observed_filters=obs,
                   sources=resolve_sources(obs, case["model_filters"],
                                           known_filters=[f for f in obs if f not in FILTER_AVERAGES]),
                   detection_limit=lim, systematics=case["systematics"], ebv_coeff=case.get("ebv_coeff"),
                   filter_nu0=case.get("filter_nu0"),
                   extinction_law="P92_SMC_host" if case.get("filter_nu0") is not None else None,
                   hubble_reference=case.get("hubble_reference"), device=device, **extra)

    # ------------------------------------------------------------------ calls
    def _dev_theta(self, theta):
        import torch
        if isinstance(theta, torch.Tensor):
            t = theta
            if t.dtype != torch.float64 or not t.is_cuda or t.device.index != self.device:
                t = t.to(device=f"cuda:{self.device}", dtype=torch.float64)
        else:
            t = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64)).to(f"cuda:{self.device}")
        if t.dim() != 2 or t.shape[1] < len(self.parameter_names):
            raise L.NMMAHipError(f"theta must be [B, >={len(self.parameter_names)}], got {tuple(t.shape)}")
        return t.contiguous()

    def _stream(self, stream=None):
        """The launch stream: an explicit torch stream of the handle's device, else torch's current stream OF THAT
        DEVICE (the C side launches there)."""
        import torch
        if stream is not None:
            if stream.device.index != self.device:
                raise L.NMMAHipError(f"stream belongs to {stream.device}, the handle to cuda:{self.device}")
            return C.c_void_p(stream.cuda_stream)
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_out(self, out, n):
        import torch
        if (not isinstance(out, torch.Tensor) or out.dtype != torch.float64 or not out.is_cuda
                or out.device.index != self.device or out.numel() < n or not out.is_contiguous()):
            raise L.NMMAHipError(f"out must be a contiguous float64 tensor on cuda:{self.device} with >= {n} elements")

    def loglike(self, theta, out=None, stream=None):
        """logL for every row of ``theta[B, D]``.  torch CUDA tensor in -> torch tensor out
        (asynchronous on ``stream``, default torch's current stream); numpy in -> numpy out (synchronous)."""
        import torch
        if not isinstance(theta, torch.Tensor):
            th = _f64(theta)
            if th.ndim != 2 or th.shape[1] < len(self.parameter_names):
                raise L.NMMAHipError(f"theta must be [B, >={len(self.parameter_names)}], got {th.shape}")
            # (a reusable result buffer per batch size, with its address: the per-call Python overhead is kept to ~2 us)
            slot = self._host_out.get(th.shape[0])
            if slot is None:
                if len(self._host_out) >= 16:
                    self._host_out.clear()
                res = np.empty(th.shape[0])
                slot = self._host_out[th.shape[0]] = (res, res.ctypes.data)
            L.check(self._lib.nmma_em_loglike_host(self._handle, th.ctypes.data, th.shape[0], th.shape[1], slot[1]),
                    "nmma_em_loglike_host")
            return slot[0].copy()
        t = self._dev_theta(theta)
        if out is None:
            out = torch.empty(t.shape[0], dtype=torch.float64, device=t.device)
        else:
            self._check_out(out, t.shape[0])
        L.check(self._lib.nmma_em_loglike(self._handle, C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0),
                                          C.c_void_p(out.data_ptr()), self._stream(stream)), "nmma_em_loglike")
        return out

    def loglike_parts(self, theta):
        """(chi[O, B], gp[O, B]) per observed filter (em_likelihood.py:337-352)."""
        import torch
        t = self._dev_theta(theta)
        n_o = self._n_obs_uploaded            # (incl. the bands without data that stand for unobserved model filters)
        chi = torch.empty((n_o, t.shape[0]), dtype=torch.float64, device=t.device)
        gp = torch.empty_like(chi)
        L.check(self._lib.nmma_em_loglike_parts(self._handle, C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0),
                                                C.c_void_p(chi.data_ptr()), C.c_void_p(gp.data_ptr()),
                                                self._stream()), "nmma_em_loglike_parts")
        n_real = len(self.observed_filters)
        return chi[:n_real], gp[:n_real]

    def lightcurves(self, theta):
        """(obs_times[B, NS], mag[B, M, NS]) -- gen_detector_lc for every row (model.py:352-404)."""
        import torch
        t = self._dev_theta(theta)
        ns, m = self.n_sample_times, len(self.model_filters)
        tobs = torch.empty((t.shape[0], ns), dtype=torch.float64, device=t.device)
        mag = torch.empty((t.shape[0], m, ns), dtype=torch.float64, device=t.device)
        L.check(self._lib.nmma_em_lightcurves(self._handle, C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0),
                                              C.c_void_p(tobs.data_ptr()), C.c_void_p(mag.data_ptr()),
                                              self._stream()), "nmma_em_lightcurves")
        return tobs, mag

    def model_lightcurves(self, theta):
        """Source-frame absolute magnitudes lc[B, M, NS] of the handle's model (generate_lightcurve)."""
        import torch
        t = self._dev_theta(theta)
        lc = torch.empty((t.shape[0], len(self.model_filters), self.n_sample_times), dtype=torch.float64,
                         device=t.device)
        L.check(self._lib.nmma_em_model_lightcurves(self._handle, C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0),
                                                    C.c_void_p(lc.data_ptr()), self._stream()),
                "nmma_em_model_lightcurves")
        return lc

    def loglike_lc(self, theta, lc):
        """logL from supplied source-frame light curves lc[B, M, NS] (torch CUDA tensor)."""
        import torch
        t = self._dev_theta(theta)
        lc = lc.to(device=t.device, dtype=torch.float64).contiguous()
        if tuple(lc.shape) != (t.shape[0], len(self.model_filters), self.n_sample_times):
            raise L.NMMAHipError(f"lc must be [B, M, NS] = {(t.shape[0], len(self.model_filters), self.n_sample_times)}")
        out = torch.empty(t.shape[0], dtype=torch.float64, device=t.device)
        L.check(self._lib.nmma_em_loglike_lc(self._handle, C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0),
                                             C.c_void_p(lc.data_ptr()), C.c_void_p(out.data_ptr()), self._stream()),
                "nmma_em_loglike_lc")
        return out

    def loglike_lc_sets(self, theta, lcs, bad_rows=None):
        """logL of a COMBINED model from its sub-models' source-frame sets ``lcs`` (list of [B, M, NS] CUDA tensors on this
        engine's grid and filters): the flux sum (``stack``) is formed on chip while each sample's curves are staged, the stacked
        set is never written.  ``bad_rows``: bool/uint8 CUDA tensor [B] of rows without a light curve (floor), or None."""
        import torch
        t = self._dev_theta(theta)
        shape = (t.shape[0], len(self.model_filters), self.n_sample_times)
        lcs = [x.to(device=t.device, dtype=torch.float64).contiguous() for x in lcs]
        if not lcs or any(tuple(x.shape) != shape for x in lcs):
            raise L.NMMAHipError(f"every curve set must be [B, M, NS] = {shape}")
        bad = None
        if bad_rows is not None:
            bad = bad_rows.to(device=t.device, dtype=torch.uint8).contiguous()
            if tuple(bad.shape) != (t.shape[0],):
                raise L.NMMAHipError("bad_rows must be [B]")
        out = torch.empty(t.shape[0], dtype=torch.float64, device=t.device)
        ptrs = (C.c_void_p * len(lcs))(*[C.c_void_p(x.data_ptr()) for x in lcs])
        L.check(self._lib.nmma_em_loglike_lc_sets(self._handle, C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0), ptrs, len(lcs),
                                                  C.c_void_p(bad.data_ptr()) if bad is not None else None, C.c_void_p(out.data_ptr()),
                                                  self._stream()), "nmma_em_loglike_lc_sets")
        return out

    def loglike_stack2(self, theta, lc2, bad_rows=None, out=None, stream=None, gap_free=False, completed=False):
        """logL of the COMBINED model {this engine's surrogate + a second transient} in one launch: ``lc2[B, M, NS]`` are the second
        transient's source-frame curves on this engine's sample_times and model filters; the flux sum is formed on the two nodes
        every datum interpolates between (``nmma_em_loglike_stack2``).  Returns None when the handle has no one-launch form (not
        created with ``stack_operands=1``, or a configuration outside it): the caller then materialises the surrogate's curves
        (``model_lightcurves``) and takes ``loglike_lc_sets`` on a likelihood-from-curves engine.  ``gap_free=True``: the caller guarantees
        that ``lc2`` has no non-finite node strictly inside the grid (e.g. afterglowpy curves: finite, or the row is in ``bad_rows``) --
        the re-evaluation launch is skipped; a row that breaks the promise poisons the engine (the next call raises).
        ``completed=True``: ``lc2`` came out of ``regrid`` (its non-finite nodes are leading / trailing only: no flux wherever they lie).
        After a None, ``stack2_reason`` says why the handle has no one-launch form."""
        import torch
        t = self._dev_theta(theta)
        shape = (t.shape[0], len(self.model_filters), self.n_sample_times)
        lc2 = lc2.to(device=t.device, dtype=torch.float64).contiguous()
        if tuple(lc2.shape) != shape:
            raise L.NMMAHipError(f"lc2 must be [B, M, NS] = {shape}, got {tuple(lc2.shape)}")
        bad = None
        if bad_rows is not None:
            bad = bad_rows.to(device=t.device, dtype=torch.uint8).contiguous()
            if tuple(bad.shape) != (t.shape[0],):
                raise L.NMMAHipError("bad_rows must be [B]")
        if out is None:
            out = torch.empty(t.shape[0], dtype=torch.float64, device=t.device)
        else:
            self._check_out(out, t.shape[0])
        status = self._lib.nmma_em_loglike_stack2(self._handle, C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0), C.c_void_p(lc2.data_ptr()),
                                                  C.c_void_p(bad.data_ptr()) if bad is not None else None, C.c_void_p(out.data_ptr()),
                                                  (L.STACK2_GAP_FREE if gap_free else 0) | (L.STACK2_COMPLETED if completed else 0),
                                                  self._stream(stream))
        if status == 2:
            self.stack2_reason = L.last_error()       # why the handle has no one-launch form
            return None
        L.check(status, "nmma_em_loglike_stack2")
        return out

    def stack(self, lcs):
        """Flux-add light-curve sets [B, M, NS] (CombinedLightCurveModelContainer.stack_magnitudes)."""
        import torch
        lcs = [x.to(dtype=torch.float64).contiguous() for x in lcs]
        out = torch.empty_like(lcs[0])
        ptrs = (C.c_void_p * len(lcs))(*[C.c_void_p(x.data_ptr()) for x in lcs])
        L.check(self._lib.nmma_lc_stack(self._handle, ptrs, len(lcs), lcs[0].shape[0], C.c_void_p(out.data_ptr()),
                                        self._stream()), "nmma_lc_stack")
        return out

    def regrid(self, lc, src_times, sources):
        """One sub-model's curves ``lc[B, Ms, NSs]`` (on ``src_times``) moved onto this engine's sample_times and model
        filters: ``sources[m]`` = list of 0-3 source-filter indices of output filter m (1: the filter itself, 2-3: mean of
        helper bands, 0: nothing -> +inf).  CombinedLightCurveModelContainer.gen_detector_lc, model.py:1434-1448."""
        import torch
        lc = lc.to(device=f"cuda:{self.device}", dtype=torch.float64).contiguous()
        if lc.dim() != 3 or lc.shape[2] != len(src_times):
            raise L.NMMAHipError(f"lc must be [B, Ms, {len(src_times)}], got {tuple(lc.shape)}")
        m = len(self.model_filters)
        if len(sources) != m:
            raise L.NMMAHipError("sources needs one entry per model filter of the engine")
        st = _f64(src_times)
        idx = np.zeros((m, L.MAX_SOURCES), dtype=np.int32)
        ns = np.zeros(m, dtype=np.int32)
        for i, src in enumerate(sources):
            ns[i] = len(src)
            idx[i, :len(src)] = src
        out = torch.empty((lc.shape[0], m, self.n_sample_times), dtype=torch.float64, device=lc.device)
        L.check(self._lib.nmma_lc_regrid(self._handle, C.c_void_p(lc.data_ptr()), lc.shape[1], lc.shape[2],
                                         _ptr(st, C.c_double), _ptr(idx, C.c_int32), _ptr(ns, C.c_int32), lc.shape[0],
                                         C.c_void_p(out.data_ptr()), self._stream()), "nmma_lc_regrid")
        return out

    def coefficients(self, theta):
        """SVD coefficients c[B, M, NC] (fp32), the surrogate output (lightcurve_generation.py:198)."""
        import torch
        t = self._dev_theta(theta)
        c = torch.empty((t.shape[0], len(self.model_filters), self.n_coeff), dtype=torch.float32, device=t.device)
        L.check(self._lib.nmma_em_coefficients(self._handle, C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0),
                                               C.c_void_p(c.data_ptr()), self._stream()), "nmma_em_coefficients")
        return c

    def debug_timeline(self, theta):
        """In-kernel shader-clock stamps of workgroup 0 (see nmma_em_debug_timeline)."""
        import torch
        t = self._dev_theta(theta)
        out = torch.empty(t.shape[0], dtype=torch.float64, device=t.device)
        torch.cuda.synchronize()
        stamps = (C.c_int64 * 512)()
        L.check(self._lib.nmma_em_debug_timeline(self._handle, C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0),
                                                 C.c_void_p(out.data_ptr()), stamps), "nmma_em_debug_timeline")
        return np.array(stamps[:], dtype=np.int64)

    def last_launch_geometry(self):
        v = [C.c_int32() for _ in range(5)]
        L.check(self._lib.nmma_em_last_launch_geometry(self._handle, *[C.byref(x) for x in v]),
                "nmma_em_last_launch_geometry")
        return dict(zip(("grid_x", "grid_y", "block", "tile_samples", "lds_bytes"), (x.value for x in v)))

    def profile_begin(self, max_launches):
        L.check(self._lib.nmma_em_profile_begin(self._handle, int(max_launches)), "nmma_em_profile_begin")

    def profile_end(self):
        f, c, n = C.c_double(), C.c_double(), C.c_int32()
        L.check(self._lib.nmma_em_profile_end(self._handle, C.byref(f), C.byref(c), C.byref(n)),
                "nmma_em_profile_end")
        return dict(fused_ms_total=f.value, combine_ms_total=c.value, n_launches=n.value)

    def set_option(self, name, value):
        """Run-time option of the handle (``nmma_em_set_option``): "walk_fuse", "walk_split" (0 / 1), "lc_group" (0, 16, 32, 64),
        "stack2_fixup" (0 / 1) -- for A/B measurements and tests."""
        L.check(self._lib.nmma_em_set_option(self._handle, str(name).encode(), int(value)), "nmma_em_set_option")

    def check(self):
        """Synchronise and raise if an earlier asynchronous launch failed (kernel watchdog)."""
        L.check(self._lib.nmma_em_check(self._handle), "nmma_em_check")

    def walk_queue(self, table, live, u0, loglstar, keys, walks, constraints=None, first_step=1, stream=None):
        """One queue of the nested sampler in ONE library call (``nmma_em_walk_queue``): ``table`` = ``device_prior_table``'s array of
        ``WalkPrior``, ``live[n_live, D]`` the unit-cube live points, ``u0[n, D]`` the chains' start points, ``loglstar`` (scalar or
        [n]) their likelihood bounds, ``keys[n]`` their random-number keys, ``walks`` (int or [n]) their lengths, ``constraints`` a
        :class:`nmma_amd.core.constraints.ConstraintProgram` or None.  Returns ``(u, v, logl, counts[n, 4])`` as numpy arrays."""
        return self.walk_queue_end(self.walk_queue_begin(table, live, u0, loglstar, keys, walks, constraints, first_step, stream))

    def walk_queue_begin(self, table, live, u0, loglstar, keys, walks, constraints=None, first_step=1, stream=None, records=None):
        """First half of :meth:`walk_queue` (``nmma_em_walk_queue_begin``): packs, uploads and enqueues the whole queue on this engine's
        device and returns a token WITHOUT waiting -- a queue sharded over several devices is begun on every engine, then collected
        with :meth:`walk_queue_end` (``GPUPool(devices=[...])``).  One queue in flight per engine.
        ``records``: a float64 CUDA tensor on this engine's device with at least ``n`` rows of ``2 D + 3`` -- the records then STAY on the
        device, packed ``u | v | logl | counts`` per row (``nmma_walk_queue::records_dev``; :func:`nmma_amd.parallel.unpack_records` reads them), instead
        of being downloaded: the send buffer of a queue sharded over ranks (``parallel.ShardedQueue``); ``walk_queue_end`` returns None."""
        import torch
        live, u0 = _f64(live), _f64(u0)
        n, ndim = u0.shape
        if len(table) != ndim or live.ndim != 2 or live.shape[1] != ndim or ndim != len(self.parameter_names):
            raise L.NMMAHipError(f"walk_queue: {ndim} sampled dimensions, a prior table of {len(table)}, live points {live.shape}, "
                                 f"the engine's {len(self.parameter_names)} columns")
        star = np.ascontiguousarray(np.broadcast_to(np.asarray(loglstar, dtype=np.float64), (n,)))
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        per_chain = np.ndim(walks) > 0
        wl = np.ascontiguousarray(walks, dtype=np.int32) if per_chain else None
        if keys.shape != (n,) or (per_chain and wl.shape != (n,)):
            raise L.NMMAHipError("walk_queue: one key (and walk length) per chain")
        if constraints is not None and constraints.handle is None:
            # (a closed program would silently run the walk unconstrained)
            raise L.NMMAHipError("walk_queue: the constraint program has been closed")
        if getattr(self, "_walk_ws", None) is None:
            ws = C.c_void_p()
            L.check(self._lib.nmma_walk_ws_create(self.device, C.byref(ws)), "nmma_walk_ws_create")
            self._walk_ws = ws
        if records is not None:
            if (not isinstance(records, torch.Tensor) or not records.is_cuda or records.device.index != self.device or records.dtype != torch.float64
                    or records.dim() != 2 or records.shape[1] != 2 * ndim + 3 or records.shape[0] < n or not records.is_contiguous()):
                raise L.NMMAHipError(f"walk_queue: records must be a contiguous float64 tensor [>= {n}, {2 * ndim + 3}] on cuda:{self.device}")
            outs = None
        else:
            outs = (np.empty((n, ndim)), np.empty((n, ndim)), np.empty(n), np.empty((n, 4), dtype=np.int32))
        q = L.WalkQueue()
        q.priors, q.ndim, q.walks = table, ndim, 0 if per_chain else int(walks)
        q.live, q.n_live, q.u0, q.loglstar, q.key = live.ctypes.data, live.shape[0], u0.ctypes.data, star.ctypes.data, keys.ctypes.data
        q.walks_per_chain = wl.ctypes.data if per_chain else None
        q.n, q.first_step = n, int(first_step)
        q.constraints = constraints.handle if constraints is not None else None
        if outs is not None:
            q.u, q.v, q.logl, q.counts = (a.ctypes.data for a in outs)
        else:
            q.records_dev = records.data_ptr()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        L.check(self._lib.nmma_em_walk_queue_begin(self._handle, self._walk_ws, C.byref(q), C.c_void_p(s.cuda_stream)), "nmma_em_walk_queue_begin")
        # (everything the library reads until `end` stays referenced by the token)
        return (q, outs, (table, live, u0, star, keys, wl, constraints, s, records))

    def walk_queue_end(self, token):
        q, outs, _keep = token
        L.check(self._lib.nmma_em_walk_queue_end(self._handle, self._walk_ws, C.byref(q)), "nmma_em_walk_queue_end")
        self.last_walk_gpu_ms = float(q.gpu_ms)
        return outs

    def close(self):
        if getattr(self, "_walk_ws", None):
            self._lib.nmma_walk_ws_destroy(self._walk_ws)
            self._walk_ws = None
        if getattr(self, "_handle", None):
            self._lib.nmma_em_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
