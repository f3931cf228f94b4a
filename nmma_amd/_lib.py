"""ctypes binding of ``libnmma_hip.so`` (C ABI: ``include/nmma_hip.h``).

There is NO CPU fallback: if the shared library is missing or fails to load, or if
no HIP device is present when a handle is created, the product path raises
:class:`NMMAHipError`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NMMA_HIP_LIB") or os.path.join(_HERE, "libnmma_hip.so")   # env: experiment builds
SRC_PATH = os.path.join(_HERE, "csrc", "em_kernels.hip")
#: translation units of libnmma_hip.so, longest first (they compile concurrently; build_library caps the number in flight).  em_logl's
#: 76 instantiations (round 5: 100) are spread over the em_logl_*.hip units, one or two task flavours each: as one unit round 4's 38
#: took 170 s; em_lc.hip holds the fourteen variants of the likelihood-from-curves kernel (inside em_kernels.hip they made it the
#: critical path).  Order = this machine's unit times, longest first (tools: _lib.UNIT_SECONDS after a forced build).
SOURCES = ("em_kernels.hip", "em_lc64.hip", "em_lc.hip", "em_logl_w3.hip", "em_logl_w2.hip", "em_logl_wc1.hip", "em_logl_w1.hip", "em_logl_f5.hip", "em_logl_f02.hip",
           "em_logl_w4.hip", "em_logl_wc2.hip", "em_logl_w5.hip", "em_logl_f8.hip", "em_logl_f6.hip", "em_logl_f9.hip", "em_logl_f4.hip", "em_logl_f3.hip", "em_logl_f7.hip",
           "gw_kernels.hip", "em_logl_f1.hip", "walk_kernels.hip")
#: per-unit flags after the common ones.  The EM unit keeps -ffp-contract=off (the reference's numpy expressions are not fused and
#: the parity tests compare bit patterns of intermediate results); the GW unit has no bit-level counterpart (its reference
#: arithmetic is third-party and absent) and lets hipcc fuse multiply-adds: a quarter fewer instructions in the bin loop.
UNIT_FLAGS = {"gw_kernels.hip": ["-ffp-contract=fast"]}

#: seconds each unit took in this process's last build_library call (tools: order SOURCES longest first)
UNIT_SECONDS = {}

ABI_VERSION = 8
STACK2_GAP_FREE = 1
STACK2_COMPLETED = 2
MAX_PARAMS = 8
MAX_COEFF = 16
MAX_SOURCES = 3
LOGL_FLOOR = -1.7976931348623157e308

OP_IDENT, OP_RAD2DEG, OP_DEG2RAD, OP_LOG10, OP_POW10, OP_THETAJN2DEG, OP_COSTHETAJN2DEG, OP_ACOS = range(8)
GW_MAX_IFO = 4
GW_CHIRP_MASS_RATIO, GW_COMPONENT_MASSES = range(2)
Z_ZERO, Z_SLOT, Z_GRID = range(3)
SYS_CONST, SYS_PARAM, SYS_NODES = range(3)
MODEL_SVD, MODEL_ME2017, MODEL_EXTERNAL = range(3)
EXT_LINEAR, EXT_P92_SMC_HOST = range(2)


class NMMAHipError(RuntimeError):
    """Raised for every failure of the HIP path (missing library, HIP error, bad config)."""


class Slot(C.Structure):
    _fields_ = [("col", C.c_int32), ("op", C.c_int32), ("value", C.c_double)]

    @classmethod
    def column(cls, col, op=OP_IDENT):
        return cls(int(col), int(op), 0.0)

    @classmethod
    def constant(cls, value):
        return cls(-1, OP_IDENT, float(value))


_pf = C.POINTER(C.c_float)
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)


class EmConfig(C.Structure):
    """Mirror of ``struct nmma_em_config`` (field order must match the header)."""
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32),
        ("model_kind", C.c_int32), ("filter_nu0", _pd),
        ("n_model_filters", C.c_int32), ("n_params", C.c_int32), ("n_hidden", C.c_int32),
        ("n_coeff", C.c_int32), ("n_tt", C.c_int32),
        ("W1", _pf), ("b1", _pf), ("W2", _pf), ("b2", _pf),
        ("VA", _pd), ("mins", _pd), ("maxs", _pd), ("tt", _pd),
        ("param_mins", _pd), ("param_maxs", _pd),
        ("n_sample_times", C.c_int32), ("sample_times", _pd),
        ("redshift_mode", C.c_int32), ("n_cosmo", C.c_int32), ("dist_grid", _pd), ("z_grid", _pd),
        ("n_dim", C.c_int32),
        ("model_param", Slot * MAX_PARAMS),
        ("luminosity_distance", Slot), ("redshift", Slot), ("timeshift", Slot), ("ebv", Slot),
        ("hubble_constant", Slot), ("hubble_reference", C.c_double),
        ("ebv_coeff", _pd), ("extinction_law", C.c_int32),
        ("n_obs_filters", C.c_int32), ("data_offsets", _pi),
        ("data_times", _pd), ("data_mags", _pd), ("data_sigmas", _pd),
        ("detection_limit", _pd), ("n_sources", _pi), ("sources", _pi),
        ("sys_kind", _pi), ("sys_const", _pd), ("sys_n_nodes", _pi), ("sys_slot_offsets", _pi),
        ("sys_slots", C.POINTER(Slot)), ("sys_node_times", _pd),
        ("stack_operands", C.c_int32), ("n_base_times", C.c_int32), ("base_times", _pd), ("null_filters", _pi),
    ]


class WalkPrior(C.Structure):
    """Mirror of ``struct nmma_walk_prior``."""
    _fields_ = [("kind", C.c_int32), ("boundary", C.c_int32), ("a", C.c_double), ("b", C.c_double), ("alpha", C.c_double), ("c", C.c_double)]


WALK_MAX_DIM = 32
(PRIOR_UNIFORM, PRIOR_SINE, PRIOR_COSINE, PRIOR_POWERLAW, PRIOR_GAUSSIAN, PRIOR_DELTA, PRIOR_TRUNC_GAUSSIAN, PRIOR_LOGNORMAL,
 PRIOR_HALF_GAUSSIAN) = range(9)
BOUNDARY_NONE, BOUNDARY_PERIODIC, BOUNDARY_REFLECTIVE = range(3)


class ConOp(C.Structure):
    """Mirror of ``struct nmma_con_op``: one operation of a constraint program (postfix)."""
    _fields_ = [("op", C.c_int32), ("col", C.c_int32), ("value", C.c_double)]


(CON_PUSH_COL, CON_PUSH_CONST, CON_ADD, CON_SUB, CON_MUL, CON_DIV, CON_POW, CON_MIN, CON_MAX, CON_NEG, CON_ABS, CON_SQRT, CON_LOG10,
 CON_LOG, CON_EXP, CON_SIN, CON_COS, CON_ACOS, CON_ASIN, CON_SIGN, CON_CHECK_GT, CON_CHECK_LT) = range(22)
CON_MAX_OPS, CON_MAX_STACK = 256, 16


class WalkQueue(C.Structure):
    """Mirror of ``struct nmma_walk_queue`` (field order must match the header)."""
    _fields_ = [
        ("priors", C.POINTER(WalkPrior)), ("ndim", C.c_int32), ("walks", C.c_int32),
        ("live", C.c_void_p), ("n_live", C.c_int64), ("u0", C.c_void_p), ("loglstar", C.c_void_p), ("key", C.c_void_p),
        ("walks_per_chain", C.c_void_p), ("n", C.c_int64), ("first_step", C.c_uint64), ("constraints", C.c_void_p),
        ("u", C.c_void_p), ("v", C.c_void_p), ("logl", C.c_void_p), ("counts", C.c_void_p), ("gpu_ms", C.c_double),
        ("records_dev", C.c_void_p),
    ]


class GwConfig(C.Structure):
    """Mirror of ``struct nmma_gw_config`` (field order must match the header)."""
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("n_ifo", C.c_int32), ("tidal", C.c_int32),
        ("n_freq", C.c_int64), ("duration", C.c_double), ("start_time", C.c_double),
        ("data", _pd), ("psd", _pd), ("mask", C.POINTER(C.c_uint8)), ("detector_tensor", _pd), ("vertex", _pd),
        ("gmst_ref_time", C.c_double), ("gmst_ref", C.c_double), ("gmst_rate", C.c_double),
        ("reference_frequency", C.c_double), ("waveform_minimum_frequency", C.c_double),
        ("waveform_maximum_frequency", C.c_double),
        ("phase_marginalization", C.c_int32), ("mass_mode", C.c_int32), ("n_dim", C.c_int32),
        ("mass_a", Slot), ("mass_b", Slot), ("chi_1", Slot), ("chi_2", Slot), ("lambda_1", Slot), ("lambda_2", Slot),
        ("luminosity_distance", Slot), ("theta_jn", Slot), ("phase", Slot), ("ra", Slot), ("dec", Slot), ("psi", Slot),
        ("geocent_time", Slot),
        ("n_distance", C.c_int32), ("pad_distance", C.c_int32), ("distance_grid", _pd), ("distance_log_weight", _pd),
        ("time_log_weight", _pd),
        ("time_jitter", Slot), ("time_prior_minimum", C.c_double), ("time_prior_maximum", C.c_double),
    ]


#: name -> (restype, argtypes); every symbol ``include/nmma_hip.h`` declares
PROTOTYPES = {
    "nmma_abi_version": (C.c_int32, []),
    "nmma_build_info": (C.c_char_p, []),
    "nmma_last_error": (C.c_char_p, []),
    "nmma_em_create": (C.c_int32, [C.POINTER(EmConfig), C.POINTER(C.c_void_p)]),
    "nmma_em_destroy": (None, [C.c_void_p]),
    "nmma_em_loglike": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    # (host pointers as integers: building a typed ctypes pointer from a numpy array costs ~3 us, a fifth of a single-point call)
    "nmma_em_loglike_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "nmma_em_loglike_parts": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "nmma_em_lightcurves": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "nmma_em_model_lightcurves": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "nmma_em_loglike_lc": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "nmma_em_loglike_lc_sets": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "nmma_em_loglike_stack2": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "nmma_lc_stack": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "nmma_lc_regrid": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, _pd, _pi, _pi, C.c_int64, C.c_void_p,
                                   C.c_void_p]),
    "nmma_gw_loglike_ratio": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_double,
                                         C.c_void_p, C.c_int32, C.c_void_p]),
    "nmma_logl_sum_floor": (C.c_int32, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "nmma_walk_propose": (C.c_int32, [C.POINTER(WalkPrior), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "nmma_walk_accept": (C.c_int32, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p]),
    "nmma_walk_step": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p]),
    "nmma_walk_step_rwalk": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                         C.c_double, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p]),
    "nmma_walk_accept_rwalk": (C.c_int32, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_int32, C.c_double, C.c_double,
                                           C.c_int32, C.c_void_p]),
    "nmma_walk_rescale": (C.c_int32, [C.POINTER(WalkPrior), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "nmma_con_create": (C.c_int32, [C.POINTER(ConOp), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "nmma_con_destroy": (None, [C.c_void_p]),
    "nmma_con_floor": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "nmma_em_loglike_walk": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p]),
    "nmma_em_set_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int32]),
    "nmma_walk_ws_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "nmma_walk_ws_destroy": (None, [C.c_void_p]),
    "nmma_em_walk_queue": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(WalkQueue), C.c_void_p]),
    "nmma_em_walk_queue_begin": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(WalkQueue), C.c_void_p]),
    "nmma_em_walk_queue_end": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(WalkQueue)]),
    "nmma_gw_create": (C.c_int32, [C.POINTER(GwConfig), C.POINTER(C.c_void_p)]),
    "nmma_gw_destroy": (None, [C.c_void_p]),
    "nmma_gw_loglike": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "nmma_gw_noise_log_likelihood": (C.c_double, [C.c_void_p]),
    "nmma_gw_strain": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "nmma_gw_inner_products": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "nmma_gw_n_bins": (C.c_int64, [C.c_void_p]),
    "nmma_gw_profile_begin": (C.c_int32, [C.c_void_p, C.c_int32]),
    "nmma_gw_profile_end": (C.c_int32, [C.c_void_p, _pd, _pi]),
    "nmma_em_coefficients": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_void_p]),
    "nmma_em_check": (C.c_int32, [C.c_void_p]),
    "nmma_em_debug_timeline": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                           C.POINTER(C.c_int64)]),
    "nmma_em_n_sample_times": (C.c_int32, [C.c_void_p]),
    "nmma_em_device": (C.c_int32, [C.c_void_p]),
    "nmma_em_flops_per_eval": (C.c_int64, [C.c_void_p]),
    "nmma_em_last_launch_geometry": (C.c_int32, [C.c_void_p] + [_pi] * 5),
    "nmma_em_profile_begin": (C.c_int32, [C.c_void_p, C.c_int32]),
    "nmma_em_profile_end": (C.c_int32, [C.c_void_p, _pd, _pd, _pi]),
}

_lib = None


HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-comment"]


def _dependencies():
    """Every file a translation unit may include: all of csrc/ (sources, .inc, .h) and the public header."""
    csrc = os.path.join(_HERE, "csrc")
    deps = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".h", ".inc"))]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "nmma_hip.h"))
    return deps


def build_library(force=False, extra_flags=()):
    """Compile the translation units under ``csrc/`` for gfx950 and link them into ``libnmma_hip.so`` (in-tree).
    A unit is recompiled when it or ANY header / .inc under ``csrc/`` or ``include/`` is newer than its object."""
    csrc = os.path.join(_HERE, "csrc")
    objdir = os.path.join(csrc, "build")
    os.makedirs(objdir, exist_ok=True)
    deps_mtime = max(os.path.getmtime(d) for d in _dependencies())
    objs, todo, rebuilt = [], [], False
    for name in SOURCES:
        src = os.path.join(csrc, name)
        obj = os.path.join(objdir, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or extra_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), deps_mtime):
            todo.append((name, ["hipcc", *HIPCC_FLAGS, *UNIT_FLAGS.get(name, []), "-c", src, "-o", obj, *extra_flags]))
            rebuilt = True
    # the units compile concurrently, at most one per core, in SOURCES order (longest first)
    import time
    slots = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    running = []
    while todo or running:
        while todo and len(running) < slots:
            name, cmd = todo.pop(0)
            running.append((name, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True), time.time()))
        for item in list(running):
            name, proc, t0 = item
            if proc.poll() is not None:
                running.remove(item)
                UNIT_SECONDS[name] = round(time.time() - t0, 1)
                if proc.returncode != 0:
                    err = proc.stderr.read()
                    for _, other, _ in running:
                        other.kill()
                    raise NMMAHipError(f"hipcc failed on {name}:\n" + err[-4000:])
        if running:
            time.sleep(0.05)
    if rebuilt or not os.path.exists(LIB_PATH) or any(os.path.getmtime(LIB_PATH) < os.path.getmtime(o) for o in objs):
        proc = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH],
                              capture_output=True, text=True)
        if proc.returncode != 0:
            raise NMMAHipError("hipcc link failed:\n" + proc.stderr[-4000:])
    return LIB_PATH


def load_library():
    """Load the HIP library (after torch, so both share one HIP runtime) and bind prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NMMAHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or tools/build_lib.sh.  nmma_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads libamdhip64 first; our .so then resolves to the same runtime)
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as exc:
        raise NMMAHipError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise NMMAHipError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.nmma_abi_version() != ABI_VERSION:
        raise NMMAHipError("libnmma_hip ABI version mismatch")
    _lib = lib
    return lib


def last_error():
    lib = load_library()
    msg = lib.nmma_last_error()
    return msg.decode() if msg else ""


def check(status, what):
    if status != 0:
        raise NMMAHipError(f"{what}: {last_error()}")
