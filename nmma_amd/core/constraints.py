"""Constraint priors on the device: lowering ``evaluate_constraints(parameter_conversion(sample))`` to a postfix program.

The reference evaluates Constraint priors per sample on the CONVERTED parameters (``nmma/core/base.py:51-82``: ``constraints``
is the dict of ``bilby.core.prior.Constraint`` entries of the prior, ``evaluate_constraints`` the product of their ``prob`` on the
output of ``parameter_conversion``; a zero product means the floor).  The conversion chain -- ``observation_angle_conversion``
(core/conversion.py:119-126), the model's ``log10_`` aliases (em/model.py:272-286), ``convert_mtot_mni`` (core/conversion.py:184-192),
bilby's mass conversions behind ``bbh_source_frame`` / ``bns_source_frame`` (:104-139) -- is elementwise arithmetic on the sampled
columns.  So instead of re-stating each derived quantity by hand, :func:`trace_constraints` runs the likelihood's OWN conversion
functions once on symbolic columns (:class:`Sym` records the arithmetic numpy would have done), takes the expressions of the
constrained keys and compiles ``minimum < expr < maximum`` into the operations of ``include/nmma_hip.h`` (``nmma_con_op``).  A
kernel then evaluates the program per row of ``theta`` (``nmma_con_floor``; fused into the accept step of the device walk), which
removes the device-to-host copy of ``theta`` a batched call with constraints used to make.

What cannot be traced -- a conversion that branches on VALUES, needs a root-find (redshift from distance) or calls into
non-ufunc numpy -- yields an *opaque* expression; a constraint on an opaque expression makes the whole set "not lowerable" and the
caller keeps the host mask (numpy on columns), as before.  ``evaluate_program`` is a numpy interpreter of the same program, used
by the CPU tests to pin the lowering against the conversion functions themselves.
"""
from __future__ import annotations

import numpy as np

from .. import _lib as L


class TraceError(Exception):
    """A conversion asked a symbolic column for its VALUE (comparison, bool, float(), array conversion)."""


_BINARY = {np.add: L.CON_ADD, np.subtract: L.CON_SUB, np.multiply: L.CON_MUL, np.divide: L.CON_DIV, np.true_divide: L.CON_DIV,
           np.power: L.CON_POW, np.float_power: L.CON_POW, np.minimum: L.CON_MIN, np.maximum: L.CON_MAX, np.fmin: L.CON_MIN,
           np.fmax: L.CON_MAX}
_UNARY = {np.negative: L.CON_NEG, np.absolute: L.CON_ABS, np.fabs: L.CON_ABS, np.sqrt: L.CON_SQRT, np.log10: L.CON_LOG10, np.log: L.CON_LOG,
          np.exp: L.CON_EXP, np.sin: L.CON_SIN, np.cos: L.CON_COS, np.arccos: L.CON_ACOS, np.arcsin: L.CON_ASIN, np.sign: L.CON_SIGN}
_OPAQUE = -1


class Sym:
    """One node of an expression over the sampled columns.  Behaves like a number under numpy ufuncs and Python arithmetic."""
    __slots__ = ("op", "args", "col", "value", "is_opaque", "size")
    __array_priority__ = 1000.0

    def __init__(self, op, args=(), col=-1, value=0.0):
        self.op, self.args, self.col, self.value = op, tuple(args), int(col), float(value)
        self.is_opaque = op == _OPAQUE or any(a.is_opaque for a in self.args)
        # operations of the expression written out as a tree (shared sub-expressions count every time they are used): kept so
        # that an expression that outgrows the device program is refused before anybody walks it
        self.size = min(1 + sum(a.size for a in self.args), 1 << 30)

    @classmethod
    def column(cls, col):
        return cls(L.CON_PUSH_COL, col=col)

    @classmethod
    def const(cls, value):
        return cls(L.CON_PUSH_CONST, value=value)

    @classmethod
    def opaque(cls):
        return cls(_OPAQUE)

    @staticmethod
    def wrap(x):
        if isinstance(x, Sym):
            return x
        a = np.asarray(x)
        if a.dtype == object or a.size != 1:
            raise TraceError("a symbolic column met a non-scalar operand")
        return Sym.const(float(a.reshape(())))

    # ---- numpy protocol
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            return Sym.opaque()
        if ufunc in _BINARY and len(inputs) == 2:
            return Sym(_BINARY[ufunc], (Sym.wrap(inputs[0]), Sym.wrap(inputs[1])))
        if ufunc in _UNARY and len(inputs) == 1:
            return Sym(_UNARY[ufunc], (Sym.wrap(inputs[0]),))
        if ufunc is np.square and len(inputs) == 1:
            return Sym(L.CON_MUL, (Sym.wrap(inputs[0]), Sym.wrap(inputs[0])))
        if ufunc is np.reciprocal and len(inputs) == 1:
            return Sym(L.CON_DIV, (Sym.const(1.0), Sym.wrap(inputs[0])))
        if ufunc is np.positive and len(inputs) == 1:
            return Sym.wrap(inputs[0])
        if ufunc in (np.less, np.less_equal, np.greater, np.greater_equal, np.equal, np.not_equal, np.isfinite, np.isnan):
            raise TraceError(f"a conversion compares symbolic values ({ufunc.__name__})")
        return Sym.opaque()                   # (a ufunc without a device counterpart: whatever depends on it cannot be constrained on the device)

    def __array__(self, *args, **kwargs):
        raise TraceError("a conversion turned a symbolic column into an array")

    # ---- Python arithmetic
    def __add__(self, o): return Sym(L.CON_ADD, (self, Sym.wrap(o)))
    def __radd__(self, o): return Sym(L.CON_ADD, (Sym.wrap(o), self))
    def __sub__(self, o): return Sym(L.CON_SUB, (self, Sym.wrap(o)))
    def __rsub__(self, o): return Sym(L.CON_SUB, (Sym.wrap(o), self))
    def __mul__(self, o): return Sym(L.CON_MUL, (self, Sym.wrap(o)))
    def __rmul__(self, o): return Sym(L.CON_MUL, (Sym.wrap(o), self))
    def __truediv__(self, o): return Sym(L.CON_DIV, (self, Sym.wrap(o)))
    def __rtruediv__(self, o): return Sym(L.CON_DIV, (Sym.wrap(o), self))
    def __pow__(self, o): return Sym(L.CON_POW, (self, Sym.wrap(o)))
    def __rpow__(self, o): return Sym(L.CON_POW, (Sym.wrap(o), self))
    def __neg__(self): return Sym(L.CON_NEG, (self,))
    def __pos__(self): return self
    def __abs__(self): return Sym(L.CON_ABS, (self,))

    def _value_needed(self, *a, **k):
        raise TraceError("a conversion branches on the VALUE of a sampled parameter")

    __bool__ = __float__ = __int__ = __index__ = __lt__ = __le__ = __gt__ = __ge__ = __len__ = __iter__ = _value_needed
    __hash__ = object.__hash__

    def __eq__(self, o):
        raise TraceError("a conversion compares symbolic values")

    def __ne__(self, o):
        raise TraceError("a conversion compares symbolic values")

    def emit(self, out):
        """Append this expression's postfix operations to ``out`` (list of (op, col, value))."""
        if self.op == _OPAQUE:
            raise TraceError("opaque expression")
        for a in self.args:
            a.emit(out)
        out.append((self.op, self.col, self.value))


def is_symbolic(x):
    return isinstance(x, Sym)


def opaque_like(x):
    """What a conversion returns for a quantity it cannot express as arithmetic on a symbolic input (root-finds, table look-ups)."""
    return Sym.opaque()


def trace_constraints(constraints, names, fixed, conversions):
    """The postfix program of ``constraints`` (name -> Constraint) over the columns ``names``: list of ``(op, col, value)``, or
    ``None`` when the set cannot be lowered (a constrained key that no conversion derives by traceable arithmetic).
    ``conversions``: the callables ``parameter_conversion`` applies, in application order; ``fixed``: name -> number."""
    if not constraints:
        return []
    sample = {n: Sym.column(i) for i, n in enumerate(names)}
    for key, val in (fixed or {}).items():
        sample.setdefault(key, float(val))
    try:
        for conv in conversions:
            out = conv(sample)
            sample = out[0] if isinstance(out, tuple) else out
        prog = []
        for key, con in constraints.items():
            if key not in sample:
                return None
            expr = sample[key]
            if not isinstance(expr, Sym):
                a = np.asarray(expr)
                if a.dtype == object:          # (np.array(Sym): a 0-d object array wrapping the expression)
                    expr = a.reshape(-1)[0] if a.size == 1 else None
                    if not isinstance(expr, Sym):
                        return None
                else:
                    expr = Sym.const(float(a.reshape(())))           # a constraint on a FIXED quantity: constant program
            if expr.is_opaque or expr.size + len(prog) + 2 > L.CON_MAX_OPS:
                return None
            expr.emit(prog)
            lo = -np.inf if con.minimum is None else float(con.minimum)
            hi = np.inf if con.maximum is None else float(con.maximum)
            prog.append((L.CON_CHECK_GT, -1, lo))
            prog.append((L.CON_CHECK_LT, -1, hi))
    except TraceError:
        return None
    except (TypeError, ValueError, KeyError, AttributeError, IndexError):
        return None
    if len(prog) > L.CON_MAX_OPS or _max_depth(prog) > L.CON_MAX_STACK:
        return None
    return prog


def _max_depth(prog):
    sp = deepest = 0
    for op, _, _ in prog:
        if op in (L.CON_PUSH_COL, L.CON_PUSH_CONST):
            sp += 1
        elif L.CON_ADD <= op <= L.CON_MAX or op == L.CON_CHECK_LT:
            sp -= 1
        deepest = max(deepest, sp)
    return deepest


def evaluate_program(prog, theta):
    """numpy interpreter of a constraint program: ``ok[B]`` for ``theta[B, D]`` -- the arithmetic of ``con_row_ok``
    (csrc/walk_kernels.hip), used by the CPU tests and by nothing on the product path."""
    theta = np.atleast_2d(np.asarray(theta, dtype=float))
    ok = np.ones(len(theta), dtype=bool)
    st = []
    two = {L.CON_ADD: np.add, L.CON_SUB: np.subtract, L.CON_MUL: np.multiply, L.CON_DIV: np.divide, L.CON_POW: np.power,
           L.CON_MIN: np.fmin, L.CON_MAX: np.fmax}
    one = {L.CON_NEG: np.negative, L.CON_ABS: np.abs, L.CON_SQRT: np.sqrt, L.CON_LOG10: np.log10, L.CON_LOG: np.log, L.CON_EXP: np.exp,
           L.CON_SIN: np.sin, L.CON_COS: np.cos, L.CON_ACOS: np.arccos, L.CON_ASIN: np.arcsin, L.CON_SIGN: np.sign}
    with np.errstate(all="ignore"):
        for op, col, value in prog:
            if op == L.CON_PUSH_COL:
                st.append(theta[:, col])
            elif op == L.CON_PUSH_CONST:
                st.append(np.full(len(theta), value))
            elif op in two:
                b = st.pop(); a = st.pop()
                st.append(two[op](a, b))
            elif op in one:
                st.append(one[op](st.pop()))
            elif op == L.CON_CHECK_GT:
                ok &= st[-1] > value
            elif op == L.CON_CHECK_LT:
                ok &= st.pop() < value
            else:
                raise ValueError(f"unknown constraint operation {op}")
    return ok


class ConstraintProgram:
    """A lowered constraint set on one device (``nmma_con_program``)."""

    def __init__(self, prog, n_cols, device=0):
        import ctypes as C
        self._lib = L.load_library()
        self.ops = list(prog)
        self.n_cols, self.device = int(n_cols), int(device)
        arr = (L.ConOp * len(prog))(*[L.ConOp(int(o), int(c), float(v)) for o, c, v in prog])
        h = C.c_void_p()
        L.check(self._lib.nmma_con_create(arr, len(prog), self.n_cols, self.device, C.byref(h)), "nmma_con_create")
        self.handle = h

    def floor(self, theta, logl, stream=None):
        """``logl[b] = floor`` where row b of the CUDA tensor ``theta[B, >= n_cols]`` fails a check (in place, asynchronous)."""
        import ctypes as C
        import torch
        if not (theta.is_cuda and logl.is_cuda and theta.dtype == torch.float64 and logl.dtype == torch.float64
                and theta.is_contiguous() and logl.is_contiguous() and theta.device.index == self.device):
            raise L.NMMAHipError("ConstraintProgram.floor: contiguous float64 CUDA tensors on the program's device")
        s = stream if stream is not None else torch.cuda.current_stream(theta.device)
        L.check(self._lib.nmma_con_floor(self.handle, C.c_void_p(theta.data_ptr()), theta.shape[0], theta.stride(0),
                                         C.c_void_p(logl.data_ptr()), C.c_void_p(s.cuda_stream)), "nmma_con_floor")
        return logl

    def close(self):
        if getattr(self, "handle", None):
            self._lib.nmma_con_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def constraint_signature(constraints):
    """Hashable description of a constraint set (the cache key of its lowered program: the set may be edited in place)."""
    return tuple((k, None if c.minimum is None else float(c.minimum), None if c.maximum is None else float(c.maximum))
                 for k, c in constraints.items())
