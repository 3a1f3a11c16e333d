"""ORACLE (test infrastructure, never shipped): run an unmodified ``Ex_*.py`` file of the reference on plain NumPy numbers.

The reference star-imports an example over its defaults (``MPC_code.py:23-28``) and hands the example's Python functions
(``User_fxm_Cont``, ``User_fobj_Cont``, ``User_fobj_mhe`` ...) to CasADi, which evaluates and differentiates them symbolically.
The oracle needs the same functions as *numbers in, numbers out* - independent of the product's tracer
(``mpc-code_amd/symtrace.py``) and of its loader (``exfile.py``): this module executes the file with the handful of CasADi names
an example body mentions bound to NumPy (``vertcat`` stacks, ``mtimes`` is ``@``, ``SX.sym`` is a shape, ``xQx`` of
``Utilities.py:247-265`` is ``x'Qx``), so that every user function accepts real or COMPLEX arrays - derivatives in the oracle are
complex-step differences of these very functions, accurate to rounding and sharing nothing with the product's symbolic ones.

Nothing in ``mpc-code_amd/`` imports this file; it imports nothing from there.
"""
from __future__ import annotations

import math
import os
import sys
import types

import numpy as np

# reference Default_Values.py:16-131, the names the examples and MPC_code.py's probes rely on (restated as data)
DEFAULTS = dict(
    estimating=False, ssjacid=False, StateFeedback=False, Fp_nominal=False, offree="no",
    umin=None, umax=None, xmin=None, xmax=None, ymin=None, ymax=None,
    umin_ss=None, umax_ss=None, xmin_ss=None, xmax_ss=None, ymin_ss=None, ymax_ss=None,
    umin_dyn=None, umax_dyn=None, xmin_dyn=None, xmax_dyn=None, ymin_dyn=None, ymax_dyn=None,
    dmin=None, dmax=None, Dumin=None, Dumax=None, wmin=None, wmax=None, vmin=None, vmax=None,
    QForm_ss=False, DUssForm=False, Adaptation=False, ContForm=False, TermCons=False, QForm=False, DUForm=False, DUFormEcon=False,
    Sol_itmax=100, kalss=False, lue=False, kal=False, ekf=False, mhe=False, Collocation=False, LinPar=True, slacks=False,
)


class NumMat:
    """A small dense matrix of NUMBERS with the part of ``casadi.SX``'s surface the examples use inside their functions: ``SX(n, m)``
    zeros, element and slice reads (a slice is a COPY, as in CasADi), element and slice assignment - an argument clamped in place
    stays clamped for the rest of the function, as in CasADi -, ``.T``, ``size1()``, arithmetic, ``mtimes``.  A column indexed with
    one integer gives the number itself."""
    __array_priority__ = 1000

    def __init__(self, a):
        a = np.array(a)
        self.a = a.reshape(-1, 1) if a.ndim < 2 else a

    zeros = classmethod(lambda cls, n, m=1: cls(np.zeros((int(n), int(m)))))
    col = classmethod(lambda cls, items: cls(np.asarray(items).reshape(-1, 1)))
    shape = property(lambda self: self.a.shape)
    T = property(lambda self: NumMat(self.a.T.copy()))
    size1 = lambda self: self.a.shape[0]
    size2 = lambda self: self.a.shape[1]
    __len__ = lambda self: self.a.shape[0]

    def __iter__(self):
        return iter(self.a.ravel()) if self.a.shape[1] == 1 else iter(NumMat(r[None].copy()) for r in self.a)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)) and self.a.shape[1] == 1:
            return self.a[idx, 0]
        r = self.a[idx]
        return NumMat(np.array(r)) if isinstance(r, np.ndarray) else r

    def __setitem__(self, idx, v):
        v = v.a if isinstance(v, NumMat) else np.asarray(v)
        if v.dtype.kind == "c" and self.a.dtype.kind != "c":
            self.a = self.a.astype(complex)
        if isinstance(idx, (int, np.integer)) and self.a.shape[1] == 1:
            self.a[idx, 0] = v.ravel()[0] if v.ndim else v
        else:
            tgt = self.a[idx]
            self.a[idx] = v.reshape(tgt.shape) if (v.ndim and isinstance(tgt, np.ndarray) and v.size == tgt.size) else v

    @staticmethod
    def _arr(o):
        if isinstance(o, NumMat):
            return o.a
        o = np.asarray(o)
        return o.reshape(-1, 1) if o.ndim == 1 else o

    def _bin(self, o, f):
        return NumMat(f(self.a, NumMat._arr(o)))
    __add__ = lambda self, o: self._bin(o, lambda a, b: a + b)
    __radd__ = lambda self, o: self._bin(o, lambda a, b: b + a)
    __sub__ = lambda self, o: self._bin(o, lambda a, b: a - b)
    __rsub__ = lambda self, o: self._bin(o, lambda a, b: b - a)
    __mul__ = lambda self, o: self._bin(o, lambda a, b: a * b)
    __rmul__ = lambda self, o: self._bin(o, lambda a, b: b * a)
    __truediv__ = lambda self, o: self._bin(o, lambda a, b: a / b)
    __rtruediv__ = lambda self, o: self._bin(o, lambda a, b: b / a)
    __pow__ = lambda self, o: self._bin(o, lambda a, b: a ** b)
    __neg__ = lambda self: NumMat(-self.a)

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)


class Shape:
    """``SX.sym(name, n[, m])``: only the shape is ever read at module level (``d.size1()``).  ``SX(n, m)`` inside a user function is a
    zero matrix to be filled in (Ex_NMPC_dis.py:72,113), ``SX(1.)`` a number."""

    def __new__(cls, name=None, n=1, m=1):
        if isinstance(name, (int, np.integer)):
            return NumMat.zeros(name, n)
        if isinstance(name, float):
            return name
        return super().__new__(cls)

    def __init__(self, name, n=1, m=1):
        self.name, self.n, self.m = name, int(n), int(m)

    sym = classmethod(lambda cls, name, n=1, m=1: cls(name, n, m))
    size1 = lambda self: self.n
    size2 = lambda self: self.m
    shape = property(lambda self: (self.n, self.m))


BATCH = False      # inside ``batched()``: vectors are [n, P] arrays (P evaluation points side by side), per-point scalars are [P] rows


class batched:
    """Context in which the example's right-hand sides are evaluated at P points at once: ``x[i]`` is then a row over the points and
    ``vertcat`` stacks rows.  (Only element-wise functions - model, plant, stage cost - are called this way.)"""

    def __enter__(self):
        global BATCH
        self.prev, BATCH = BATCH, True

    def __exit__(self, *exc):
        global BATCH
        BATCH = self.prev


def vertcat(*parts):
    if not BATCH:
        return np.concatenate([np.ravel(np.asarray(a)) for a in parts])
    rows = []
    for a in parts:
        a = np.asarray(a)
        if a.ndim == 2:
            rows.extend(a)
        else:
            rows.append(a)
    return np.stack(np.broadcast_arrays(*rows))


def mtimes(*ms):
    out = ms[0]
    for b in ms[1:]:
        out = out * b if (np.ndim(out) == 0 or np.ndim(b) == 0) else np.asarray(out) @ np.asarray(b)
    return out.ravel()[0] if isinstance(out, np.ndarray) and out.shape == (1, 1) else out


def xQx(x, Q):
    """Utilities.py:247-265: the quadratic form x'Qx."""
    x = np.asarray(x)
    return x @ (np.asarray(Q) @ x)


def old_div(a, b):
    import numbers
    return a // b if isinstance(a, numbers.Integral) and isinstance(b, numbers.Integral) else a / b


def _standins():
    cas = types.ModuleType("casadi")
    cas.SX = cas.MX = Shape
    cas.DM = np.asarray
    cas.vertcat, cas.mtimes, cas.inv = vertcat, mtimes, np.linalg.inv
    cas.pi, cas.inf = math.pi, math.inf
    for fn in ("exp", "log", "sqrt", "sin", "cos", "tan", "tanh", "fabs"):
        setattr(cas, fn, getattr(np, fn))
    cas.if_else = lambda c, a, b: np.where(c, a, b)
    cas.__all__ = [k for k in vars(cas) if not k.startswith("_")]
    tools = types.ModuleType("casadi.tools"); tools.__all__ = []
    cas.tools = tools
    util = types.ModuleType("Utilities"); util.xQx = xQx; util.__all__ = ["xQx"]
    past = types.ModuleType("past"); putils = types.ModuleType("past.utils"); putils.old_div = old_div; past.utils = putils
    return {"casadi": cas, "casadi.tools": tools, "Utilities": util, "past": past, "past.utils": putils}


def load(path, overrides=None):
    """Execute the example over the defaults; returns its namespace (a dict)."""
    path = os.path.abspath(path)
    ns = dict(DEFAULTS)
    ns["__name__"] = os.path.splitext(os.path.basename(path))[0]
    ns["__file__"] = path
    mods = _standins()
    saved = {k: sys.modules.get(k) for k in mods}
    os.environ.setdefault("MPLBACKEND", "Agg")
    try:
        sys.modules.update(mods)
        with open(path) as fh:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")      # np.row_stack of the examples is deprecated
                exec(compile(fh.read(), path, "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    if overrides:
        ns.update(overrides)
    return ns
