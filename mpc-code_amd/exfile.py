"""Ex-file loader: run an unmodified ``Ex_*.py`` problem-definition file without CasADi.

The reference selects a problem by star-importing an example module over a set of
defaults (reference ``MPC_code.py:23`` then ``:25-28``) and then probes for optional
names with ``'Name' in locals()`` (``MPC_code.py:94-257``).  This module reproduces
that *surface*: :func:`load_exfile` executes the file into a namespace that was
seeded with the defaults of reference ``Default_Values.py:16-131`` and returns that
namespace as a plain ``dict``.  The hot path only needs the numbers in it.

The shipped examples do ``from casadi import *``, ``from casadi.tools import *``,
``from Utilities import *`` and (non-linear ones) ``from past.utils import old_div``
on their first lines.  None of those packages is a dependency of this project, so
while the file executes we install tiny stand-in modules in ``sys.modules``:

* ``casadi``      - ``SX.sym(name, n[, m])`` objects that know their shape
  (``.size1()`` is all the driver reads: reference ``MPC_code.py:31-35``) plus the
  handful of element-wise names the example bodies mention (``exp``, ``sqrt``,
  ``vertcat`` ...), bound to NumPy: nothing symbolic is evaluated, but a user
  *plant* function (``User_fxp_Cont``) can be called on arrays ``x[nx, B]``
  (:meth:`LinearMPCProblem.plant_step`).
* ``casadi.tools``, ``Utilities`` - empty.
* ``past.utils``  - ``old_div`` (floor division of two integers, true division otherwise).

The stand-ins are removed again afterwards; a real CasADi, if one is installed,
is never touched.
"""
from __future__ import annotations

import math
import os
import sys
import types
from typing import Any, Dict, Optional

import numpy as np

__all__ = ["DEFAULTS", "load_exfile", "SymVec"]


# --------------------------------------------------------------------------------------
# Defaults: reference Default_Values.py:16-131 restated as data (same names, same values)
# --------------------------------------------------------------------------------------
DEFAULTS: Dict[str, Any] = dict(
    estimating=False, ssjacid=False, StateFeedback=False, Fp_nominal=False, offree="no",
    umin=None, umax=None, xmin=None, xmax=None, ymin=None, ymax=None,
    umin_ss=None, umax_ss=None, xmin_ss=None, xmax_ss=None, ymin_ss=None, ymax_ss=None,
    umin_dyn=None, umax_dyn=None, xmin_dyn=None, xmax_dyn=None, ymin_dyn=None, ymax_dyn=None,
    dmin=None, dmax=None, Dumin=None, Dumax=None, wmin=None, wmax=None, vmin=None, vmax=None,
    QForm_ss=False, DUssForm=False, Adaptation=False,
    ContForm=False, TermCons=False, QForm=False, DUForm=False, DUFormEcon=False,
    Sol_itmax=100, Sol_Hess_constss="no", Sol_Hess_constdyn="no", Sol_Hess_constmhe="no",
    kalss=False, lue=False, kal=False, ekf=False, mhe=False,
    Collocation=False, LinPar=True, slacks=False, slacksG=True, slacksH=True,
)


class SymVec:
    """Shape-only stand-in for ``casadi.SX.sym``.

    Indexing returns another :class:`SymVec` so that expressions inside ``def`` blocks of
    an example parse; arithmetic on it raises, because the linear hot path never
    evaluates symbolic model code (the non-linear front end is a later scope row).
    """

    def __new__(cls, name=None, n: int = 1, m: int = 1):
        if isinstance(name, (int, np.integer)):      # SX(n, m): a zero matrix to be filled in (Ex_NMPC_dis.py:67,89)
            from .symtrace import SymMat
            return SymMat.zeros(name, n)
        if isinstance(name, float):                  # SX(1.): a number
            return name
        return super().__new__(cls)

    def __init__(self, name: str, n: int = 1, m: int = 1):
        self.name, self._n, self._m = name, int(n), int(m)

    @classmethod
    def sym(cls, name: str, n: int = 1, m: int = 1) -> "SymVec":
        if isinstance(n, tuple):
            n, m = n
        return cls(name, n, m)

    def size1(self) -> int:
        return self._n

    def size2(self) -> int:
        return self._m

    @property
    def shape(self):
        return (self._n, self._m)

    def __getitem__(self, idx):
        return SymVec(f"{self.name}[{idx}]", 1, 1)

    def __repr__(self):
        return f"SymVec({self.name!r}, {self._n}, {self._m})"


def _vertcat(*parts):
    """Numeric ``vertcat``: stacks scalars / arrays along axis 0 (the state index), broadcasting over trailing batch
    axes, so that a user plant function written for CasADi evaluates on ``x[nx, B]`` arrays; shape objects pass through;
    traced expressions (:mod:`symtrace`) come back as a flat list of nodes."""
    from .symtrace import Sym, SymMat, flatten
    if any(isinstance(a, SymVec) for a in parts):
        return list(parts)
    if any(isinstance(a, (Sym, SymMat)) or (isinstance(a, list) and a and isinstance(a[0], Sym)) for a in parts):
        return SymMat.col(flatten(parts))
    rows = [np.asarray(a, dtype=np.float64) for a in parts]
    tail = np.broadcast_shapes(*[r.shape[1:] if r.ndim >= 2 else r.shape for r in rows])
    out = []
    for r in rows:
        if r.ndim >= 2:                       # an [n, B] block: its rows are appended one by one
            out.extend(np.broadcast_to(r, (r.shape[0],) + tail))
        else:                                 # a scalar, or one component over the batch [B]
            out.append(np.broadcast_to(r, tail))
    return np.stack(out, axis=0)


def _old_div(a, b):
    """``past.utils.old_div``: floor division when both operands are integers, true division otherwise."""
    import numbers
    if isinstance(a, numbers.Integral) and isinstance(b, numbers.Integral):
        return a // b
    return a / b


def _make_standins() -> Dict[str, types.ModuleType]:
    from .symtrace import mtimes as symtrace_mtimes
    cas = types.ModuleType("casadi")
    cas.SX = SymVec
    cas.MX = SymVec
    cas.DM = np.asarray
    cas.vertcat = _vertcat
    cas.horzcat = lambda *a: list(a)
    cas.mtimes = symtrace_mtimes
    cas.pi = math.pi
    cas.inf = math.inf
    from . import symtrace

    def _elementwise(fn):      # NumPy's on numbers and arrays; the node's own method on a traced expression
        npf = getattr(np, fn)
        return lambda v: getattr(v, fn)() if isinstance(v, symtrace.Sym) else npf(v)
    for fn in ("exp", "log", "sqrt", "sin", "cos", "tan", "fabs", "tanh"):
        setattr(cas, fn, _elementwise(fn))
    cas.if_else = symtrace.if_else
    cas.inv = np.linalg.inv      # numeric weights only (Ex_ENMPC.py:171: inv(Q) of a constant matrix)
    cas.__all__ = [k for k in vars(cas) if not k.startswith("_")]
    tools = types.ModuleType("casadi.tools")
    tools.__all__ = []
    cas.tools = tools
    util = types.ModuleType("Utilities")

    def xQx(x, Q):      # reference Utilities.py:247-265: the quadratic form x'Qx (the economic example's estimator cost uses it)
        return symtrace_mtimes(x.T if hasattr(x, "T") else x, symtrace_mtimes(np.asarray(Q, dtype=np.float64), x))
    util.xQx = xQx
    util.__all__ = ["xQx"]
    past = types.ModuleType("past")
    putils = types.ModuleType("past.utils")
    putils.old_div = _old_div
    past.utils = putils
    mods = {"casadi": cas, "casadi.tools": tools, "Utilities": util, "past": past, "past.utils": putils}
    try:  # the examples import pylab but never use it at module level
        import matplotlib  # noqa: F401
    except Exception:  # pragma: no cover - matplotlib is present in the image
        mpl = types.ModuleType("matplotlib")
        mpl.pylab = types.ModuleType("matplotlib.pylab")
        mods.update({"matplotlib": mpl, "matplotlib.pylab": mpl.pylab})
    return mods


def load_exfile(path: str, overrides: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """Execute an Ex-style file over the defaults and return the resulting namespace.

    ``overrides`` are applied *after* the file ran (the driver-level ``N`` / ``N_mhe``
    overrides BASELINE.json's configs need).  Mirrors reference ``MPC_code.py:23-28``.
    """
    path = os.path.abspath(path)
    with open(path, "r") as fh:
        src = fh.read()
    ns: Dict[str, Any] = dict(DEFAULTS)
    ns["__name__"] = os.path.splitext(os.path.basename(path))[0]
    ns["__file__"] = path
    standins = _make_standins()
    saved = {k: sys.modules.get(k) for k in standins}
    os.environ.setdefault("MPLBACKEND", "Agg")
    try:
        sys.modules.update(standins)
        exec(compile(src, path, "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    if overrides:
        ns.update(overrides)
    ns["__defined__"] = {k for k in ns if not k.startswith("__")}
    return ns
