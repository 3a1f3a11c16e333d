"""A small expression tracer: what the non-linear front end needs from ``casadi.SX`` and nothing more.

The reference builds its non-linear models symbolically: the Ex-file's ``User_fxm_Cont(x,u,d,t,px)`` /
``User_fym(x,u,d,t,py)`` (and the plant's ``User_fxp_Cont`` / ``User_fyp``) are Python functions that CasADi
evaluates on ``SX`` symbols (reference ``Utilities.py:157-183``, ``:58-82``), differentiates (``jacobian``,
``Estimator.py:343-373``) and compiles for its virtual machine.  Here the same functions are called on
:class:`Sym` objects - operator overloading records the expression DAG - and the DAG is

* differentiated symbolically (:func:`jacobian`, forward accumulation with memoisation),
* evaluated with NumPy over a batch (:func:`evaluate`), and
* emitted as straight-line C++ with common sub-expressions shared (:func:`emit_cpp`) - the ``__device__``
  model functions of the NMPC kernels.

Supported: ``+ - * / **`` (constant or general exponent), unary minus, ``exp log sqrt sin cos tan tanh fabs``
(also as NumPy ufuncs: ``np.exp(sym)`` dispatches to ``sym.exp()``), comparisons and ``if_else``.  Nodes are
hash-consed, so equal sub-expressions are one object.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence

import numpy as np

_UNARY = {"exp": np.exp, "log": np.log, "sqrt": np.sqrt, "sin": np.sin, "cos": np.cos, "tan": np.tan, "tanh": np.tanh,
          "fabs": np.abs, "neg": np.negative}
_CMP = {"le": np.less_equal, "lt": np.less, "ge": np.greater_equal, "gt": np.greater}
_pool: Dict[tuple, "Sym"] = {}


class Sym:
    """One node of the DAG.  ``op``: 'const' | 'var' | 'add' 'sub' 'mul' 'div' 'pow' | unary name | cmp name | 'sel'."""
    __slots__ = ("op", "args", "val", "_id")
    __array_priority__ = 1000.0

    def __new__(cls, op, args=(), val=None):
        key = (op, tuple(id(a) for a in args), val)
        hit = _pool.get(key)
        if hit is not None:
            return hit
        self = object.__new__(cls)
        self.op, self.args, self.val, self._id = op, tuple(args), val, len(_pool)
        _pool[key] = self
        return self

    # ---- construction helpers
    @staticmethod
    def const(v) -> "Sym":
        return Sym("const", (), float(v))

    @staticmethod
    def var(name: str) -> "Sym":
        return Sym("var", (), name)

    @staticmethod
    def lift(v) -> "Sym":
        return v if isinstance(v, Sym) else Sym.const(v)

    def is_const(self, v=None) -> bool:
        return self.op == "const" and (v is None or self.val == v)

    # ---- arithmetic with local simplification (keeps the generated code small)
    def __add__(self, o):
        o = Sym.lift(o)
        if self.is_const() and o.is_const(): return Sym.const(self.val + o.val)
        if self.is_const(0.0): return o
        if o.is_const(0.0): return self
        return Sym("add", (self, o))
    __radd__ = lambda self, o: Sym.lift(o).__add__(self)

    def __sub__(self, o):
        o = Sym.lift(o)
        if self.is_const() and o.is_const(): return Sym.const(self.val - o.val)
        if o.is_const(0.0): return self
        if self.is_const(0.0): return -o
        return Sym("sub", (self, o))
    __rsub__ = lambda self, o: Sym.lift(o).__sub__(self)

    def __mul__(self, o):
        o = Sym.lift(o)
        if self.is_const() and o.is_const(): return Sym.const(self.val * o.val)
        if self.is_const(0.0) or o.is_const(0.0): return Sym.const(0.0)
        if self.is_const(1.0): return o
        if o.is_const(1.0): return self
        if self.is_const(-1.0): return -o
        if o.is_const(-1.0): return -self
        return Sym("mul", (self, o))
    __rmul__ = lambda self, o: Sym.lift(o).__mul__(self)

    def __truediv__(self, o):
        o = Sym.lift(o)
        if self.is_const() and o.is_const(): return Sym.const(self.val / o.val)
        if self.is_const(0.0): return Sym.const(0.0)
        if o.is_const(1.0): return self
        return Sym("div", (self, o))
    __rtruediv__ = lambda self, o: Sym.lift(o).__truediv__(self)
    __floordiv__ = __truediv__            # old_div of the examples: floor division only for two Python integers
    __rfloordiv__ = __rtruediv__

    def __neg__(self):
        if self.is_const(): return Sym.const(-self.val)
        if self.op == "neg": return self.args[0]
        return Sym("neg", (self,))

    def __pos__(self):
        return self

    def __pow__(self, o):
        o = Sym.lift(o)
        if self.is_const() and o.is_const(): return Sym.const(self.val ** o.val)
        if o.is_const(1.0): return self
        if o.is_const(0.0): return Sym.const(1.0)
        if o.is_const(2.0): return self * self
        if o.is_const(0.5): return self.sqrt()
        return Sym("pow", (self, o))
    __rpow__ = lambda self, o: Sym.lift(o).__pow__(self)

    def _un(self, name):
        if self.is_const(): return Sym.const(float(_UNARY[name](self.val)))
        return Sym(name, (self,))

    exp = lambda self: self._un("exp")
    log = lambda self: self._un("log")
    sqrt = lambda self: self._un("sqrt")
    sin = lambda self: self._un("sin")
    cos = lambda self: self._un("cos")
    tan = lambda self: self._un("tan")
    tanh = lambda self: self._un("tanh")
    fabs = lambda self: self._un("fabs")
    __abs__ = fabs

    def _cmp(self, name, o):
        return Sym(name, (self, Sym.lift(o)))
    __le__ = lambda self, o: self._cmp("le", o)
    __lt__ = lambda self, o: self._cmp("lt", o)
    __ge__ = lambda self, o: self._cmp("ge", o)
    __gt__ = lambda self, o: self._cmp("gt", o)
    __hash__ = lambda self: self._id

    def __bool__(self):
        raise TypeError("the truth value of a traced expression is not known: use if_else(cond, a, b)")

    def __repr__(self):
        if self.op == "const": return repr(self.val)
        if self.op == "var": return str(self.val)
        return f"<{self.op} node {self._id}>"      # never the whole tree: shared sub-expressions make it exponentially long


def if_else(cond, a, b):
    """``casadi.if_else``: traced when the condition (or a branch) is symbolic, ``np.where`` on numbers."""
    if isinstance(cond, Sym) or isinstance(a, Sym) or isinstance(b, Sym):
        cond = Sym.lift(cond)
        if cond.op == "const":
            return Sym.lift(a) if cond.val else Sym.lift(b)
        return Sym("sel", (cond, Sym.lift(a), Sym.lift(b)))
    return np.where(cond, a, b)


def symvec(name: str, n: int) -> List[Sym]:
    return [Sym.var(f"{name}[{i}]") for i in range(n)]


class SymMat:
    """A small dense matrix of traced expressions / numbers with the part of ``casadi.SX``'s surface the Ex-files use inside their
    functions: ``SX(n, m)`` zeros, element and slice reads (a slice is a COPY, as in CasADi), element and slice assignment, ``.T``,
    ``.shape``, ``size1()``, element-wise arithmetic with numbers / NumPy arrays / other matrices, ``mtimes``.  A column vector
    indexed with one integer gives the element itself."""
    __array_priority__ = 1000      # NumPy lets our operators win: np.array * SymMat -> SymMat.__rmul__

    def __init__(self, a):
        arr = np.empty(np.shape(a), dtype=object)
        arr[...] = a
        self.a = arr.reshape(-1, 1) if arr.ndim < 2 else arr

    @classmethod
    def zeros(cls, n, m=1):
        z = np.empty((int(n), int(m)), dtype=object); z[...] = 0.0
        return cls(z)

    @classmethod
    def col(cls, items):
        z = np.empty((len(items), 1), dtype=object)
        for i, v in enumerate(items): z[i, 0] = v
        return cls(z)

    shape = property(lambda self: self.a.shape)
    T = property(lambda self: SymMat(self.a.T.copy()))
    def size1(self): return self.a.shape[0]
    def size2(self): return self.a.shape[1]
    def __len__(self): return self.a.shape[0]
    def __iter__(self): return iter(self.a.ravel()) if self.a.shape[1] == 1 else iter(SymMat(r[None].copy()) for r in self.a)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)) and self.a.shape[1] == 1:
            return self.a[idx, 0]
        r = self.a[idx]
        return SymMat(np.array(r, dtype=object).copy()) if isinstance(r, np.ndarray) else r

    def __setitem__(self, idx, v):
        if isinstance(v, SymMat): v = v.a
        elif isinstance(v, (list, tuple)): v = SymMat.col(flatten(v)).a
        if isinstance(idx, (int, np.integer)) and self.a.shape[1] == 1:
            if isinstance(v, np.ndarray): v = v.ravel()[0]
            self.a[idx, 0] = v
        else:
            tgt = self.a[idx]
            if isinstance(v, np.ndarray) and isinstance(tgt, np.ndarray) and v.size == tgt.size: v = v.reshape(tgt.shape)
            self.a[idx] = v

    @staticmethod
    def _arr(o):
        if isinstance(o, SymMat): return o.a
        if isinstance(o, (list, tuple)): return SymMat.col(flatten(o)).a
        if isinstance(o, np.ndarray) and o.ndim == 1: return o.reshape(-1, 1)
        return o

    def _bin(self, o, f): return SymMat(f(self.a, SymMat._arr(o)))
    __add__ = lambda self, o: self._bin(o, lambda a, b: a + b)
    __radd__ = lambda self, o: self._bin(o, lambda a, b: b + a)
    __sub__ = lambda self, o: self._bin(o, lambda a, b: a - b)
    __rsub__ = lambda self, o: self._bin(o, lambda a, b: b - a)
    __mul__ = lambda self, o: self._bin(o, lambda a, b: a * b)
    __rmul__ = lambda self, o: self._bin(o, lambda a, b: b * a)
    __truediv__ = lambda self, o: self._bin(o, lambda a, b: a / b)
    __rtruediv__ = lambda self, o: self._bin(o, lambda a, b: b / a)
    __pow__ = lambda self, o: self._bin(o, lambda a, b: a ** b)
    __neg__ = lambda self: SymMat(-self.a)

    def __array__(self, dtype=None, copy=None):
        return np.array(self.a.tolist(), dtype=np.float64 if dtype is None else dtype)

    def __repr__(self):
        return f"SymMat{self.a.shape}"


def mtimes(*ms):
    """``casadi.mtimes``: matrix product of numbers, NumPy arrays and :class:`SymMat`; a 1 x 1 result is returned as its element."""
    def arr(m):
        if isinstance(m, SymMat): return m.a
        if isinstance(m, (list, tuple)): return SymMat.col(flatten(m)).a
        m = np.asarray(m)
        return m.reshape(-1, 1) if m.ndim == 1 else m
    out = ms[0]
    for m in ms[1:]:
        a, b = arr(out), arr(m)
        out = a * b if (np.ndim(a) == 0 or np.ndim(b) == 0) else SymMat(np.dot(a, b))
    if isinstance(out, SymMat) and out.a.shape == (1, 1):
        return out.a[0, 0]
    return out


def flatten(v) -> List[Sym]:
    """What a traced ``vertcat`` / list / scalar returns, as a flat list of nodes."""
    if isinstance(v, Sym):
        return [v]
    if isinstance(v, (int, float, np.floating, np.integer)):
        return [Sym.const(float(v))]
    if isinstance(v, SymMat):
        return [Sym.lift(e) for e in v.a.ravel()]
    out: List[Sym] = []
    for a in v:
        out.extend(flatten(a))
    return out


# ---------------------------------------------------------------------------------------------------
def _topo(outputs: Iterable[Sym]) -> List[Sym]:
    seen, order = set(), []
    stack = [(o, False) for o in outputs]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n); continue
        if id(n) in seen:
            continue
        seen.add(id(n))
        stack.append((n, True))
        stack.extend((a, False) for a in n.args)
    return order


def diff(expr: Sym, wrt: Sym, memo=None) -> Sym:
    """d expr / d wrt (wrt a 'var' node), forward accumulation over the DAG."""
    memo = {} if memo is None else memo
    for n in _topo([expr]):
        if id(n) in memo:
            continue
        a = n.args
        d = [memo[id(x)] for x in a]
        if n.op == "const": r = Sym.const(0.0)
        elif n.op == "var": r = Sym.const(1.0 if n is wrt else 0.0)
        elif n.op == "add": r = d[0] + d[1]
        elif n.op == "sub": r = d[0] - d[1]
        elif n.op == "mul": r = d[0] * a[1] + a[0] * d[1]
        elif n.op == "div": r = (d[0] - n * d[1]) / a[1]
        elif n.op == "neg": r = -d[0]
        elif n.op == "pow":
            if a[1].is_const(): r = a[1] * a[0] ** (a[1].val - 1.0) * d[0]
            else: r = n * (d[1] * a[0].log() + a[1] * d[0] / a[0])
        elif n.op == "exp": r = n * d[0]
        elif n.op == "log": r = d[0] / a[0]
        elif n.op == "sqrt": r = d[0] / (2.0 * n)
        elif n.op == "sin": r = a[0].cos() * d[0]
        elif n.op == "cos": r = -(a[0].sin() * d[0])
        elif n.op == "tan": r = d[0] * (1.0 + n * n)
        elif n.op == "tanh": r = d[0] * (1.0 - n * n)
        elif n.op == "fabs": r = if_else(a[0] >= 0.0, d[0], -d[0])
        elif n.op in _CMP: r = Sym.const(0.0)
        elif n.op == "sel": r = if_else(a[0], d[1], d[2])
        else: raise ValueError(n.op)
        memo[id(n)] = Sym.lift(r)
    return memo[id(expr)]


def jacobian(exprs: Sequence[Sym], wrt: Sequence[Sym]) -> List[List[Sym]]:
    """[d exprs[i] / d wrt[j]]."""
    cols = []
    for v in wrt:
        memo = {}
        cols.append([diff(e, v, memo) for e in exprs])
    return [[cols[j][i] for j in range(len(wrt))] for i in range(len(exprs))]


def evaluate(outputs: Sequence[Sym], values: Dict[str, np.ndarray]):
    """NumPy evaluation; ``values`` maps variable names ('x[0]', ...) to scalars or arrays that broadcast."""
    val = {}
    for n in _topo(outputs):
        a = [val[id(x)] for x in n.args]
        if n.op == "const": r = n.val
        elif n.op == "var": r = values[n.val]
        elif n.op == "add": r = a[0] + a[1]
        elif n.op == "sub": r = a[0] - a[1]
        elif n.op == "mul": r = a[0] * a[1]
        elif n.op == "div": r = a[0] / a[1]
        elif n.op == "pow": r = a[0] ** a[1]
        elif n.op in _UNARY: r = _UNARY[n.op](a[0])
        elif n.op in _CMP: r = _CMP[n.op](a[0], a[1])
        elif n.op == "sel": r = np.where(a[0], a[1], a[2])
        else: raise ValueError(n.op)
        val[id(n)] = r
    return [val[id(o)] for o in outputs]


_CFUN = {"exp": "exp", "log": "log", "sqrt": "sqrt", "sin": "sin", "cos": "cos", "tan": "tan", "tanh": "tanh", "fabs": "fabs"}
_CCMP = {"le": "<=", "lt": "<", "ge": ">=", "gt": ">"}


def emit_cpp(outputs: Sequence[Sym], out_names: Sequence[str], var_map: Dict[str, str], tmp: str = "t") -> str:
    """Straight-line C++ (double arithmetic) computing ``out_names[i] = outputs[i]``; every shared node is computed once.
    ``var_map``: variable name -> C++ expression (``'x[0]' -> 'x[0]'``)."""
    lines, name = [], {}
    uses = {}
    order = _topo(outputs)
    for n in order:
        for a in n.args:
            uses[id(a)] = uses.get(id(a), 0) + 1
    k = 0
    for n in order:
        a = [name[id(x)] for x in n.args]
        if n.op == "const":
            v = n.val
            name[id(n)] = ("%.17g" % v) if math.isfinite(v) else ("INFINITY" if v > 0 else "-INFINITY")
            if "." not in name[id(n)] and "e" not in name[id(n)] and "INF" not in name[id(n)]:
                name[id(n)] += ".0"
            if v < 0: name[id(n)] = "(" + name[id(n)] + ")"
            continue
        if n.op == "var":
            name[id(n)] = var_map[n.val]; continue
        if n.op == "add": e = f"{a[0]} + {a[1]}"
        elif n.op == "sub": e = f"{a[0]} - {a[1]}"
        elif n.op == "mul": e = f"{a[0]} * {a[1]}"
        elif n.op == "div": e = f"{a[0]} / {a[1]}"
        elif n.op == "neg": e = f"-{a[0]}"
        elif n.op == "pow": e = f"pow({a[0]}, {a[1]})"
        elif n.op in _CFUN: e = f"{_CFUN[n.op]}({a[0]})"
        elif n.op in _CCMP: e = f"({a[0]} {_CCMP[n.op]} {a[1]})"
        elif n.op == "sel": e = f"({a[0]} ? {a[1]} : {a[2]})"
        else: raise ValueError(n.op)
        if n.op in _CCMP:            # conditions are inlined (bool), never stored in a double
            name[id(n)] = e; continue
        nm = f"{tmp}{k}"; k += 1
        lines.append(f"const double {nm} = {e};")
        name[id(n)] = nm
    for o, on in zip(outputs, out_names):
        lines.append(f"{on} = {name[id(o)]};")
    return "\n".join(lines)
