"""Problem descriptor of the economic MPC path (BASELINE configs 4 and 5: ``Ex_ENMPC.py``).

What the reference builds symbolically for such an example, and what is traced here instead (:mod:`symtrace`):

* ``opt_dyn`` with ``ContForm`` (``Control_Calc.py:102-111,153-158``): every shooting interval integrates the Ex-file's
  ``User_fxm_Cont(x,u,d,t,px) + px`` together with the quadrature of ``User_fobj_Cont(x,u,y,xs,us,ys)``; terminal cost
  ``User_vfin(x_N, xs)`` (``:194-210``).  Traced: the augmented right-hand side ``[f; l]`` in the variables ``x, u`` with ``d, xs, us``
  as data (``y = Fy_model(x,u,d)`` and ``ys = Fy_model(xs,us,d)`` substituted: ``StateFeedback`` makes them ``x + Cd d``,
  ``Utilities.py:200-204``).
* ``opt_ss`` with ``User_fssobj`` (``Target_Calc.py:20-161``): traced cost in ``(xs, us, ys)``; the model's fixed-point equation
  uses the same ``User_fxm_Cont`` through ``Mx`` Runge-Kutta steps (``Utilities.py:157-183``).
* ``mhe_opt`` (``Utilities.py:825-990``) with ``User_fx_mhe_Cont`` and ``User_fobj_mhe``: traced model of the estimator (it must not
  use the noise ``w`` or the disturbance inside the differential equation: with ``offree = "lin"`` both enter the discrete map
  linearly, ``Utilities.py:798-821``) and traced cost in ``(w, v)``.

Anything outside this is refused with :class:`UnsupportedProblem`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from . import symtrace as st
from .problem import UnsupportedProblem, _mat, _vec

INF = float("inf")


@dataclass(repr=False)
class EconomicMPCProblem:
    nx: int
    nu: int
    ny: int
    nd: int
    nxp: int
    N: int
    h: float
    Nsim: int
    Mx: int
    quad_steps: int                 # RK4 steps per shooting interval of the ContForm OCP (the reference: IDAS, adaptive)
    f: List[st.Sym] = None          # model dx/dt in x[i], u[i], d[i], t
    fp: List[st.Sym] = None         # plant dxp/dt in xp[i], u[i], t
    ell: st.Sym = None              # stage cost rate in x[i], u[i], d[i], xs[i], us[i], t
    fss: st.Sym = None              # target cost in xs[i], us[i], ys[i] (+ xsp[i], usp[i], ysp[i])
    vfin: st.Sym = None             # terminal cost in x[i], xs[i]
    f_mhe: List[st.Sym] = None      # estimator model dx/dt in x[i], u[i], t
    c_mhe: st.Sym = None            # estimator stage cost in w[i], v[i], t
    g_ineq: List[st.Sym] = field(default_factory=list)      # user inequality rows of the OCP, each <= 0, in x[i], u[i], d[i], t (User_g_ineq)
    Bd: np.ndarray = None
    Cd: np.ndarray = None
    umin: np.ndarray = None
    umax: np.ndarray = None
    xmin: np.ndarray = None
    xmax: np.ndarray = None
    umin_ss: np.ndarray = None
    umax_ss: np.ndarray = None
    xmin_ss: np.ndarray = None
    xmax_ss: np.ndarray = None
    ymin_ss: np.ndarray = None
    ymax_ss: np.ndarray = None
    dmin: Optional[np.ndarray] = None
    dmax: Optional[np.ndarray] = None
    x0_p: np.ndarray = None
    x0_m: np.ndarray = None
    u0: np.ndarray = None
    max_iter: int = 100
    N_mhe: int = 0
    mhe_up: str = "smooth"          # update of the arrival cost: 'smooth' | 'filter' (Estimator.py:626-736)
    n_w: int = 0
    G_mhe: np.ndarray = None
    P0: np.ndarray = None
    x_bar: np.ndarray = None
    xmin_mhe: np.ndarray = None
    xmax_mhe: np.ndarray = None
    wmin: np.ndarray = None         # bounds of the estimator's state noise (Utilities.py:881-884,974-977); +-inf = absent
    wmax: np.ndarray = None
    estimator: str = "mhe"          # 'mhe' | 'ekf': the example's estimator switch (Ex_ENMPC.py:109-123, mhe_mod)
    R_wn: Optional[np.ndarray] = None      # white noise of the loop (MPC_code.py:537-541, 822-827; Ex_ENMPC.py:68-69 carries the state noise commented out): covariance of the
    G_wn: Optional[np.ndarray] = None      # measurement noise, input matrix and covariance of the state noise - enmpc.run_enmpc_stepwise(noise_seed=...) draws them
    Q_wn: Optional[np.ndarray] = None
    Q_kf: Optional[np.ndarray] = None      # extended Kalman filter on [x; d]: process / measurement noise covariances (Estimator.py:313-386); P0 is P(0|-1)
    R_kf: Optional[np.ndarray] = None
    name: str = ""
    funcs: Dict[str, Any] = field(default_factory=dict)

    def __repr__(self):
        return f"EconomicMPCProblem({self.name!r}, nx={self.nx}, nu={self.nu}, ny={self.ny}, nd={self.nd}, N={self.N}, N_mhe={self.N_mhe})"

    @property
    def nw(self) -> int:
        return self.nx * (self.N + 1) + self.nu * self.N


def _trace(fn, args, what):
    try:
        return st.flatten(fn(*args))
    except Exception as e:      # noqa: BLE001
        raise UnsupportedProblem(f"{what} cannot be traced: {e}") from e


def _depends_on(exprs, names):
    seen, stack = set(), list(exprs)
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        if n.op == "var" and any(str(n.val).startswith(p + "[") for p in names):
            return True
        stack.extend(n.args)
    return False


def is_economic(ns: Dict[str, Any]) -> bool:
    return ns.get("User_fobj_Cont") is not None or ns.get("User_fssobj") is not None or bool(ns.get("mhe", False))


def econ_problem_from_namespace(ns: Dict[str, Any], name: str = "", quad_steps: int = 20) -> EconomicMPCProblem:
    """Classify an economic Ex-file namespace (reference MPC_code.py:84-257,368-438 probes) and trace its functions."""
    has = lambda k: k in ns and ns[k] is not None and not k.startswith("__")
    for bad in ("User_fobj_Dis", "User_fobj_Coll", "User_h_eq", "User_g_ineq_SS", "User_h_eq_SS", "r_x", "rss_y", "Q", "Qss",
                "def_px", "def_py", "def_pxmp", "def_pymp", "def_pxp", "def_pyp", "A", "User_fxm_Dis", "User_fxp_Dis", "User_fym", "User_fyp",
                "defSP", "ymin", "ymax", "ymin_dyn", "ymax_dyn", "Dumin", "Dumax", "vmin", "vmax", "User_fx_mhe_Dis",
                "r_w", "Q_mhe"):
        if has(bad):
            raise UnsupportedProblem(f"'{bad}' is outside the economic path built so far")
    for flag in ("ssjacid", "Fp_nominal", "Adaptation", "Collocation", "slacks", "TermCons", "DUFormEcon", "kal", "kalss", "lue", "estimating"):
        if ns.get(flag, False) is True:
            raise UnsupportedProblem(f"flag {flag}=True is outside the economic path built so far")
    if ns.get("LinPar", True) is not True:
        raise UnsupportedProblem("LinPar = False is outside the economic path built so far")
    for req in ("User_fxm_Cont", "User_fxp_Cont", "User_fobj_Cont", "User_fssobj", "N", "h", "Nsim", "x", "u", "y", "d", "xp", "Bd", "Cd"):
        if not has(req):
            raise UnsupportedProblem(f"'{req}' missing: not an economic example with continuous model, plant and cost")
    if ns.get("StateFeedback", False) is not True or ns.get("offree", "no") != "lin":
        raise UnsupportedProblem("the economic path needs StateFeedback = True and offree = 'lin' (outputs y = x + Cd d)")
    use_mhe, use_ekf = bool(ns.get("mhe", False)), bool(ns.get("ekf", False))
    if use_mhe == use_ekf:
        raise UnsupportedProblem("the economic path needs one estimator: the moving-horizon estimator (mhe = True) or the extended Kalman filter (ekf = True)")
    if use_mhe and ns.get("mhe_up", "smooth") not in ("smooth", "filter"):
        raise UnsupportedProblem("mhe_up: 'smooth' or 'filter' (Estimator.py:626-736)")
    for req in (("N_mhe", "w", "User_fx_mhe_Cont", "User_fobj_mhe", "P0", "x_bar") if use_mhe else ("Q_kf", "R_kf", "P0")):
        if not has(req):
            raise UnsupportedProblem(f"'{req}' missing for the " + ("moving-horizon estimator" if use_mhe else "extended Kalman filter"))
    if has("G_wn") != has("Q_wn"):
        raise UnsupportedProblem("G_wn and Q_wn (the state noise, MPC_code.py:822-827) have to come together")
    if has("R_wn") or has("G_wn"):
        import warnings
        warnings.warn("R_wn / G_wn: the white noise of the example (unseeded in the reference, MPC_code.py:537-541, 822-827) is simulated only on request: the loops run "
                      "noise-free unless called with noise_seed=...", UserWarning, stacklevel=3)
    if (ns.get("dmin") is None) != (ns.get("dmax") is None):
        raise UnsupportedProblem("dmin and dmax have to come together")
    nx, nu, ny, nd, nxp = ns["x"].size1(), ns["u"].size1(), ns["y"].size1(), ns["d"].size1(), ns["xp"].size1()
    n_w = ns["w"].size1() if use_mhe else nx + nd
    if ny != nx or nd != ny:
        raise UnsupportedProblem("StateFeedback with an output disturbance needs ny = nx = nd")
    if nxp != nx:
        raise UnsupportedProblem("StateFeedback: the measurement is the plant state, so the plant needs the model's state dimension (nxp = nx)")
    if has("dhat0") and np.any(np.asarray(ns["dhat0"], dtype=float) != 0.0):      # MPC_code.py:459-462 seeds dhat_k with it; this path starts from zero
        raise UnsupportedProblem("a non-zero dhat0 is outside the economic path built so far")
    N_mhe = int(ns["N_mhe"]) if use_mhe else 2      # (the filter has no window; the per-model library still carries the estimator kernels, with the defaults below)
    if N_mhe < 2 or N_mhe > 63 or int(ns["N"]) < 2 or int(ns["N"]) > 64:
        raise UnsupportedProblem("horizons: 2 <= N <= 64, 2 <= N_mhe <= 63 (one stage per lane of a wavefront)")
    if n_w != nx + nd:
        raise UnsupportedProblem("the estimator's noise vector has to have nx + nd components")
    Bd, Cd = _mat(ns["Bd"], nx, nd, "Bd"), _mat(ns["Cd"], ny, nd, "Cd")
    vx, vu, vd, vxp, vt = st.symvec("x", nx), st.symvec("u", nu), st.symvec("d", nd), st.symvec("xp", nxp), st.Sym.var("t")
    vxs, vus, vys = st.symvec("xs", nx), st.symvec("us", nu), st.symvec("ys", ny)
    vw, vv = st.symvec("w", n_w), st.symvec("v", ny)
    col, zero = st.SymMat.col, (lambda n: st.SymMat.zeros(n))
    f = _trace(ns["User_fxm_Cont"], (col(vx), col(vu), col(vd), vt, zero(nx)), "User_fxm_Cont")
    fp = _trace(ns["User_fxp_Cont"], (col(vxp), vt, col(vu), zero(nxp), zero(nxp)), "User_fxp_Cont")
    # without a moving-horizon estimator its model is the controller's and its cost the unit quadratic (never evaluated: the library's estimator kernels are not launched)
    f_mhe = _trace(ns["User_fx_mhe_Cont"], (col(vx), col(vu), col(vd), vt, zero(nx), col(vw)), "User_fx_mhe_Cont") if use_mhe else f
    if len(f) != nx or len(fp) != nxp or len(f_mhe) != nx:
        raise UnsupportedProblem("a user function returns a vector of the wrong length")
    if _depends_on(f_mhe, ("w", "d")):
        raise UnsupportedProblem(("User_fx_mhe_Cont" if use_mhe else "User_fxm_Cont (the filter's state map)") + " uses w or d inside the differential equation: only their linear entry (+ Bd d, + G w) is built")
    # Fy_model with StateFeedback: x + Cd d (Utilities.py:200-204)
    fy = lambda xv: [xv[i] + sum((float(Cd[i, j]) * vd[j] for j in range(nd)), st.Sym.const(0.0)) for i in range(ny)]
    ell = _trace(ns["User_fobj_Cont"], (col(vx), col(vu), col(fy(vx)), col(vxs), col(vus), col(fy(vxs))), "User_fobj_Cont")
    sp = (col([st.Sym.const(0.0)] * nx), col([st.Sym.const(0.0)] * nu), col([st.Sym.const(0.0)] * ny))      # no defSP: the set points stay zero (MPC_code.py:440)
    fss = _trace(ns["User_fssobj"], (col(vxs), col(vus), col(vys), sp[0], sp[1], sp[2]), "User_fssobj")
    vfin = [st.Sym.const(0.0)]
    if has("User_vfin"):
        vfin = _trace(ns["User_vfin"], (col(vx), col(vxs)), "User_vfin")
    # user inequality rows of the OCP, G(x_k, u_k, y_k, d, t, px_k, py_k) <= 0 for k = 0..N-1 (Control_Calc.py:94-100,132-147,g4; MPC_code.py:306-314), with
    # y = Fy_model(x, u, d) substituted as in the cost and px = py = 0 (LinPar).  The reference's solver turns every such row into an equality with a slack
    # variable, g - s = 0, s <= 0 [ext]: here that slack is one more stage state (csrc/mpc_enmpc.hip:phase_ocp).  Slack variables of the Ex-file (`slacks`) stay refused.
    g_ineq = []
    if has("User_g_ineq"):
        g_ineq = _trace(ns["User_g_ineq"], (col(vx), col(vu), col(fy(vx)), col(vd), vt, zero(nx), zero(ny)), "User_g_ineq")
        if not g_ineq or len(g_ineq) > 4:
            raise UnsupportedProblem("User_g_ineq: between one and four rows are carried")
        if nx + len(g_ineq) > 8:
            raise UnsupportedProblem("User_g_ineq: the stage state (nx + rows) exceeds 8")
    c_mhe = _trace(ns["User_fobj_mhe"], (col(vw), col(vv), vt), "User_fobj_mhe") if use_mhe else [sum((a * a for a in list(vw) + list(vv)), st.Sym.const(0.0)) * st.Sym.const(0.5)]
    if len(ell) != 1 or len(fss) != 1 or len(vfin) != 1 or len(c_mhe) != 1:
        raise UnsupportedProblem("a cost function does not return a scalar")

    def pick(base, suffix, n, fill):
        v = ns.get(base + suffix)
        if v is None:
            v = ns.get(base)
        return _vec(v, n, fill)
    G = np.eye(nx + nd) if not has("G_mhe") else _mat(ns["G_mhe"], nx + nd, n_w, "G_mhe")      # MPC_code.py:387
    return EconomicMPCProblem(
        nx=nx, nu=nu, ny=ny, nd=nd, nxp=nxp, N=int(ns["N"]), h=float(ns["h"]), Nsim=int(ns["Nsim"]), Mx=int(ns.get("Mx", 10)), quad_steps=int(quad_steps),
        f=f, fp=fp, ell=ell[0], fss=fss[0], vfin=vfin[0], f_mhe=f_mhe, c_mhe=c_mhe[0], g_ineq=g_ineq, Bd=Bd, Cd=Cd,
        umin=pick("umin", "_dyn", nu, -INF), umax=pick("umax", "_dyn", nu, INF), xmin=pick("xmin", "_dyn", nx, -INF), xmax=pick("xmax", "_dyn", nx, INF),
        umin_ss=pick("umin", "_ss", nu, -INF), umax_ss=pick("umax", "_ss", nu, INF), xmin_ss=pick("xmin", "_ss", nx, -INF), xmax_ss=pick("xmax", "_ss", nx, INF),
        ymin_ss=pick("ymin", "_ss", ny, -INF), ymax_ss=pick("ymax", "_ss", ny, INF),
        dmin=None if ns.get("dmin") is None else _vec(ns["dmin"], nd, -INF), dmax=None if ns.get("dmax") is None else _vec(ns["dmax"], nd, INF),
        x0_p=_vec(ns["x0_p"], nxp, 0.0), x0_m=_vec(ns["x0_m"], nx, 0.0), u0=_vec(ns["u0"], nu, 0.0), max_iter=int(ns.get("Sol_itmax", 100)),
        N_mhe=N_mhe, mhe_up=str(ns.get("mhe_up", "smooth")), n_w=n_w, G_mhe=G, P0=_mat(ns["P0"], nx + nd, nx + nd, "P0"),
        x_bar=(np.asarray(ns["x_bar"], dtype=np.float64).reshape(nx + nd) if use_mhe else np.concatenate([_vec(ns["x0_m"], nx, 0.0), np.zeros(nd)])),
        wmin=_vec(ns.get("wmin") if use_mhe else None, n_w, -INF), wmax=_vec(ns.get("wmax") if use_mhe else None, n_w, INF),
        R_wn=_mat(ns["R_wn"], ny, ny, "R_wn") if has("R_wn") else None,
        G_wn=_mat(ns["G_wn"], nxp, nxp, "G_wn") if has("G_wn") else None, Q_wn=_mat(ns["Q_wn"], nxp, nxp, "Q_wn") if has("Q_wn") else None,
        estimator="mhe" if use_mhe else "ekf",
        Q_kf=None if use_mhe else _mat(ns["Q_kf"], nx + nd, nx + nd, "Q_kf"), R_kf=None if use_mhe else _mat(ns["R_kf"], ny, ny, "R_kf"),
        xmin_mhe=np.concatenate([_vec(ns.get("xmin"), nx, -INF), _vec(ns.get("dmin"), nd, -INF)]),      # MPC_code.py:397-402
        xmax_mhe=np.concatenate([_vec(ns.get("xmax"), nx, INF), _vec(ns.get("dmax"), nd, INF)]),
        name=name or str(ns.get("__name__", "")),
        funcs={k: ns[k] for k in ("User_fxm_Cont", "User_fxp_Cont", "User_fobj_Cont", "User_fssobj", "User_vfin", "User_fx_mhe_Cont", "User_fobj_mhe", "User_g_ineq") if has(k)},
    )
