"""Numeric problem descriptor for the linear MPC hot path.

Turns the namespace of an Ex-style file (see :mod:`exfile`) into the numbers the HIP
solver consumes.  It replaces the *symbolic* builders of the reference - ``defF_model``
(``Utilities.py:102-245``), ``defF_p`` (``:21-100``), ``defFss_obj`` (``:267-321``),
``defF_obj`` (``:323-381``), ``defVfin`` (``:383-420``), ``opt_ss``
(``Target_Calc.py:20-161``), ``opt_dyn`` (``Control_Calc.py:20-260``) and the gain
computation ``Kkalss`` (``Estimator.py:103-229``) - for the case where model, plant and
cost are given as matrices (``A,B,C``, ``Ap,Bp,Cp``, ``Q`` with ``R`` or ``S``, ``Qss`` with
``Rss`` or ``Sss``), which is what ``Ex_LMPC_CSTR.py`` and ``Ex_LMPC_WB.py`` do.

Anything outside that (user functions ``User_f*``, LP costs ``r_x``/``rss_y``, slacks,
collocation, adaptation, MHE, time-varying ``def_px``/``def_py``) raises
:class:`UnsupportedProblem` - the accelerated path never silently falls back.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Optional

import numpy as np
import scipy.linalg as scla

__all__ = ["LinearMPCProblem", "UnsupportedProblem", "problem_from_namespace"]

INF = float("inf")


class UnsupportedProblem(NotImplementedError):
    """The Ex-file asks for a feature the batched linear path does not implement."""


def _vec(v, n, fill):
    if v is None:
        return np.full(n, fill, dtype=np.float64)
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.size != n:
        raise ValueError(f"bound vector has {a.size} entries, expected {n}")
    return a.copy()


def _mat(v, r, c, name):
    a = np.array(v, dtype=np.float64)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    if a.shape != (r, c):
        raise ValueError(f"{name} has shape {a.shape}, expected {(r, c)}")
    return np.ascontiguousarray(a)


@dataclass
class LinearMPCProblem:
    # dimensions (reference MPC_code.py:31-52)
    nx: int
    nu: int
    ny: int
    nd: int
    nxp: int
    N: int
    h: float
    Nsim: int
    # model x+ = A(x-xlin)+B(u-ulin)+xlin+Bd d (+px), y = C(x-xlin)+ylin+Cd d (+py)  (Utilities.py:135-155,208-244)
    A: np.ndarray
    B: np.ndarray
    C: np.ndarray
    Bd: np.ndarray
    Cd: np.ndarray
    fx_const: np.ndarray  # xlin - A xlin - B ulin
    fy_const: np.ndarray  # ylin - C xlin
    # plant xp+ = Ap xp + Bp u + pxp (+pxmp), y = Cp xp + pyp (+pymp)                 (Utilities.py:45-49,88-91)
    Ap: np.ndarray
    Bp: np.ndarray
    Cp: np.ndarray
    # dynamic cost 1/2 dx'Q dx + 1/2 du'R du  (du = u-us, or u-u_prev when DUForm)    (Utilities.py:353-367, Control_Calc.py:173-188)
    Q: np.ndarray
    R: np.ndarray          # R, or S when DUForm
    DUForm: bool
    P: np.ndarray          # terminal weight, DARE(A,B,Q,R)                              (Utilities.py:403-413, MPC_code.py:250-255)
    # target cost 1/2 (ys-ysp)'Qss(ys-ysp) + 1/2 dus'Rss dus                             (Utilities.py:299-313, Target_Calc.py:112-124)
    Qss: np.ndarray
    Rss: np.ndarray
    DUssForm: bool
    # bounds (+-inf = absent), dynamic and target problems                               (MPC_code.py:291-304)
    umin: np.ndarray
    umax: np.ndarray
    xmin: np.ndarray
    xmax: np.ndarray
    ymin: np.ndarray
    ymax: np.ndarray
    y_bounded: bool        # yFree is False (Control_Calc.py:60-63): the g1 rows exist
    umin_ss: np.ndarray = None
    umax_ss: np.ndarray = None
    xmin_ss: np.ndarray = None
    xmax_ss: np.ndarray = None
    ymin_ss: np.ndarray = None
    ymax_ss: np.ndarray = None
    dmin: Optional[np.ndarray] = None
    dmax: Optional[np.ndarray] = None
    # bounds on u_k - u_{k-1} (u_0 - u_prev for k = 0): the g2 rows of opt_dyn, Control_Calc.py:163-169,241-243; None = absent
    Dumin: Optional[np.ndarray] = None
    Dumax: Optional[np.ndarray] = None
    # estimator
    estimator: str = "none"   # 'kal' (Estimator.py:263-311) | 'kalss' (:231-261, also for lue) | 'none'
    Q_kf: Optional[np.ndarray] = None
    R_kf: Optional[np.ndarray] = None
    P0: Optional[np.ndarray] = None
    K: Optional[np.ndarray] = None
    # initial conditions (MPC_code.py:449-463)
    x0_p: np.ndarray = None
    x0_m: np.ndarray = None
    u0: np.ndarray = None
    dhat0: np.ndarray = None
    max_iter: int = 100
    # schedules (python callables of t, same contract as the Ex-file: each returns a list)
    defSP: Optional[Callable] = None
    def_pxp: Optional[Callable] = None
    def_pyp: Optional[Callable] = None
    name: str = ""
    extras: Dict[str, Any] = field(default_factory=dict)
    # non-linear continuous-time plant (Utilities.py:58-82): User_fxp_Cont(x, t, u, pxp, pxmp) integrated by classical RK4
    # with Mx sub-steps per sampling interval; the controller path (estimator, target, OCP) stays linear.
    plant_fx_cont: Optional[Callable] = None
    plant_Mx: int = 10
    TermCons: bool = False    # terminal equality x_N = xs (Control_Calc.py:197-198)
    def_px: Optional[Callable] = None     # time-varying model parameters over the horizon (MPC_code.py:492-497)
    def_py: Optional[Callable] = None
    # soft output constraints (Control_Calc.py:39-40,186-192,228-239): one slack vector [sl_ub; sl_lb] >= 0 shared by all stages, Sl' Ws Sl in every stage's cost
    slacks: bool = False
    Ws: Optional[np.ndarray] = None
    # affine user inequality rows of the OCP (User_g_ineq, Control_Calc.py:94-100,132-147): Gx x_k + Gu u_k + Gd dhat + g0 <= 0 for k = 0..N-1, with
    # y_k = C x_k + Cd dhat + fy_const already substituted ([ng, nx], [ng, nu], [ng, nd], [ng]); None: no rows
    Gx: Optional[np.ndarray] = None
    Gu: Optional[np.ndarray] = None
    Gd: Optional[np.ndarray] = None
    g0: Optional[np.ndarray] = None

    @property
    def n_user_rows(self) -> int:
        return 0 if self.Gx is None else int(self.Gx.shape[0])

    # ------------------------------------------------------------------ schedules
    def schedules(self, nsteps: int, k0: int = 0) -> Dict[str, np.ndarray]:
        """Evaluate the time callbacks the driver calls once per step (MPC_code.py:487-515,677-680)."""
        ysp = np.zeros((nsteps, self.ny))
        usp = np.zeros((nsteps, self.nu))
        xsp = np.zeros((nsteps, self.nx))
        pxp = np.zeros((nsteps, self.nxp))
        pyp = np.zeros((nsteps, self.ny))
        for i in range(nsteps):
            t = (k0 + i) * self.h
            if self.defSP is not None:
                a, b, c = self.defSP(t)
                ysp[i], usp[i], xsp[i] = np.ravel(a), np.ravel(b), np.ravel(c)
            if self.def_pxp is not None:
                pxp[i] = np.ravel(self.def_pxp(t)[0])
            if self.def_pyp is not None:
                pyp[i] = np.ravel(self.def_pyp(t)[0])
        return dict(ysp=ysp, usp=usp, xsp=xsp, pxp=pxp, pyp=pyp)

    @property
    def has_model_params(self) -> bool:
        return self.def_px is not None or self.def_py is not None

    def horizon_params(self, t_k: float):
        """``p_xk[:, i] = def_px(t_k + i)``, ``p_yk[:, i] = def_py(t_k + i)`` for i = 0..N-1 - the reference's own indexing
        (time plus stage index, not stage index times h), MPC_code.py:492-497.  Returns (px [N, nx], py [N, ny]); zeros when a
        callback is absent."""
        px = np.zeros((self.N, self.nx)); py = np.zeros((self.N, self.ny))
        for i in range(self.N):
            if self.def_px is not None:
                px[i] = np.ravel(self.def_px(t_k + i)[0])
            if self.def_py is not None:
                py[i] = np.ravel(self.def_py(t_k + i)[0])
        return px, py

    # ------------------------------------------------------------------ plant
    @property
    def plant_is_linear(self) -> bool:
        return self.plant_fx_cont is None

    def plant_step(self, xp: np.ndarray, u: np.ndarray, t: float, pxp: np.ndarray) -> np.ndarray:
        """x_p(t+h) for a batch ``xp[B, nxp]``, ``u[B, nu]`` (MPC_code.py:813-816).

        Linear plant: ``Ap x + Bp u + pxp`` (Utilities.py:45-49).  Non-linear continuous plant (Utilities.py:58-82): the user
        function on the augmented state ``[x; t]`` (``dt/dt = 1``), integrated over ``h`` by ``casadi.tools.simpleRK(f, Mx)`` -
        ``Mx`` classical Runge-Kutta-4 steps of ``h / Mx`` with the inputs held - then ``+ pxp`` (LinPar)."""
        if self.plant_fx_cont is None:
            return xp @ self.Ap.T + u @ self.Bp.T + pxp
        x = np.ascontiguousarray(np.asarray(xp, dtype=np.float64).T)           # [nxp, B]: x[0] is a component over the batch
        uu = np.ascontiguousarray(np.asarray(u, dtype=np.float64).T)
        pp = np.broadcast_to(np.asarray(pxp, dtype=np.float64).reshape(self.nxp, -1), x.shape)
        zero = np.zeros_like(pp)
        f = lambda xx, tt: np.asarray(self.plant_fx_cont(xx, tt, uu, pp, zero), dtype=np.float64)
        dt = self.h / self.plant_Mx
        tt = float(t)
        for _ in range(self.plant_Mx):
            k1 = f(x, tt)
            k2 = f(x + 0.5 * dt * k1, tt + 0.5 * dt)
            k3 = f(x + 0.5 * dt * k2, tt + 0.5 * dt)
            k4 = f(x + dt * k3, tt + dt)
            x = x + (dt / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
            tt += dt
        return np.ascontiguousarray(x.T) + np.asarray(pxp, dtype=np.float64)

    # ------------------------------------------------------------------ derived
    @property
    def nw(self) -> int:          # MPC_code.py:52
        return self.nx * (self.N + 1) + self.nu * self.N

    def aug_estimator_matrices(self):
        """[[A,Bd],[0,I]] and [C,Cd] - what kalman() re-derives every step (Estimator.py:288-291)."""
        n, nd = self.nx, self.nd
        Aa = np.eye(n + nd)
        Aa[:n, :n] = self.A
        Aa[:n, n:] = self.Bd
        Ca = np.hstack([self.C, self.Cd])
        return Aa, Ca


def _has(ns, name):
    """'name' in locals() in the reference: defaults count as defined (they are star-imported first)."""
    return name in ns and not name.startswith("__")


def problem_from_namespace(ns: Dict[str, Any], name: str = "") -> LinearMPCProblem:
    """Classify an Ex-file namespace and emit the numeric descriptor (or raise)."""
    for bad in ("User_fxm_Cont", "User_fxm_Dis", "User_fym", "User_fxp_Dis", "User_fyp",
                "User_fobj_Cont", "User_fobj_Dis", "User_fobj_Coll", "User_fssobj", "User_vfin",
                "User_h_eq", "User_g_ineq_SS", "User_h_eq_SS", "r_x", "rss_y",
                "def_pxmp", "def_pymp", "R_wn", "G_wn"):
        if _has(ns, bad) and ns[bad] is not None:
            raise UnsupportedProblem(f"'{bad}' is outside the batched linear hot path (later scope row)")
    for flag in ("ssjacid", "StateFeedback", "Fp_nominal", "Adaptation", "Collocation",
                 "mhe", "ekf", "estimating", "ContForm", "DUFormEcon"):
        if ns.get(flag, False) is True:
            raise UnsupportedProblem(f"flag {flag}=True is outside the batched linear hot path")
    if not ns.get("LinPar", True):
        raise UnsupportedProblem("LinPar=False")
    # plant only: the controller stays linear.  The reference tests 'Ap' first and falls through to the user plant only when it
    # is absent (MPC_code.py:176-199): an Ex-file that defines both simulates the linear plant
    nl_plant = _has(ns, "User_fxp_Cont") and ns["User_fxp_Cont"] is not None and not (_has(ns, "Ap") and ns["Ap"] is not None)
    if (ns.get("dmin") is None) != (ns.get("dmax") is None):
        raise UnsupportedProblem("dmin and dmax come as a pair (the reference saturates dhat with both, MPC_code.py:655-668)")
    for req in ("A", "B", "C", "Cp", "Q", "Qss", "N", "h", "Nsim", "x", "u", "y", "d", "xp") + (() if nl_plant else ("Ap", "Bp")):
        if not _has(ns, req):
            raise UnsupportedProblem(f"'{req}' missing: not a matrix-defined linear example")

    nx, nu, ny = ns["x"].size1(), ns["u"].size1(), ns["y"].size1()
    nd, nxp = ns["d"].size1(), ns["xp"].size1()
    N, h, Nsim = int(ns["N"]), float(ns["h"]), int(ns["Nsim"])

    A = _mat(ns["A"], nx, nx, "A")
    B = _mat(ns["B"], nx, nu, "B")
    C = _mat(ns["C"], ny, nx, "C")
    offree = ns.get("offree", "no")
    if offree == "lin":
        Bd = _mat(ns["Bd"], nx, nd, "Bd")
        Cd = _mat(ns["Cd"], ny, nd, "Cd")
    elif offree == "no":
        if nd != 0:
            # reference MPC_code.py:564-566 exits; a zero-size d is the only consistent choice
            raise UnsupportedProblem("nd != 0 but offree == 'no'")
        Bd, Cd = np.zeros((nx, 0)), np.zeros((ny, 0))
    else:
        raise UnsupportedProblem("offree='nl'")
    xlin = _vec(ns.get("xlin"), nx, 0.0) if _has(ns, "xlin") else np.zeros(nx)
    ulin = _vec(ns.get("ulin"), nu, 0.0) if _has(ns, "ulin") else np.zeros(nu)
    ylin = _vec(ns.get("ylin"), ny, 0.0) if _has(ns, "ylin") else np.zeros(ny)
    fx_const = xlin - A @ xlin - B @ ulin if _has(ns, "xlin") else np.zeros(nx)
    # Utilities.py:210-227: ylin alone -> C x + ylin ; xlin and ylin -> C (x-xlin) + ylin ; xlin alone -> C x
    if _has(ns, "ylin"):
        fy_const = ylin - (C @ xlin if _has(ns, "xlin") else 0.0)
    else:
        fy_const = np.zeros(ny)

    if nl_plant:         # MPC_code.py:176-199: a user plant takes the place of Ap, Bp (zeros here: only plant_step uses them)
        Ap, Bp = np.zeros((nxp, nxp)), np.zeros((nxp, nu))
    else:
        Ap = _mat(ns["Ap"], nxp, nxp, "Ap")
        Bp = _mat(ns["Bp"], nxp, nu, "Bp")
    Cp = _mat(ns["Cp"], ny, nxp, "Cp")

    Q = _mat(ns["Q"], nx, nx, "Q")
    if _has(ns, "R"):
        R, DUForm = _mat(ns["R"], nu, nu, "R"), False
    elif _has(ns, "S"):
        R, DUForm = _mat(ns["S"], nu, nu, "S"), True          # MPC_code.py:237-239,253-255
    else:
        raise UnsupportedProblem("Q given without R or S")
    P = scla.solve_discrete_are(A, B, Q, R)                   # Utilities.py:409
    P = 0.5 * (P + P.T)

    Qss = _mat(ns["Qss"], ny, ny, "Qss")
    if _has(ns, "Rss"):
        Rss, DUssForm = _mat(ns["Rss"], nu, nu, "Rss"), False
    elif _has(ns, "Sss"):
        Rss, DUssForm = _mat(ns["Sss"], nu, nu, "Sss"), True  # MPC_code.py:216-218
    else:
        raise UnsupportedProblem("Qss given without Rss or Sss")

    def pick(base, suffix, n, fill):
        v = ns.get(base + suffix)
        if v is None:
            v = ns.get(base)
        return _vec(v, n, fill)

    y_dyn_lo = ns.get("ymin_dyn") if ns.get("ymin_dyn") is not None else ns.get("ymin")
    y_dyn_hi = ns.get("ymax_dyn") if ns.get("ymax_dyn") is not None else ns.get("ymax")
    y_bounded = not (y_dyn_lo is None and y_dyn_hi is None)   # Control_Calc.py:60-63
    # soft constraints (Control_Calc.py:39-40,186-192,228-239): ONE slack vector [sl_ub; sl_lb] >= 0 for all stages, weight Ws in every stage's cost.  The reference
    # sizes it by Ws (MPC_code.py:55-57: ns = Ws.shape[0]); without user constraint rows - not carried on this path - that is 2 ny
    slacks = bool(ns.get("slacks", False))
    Ws = None
    if slacks:
        if not _has(ns, "Ws") or ns["Ws"] is None:
            raise UnsupportedProblem("slacks = True needs the slack weight Ws (MPC_code.py:55-57)")
        Ws = _mat(ns["Ws"], 2 * ny, 2 * ny, "Ws")
        if not y_bounded:
            raise UnsupportedProblem("slacks = True without output bounds: the slack vector would be unused and unbounded")
        if ns.get("TermCons", False) or _has(ns, "def_px") and ns["def_px"] is not None or _has(ns, "def_py") and ns["def_py"] is not None:
            raise UnsupportedProblem("soft constraints together with a terminal equality or horizon parameters are not carried")
    du_bounded = ns.get("Dumin") is not None or ns.get("Dumax") is not None      # DuFree False, Control_Calc.py:64-67

    if ns.get("kal", False):
        est = "kal"
    elif ns.get("kalss", False) or ns.get("lue", False):
        est = "kalss"                                          # MPC_code.py:577-581
    else:
        est = "none"
    if est != "none" and offree == "no":
        raise UnsupportedProblem("estimator without disturbance model")
    Q_kf = R_kf = P0 = K = None
    nxd = nx + nd
    if est == "kal":
        Q_kf = _mat(ns["Q_kf"], nxd, nxd, "Q_kf")
        R_kf = _mat(ns["R_kf"], ny, ny, "R_kf")
        P0 = _mat(ns["P0"], nxd, nxd, "P0") if _has(ns, "P0") else np.zeros((nxd, nxd))  # MPC_code.py:455-458
    elif est == "kalss":
        if ns.get("kalss", False):
            # Estimator.py:189-223: DARE on the augmented pair, K = P C'(C P C' + R)^-1
            Aa = np.eye(nxd); Aa[:nx, :nx] = A; Aa[:nx, nx:] = Bd
            Ca = np.hstack([C, Cd])
            Qe = _mat(ns["Q_kf"], nxd, nxd, "Q_kf"); Re = _mat(ns["R_kf"], ny, ny, "R_kf")
            Pe = scla.solve_discrete_are(Aa.T, Ca.T, Qe, Re)
            K = Pe @ Ca.T @ np.linalg.inv(Ca @ Pe @ Ca.T + Re)
        else:
            K = _mat(ns["K"], nxd, ny, "K")

    Gx = Gu = Gd = g0 = None
    if _has(ns, "User_g_ineq") and ns["User_g_ineq"] is not None:
        Gx, Gu, Gd, g0 = _affine_user_rows(ns["User_g_ineq"], nx, nu, ny, nd, C, Cd, fy_const)
        if slacks or ns.get("TermCons", False) or (_has(ns, "def_px") and ns["def_px"] is not None) or (_has(ns, "def_py") and ns["def_py"] is not None):
            raise UnsupportedProblem("User_g_ineq together with slacks, a terminal equality or horizon parameters is not carried")
    prob = LinearMPCProblem(
        nx=nx, nu=nu, ny=ny, nd=nd, nxp=nxp, N=N, h=h, Nsim=Nsim,
        A=A, B=B, C=C, Bd=Bd, Cd=Cd, fx_const=fx_const, fy_const=fy_const,
        Ap=Ap, Bp=Bp, Cp=Cp, Q=Q, R=R, DUForm=DUForm, P=P, Qss=Qss, Rss=Rss, DUssForm=DUssForm,
        umin=pick("umin", "_dyn", nu, -INF), umax=pick("umax", "_dyn", nu, INF),
        xmin=pick("xmin", "_dyn", nx, -INF), xmax=pick("xmax", "_dyn", nx, INF),
        ymin=pick("ymin", "_dyn", ny, -INF), ymax=pick("ymax", "_dyn", ny, INF), y_bounded=y_bounded,
        umin_ss=pick("umin", "_ss", nu, -INF), umax_ss=pick("umax", "_ss", nu, INF),
        xmin_ss=pick("xmin", "_ss", nx, -INF), xmax_ss=pick("xmax", "_ss", nx, INF),
        ymin_ss=pick("ymin", "_ss", ny, -INF), ymax_ss=pick("ymax", "_ss", ny, INF),
        dmin=None if ns.get("dmin") is None else _vec(ns["dmin"], nd, -INF),
        dmax=None if ns.get("dmax") is None else _vec(ns["dmax"], nd, INF),
        Dumin=_vec(ns.get("Dumin"), nu, -INF) if du_bounded else None, Dumax=_vec(ns.get("Dumax"), nu, INF) if du_bounded else None,
        estimator=est, Q_kf=Q_kf, R_kf=R_kf, P0=P0, K=K,
        x0_p=_vec(ns["x0_p"], nxp, 0.0), x0_m=_vec(ns["x0_m"], nx, 0.0), u0=_vec(ns["u0"], nu, 0.0),
        dhat0=_vec(ns.get("dhat0"), nd, 0.0) if _has(ns, "dhat0") else np.zeros(nd),
        max_iter=int(ns.get("Sol_itmax", 100)),
        defSP=ns.get("defSP"), def_pxp=ns.get("def_pxp"), def_pyp=ns.get("def_pyp"),
        name=name or str(ns.get("__name__", "")),
        plant_fx_cont=ns["User_fxp_Cont"] if nl_plant else None, plant_Mx=int(ns.get("Mx", 10)),
        TermCons=bool(ns.get("TermCons", False)), def_px=ns.get("def_px"), def_py=ns.get("def_py"),
        slacks=slacks, Ws=Ws, Gx=Gx, Gu=Gu, Gd=Gd, g0=g0,
    )
    return prob


def _affine_user_rows(fn, nx, nu, ny, nd, C, Cd, fy_const):
    """User_g_ineq(x, u, y, d, t, px, py) <= 0 (Control_Calc.py:94-100; a row per stage k = 0..N-1 on (X[k], U[k], Y_k), :132-147) for the LINEAR hot path: the rows have
    to be affine in (x, u, y, d) and independent of t, px, py (LinPar).  Traced once with symbols, then read off numerically - value at the origin, one unit vector per
    argument - and checked at random points; y_k = C x_k + Cd d + fy_const is substituted.  Returns (Gx, Gu, Gd, g0)."""
    from . import symtrace as st
    vx, vu, vy, vd, vt = st.symvec("x", nx), st.symvec("u", nu), st.symvec("y", ny), st.symvec("d", nd), st.Sym.var("t")
    col, zero = st.SymMat.col, (lambda n: st.SymMat.zeros(n))
    try:
        rows = st.flatten(fn(col(vx), col(vu), col(vy), col(vd), vt, zero(nx), zero(ny)))
    except Exception as e:      # noqa: BLE001
        raise UnsupportedProblem(f"User_g_ineq cannot be traced: {e}") from e
    ng = len(rows)
    if ng < 1 or ng > 4:
        raise UnsupportedProblem("User_g_ineq: between one and four rows are carried")
    nz = nx + nu + ny + nd
    names = [f"x[{i}]" for i in range(nx)] + [f"u[{i}]" for i in range(nu)] + [f"y[{i}]" for i in range(ny)] + [f"d[{i}]" for i in range(nd)]

    def ev(z, t=0.0):
        vals = {n: float(v) for n, v in zip(names, z)}; vals["t"] = float(t)
        return np.array([float(v) for v in st.evaluate(rows, vals)])
    c0 = ev(np.zeros(nz))
    J = np.stack([ev(np.eye(nz)[j]) - c0 for j in range(nz)], axis=1)
    rng = np.random.default_rng(0)
    for _ in range(4):
        z = rng.normal(size=nz) * 3.0
        if np.abs(ev(z, t=rng.normal()) - (c0 + J @ z)).max() > 1e-9 * (1.0 + np.abs(J).max() * 10.0):
            raise UnsupportedProblem("User_g_ineq: only rows that are affine in (x, u, y, d) and independent of t are carried on the linear path")
    Jx, Ju, Jy, Jd = J[:, :nx], J[:, nx:nx + nu], J[:, nx + nu:nx + nu + ny], J[:, nx + nu + ny:]
    return Jx + Jy @ C, Ju, (Jd + Jy @ Cd if nd else np.zeros((ng, 0))), c0 + Jy @ fy_const
