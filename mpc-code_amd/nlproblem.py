"""Problem descriptor of the non-linear MPC path (BASELINE config 3: ``Ex_NMPC.py``).

The reference builds a non-linear example symbolically: ``defF_model`` integrates the Ex-file's continuous model
``User_fxm_Cont(x,u,d,t,px)`` with ``Mx`` classical Runge-Kutta steps per sampling interval (``Utilities.py:157-183``,
``casadi.simpleRK``), takes ``User_fym`` as output map, ``defF_p`` does the same for the plant (``:58-82``), and IPOPT
differentiates the unrolled graphs.  Here the same Ex-file functions are *traced* (:mod:`symtrace`): the expression
DAGs of model, output, plant and their Jacobians are what the device code generator and the host-side simulation use.

Covered: continuous-time model and plant (``User_fxm_Cont``, ``User_fym``, ``User_fxp_Cont``, ``User_fyp``), disturbance
entering the model non-linearly (``offree = "nl"``: the estimate ``dhat`` is an argument of the model, ``Utilities.py:126-129``),
quadratic stage cost on ``x - xs`` and ``u - us`` (``Q``, ``R``; no terminal cost: ``defVfin`` returns 0 without ``A``,
``Utilities.py:398-399``), quadratic target cost (``Qss``, ``Rss``), bounds on u, x, y (outputs that are single states),
saturation of ``dhat``, extended Kalman filter (``Estimator.py:313-386``).  Anything else raises ``UnsupportedProblem``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from . import symtrace as st
from .problem import UnsupportedProblem, _mat, _vec

INF = float("inf")


@dataclass(repr=False)
class NonlinearMPCProblem:
    nx: int
    nu: int
    ny: int
    nd: int
    nxp: int
    N: int
    h: float
    Nsim: int
    Mx: int
    # traced expressions in the variables x[i], u[i], d[i], t (model) / xp[i], u[i], t (plant)
    f: List[st.Sym] = None          # model: dx/dt
    hy: List[st.Sym] = None         # model output
    fp: List[st.Sym] = None         # plant: dxp/dt
    hp: List[st.Sym] = None         # plant output
    Q: np.ndarray = None
    R: np.ndarray = None
    Qss: np.ndarray = None
    Rss: np.ndarray = None
    umin: np.ndarray = None
    umax: np.ndarray = None
    xmin: np.ndarray = None
    xmax: np.ndarray = None
    ymin: np.ndarray = None
    ymax: np.ndarray = None
    umin_ss: np.ndarray = None
    umax_ss: np.ndarray = None
    xmin_ss: np.ndarray = None
    xmax_ss: np.ndarray = None
    ymin_ss: np.ndarray = None
    ymax_ss: np.ndarray = None
    dmin: Optional[np.ndarray] = None
    dmax: Optional[np.ndarray] = None
    Q_kf: np.ndarray = None
    R_kf: np.ndarray = None
    P0: np.ndarray = None
    x0_p: np.ndarray = None
    x0_m: np.ndarray = None
    u0: np.ndarray = None
    dhat0: np.ndarray = None
    max_iter: int = 100
    defSP: Optional[Callable] = None
    R_wn: Optional[np.ndarray] = None      # covariance of the white noise the reference adds to every measurement (MPC_code.py:538-541): run_nmpc_stepwise(noise_seed=...)
    name: str = ""
    ycols: List[int] = None         # output row i is state ycols[i] (bounded outputs are boxes on states)
    discrete: bool = False          # f is the discrete map Fx (User_fxm_Dis, Utilities.py:186-198), not a right-hand side to integrate
    plant_discrete: bool = False
    offree: str = "nl"              # "nl": d is an argument of the model; "lin": + Bd d, + Cd d (folded into f / hy)
    Bd: Optional[np.ndarray] = None
    Cd: Optional[np.ndarray] = None
    estimator: str = "ekf"          # "ekf" | "lue" (fixed gain K)
    K: Optional[np.ndarray] = None
    DUForm: bool = False            # R weighs u_k - u_{k-1} (the Ex-file's S)
    DUssForm: bool = False          # Rss weighs us - us_prev (Sss)
    Dumin: Optional[np.ndarray] = None
    Dumax: Optional[np.ndarray] = None
    Pf: np.ndarray = None           # terminal weight: Vfin = 1/2 dx' Pf dx (User_vfin when it is such a form, else zero)
    def_pxp: Optional[Callable] = None
    def_pyp: Optional[Callable] = None
    funcs: Dict[str, Any] = field(default_factory=dict)      # the Ex-file's own Python functions (host-side checks)

    def __repr__(self):      # the traced expressions print as whole trees: keep them out of reprs (pytest renders them on a failure)
        return f"NonlinearMPCProblem({self.name!r}, nx={self.nx}, nu={self.nu}, ny={self.ny}, nd={self.nd}, N={self.N}, discrete={self.discrete})"

    # ------------------------------------------------------------------ derived symbolic Jacobians
    def __post_init__(self):
        self.vx, self.vu, self.vd = st.symvec("x", self.nx), st.symvec("u", self.nu), st.symvec("d", self.nd)
        self.vxp = st.symvec("xp", self.nxp)
        self.vt = st.Sym.var("t")
        if self.f is not None:
            self.f_x = st.jacobian(self.f, self.vx); self.f_u = st.jacobian(self.f, self.vu); self.f_d = st.jacobian(self.f, self.vd)
            self.h_x = st.jacobian(self.hy, self.vx); self.h_d = st.jacobian(self.hy, self.vd)

    @property
    def nw(self) -> int:
        return self.nx * (self.N + 1) + self.nu * self.N

    def schedules(self, nsteps: int, k0: int = 0) -> Dict[str, np.ndarray]:
        ysp = np.zeros((nsteps, self.ny)); usp = np.zeros((nsteps, self.nu)); xsp = np.zeros((nsteps, self.nx))
        pxp = np.zeros((nsteps, self.nxp)); pyp = np.zeros((nsteps, self.ny))
        for i in range(nsteps):
            t = (k0 + i) * self.h
            if self.defSP is not None:
                a, b, c = self.defSP(t)
                ysp[i], usp[i], xsp[i] = np.ravel(a), np.ravel(b), np.ravel(c)
            if self.def_pxp is not None:
                pxp[i] = np.ravel(self.def_pxp(t)[0])          # plant disturbances, MPC_code.py:512-515
            if self.def_pyp is not None:
                pyp[i] = np.ravel(self.def_pyp(t)[0])
        return dict(ysp=ysp, usp=usp, xsp=xsp, pxp=pxp, pyp=pyp)

    # ------------------------------------------------------------------ host-side evaluation (NumPy, batch [B, .])
    def _vals(self, x=None, u=None, d=None, t=0.0, xp=None):
        v = {"t": t}
        if x is not None: v.update({f"x[{i}]": x[..., i] for i in range(self.nx)})
        if xp is not None: v.update({f"xp[{i}]": xp[..., i] for i in range(self.nxp)})
        if u is not None: v.update({f"u[{i}]": u[..., i] for i in range(self.nu)})
        if d is not None: v.update({f"d[{i}]": d[..., i] for i in range(self.nd)})
        return v

    @staticmethod
    def _stack(vals, shape):
        return np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), shape) for v in vals], axis=-1)

    def plant_step(self, xp, u, t):
        """x_p(t + h): ``Mx`` classical RK4 steps of the plant with time as a state (Utilities.py:58-82)."""
        xp = np.array(xp, dtype=np.float64); dt = self.h / self.Mx
        f = lambda x_, t_: self._stack(st.evaluate(self.fp, self._vals(xp=x_, u=u, t=t_)), x_.shape[:-1])
        if self.plant_discrete:                      # User_fxp_Dis is the step itself (Utilities.py:84-87)
            return f(xp, t)
        for s in range(self.Mx):
            ts = t + s * dt
            k1 = f(xp, ts); k2 = f(xp + 0.5 * dt * k1, ts + 0.5 * dt); k3 = f(xp + 0.5 * dt * k2, ts + 0.5 * dt); k4 = f(xp + dt * k3, ts + dt)
            xp = xp + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        return xp

    def plant_output(self, xp, u, t):
        return self._stack(st.evaluate(self.hp, self._vals(xp=xp, u=u, t=t)), np.shape(xp)[:-1])

    def model_output(self, x, u, d, t):
        """Fy_model(x, u, d, t) (Utilities.py:208-244; with offree = 'lin' the + Cd d is part of ``hy``)."""
        return self._stack(st.evaluate(self.hy, self._vals(x=x, u=u, d=d if self.nd else None, t=t)), np.shape(x)[:-1])


def _trace(fn, args):
    try:
        return st.flatten(fn(*args))
    except Exception as e:      # noqa: BLE001 - whatever the user function trips over
        raise UnsupportedProblem(f"{getattr(fn, '__name__', fn)} cannot be traced: {e}") from e


def nl_problem_from_namespace(ns: Dict[str, Any], name: str = "") -> NonlinearMPCProblem:
    """Classify a non-linear Ex-file namespace (reference MPC_code.py:84-257 probes) and trace its functions."""
    has = lambda k: k in ns and ns[k] is not None and not k.startswith("__")
    for bad in ("User_fobj_Cont", "User_fobj_Dis", "User_fobj_Coll", "User_fssobj",
                "User_g_ineq", "User_h_eq", "User_g_ineq_SS", "User_h_eq_SS", "r_x", "rss_y", "def_px", "def_py", "def_pxmp", "def_pymp", "A", "G_wn"):
        if has(bad):
            raise UnsupportedProblem(f"'{bad}' is outside the non-linear path built so far")
    if has("R_wn"):
        # the reference adds UNSEEDED Gaussian noise to every measurement (MPC_code.py:538-541): a run is not reproducible even there.
        # The batched loops are deterministic and run noise-free unless asked otherwise - said out loud, not silently (Ex_NMPC.py ships R_wn): with noise_seed = s
        # nmpc.run_nmpc_closed_loop (draws on the device, nmpc_set_noise) and nmpc.run_nmpc_stepwise (the measurement is the caller's) add seeded noise of this covariance
        import warnings
        warnings.warn("R_wn: the measurement noise of the example (unseeded in the reference, MPC_code.py:538-541) is simulated only on request: the loops run noise-free unless called with noise_seed=...", UserWarning, stacklevel=3)
    for flag in ("ssjacid", "StateFeedback", "Fp_nominal", "Adaptation", "Collocation", "slacks", "TermCons", "mhe", "ContForm",
                 "DUFormEcon", "kal", "kalss", "estimating"):
        if ns.get(flag, False) is True:
            raise UnsupportedProblem(f"flag {flag}=True is outside the non-linear path built so far")
    if ns.get("LinPar", True) is not True:
        raise UnsupportedProblem("LinPar = False is outside the non-linear path built so far")
    if (ns.get("dmin") is None) != (ns.get("dmax") is None):
        raise UnsupportedProblem("dmin and dmax have to come together (the estimate is clipped to both, MPC_code.py:657-664)")
    discrete = has("User_fxm_Dis")
    plant_discrete = has("User_fxp_Dis")
    for req in ("User_fxm_Dis" if discrete else "User_fxm_Cont", "User_fym", "User_fxp_Dis" if plant_discrete else "User_fxp_Cont", "User_fyp",
                "Q", "Qss", "N", "h", "Nsim", "x", "u", "y", "d", "xp"):
        if not has(req):
            raise UnsupportedProblem(f"'{req}' missing: not a non-linear example with a quadratic cost")
    offree = ns.get("offree", "no")
    if offree not in ("nl", "lin"):
        raise UnsupportedProblem("the non-linear path needs a disturbance model (offree = 'nl' or 'lin')")
    if offree == "lin" and not discrete:
        raise UnsupportedProblem("offree = 'lin' with a continuous-time model is outside the non-linear path built so far")
    if ns.get("ekf", False):
        est = "ekf"
        for req in ("Q_kf", "R_kf"):
            if not has(req):
                raise UnsupportedProblem(f"'{req}' missing for the extended Kalman filter")
    elif ns.get("lue", False):
        est = "lue"                                   # xi+ = xi + K (y - yhat), Estimator.py:231-261
        if not has("K"):
            raise UnsupportedProblem("lue = True without K")
    else:
        raise UnsupportedProblem("the non-linear path needs ekf = True or lue = True")
    nx, nu, ny, nd, nxp = ns["x"].size1(), ns["u"].size1(), ns["y"].size1(), ns["d"].size1(), ns["xp"].size1()
    vx, vu, vd, vxp, vt = st.symvec("x", nx), st.symvec("u", nu), st.symvec("d", nd), st.symvec("xp", nxp), st.Sym.var("t")
    col = st.SymMat.col
    zero = lambda n: st.SymMat.zeros(n)
    f = _trace(ns["User_fxm_Dis" if discrete else "User_fxm_Cont"], (col(vx), col(vu), col(vd), vt, zero(nx)))     # Utilities.py:160,188
    hy = _trace(ns["User_fym"], (col(vx), col(vu), col(vd), vt, zero(ny)))                # Utilities.py:229
    fp = _trace(ns["User_fxp_Dis" if plant_discrete else "User_fxp_Cont"], (col(vxp), vt, col(vu), zero(nxp), zero(nxp)))  # Utilities.py:61,85
    hp = _trace(ns["User_fyp"], (col(vxp), col(vu), vt, zero(ny), zero(ny)))         # Utilities.py:94
    if len(f) != nx or len(hy) != ny or len(fp) != nxp or len(hp) != ny:
        raise UnsupportedProblem("a user function returns a vector of the wrong length")
    hy_user = list(hy)      # the user's map alone: with offree = 'lin' the + Cd d below is carried per instance by the kernels' output rows
    Bd = Cd = None
    if offree == "lin":        # Fx_model = F(x,u,d,t) + Bd d, Fy_model = h(x,u,d,t) + Cd d (Utilities.py:189-193,231-233)
        Bd, Cd = _mat(ns["Bd"], nx, nd, "Bd"), _mat(ns["Cd"], ny, nd, "Cd")
        f = [f[i] + sum((float(Bd[i, j]) * vd[j] for j in range(nd)), st.Sym.const(0.0)) for i in range(nx)]
        hy = [hy[i] + sum((float(Cd[i, j]) * vd[j] for j in range(nd)), st.Sym.const(0.0)) for i in range(ny)]
    if any(not e.is_const(0.0) for row in st.jacobian(hy, vu) for e in row):
        raise UnsupportedProblem("User_fym depends on u: the target and the OCP linearise the output map in x only")
    ycols = []
    hx = st.jacobian(hy, vx)
    for i in range(ny):      # bounded outputs have to be single states with unit gain (then their bounds are boxes on states)
        nz = [(j, e) for j, e in enumerate(hx[i]) if not e.is_const(0.0)]
        ycols.append(nz[0][0] if len(nz) == 1 and nz[0][1].is_const(1.0) else -1)

    def pick(base, suffix, n, fill):
        v = ns.get(base + suffix)
        if v is None:
            v = ns.get(base)
        return _vec(v, n, fill)
    y_lo, y_hi = pick("ymin", "_dyn", ny, -INF), pick("ymax", "_dyn", ny, INF)
    if any(c < 0 and (np.isfinite(y_lo[i]) or np.isfinite(y_hi[i])) for i, c in enumerate(ycols)):
        raise UnsupportedProblem("a bounded output is not a single state")
    for i, c in enumerate(ycols):      # a bounded row has to BE the state (its box is put on the state): nothing of d or a constant beside it
        if c >= 0 and (np.isfinite(y_lo[i]) or np.isfinite(y_hi[i])):
            others = [st.diff(hy_user[i], v) for j, v in enumerate(vx) if j != c] + [st.diff(hy_user[i], v) for v in vd]
            at0 = float(np.asarray(st.evaluate([hy_user[i]], {**{f"x[{j}]": 0.0 for j in range(nx)}, **{f"u[{j}]": 0.0 for j in range(nu)}, **{f"d[{j}]": 0.0 for j in range(nd)}, "t": 0.0})[0]))
            if any(not e.is_const(0.0) for e in others) or at0 != 0.0:
                raise UnsupportedProblem(f"bounded output {i} is state {c} plus something else (d, a constant): its bounds are not a box on the state")
    if has("R"):
        R, DUForm = _mat(ns["R"], nu, nu, "R"), False
    elif has("S"):
        R, DUForm = _mat(ns["S"], nu, nu, "S"), True          # cost on u_k - u_{k-1}, MPC_code.py:237-239
    else:
        raise UnsupportedProblem("Q given without R or S")
    if has("Rss"):
        Rss, DUssForm = _mat(ns["Rss"], nu, nu, "Rss"), False
    elif has("Sss"):
        Rss, DUssForm = _mat(ns["Sss"], nu, nu, "Sss"), True  # MPC_code.py:216-218
    else:
        Rss, DUssForm = np.zeros((nu, nu)), False
    du_bounded = has("Dumin") or has("Dumax")
    Pf = np.zeros((nx, nx))
    if has("User_vfin"):      # Vfin(dx, xs) with dx = X[N] - xs (Control_Calc.py:193-210): accepted when it is a quadratic form of dx alone
        vdx = st.symvec("dx", nx)
        v = ns["User_vfin"](col(vdx), col(st.symvec("xs", nx)))
        if isinstance(v, st.SymMat):
            v = v.a.ravel()[0]
        v = st.Sym.lift(v)
        g = [st.diff(v, w) for w in vdx]
        Hs = st.jacobian(g, vdx)
        if not all(e.is_const() for row in Hs for e in row) or any(abs(float(st.evaluate([e], {f"dx[{i}]": 0.0 for i in range(nx)} | {f"xs[{i}]": 1.0 for i in range(nx)})[0])) > 0 for e in g):
            raise UnsupportedProblem("User_vfin is not a quadratic form of x - xs")
        Pf = np.array([[e.val for e in row] for row in Hs], dtype=np.float64)
        Pf = 0.5 * (Pf + Pf.T)
    P0 = np.array(ns["P0"], dtype=np.float64) if has("P0") else np.zeros((nx + nd, nx + nd))
    return NonlinearMPCProblem(
        nx=nx, nu=nu, ny=ny, nd=nd, nxp=nxp, N=int(ns["N"]), h=float(ns["h"]), Nsim=int(ns["Nsim"]), Mx=int(ns.get("Mx", 10)),
        f=f, hy=hy, fp=fp, hp=hp, Q=_mat(ns["Q"], nx, nx, "Q"), R=R, Qss=_mat(ns["Qss"], ny, ny, "Qss"), Rss=Rss,
        umin=pick("umin", "_dyn", nu, -INF), umax=pick("umax", "_dyn", nu, INF), xmin=pick("xmin", "_dyn", nx, -INF), xmax=pick("xmax", "_dyn", nx, INF),
        ymin=y_lo, ymax=y_hi,
        umin_ss=pick("umin", "_ss", nu, -INF), umax_ss=pick("umax", "_ss", nu, INF), xmin_ss=pick("xmin", "_ss", nx, -INF), xmax_ss=pick("xmax", "_ss", nx, INF),
        ymin_ss=pick("ymin", "_ss", ny, -INF), ymax_ss=pick("ymax", "_ss", ny, INF),
        dmin=None if ns.get("dmin") is None else _vec(ns["dmin"], nd, -INF), dmax=None if ns.get("dmax") is None else _vec(ns["dmax"], nd, INF),
        Q_kf=_mat(ns["Q_kf"], nx + nd, nx + nd, "Q_kf") if est == "ekf" else None, R_kf=_mat(ns["R_kf"], ny, ny, "R_kf") if est == "ekf" else None, P0=P0,
        x0_p=_vec(ns["x0_p"], nxp, 0.0), x0_m=_vec(ns["x0_m"], nx, 0.0), u0=_vec(ns["u0"], nu, 0.0),
        dhat0=_vec(ns.get("dhat0"), nd, 0.0) if has("dhat0") else np.zeros(nd),
        max_iter=int(ns.get("Sol_itmax", 100)), defSP=ns.get("defSP"), R_wn=(_mat(ns["R_wn"], ny, ny, "R_wn") if has("R_wn") else None), name=name or str(ns.get("__name__", "")), ycols=ycols,
        funcs={k: ns[k] for k in ("User_fxm_Cont", "User_fxm_Dis", "User_fym", "User_fxp_Cont", "User_fxp_Dis", "User_fyp") if has(k)},
        discrete=discrete, plant_discrete=plant_discrete, offree=offree, Bd=Bd, Cd=Cd, estimator=est,
        K=_mat(ns["K"], nx + nd, ny, "K") if est == "lue" else None, DUForm=DUForm, DUssForm=DUssForm,
        Dumin=_vec(ns.get("Dumin"), nu, -INF) if du_bounded else None, Dumax=_vec(ns.get("Dumax"), nu, INF) if du_bounded else None, Pf=Pf,
        def_pxp=ns.get("def_pxp"), def_pyp=ns.get("def_pyp"),
    )
