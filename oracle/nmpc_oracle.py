"""ORACLE (test infrastructure, never shipped): the non-linear MPC loop of the reference, restated with NumPy.

PARITY UNPINNED against the reference's own solver: CasADi/IPOPT cannot run here and the reference ships no vectors
(SURVEY.md section 8c).  This file pins the GPU path to the *mathematics* of the reference's non-linear case instead:

* model / plant      ``defF_model`` / ``defF_p``: ``Mx`` classical RK4 steps of the Ex-file's continuous functions per sampling
                     interval, time carried as a state (``Utilities.py:157-183``, ``:58-82``; ``casadi.simpleRK``).  The
                     Ex-file is executed by the oracle's own loader (``load_problem`` -> ``exnum.py``) and its Python functions are
                     called on NumPy arrays - nothing of the product's loader, stand-ins, tracer or generated code;
* Jacobians          central finite differences of that discrete map (independent of the product's symbolic ones);
* estimator          ``ekf`` (``Estimator.py:313-386``) on ``[x; d]`` with ``d+ = d`` (``MPC_code.py:546-561``);
* target             the NLP of ``opt_ss`` (``Target_Calc.py:20-161``) for a non-linear model, by Newton-type SQP: every iteration
                     is the linear target QP of ``oracle/mpc_oracle.py:target_qp`` with the model linearised at the iterate;
* OCP                the NLP of ``opt_dyn`` (``Control_Calc.py:20-260``: multiple shooting, quadratic cost, no terminal cost -
                     ``Utilities.py:398-399`` - bounds on x_1..x_N, u, and the output rows at k = 0..N-1), by SQP with the exact
                     (Gauss-Newton = exact: the cost is quadratic) Hessian of the cost: each iteration a dense QP in opt_dyn's own
                     variable order, solved by ``mpc_oracle.qp_ipm_dense``; iterated to a fixed point, which is a KKT point of the
                     NLP whatever Hessian the iteration uses;
* loop               ``MPC_code.py:485-827``: measure, estimate, target (previous target kept when infeasible), OCP (shifted
                     warm start ``:764``, hold rule ``:804-805``), plant.

``kkt_nlp`` certifies a returned OCP point against the NLP itself (stationarity with finite-difference constraint Jacobians).
"""
from __future__ import annotations

import numpy as np

import exnum
import mpc_oracle as o

STATUS_SOLVED, STATUS_MAXITER, STATUS_INFEASIBLE = 0, 1, 2
INF = float("inf")


# ---------------------------------------------------------------------------------------------------
# the problem, read from the Ex-file by the oracle's own loader (exnum.py): nothing of the product's loader, tracer or stand-ins
# ---------------------------------------------------------------------------------------------------
class NlProblem:
    def __repr__(self):
        return f"oracle NlProblem({self.name!r}, nx={self.nx}, nu={self.nu}, ny={self.ny}, nd={self.nd}, N={self.N}, discrete={self.discrete})"

    def schedules(self, nsteps, k0=0):
        ysp = np.zeros((nsteps, self.ny)); usp = np.zeros((nsteps, self.nu)); xsp = np.zeros((nsteps, self.nx))
        pxp = np.zeros((nsteps, self.nxp)); pyp = np.zeros((nsteps, self.ny))
        for i in range(nsteps):
            t = (k0 + i) * self.h
            if self.defSP is not None:
                a, b, c = self.defSP(t)
                ysp[i], usp[i], xsp[i] = np.ravel(a), np.ravel(b), np.ravel(c)
            if self.def_pxp is not None:
                pxp[i] = np.ravel(self.def_pxp(t)[0])      # MPC_code.py:512-515
            if self.def_pyp is not None:
                pyp[i] = np.ravel(self.def_pyp(t)[0])
        return dict(ysp=ysp, usp=usp, xsp=xsp, pxp=pxp, pyp=pyp)


def _vec(v, n, fill):
    return np.full(n, fill, dtype=np.float64) if v is None else np.asarray(v, dtype=np.float64).reshape(n)


def load_problem(path, overrides=None):
    """The namespace of a non-linear tracking example as numbers and plain Python functions (reference MPC_code.py:31-60,84-257 probes;
    defaults Default_Values.py:16-131)."""
    ns = exnum.load(path, overrides)
    has = lambda k: ns.get(k) is not None
    p = NlProblem()
    p.name = ns["__name__"]
    p.nx, p.nu, p.ny, p.nd, p.nxp = (ns[k].size1() for k in ("x", "u", "y", "d", "xp"))
    p.N, p.h, p.Nsim, p.Mx = int(ns["N"]), float(ns["h"]), int(ns["Nsim"]), int(ns.get("Mx", 10))
    p.discrete, p.plant_discrete = has("User_fxm_Dis"), has("User_fxp_Dis")
    p.funcs = {k: ns[k] for k in ("User_fxm_Cont", "User_fxm_Dis", "User_fym", "User_fxp_Cont", "User_fxp_Dis", "User_fyp", "User_vfin") if has(k)}
    p.offree = ns.get("offree", "no")
    assert p.offree in ("nl", "lin")
    p.Bd = np.asarray(ns["Bd"], dtype=float).reshape(p.nx, p.nd) if p.offree == "lin" else None
    p.Cd = np.asarray(ns["Cd"], dtype=float).reshape(p.ny, p.nd) if p.offree == "lin" else None
    p.Q = np.asarray(ns["Q"], dtype=float).reshape(p.nx, p.nx)
    p.DUForm = not has("R")                                   # S instead of R: cost on input moves (MPC_code.py:237-239)
    p.R = np.asarray(ns["S"] if p.DUForm else ns["R"], dtype=float).reshape(p.nu, p.nu)
    p.Qss = np.asarray(ns["Qss"], dtype=float).reshape(p.ny, p.ny)
    p.DUssForm = (not has("Rss")) and has("Sss")              # :216-218
    p.Rss = np.asarray(ns["Sss"] if p.DUssForm else (ns["Rss"] if has("Rss") else np.zeros((p.nu, p.nu))), dtype=float).reshape(p.nu, p.nu)
    pick = lambda b, sfx, n, f: _vec(ns.get(b + sfx) if ns.get(b + sfx) is not None else ns.get(b), n, f)
    for b, n in (("u", p.nu), ("x", p.nx), ("y", p.ny)):
        setattr(p, b + "min", pick(b + "min", "_dyn", n, -INF)); setattr(p, b + "max", pick(b + "max", "_dyn", n, INF))
        setattr(p, b + "min_ss", pick(b + "min", "_ss", n, -INF)); setattr(p, b + "max_ss", pick(b + "max", "_ss", n, INF))
    p.dmin = None if not has("dmin") else _vec(ns["dmin"], p.nd, -INF)
    p.dmax = None if not has("dmax") else _vec(ns["dmax"], p.nd, INF)
    p.Dumin = _vec(ns.get("Dumin"), p.nu, -INF) if (has("Dumin") or has("Dumax")) else None
    p.Dumax = _vec(ns.get("Dumax"), p.nu, INF) if (has("Dumin") or has("Dumax")) else None
    p.estimator = "ekf" if ns.get("ekf", False) else "lue"
    ne = p.nx + p.nd
    p.Q_kf = np.asarray(ns["Q_kf"], dtype=float).reshape(ne, ne) if p.estimator == "ekf" else None
    p.R_kf = np.asarray(ns["R_kf"], dtype=float).reshape(p.ny, p.ny) if p.estimator == "ekf" else None
    p.K = np.asarray(ns["K"], dtype=float).reshape(ne, p.ny) if p.estimator == "lue" else None
    p.P0 = np.asarray(ns["P0"], dtype=float).reshape(ne, ne) if has("P0") else np.zeros((ne, ne))
    p.x0_p, p.x0_m, p.u0 = _vec(ns["x0_p"], p.nxp, 0.0), _vec(ns["x0_m"], p.nx, 0.0), _vec(ns["u0"], p.nu, 0.0)
    p.dhat0 = _vec(ns.get("dhat0"), p.nd, 0.0)
    p.max_iter = int(ns.get("Sol_itmax", 100))
    p.defSP, p.def_pxp, p.def_pyp = ns.get("defSP"), ns.get("def_pxp"), ns.get("def_pyp")
    p.Pf = np.zeros((p.nx, p.nx))
    if has("User_vfin"):      # Vfin(dx, xs), a quadratic form of dx = X[N] - xs (Control_Calc.py:193-210): its Hessian by central differences (exact for a quadratic)
        vf = lambda dx: float(np.real(np.ravel(ns["User_vfin"](_sm(dx), _sm(np.ones(p.nx))))[0]))
        e = np.eye(p.nx)
        H = np.array([[(vf(e[i] + e[j]) - vf(e[i] - e[j]) - vf(e[j] - e[i]) + vf(-e[i] - e[j])) / 4.0 for j in range(p.nx)] for i in range(p.nx)])
        p.Pf = 0.5 * (H + H.T)
    return p


# ---------------------------------------------------------------------------------------------------
# discrete maps from the Ex-file's continuous functions
# ---------------------------------------------------------------------------------------------------
def _col(v, n):
    return np.asarray(v, dtype=np.float64).reshape(n)


def _sm(v):
    """The container the Ex-file functions index, slice and assign into (numbers inside)."""
    return exnum.NumMat.col([float(a) for a in np.ravel(v)])


def _rk4(f, x, t, h, Mx):
    dt = h / Mx
    x = np.array(x, dtype=np.float64)
    for s in range(Mx):
        ts = t + s * dt
        k1 = f(x, ts); k2 = f(x + 0.5 * dt * k1, ts + 0.5 * dt); k3 = f(x + 0.5 * dt * k2, ts + 0.5 * dt); k4 = f(x + dt * k3, ts + dt)
        x = x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    return x


def model_fx(p, x, u, d, t=0.0):
    """Fx_model(x,u,h,d,t): Mx RK4 steps of User_fxm_Cont over h (Utilities.py:160-172), or the user's discrete map
    (User_fxm_Dis, :186-198); + Bd d when offree = 'lin' (:189-190)."""
    if getattr(p, "discrete", False):
        out = _col(p.funcs["User_fxm_Dis"](_sm(x), _sm(u), _sm(d), t, _sm(np.zeros(p.nx))), p.nx)
    else:
        out = _rk4(lambda x_, t_: _col(p.funcs["User_fxm_Cont"](x_, u, d, t_, np.zeros(p.nx)), p.nx), x, t, p.h, p.Mx)
    return out + p.Bd @ d if getattr(p, "offree", "nl") == "lin" else out


def model_fy(p, x, u, d, t=0.0):
    out = _col(p.funcs["User_fym"](_sm(x), _sm(u), _sm(d), t, _sm(np.zeros(p.ny))), p.ny)
    return out + p.Cd @ d if getattr(p, "offree", "nl") == "lin" else out


def plant_fx(p, xp, u, t, pxp=None):
    if getattr(p, "plant_discrete", False):
        out = _col(p.funcs["User_fxp_Dis"](_sm(xp), t, _sm(u), _sm(np.zeros(p.nxp)), _sm(np.zeros(p.nxp))), p.nxp)
    else:
        out = _rk4(lambda x_, t_: _col(p.funcs["User_fxp_Cont"](x_, t_, u, np.zeros(p.nxp), np.zeros(p.nxp)), p.nxp), xp, t, p.h, p.Mx)
    return out if pxp is None else out + pxp                     # LinPar: + pxp (Utilities.py:78-82,86-87)


def plant_fy(p, xp, u, t, pyp=None):
    out = _col(p.funcs["User_fyp"](_sm(xp), _sm(u), t, _sm(np.zeros(p.ny)), _sm(np.zeros(p.ny))), p.ny)
    return out if pyp is None else out + pyp


def _fd(fun, v, rel=1e-6):
    v = np.asarray(v, dtype=np.float64)
    f0 = fun(v)
    J = np.zeros((f0.size, v.size))
    for j in range(v.size):
        e = np.zeros(v.size); e[j] = rel * max(1.0, abs(v[j]))
        J[:, j] = (fun(v + e) - fun(v - e)) / (2.0 * e[j])
    return J


def linearize(p, x, u, d, t=0.0):
    """A = dF/dx, B = dF/du, G = dF/dd of the discrete model at (x,u,d), and F itself."""
    z = np.concatenate([x, u, d])
    J = _fd(lambda v: model_fx(p, v[:p.nx], v[p.nx:p.nx + p.nu], v[p.nx + p.nu:], t), z)
    return J[:, :p.nx], J[:, p.nx:p.nx + p.nu], J[:, p.nx + p.nu:], model_fx(p, x, u, d, t)


def output_jac(p, x, u, d, t=0.0):
    z = np.concatenate([x, d])
    J = _fd(lambda v: model_fy(p, v[:p.nx], u, v[p.nx:], t), z)
    return J[:, :p.nx], J[:, p.nx:]


# ---------------------------------------------------------------------------------------------------
# estimator: Estimator.py:313-386
# ---------------------------------------------------------------------------------------------------
def ekf(p, xi, Pm, y, u, t=0.0):
    n = p.nx
    x, d = xi[:n], xi[n:]
    yhat = model_fy(p, x, u, d, t)
    Cx, Cd = output_jac(p, x, u, d, t)
    C = np.hstack([Cx, Cd])
    K = Pm @ C.T @ np.linalg.inv(C @ Pm @ C.T + p.R_kf)
    Pc = Pm - K @ C @ Pm
    xi_c = xi + K @ (y - yhat)
    A, _, G, _ = linearize(p, xi_c[:n], u, xi_c[n:], t)
    Aa = np.block([[A, G], [np.zeros((p.nd, n)), np.eye(p.nd)]])
    return xi_c, Aa @ Pc @ Aa.T + p.Q_kf


# ---------------------------------------------------------------------------------------------------
# a linear stand-in problem object for mpc_oracle's dense QP builders
# ---------------------------------------------------------------------------------------------------
class _Lin:
    pass


def _linear_view(p, A, B, c, C, e):
    """The linear problem 'x+ = A x + B u + c, y = C x + e' with p's costs and bounds, in the attribute names mpc_oracle expects."""
    q = _Lin()
    q.nx, q.nu, q.ny, q.nd, q.N = p.nx, p.nu, p.ny, 0, p.N
    q.A, q.B, q.C, q.fx_const, q.fy_const = A, B, C, c, e
    q.Bd, q.Cd = np.zeros((p.nx, 0)), np.zeros((p.ny, 0))
    q.Q, q.R, q.P, q.DUForm = p.Q, p.R, np.zeros((p.nx, p.nx)), False
    q.Qss, q.Rss, q.DUssForm = p.Qss, p.Rss, bool(getattr(p, "DUssForm", False))
    for k in ("umin", "umax", "xmin", "xmax", "ymin", "ymax", "umin_ss", "umax_ss", "xmin_ss", "xmax_ss", "ymin_ss", "ymax_ss"):
        setattr(q, k, getattr(p, k))
    q.y_bounded = bool(np.isfinite(p.ymin).any() or np.isfinite(p.ymax).any())
    q.Dumin = q.Dumax = None
    return q


# ---------------------------------------------------------------------------------------------------
# target: opt_ss for a non-linear model (Target_Calc.py:20-161), SQP on the linear target QP
# ---------------------------------------------------------------------------------------------------
def target_solve(p, usp, ysp, d, xs0, us0, tol=1e-10, max_sqp=30, t=0.0, us_prev=None):
    xs, us = np.array(xs0, dtype=np.float64), np.array(us0, dtype=np.float64)
    for it in range(max_sqp):
        A, B, _, F = linearize(p, xs, us, d, t)
        Cx, _ = output_jac(p, xs, us, d, t)
        c = F - A @ xs - B @ us
        e = model_fy(p, xs, us, d, t) - Cx @ xs
        q = _linear_view(p, A, B, c, Cx, e)
        H, g, E, ee, G, lo, hi = o.target_qp(q, usp, ysp, np.zeros(p.nx), np.zeros(0), us0 if us_prev is None else us_prev)
        r = o.qp_ipm_dense(H, g, E, ee, G, lo, hi, tol=1e-12)
        if r["status"] != STATUS_SOLVED:
            return dict(xs=xs, us=us, status=r["status"], sqp_iters=it)
        w = r["w"]
        step = max(np.abs(w[:p.nx] - xs).max(), np.abs(w[p.nx:p.nx + p.nu] - us).max())
        xs, us = w[:p.nx].copy(), w[p.nx:p.nx + p.nu].copy()
        if step < tol:
            return dict(xs=xs, us=us, ys=model_fy(p, xs, us, d, t), status=STATUS_SOLVED, sqp_iters=it + 1)
    return dict(xs=xs, us=us, ys=model_fy(p, xs, us, d, t), status=STATUS_MAXITER, sqp_iters=max_sqp)


# ---------------------------------------------------------------------------------------------------
# OCP: opt_dyn for a non-linear model (Control_Calc.py:20-260), SQP on a dense QP in opt_dyn's own order
# ---------------------------------------------------------------------------------------------------
def ocp_qp_ltv(p, Ak, Bk, ck, C, e, xhat, xs, us, u_prev=None):
    """QP of one SQP iteration: min sum_k 1/2 (x_k-xs)'Q(x_k-xs) + 1/2 (u_k-us)'R(u_k-us), x_0 = xhat,
    x_{k+1} = A_k x_k + B_k u_k + c_k, bounds on x_1..x_N and u, output rows ymin <= C x_k + e <= ymax for k = 1..N-1
    (the k = 0 row constrains the given x_0: feasibility test made by the caller)."""
    n, m, N = p.nx, p.nu, p.N
    nxu = n + m; nw = nxu * N + n
    ix = lambda k: slice(nxu * k, nxu * k + n)
    iu = lambda k: slice(nxu * k + n, nxu * k + nxu)
    H = np.zeros((nw, nw)); g = np.zeros(nw)
    DU = bool(getattr(p, "DUForm", False))
    for k in range(N):
        H[ix(k), ix(k)] += p.Q; g[ix(k)] += -p.Q @ xs
        H[iu(k), iu(k)] += p.R
        if not DU:
            g[iu(k)] += -p.R @ us
        elif k == 0:                       # du = U[0] - um1 (Control_Calc.py:163-166,180-181)
            g[iu(k)] += -p.R @ u_prev
        else:                              # du = U[k] - U[k-1]
            H[iu(k - 1), iu(k - 1)] += p.R; H[iu(k), iu(k - 1)] -= p.R; H[iu(k - 1), iu(k)] -= p.R
    Pf = getattr(p, "Pf", None)
    if Pf is not None:                     # Vfin = 1/2 (x_N - xs)' Pf (x_N - xs)
        H[ix(N), ix(N)] += Pf; g[ix(N)] += -Pf @ xs
    E = np.zeros((n * (N + 1), nw)); ee = np.zeros(n * (N + 1))
    E[0:n, ix(0)] = np.eye(n); ee[0:n] = xhat
    for k in range(N):
        r = slice(n * (k + 1), n * (k + 2))
        E[r, ix(k)] = Ak[k]; E[r, iu(k)] = Bk[k]; E[r, ix(k + 1)] = -np.eye(n); ee[r] = -ck[k]
    rows, lo, hi = [], [], []
    for k in range(1, N + 1):
        for i in range(n):
            if np.isfinite(p.xmin[i]) or np.isfinite(p.xmax[i]):
                row = np.zeros(nw); row[nxu * k + i] = 1.0; rows.append(row); lo.append(p.xmin[i]); hi.append(p.xmax[i])
    for k in range(N):
        for i in range(m):
            if np.isfinite(p.umin[i]) or np.isfinite(p.umax[i]):
                row = np.zeros(nw); row[nxu * k + n + i] = 1.0; rows.append(row); lo.append(p.umin[i]); hi.append(p.umax[i])
    for k in range(1, N):
        for i in range(p.ny):
            if np.isfinite(p.ymin[i]) or np.isfinite(p.ymax[i]):
                row = np.zeros(nw); row[ix(k)] = C[i]; rows.append(row); lo.append(p.ymin[i] - e[i]); hi.append(p.ymax[i] - e[i])
    if getattr(p, "Dumin", None) is not None or getattr(p, "Dumax", None) is not None:      # g2 rows, Control_Calc.py:163-169,241-243
        dlo = p.Dumin if p.Dumin is not None else np.full(m, -np.inf); dhi = p.Dumax if p.Dumax is not None else np.full(m, np.inf)
        for k in range(N):
            for i in range(m):
                if np.isfinite(dlo[i]) or np.isfinite(dhi[i]):
                    row = np.zeros(nw); row[nxu * k + n + i] = 1.0
                    off = u_prev[i] if k == 0 else 0.0
                    if k > 0:
                        row[nxu * (k - 1) + n + i] = -1.0
                    rows.append(row); lo.append(dlo[i] + off); hi.append(dhi[i] + off)
    G = np.array(rows) if rows else np.zeros((0, nw))
    return H, g, E, ee, G, np.array(lo, dtype=float), np.array(hi, dtype=float)


def ocp_solve(p, xhat, xs, us, d, w_guess, max_sqp=50, tol=1e-9, t=0.0, u_prev=None):
    """SQP from the trajectory ``w_guess`` (opt_dyn's order).  ``max_sqp = 1`` is one real-time iteration.
    Returns dict(u0, x1, w, status, sqp_iters, step)."""
    n, m, N = p.nx, p.nu, p.N
    nxu = n + m
    w = np.array(w_guess, dtype=np.float64); w[:n] = xhat
    y0 = model_fy(p, xhat, us, d, t)
    rl = o.BOUND_RELAX * np.maximum(1.0, np.abs(p.ymin)); rh = o.BOUND_RELAX * np.maximum(1.0, np.abs(p.ymax))
    if np.any(y0 < p.ymin - rl) or np.any(y0 > p.ymax + rh):
        return dict(u0=None, x1=None, w=w, status=STATUS_INFEASIBLE, sqp_iters=0, step=np.inf)
    step = np.inf
    for it in range(max_sqp):
        Ak, Bk, ck = [], [], []
        for k in range(N):
            xk, uk = w[nxu * k:nxu * k + n], w[nxu * k + n:nxu * (k + 1)]
            A, B, _, F = linearize(p, xk, uk, d, t)
            Ak.append(A); Bk.append(B); ck.append(F - A @ xk - B @ uk)
        C, _ = output_jac(p, xhat, us, d, t)                      # outputs that are single states: constant selection rows
        e = model_fy(p, xhat, us, d, t) - C @ xhat
        H, g, E, ee, G, lo, hi = ocp_qp_ltv(p, Ak, Bk, ck, C, e, xhat, xs, us, u_prev)
        r = o.qp_ipm_dense(H, g, E, ee, G, lo, hi, tol=1e-11)
        if r["status"] == STATUS_INFEASIBLE:
            return dict(u0=None, x1=None, w=w, status=STATUS_INFEASIBLE, sqp_iters=it, step=step)
        pol = o.qp_polish(H, g, E, ee, G, lo, hi, r["w"], r["z_lo"], r["z_hi"]) if r["status"] == STATUS_SOLVED else None
        wn = pol["w"] if pol is not None else r["w"]
        step = float(np.abs(wn - w).max())
        w = wn
        if step < tol:
            break
    return dict(u0=w[n:n + m].copy(), x1=w[n + m:2 * n + m].copy(), w=w, status=STATUS_SOLVED if step < tol or max_sqp == 1 else STATUS_MAXITER,
                sqp_iters=it + 1, step=step)


def kkt_nlp(p, w, xhat, xs, us, d, t=0.0, u_prev=None):
    """Certificate of an OCP point against the NLP itself: dynamics defects and the least-squares stationarity residual
    |grad f + J' lambda + bound multipliers| with multipliers fitted on the active set (bounds within 1e-7)."""
    n, m, N = p.nx, p.nu, p.N
    nxu = n + m; nw = w.size
    defect = 0.0
    Ak, Bk = [], []
    for k in range(N):
        xk, uk = w[nxu * k:nxu * k + n], w[nxu * k + n:nxu * (k + 1)]
        A, B, _, F = linearize(p, xk, uk, d, t)
        Ak.append(A); Bk.append(B)
        defect = max(defect, np.abs(F - w[nxu * (k + 1):nxu * (k + 1) + n]).max())
    C, _ = output_jac(p, xhat, us, d, t)
    e = model_fy(p, xhat, us, d, t) - C @ xhat
    H, g, E, ee, G, lo, hi = ocp_qp_ltv(p, Ak, Bk, [np.zeros(n)] * N, C, e, xhat, xs, us, u_prev)
    grad = H @ w + g
    Gw = G @ w
    act = (np.abs(Gw - lo) < 1e-7) | (np.abs(Gw - hi) < 1e-7)
    M = np.vstack([E, G[act]]).T
    lam = np.linalg.lstsq(M, -grad, rcond=None)[0]
    return dict(defect=defect, stationarity=float(np.abs(grad + M @ lam).max()), n_active=int(act.sum()),
                bound_violation=float(max(0.0, (lo - Gw).max(initial=0.0), (Gw - hi).max(initial=0.0))))


# ---------------------------------------------------------------------------------------------------
# the closed loop, one instance: MPC_code.py:485-827
# ---------------------------------------------------------------------------------------------------
def closed_loop(p, nsteps, x0_p=None, x0_m=None, max_sqp=50, sqp_tol=1e-9, certify=False, v_wn=None):
    n, m, N = p.nx, p.nu, p.N
    nxu = n + m
    x = np.array(p.x0_p if x0_p is None else x0_p, dtype=np.float64)
    xhat = np.array(p.x0_m if x0_m is None else x0_m, dtype=np.float64)
    u = p.u0.copy(); dhat = p.dhat0.copy(); Pk = p.P0.copy()
    lue = getattr(p, "estimator", "ekf") == "lue"
    xs, us = xhat.copy(), u.copy()
    sched = p.schedules(nsteps)
    w = np.concatenate([np.tile(np.concatenate([xhat, u]), N), xhat])          # :740-756
    L = {k: [] for k in ("U", "X_HAT", "XS", "US", "Xp", "Yp", "D_HAT", "STATUS_DYN", "STATUS_SS", "SQP_DYN", "SQP_SS", "W",
                              "KKT_DEFECT", "KKT_STAT", "KKT_VIOL")}
    for k in range(nsteps):
        t = k * p.h
        L["Xp"].append(x.copy()); L["X_HAT"].append(xhat.copy())
        y = plant_fy(p, x, u, t, sched["pyp"][k])                              # :531-534
        if v_wn is not None:                                                   # :537-541: white noise on the measurement (v_wn [nsteps, ny]: the draws, sqrtm(R_wn) applied)
            y = y + v_wn[k]
        L["Yp"].append(y.copy())
        if lue:                                                               # xi+ = xi + K (y - yhat), Estimator.py:231-261
            xi = np.concatenate([xhat, dhat]) + p.K @ (y - model_fy(p, xhat, u, dhat, t))
        else:
            xi, Pk = ekf(p, np.concatenate([xhat, dhat]), Pk, y, u, t)         # :577-650
        xhat, dhat = xi[:n].copy(), xi[n:].copy()
        if p.dmin is not None:
            dhat = np.minimum(np.maximum(dhat, p.dmin), p.dmax)               # :655-668
        L["D_HAT"].append(dhat.copy())
        us_prev, xs_prev = us.copy(), xs.copy()
        tg = target_solve(p, sched["usp"][k], sched["ysp"][k], dhat, xs, us, t=t, us_prev=us_prev)   # :693-718
        if tg["status"] != STATUS_INFEASIBLE:
            xs, us = tg["xs"], tg["us"]
        L["XS"].append(xs.copy()); L["US"].append(us.copy()); L["STATUS_SS"].append(tg["status"]); L["SQP_SS"].append(tg["sqp_iters"])
        r = ocp_solve(p, xhat, xs, us, dhat, w, max_sqp=max_sqp, tol=sqp_tol, t=t, u_prev=u)    # :733-805
        L["W"].append(r["w"].copy())
        if certify and r["status"] == STATUS_SOLVED:
            c = kkt_nlp(p, r["w"], xhat, xs, us, dhat, t, u_prev=u)
            L["KKT_DEFECT"].append(c["defect"]); L["KKT_STAT"].append(c["stationarity"]); L["KKT_VIOL"].append(c["bound_violation"])
        else:
            L["KKT_DEFECT"].append(np.nan); L["KKT_STAT"].append(np.nan); L["KKT_VIOL"].append(np.nan)
        if r["status"] != STATUS_INFEASIBLE:
            u, xhat = r["u0"], r["x1"]                                         # :798-799
            w = np.concatenate([r["w"][nxu:], us_prev, xs_prev])               # :764
        else:
            xhat = model_fx(p, xhat, u, dhat, t)                              # :804-805
        L["U"].append(u.copy()); L["STATUS_DYN"].append(r["status"]); L["SQP_DYN"].append(r["sqp_iters"])
        x = plant_fx(p, x, u, t, sched["pxp"][k])                              # :813-816
    return {k: np.array(v) for k, v in L.items()}
