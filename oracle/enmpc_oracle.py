"""ORACLE (test infrastructure, never shipped): economic NMPC with a moving-horizon estimator, restated with NumPy.

PARITY UNPINNED against the reference's own solver: CasADi / IPOPT / IDAS cannot run here and the reference ships no vectors
(SURVEY.md section 8c).  What this file pins instead is the *mathematics* of the reference's economic example
(``Ex_ENMPC.py``; BASELINE configs[3] and [4]):

* model / plant      ``defF_model`` / ``defF_p`` with ``StateFeedback`` and ``offree = "lin"``: ``Mx`` classical RK4 steps of the
                     Ex-file's continuous functions per sampling interval, ``+ Bd d`` (``Utilities.py:157-183,174-177``), outputs
                     ``y = x + Cd d`` (``:200-204``), plant output ``y = x_p`` (``:84-86``).
* OCP                ``opt_dyn`` with ``ContForm`` (``Control_Calc.py:102-111,153-158``): multiple shooting where every interval
                     integrates ``xdot = f(x,u,d,t,px) + px`` *together with the cost quadrature* of ``User_fobj_Cont(x,u,y,xs,us,ys)``
                     (``y = Fy_model(x,u,d)``), terminal cost ``User_vfin(x_N, xs)`` (``:194-210``), bounds on ``x_1..x_N`` and ``u``
                     (``:248-252``).  The reference integrates with SUNDIALS IDAS (adaptive BDF, CasADi's default tolerances
                     reltol 1e-6 / abstol 1e-8); here - and in the product - the interval is ``quad_steps`` classical RK4 steps
                     (default 20: within 1e-6 of the exact flow and quadrature on this example, ``test_enmpc_oracle.py``), so parity
                     with the reference's NLP is *to integrator tolerance* by construction.
* target             ``opt_ss`` with ``User_fssobj`` (``Target_Calc.py:20-161``): variables ``[xs, us, ys]``, fixed point of the
                     discrete model, ``ys = Fy_model(xs, us, d)``, cold-started from ``(x0_m, u0)`` every step (``MPC_code.py:696-700``).
* estimator          ``mhe`` (``Estimator.py:388-768``) on ``mhe_opt``'s NLP (``Utilities.py:825-990``): growing window for the
                     first ``N_mhe`` steps (``MPC_code.py:591-598``), arrival cost ``1/2 (X_0 - x_bar)' P^-1 (X_0 - x_bar)``
                     (``Utilities.py:944-945``), prior update by the smoothing recursion (``Estimator.py:652-665``) and
                     ``x_bar`` = the second state of the optimal window (``:750-753``).  The smoothing *correction* of the cost
                     (``Utilities.py:949-952``) is dead code in the reference: the solver object is built for the last time at
                     ``ksim = N_mhe - 1`` (``MPC_code.py:591-598``), where ``ksim >= N_mhe`` is false - so it is not here either.
* loop               ``MPC_code.py:485-827``.

All three NLPs are solved by one dense primal-dual interior point method (``ipm_dense``: the outer algorithm of the reference's
solver, exact Hessian of the Lagrangian, dense LU of the Newton system).  Derivatives are complex-step differences of
the Ex-file's own functions (``exnum.py``; second derivatives: central differences of complex-step gradients) - nothing is
shared with the product's tracer, generated code or Riccati solver.  ``kkt_nlp`` certifies a returned point against the NLP
itself: the reference's IPOPT terminates on exactly these conditions.
"""
from __future__ import annotations

import numpy as np

import exnum

STATUS_SOLVED, STATUS_MAXITER, STATUS_INFEASIBLE = 0, 1, 2
INF = float("inf")


class EconProblem:
    def __repr__(self):
        return f"EconProblem({self.name!r}, nx={self.nx}, nu={self.nu}, ny={self.ny}, nd={self.nd}, N={self.N}, N_mhe={getattr(self, 'N_mhe', None)})"


def _vec(v, n, fill):
    if v is None:
        return np.full(n, fill, dtype=np.float64)
    return np.asarray(v, dtype=np.float64).reshape(n)


def load_problem(path, overrides=None, quad_steps=20):
    """The namespace of an economic example as numbers and plain Python functions (reference MPC_code.py:31-60,84-257,368-438)."""
    ns = exnum.load(path, overrides)
    p = EconProblem()
    p.ns = ns
    p.name = ns["__name__"]
    p.nx, p.nu, p.ny, p.nd, p.nxp = (ns[k].size1() for k in ("x", "u", "y", "d", "xp"))
    p.N, p.h, p.Nsim, p.Mx = int(ns["N"]), float(ns["h"]), int(ns["Nsim"]), int(ns.get("Mx", 10))
    p.quad_steps = int(quad_steps)
    assert ns.get("User_fxm_Cont") and ns.get("User_fobj_Cont") and ns.get("User_fssobj") and ns.get("User_fxp_Cont"), "an economic example with continuous model, plant and cost"
    assert ns["StateFeedback"] is True and ns["offree"] == "lin"
    p.fxm, p.fxp, p.fobj, p.fssobj, p.vfin = ns["User_fxm_Cont"], ns["User_fxp_Cont"], ns["User_fobj_Cont"], ns["User_fssobj"], ns.get("User_vfin")
    p.g_ineq = ns.get("User_g_ineq")      # user inequality rows of the OCP, G(x, u, y, d, t, px, py) <= 0 at every stage (Control_Calc.py:94-100,132-147; MPC_code.py:306-314)
    for k in ("User_h_eq", "User_g_ineq_SS", "User_h_eq_SS"):
        assert ns.get(k) is None, k + " is not restated here"
    p.Bd, p.Cd = np.asarray(ns["Bd"], dtype=float).reshape(p.nx, p.nd), np.asarray(ns["Cd"], dtype=float).reshape(p.ny, p.nd)
    pick = lambda b, s, n, f: _vec(ns.get(b + s) if ns.get(b + s) is not None else ns.get(b), n, f)
    p.umin, p.umax = pick("umin", "_dyn", p.nu, -INF), pick("umax", "_dyn", p.nu, INF)
    p.xmin, p.xmax = pick("xmin", "_dyn", p.nx, -INF), pick("xmax", "_dyn", p.nx, INF)
    p.umin_ss, p.umax_ss = pick("umin", "_ss", p.nu, -INF), pick("umax", "_ss", p.nu, INF)
    p.xmin_ss, p.xmax_ss = pick("xmin", "_ss", p.nx, -INF), pick("xmax", "_ss", p.nx, INF)
    p.ymin_ss, p.ymax_ss = pick("ymin", "_ss", p.ny, -INF), pick("ymax", "_ss", p.ny, INF)
    assert ns.get("ymin") is None and ns.get("ymax") is None, "output rows in the OCP are not restated here"
    p.dmin = None if ns.get("dmin") is None else _vec(ns["dmin"], p.nd, -INF)
    p.dmax = None if ns.get("dmax") is None else _vec(ns["dmax"], p.nd, INF)
    p.x0_p, p.x0_m, p.u0 = _vec(ns["x0_p"], p.nxp, 0.0), _vec(ns["x0_m"], p.nx, 0.0), _vec(ns["u0"], p.nu, 0.0)
    p.max_iter = int(ns.get("Sol_itmax", 100))
    p.mhe = bool(ns.get("mhe", False))
    if p.mhe:
        p.N_mhe, p.mhe_up = int(ns["N_mhe"]), ns.get("mhe_up", "smooth")
        assert p.mhe_up in ("smooth", "filter") and p.N_mhe >= 2, "prior updates: 'smooth' (the shipped example) and 'filter'"
        p.n_w = ns["w"].size1()
        p.fx_mhe, p.fobj_mhe = ns["User_fx_mhe_Cont"], ns["User_fobj_mhe"]
        p.G_mhe = np.asarray(ns["G_mhe"], dtype=float) if ns.get("G_mhe") is not None else np.eye(p.nx + p.nd)      # MPC_code.py:387
        p.P0 = np.asarray(ns["P0"], dtype=float)
        p.x_bar = np.asarray(ns["x_bar"], dtype=float).reshape(p.nx + p.nd)
        ne = p.nx + p.nd
        p.xmin_mhe = np.concatenate([_vec(ns.get("xmin"), p.nx, -INF), _vec(ns.get("dmin"), p.nd, -INF)])      # MPC_code.py:397-402
        p.xmax_mhe = np.concatenate([_vec(ns.get("xmax"), p.nx, INF), _vec(ns.get("dmax"), p.nd, INF)])
        p.wmin, p.wmax = _vec(ns.get("wmin"), p.n_w, -INF), _vec(ns.get("wmax"), p.n_w, INF)      # bounds of the state noise (Utilities.py:881-884,974-977)
        for k in ("vmin", "vmax"):
            assert ns.get(k) is None, "bounds of the output noise are not restated here"
        assert p.G_mhe.shape == (ne, p.n_w)
    p.ekf = bool(ns.get("ekf", False))
    if not p.mhe:      # the example's other estimator (Ex_ENMPC.py:109-123, mhe_mod = 'off'): the extended Kalman filter on [x; d]
        assert p.ekf, "estimator: the moving-horizon estimator (mhe = True) or the extended Kalman filter (ekf = True)"
        ne = p.nx + p.nd
        p.Q_kf, p.R_kf, p.P0 = (np.asarray(ns[k], dtype=float) for k in ("Q_kf", "R_kf", "P0"))
        assert p.Q_kf.shape == (ne, ne) and p.R_kf.shape == (p.ny, p.ny) and p.P0.shape == (ne, ne)
    return p


# ---------------------------------------------------------------------------------------------------
# the example's functions over P points at once (columns): real or complex
# ---------------------------------------------------------------------------------------------------
def _cols(v, P):
    v = np.asarray(v)
    return v if v.ndim == 2 else np.repeat(v.reshape(-1, 1), P, axis=1)


def _rk4(rhs, Z, h, M):
    dt = h / M
    for _ in range(M):
        k1 = rhs(Z); k2 = rhs(Z + 0.5 * dt * k1); k3 = rhs(Z + 0.5 * dt * k2); k4 = rhs(Z + dt * k3)
        Z = Z + dt / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    return Z


def fy_model(p, X, D):
    """Fy_model with StateFeedback: x + Cd d (+ py = 0), Utilities.py:200-204,243."""
    return X + p.Cd @ D


def fx_model(p, X, U, D, t=0.0):
    """Fx_model(x,u,h,d,t,px): Mx RK4 steps of User_fxm_Cont, + Bd d (Utilities.py:160-177); columns are evaluation points."""
    P = np.shape(X)[1]
    zero = np.zeros((p.nx, 1))
    with exnum.batched():
        out = _rk4(lambda Z: np.asarray(p.fxm(Z, U, D, t, zero)) + 0 * Z, np.asarray(X), p.h, p.Mx)
    return out + p.Bd @ _cols(D, P)


def g_rows(p, X, U, D, t=0.0):
    """User_g_ineq(x, u, y, d, t, px, py) with y = Fy_model(x, u, d), px = py = 0 (LinPar): [ng, P], columns are evaluation points"""
    X = np.asarray(X); P = X.shape[1]
    with exnum.batched():
        g = p.g_ineq(X, U, fy_model(p, X, _cols(D, P)), D, t, np.zeros((p.nx, 1)), np.zeros((p.ny, 1)))
    g = np.asarray(g.a if hasattr(g, "a") else g)
    return (g.reshape(-1, P) if g.ndim else g.reshape(1, 1)) + 0 * X[:1]


def fx_plant(p, X, U, t=0.0):
    """Fx_p: Mx RK4 steps of User_fxp_Cont(xp, t, u, pxp, pxmp) (Utilities.py:58-82)."""
    zero = np.zeros((p.nxp, 1))
    with exnum.batched():
        return _rk4(lambda Z: np.asarray(p.fxp(Z, t, U, zero, zero)) + 0 * Z, np.asarray(X), p.h, p.Mx)


def ocp_stage(p, X, U, D, Xs, Us, t=0.0):
    """One shooting interval of the ContForm OCP: x(h) and the integral of the stage cost (Control_Calc.py:102-111,153-158), by
    ``quad_steps`` RK4 steps of [f(x,u,d,t,px) + px; User_fobj_Cont(x, u, y, xs, us, ys)].  Note: no ``Bd d`` in these dynamics
    (SURVEY App. C), unlike Fx_model."""
    X = np.asarray(X); P = X.shape[1]
    zero = np.zeros((p.nx, 1))
    Ys = fy_model(p, _cols(Xs, P), _cols(D, P))

    def rhs(Z):
        x = Z[:p.nx]
        with exnum.batched():
            xd = np.asarray(p.fxm(x, U, D, t, zero)) + 0 * x
            q = np.asarray(p.fobj(x, U, fy_model(p, x, _cols(D, P)), Xs, Us, Ys)) + 0 * x[0]
        return np.vstack([xd, q[None]])
    Z = _rk4(rhs, np.vstack([X, np.zeros((1, P), dtype=X.dtype)]), p.h, p.quad_steps)
    return Z[:p.nx], Z[p.nx]


def fx_mhe(p, Xi, U, W, t=0.0):
    """Fx_mhe(csi,u,k,t,w,px) for offree = 'lin': [RK4 of User_fx_mhe_Cont(x,u,d,t,px,w) + Bd d; d] + G w (Utilities.py:749-821)."""
    Xi = np.asarray(Xi); P = Xi.shape[1]
    zero = np.zeros((p.nx, 1))
    x, d = Xi[:p.nx], Xi[p.nx:]
    with exnum.batched():
        xn = _rk4(lambda Z: np.asarray(p.fx_mhe(Z, U, d, t, zero, W)) + 0 * Z, x, p.h, p.Mx)
    return np.vstack([xn + p.Bd @ d, d]) + p.G_mhe @ _cols(W, P)


def fy_es(p, Xi):
    """Fy_es(csi,u,t,py) = Fy_model(x1,u,d1,t,py) (MPC_code.py:563-564)."""
    return fy_model(p, Xi[:p.nx], Xi[p.nx:])


# ---------------------------------------------------------------------------------------------------
# derivatives: complex step (first), central differences of complex-step gradients (second)
# ---------------------------------------------------------------------------------------------------
CS = 1e-30


def jac_cs(fun, z):
    """fun: [n, P] -> [m, P] (columns = points); returns fun(z) [m] and d fun / d z [m, n] at the single point z."""
    n = len(z)
    Z = np.repeat(np.asarray(z, dtype=complex).reshape(n, 1), n + 1, axis=1)
    Z[np.arange(n), np.arange(n)] += 1j * CS
    out = np.asarray(fun(Z))
    return out[:, n].real, out[:, :n].imag / CS


def hess_fd(fun, z, rel=1e-5):
    """Hessian of the scalar fun ([n, P] -> [P]) at z: central differences of complex-step gradients, symmetrised."""
    n = len(z)
    z = np.asarray(z, dtype=float)
    dz = rel * np.maximum(1.0, np.abs(z))
    cols = []
    for j in range(n):
        for sgn in (1.0, -1.0):
            zz = z.copy(); zz[j] += sgn * dz[j]
            Z = np.repeat(zz.astype(complex).reshape(n, 1), n, axis=1)
            Z[np.arange(n), np.arange(n)] += 1j * CS
            cols.append(Z)
    G = np.asarray(fun(np.hstack(cols))).imag / CS      # [2 n * n]
    G = G.reshape(n, 2, n)
    H = (G[:, 0, :] - G[:, 1, :]) / (2.0 * dz[:, None])
    return 0.5 * (H + H.T)


# ---------------------------------------------------------------------------------------------------
# dense primal-dual interior point method on  min f(w)  s.t.  g(w) = 0,  lo <= w <= hi
# ---------------------------------------------------------------------------------------------------
# The algorithm is the reference solver's: IPOPT 3.12 at the options MPC_code.py:262-263 leaves at their defaults [ext] (Waechter & Biegler, Math. Program.
# 106 (2006), referred to as [WB] below, and the implementation's file names).  The reference's solver is not installable here, so this is a RESTATEMENT
# FROM ITS PUBLISHED DESCRIPTION - parity with a reference run stays unpinned (header) - of:
#   * problem scaling (IpGradientScaling): the objective is multiplied by df = min(1, 100 / |grad f(w0)|_inf) at the caller's starting point, before the push;
#   * first iterate (IpDefaultIterateInitializer): w0 pushed into the box by bound_push = bound_frac = 0.01 [WB 3.6], bound multipliers 1, equality
#     multipliers by least squares  min |grad f - z_L + z_U + J'y|  [WB (36)], dropped (zero) when |y|_inf > 1000 or the problem is square;
#   * monotone barrier parameter [WB (7), (8)]: mu_init 0.1, kappa_eps 10, kappa_mu 0.2, theta_mu 1.5, floor tol / (kappa_eps + 1), several decreases in a
#     row when the error allows; tau = max(0.99, 1 - mu); the filter is emptied whenever mu changes;
#   * search direction [WB (11), (13)] from the primal-dual system with Sigma = z / s, the Hessian of the Lagrangian shifted by delta I while the
#     reduced Hessian lacks positive curvature [WB 3.1]: delta = 1e-4 first, x100 / x8, restarted from a third of the last one; damping kappa_d = 1e-5 of
#     variables with one bound [WB 3.7];
#   * fraction to the boundary [WB (15)] for primal and dual step separately; equality multipliers move with the primal step length;
#   * FILTER LINE SEARCH [WB 2.3; IpFilterLSAcceptor, IpBacktrackingLineSearch]: theta = |c|_1, phi = barrier function; switching condition with
#     s_phi 2.3, s_theta 1.1, delta 1; Armijo eta_phi 1e-8; sufficient decrease gamma_theta 1e-5, gamma_phi 1e-8; theta_max / theta_min = 1e4 / 1e-4
#     max(1, theta(w0)); backtracking by halves down to alpha_min [WB (23)] with alpha_min_frac 0.05; comparisons relaxed by 10 eps |reference|
#     (Compare_le); second-order correction [WB 2.4] when the first trial step is rejected and does not reduce theta: up to 4, kappa_soc 0.99; tiny steps
#     (|dw| / (1 + |w|) < 10 eps with theta <= 1e-4) accepted unchecked, two in a row force mu down or end the solve;
#   * slacks: IpIpoptCalculatedQuantities::CalculateSafeSlack - a slack below eps min(1, mu) becomes min(max(mu / z, eps min(1, mu)), max(s, 0) +
#     eps^(3/4) max(1, |bound|)) and the bound of THIS solve moves along; bound multipliers kept within kappa_Sigma = 1e10 of mu / s [WB (16)];
#   * bounds: every finite bound relaxed by bound_relax_factor max(1, |bound|) = 1e-8 ... before the first iterate is pushed (OrigIpoptNLP::relax_bounds), the final
#     point projected back into the caller's bounds (honor_original_bounds = yes: the 3.12 series' default, the series CasADi 3.4 / 3.5 bundle [ext]);
#   * RESTORATION PHASE [WB 3.3; IpRestoMinC_1Nrm, IpRestoIpoptNLP, IpRestoFilterConvCheck] for every NLP (``resto``; since round 5 also the OCP and the estimator): _restore;
#   * stop [WB (5), (6); IpOptErrorConvCheck]: scaled error E_0 <= tol with the unscaled side conditions dual_inf_tol 1, constr_viol_tol 1e-4,
#     compl_inf_tol 1e-4; "acceptable" stop after 15 iterations in a row within 1e-6 (1e10, 1e-2, 1e-2); iteration limit.
# NOT RESTATED (the one list of it; DESIGN.md section 14 repeats it): the watchdog procedure (watchdog_shortened_iter_trigger 10: a relaxed acceptance after ten
# shortened steps in a row); scaling of the CONSTRAINTS (no row of the Jacobian exceeds 100 on the models here: asserted by the tests through ``scale_rows``);
# the filter's reset heuristic (max_filter_resets 5); delta_c (the regularisation of a rank-deficient Jacobian: the dynamics' Jacobian has full row rank by
# construction); mu_target, the quality function and every non-default branch; the restoration phase's own restoration ('Restoration_Failed' there).
# The product runs the SAME algorithm on a Riccati factorisation of the same Newton system (csrc/mpc_enmpc.hpp).
BOUND_RELAX_FACTOR = 1e-8
KAPPA_PUSH, MU_INIT, KAPPA_EPS, KAPPA_MU, THETA_MU, TAU_MIN, KAPPA_SIGMA, S_MAX = 1e-2, 0.1, 10.0, 0.2, 1.5, 0.99, 1e10, 100.0
DELTA_FIRST, DELTA_MAX = 1e-4, 1e40
EPS = float(np.finfo(float).eps)
SLACK_MOVE = EPS ** 0.75
SCALE_MAX_GRAD, SCALE_MIN = 100.0, 1e-8
Y_INIT_MAX = 1e3
KAPPA_D = 1e-5
GAMMA_THETA, GAMMA_PHI, LS_DELTA, S_THETA, S_PHI, ETA_PHI = 1e-5, 1e-8, 1.0, 1.1, 2.3, 1e-8
THETA_MAX_FACT, THETA_MIN_FACT, ALPHA_MIN_FRAC, ALPHA_RED, OBJ_MAX_INC = 1e4, 1e-4, 0.05, 0.5, 5.0
MAX_SOC, KAPPA_SOC = 4, 0.99
TINY_STEP_TOL, TINY_STEP_Y_TOL = 10.0 * EPS, 1e-2
DUAL_INF_TOL, CONSTR_VIOL_TOL, COMPL_INF_TOL = 1.0, 1e-4, 1e-4
ACC_TOL, ACC_ITER, ACC_DUAL_INF_TOL, ACC_CONSTR_VIOL_TOL, ACC_COMPL_INF_TOL = 1e-6, 15, 1e10, 1e-2, 1e-2
FILTER_CAP = 16     # entries the product keeps (registers); a ninth is merged into the last, conservatively - never seen to happen


def _null(E):
    u, s, vt = np.linalg.svd(E)
    r = int((s > 1e-12 * max(s.max(initial=0.0), 1.0)).sum())
    return vt[r:].T


def push_interior(w, lo, hi):
    """IPOPT's projection of the first guess: at least min(kappa max(1, |bound|), kappa (hi - lo)) away from every finite bound."""
    w = np.array(w, dtype=float)
    fl, fh = np.isfinite(lo), np.isfinite(hi)
    with np.errstate(invalid="ignore"):
        gap = np.where(fl & fh, KAPPA_PUSH * (hi - lo), INF)
        pl = np.minimum(KAPPA_PUSH * np.maximum(1.0, np.abs(lo)), gap)
        ph = np.minimum(KAPPA_PUSH * np.maximum(1.0, np.abs(hi)), gap)
        w = np.where(fl, np.maximum(w, lo + pl), w)
        w = np.where(fh, np.minimum(w, hi - ph), w)
    return w


def _safe_slacks(w, lo, hi, zl, zh, mu, fl, fh):
    """Slacks of the bounds at w with IPOPT's CalculateSafeSlack; returns (sl, sh, lo', hi'): a corrected slack moves its bound (the caller keeps the
    moved bounds when it keeps the point)."""
    lo, hi = lo.copy(), hi.copy()
    sl, sh = np.where(fl, w - lo, 1.0), np.where(fh, hi - w, 1.0)
    s_min = EPS * min(1.0, mu)
    with np.errstate(divide="ignore", invalid="ignore"):
        for s, z, b, f, sign in ((sl, zl, lo, fl, 1.0), (sh, zh, hi, fh, -1.0)):
            fix = f & (s < s_min)
            if fix.any():
                s[fix] = np.minimum(np.maximum(mu / z[fix], s_min), np.maximum(s[fix], 0.0) + SLACK_MOVE * np.maximum(1.0, np.abs(b[fix])))
                b[fix] = w[fix] - sign * s[fix]
    return sl, sh, lo, hi


def _le(lhs, rhs, bas):
    """IPOPT's Compare_le: lhs <= rhs up to 10 eps |bas|"""
    return lhs - rhs <= 10.0 * EPS * abs(bas)


def ipm_dense(evalf, w0, lo, hi, tol=1e-8, max_iter=200, trace=None, info=None, resto=True):
    """evalf(w, lam) -> f, grad f [n], g [m], dg/dw [m, n], Hessian of f + lam'g [n, n] (``lam`` of length 0: no Hessian wanted).  Variables with
    lo == hi are PARAMETERS, as IPOPT treats them (fixed_variable_treatment = make_parameter, its default [ext]): the initial state of the OCP
    (MPC_code.py:734) drops out of the variables, and the rows that only restate it (Control_Calc.py:126) drop out of the constraints."""
    fixed = lo == hi
    if fixed.any():
        free = ~fixed
        wf = np.where(fixed, lo, w0)
        rows = None

        def sub(wv, lam):
            nonlocal rows
            wfull = wf.copy(); wfull[free] = wv
            if rows is None:
                J0 = evalf(wfull, np.zeros(0))[3]
                rows = np.abs(J0[:, free]).sum(axis=1) > 0.0
            lfull = np.zeros(len(rows)); 
            if len(lam):
                lfull[rows] = lam
            f, gf, g, J, H = evalf(wfull, lfull if len(lam) else lam)
            return f, gf[free], g[rows], J[np.ix_(rows, free)], H[np.ix_(free, free)]
        r = ipm_dense(sub, np.asarray(w0, dtype=float)[free], lo[free], hi[free], tol=tol, max_iter=max_iter, trace=trace, info=info, resto=resto)
        wfull = wf.copy(); wfull[free] = r["w"]
        lfull = np.zeros(len(rows)); lfull[rows] = r["lam"]
        # multipliers of the dropped rows / bounds of the fixed variables, for the certificate of the full statement: the rows that
        # restate a fixed variable take whatever makes its stationarity row vanish
        f, gf, g, J, H = evalf(wfull, lfull)
        zl = np.zeros(len(wfull)); zh = np.zeros(len(wfull)); zl[free] = r["z_lo"]; zh[free] = r["z_hi"]
        st = gf + J.T @ lfull
        Jd = J[np.ix_(~rows, fixed)]
        if Jd.size:
            lfull[~rows] = np.linalg.lstsq(Jd.T, -st[fixed], rcond=None)[0]
        r.update(w=wfull, lam=lfull, z_lo=zl, z_hi=zh)
        return r
    n = len(w0)
    lo_user, hi_user = np.array(lo, dtype=float), np.array(hi, dtype=float)
    # IPOPT relaxes every finite bound by bound_relax_factor max(1, |bound|) before it starts (OrigIpoptNLP::relax_bounds [ext]; default 1e-8, left there by
    # MPC_code.py:262-263) and projects the final point back into the caller's bounds (honor_original_bounds = yes, the default of the 3.12 series [ext])
    with np.errstate(invalid="ignore"):
        lo = np.where(np.isfinite(lo_user), lo_user - BOUND_RELAX_FACTOR * np.maximum(1.0, np.abs(lo_user)), lo_user)      # this solve's own bounds: the safe slack moves them
        hi = np.where(np.isfinite(hi_user), hi_user + BOUND_RELAX_FACTOR * np.maximum(1.0, np.abs(hi_user)), hi_user)
    fl, fh = np.isfinite(lo), np.isfinite(hi)
    # ---- scaling of the objective at the caller's point (IpGradientScaling) ------------------------------------------------------------
    w0 = np.asarray(w0, dtype=float)
    _, gf0, c0, J0, _ = evalf(w0, np.zeros(0))
    gmax = float(np.abs(gf0).max(initial=0.0))
    df = max(SCALE_MAX_GRAD / gmax, SCALE_MIN) if gmax > SCALE_MAX_GRAD else 1.0
    stats = dict(ls_steps=0, soc=0, tiny=0, filter_max=0, resto=0, resto_iters=0, stop="iteration limit", df=df, scale_rows=bool(np.abs(J0).max(initial=0.0) > SCALE_MAX_GRAD))
    m = len(c0)

    def ev(w_, lam_, mu_=None):
        """the scaled problem: df f, with the Hessian of df f + lam'g"""
        if len(lam_) and m:
            f_, gf_, c_, J_, H_ = evalf(w_, lam_ / df)
            return df * f_, df * gf_, c_, J_, df * H_
        f_, gf_, c_, J_, H_ = evalf(w_, lam_)
        return df * f_, df * gf_, c_, J_, (df * H_ if np.size(H_) else H_)
    # ---- first iterate -------------------------------------------------------------------------------------------------------------------
    w = push_interior(w0, lo, hi)
    zl, zh = np.where(fl, 1.0, 0.0), np.where(fh, 1.0, 0.0)
    lam = np.zeros(m)
    if 0 < m < n:
        _, gf, c, J, _ = ev(w, np.zeros(0))
        if np.all(np.isfinite(gf)) and np.all(np.isfinite(J)):
            try:
                y = np.linalg.solve(np.block([[np.eye(n), J.T], [J, np.zeros((m, m))]]), np.concatenate([-(gf - zl + zh), np.zeros(m)]))[n:]
                if np.all(np.isfinite(y)) and np.abs(y).max() <= Y_INIT_MAX:
                    lam = y
            except np.linalg.LinAlgError:
                pass
    r = _ipm_core(ev, w, lo, hi, zl, zh, lam, MU_INIT, tol, max_iter, 0, THETA_MAX_FACT, None, resto, stats, trace, df)
    if info is not None:
        info.update(stats)
    r["w"] = np.minimum(np.maximum(r["w"], lo_user), hi_user)      # honor_original_bounds
    return dict(w=r["w"], lam=r["lam"] / df, z_lo=r["zl"] / df, z_hi=r["zh"] / df, status=r["status"], iters=r["it"], mu=r["mu"], df=df, stop=stats["stop"])      # (multipliers of the unscaled problem)


RESTO_RHO, RESTO_KAPPA, RESTO_THETA_MAX_FACT, RESTO_BOUND_MULT_RESET, RESTO_FEAS_FACT = 1000.0, 0.9, 1e8, 1e3, 1e2


def _ipm_core(ev, w, lo, hi, zl, zh, lam, mu, tol, max_iter, it0, theta_max_fact, hook, resto, stats, trace, df):
    """The iteration of ipm_dense from a given first iterate (w, zl, zh, lam, mu); ev(w, lam, mu) evaluates the problem (its objective may depend on mu: the
    restoration problem's does).  hook(w) -> True ends the solve (the restoration phase's own test); resto: whether a failed line search may enter the
    restoration phase.  lo / hi are modified in place (moved bounds)."""
    n, m = len(w), len(lam)
    fl, fh = np.isfinite(lo), np.isfinite(hi)
    one_l, one_h = fl & ~fh, fh & ~fl                                   # variables with one bound: damped [WB 3.7]
    nb = int(fl.sum() + fh.sum())

    def barrier(f_, sl_, sh_, mu_):
        return f_ - mu_ * (np.log(sl_[fl]).sum() + np.log(sh_[fh]).sum()) + KAPPA_D * mu_ * (sl_[one_l].sum() + sh_[one_h].sum())
    delta_last = 0.0
    tau = max(TAU_MIN, 1.0 - mu)
    filt = []                                        # entries (phi, theta)
    theta_max = theta_min = -1.0
    acc_count, tiny_last, tiny_flag = 0, False, False
    status, it = STATUS_MAXITER, it0
    it = it0
    while True:
        f, gf, c, J, H = ev(w, lam, mu)
        if not (np.all(np.isfinite(w)) and np.isfinite(f) and np.all(np.isfinite(gf)) and np.all(np.isfinite(c))):
            status = STATUS_INFEASIBLE; stats["stop"] = "not finite"
            break
        sl, sh, lo_, hi_ = _safe_slacks(w, lo, hi, zl, zh, mu, fl, fh)
        lo[:], hi[:] = lo_, hi_
        stat = gf + J.T @ lam - zl + zh
        if not np.all(np.isfinite(stat)):
            status = STATUS_INFEASIBLE; stats["stop"] = "not finite"
            break
        if hook is not None and hook(w):
            status = STATUS_SOLVED; stats["stop"] = "restored"
            break
        s_d = max(S_MAX, (np.abs(lam).sum() + zl.sum() + zh.sum()) / max(m + nb, 1)) / S_MAX
        s_c = max(S_MAX, (zl.sum() + zh.sum()) / max(nb, 1)) / S_MAX
        e_st, e_c = float(np.abs(stat).max(initial=0.0)), float(np.abs(c).max(initial=0.0))

        def compl(mu_):
            return max(np.abs(np.where(fl, sl * zl - mu_, 0.0)).max(initial=0.0), np.abs(np.where(fh, sh * zh - mu_, 0.0)).max(initial=0.0))

        def err(mu_):
            return max(e_st / s_d, e_c, compl(mu_) / s_c)
        if trace is not None:
            trace.append(dict(it=it, f=f / df, E0=err(0.0), mu=mu, w=w.copy(), e_st=e_st, e_c=e_c, compl=compl(0.0), theta=float(np.abs(c).sum()), resto=hook is not None))
        c0_ = compl(0.0)
        if err(0.0) <= tol and e_st <= DUAL_INF_TOL and e_c <= CONSTR_VIOL_TOL and c0_ <= COMPL_INF_TOL:
            status = STATUS_SOLVED; stats["stop"] = "converged"
            break
        if err(0.0) <= ACC_TOL and e_st <= ACC_DUAL_INF_TOL and e_c <= ACC_CONSTR_VIOL_TOL and c0_ <= ACC_COMPL_INF_TOL:
            acc_count += 1
            if acc_count >= ACC_ITER:
                status = STATUS_SOLVED; stats["stop"] = "acceptable"
                break
        else:
            acc_count = 0
        if it >= max_iter:
            stats["stop"] = "iteration limit"
            break
        # ---- barrier parameter (IpMonotoneMuUpdate) -----------------------------------------------------------------------------------------
        mu_min = min(tol, COMPL_INF_TOL) / (KAPPA_EPS + 1.0)
        mu_changed = False
        stop_tiny = False
        while err(mu) <= KAPPA_EPS * mu or tiny_flag:
            new_mu = max(min(KAPPA_MU * mu, mu ** THETA_MU), mu_min)
            if new_mu == mu:
                stop_tiny = tiny_flag
                break
            mu, mu_changed, tiny_flag = new_mu, True, False
        if stop_tiny:
            status = STATUS_MAXITER; stats["stop"] = "tiny step"      # 'Search_Direction_Becomes_Too_Small': the reference accepts the point
            break
        tiny_flag = False
        if mu_changed:
            filt = []
            tau = max(TAU_MIN, 1.0 - mu)
            if hook is not None:                     # (the restoration problem's objective changes with mu)
                f, gf, c, J, H = ev(w, lam, mu)
        # ---- search direction -------------------------------------------------------------------------------------------------------------------
        Sig = np.where(fl, zl / sl, 0.0) + np.where(fh, zh / sh, 0.0)
        gphi = gf - np.where(fl, mu / sl, 0.0) + np.where(fh, mu / sh, 0.0) + KAPPA_D * mu * (one_l.astype(float) - one_h.astype(float))      # gradient of the barrier function
        Z = _null(J)
        delta = 0.0
        while True:
            Hk = H + np.diag(Sig) + delta * np.eye(n)
            if Z.shape[1] == 0 or np.linalg.eigvalsh(Z.T @ Hk @ Z).min() > 0.0:
                break
            delta = max(DELTA_FIRST, delta_last / 3.0) if delta == 0.0 else delta * (100.0 if delta_last == 0.0 else 8.0)
            if delta > DELTA_MAX:
                break
        if delta > DELTA_MAX:
            status = STATUS_INFEASIBLE; stats["stop"] = "no curvature"
            break
        if delta > 0.0:
            delta_last = delta
        KKT = np.block([[Hk, J.T], [J, np.zeros((m, m))]])

        def direction(c_rhs):
            sol = np.linalg.solve(KKT, np.concatenate([-gphi, -c_rhs]))
            return sol[:n], sol[n:]                  # dw, the new equality multipliers

        def maxstep(v, dv, mask):
            neg = mask & (dv < 0)
            return min(1.0, float(np.min(-tau * v[neg] / dv[neg]))) if neg.any() else 1.0
        dw, lam_new = direction(c)
        a_max = min(maxstep(sl, dw, fl), maxstep(sh, -dw, fh))
        # ---- filter line search ------------------------------------------------------------------------------------------------------------------
        theta, phi = float(np.abs(c).sum()), barrier(f, sl, sh, mu)
        gbd = float(gphi @ dw)
        a_min = GAMMA_THETA
        if gbd < 0.0:
            a_min = min(GAMMA_THETA, GAMMA_PHI * theta / (-gbd))
            if theta <= theta_min:
                a_min = min(a_min, LS_DELTA * theta ** S_THETA / (-gbd) ** S_PHI)
        a_min *= ALPHA_MIN_FRAC
        if theta_max < 0.0:
            theta_max, theta_min = theta_max_fact * max(1.0, theta), THETA_MIN_FACT * max(1.0, theta)

        def trial(alpha, d_):
            wt = w + alpha * d_
            ft, _, ct, _, _ = ev(wt, np.zeros(0), mu)
            slt, sht, lot, hit = _safe_slacks(wt, lo, hi, zl, zh, mu, fl, fh)
            ok = bool(np.isfinite(ft) and np.all(np.isfinite(ct)))
            return wt, (float(np.abs(ct).sum()) if ok else INF), (barrier(ft, slt, sht, mu) if ok else INF), ct, (slt, sht, lot, hit), ok

        def ftype(alpha):
            if theta == 0.0 and 0.0 < gbd < 100.0 * EPS:
                return True
            return gbd < 0.0 and alpha * (-gbd) ** S_PHI > LS_DELTA * theta ** S_THETA

        def armijo(alpha, phi_t):
            return _le(phi_t - phi, ETA_PHI * alpha * gbd, phi)

        def to_iterate(theta_t, phi_t, from_resto=False):
            if not from_resto and phi_t > phi:
                bas = np.log10(abs(phi)) if abs(phi) > 10.0 else 1.0
                if np.log10(phi_t - phi) > OBJ_MAX_INC + bas:
                    return False
            return _le(theta_t, (1.0 - GAMMA_THETA) * theta, theta) or _le(phi_t - phi, -GAMMA_PHI * theta, phi)

        def to_filter(theta_t, phi_t):               # acceptable to an entry: better than it in one of the two measures
            return not any(_le(ph_j, phi_t, ph_j) and _le(th_j, theta_t, th_j) for (ph_j, th_j) in filt)

        def acceptable(alpha, theta_t, phi_t):
            if theta_t > theta_max:
                return False
            ok = armijo(alpha, phi_t) if (alpha > 0.0 and ftype(alpha) and theta <= theta_min) else to_iterate(theta_t, phi_t)
            return ok and to_filter(theta_t, phi_t)

        def augment():
            nonlocal filt
            ent = (phi - GAMMA_PHI * theta, (1.0 - GAMMA_THETA) * theta)
            filt = [e for e in filt if not (e[0] >= ent[0] and e[1] >= ent[1])]      # entries the new one dominates are dropped
            if len(filt) >= FILTER_CAP:
                filt[-1] = (min(filt[-1][0], ent[0]), min(filt[-1][1], ent[1]))
            else:
                filt.append(ent)
            stats["filter_max"] = max(stats["filter_max"], len(filt))
        tiny = bool((np.abs(dw) / (1.0 + np.abs(w))).max(initial=0.0) < TINY_STEP_TOL and theta <= 1e-4)
        accepted, alpha, d_acc, lam_acc = None, a_max, dw, lam_new
        if tiny:
            T = trial(a_max, dw)
            if T[5]:
                accepted = T; stats["tiny"] += 1
                tiny_flag = tiny_last
                tiny_last = bool(np.abs(lam_new - lam).max(initial=0.0) < TINY_STEP_Y_TOL)
            else:
                tiny = False
        if not tiny:
            tiny_last = False
            n_steps = 0
            while alpha > a_min or n_steps == 0:
                T = trial(alpha, dw)
                if T[5] and acceptable(alpha, T[1], T[2]):
                    accepted = T
                    break
                if T[5] and n_steps == 0 and theta <= T[1]:      # second-order correction: the first trial step did not reduce the infeasibility
                    c_soc, a_soc, theta_old, theta_t, cnt = c.copy(), alpha, 0.0, T[1], 0
                    ct = T[3]
                    while cnt < MAX_SOC and accepted is None and (cnt == 0 or theta_t <= KAPPA_SOC * theta_old):
                        theta_old = theta_t
                        c_soc = a_soc * c_soc + ct
                        d_s, lam_s = direction(c_soc)
                        a_soc = min(maxstep(sl, d_s, fl), maxstep(sh, -d_s, fh))
                        Ts = trial(a_soc, d_s)
                        stats["soc"] += 1
                        if Ts[5] and acceptable(alpha, Ts[1], Ts[2]):      # (the tests keep the original step length)
                            accepted, d_acc, lam_acc, alpha_soc = Ts, d_s, lam_s, a_soc
                        else:
                            cnt += 1; theta_t, ct = Ts[1], Ts[3]
                            if not Ts[5]:
                                break
                    if accepted is not None:
                        break
                alpha *= ALPHA_RED
                n_steps += 1
            stats["ls_steps"] += n_steps
            if accepted is None:
                # ---- IPOPT's restoration phase (IpRestoMinC_1Nrm, IpRestoIpoptNLP, IpRestoFilterConvCheck) -----------------------------------------
                if theta <= 1e-2 * tol:
                    status = STATUS_MAXITER; stats["stop"] = "line search failed at a feasible point"      # 'Restoration_Failed': the reference accepts the point
                    break
                if not resto or hook is not None:
                    # not restated for this problem (the OCP and the estimator in the product: their Newton systems would need another recursion) / no
                    # restoration inside the restoration phase ('Restoration_Failed' there: the caller decides)
                    status = STATUS_INFEASIBLE; stats["stop"] = "restoration needed"
                    break
                augment()                            # the point the restoration starts from is never returned to
                stats["resto"] += 1

                def orig_ok(x_):                     # the restoration phase's own test: enough less infeasible, acceptable to the filter and to the point left
                    ft, _, ct, _, _ = ev(x_, np.zeros(0), mu)
                    if not (np.isfinite(ft) and np.all(np.isfinite(ct))):
                        return False
                    th_t = float(np.abs(ct).sum())
                    if th_t > RESTO_KAPPA * theta:
                        return False
                    slt, sht, _, _ = _safe_slacks(x_, lo, hi, zl, zh, mu, fl, fh)
                    ph_t = barrier(ft, slt, sht, mu)
                    return to_filter(th_t, ph_t) and to_iterate(th_t, ph_t, from_resto=True)
                rr = _restore(ev, w, lo, hi, zl, zh, mu, c, tol, max_iter, it + 1, orig_ok, stats, df, trace)
                it = rr["it"] - 1
                if rr["status"] != "restored":
                    if rr["status"] == "limit":
                        status = STATUS_MAXITER; stats["stop"] = "iteration limit"
                    elif rr["status"] == "converged":      # the restoration problem has a minimiser here: infeasible, or feasible and not acceptable
                        cx = ev(rr["x"], np.zeros(0), mu)[2]
                        if np.abs(cx).max(initial=0.0) <= RESTO_FEAS_FACT * tol:
                            status = STATUS_MAXITER; stats["stop"] = "restoration converged to a feasible point"
                        else:
                            status = STATUS_INFEASIBLE; stats["stop"] = "locally infeasible"
                    else:
                        status = STATUS_MAXITER; stats["stop"] = "restoration failed"      # 'Restoration_Failed': the reference accepts the point
                    break
                # back from the restoration: bound multipliers as if the whole move had been one Newton step, reset to 1 when they grew beyond 1e3; equality
                # multipliers zero (constr_mult_reset_threshold 0)
                x_new = rr["x"]
                slt, sht, lot, hit = _safe_slacks(x_new, lo, hi, zl, zh, mu, fl, fh)
                dzl = np.where(fl, mu / sl - zl - zl / sl * (slt - sl), 0.0)
                dzh = np.where(fh, mu / sh - zh - zh / sh * (sht - sh), 0.0)
                a_du = min(maxstep(zl, dzl, fl), maxstep(zh, dzh, fh))
                zl, zh = zl + a_du * dzl, zh + a_du * dzh
                if max(zl.max(initial=0.0), zh.max(initial=0.0)) > RESTO_BOUND_MULT_RESET:
                    zl, zh = np.where(fl, 1.0, 0.0), np.where(fh, 1.0, 0.0)
                w, lam = x_new, np.zeros(m)
                lo[:], hi[:] = lot, hit
                zl = np.where(fl, np.clip(zl, mu / (KAPPA_SIGMA * slt), KAPPA_SIGMA * mu / slt), 0.0)
                zh = np.where(fh, np.clip(zh, mu / (KAPPA_SIGMA * sht), KAPPA_SIGMA * mu / sht), 0.0)
                it += 1
                continue
            # the filter grows unless the step was an Armijo step on the barrier function (IpFilterLSAcceptor::UpdateForNextIteration)
            if not ftype(alpha) or not armijo(alpha, accepted[2]):
                augment()
        a_pr = alpha_soc if d_acc is not dw else alpha
        # ---- the accepted point; bounds move with corrected slacks; multipliers within kappa_Sigma of mu / s ---------------------------
        dzl = np.where(fl, mu / sl - zl - zl / sl * d_acc, 0.0)      # (of the direction that was taken: the corrected one after a second-order correction)
        dzh = np.where(fh, mu / sh - zh + zh / sh * d_acc, 0.0)
        a_du = min(maxstep(zl, dzl, fl), maxstep(zh, dzh, fh))
        w = accepted[0]
        sl, sh, lo_, hi_ = accepted[4]
        lo[:], hi[:] = lo_, hi_
        lam = lam + a_pr * (lam_acc - lam)
        zl, zh = zl + a_du * dzl, zh + a_du * dzh
        zl = np.where(fl, np.clip(zl, mu / (KAPPA_SIGMA * sl), KAPPA_SIGMA * mu / sl), 0.0)
        zh = np.where(fh, np.clip(zh, mu / (KAPPA_SIGMA * sh), KAPPA_SIGMA * mu / sh), 0.0)
        it += 1
    return dict(w=w, lam=lam, zl=zl, zh=zh, mu=mu, status=status, it=it)


def _restore(ev, x_r, lo, hi, zl, zh, mu, c_r, tol, max_iter, it0, orig_ok, stats, df, trace=None):
    """IPOPT's restoration phase: the same interior point iteration on
        min  rho sum(n + p) + eta(mu) / 2 |D_R (x - x_R)|^2   s.t.  c(x) + n - p = 0,  lo <= x <= hi,  n, p >= 0
    (rho = 1000, eta = sqrt(mu), D_R = diag(1 / max(1, |x_R|)); [WB 3.3]) from x_R with n, p from [WB (32), (33)] at mu_R = max(mu, |c(x_R)|_inf), until
    ``orig_ok(x)``.  Returns status 'restored' | 'converged' | 'limit' | 'failed', x, the iteration counter."""
    n, m = len(x_r), len(c_r)
    d_r = 1.0 / np.maximum(1.0, np.abs(x_r))
    mu_r = max(mu, float(np.abs(c_r).max(initial=0.0)))
    a = mu_r / (2.0 * RESTO_RHO) - 0.5 * c_r
    nn = a + np.sqrt(a * a + mu_r * c_r / (2.0 * RESTO_RHO))
    pp = c_r + nn
    wb = np.concatenate([x_r, nn, pp])
    lob, hib = np.concatenate([lo, np.zeros(2 * m)]), np.concatenate([hi, np.full(2 * m, INF)])
    zlb = np.concatenate([np.minimum(RESTO_RHO, zl), mu_r / nn, mu_r / pp]); zhb = np.concatenate([np.minimum(RESTO_RHO, zh), np.zeros(2 * m)])
    zlb[:n] = np.where(np.isfinite(lo), zlb[:n], 0.0)
    Im = np.eye(m)

    def ev_r(wb_, lam_, mu_):
        x_, n_, p_ = wb_[:n], wb_[n:n + m], wb_[n + m:]
        eta = np.sqrt(mu_)
        _, _, c_, J_, _ = ev(x_, np.zeros(0), mu_)
        e_ = d_r * (x_ - x_r)
        f_ = RESTO_RHO * (n_.sum() + p_.sum()) + 0.5 * eta * float(e_ @ e_)
        gf_ = np.concatenate([eta * d_r * e_, np.full(2 * m, RESTO_RHO)])
        Hb = np.zeros((0, 0))
        if len(lam_):
            Hc = ev(x_, lam_, mu_)[4] - ev(x_, np.zeros(m), mu_)[4]      # Hessian of lam'c alone
            Hb = np.zeros((n + 2 * m, n + 2 * m)); Hb[:n, :n] = Hc + eta * np.diag(d_r * d_r)
        return f_, gf_, c_ + n_ - p_, np.hstack([J_, Im, -Im]), Hb
    sub = dict(ls_steps=0, soc=0, tiny=0, filter_max=0, resto=0, resto_iters=0, stop="")
    r = _ipm_core(ev_r, wb, lob, hib, zlb, zhb, np.zeros(m), mu_r, tol, max_iter, it0, RESTO_THETA_MAX_FACT, lambda wb_: orig_ok(wb_[:n]), False, sub, trace, df)
    stats["resto_iters"] += r["it"] - it0
    lo[:], hi[:] = lob[:n], hib[:n]                  # (bounds the restoration moved stay moved)
    st = {"restored": "restored", "converged": "converged", "acceptable": "converged", "tiny step": "converged", "iteration limit": "limit"}.get(sub["stop"], "failed")
    return dict(status=st, x=r["w"][:n], it=r["it"])


def kkt_nlp(evalf, sol, lo, hi):
    """Residuals of the NLP's own first-order conditions at sol: stationarity, constraint violation, complementarity, dual sign."""
    w, lam, z_lo, z_hi = sol["w"], sol["lam"], sol["z_lo"], sol["z_hi"]
    f, gf, g, J, _ = evalf(w, lam)
    stat = gf + J.T @ lam + z_hi - z_lo
    with np.errstate(invalid="ignore"):
        comp = max(np.abs(np.where(np.isfinite(lo), z_lo * (w - lo), 0.0)).max(), np.abs(np.where(np.isfinite(hi), z_hi * (hi - w), 0.0)).max())
    viol = float(np.maximum(np.maximum(lo - w, w - hi), 0.0).max())
    return dict(stat=float(np.abs(stat).max()), eq=float(np.abs(g).max(initial=0.0)), viol=viol, comp=float(comp),
                dual=float(max(np.maximum(-z_lo, 0).max(), np.maximum(-z_hi, 0).max())))


# ---------------------------------------------------------------------------------------------------
# OCP (opt_dyn with ContForm)
# ---------------------------------------------------------------------------------------------------
def ocp_eval(p, xhat, xs, us, d, t=0.0):
    """The OCP's evalf and bounds; with user inequality rows (``User_g_ineq``: the g4 rows G_k <= 0 of Control_Calc.py:132-147,g_lb = -inf :246-247) in the form the
    reference's solver gives them internally [ext]: one slack variable per row, G_k(x_k, u_k) - s_k = 0 with s_k <= 0, appended to ``w`` as [w; s_0; ...; s_{N-1}]."""
    evalf0, lo0, hi0 = _ocp_eval_plain(p, xhat, xs, us, d, t)
    if getattr(p, "g_ineq", None) is None:
        return evalf0, lo0, hi0
    n, m, N = p.nx, p.nu, p.N
    nz = n + m
    nw = nz * N + n
    ng = g_rows(p, np.zeros((n, 1)), np.zeros((m, 1)), d.reshape(-1, 1), t).shape[0]
    D = d.reshape(-1, 1)

    def evalf(wa, lam):
        w, sv = wa[:nw], wa[nw:].reshape(N, ng)
        m0 = (N + 1) * n
        f, gf, g, J, H = evalf0(w, lam[:m0] if len(lam) else lam)
        ga = np.zeros(m0 + N * ng); Ja = np.zeros((m0 + N * ng, nw + N * ng)); Ha = np.zeros((nw + N * ng, nw + N * ng))
        ga[:m0] = g; Ja[:m0, :nw] = J
        if len(lam):
            Ha[:nw, :nw] = H
        for k in range(N):
            zk = w[nz * k: nz * (k + 1)]
            gv, gj = jac_cs(lambda Zc: g_rows(p, Zc[:n], Zc[n:], D, t), zk)
            r = slice(m0 + ng * k, m0 + ng * (k + 1))
            ga[r] = gv - sv[k]
            Ja[r, nz * k: nz * (k + 1)] = gj
            Ja[r, nw + ng * k: nw + ng * (k + 1)] = -np.eye(ng)
            if len(lam):
                lg = lam[r]
                Ha[nz * k: nz * (k + 1), nz * k: nz * (k + 1)] += hess_fd(lambda Zc: (lg[:, None] * g_rows(p, Zc[:n], Zc[n:], D, t)).sum(axis=0), zk)
        return f, np.concatenate([gf, np.zeros(N * ng)]), ga, Ja, Ha
    evalf.nw, evalf.ng = nw, ng
    return evalf, np.concatenate([lo0, np.full(N * ng, -INF)]), np.concatenate([hi0, np.zeros(N * ng)])


def _ocp_eval_plain(p, xhat, xs, us, d, t=0.0):
    """evalf of the OCP in opt_dyn's own layout w = [x0,u0,x1,...,x_N] (Control_Calc.py:31-37); g = [x0 - X0; X_{k+1} - F_k ...]
    (:126,156); f = sum of the interval quadratures + Vfin(X_N, xs) (:158,194-210)."""
    n, m, N = p.nx, p.nu, p.N
    nz = n + m
    nw = nz * N + n

    def evalf(w, lam):
        X = np.array([w[nz * k: nz * k + n] for k in range(N + 1)]); U = np.array([w[nz * k + n: nz * (k + 1)] for k in range(N)])
        Lm = lam.reshape(N + 1, n)[1:] if len(lam) else np.zeros((N, n))      # multipliers of X_{k+1} - F_k = 0
        D = np.repeat(d.reshape(-1, 1), 1, axis=1)
        # values and first derivatives of every interval: N * (nz + 1) points in one pass
        Zb = np.hstack([np.concatenate([X[k], U[k]]).astype(complex).reshape(nz, 1).repeat(nz + 1, axis=1) for k in range(N)])
        for k in range(N):
            Zb[np.arange(nz), k * (nz + 1) + np.arange(nz)] += 1j * CS
        Xn, q = ocp_stage(p, Zb[:n], Zb[n:], D, xs.reshape(-1, 1), us.reshape(-1, 1), t)
        Xn = Xn.reshape(n, N, nz + 1); q = q.reshape(N, nz + 1)
        F = Xn[:, :, nz].real.T; AB = np.transpose(Xn[:, :, :nz].imag / CS, (1, 0, 2)); ql = q[:, nz].real; gq = q[:, :nz].imag / CS
        f = float(ql.sum())
        gf = np.zeros(nw); g = np.zeros((N + 1) * n); J = np.zeros(((N + 1) * n, nw)); H = np.zeros((nw, nw))
        g[:n] = xhat - X[0]; J[:n, :n] = -np.eye(n)
        for k in range(N):
            sl = slice(nz * k, nz * (k + 1))
            gf[sl] += gq[k]
            r = slice(n * (k + 1), n * (k + 2))
            g[r] = X[k + 1] - F[k]
            J[r, sl] = -AB[k]; J[r, nz * (k + 1): nz * (k + 1) + n] = np.eye(n)
        if len(lam):
            # Hessian of q_k - lam_{k+1}' F_k with respect to (x_k, u_k), every interval in one pass
            def phi(Zall):      # [nz, N * P] -> [N * P]
                Pk = Zall.shape[1] // N
                Xn_, q_ = ocp_stage(p, Zall[:n], Zall[n:], D, xs.reshape(-1, 1), us.reshape(-1, 1), t)
                lm = np.repeat(Lm, Pk, axis=0).T      # [n, N * Pk]
                return q_ - (lm * Xn_).sum(axis=0)
            # assemble the per-interval stencils side by side
            zs = [np.concatenate([X[k], U[k]]) for k in range(N)]
            dz = [1e-5 * np.maximum(1.0, np.abs(z)) for z in zs]
            cols = []
            for k in range(N):
                blk = []
                for j in range(nz):
                    for sgn in (1.0, -1.0):
                        zz = zs[k].copy(); zz[j] += sgn * dz[k][j]
                        Zc = np.repeat(zz.astype(complex).reshape(nz, 1), nz, axis=1)
                        Zc[np.arange(nz), np.arange(nz)] += 1j * CS
                        blk.append(Zc)
                cols.append(np.hstack(blk))
            Gr = np.asarray(phi(np.hstack(cols))).imag / CS
            Gr = Gr.reshape(N, nz, 2, nz)
            for k in range(N):
                Hk = (Gr[k, :, 0, :] - Gr[k, :, 1, :]) / (2.0 * dz[k][:, None])
                H[nz * k: nz * (k + 1), nz * k: nz * (k + 1)] = 0.5 * (Hk + Hk.T)
        # terminal cost
        xN = X[N]
        if p.vfin is not None:
            vf = lambda Zc: np.array([p.vfin(Zc[:, i], xs) for i in range(Zc.shape[1])])
            v, gv = jac_cs(lambda Zc: vf(Zc)[None], xN)
            f += float(v[0]); gf[nz * N:] += gv[0]
            if len(lam):
                H[nz * N:, nz * N:] = hess_fd(vf, xN)
        return f, gf, g, J, H
    lo = np.full(nw, -INF); hi = np.full(nw, INF)
    lo[:n] = hi[:n] = xhat                                   # MPC_code.py:734
    for k in range(1, N + 1):                                # Control_Calc.py:248-252
        lo[nz * k: nz * k + n], hi[nz * k: nz * k + n] = p.xmin, p.xmax
        lo[nz * k - m: nz * k], hi[nz * k - m: nz * k] = p.umin, p.umax
    return evalf, lo, hi


def ocp_solve(p, xhat, xs, us, d, w_guess, max_iter=None, tol=1e-8, t=0.0):
    max_iter = p.max_iter if max_iter is None else max_iter
    evalf, lo, hi = ocp_eval(p, xhat, xs, us, d, t)
    w0 = np.array(w_guess, dtype=float)
    w0[:p.nx] = xhat
    if getattr(p, "g_ineq", None) is not None:
        # the solver's own slacks start from the rows' values at the PUSHED first iterate and are then pushed into their bound themselves (IpDefaultIterateInitializer [ext])
        nw, ng, n, m = evalf.nw, evalf.ng, p.nx, p.nu
        fixed = lo[:nw] == hi[:nw]
        with np.errstate(invalid="ignore"):
            lor = np.where(np.isfinite(lo[:nw]), lo[:nw] - BOUND_RELAX_FACTOR * np.maximum(1.0, np.abs(lo[:nw])), lo[:nw])
            hir = np.where(np.isfinite(hi[:nw]), hi[:nw] + BOUND_RELAX_FACTOR * np.maximum(1.0, np.abs(hi[:nw])), hi[:nw])
        wp = np.where(fixed, lo[:nw], push_interior(w0[:nw], lor, hir))
        s0 = [g_rows(p, wp[(n + m) * k: (n + m) * k + n].reshape(-1, 1), wp[(n + m) * k + n: (n + m) * (k + 1)].reshape(-1, 1), d.reshape(-1, 1), t)[:, 0].real for k in range(p.N)]
        w0 = np.concatenate([w0[:nw], np.concatenate(s0)])
    sol = ipm_dense(evalf, w0, lo, hi, tol=tol, max_iter=max_iter)
    sol["evalf"], sol["lo"], sol["hi"] = evalf, lo, hi
    if getattr(p, "g_ineq", None) is not None:
        sol["w_all"] = sol["w"]; sol["w"] = sol["w"][:evalf.nw]      # (the loop shifts opt_dyn's own w; the slacks are the solver's)
    return sol


# ---------------------------------------------------------------------------------------------------
# target (opt_ss with User_fssobj)
# ---------------------------------------------------------------------------------------------------
def target_eval(p, d, usp, ysp, xsp, t=0.0):
    n, m, q = p.nx, p.nu, p.ny
    nv = n + m + q

    def cons(Wc):      # [nv, P] -> [n + q, P]: Fx_model(xs,us,h,d) - xs; Fy_model(xs,us,d) - ys   (Target_Calc.py:73-81; lambdaT = 0)
        P = Wc.shape[1]
        D = _cols(d, P)
        return np.vstack([fx_model(p, Wc[:n], Wc[n:n + m], D, t) - Wc[:n], fy_model(p, Wc[:n], D) - Wc[n + m:]])

    def cost(Wc):      # Fss_obj(dx, du, dy, xsp, usp, ysp) with QForm_ss False (Target_Calc.py:109-124)
        return np.array([p.fssobj(Wc[:n, i], Wc[n:n + m, i], Wc[n + m:, i], xsp, usp, ysp) for i in range(Wc.shape[1])])

    def evalf(w, lam):
        g, J = jac_cs(cons, w)
        f, gf = jac_cs(lambda Wc: cost(Wc)[None], w)
        H = np.zeros((nv, nv))
        if len(lam):
            H = hess_fd(lambda Wc: cost(Wc) + (lam[:, None] * cons(Wc)).sum(axis=0), w)
        return float(f[0]), gf[0], g, J, H
    lo = np.concatenate([p.xmin_ss, p.umin_ss, p.ymin_ss]); hi = np.concatenate([p.xmax_ss, p.umax_ss, p.ymax_ss])
    return evalf, lo, hi


def target_solve(p, d, t=0.0, usp=None, ysp=None, xsp=None, max_iter=None, tol=1e-8):
    max_iter = p.max_iter if max_iter is None else max_iter
    usp = np.zeros(p.nu) if usp is None else usp; ysp = np.zeros(p.ny) if ysp is None else ysp; xsp = np.zeros(p.nx) if xsp is None else xsp
    evalf, lo, hi = target_eval(p, d, usp, ysp, xsp, t)
    y0 = fy_model(p, p.x0_m.reshape(-1, 1), d.reshape(-1, 1))[:, 0]
    w0 = np.concatenate([p.x0_m, p.u0, y0])                 # MPC_code.py:696-700: cold start, every step
    sol = ipm_dense(evalf, w0, lo, hi, tol=tol, max_iter=max_iter, resto=True)      # (the target: with the restoration phase, as in the product)
    sol["evalf"], sol["lo"], sol["hi"] = evalf, lo, hi
    sol["xs"], sol["us"], sol["ys"] = sol["w"][:p.nx], sol["w"][p.nx:p.nx + p.nu], sol["w"][p.nx + p.nu:]
    return sol


# ---------------------------------------------------------------------------------------------------
# extended Kalman filter on the augmented state (Estimator.py:313-386 with the Fx_es / Fy_es of MPC_code.py:546-561)
# ---------------------------------------------------------------------------------------------------
def ekf_step(p, P_min, x_es, y_k, u_k, t_k=0.0):
    """ekf(Fx_es, Fy_es, y, u, Q, R, P_min, xhat_min, h, t): returns P(k+1|k), P(k|k), [x; d](k|k).  Fy_es(csi) = x + Cd d, so C = [I, Cd];
    Fx_es(csi, u) = [Fx_model(x, u, d); d], linearised at the CORRECTED estimate (:371-379)."""
    n, nd = p.nx, p.nd
    C = np.hstack([np.eye(p.ny, n), p.Cd])                                   # jac Fy_x (:343-350; StateFeedback: ny = nx)
    yhat = x_es[:n] + p.Cd @ x_es[n:]                                        # :339
    K = P_min @ C.T @ np.linalg.inv(C @ P_min @ C.T + p.R_kf)                # :356-357
    P_corr = P_min - K @ C @ P_min                                           # :360
    x_corr = x_es + K @ (y_k - yhat)                                         # :363-369
    _, A = jac_cs(lambda Zc: np.vstack([fx_model(p, Zc[:n], np.asarray(u_k, dtype=float).reshape(-1, 1), Zc[n:], t_k), Zc[n:]]), x_corr)      # :372-379
    P_plus = A @ P_corr @ A.T + p.Q_kf                                       # :382
    return P_plus, P_corr, x_corr


# ---------------------------------------------------------------------------------------------------
# MHE (mhe_opt's NLP + mhe()'s bookkeeping)
# ---------------------------------------------------------------------------------------------------
def mhe_eval(p, N, Us, Ys, x_bar, Pinv, t=0.0):
    """mhe_opt's NLP for a window of N stages in its own layout w = [x0,v0,w0,x1,...,x_N] (Utilities.py:831-846):
    g = [Fy(X_k) + V_k - Y_k; Fx_mhe(X_k, U_k, W_k) - X_{k+1}] (k < N) (:909-926), f = sum F_obj_mhe(W_k, V_k) + arrival cost (:928-945)."""
    ne, q, nw_ = p.nx + p.nd, p.ny, p.n_w
    nb = ne + q + nw_
    nopt = N * nb + ne
    ng = N * (q + ne)

    def evalf(w, lam):
        f = 0.0; gf = np.zeros(nopt); g = np.zeros(ng); J = np.zeros((ng, nopt)); H = np.zeros((nopt, nopt))
        Lm = lam.reshape(N, q + ne) if len(lam) else None
        for k in range(N):
            o0 = nb * k
            X, V, W = w[o0:o0 + ne], w[o0 + ne:o0 + ne + q], w[o0 + ne + q:o0 + nb]
            zk = np.concatenate([X, W])
            Uk = Us[k].reshape(-1, 1)
            Fv, Fj = jac_cs(lambda Zc: fx_mhe(p, Zc[:ne], Uk, Zc[ne:], t), zk)
            yv, yj = jac_cs(lambda Zc: fy_es(p, Zc), X)
            cv, cj = jac_cs(lambda Zc: np.array([p.fobj_mhe(Zc[:nw_, i], Zc[nw_:, i], t) for i in range(Zc.shape[1])])[None], np.concatenate([W, V]))
            f += float(cv[0])
            gf[o0 + ne + q:o0 + nb] += cj[0][:nw_]; gf[o0 + ne:o0 + ne + q] += cj[0][nw_:]
            r0 = (q + ne) * k
            g[r0:r0 + q] = yv + V - Ys[k]
            J[r0:r0 + q, o0:o0 + ne] = yj; J[r0:r0 + q, o0 + ne:o0 + ne + q] = np.eye(q)
            g[r0 + q:r0 + q + ne] = Fv - w[o0 + nb:o0 + nb + ne]
            J[r0 + q:r0 + q + ne, o0:o0 + ne] = Fj[:, :ne]; J[r0 + q:r0 + q + ne, o0 + ne + q:o0 + nb] = Fj[:, ne:]
            J[r0 + q:r0 + q + ne, o0 + nb:o0 + nb + ne] = -np.eye(ne)
            if Lm is not None:
                Hc = hess_fd(lambda Zc: np.array([p.fobj_mhe(Zc[:nw_, i], Zc[nw_:, i], t) for i in range(Zc.shape[1])]), np.concatenate([W, V]))
                iw = np.arange(o0 + ne + q, o0 + nb); iv = np.arange(o0 + ne, o0 + ne + q)
                idx = np.concatenate([iw, iv])
                H[np.ix_(idx, idx)] += Hc
                lf, ly = Lm[k, q:], Lm[k, :q]
                Hd = hess_fd(lambda Zc: (lf[:, None] * fx_mhe(p, Zc[:ne], Uk, Zc[ne:], t)).sum(axis=0) + (ly[:, None] * fy_es(p, Zc[:ne])).sum(axis=0), zk)
                idz = np.concatenate([np.arange(o0, o0 + ne), iw])
                H[np.ix_(idz, idz)] += Hd
        e0 = w[:ne] - x_bar
        f += 0.5 * float(e0 @ Pinv @ e0); gf[:ne] += Pinv @ e0; H[:ne, :ne] += 0.5 * (Pinv + Pinv.T)
        return f, gf, g, J, H
    lo = np.full(nopt, -INF); hi = np.full(nopt, INF)
    for k in range(N + 1):                                   # Utilities.py:956-966: the state boxes on every X_k
        lo[nb * k: nb * k + ne], hi[nb * k: nb * k + ne] = p.xmin_mhe, p.xmax_mhe
    for k in range(N):                                       # :968-977: the boxes of the state noise
        lo[nb * k + ne + q: nb * (k + 1)], hi[nb * k + ne + q: nb * (k + 1)] = p.wmin, p.wmax
    return evalf, lo, hi


class MheState:
    """What mhe() carries from call to call (Estimator.py:388-768 arguments / MPC_code.py:405-438 initial values)."""

    def __init__(self, p):
        ne = p.nx + p.nd
        self.U, self.Y, self.T = [], [], []
        self.w_k, self.v_k = np.zeros(p.n_w), np.zeros(p.ny)
        self.x_bar, self.P_k = p.x_bar.copy(), p.P0.copy()
        self.bigA, self.bigP, self.bigPc = [], [], []
        self.P_kal = p.P0.copy()
        self.X, self.V, self.W = [], [], []                  # the lists of x(k+1|k), v_k, w_k the 'filter' update reads (:541-554)
        self.last = None


def mhe_step(p, S, ksim, y_act, u_k, t_k=0.0, max_iter=None, tol=1e-10):
    max_iter = p.max_iter if max_iter is None else max_iter      # ipopt.tol = 1e-10 for the estimator (MPC_code.py:383)
    """One call of mhe() (Estimator.py:388-768), ``mhe_up = 'smooth'`` or ``'filter'``; returns the corrected estimate [x; d](k|k)."""
    ne, q, nw_, m = p.nx + p.nd, p.ny, p.n_w, p.nu
    nb = ne + q + nw_
    N = min(ksim + 1, p.N_mhe)                               # MPC_code.py:591-593
    # stacking (Estimator.py:474-501): the last input is doubled, the copy is the fictitious input of the prediction stage
    if ksim < p.N_mhe:
        S.U = S.U + ([u_k.copy()] if ksim == 0 else [u_k.copy(), u_k.copy()])
        S.Y = S.Y + [y_act.copy()]; S.T = S.T + [t_k]
    else:
        S.U = S.U[1:] + [u_k.copy(), u_k.copy()]
        S.Y = S.Y[1:] + [y_act.copy()]; S.T = S.T[1:] + [t_k]
    assert len(S.U) == N and len(S.Y) == N
    # first guess: x_bar propagated without noise (:503-512)
    w0 = np.zeros(N * nb + ne)
    xg = S.x_bar.copy()
    for k in range(N):
        w0[nb * k: nb * k + ne] = xg
        xg = fx_mhe(p, xg.reshape(-1, 1), S.U[k].reshape(-1, 1), np.zeros((nw_, 1)), t_k)[:, 0]
    w0[N * nb:] = xg
    Pinv = np.linalg.inv(S.P_k)                              # :515-517
    evalf, lo, hi = mhe_eval(p, N, S.U, S.Y, S.x_bar, Pinv, t_k)
    sol = ipm_dense(evalf, w0, lo, hi, tol=tol, max_iter=max_iter)
    sol["evalf"], sol["lo"], sol["hi"] = evalf, lo, hi
    S.last = sol
    w = sol["w"]
    xkp1k = w[-ne:]; xhat_corr = w[-ne - nb:-nb]            # :532-534
    S.v_k = w[-nb:-ne - nw_].copy()
    if ksim != 0:
        S.w_k = w[-ne - nw_:-ne].copy()                      # :536-538
    # the lists of one-step predictions and noises (:541-554)
    if ksim < p.N_mhe:
        S.X, S.V, S.W = S.X + [xkp1k.copy()], S.V + [S.v_k.copy()], S.W + [S.w_k.copy()]
    else:
        S.X, S.V, S.W = S.X[1:] + [xkp1k.copy()], S.V[1:] + [S.v_k.copy()], S.W[1:] + [S.w_k.copy()]

    def ekf_covariance(P_in, w_, v_, x_c, x_a, u_, t_):
        """P(k|k) and P(k+1|k) of an extended Kalman step with correlated noises, weights from the estimator's cost (:557-623 and
        :629-649 are this with different arguments): C at x_c, A and G at (x_a, u_, w_)."""
        Hk = np.linalg.inv(hess_fd(lambda Zc: np.array([p.fobj_mhe(Zc[:nw_, i], Zc[nw_:, i], t_) for i in range(Zc.shape[1])]), np.concatenate([w_, v_])))
        Q_k, R_k, S_k = Hk[:nw_, :nw_], Hk[-q:, -q:], Hk[:nw_, -q:]
        _, C_k = jac_cs(lambda Zc: fy_es(p, Zc), x_c)
        _, Fj = jac_cs(lambda Zc: fx_mhe(p, Zc[:ne], u_.reshape(-1, 1), Zc[ne:], t_), np.concatenate([x_a, w_]))
        A_k, G_k = Fj[:, :ne], Fj[:, ne:]
        K_k = P_in @ C_k.T @ np.linalg.inv(C_k @ P_in @ C_k.T + R_k)
        P_corr = P_in - K_k @ C_k @ P_in
        M_k = -K_k @ S_k.T
        return A_k, P_corr, A_k @ P_corr @ A_k.T + G_k @ Q_k @ G_k.T + A_k @ M_k @ G_k.T + G_k @ M_k @ A_k.T

    if p.mhe_up == "filter":
        # (:557-623 also run for 'filter' in the reference, into lists nothing reads afterwards: left out here)
        if ksim >= p.N_mhe - 1:                              # :627-649: one Kalman step on the prior weight, at the window's first entries
            # (C is evaluated at Xmin[0] in the reference: Fy_es is linear in the state here, so the point does not matter)
            _, _, S.P_k = ekf_covariance(S.P_k, S.W[0], S.V[0], S.X[0], S.X[0], S.U[0], S.T[0])
            S.x_bar = S.X[0].copy()                          # :740-748
        S.U = [] if ksim == 0 else S.U[:-1]
        return xhat_corr.copy()
    # Kalman quantities for the smoothing update (:558-623)
    Pi = S.P_kal
    A_k, P_corr, S.P_kal = ekf_covariance(S.P_kal, S.w_k, S.v_k, xhat_corr, xhat_corr, u_k, t_k)
    S.bigA.append(A_k); S.bigP.append(Pi); S.bigPc.append(P_corr)
    if ksim >= p.N_mhe - 1:                                  # :626-665: smoothed covariance of the window's second state
        Nm = p.N_mhe
        Pis = [None] * Nm
        Pis[Nm - 1] = S.bigPc[Nm - 1]
        for i in range(Nm - 2, -1, -1):
            Pim = np.linalg.inv(S.bigP[i + 1])
            Pis[i] = S.bigPc[i] + S.bigPc[i] @ S.bigA[i].T @ Pim @ (Pis[i + 1] - S.bigP[i + 1]) @ Pim @ S.bigA[i] @ S.bigPc[i]
        S.P_k = Pis[1]
        S.bigA, S.bigP, S.bigPc = S.bigA[1:], S.bigP[1:], S.bigPc[1:]
        S.x_bar = w[nb:nb + ne].copy()                       # :750-753
    S.U = [] if ksim == 0 else S.U[:-1]                      # :756-760
    return xhat_corr.copy()


# ---------------------------------------------------------------------------------------------------
# closed loop (MPC_code.py:485-827)
# ---------------------------------------------------------------------------------------------------
def closed_loop(p, nsteps, x0_p=None, x0_m=None, certify=False, verbose=False, v_wn=None, w_wn=None):
    n, m, nd, N = p.nx, p.nu, p.nd, p.N
    nz = n + m
    x_k = (p.x0_p if x0_p is None else np.asarray(x0_p, dtype=float)).copy()
    x0_m = (p.x0_m if x0_m is None else np.asarray(x0_m, dtype=float)).copy()
    xhat = x0_m.copy(); dhat = np.zeros(nd); u_k = p.u0.copy()
    S = MheState(p) if p.mhe else None
    P_k = None if p.mhe else p.P0.copy()                    # MPC_code.py:455-458
    if S is not None and x0_m is not None:
        S.x_bar[:n] = x0_m                                   # Ex-file: x_bar = [x0_m; 0]
    xs_k, us_k = x0_m.copy(), u_k.copy()                     # MPC_code.py:682-684
    w_opt = None; last_ok = True
    log = {k: [] for k in ("U", "X_HAT", "D_HAT", "XS", "US", "Xp", "Yp", "STATUS_DYN", "STATUS_SS", "ITERS_DYN", "ITERS_SS", "ITERS_MHE", "STATUS_MHE", "X_ES", "KKT_DYN", "KKT_SS", "KKT_MHE", "P_K")}
    for ksim in range(nsteps):
        t_k = ksim * p.h
        log["Xp"].append(x_k.copy()); log["X_HAT"].append(xhat.copy())
        y_k = x_k.copy()                                     # Fy_p with StateFeedback (Utilities.py:84-86)
        if v_wn is not None:
            y_k = y_k + v_wn[ksim]                           # white noise on the measurement, MPC_code.py:537-541 (the draws, sqrtm(R_wn) applied)
        log["Yp"].append(y_k.copy())
        if p.mhe:
            x_es = mhe_step(p, S, ksim, y_k, u_k, t_k)
            log["ITERS_MHE"].append(S.last["iters"]); log["STATUS_MHE"].append(S.last["status"])
            if certify:
                log["KKT_MHE"].append(max(kkt_nlp(S.last["evalf"], S.last, S.last["lo"], S.last["hi"]).values()))
            log["P_K"].append(S.P_k.copy())
        else:                                                # MPC_code.py:640-650: estype 'ekf', P_k <- P_plus
            P_k, _, x_es = ekf_step(p, P_k, np.concatenate([xhat, dhat]), y_k, u_k, t_k)
            log["ITERS_MHE"].append(0); log["STATUS_MHE"].append(0); log["P_K"].append(P_k.copy())
        xhat, dhat = x_es[:n].copy(), x_es[n:].copy()
        if p.dmin is not None:
            dhat = np.minimum(np.maximum(dhat, p.dmin), p.dmax)
        log["X_ES"].append(x_es.copy()); log["D_HAT"].append(dhat.copy())
        us_prev, xs_prev = us_k.copy(), xs_k.copy()
        ts = target_solve(p, dhat, t_k)
        if ts["status"] != STATUS_INFEASIBLE:
            xs_k, us_k = ts["xs"].copy(), ts["us"].copy()
        log["XS"].append(xs_k.copy()); log["US"].append(us_k.copy()); log["STATUS_SS"].append(ts["status"]); log["ITERS_SS"].append(ts["iters"])
        if certify:
            log["KKT_SS"].append(max(kkt_nlp(ts["evalf"], ts, ts["lo"], ts["hi"]).values()))
        if w_opt is None:                                    # MPC_code.py:740-756
            w_guess = np.zeros(nz * N + n)
            for k in range(1, N + 1):
                w_guess[k * nz - m:k * nz] = p.u0; w_guess[k * nz:k * nz + n] = x0_m
            w_guess[:n] = x0_m
        elif last_ok:
            w_guess = np.concatenate([w_opt[nz:], us_prev, xs_prev])      # :764
        sol = ocp_solve(p, xhat, xs_k, us_k, dhat, w_guess, t=t_k)
        last_ok = sol["status"] != STATUS_INFEASIBLE
        log["STATUS_DYN"].append(sol["status"]); log["ITERS_DYN"].append(sol["iters"])
        if last_ok:
            w_opt = sol["w"]
            u_k = w_opt[n:nz].copy(); xhat = w_opt[nz:nz + n].copy()      # :798-799
            if certify:
                log["KKT_DYN"].append(max(kkt_nlp(sol["evalf"], dict(sol, w=sol.get("w_all", sol["w"])), sol["lo"], sol["hi"]).values()))
        else:
            xhat = fx_model(p, xhat.reshape(-1, 1), u_k.reshape(-1, 1), dhat.reshape(-1, 1), t_k)[:, 0]      # :804-805
            if certify:
                log["KKT_DYN"].append(np.nan)
        log["U"].append(u_k.copy())
        if verbose:
            print(f"step {ksim}: u {u_k}, xs {xs_k}, us {us_k}, iters {sol['iters']} / {ts['iters']} / {log['ITERS_MHE'][-1] if p.mhe else 0}, status {sol['status']}")
        x_k = fx_plant(p, x_k.reshape(-1, 1), u_k.reshape(-1, 1), t_k)[:, 0]      # :813-816
        if w_wn is not None:
            x_k = x_k + w_wn[ksim]                           # white noise on the state, :822-827 (G_wn sqrtm(Q_wn) applied)
    return {k: np.array(v) for k, v in log.items() if len(v)}
