"""ORACLE (test infrastructure, never shipped): float64 NumPy restatement of the hot path of
CPCLAB-UNIPI/MPC-code for matrix-defined linear examples.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  The product (``mpc-code_amd/``) must never do so.

PARITY STATUS: **unpinned against the reference implementation.**  The reference delegates
every floating-point operation of this path to CasADi 3.4 / IPOPT 3.12 / MUMPS
(``Control_Calc.py:258``, ``Target_Calc.py:159``), none of which exists under
``/root/reference`` or in this image, and the reference ships no tests or golden vectors
(SURVEY.md section 4, section 8c).  What pins this oracle instead:

* the dynamic problem is a strictly convex QP in ``u`` (``R`` or ``S`` positive definite), the
  target problem has a positive definite reduced Hessian, so the optimum each step is unique
  and any solver must return it; every solution produced here carries its KKT residual
  (:func:`kkt_residual`) so fixtures are self-certifying;
* the LQR known answer ``u0* = us + K (xhat - xs)`` with ``P`` from
  ``scipy.linalg.solve_discrete_are`` - the same library call the reference makes
  (``Utilities.py:409``) - when no bound is active;
* an independent second solver (SciPy ``trust-constr`` / HiGHS feasibility) in the tests.

The problem statement follows the reference line by line:

=====================  ==========================================================
``ocp_qp``             ``opt_dyn``, Control_Calc.py:20-260 (variable layout ``:31-37``,
                       parameter vector ``:43-52``, rows ``:126-171``, bounds ``:213-252``)
``target_qp``          ``opt_ss``, Target_Calc.py:20-161
``kalman`` / ``kalss`` Estimator.py:263-311 / :231-261
``model_fx/model_fy``  ``defF_model``, Utilities.py:135-155, :208-244
``plant_fx/plant_fy``  ``defF_p``, Utilities.py:45-49, :88-91
``closed_loop``        MPC_code.py:485-827 (order: measure, estimate, target, OCP, plant)
=====================  ==========================================================

The QP solver itself (:func:`qp_ipm_dense`) is a textbook Mehrotra predictor-corrector on the
*dense* KKT system - deliberately a different factorisation from the Riccati recursion the
HIP kernels use, so that agreement between the two is evidence and not tautology.
"""
from __future__ import annotations

import numpy as np

INF = np.inf
STATUS_SOLVED, STATUS_MAXITER, STATUS_INFEASIBLE = 0, 1, 2
BOUND_RELAX = 1e-8   # IPOPT's bound_relax_factor default [ext]; used only for the stage-0 output rows


# ----------------------------------------------------------------------------------------
# model / plant maps
# ----------------------------------------------------------------------------------------
def model_fx(p, x, u, d, px=None):
    """Fx_model: A(x-xlin)+B(u-ulin)+xlin + Bd d + px   (Utilities.py:135-155)."""
    out = p.A @ x + p.B @ u + p.fx_const
    if p.nd:
        out = out + p.Bd @ d
    return out if px is None else out + px


def model_fy(p, x, d, py=None):
    """Fy_model: C(x-xlin)+ylin + Cd d + py             (Utilities.py:208-244)."""
    out = p.C @ x + p.fy_const
    if p.nd:
        out = out + p.Cd @ d
    return out if py is None else out + py


def plant_fx(p, xp, u, pxp, t=0.0):
    """Fx_p: Ap x + Bp u + pxp + pxmp (pxmp = 0 without def_px)   (Utilities.py:45-49); a non-linear continuous plant
    (Utilities.py:58-82) is integrated by the problem object's RK4 (host code, outside the hot path)."""
    if getattr(p, "plant_fx_cont", None) is not None:
        return p.plant_step(np.asarray(xp)[None], np.asarray(u)[None], t, pxp)[0]
    return p.Ap @ xp + p.Bp @ u + pxp


def plant_fy(p, xp, pyp):
    """Fy_p: Cp x + pyp + pymp (pymp = 0 without def_py)           (Utilities.py:88-91)."""
    return p.Cp @ xp + pyp


# ----------------------------------------------------------------------------------------
# dense QP data in the reference's own layouts
# ----------------------------------------------------------------------------------------
def ocp_qp(p, xhat, xs, us, dhat, u_prev, drop_stage0_rows=False, px=None, py=None):
    """QP of ``opt_dyn`` in its own variable order w=[x0,u0,...,x_{N-1},u_{N-1},x_N].

    Returns ``H, g, E, e, G, lo, hi``:  min 1/2 w'Hw + g'w  s.t.  E w = e,  lo <= G w <= hi.
    Equality rows: ``x0 - X[0]`` (Control_Calc.py:126) then ``X_next - X[k+1]`` (:171).
    Inequality rows: variable bounds on x_1..x_N, u_0..u_{N-1} (:248-252; x_0 is fixed by
    ``w_lb[0:nx]=w_ub[0:nx]=xhat``, MPC_code.py:734, so its bound rows are dropped), then the
    ``g1`` rows ``ymin <= C x_k + Cd d + .. <= ymax`` for k=0..N-1 (:130,150-151,229-230).
    ``px`` [N, nx] / ``py`` [N, ny]: the model parameters over the horizon, ``par_xmk[:, k]`` in the dynamics and ``par_ymk[:, k]`` in the
    output rows (Control_Calc.py:43-57,130,161).
    ``drop_stage0_rows`` leaves out the k=0 rows: they constrain the given x_0, i.e. they are a feasibility
    test that :func:`ocp_solve` makes up front with IPOPT's bound relaxation.

    ``p.slacks`` (soft constraints, Control_Calc.py:39-40,186-192,228-239; ``Default_Values.py:128``): ONE vector ``Sl = [sl_ub (ny), sl_lb (ny)] >= 0`` appended to
    ``w`` (:31,39-40), shared by all stages, penalised ``Sl' Ws Sl`` in EVERY stage's cost (:186-188: N times) and widening every stage's output rows,
    ``ymin - Y_k - sl_lb <= 0``, ``-ymax + Y_k - sl_ub <= 0`` (:231-239; k = 0 included: there the row only bounds the slack from below).  A missing side of
    the output box is +-1e12 with slacks on (:64-72).  The state and input bounds stay hard (:213-252).
    """
    n, m, N = p.nx, p.nu, p.N
    nxu = n + m
    soft = bool(getattr(p, "slacks", False))
    ns = 2 * p.ny if soft else 0
    nw = nxu * N + n + ns
    H = np.zeros((nw, nw))
    g = np.zeros(nw)
    ix = lambda k: slice(nxu * k, nxu * k + n)
    iu = lambda k: slice(nxu * k + n, nxu * k + nxu)
    Q, R, P = p.Q, p.R, p.P
    for k in range(N):
        # F_obj(dx,du,..) = 1/2 (dx'Q dx + du'R du), dx = x_k - xs (Control_Calc.py:173-188)
        H[ix(k), ix(k)] += Q
        g[ix(k)] += -Q @ xs
        if not p.DUForm:
            H[iu(k), iu(k)] += R
            g[iu(k)] += -R @ us
        else:  # du = U[k]-U[k-1], or U[0]-um1 (Control_Calc.py:163-166,180-181)
            H[iu(k), iu(k)] += R
            if k == 0:
                g[iu(k)] += -R @ u_prev
            else:
                H[iu(k - 1), iu(k - 1)] += R
                H[iu(k), iu(k - 1)] -= R
                H[iu(k - 1), iu(k)] -= R
    H[ix(N), ix(N)] += P                       # Vfin(dx) = 1/2 dx'P dx (Control_Calc.py:194-210)
    g[ix(N)] += -P @ xs
    if soft:
        Ws = np.asarray(p.Ws, dtype=float).reshape(ns, ns)
        H[nw - ns:, nw - ns:] += N * (Ws + Ws.T)  # + Sl' Ws Sl in each of the N stage costs (Control_Calc.py:186-188)
    # equalities
    E = np.zeros((n * (N + 1), nw))
    e = np.zeros(n * (N + 1))
    E[0:n, ix(0)] = np.eye(n)
    e[0:n] = xhat
    c = p.fx_const + (p.Bd @ dhat if p.nd else 0.0)
    for k in range(N):
        r = slice(n * (k + 1), n * (k + 2))
        E[r, ix(k)] = p.A
        E[r, iu(k)] = p.B
        E[r, ix(k + 1)] = -np.eye(n)
        e[r] = -(c + (px[k] if px is not None else 0.0))
    if getattr(p, "TermCons", False):          # g.append(X[N] - xs), Control_Calc.py:193-198
        Et = np.zeros((n, nw)); Et[:, ix(N)] = np.eye(n)
        E = np.vstack([E, Et]); e = np.concatenate([e, xs])
    # inequalities
    rows, lo, hi = [], [], []
    for k in range(1, N + 1):
        for i in range(n):
            if np.isfinite(p.xmin[i]) or np.isfinite(p.xmax[i]):
                row = np.zeros(nw); row[nxu * k + i] = 1.0
                rows.append(row); lo.append(p.xmin[i]); hi.append(p.xmax[i])
    for k in range(N):
        for i in range(m):
            if np.isfinite(p.umin[i]) or np.isfinite(p.umax[i]):
                row = np.zeros(nw); row[nxu * k + n + i] = 1.0
                rows.append(row); lo.append(p.umin[i]); hi.append(p.umax[i])
    if p.y_bounded and soft:
        yc = p.fy_const + (p.Cd @ dhat if p.nd else 0.0)
        ymin = np.where(np.isfinite(p.ymin), p.ymin, -1e12); ymax = np.where(np.isfinite(p.ymax), p.ymax, 1e12)      # Control_Calc.py:64-72
        for k in range(N):                     # (k = 0 stays: it bounds the slack)
            for i in range(p.ny):
                yk = yc[i] + (py[k][i] if py is not None else 0.0)
                row = np.zeros(nw); row[ix(k)] = p.C[i]; row[nw - ns + p.ny + i] = 1.0      # ymin - Y_k - sl_lb <= 0
                rows.append(row); lo.append(ymin[i] - yk); hi.append(np.inf)
                row = np.zeros(nw); row[ix(k)] = p.C[i]; row[nw - ns + i] = -1.0            # -ymax + Y_k - sl_ub <= 0
                rows.append(row); lo.append(-np.inf); hi.append(ymax[i] - yk)
    elif p.y_bounded:
        yc = p.fy_const + (p.Cd @ dhat if p.nd else 0.0)
        for k in range(1 if drop_stage0_rows else 0, N):
            for i in range(p.ny):
                if np.isfinite(p.ymin[i]) or np.isfinite(p.ymax[i]):
                    row = np.zeros(nw); row[ix(k)] = p.C[i]
                    yk = yc[i] + (py[k][i] if py is not None else 0.0)
                    rows.append(row); lo.append(p.ymin[i] - yk); hi.append(p.ymax[i] - yk)
    if getattr(p, "Dumin", None) is not None or getattr(p, "Dumax", None) is not None:
        # g2 rows: DU_k = U[k] - um1 (k = 0) | U[k] - U[k-1], Control_Calc.py:163-169, bounds tiled :241-243
        dlo = p.Dumin if p.Dumin is not None else np.full(m, -np.inf); dhi = p.Dumax if p.Dumax is not None else np.full(m, np.inf)
        for k in range(N):
            for i in range(m):
                if np.isfinite(dlo[i]) or np.isfinite(dhi[i]):
                    row = np.zeros(nw); row[nxu * k + n + i] = 1.0
                    off = u_prev[i] if k == 0 else 0.0
                    if k > 0:
                        row[nxu * (k - 1) + n + i] = -1.0
                    rows.append(row); lo.append(dlo[i] + off); hi.append(dhi[i] + off)
    if getattr(p, "Gx", None) is not None:      # g4 rows: G_ineq(X[k], U[k], Y_k, d, ...) <= 0, k = 0..N-1 (Control_Calc.py:132-147,245); affine on this path, Y_k substituted by the loader
        gc = p.g0 + (p.Gd @ dhat if p.nd else 0.0)
        for k in range(N):
            for i in range(p.Gx.shape[0]):
                row = np.zeros(nw); row[ix(k)] = p.Gx[i]; row[iu(k)] = p.Gu[i]
                rows.append(row); lo.append(-np.inf); hi.append(-gc[i])
    for j in range(ns):                        # w_lb[nw-ns:nw] = 0 (Control_Calc.py:217)
        row = np.zeros(nw); row[nw - ns + j] = 1.0
        rows.append(row); lo.append(0.0); hi.append(np.inf)
    G = np.array(rows) if rows else np.zeros((0, nw))
    return H, g, E, e, G, np.array(lo, dtype=float), np.array(hi, dtype=float)


def target_qp(p, usp, ysp, xsp, dhat, us_prev, px0=None, py0=None):
    """QP of ``opt_ss`` in its own variable order wss=[xs,us,ys] (Target_Calc.py:29-38).

    Cost 1/2 (ys-ysp)'Qss(ys-ysp) + 1/2 dus'Rss dus with dus = us-usp, or us-us_prev when
    DUssForm (:112-124); equalities Fx(xs,us)-xs = 0 and Fy(xs,us)-ys = 0 (:75-81); bounds on
    all three blocks (:127-134).
    """
    n, m, q = p.nx, p.nu, p.ny
    nv = n + m + q
    H = np.zeros((nv, nv)); g = np.zeros(nv)
    sy = slice(n + m, nv); su = slice(n, n + m)
    H[sy, sy] = p.Qss; g[sy] = -p.Qss @ ysp
    H[su, su] = p.Rss; g[su] = -p.Rss @ (us_prev if p.DUssForm else usp)
    E = np.zeros((n + q, nv)); e = np.zeros(n + q)
    E[:n, :n] = p.A - np.eye(n); E[:n, su] = p.B
    e[:n] = -(p.fx_const + (p.Bd @ dhat if p.nd else 0.0) + (px0 if px0 is not None else 0.0))
    E[n:, :n] = p.C; E[n:, sy] = -np.eye(q)
    e[n:] = -(p.fy_const + (p.Cd @ dhat if p.nd else 0.0) + (py0 if py0 is not None else 0.0))
    lo = np.concatenate([p.xmin_ss, p.umin_ss, p.ymin_ss])
    hi = np.concatenate([p.xmax_ss, p.umax_ss, p.ymax_ss])
    keep = np.isfinite(lo) | np.isfinite(hi)
    G = np.eye(nv)[keep]
    return H, g, E, e, G, lo[keep], hi[keep]


# ----------------------------------------------------------------------------------------
# dense Mehrotra predictor-corrector
# ----------------------------------------------------------------------------------------
def _one_sided(G, lo, hi):
    fl, fu = np.isfinite(lo), np.isfinite(hi)
    Gi = np.vstack([-G[fl], G[fu]])
    hh = np.concatenate([-lo[fl], hi[fu]])
    return Gi, hh, fl, fu


def qp_ipm_dense(H, g, E, e, G, lo, hi, tol=1e-11, max_iter=200):
    """min 1/2 w'Hw+g'w, Ew=e, lo<=Gw<=hi by Mehrotra's predictor-corrector, dense KKT solves.

    Returns dict(w, nu, z_lo, z_hi, status, iters, res) where ``res`` is the final
    :func:`kkt_residual`.  status 2 (infeasible) is declared when the iteration diverges or
    stalls with a primal residual that does not vanish; see tests for the LP cross-check.
    """
    nv, ne = H.shape[0], E.shape[0]
    Gi, hh, fl, fu = _one_sided(G, lo, hi)
    mi = Gi.shape[0]
    # start: equality-constrained minimiser of the regularised cost, slacks pushed positive
    KK = np.block([[H + 1e-8 * np.eye(nv), E.T], [E, np.zeros((ne, ne))]])
    sol = np.linalg.lstsq(KK, np.concatenate([-g, e]), rcond=None)[0]
    w, nu = sol[:nv], sol[nv:]
    if mi:
        s = hh - Gi @ w
        s = np.maximum(s, 1.0)
        z = np.ones(mi)
    else:
        s = z = np.zeros(0)
    status, it = STATUS_MAXITER, 0
    scale = max(1.0, np.abs(g).max() if nv else 1.0)
    for it in range(max_iter + 1):
        r_d = H @ w + g + E.T @ nu + (Gi.T @ z if mi else 0.0)
        r_e = E @ w - e
        r_p = Gi @ w + s - hh if mi else np.zeros(0)
        mu = float(s @ z) / mi if mi else 0.0
        if (np.abs(r_d).max(initial=0) <= tol * scale and np.abs(r_e).max(initial=0) <= tol
                and np.abs(r_p).max(initial=0) <= tol and mu <= tol):
            status = STATUS_SOLVED
            break
        if it == max_iter:
            break
        if mi and (z.max() > 1e14 * scale or not np.all(np.isfinite(w))):
            status = STATUS_INFEASIBLE
            break
        D = z / s if mi else np.zeros(0)
        Kmat = np.block([[H + (Gi.T * D) @ Gi if mi else H, E.T], [E, np.zeros((ne, ne))]])
        lu = _lu(Kmat)

        def newton(r_c):
            rhs_w = -r_d - (Gi.T @ ((-r_c + z * r_p) / s) if mi else 0.0)
            dsol = _lu_solve(lu, np.concatenate([rhs_w, -r_e]))
            dw, dnu = dsol[:nv], dsol[nv:]
            if mi:
                ds = -r_p - Gi @ dw
                dz = (-r_c - z * ds) / s
            else:
                ds = dz = np.zeros(0)
            return dw, dnu, ds, dz

        def maxstep(v, dv):
            neg = dv < 0
            return min(1.0, float(np.min(-v[neg] / dv[neg]))) if np.any(neg) else 1.0

        dw, dnu, ds, dz = newton(s * z)
        if mi:
            a_aff = min(maxstep(s, ds), maxstep(z, dz))
            mu_aff = float((s + a_aff * ds) @ (z + a_aff * dz)) / mi
            sigma = (mu_aff / mu) ** 3 if mu > 0 else 0.0
            dw, dnu, ds, dz = newton(s * z - sigma * mu + ds * dz)
            a = min(1.0, 0.995 * min(maxstep(s, ds), maxstep(z, dz)))
        else:
            a = 1.0
        w = w + a * dw; nu = nu + a * dnu
        if mi:
            s = s + a * ds; z = z + a * dz
    if status == STATUS_MAXITER and mi and np.abs(Gi @ w + s - hh).max() > 1e-6:
        status = STATUS_INFEASIBLE
    z_lo = np.zeros(len(lo)); z_hi = np.zeros(len(hi))
    if mi:
        z_lo[fl] = z[: fl.sum()]; z_hi[fu] = z[fl.sum():]
    res = kkt_residual(H, g, E, e, G, lo, hi, w, nu, z_lo, z_hi)
    return dict(w=w, nu=nu, z_lo=z_lo, z_hi=z_hi, status=status, iters=it, res=res)


def _lu(K):
    import scipy.linalg as sl
    return sl.lu_factor(K, check_finite=False)


def _lu_solve(lu, b):
    import scipy.linalg as sl
    x = sl.lu_solve(lu, b, check_finite=False)
    return x


def kkt_residual(H, g, E, e, G, lo, hi, w, nu, z_lo, z_hi):
    """max-norm KKT certificate of (w, nu, z_lo>=0, z_hi>=0):

    stationarity  H w + g + E'nu + G'(z_hi - z_lo),   primal  E w - e,  bound violation,
    complementarity  z_lo (Gw - lo),  z_hi (hi - Gw),  dual sign.
    """
    Gw = G @ w if G.shape[0] else np.zeros(0)
    stat = H @ w + g + E.T @ nu + (G.T @ (z_hi - z_lo) if G.shape[0] else 0.0)
    viol = np.maximum(np.maximum(lo - Gw, Gw - hi), 0.0) if G.shape[0] else np.zeros(0)
    with np.errstate(invalid="ignore"):
        comp_lo = np.where(np.isfinite(lo), z_lo * (Gw - lo), 0.0) if G.shape[0] else np.zeros(0)
        comp_hi = np.where(np.isfinite(hi), z_hi * (hi - Gw), 0.0) if G.shape[0] else np.zeros(0)
    return dict(
        stat=float(np.abs(stat).max(initial=0)), eq=float(np.abs(E @ w - e).max(initial=0)),
        viol=float(viol.max(initial=0)),
        comp=float(max(np.abs(comp_lo).max(initial=0), np.abs(comp_hi).max(initial=0))),
        dual=float(max(np.maximum(-z_lo, 0).max(initial=0), np.maximum(-z_hi, 0).max(initial=0))),
    )


def kkt_max(res):
    return max(res.values())


def lp_feasible(E, e, G, lo, hi, slack=0.0):
    """Definitive feasibility label by HiGHS: is {Ew=e, lo-slack <= Gw <= hi+slack} non-empty?"""
    from scipy.optimize import linprog
    nv = E.shape[1]
    A_ub = np.vstack([G[np.isfinite(hi)], -G[np.isfinite(lo)]])
    b_ub = np.concatenate([hi[np.isfinite(hi)] + slack, -(lo[np.isfinite(lo)] - slack)])
    r = linprog(np.zeros(nv), A_ub=A_ub if len(b_ub) else None, b_ub=b_ub if len(b_ub) else None,
                A_eq=E, b_eq=e, bounds=[(None, None)] * nv, method="highs")
    return r.status == 0


# ----------------------------------------------------------------------------------------
# per-step building blocks with the reference's read-out rules
# ----------------------------------------------------------------------------------------
def ocp_solve(p, xhat, xs, us, dhat, u_prev, tol=1e-11, px=None, py=None):
    """One ``solver(...)`` call of MPC_code.py:776-781 + read-out ``:798-799``.

    Returns dict(u0, x1, w, status, iters, res).  x_0 sits on a g1 row too
    (Control_Calc.py:128-151): if ``C xhat + ..`` violates [ymin,ymax] the problem is infeasible
    whatever u is - reported as status 2 before any iteration.
    """
    H, g, E, e, G, lo, hi = ocp_qp(p, xhat, xs, us, dhat, u_prev, drop_stage0_rows=True, px=px, py=py)
    n, m = p.nx, p.nu
    if p.y_bounded and not getattr(p, "slacks", False):
        y0 = model_fy(p, xhat, dhat, None if py is None else py[0])
        rl = BOUND_RELAX * np.maximum(1.0, np.abs(p.ymin)); rh = BOUND_RELAX * np.maximum(1.0, np.abs(p.ymax))
        if np.any(y0 < p.ymin - rl) or np.any(y0 > p.ymax + rh):
            return dict(u0=None, x1=None, w=None, status=STATUS_INFEASIBLE, iters=0, res=None)
    r = qp_ipm_dense(H, g, E, e, G, lo, hi, tol=tol)
    w = r["w"]
    out = dict(u0=w[n:n + m].copy(), x1=w[n + m:2 * n + m].copy(), w=w, status=r["status"],
               iters=r["iters"], res=r["res"], nu=r["nu"], z_lo=r["z_lo"], z_hi=r["z_hi"])
    if getattr(p, "slacks", False):
        out["sl"] = w[-2 * p.ny:].copy()      # sl_k = w_opt[nw-ns:nw], MPC_code.py:800
    return out


def target_solve(p, usp, ysp, xsp, dhat, us_prev, tol=1e-11, px0=None, py0=None):
    """One ``solver_ss(...)`` call of MPC_code.py:704-709 + read-out ``:715-718``."""
    H, g, E, e, G, lo, hi = target_qp(p, usp, ysp, xsp, dhat, us_prev, px0, py0)
    r = qp_ipm_dense(H, g, E, e, G, lo, hi, tol=tol)
    n, m = p.nx, p.nu
    w = r["w"]
    return dict(xs=w[:n].copy(), us=w[n:n + m].copy(), ys=w[n + m:].copy(), status=r["status"],
                iters=r["iters"], res=r["res"])


def kalman(p, xi, Pm, y, yhat):
    """Time-varying Kalman filter on xi=[x;d] (Estimator.py:263-311).

    K = P C'(C P C' + R)^-1 (:297), P_corr = (I-KC)P (:300), xi+ = xi + K(y-yhat) (:303-306),
    P_plus = A P_corr A' + Q (:309).  The state *prediction* is not done here.
    """
    Aa, Ca = p.aug_estimator_matrices()
    S = Ca @ Pm @ Ca.T + p.R_kf
    K = np.linalg.solve(S.T, (Pm @ Ca.T).T).T
    P_corr = (np.eye(Aa.shape[0]) - K @ Ca) @ Pm
    xi_c = xi + K @ (y - yhat)
    P_plus = Aa @ P_corr @ Aa.T + p.Q_kf
    return xi_c, P_plus


def kalss(p, xi, y, yhat):
    """Fixed-gain observer xi+ = xi + K (y - yhat) (Estimator.py:231-261; also the ``lue`` case)."""
    return xi + p.K @ (y - yhat)


# ----------------------------------------------------------------------------------------
# the closed loop
# ----------------------------------------------------------------------------------------
def closed_loop(p, nsteps, x0_p=None, x0_m=None, u0=None, dhat0=None, P0=None, sched=None,
                tol=1e-11, ocp=ocp_solve, target=target_solve):
    """One instance of the loop MPC_code.py:485-827 (single instance, like the reference).

    Step order: store x, xhat -> yhat = Fy_model(xhat) (:524) -> y = Fy_p(x)
    (:534) -> estimator (:577-650) -> target (:693-718; hold previous on infeasible) ->
    ys (:730) -> OCP (:734-800; on infeasible hold u and propagate the model, :804-805) ->
    plant (:816).  Returns the log arrays under the reference's names (:877-895).
    """
    x = np.array(p.x0_p if x0_p is None else x0_p, dtype=float)
    xhat = np.array(p.x0_m if x0_m is None else x0_m, dtype=float)
    u = np.array(p.u0 if u0 is None else u0, dtype=float)
    dhat = np.array(p.dhat0 if dhat0 is None else dhat0, dtype=float)
    Pk = None if p.estimator != "kal" else np.array(p.P0 if P0 is None else P0, dtype=float)
    sched = p.schedules(nsteps) if sched is None else sched
    x0_m_fixed = np.array(p.x0_m if x0_m is None else x0_m, dtype=float)
    us_k, xs_k = u.copy(), x0_m_fixed.copy()                    # MPC_code.py:682-684
    log = {k: [] for k in ("Xp", "X_HAT", "Yp", "Y_HAT", "D_HAT", "XS", "US", "YS", "U",
                           "STATUS_SS", "STATUS_DYN", "KKT_DYN", "KKT_SS", "ITERS_DYN", "XHAT_C", "U_PREV",
                           "EXACT_DYN", "EXACT_SS")}
    n = p.nx
    has_par = getattr(p, "def_px", None) is not None or getattr(p, "def_py", None) is not None
    px0 = py0 = None; kw_t, kw_o = {}, {}
    for k in range(nsteps):
        if has_par:        # MPC_code.py:492-510: parameters over the horizon; p_x_k, p_y_k also reach the plant (p_xmp, p_ymp)
            pxh, pyh = p.horizon_params(k * p.h)
            px0, py0 = pxh[0], pyh[0]
            kw_t, kw_o = dict(px0=px0, py0=py0), dict(px=pxh, py=pyh)
        log["Xp"].append(x.copy()); log["X_HAT"].append(xhat.copy())
        yhat = model_fy(p, xhat, dhat, py0)
        y = plant_fy(p, x, sched["pyp"][k] + (py0 if py0 is not None else 0.0))
        log["Yp"].append(y.copy()); log["Y_HAT"].append(yhat.copy())
        xi = np.concatenate([xhat, dhat])
        if p.estimator == "kal":
            xi, Pk = kalman(p, xi, Pk, y, yhat)
        elif p.estimator == "kalss":
            xi = kalss(p, xi, y, yhat)
        xhat, dhat = xi[:n].copy(), xi[n:].copy()
        if p.dmin is not None:                                   # MPC_code.py:660-665
            dhat = np.minimum(np.maximum(dhat, p.dmin), p.dmax)
        log["D_HAT"].append(dhat.copy()); log["XHAT_C"].append(xhat.copy()); log["U_PREV"].append(u.copy())
        us_prev = us_k
        t = target(p, sched["usp"][k], sched["ysp"][k], sched["xsp"][k], dhat, us_prev, tol=tol, **kw_t)
        if t["status"] != STATUS_INFEASIBLE:
            xs_k, us_k = t["xs"], t["us"]
        log["XS"].append(xs_k.copy()); log["US"].append(us_k.copy())
        log["YS"].append(model_fy(p, xs_k, dhat, py0))
        log["STATUS_SS"].append(t["status"]); log["KKT_SS"].append(kkt_max(t["res"]))
        log["EXACT_SS"].append(bool(t.get("exact", False)))
        o = ocp(p, xhat, xs_k, us_k, dhat, u, tol=tol, **kw_o)
        if o["status"] != STATUS_INFEASIBLE:
            u, xhat = o["u0"].copy(), o["x1"].copy()             # MPC_code.py:798-799
        else:
            xhat = model_fx(p, xhat, u, dhat, px0)               # MPC_code.py:804-805
        log["U"].append(u.copy()); log["STATUS_DYN"].append(o["status"])
        log["KKT_DYN"].append(kkt_max(o["res"]) if o["res"] else np.nan)
        log["ITERS_DYN"].append(o["iters"]); log["EXACT_DYN"].append(bool(o.get("exact", False)))
        x = plant_fx(p, x, u, sched["pxp"][k] + (px0 if px0 is not None else 0.0), k * p.h)                 # MPC_code.py:816
    return {k: np.array(v) for k, v in log.items()}


def lqr_gain(p):
    """K with u0* = us + K (xhat - xs) when no bound is active."""
    return -np.linalg.solve(p.R + p.B.T @ p.P @ p.B, p.B.T @ p.P @ p.A)


# ----------------------------------------------------------------------------------------
# exact optimum by active-set polish (ground truth for fixtures)
# ----------------------------------------------------------------------------------------
def qp_polish(H, g, E, e, G, lo, hi, w, z_lo, z_hi, act_tol=None):
    """Turn an interior-point answer into the exact optimum.

    Guess the active set from the IPM multipliers/slacks, solve the equality-constrained QP
    on it with one dense symmetric solve, and *verify* the result: primal feasibility of the
    inactive rows and non-negativity of the active multipliers.  If the verification holds the
    returned point satisfies the KKT conditions to rounding (complementarity exactly), so it is
    the unique optimum irrespective of any solver tolerance.  Returns None if it does not hold
    (degenerate or wrongly guessed set) - callers then keep the IPM point.
    """
    Gw = G @ w
    s_lo, s_hi = Gw - lo, hi - Gw
    with np.errstate(invalid="ignore"):
        a_lo = np.isfinite(lo) & (z_lo > s_lo)
        a_hi = np.isfinite(hi) & (z_hi > s_hi)
    rows = np.vstack([E, G[a_lo], G[a_hi]])
    rhs = np.concatenate([e, lo[a_lo], hi[a_hi]])
    nv, nr = H.shape[0], rows.shape[0]
    KK = np.block([[H, rows.T], [rows, np.zeros((nr, nr))]])
    # duplicated active rows (e.g. an x bound and the identical y bound when C = I) make KK
    # singular but consistent: take the minimum-norm multipliers
    sol = np.linalg.lstsq(KK, np.concatenate([-g, rhs]), rcond=1e-13)[0]
    w2, mult = sol[:nv], sol[nv:]
    ne = E.shape[0]
    nu = mult[:ne]
    zl = np.zeros(len(lo)); zh = np.zeros(len(hi))
    zl[a_lo] = -mult[ne:ne + a_lo.sum()]          # rows G w = lo carry multiplier -z_lo
    zh[a_hi] = mult[ne + a_lo.sum():]
    res = kkt_residual(H, g, E, e, G, lo, hi, w2, nu, zl, zh)
    if kkt_max(res) > 1e-9:
        return None
    return dict(w=w2, nu=nu, z_lo=zl, z_hi=zh, res=res, n_active=int(a_lo.sum() + a_hi.sum()))


def ocp_solve_exact(p, xhat, xs, us, dhat, u_prev, tol=1e-11, px=None, py=None):
    """:func:`ocp_solve` followed by :func:`qp_polish`; ``exact`` tells whether the polish verified."""
    r = ocp_solve(p, xhat, xs, us, dhat, u_prev, tol=tol, px=px, py=py)
    r["exact"] = False
    if r["status"] != STATUS_SOLVED:
        return r
    H, g, E, e, G, lo, hi = ocp_qp(p, xhat, xs, us, dhat, u_prev, drop_stage0_rows=True, px=px, py=py)
    pol = qp_polish(H, g, E, e, G, lo, hi, r["w"], r["z_lo"], r["z_hi"])
    if pol is not None:
        n, m = p.nx, p.nu
        w = pol["w"]
        r.update(w=w, u0=w[n:n + m].copy(), x1=w[n + m:2 * n + m].copy(), res=pol["res"], exact=True,
                 n_active=pol["n_active"], nu=pol["nu"], z_lo=pol["z_lo"], z_hi=pol["z_hi"])
    return r


def target_solve_exact(p, usp, ysp, xsp, dhat, us_prev, tol=1e-11, px0=None, py0=None):
    H, g, E, e, G, lo, hi = target_qp(p, usp, ysp, xsp, dhat, us_prev, px0, py0)
    r = qp_ipm_dense(H, g, E, e, G, lo, hi, tol=tol)
    n, m = p.nx, p.nu
    exact = False
    if r["status"] == STATUS_SOLVED:
        pol = qp_polish(H, g, E, e, G, lo, hi, r["w"], r["z_lo"], r["z_hi"])
        if pol is not None:
            r.update(w=pol["w"], res=pol["res"]); exact = True
    w = r["w"]
    return dict(xs=w[:n].copy(), us=w[n:n + m].copy(), ys=w[n + m:].copy(), status=r["status"],
                iters=r["iters"], res=r["res"], exact=exact)
