"""ORACLE (test infrastructure, never shipped): NumPy, batch-vectorised statement of the
Riccati-based primal-dual interior-point iteration ("RPDIP") that the C restatement
(``oracle/mpc_oracle.c``) and the HIP kernels implement.

It exists to (i) fix the algorithm's specification in ~200 readable lines, (ii) cross-check the
C restatement on moderate batches, and (iii) be checked itself against the dense certifier
``mpc_oracle.qp_ipm_dense`` (different factorisation).  Parity status: see ``mpc_oracle.py``.

Stage form of the OCP of ``opt_dyn`` (Control_Calc.py:20-260), after eliminating the fixed
``x_0`` (MPC_code.py:734):

    z_{k+1} = A z_k + B u_k + c,                 k = 0..N-1,  z_0 given
    cost    = sum_k 1/2 (z_k-zr)'Q(z_k-zr) + (z_k-zr)'M(u_k-ur) + 1/2 (u_k-ur)'R(u_k-ur)
              + 1/2 (z_N-zr)'Pf(z_N-zr)
    bounds  ulo <= u_k <= uhi (k<N),  zlo_k <= z_k <= zhi_k (k=1..N)

For the R-form (``Ex_LMPC_CSTR``): z=x, zr=xs, ur=us, M=0.  For the Delta-u form
(``Ex_LMPC_WB``, Control_Calc.py:163-166,180-181): z=[x;u_prev], zr=[xs;0], ur=0,
Q=blkdiag(Q,S), M=[0;-S], R=S, A=[[A,0],[0,0]], B=[[B],[I]].
"""
from __future__ import annotations

import numpy as np

# ---- algorithm constants (shared verbatim with mpc_oracle.c and csrc/rpdip.hpp) -------------
MU0 = 1.0          # initial complementarity product
S_MIN = 1.0        # minimum initial slack ("push"), absolute
TAU = 0.995        # fraction to the boundary
TOL_STAT = 1e-9    # |grad_u L|_inf, relative to max(1,|grad_u L|_inf at the start)
TOL_STAT_ACC = 1e-6  # accepted after STALL_MAX stalled iterations (rounding floor eps*l/s*|step|)
STALL_MAX = 2
TOL_FEAS = 1e-9    # bound residual |v + s - hi|_inf
TOL_C = 1e-9       # complementarity, per bound: min(s, l) <= TOL_C ...
TOL_MU = 1e-14     # ... or s*l <= TOL_MU (degenerate bounds, s* = l* = 0, converge only like sqrt(mu))
MU_FLOOR = 1e-15   # the centring target sigma*mu is never below this ...
S_FLOOR = 1e-11    # ... nor below l*S_FLOOR: no slack is driven under S_FLOOR (keeps l/s bounded)
BOUND_RELAX = 1e-8 # relaxation of the stage-0 output rows (constraints on a given quantity)
WS_DELTA = 0.3     # closed-loop warm start: used when (xhat - prediction, dhat, xs, us) moved less than this
WS_KAPPA = 1e-2    # closed-loop warm start: minimum slack = clip(WS_KAPPA * movement, WS_SMIN_LO, WS_SMIN_HI) ...
WS_SMIN_LO = 1e-9
WS_SMIN_HI = 1e-6
WS_MU_FACTOR = 1e4 # ... and minimum complementarity product = WS_MU_FACTOR * (minimum slack)^2   (1e-14 .. 1e-8)
POLISH_AT = (1, 5, 9, 13)  # interior-point iteration counts after which an active-set polish is attempted
POLISH_W = 1e8     # augmented-Lagrangian weight on the active bounds of the polish
POLISH_TOL = 1e-9  # polish accepted if bound violation, negative multipliers and the last correction are below this
INFEAS_Z = 1e10    # dual blow-up threshold (times max(1,|g|)): infeasible problem
STATUS_SOLVED, STATUS_MAXITER, STATUS_INFEASIBLE = 0, 1, 2


def stage_data(p):
    """Stage matrices of a LinearMPCProblem-like object (duck-typed, see module docstring)."""
    n, m = p.nx, p.nu
    if not p.DUForm:
        A, B, Q, R, Pf = p.A, p.B, p.Q, p.R, p.P
        M = np.zeros((n, m))
        zlo_m, zhi_m = p.xmin.copy(), p.xmax.copy()
        zlo_e, zhi_e = p.xmin.copy(), p.xmax.copy()
    else:
        A = np.zeros((n + m, n + m)); A[:n, :n] = p.A
        B = np.vstack([p.B, np.eye(m)])
        Q = np.zeros((n + m, n + m)); Q[:n, :n] = p.Q; Q[n:, n:] = p.R
        M = np.vstack([np.zeros((n, m)), -p.R])
        R = p.R
        Pf = np.zeros((n + m, n + m)); Pf[:n, :n] = p.P
        inf = np.full(m, np.inf)
        zlo_m, zhi_m = np.concatenate([p.xmin, -inf]), np.concatenate([p.xmax, inf])
        zlo_e, zhi_e = zlo_m.copy(), zhi_m.copy()
    # Output rows C_i x with several (or no) non-zero entries: one extra stage state w_i = C_i x each, carried by
    # w+ = C_i (A x + B u + c), so that the row becomes a box on a state (k = 1..N-1; the terminal state has no output row,
    # Control_Calc.py:150-151,229-230).  No cost on w.
    yg = general_output_rows(p)
    if len(yg):
        na, ng = A.shape[0], len(yg)
        Cg = p.C[yg]
        A2 = np.zeros((na + ng, na + ng)); A2[:na, :na] = A; A2[na:, :n] = Cg @ p.A
        B = np.vstack([B, Cg @ p.B])
        Q2 = np.zeros((na + ng, na + ng)); Q2[:na, :na] = Q
        Pf2 = np.zeros((na + ng, na + ng)); Pf2[:na, :na] = Pf
        M = np.vstack([M, np.zeros((ng, m))])
        inf = np.full(ng, np.inf)
        zlo_m, zhi_m = np.concatenate([zlo_m, -inf]), np.concatenate([zhi_m, inf])
        zlo_e, zhi_e = np.concatenate([zlo_e, -inf]), np.concatenate([zhi_e, inf])
        A, Q, Pf = A2, Q2, Pf2
    return dict(A=A, B=B, Q=Q, M=M, R=R, Pf=Pf, ulo=p.umin.copy(), uhi=p.umax.copy(),
                zlo_m=zlo_m, zhi_m=zhi_m, zlo_e=zlo_e, zhi_e=zhi_e, n=A.shape[0], m=m, N=p.N, yg=yg)


def general_output_rows(p):
    """Indices of the bounded output rows that are not a multiple of one state (those map onto that state's box)."""
    if not p.y_bounded:
        return np.zeros(0, dtype=int)
    return np.array([i for i in range(p.ny) if np.count_nonzero(p.C[i]) != 1
                     and (np.isfinite(p.ymin[i]) or np.isfinite(p.ymax[i]))], dtype=int)


def _ymap(p):
    idx = np.full(p.ny, -1, dtype=int); scale = np.zeros(p.ny)
    for i in range(p.ny):
        nz = np.nonzero(p.C[i])[0]
        if len(nz) == 1:
            idx[i], scale[i] = nz[0], p.C[i, nz[0]]
    return idx, scale      # idx < 0: a general row (own stage state) or an unbounded one


def instance_data(p, sd, xhat, xs, us, dhat, u_prev):
    """Per-instance vectors z0, zr, ur, c, state boxes and the stage-0 output check (all [B,.])."""
    xhat, xs, us, dhat, u_prev = (np.atleast_2d(np.asarray(a, float)) for a in (xhat, xs, us, dhat, u_prev))
    Bsz, n, m = xhat.shape[0], p.nx, p.nu
    c = np.broadcast_to(p.fx_const, (Bsz, n)) + (dhat @ p.Bd.T if p.nd else 0.0)
    if not p.DUForm:
        z0, zr, ur = xhat.copy(), xs.copy(), us.copy()
    else:
        z0 = np.hstack([xhat, u_prev]); zr = np.hstack([xs, np.zeros((Bsz, m))]); ur = np.zeros((Bsz, m))
        c = np.hstack([c, np.zeros((Bsz, m))])
    yg = sd.get("yg", ())
    if len(yg):
        Cg = p.C[yg]
        z0 = np.hstack([z0, xhat @ Cg.T]); zr = np.hstack([zr, xs @ Cg.T]); c = np.hstack([c, c[:, :n] @ Cg.T])
    na = sd["n"]
    zlo_m = np.broadcast_to(sd["zlo_m"], (Bsz, na)).copy(); zhi_m = np.broadcast_to(sd["zhi_m"], (Bsz, na)).copy()
    ok0 = np.ones(Bsz, dtype=bool)
    if p.y_bounded:
        idx, scale = _ymap(p)
        e = np.broadcast_to(p.fy_const, (Bsz, p.ny)) + (dhat @ p.Cd.T if p.nd else 0.0)
        y0 = xhat @ p.C.T + e
        # stage-0 rows constrain a given quantity: a pure feasibility test, with the relaxation
        # IPOPT applies to every bound before it starts (bound_relax_factor = 1e-8, [ext])
        rl = BOUND_RELAX * np.maximum(1.0, np.abs(p.ymin)); rh = BOUND_RELAX * np.maximum(1.0, np.abs(p.ymax))
        ok0 = np.all((y0 >= p.ymin - rl) & (y0 <= p.ymax + rh), axis=1)
        for i in range(p.ny):
            if idx[i] < 0:
                continue
            a, b = (p.ymin[i] - e[:, i]) / scale[i], (p.ymax[i] - e[:, i]) / scale[i]
            lo_i, hi_i = (a, b) if scale[i] > 0 else (b, a)
            zlo_m[:, idx[i]] = np.maximum(zlo_m[:, idx[i]], lo_i)
            zhi_m[:, idx[i]] = np.minimum(zhi_m[:, idx[i]], hi_i)
        for g, i in enumerate(yg):
            zlo_m[:, na - len(yg) + g] = p.ymin[i] - e[:, i]; zhi_m[:, na - len(yg) + g] = p.ymax[i] - e[:, i]
    zlo_e = np.broadcast_to(sd["zlo_e"], (Bsz, na)).copy(); zhi_e = np.broadcast_to(sd["zhi_e"], (Bsz, na)).copy()
    return dict(z0=z0, zr=zr, ur=ur, c=c, zlo_m=zlo_m, zhi_m=zhi_m, zlo_e=zlo_e, zhi_e=zhi_e, ok0=ok0,
                us=us.copy())


def rpdip_solve(sd, inst, max_iter=100, verbose=False, trace=None, polish=False, warm=None):
    """Batched Mehrotra predictor-corrector with Riccati KKT solves.  Returns dict of [B,..] arrays."""
    A, Bm, Q, M, R, Pf = sd["A"], sd["B"], sd["Q"], sd["M"], sd["R"], sd["Pf"]
    n, m, N = sd["n"], sd["m"], sd["N"]
    z0, zr, ur, c = inst["z0"], inst["zr"], inst["ur"], inst["c"]
    Bsz = z0.shape[0]
    # bounds per block k (u_k , z_{k+1}) : v = [u; z]  -> [B,N,m+n]
    nv = m + n
    lo = np.empty((Bsz, N, nv)); hi = np.empty((Bsz, N, nv))
    lo[:, :, :m] = sd["ulo"]; hi[:, :, :m] = sd["uhi"]
    lo[:, :N - 1, m:] = inst["zlo_m"][:, None, :]; hi[:, :N - 1, m:] = inst["zhi_m"][:, None, :]
    lo[:, N - 1, m:] = inst["zlo_e"]; hi[:, N - 1, m:] = inst["zhi_e"]
    fl, fh = np.isfinite(lo), np.isfinite(hi)
    ncon = (fl.sum(axis=(1, 2)) + fh.sum(axis=(1, 2))).astype(float)
    lo_f = np.where(fl, lo, 0.0); hi_f = np.where(fh, hi, 0.0)

    def simulate(u):
        z = np.empty((Bsz, N + 1, n)); z[:, 0] = z0
        for k in range(N):
            z[:, k + 1] = z[:, k] @ A.T + u[:, k] @ Bm.T + c
        return z

    # ---- initial point ------------------------------------------------------------------
    us0 = inst["us"]
    u = np.broadcast_to(us0[:, None, :], (Bsz, N, m)).copy()
    ulo, uhi = sd["ulo"], sd["uhi"]
    both = np.isfinite(ulo) & np.isfinite(uhi)
    push = np.where(both, 0.1 * (np.where(both, uhi, 0) - np.where(both, ulo, 0)), 0.1 * np.maximum(1.0, np.abs(np.where(np.isfinite(ulo), ulo, np.where(np.isfinite(uhi), uhi, 0)))))
    u = np.minimum(np.maximum(u, np.where(np.isfinite(ulo), ulo + push, -np.inf)), np.where(np.isfinite(uhi), uhi - push, np.inf))
    z = simulate(u)
    v = np.concatenate([u, z[:, 1:]], axis=2)
    s_lo = np.where(fl, np.maximum(v - lo_f, S_MIN), 1.0); s_hi = np.where(fh, np.maximum(hi_f - v, S_MIN), 1.0)
    l_lo = np.where(fl, MU0 / s_lo, 0.0); l_hi = np.where(fh, MU0 / s_hi, 0.0)
    if warm is not None:     # primal-dual warm start from the previous closed-loop step
        use = warm["use"]
        uw = np.minimum(np.maximum(warm["u"], np.where(np.isfinite(ulo), ulo, -np.inf)), np.where(np.isfinite(uhi), uhi, np.inf))
        zw = simulate(uw)
        vw = np.concatenate([uw, zw[:, 1:]], axis=2)
        # floors scale with how far the problem data moved: an unchanged problem resumes from its solution
        smin_w = np.clip(WS_KAPPA * warm["delta"], WS_SMIN_LO, WS_SMIN_HI)[:, None, None]
        mu_w = WS_MU_FACTOR * smin_w * smin_w
        sw_lo = np.where(fl, np.maximum(vw - lo_f, smin_w), 1.0); sw_hi = np.where(fh, np.maximum(hi_f - vw, smin_w), 1.0)
        lw_lo = np.where(fl, np.maximum(warm["l_lo"], mu_w / sw_lo), 0.0); lw_hi = np.where(fh, np.maximum(warm["l_hi"], mu_w / sw_hi), 0.0)
        m3 = use[:, None, None]
        u = np.where(m3, uw, u); z = np.where(m3, zw, z)
        s_lo = np.where(m3, sw_lo, s_lo); s_hi = np.where(m3, sw_hi, s_hi); l_lo = np.where(m3, lw_lo, l_lo); l_hi = np.where(m3, lw_hi, l_hi)

    status = np.full(Bsz, -1, dtype=np.int32)
    iters = np.zeros(Bsz, dtype=np.int32)
    active = inst["ok0"].copy()
    gscale = None
    stall = np.zeros(Bsz, dtype=np.int64)
    polished = np.zeros(Bsz, dtype=bool)
    res_out = np.zeros((Bsz, 3))
    K = np.empty((Bsz, N, m, n)); Linv = np.empty((Bsz, N, m, m)); Acl = np.empty((Bsz, N, n, n))

    for it in range(max_iter + 1):
        v = np.concatenate([u, z[:, 1:]], axis=2)
        r_lo = np.where(fl, v - s_lo - lo_f, 0.0); r_hi = np.where(fh, v + s_hi - hi_f, 0.0)
        mu = ((s_lo * l_lo).sum(axis=(1, 2)) + (s_hi * l_hi).sum(axis=(1, 2))) / np.maximum(ncon, 1.0)
        sig = l_lo / s_lo + l_hi / s_hi                         # barrier Hessian diag [B,N,nv]
        dl = l_hi - l_lo                                        # net bound multiplier
        # gradient pieces of the *current* point (cost + bound multipliers), no costates
        dz = z - zr[:, None, :]; du = u - ur[:, None, :]
        gz = np.empty((Bsz, N + 1, n)); gu = np.empty((Bsz, N, m))
        gz[:, :N] = dz[:, :N] @ Q.T + du @ M.T
        gz[:, N] = dz[:, N] @ Pf.T
        gu[:] = du @ R.T + dz[:, :N] @ M
        gz[:, 1:] += dl[:, :, m:]; gu += dl[:, :, :m]
        # adjoint recursion pi_k = gz_k + A' pi_{k+1}; stationarity residual r_u = gu_k + B' pi_{k+1}
        pi = gz[:, N].copy(); r_u = np.empty((Bsz, N, m))
        for k in range(N - 1, -1, -1):
            r_u[:, k] = gu[:, k] + pi @ Bm
            pi = gz[:, k] + pi @ A
        if gscale is None:
            gscale = np.maximum(1.0, np.abs(r_u).max(axis=(1, 2)))
        res_s = np.abs(r_u).max(axis=(1, 2)); res_p = np.maximum(np.abs(r_lo).max(axis=(1, 2)), np.abs(r_hi).max(axis=(1, 2)))
        res_out[active] = np.stack([res_s, res_p, mu], axis=1)[active]
        cres = np.maximum(_comp(s_lo, l_lo).max(axis=(1, 2)), _comp(s_hi, l_hi).max(axis=(1, 2)))
        ok_cp = (cres <= 1.0) & (res_p <= TOL_FEAS)
        stall = np.where(ok_cp, stall + 1, 0)
        conv = active & ok_cp & ((res_s <= TOL_STAT * gscale) | ((stall > STALL_MAX) & (res_s <= TOL_STAT_ACC * gscale)))
        status[conv] = STATUS_SOLVED; iters[conv] = it; active &= ~conv
        lmax = np.maximum(l_lo.max(axis=(1, 2)), l_hi.max(axis=(1, 2)))
        bad = active & ((lmax > INFEAS_Z * gscale) | ~np.isfinite(mu))
        status[bad] = STATUS_INFEASIBLE; iters[bad] = it; active &= ~bad
        if polish and it in POLISH_AT and active.any():
            pu, pz, pok, pres = _polish(sd, inst, lo_f, hi_f, fl, fh, u, z, s_lo, s_hi, l_lo, l_hi)
            acc = active & pok
            u = np.where(acc[:, None, None], pu, u); z = np.where(acc[:, None, None], pz, z)
            status[acc] = STATUS_SOLVED; iters[acc] = it; polished[acc] = True; active &= ~acc
            res_out[acc] = pres[acc]
        if verbose:
            print(it, "active", active.sum(), "res", res_s.max(), res_p.max(), mu.max())
        if trace is not None:
            smin = np.minimum(np.where(fl, s_lo, np.inf).min(axis=(1, 2)), np.where(fh, s_hi, np.inf).min(axis=(1, 2)))
            trace.append(dict(it=it, res_s=res_s.copy(), res_p=res_p.copy(), mu=mu.copy(), smin=smin, lmax=lmax.copy(),
                              sigmax=sig.max(axis=(1, 2)), active=active.copy()))
        if not active.any() or it == max_iter:
            break
        # ---- factorisation (depends on sig only) -------------------------------------------
        Pn = np.broadcast_to(Pf, (Bsz, n, n)) + _diag(sig[:, N - 1, m:])
        for k in range(N - 1, -1, -1):
            PB = Pn @ Bm                              # [B,n,m]
            Lam = R + _diag(sig[:, k, :m]) + Bm.T @ PB
            Psi = M.T + np.swapaxes(PB, 1, 2) @ A     # [B,m,n]
            Li = np.linalg.inv(Lam)
            Linv[:, k] = Li
            K[:, k] = -Li @ Psi
            Acl[:, k] = A + Bm @ K[:, k]              # closed-loop matrix, reused by the rhs sweeps
            if k > 0:
                # Joseph (closed-loop Lyapunov) form: no cancellation of the O(sig) terms
                Kk = K[:, k]; Rt = R + _diag(sig[:, k, :m])
                MK = M @ Kk
                Pn = (Q + _diag(sig[:, k - 1, m:]) + np.swapaxes(Acl[:, k], 1, 2) @ Pn @ Acl[:, k]
                      + np.swapaxes(Kk, 1, 2) @ Rt @ Kk + MK + np.swapaxes(MK, 1, 2))
                Pn = 0.5 * (Pn + np.swapaxes(Pn, 1, 2))

        def solve(rc_lo, rc_hi):
            """Newton step for complementarity targets s*l + ds*dl = -rc (rc given as residual)."""
            h = (-rc_hi + l_hi * r_hi) / s_hi + (rc_lo + l_lo * r_lo) / s_lo
            qz = gz.copy(); qu = gu.copy()
            qz[:, 1:] += h[:, :, m:]; qu += h[:, :, :m]
            kff = np.empty((Bsz, N, m))
            pv = qz[:, N].copy()
            for k in range(N - 1, -1, -1):
                psi = qu[:, k] + pv @ Bm
                kff[:, k] = -np.einsum("bij,bj->bi", Linv[:, k], psi)
                pv = qz[:, k] + np.einsum("bji,bj->bi", Acl[:, k], pv) + np.einsum("bji,bj->bi", K[:, k], qu[:, k])
            d_z = np.zeros((Bsz, N + 1, n)); d_u = np.empty((Bsz, N, m))
            for k in range(N):
                d_u[:, k] = np.einsum("bij,bj->bi", K[:, k], d_z[:, k]) + kff[:, k]
                d_z[:, k + 1] = d_z[:, k] @ A.T + d_u[:, k] @ Bm.T
            dv = np.concatenate([d_u, d_z[:, 1:]], axis=2)
            ds_hi = np.where(fh, -r_hi - dv, 0.0); ds_lo = np.where(fl, r_lo + dv, 0.0)
            dl_hi = np.where(fh, (-rc_hi - l_hi * ds_hi) / s_hi, 0.0)
            dl_lo = np.where(fl, (-rc_lo - l_lo * ds_lo) / s_lo, 0.0)
            return d_u, d_z, ds_lo, ds_hi, dl_lo, dl_hi

        def maxstep(xs_, dxs_, cap=1.0):
            out = np.full(Bsz, cap)
            for x_, d_ in zip(xs_, dxs_):
                with np.errstate(divide="ignore", invalid="ignore"):
                    r = np.where(d_ < 0, -x_ / d_, np.inf)
                out = np.minimum(out, r.min(axis=(1, 2)))
            return out

        d_u, d_z, ds_lo, ds_hi, dl_lo, dl_hi = solve(np.where(fl, s_lo * l_lo, 0.0), np.where(fh, s_hi * l_hi, 0.0))
        a_aff = maxstep((s_lo, s_hi, l_lo, l_hi), (ds_lo, ds_hi, dl_lo, dl_hi))
        aa = a_aff[:, None, None]
        mu_aff = (((s_lo + aa * ds_lo) * (l_lo + aa * dl_lo)).sum(axis=(1, 2)) + ((s_hi + aa * ds_hi) * (l_hi + aa * dl_hi)).sum(axis=(1, 2))) / np.maximum(ncon, 1.0)
        sigma = np.where(mu > 0, (mu_aff / np.where(mu > 0, mu, 1.0)) ** 3, 0.0)
        sm = np.maximum(sigma * mu, MU_FLOOR)[:, None, None]
        d_u, d_z, ds_lo, ds_hi, dl_lo, dl_hi = solve(
            np.where(fl, s_lo * l_lo - np.maximum(sm, l_lo * S_FLOOR) + ds_lo * dl_lo, 0.0),
            np.where(fh, s_hi * l_hi - np.maximum(sm, l_hi * S_FLOOR) + ds_hi * dl_hi, 0.0))
        # full step whenever the boundary is further than 1/TAU away (a Newton step solves an unconstrained QP exactly)
        a = np.minimum(1.0, TAU * maxstep((s_lo, s_hi, l_lo, l_hi), (ds_lo, ds_hi, dl_lo, dl_hi), cap=np.inf))
        a = np.where(active, a, 0.0)[:, None, None]
        u = u + a * d_u; z = z + a * d_z
        s_lo = s_lo + a * ds_lo; s_hi = s_hi + a * ds_hi; l_lo = l_lo + a * dl_lo; l_hi = l_hi + a * dl_hi
    left = status < 0
    status[left] = STATUS_MAXITER; iters[left] = max_iter
    status[~inst["ok0"]] = STATUS_INFEASIBLE; iters[~inst["ok0"]] = 0
    return dict(u=u, z=z, u0=u[:, 0].copy(), z1=z[:, 1].copy(), status=status, iters=iters, res=res_out,
                l_lo=l_lo, l_hi=l_hi, s_lo=s_lo, s_hi=s_hi, polished=polished)


def _polish(sd, inst, lo_f, hi_f, fl, fh, u, z, s_lo, s_hi, l_lo, l_hi):
    """Active-set polish of an interior-point iterate - EXPERIMENT, off by default, not in the C / HIP code
    (it makes the easy majority exact after one interior-point iteration, but the regime that
    dominates the benchmark - a long arc riding a state bound with geometrically decaying multipliers - neither
    polishes nor survives primal-dual active-set updates, and at 4096 instances per GPU the slowest lane sets the pace).

    Guess the active set from the iterate (a bound is active when its multiplier exceeds its slack), solve the
    equality-constrained QP on it with an augmented-Lagrangian weight POLISH_W and the interior-point multipliers
    as first estimate - one Riccati factorisation, two solves with a multiplier update in between - and VERIFY the
    result: every bound satisfied, every active multiplier non-negative, second correction negligible.  A verified
    point satisfies the KKT conditions with zero complementarity: it is the exact optimum.  Returns
    (u, z, ok[B], res[B,3]); the caller keeps the interior-point iterate where ok is False.
    """
    A, Bm, Q, M, R, Pf = sd["A"], sd["B"], sd["Q"], sd["M"], sd["R"], sd["Pf"]
    n, m, N = sd["n"], sd["m"], sd["N"]
    zr, ur = inst["zr"], inst["ur"]
    Bsz = u.shape[0]
    a_lo = fl & (l_lo > s_lo); a_hi = fh & (l_hi > s_hi)
    lam_lo = np.where(a_lo, l_lo, 0.0); lam_hi = np.where(a_hi, l_hi, 0.0)
    wgt = POLISH_W * (a_lo.astype(float) + a_hi.astype(float))
    K = np.empty((Bsz, N, m, n)); Linv = np.empty((Bsz, N, m, m)); Acl = np.empty((Bsz, N, n, n))
    Pn = np.broadcast_to(Pf, (Bsz, n, n)) + _diag(wgt[:, N - 1, m:])
    for k in range(N - 1, -1, -1):
        PB = Pn @ Bm
        Lam = R + _diag(wgt[:, k, :m]) + Bm.T @ PB
        Psi = M.T + np.swapaxes(PB, 1, 2) @ A
        Li = np.linalg.inv(Lam); Linv[:, k] = Li; K[:, k] = -Li @ Psi; Acl[:, k] = A + Bm @ K[:, k]
        if k > 0:
            Kk = K[:, k]; Rt = R + _diag(wgt[:, k, :m]); MK = M @ Kk
            Pn = (Q + _diag(wgt[:, k - 1, m:]) + np.swapaxes(Acl[:, k], 1, 2) @ Pn @ Acl[:, k]
                  + np.swapaxes(Kk, 1, 2) @ Rt @ Kk + MK + np.swapaxes(MK, 1, 2))
            Pn = 0.5 * (Pn + np.swapaxes(Pn, 1, 2))
    u = u.copy(); z = z.copy()
    dlast = np.zeros(Bsz)
    for _ in range(2):
        v = np.concatenate([u, z[:, 1:]], axis=2)
        al = np.where(a_lo, -lam_lo + POLISH_W * (v - lo_f), 0.0) + np.where(a_hi, lam_hi + POLISH_W * (v - hi_f), 0.0)
        dz = z - zr[:, None, :]; du = u - ur[:, None, :]
        gz = np.empty((Bsz, N + 1, n)); gu = np.empty((Bsz, N, m))
        gz[:, :N] = dz[:, :N] @ Q.T + du @ M.T; gz[:, N] = dz[:, N] @ Pf.T
        gu[:] = du @ R.T + dz[:, :N] @ M
        gz[:, 1:] += al[:, :, m:]; gu += al[:, :, :m]
        kff = np.empty((Bsz, N, m)); pv = gz[:, N].copy()
        for k in range(N - 1, -1, -1):
            psi = gu[:, k] + pv @ Bm
            kff[:, k] = -np.einsum("bij,bj->bi", Linv[:, k], psi)
            pv = gz[:, k] + np.einsum("bji,bj->bi", Acl[:, k], pv) + np.einsum("bji,bj->bi", K[:, k], gu[:, k])
        d_z = np.zeros((Bsz, N + 1, n)); d_u = np.empty((Bsz, N, m))
        for k in range(N):
            d_u[:, k] = np.einsum("bij,bj->bi", K[:, k], d_z[:, k]) + kff[:, k]
            d_z[:, k + 1] = d_z[:, k] @ A.T + d_u[:, k] @ Bm.T
        u = u + d_u; z = z + d_z
        v = np.concatenate([u, z[:, 1:]], axis=2)
        lam_lo = np.where(a_lo, lam_lo - POLISH_W * (v - lo_f), 0.0); lam_hi = np.where(a_hi, lam_hi + POLISH_W * (v - hi_f), 0.0)
        dlast = np.maximum(np.abs(d_u).max(axis=(1, 2)), np.abs(d_z).max(axis=(1, 2)))
    viol = np.maximum(np.where(fl, lo_f - v, -np.inf), np.where(fh, v - hi_f, -np.inf)).max(axis=(1, 2))
    dneg = np.minimum(np.where(a_lo, lam_lo, np.inf), np.where(a_hi, lam_hi, np.inf)).min(axis=(1, 2))
    ok = (viol <= POLISH_TOL) & (dneg >= -POLISH_TOL) & (dlast <= POLISH_TOL) & np.isfinite(dlast)
    res = np.stack([dlast, np.maximum(viol, 0.0), np.zeros(Bsz)], axis=1)
    return u, z, ok, res


def _comp(s, l):
    """Per-bound complementarity measure, <= 1 means converged (see TOL_C / TOL_MU)."""
    return np.minimum(np.minimum(s, l) / TOL_C, s * l / TOL_MU)


def _diag(d):
    out = np.zeros(d.shape + (d.shape[-1],))
    i = np.arange(d.shape[-1])
    out[..., i, i] = d
    return out


# ==========================================================================================
# target problem (opt_ss, Target_Calc.py:20-161) in null-space coordinates
# ==========================================================================================
def target_data(p):
    """Constant data of the reduced target QP.

    ``[A-I, B] [xs;us] = -(Bd d + const)`` (Target_Calc.py:75-77) is eliminated with a QR of
    ``[A-I, B]'``: [xs;us] = Ep (-(Bd d + const)) + Z y, y in R^nr.  ys = C xs + Cd d + const
    (:80-81) is substituted.  What is left is a strictly convex QP in y with the box rows of
    xs, us, ys (:127-134) as W y + w0 in [lo, hi].
    """
    n, m, q = p.nx, p.nu, p.ny
    E = np.hstack([p.A - np.eye(n), p.B])
    Qf, Rf = np.linalg.qr(E.T, mode="complete")          # E' = Qf Rf
    if np.abs(np.diag(Rf[:n])).min() < 1e-12 * np.abs(Rf).max():
        raise ValueError("[A-I, B] is rank deficient: no steady state for arbitrary disturbances")
    Q1, Z = Qf[:, :n], Qf[:, n:]
    Ep = Q1 @ np.linalg.inv(Rf[:n].T)                     # E Ep = I
    Zx, Zu = Z[:n], Z[n:]
    CZx = p.C @ Zx
    Hr = CZx.T @ p.Qss @ CZx + Zu.T @ p.Rss @ Zu
    W = np.vstack([Zx, Zu, CZx])
    lo = np.concatenate([p.xmin_ss, p.umin_ss, p.ymin_ss]); hi = np.concatenate([p.xmax_ss, p.umax_ss, p.ymax_ss])
    return dict(Ep=Ep, Z=Z, Zx=Zx, Zu=Zu, CZx=CZx, Hr=0.5 * (Hr + Hr.T), W=W, lo=lo, hi=hi, nr=Z.shape[1])


def target_solve(p, td, usp, ysp, xsp, dhat, us_prev, max_iter=100, warm=None):
    """Batched Mehrotra predictor-corrector on the reduced target QP.  Returns xs, us, ys, status, iters, and `warm`:
    the data a later call may start from (closed loop only: y, multipliers and the QP vectors gr, w0 they belong to).
    A call given `warm` starts an instance from it when that solve succeeded and (gr, w0) moved by at most WS_DELTA,
    with the same slack / multiplier floors as the OCP warm start."""
    usp, ysp, dhat, us_prev = (np.atleast_2d(np.asarray(a, float)) for a in (usp, ysp, dhat, us_prev))
    Bsz, n, m, q, nr = dhat.shape[0], p.nx, p.nu, p.ny, td["nr"]
    usp = np.broadcast_to(usp, (Bsz, m)); ysp = np.broadcast_to(ysp, (Bsz, q))
    cx = np.broadcast_to(p.fx_const, (Bsz, n)) + (dhat @ p.Bd.T if p.nd else 0.0)
    e = np.broadcast_to(p.fy_const, (Bsz, q)) + (dhat @ p.Cd.T if p.nd else 0.0)
    vp = -cx @ td["Ep"].T                                       # particular [xs;us]
    yp = vp[:, :n] @ p.C.T + e
    uref = us_prev if p.DUssForm else usp
    gr = (yp - ysp) @ p.Qss.T @ td["CZx"] + (vp[:, n:] - uref) @ p.Rss.T @ td["Zu"]
    w0 = np.hstack([vp, yp]); W = td["W"]; Hr = td["Hr"]
    lo, hi = td["lo"], td["hi"]
    fl, fh = np.isfinite(lo), np.isfinite(hi)
    ncon = float(fl.sum() + fh.sum())
    lo_f, hi_f = np.where(fl, lo, 0.0), np.where(fh, hi, 0.0)
    y = -np.linalg.solve(Hr, gr.T).T
    v = w0 + y @ W.T
    s_lo = np.where(fl, np.maximum(v - lo_f, S_MIN), 1.0); s_hi = np.where(fh, np.maximum(hi_f - v, S_MIN), 1.0)
    l_lo = np.where(fl, MU0 / s_lo, 0.0); l_hi = np.where(fh, MU0 / s_hi, 0.0)
    if warm is not None:
        delta = np.maximum(np.abs(gr - warm["gr"]).max(axis=1), np.abs(w0 - warm["w0"]).max(axis=1))
        use = warm["valid"] & (delta <= WS_DELTA)
        smin = np.clip(WS_KAPPA * delta, WS_SMIN_LO, WS_SMIN_HI)[:, None]; wmu = WS_MU_FACTOR * smin * smin
        yw = warm["y"]; vw = w0 + yw @ W.T
        sw_lo = np.where(fl, np.maximum(vw - lo_f, smin), 1.0); sw_hi = np.where(fh, np.maximum(hi_f - vw, smin), 1.0)
        lw_lo = np.where(fl, np.maximum(warm["l_lo"], wmu / sw_lo), 0.0); lw_hi = np.where(fh, np.maximum(warm["l_hi"], wmu / sw_hi), 0.0)
        u_ = use[:, None]
        y = np.where(u_, yw, y); s_lo = np.where(u_, sw_lo, s_lo); s_hi = np.where(u_, sw_hi, s_hi)
        l_lo = np.where(u_, lw_lo, l_lo); l_hi = np.where(u_, lw_hi, l_hi)
    status = np.full(Bsz, -1, dtype=np.int32); iters = np.zeros(Bsz, dtype=np.int32); active = np.ones(Bsz, dtype=bool)
    gscale = None
    stall = np.zeros(Bsz, dtype=np.int64)
    for it in range(max_iter + 1):
        v = w0 + y @ W.T
        r_lo = np.where(fl, v - s_lo - lo_f, 0.0); r_hi = np.where(fh, v + s_hi - hi_f, 0.0)
        mu = ((s_lo * l_lo).sum(1) + (s_hi * l_hi).sum(1)) / max(ncon, 1.0)
        grad = y @ Hr.T + gr + (l_hi - l_lo) @ W
        if gscale is None:
            gscale = np.maximum(1.0, np.abs(gr).max(axis=1))
        res_s = np.abs(grad).max(axis=1); res_p = np.maximum(np.abs(r_lo).max(axis=1), np.abs(r_hi).max(axis=1))
        cres = np.maximum(_comp(s_lo, l_lo).max(axis=1), _comp(s_hi, l_hi).max(axis=1))
        ok_cp = (cres <= 1.0) & (res_p <= TOL_FEAS)
        stall = np.where(ok_cp, stall + 1, 0)
        conv = active & ok_cp & ((res_s <= TOL_STAT * gscale) | ((stall > STALL_MAX) & (res_s <= TOL_STAT_ACC * gscale)))
        status[conv] = STATUS_SOLVED; iters[conv] = it; active &= ~conv
        lmax = np.maximum(l_lo.max(axis=1), l_hi.max(axis=1))
        bad = active & ((lmax > INFEAS_Z * gscale) | ~np.isfinite(mu))
        status[bad] = STATUS_INFEASIBLE; iters[bad] = it; active &= ~bad
        if not active.any() or it == max_iter:
            break
        sig = l_lo / s_lo + l_hi / s_hi
        Ht = Hr + np.einsum("bi,ij,ik->bjk", sig, W, W)
        Hti = np.linalg.inv(Ht)

        def solve(rc_lo, rc_hi):
            h = (-rc_hi + l_hi * r_hi) / s_hi + (rc_lo + l_lo * r_lo) / s_lo
            dy = -np.einsum("bij,bj->bi", Hti, grad + h @ W)
            dv = dy @ W.T
            ds_hi = np.where(fh, -r_hi - dv, 0.0); ds_lo = np.where(fl, r_lo + dv, 0.0)
            dl_hi = np.where(fh, (-rc_hi - l_hi * ds_hi) / s_hi, 0.0); dl_lo = np.where(fl, (-rc_lo - l_lo * ds_lo) / s_lo, 0.0)
            return dy, ds_lo, ds_hi, dl_lo, dl_hi

        def maxstep(xs_, dxs_, cap=1.0):
            out = np.full(Bsz, cap)
            for x_, d_ in zip(xs_, dxs_):
                with np.errstate(divide="ignore", invalid="ignore"):
                    out = np.minimum(out, np.where(d_ < 0, -x_ / d_, np.inf).min(axis=1))
            return out

        dy, ds_lo, ds_hi, dl_lo, dl_hi = solve(np.where(fl, s_lo * l_lo, 0.0), np.where(fh, s_hi * l_hi, 0.0))
        aa = maxstep((s_lo, s_hi, l_lo, l_hi), (ds_lo, ds_hi, dl_lo, dl_hi))[:, None]
        mu_aff = (((s_lo + aa * ds_lo) * (l_lo + aa * dl_lo)).sum(1) + ((s_hi + aa * ds_hi) * (l_hi + aa * dl_hi)).sum(1)) / max(ncon, 1.0)
        sigma = np.where(mu > 0, (mu_aff / np.where(mu > 0, mu, 1.0)) ** 3, 0.0)
        sm = np.maximum(sigma * mu, MU_FLOOR)[:, None]
        dy, ds_lo, ds_hi, dl_lo, dl_hi = solve(np.where(fl, s_lo * l_lo - np.maximum(sm, l_lo * S_FLOOR) + ds_lo * dl_lo, 0.0),
                                               np.where(fh, s_hi * l_hi - np.maximum(sm, l_hi * S_FLOOR) + ds_hi * dl_hi, 0.0))
        a = np.minimum(1.0, TAU * maxstep((s_lo, s_hi, l_lo, l_hi), (ds_lo, ds_hi, dl_lo, dl_hi), cap=np.inf))
        a = np.where(active, a, 0.0)[:, None]
        y = y + a * dy; s_lo = s_lo + a * ds_lo; s_hi = s_hi + a * ds_hi; l_lo = l_lo + a * dl_lo; l_hi = l_hi + a * dl_hi
    left = status < 0
    status[left] = STATUS_MAXITER; iters[left] = max_iter
    vv = vp + y @ td["Z"].T
    xs, us = vv[:, :n], vv[:, n:]
    return dict(xs=xs, us=us, ys=xs @ p.C.T + e, status=status, iters=iters,
                warm=dict(y=y, l_lo=l_lo, l_hi=l_hi, gr=gr, w0=w0, valid=status == STATUS_SOLVED))


# ==========================================================================================
# estimator + closed loop over a batch
# ==========================================================================================
def kalman_batch(p, xi, Pm, y, yhat):
    """Estimator.py:263-311 for a batch: xi [B,n+nd], Pm [B,n+nd,n+nd]."""
    Aa, Ca = p.aug_estimator_matrices()
    S = Ca @ Pm @ Ca.T + p.R_kf
    PCt = Pm @ Ca.T
    K = np.linalg.solve(np.swapaxes(S, 1, 2), np.swapaxes(PCt, 1, 2))
    K = np.swapaxes(K, 1, 2)
    P_corr = (np.eye(Aa.shape[0]) - K @ Ca) @ Pm
    xi_c = xi + np.einsum("bij,bj->bi", K, y - yhat)
    return xi_c, Aa @ P_corr @ Aa.T + p.Q_kf


def closed_loop_batch(p, nsteps, x0_p, x0_m, sched=None, max_iter=100, warm_start=True, noise=None):
    """The loop of MPC_code.py:485-827 for B instances that share the problem and the schedules."""
    sd, td = stage_data(p), target_data(p)
    x0_p = np.atleast_2d(np.asarray(x0_p, float)); x0_m = np.atleast_2d(np.asarray(x0_m, float))
    Bsz, n, m = x0_p.shape[0], p.nx, p.nu
    sched = p.schedules(nsteps) if sched is None else sched
    x, xhat = x0_p.copy(), x0_m.copy()
    u = np.broadcast_to(p.u0, (Bsz, m)).copy(); dhat = np.broadcast_to(p.dhat0, (Bsz, p.nd)).copy()
    Pk = np.broadcast_to(p.P0, (Bsz,) + p.P0.shape).copy() if p.estimator == "kal" else None
    us_k, xs_k = u.copy(), x0_m.copy()
    keys = ("Xp", "X_HAT", "Yp", "Y_HAT", "D_HAT", "XS", "US", "YS", "U", "STATUS_SS", "STATUS_DYN", "ITERS_DYN", "ITERS_SS")
    log = {k: [] for k in keys}
    for k in range(nsteps):
        log["Xp"].append(x.copy()); log["X_HAT"].append(xhat.copy())
        e = np.broadcast_to(p.fy_const, (Bsz, p.ny)) + (dhat @ p.Cd.T if p.nd else 0.0)
        yhat = xhat @ p.C.T + e
        y = x @ p.Cp.T + sched["pyp"][k]
        log["Yp"].append(y.copy()); log["Y_HAT"].append(yhat.copy())
        xi = np.hstack([xhat, dhat])
        if p.estimator == "kal":
            xi, Pk = kalman_batch(p, xi, Pk, y, yhat)
        elif p.estimator == "kalss":
            xi = xi + (y - yhat) @ p.K.T
        xhat, dhat = xi[:, :n].copy(), xi[:, n:].copy()
        if p.dmin is not None:
            dhat = np.minimum(np.maximum(dhat, p.dmin), p.dmax)
        log["D_HAT"].append(dhat.copy())
        t = target_solve(p, td, sched["usp"][k], sched["ysp"][k], sched["xsp"][k], dhat, us_k, max_iter=max_iter,
                         warm=tw if (warm_start and k > 0) else None)
        tw = t["warm"]
        okt = (t["status"] != STATUS_INFEASIBLE)[:, None]
        xs_k = np.where(okt, t["xs"], xs_k); us_k = np.where(okt, t["us"], us_k)
        log["XS"].append(xs_k.copy()); log["US"].append(us_k.copy())
        e = np.broadcast_to(p.fy_const, (Bsz, p.ny)) + (dhat @ p.Cd.T if p.nd else 0.0)
        log["YS"].append(xs_k @ p.C.T + e)
        inst = instance_data(p, sd, xhat, xs_k, us_k, dhat, u)
        warm = None
        if warm_start and k > 0:
            sh = lambda a: np.concatenate([a[:, 1:], a[:, -1:]], axis=1)
            delta = np.maximum(np.maximum(np.abs(xhat - prev_pred).max(axis=1), np.abs(dhat - prev_d).max(axis=1)),
                               np.maximum(np.abs(xs_k - prev_xs).max(axis=1), np.abs(us_k - prev_us).max(axis=1)))
            log.setdefault("WS_DELTA", []).append(delta.copy())
            warm = dict(u=sh(prev["u"]), l_lo=sh(prev["l_lo"]), l_hi=sh(prev["l_hi"]), delta=delta,
                        use=(prev["status"] == STATUS_SOLVED) & (delta <= WS_DELTA))
        o = rpdip_solve(sd, inst, max_iter=max_iter, warm=warm)
        prev = o; prev_d = dhat.copy(); prev_xs = xs_k.copy(); prev_us = us_k.copy()
        oko = (o["status"] != STATUS_INFEASIBLE)[:, None]
        cx = np.broadcast_to(p.fx_const, (Bsz, n)) + (dhat @ p.Bd.T if p.nd else 0.0)
        xhat_hold = xhat @ p.A.T + u @ p.B.T + cx
        u = np.where(oko, o["u0"], u)
        xhat = np.where(oko, o["z1"][:, :n], xhat_hold)
        prev_pred = xhat.copy()
        log["U"].append(u.copy()); log["STATUS_SS"].append(t["status"].copy()); log["STATUS_DYN"].append(o["status"].copy())
        log["ITERS_DYN"].append(o["iters"].copy()); log["ITERS_SS"].append(t["iters"].copy())
        x = p.plant_step(x, u, k * p.h, sched["pxp"][k]) if hasattr(p, "plant_step") else x @ p.Ap.T + u @ p.Bp.T + sched["pxp"][k]
        if noise is not None:        # robustness studies only: seeded process noise [nsteps,B,nxp]
            x = x + noise[k]
    return {k: np.array(v) for k, v in log.items()}
