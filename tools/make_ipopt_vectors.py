#!/usr/bin/env python3
"""Optional: golden vectors from the reference's own solver stack (CasADi + IPOPT), SURVEY.md section 8c.

The build container has no CasADi, so parity with IPOPT is pinned by mathematics (tests/golden/make_golden.py).  On any
machine where ``import casadi`` succeeds this script adds true IPOPT vectors:

    python3 tools/make_ipopt_vectors.py            # writes tests/golden/ipopt_*.npz
    python3 tools/make_ipopt_vectors.py econ /path/to/MPC-code      # the economic example's target NLP, OCP and estimator NLP: tests/golden/ipopt_enmpc.npz (see econ_vectors)

It poses the NLP of ``opt_dyn`` (Control_Calc.py:20-260) in its own variable / constraint layout - built here from the
dense matrices of oracle/mpc_oracle.py:ocp_qp, which restates Control_Calc.py:126-252 row by row - hands it to
``nlpsol('solver', 'ipopt', ...)`` with the reference's options (MPC_code.py:262-263: max_iter = Sol_itmax,
hessian_constant = yes, print_level 0, sb yes, print_time 0; everything else IPOPT's default) and the reference's
cold guess (MPC_code.py:743-756), and stores inputs, ``sol['x']``, ``u* = w[nx:nx+nu]``, ``x+ = w[nxu:nxu+nx]`` and
``return_status``.  The same for the target NLP of ``opt_ss`` (Target_Calc.py:20-161).  Nothing of the reference's Python
is imported or copied; the fixtures are data.  tests/test_oracle.py::test_ipopt_vectors_if_present compares the oracle
(and, on the GPU, tests/test_gpu_parity.py::test_ocp_matches_ipopt_vectors_if_present the HIP path) with them at 1e-6.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def ipopt_qp(ca, H, g, E, e, G, lo, hi, w0, max_iter):
    """min 1/2 w'Hw + g'w  s.t.  E w = e, lo <= G w <= hi  with IPOPT under the reference's options."""
    nw = H.shape[0]
    w = ca.MX.sym("w", nw)
    f = 0.5 * ca.mtimes([w.T, ca.DM(H), w]) + ca.mtimes(ca.DM(g).T, w)
    cons = ca.vertcat(ca.mtimes(ca.DM(E), w), ca.mtimes(ca.DM(G), w)) if G.shape[0] else ca.mtimes(ca.DM(E), w)
    opts = {"ipopt.max_iter": int(max_iter), "ipopt.hessian_constant": "yes", "ipopt.print_level": 0, "ipopt.sb": "yes", "print_time": 0}
    solver = ca.nlpsol("solver", "ipopt", {"x": w, "f": f, "g": cons}, opts)
    lbg = np.concatenate([e, lo]) if G.shape[0] else e
    ubg = np.concatenate([e, hi]) if G.shape[0] else e
    sol = solver(x0=w0, lbg=lbg, ubg=ubg)
    return np.array(sol["x"]).ravel(), float(sol["f"]), solver.stats()["return_status"]


def econ_vectors(ca, ref_dir):
    """The economic example's target NLP and OCP (mpc-code_amd/examples/reactor_enmpc.py, which needs CasADi and the reference's ``Utilities.py`` to run as it is)
    as THIS project discretises them - Mx resp. quad_steps classical Runge-Kutta steps per interval, cost quadrature in the integrator state (oracle/enmpc_oracle.py:
    fx_model, ocp_stage; the reference integrates the interval with IDAS) - solved by IPOPT at its defaults + max_iter = Sol_itmax (MPC_code.py:262-263).  What the vectors
    pin is the restated interior point: solution, objective, ITERATION COUNT and return status of the same NLP from the same guess.
    Written without a CasADi to run it against (the build container has none): treat a failure in here as a bug of this script, not of the library."""
    import runpy
    sys.path.insert(0, ref_dir)      # the Ex-file does `from Utilities import *`
    ns = runpy.run_path(os.path.join(ROOT, "mpc-code_amd", "examples", "reactor_enmpc.py"))
    nx, nu, ny, nd = (ns[k].size1() for k in ("x", "u", "y", "d"))
    N, h, Mx, quad = int(ns["N"]), float(ns["h"]), int(ns.get("Mx", 10)), 20
    Bd, Cd = ca.DM(np.asarray(ns["Bd"], dtype=float)), ca.DM(np.asarray(ns["Cd"], dtype=float))
    fxm, fobj, fss, vfin = ns["User_fxm_Cont"], ns["User_fobj_Cont"], ns["User_fssobj"], ns.get("User_vfin")
    vec = lambda v, n_, fill: np.full(n_, fill) if v is None else np.asarray(v, dtype=float).reshape(n_)
    opts = {"ipopt.max_iter": int(ns.get("Sol_itmax", 100)), "ipopt.print_level": 0, "ipopt.sb": "yes", "print_time": 0}

    def rk4(rhs, z, steps):
        dt = h / steps
        for _ in range(steps):
            k1 = rhs(z); k2 = rhs(z + 0.5 * dt * k1); k3 = rhs(z + 0.5 * dt * k2); k4 = rhs(z + dt * k3)
            z = z + dt / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        return z
    zero_x = ca.DM.zeros(nx)
    dpar = ca.SX.sym("dpar", nd)
    # ---- target: w = [xs; us; ys], g = [Fx_model(xs, us, d) - xs; xs + Cd d - ys] (Target_Calc.py:73-81), cold start (MPC_code.py:696-700)
    wt = ca.SX.sym("wt", nx + nu + ny)
    xs_, us_, ys_ = wt[:nx], wt[nx:nx + nu], wt[nx + nu:]
    Fx = rk4(lambda z: fxm(z, us_, dpar, 0.0, zero_x), xs_, Mx) + ca.mtimes(Bd, dpar)
    gt = ca.vertcat(Fx - xs_, xs_ + ca.mtimes(Cd, dpar) - ys_)
    ft = fss(xs_, us_, ys_, ca.DM.zeros(nx), ca.DM.zeros(nu), ca.DM.zeros(ny))
    st = ca.nlpsol("target", "ipopt", {"x": wt, "p": dpar, "f": ft, "g": gt}, opts)
    pick = lambda b, sfx, n_, fill: vec(ns.get(b + sfx) if ns.get(b + sfx) is not None else ns.get(b), n_, fill)
    lot = np.concatenate([pick("xmin", "_ss", nx, -np.inf), pick("umin", "_ss", nu, -np.inf), pick("ymin", "_ss", ny, -np.inf)])
    hit = np.concatenate([pick("xmax", "_ss", nx, np.inf), pick("umax", "_ss", nu, np.inf), pick("ymax", "_ss", ny, np.inf)])
    # ---- OCP: w = [x0, u0, x1, ..., x_N] (Control_Calc.py:31-37), x0 fixed by equal bounds (MPC_code.py:734), g_k = x_{k+1} - X_k(h), f = sum q_k + Vfin
    nz = nx + nu
    wd = ca.SX.sym("wd", nz * N + nx)
    par = ca.SX.sym("par", nx + nu + nd)      # xs, us, d
    pxs, pus, pd = par[:nx], par[nx:nx + nu], par[nx + nu:]
    pys = pxs + ca.mtimes(Cd, pd)
    g, f = [], 0
    for k in range(N):
        xk, uk, xn = wd[nz * k: nz * k + nx], wd[nz * k + nx: nz * (k + 1)], wd[nz * (k + 1): nz * (k + 1) + nx]
        zk = rk4(lambda z: ca.vertcat(fxm(z[:nx], uk, pd, 0.0, zero_x), fobj(z[:nx], uk, z[:nx] + ca.mtimes(Cd, pd), pxs, pus, pys)), ca.vertcat(xk, 0), quad)
        g.append(xn - zk[:nx]); f = f + zk[nx]
    if vfin is not None:
        f = f + vfin(wd[nz * N:], pxs)
    sd = ca.nlpsol("ocp", "ipopt", {"x": wd, "p": par, "f": f, "g": ca.vertcat(*g)}, opts)
    umin, umax, xmin, xmax = pick("umin", "_dyn", nu, -np.inf), pick("umax", "_dyn", nu, np.inf), pick("xmin", "_dyn", nx, -np.inf), pick("xmax", "_dyn", nx, np.inf)
    x0_m, u0 = vec(ns["x0_m"], nx, 0.0), vec(ns["u0"], nu, 0.0)
    rng = np.random.default_rng(20250614)
    rec = {k: [] for k in ("D", "XHAT", "WT", "FT", "ITERS_T", "STATUS_T", "W", "F", "ITERS", "STATUS")}
    for _ in range(24):
        d = rng.uniform(-0.05, 0.05, nd); xhat = rng.uniform([0.5, 0.0], [1.0, 0.5])
        w0t = np.concatenate([x0_m, u0, x0_m + np.asarray(ns["Cd"], dtype=float) @ d])
        sol = st(x0=w0t, p=d, lbx=lot, ubx=hit, lbg=0, ubg=0)
        wts = np.array(sol["x"]).ravel()
        rec["D"].append(d); rec["XHAT"].append(xhat); rec["WT"].append(wts); rec["FT"].append(float(sol["f"]))
        rec["ITERS_T"].append(int(st.stats()["iter_count"])); rec["STATUS_T"].append(st.stats()["return_status"])
        lo = np.concatenate([np.concatenate([xmin, umin])] * N + [xmin]); hi = np.concatenate([np.concatenate([xmax, umax])] * N + [xmax])
        lo[:nx] = hi[:nx] = xhat
        w0 = np.concatenate([np.concatenate([x0_m, u0])] * N + [x0_m]); w0[:nx] = xhat      # MPC_code.py:740-756
        sol = sd(x0=w0, p=np.concatenate([wts[:nx], wts[nx:nx + nu], d]), lbx=lo, ubx=hi, lbg=0, ubg=0)
        rec["W"].append(np.array(sol["x"]).ravel()); rec["F"].append(float(sol["f"]))
        rec["ITERS"].append(int(sd.stats()["iter_count"])); rec["STATUS"].append(sd.stats()["return_status"])
    # ---- estimator: mhe_opt's NLP for a window of Nw stages in its own layout w = [x0, v0, w0, x1, ..., x_Nw] with [x; d] for x (Utilities.py:831-846),
    # g = [Fy(X_k) + V_k - Y_k; Fx_mhe(X_k, U_k, W_k) - X_{k+1}] (:909-926), f = sum F_obj_mhe(W_k, V_k) + 1/2 (x0 - x_bar)' P^-1 (x0 - x_bar) (:928-945), boxes on every X_k
    # (:956-966); IPOPT at the estimator's options (MPC_code.py:383: tol 1e-10), first guess = x_bar propagated without noise (Estimator.py:503-512)
    if ns.get("mhe", False):
        ne, nwv = nx + nd, ns["w"].size1()
        fmhe, cmhe = ns["User_fx_mhe_Cont"], ns["User_fobj_mhe"]
        G = ca.DM(np.eye(ne) if ns.get("G_mhe") is None else np.asarray(ns["G_mhe"], dtype=float))
        P0, xbar0 = np.asarray(ns["P0"], dtype=float), vec(ns["x_bar"], ne, 0.0)
        lo_e = np.concatenate([vec(ns.get("xmin_mhe", ns.get("xmin")), nx, -np.inf), vec(ns.get("dmin"), nd, -np.inf)])
        hi_e = np.concatenate([vec(ns.get("xmax_mhe", ns.get("xmax")), nx, np.inf), vec(ns.get("dmax"), nd, np.inf)])
        opts_mhe = dict(opts); opts_mhe["ipopt.tol"] = 1e-10
        nb = ne + ny + nwv

        def fx_mhe(X, U, W):
            xn = rk4(lambda z: fmhe(z, U, X[nx:], 0.0, zero_x, W), X[:nx], Mx) + ca.mtimes(Bd, X[nx:])
            return ca.vertcat(xn, X[nx:]) + ca.mtimes(G, W)
        rec.update({k: [] for k in ("MHE_N", "MHE_U", "MHE_Y", "MHE_XBAR", "MHE_P", "MHE_W0", "MHE_W", "MHE_F", "MHE_ITERS", "MHE_STATUS")})
        for Nw in (1, 2, 3, 5, 8, 12, 20, 20):
            Us, Ys = rng.uniform(0.2, 1.2, (Nw, nu)), rng.uniform([0.5, 0.0], [1.0, 0.5], (Nw, ny))
            xb = xbar0 + np.concatenate([rng.uniform(-0.1, 0.1, nx), np.zeros(nd)]); Pk = P0 * rng.uniform(0.5, 2.0)
            wm = ca.SX.sym("wm", Nw * nb + ne)
            g, f = [], 0
            for k in range(Nw):
                X, V, W, Xn = wm[nb * k: nb * k + ne], wm[nb * k + ne: nb * k + ne + ny], wm[nb * k + ne + ny: nb * (k + 1)], wm[nb * (k + 1): nb * (k + 1) + ne]
                g += [X[:nx] + ca.mtimes(Cd, X[nx:]) + V - ca.DM(Ys[k]), fx_mhe(X, ca.DM(Us[k]), W) - Xn]
                f = f + cmhe(W, V, 0.0)
            e0 = wm[:ne] - ca.DM(xb)
            f = f + 0.5 * ca.mtimes([e0.T, ca.DM(np.linalg.inv(Pk)), e0])
            sm = ca.nlpsol("mhe", "ipopt", {"x": wm, "f": f, "g": ca.vertcat(*g)}, opts_mhe)
            lo, hi = np.full(Nw * nb + ne, -np.inf), np.full(Nw * nb + ne, np.inf)
            w0, xg = np.zeros(Nw * nb + ne), xb.copy()
            prop = ca.Function("prop", [wm[:ne], wm[ne:ne + nu]], [fx_mhe(wm[:ne], wm[ne:ne + nu], ca.DM.zeros(nwv))])
            for k in range(Nw + 1):
                lo[nb * k: nb * k + ne], hi[nb * k: nb * k + ne] = lo_e, hi_e
                w0[nb * k: nb * k + ne] = xg
                if k < Nw:
                    xg = np.array(prop(xg, Us[k])).ravel()
            sol = sm(x0=w0, lbx=lo, ubx=hi, lbg=0, ubg=0)
            for k_, v_ in (("MHE_N", Nw), ("MHE_U", Us), ("MHE_Y", Ys), ("MHE_XBAR", xb), ("MHE_P", Pk), ("MHE_W0", w0), ("MHE_W", np.array(sol["x"]).ravel()), ("MHE_F", float(sol["f"])),
                           ("MHE_ITERS", int(sm.stats()["iter_count"])), ("MHE_STATUS", sm.stats()["return_status"])):
                rec[k_].append(v_)
    np.savez_compressed(os.path.join(GOLD, "ipopt_enmpc.npz"), **{k: (np.array(v, dtype=object) if k.startswith("MHE_") and k not in ("MHE_N", "MHE_F", "MHE_ITERS", "MHE_STATUS") else np.array(v)) for k, v in rec.items()})
    print("enmpc written: 24 target + OCP solves, iterations", rec["ITERS_T"][:6], rec["ITERS"][:6], "; estimator windows", rec.get("MHE_N"), "iterations", rec.get("MHE_ITERS"))


def main():
    try:
        import casadi as ca
    except ImportError:
        print("casadi is not importable here: no IPOPT vectors written (the mathematical pinning of tests/golden/ stands)")
        return 0
    if len(sys.argv) > 2 and sys.argv[1] == "econ":      # python3 tools/make_ipopt_vectors.py econ /path/to/MPC-code   (the reference tree: its Utilities.py)
        return econ_vectors(ca, sys.argv[2]) or 0
    import mpc_code_amd as m
    import mpc_oracle as o
    rng = np.random.default_rng(20250614)
    for name, ex, B in (("cstr", "cstr_lmpc.py", 48), ("wb", "wood_berry_lmpc.py", 16)):
        p = m.load_problem(m.example_path(ex))
        n, mu, N = p.nx, p.nu, p.N
        if name == "cstr":
            xh = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3))
        else:
            xh = 0.05 * rng.standard_normal((B, n))
        dh = 0.02 * rng.standard_normal((B, p.nd)); up = np.tile(p.u0, (B, 1)) + 0.1 * rng.standard_normal((B, mu))
        sched = p.schedules(1)
        rec = dict(XHAT=xh, DHAT=dh, U_PREV=up, XS=[], US=[], W=[], U=[], XNEXT=[], F=[], STATUS=[], XS_T=[], US_T=[], STATUS_T=[])
        for b in range(B):
            Ht, gt, Et, et, Gt, lot, hit = o.target_qp(p, sched["usp"][0], sched["ysp"][0], sched["xsp"][0], dh[b], up[b])
            w0t = np.concatenate([p.x0_m, p.u0, o.model_fy(p, p.x0_m, dh[b])])                 # MPC_code.py:696-700: cold, always
            wt, _, stt = ipopt_qp(ca, Ht, gt, Et, et, Gt, lot, hit, w0t, p.max_iter)
            xs, us = wt[:n], wt[n:n + mu]
            H, g, E, e, G, lo, hi = o.ocp_qp(p, xh[b], xs, us, dh[b], up[b])
            w0 = np.concatenate([np.tile(np.concatenate([p.x0_m, p.u0]), N), p.x0_m])           # MPC_code.py:743-756
            w, f, st = ipopt_qp(ca, H, g, E, e, G, lo, hi, w0, p.max_iter)
            for k, v in (("XS", xs), ("US", us), ("W", w), ("U", w[n:n + mu]), ("XNEXT", w[n + mu:2 * n + mu]), ("F", f), ("STATUS", st),
                         ("XS_T", xs), ("US_T", us), ("STATUS_T", stt)):
                rec[k].append(v)
        out = {k: np.array(v) for k, v in rec.items()}
        np.savez_compressed(os.path.join(GOLD, f"ipopt_{name}.npz"), **out)
        print(name, "written:", B, "instances,", sum(s == "Solve_Succeeded" for s in rec["STATUS"]), "Solve_Succeeded,",
              sum(s == "Infeasible_Problem_Detected" for s in rec["STATUS"]), "infeasible")
    return 0


if __name__ == "__main__":
    sys.exit(main())
