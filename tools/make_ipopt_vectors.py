#!/usr/bin/env python3
"""Optional: golden vectors from the reference's own solver stack (CasADi + IPOPT), SURVEY.md section 8c.

The build container has no CasADi, so parity with IPOPT is pinned by mathematics (tests/golden/make_golden.py).  On any
machine where ``import casadi`` succeeds this script adds true IPOPT vectors:

    python3 tools/make_ipopt_vectors.py            # writes tests/golden/ipopt_*.npz

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


def main():
    try:
        import casadi as ca
    except ImportError:
        print("casadi is not importable here: no IPOPT vectors written (the mathematical pinning of tests/golden/ stands)")
        return 0
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
