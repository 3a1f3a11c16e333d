#!/usr/bin/env python3
"""Generate tests/golden/nmpc_cstr.npz (run in the build container; about ten minutes).

The non-linear path's reference run needs CasADi + IPOPT, which are not installable here, and the reference ships no
vectors for it: PARITY UNPINNED against a reference run.  The fixture comes from oracle/nmpc_oracle.py (which reads the example with its own
loader, oracle/exnum.py: nothing of the product is imported here) - SQP whose QPs are
solved by the dense interior point + exact active-set polish of oracle/mpc_oracle.py - and is self-certifying for the
converged mode: every OCP row carries the residuals of the NLP's own KKT conditions (dynamics defect of the RK4 model,
stationarity with multipliers fitted on the active set, bound violation), the conditions IPOPT terminates on.

  rti_*   one real-time iteration per step (max_sqp = 1), 40 steps (crossing the feed-flow change at t = 5), 3 instances
  sqp_*   every OCP iterated to its KKT point (tolerance 1e-9 on the trajectory step), 8 steps, 2 instances
Instance 0 is the shipped start; the others start from perturbed plant/model states.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nmpc_oracle as no          # noqa: E402


def starts(p, n):
    rng = np.random.default_rng(20240611)
    x0 = np.tile(p.x0_p, (n, 1))
    x0[1:] += rng.uniform(-1.0, 1.0, size=(n - 1, p.nx)) * np.array([0.02, 2.0, 0.02])
    return x0


def run(p, nsteps, x0s, **kw):
    logs = [no.closed_loop(p, nsteps, x0_p=x0, x0_m=x0, **kw) for x0 in x0s]
    return {k: np.stack([lg[k] for lg in logs], axis=1) for k in logs[0]}


def quadtank():
    """tests/golden/nmpc_quadtank.npz: the discrete-time quadruple tank (examples/quadtank_nmpc_dis.py, N = 20): real-time iteration
    over 16 steps (plant disturbance from t = 0, the set-point change of tests/nmpc_cases.py at t > 30) for the shipped start and a perturbed one; 9 steps with
    every OCP iterated to its KKT point."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from nmpc_cases import quadtank_mild_setpoints
    p = no.load_problem(os.path.join(ROOT, "mpc-code_amd", "examples", "quadtank_nmpc_dis.py"), overrides={"defSP": quadtank_mild_setpoints})
    rng = np.random.default_rng(20240612)
    x0 = np.tile(p.x0_p, (2, 1)); x0[1, 2:] += rng.uniform(-1, 1, 4) * [1.0, 1.0, 0.3, 0.3]
    out = {}
    t0 = time.time()
    r = run(p, 16, x0, max_sqp=1)
    out.update({"rti_" + k: v for k, v in r.items()}); out["rti_x0"] = x0
    print("quadtank rti", time.time() - t0, flush=True)
    r = run(p, 9, x0[:1], max_sqp=50, sqp_tol=1e-9, certify=True)
    out.update({"sqp_" + k: v for k, v in r.items()}); out["sqp_x0"] = x0[:1]
    print("quadtank sqp", time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(HERE, "nmpc_quadtank.npz"), **out)
    W = out["rti_W"]; lev = np.stack([W[:, :, 8 * j + 2:8 * j + 6] for j in range(p.N + 1)], axis=2)
    print("smallest predicted level", lev.min())
    print("max KKT: defect", np.nanmax(out["sqp_KKT_DEFECT"]), "stationarity", np.nanmax(out["sqp_KKT_STAT"]), "violation", np.nanmax(out["sqp_KKT_VIOL"]),
          "status", out["sqp_STATUS_DYN"].ravel(), out["rti_STATUS_DYN"].ravel())


def reactor():
    """tests/golden/nmpc_reactor.npz: the two-state, one-input reactor (examples/reactor_nmpc.py, N = 25): real-time iteration over 30
    steps (plant / model mismatch from t = 0, set-point change at step 21) for the shipped start and two perturbed ones (plant and model
    start apart); 6 steps with every OCP iterated to its KKT point."""
    p = no.load_problem(os.path.join(ROOT, "mpc-code_amd", "examples", "reactor_nmpc.py"))
    rng = np.random.default_rng(20240613)
    x0 = np.tile(p.x0_p, (3, 1)); x0[1:] += rng.uniform(-1, 1, size=(2, 2)) * 0.05
    xm = x0.copy(); xm[2] += rng.uniform(-1, 1, 2) * 0.03
    out = {}
    t0 = time.time()
    logs = [no.closed_loop(p, 30, x0_p=a, x0_m=b, max_sqp=1) for a, b in zip(x0, xm)]
    out.update({"rti_" + k: np.stack([lg[k] for lg in logs], axis=1) for k in logs[0]}); out["rti_x0"] = x0; out["rti_xm"] = xm
    print("reactor rti", time.time() - t0, flush=True)
    logs = [no.closed_loop(p, 6, x0_p=a, x0_m=b, max_sqp=50, sqp_tol=1e-9, certify=True) for a, b in zip(x0[:2], xm[:2])]
    out.update({"sqp_" + k: np.stack([lg[k] for lg in logs], axis=1) for k in logs[0]}); out["sqp_x0"] = x0[:2]; out["sqp_xm"] = xm[:2]
    print("reactor sqp", time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(HERE, "nmpc_reactor.npz"), **out)
    print("max KKT: defect", np.nanmax(out["sqp_KKT_DEFECT"]), "stationarity", np.nanmax(out["sqp_KKT_STAT"]), "violation", np.nanmax(out["sqp_KKT_VIOL"]),
          "status", out["sqp_STATUS_DYN"].ravel(), np.bincount(out["rti_STATUS_DYN"].ravel()), "sqp iterations", out["sqp_SQP_DYN"].ravel())


def main():
    if "quadtank" in sys.argv:
        return quadtank()
    if "reactor" in sys.argv:
        return reactor()
    p = no.load_problem(os.path.join(ROOT, "mpc-code_amd", "examples", "cstr_nmpc.py"))
    out = {}
    t0 = time.time()
    x0 = starts(p, 3)
    r = run(p, 40, x0, max_sqp=1)
    out.update({"rti_" + k: v for k, v in r.items()}); out["rti_x0"] = x0
    print("rti", time.time() - t0, flush=True)
    r = run(p, 8, x0[:2], max_sqp=50, sqp_tol=1e-9, certify=True)
    out.update({"sqp_" + k: v for k, v in r.items()}); out["sqp_x0"] = x0[:2]
    print("sqp", time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(HERE, "nmpc_cstr.npz"), **out)
    for k, v in out.items():
        print(k, v.shape)
    print("max KKT: defect", out["sqp_KKT_DEFECT"].max(), "stationarity", out["sqp_KKT_STAT"].max(), "violation", out["sqp_KKT_VIOL"].max())


if __name__ == "__main__":
    main()
