#!/usr/bin/env python3
"""Generate tests/golden/enmpc_reactor.npz (run in the build container; about three minutes).

The economic example's reference run needs CasADi + IPOPT + IDAS, which are not installable here, and the reference ships no vectors:
PARITY UNPINNED against a reference run.  The fixture comes from oracle/enmpc_oracle.py (dense interior point with complex-step
derivatives) on mpc-code_amd/examples/reactor_enmpc.py - the problem of the reference's Ex_ENMPC.py - and is self-certifying: every
NLP solved on the way (estimator, target, OCP) carries the largest residual of its own KKT conditions, the conditions IPOPT
terminates on.

  ship_*   the example as shipped (N = 25, N_mhe = 10), 21 steps = its Nsim, the shipped start
  c4_*     BASELINE configs[3]: N = 40, 10 steps, 3 starts of the benchmark box x0_p ~ U([0.5, 1] x [0, 0.5])
  c5_*     BASELINE configs[4]: N_mhe = 20, 24 steps (the window fills at step 19: growing window, then the smoothing update), 2 starts
  flt_*    the example with mhe_up = 'filter' (Estimator.py:627-649,740-748) and N_mhe = 6, 16 steps, 2 starts

``make_enmpc_golden.py ekf`` writes tests/golden/enmpc_reactor_ekf.npz instead (half a minute): the example with the other position of its estimator switch
(mpc-code_amd/examples/reactor_enmpc_ekf.py: extended Kalman filter on [x; d], Ex_ENMPC.py:109-123)
  ekf_*    21 steps from the shipped start and two starts of the benchmark box
  sat_*    the same with the disturbance estimate saturated (dmin / dmax, MPC_code.py:657-664), 12 steps, 2 starts
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import enmpc_oracle as eo          # noqa: E402

EX = os.path.join(ROOT, "mpc-code_amd", "examples", "reactor_enmpc.py")
KEYS = ("U", "X_HAT", "D_HAT", "XS", "US", "Xp", "X_ES", "STATUS_DYN", "STATUS_SS", "ITERS_DYN", "ITERS_SS", "ITERS_MHE", "KKT_DYN", "KKT_SS", "KKT_MHE", "P_K")


def box(n, seed=20250614):
    return np.random.default_rng(seed).uniform([0.5, 0.0], [1.0, 0.5], size=(n, 2))


def run(p, nsteps, x0s):
    logs = [eo.closed_loop(p, nsteps, x0_p=x0, certify=True) for x0 in x0s]
    return {k: np.stack([lg[k] for lg in logs], axis=1) for k in KEYS if k in logs[0]}


def main():
    out = {}
    t0 = time.time()
    p = eo.load_problem(EX)
    out.update({"ship_" + k: v for k, v in run(p, p.Nsim, p.x0_p[None]).items()}); out["ship_x0"] = p.x0_p[None]
    print("shipped", time.time() - t0)
    p4 = eo.load_problem(EX, overrides={"N": 40})
    x4 = box(3)
    out.update({"c4_" + k: v for k, v in run(p4, 10, x4).items()}); out["c4_x0"] = x4
    print("config 4", time.time() - t0)
    p5 = eo.load_problem(EX, overrides={"N_mhe": 20})
    x5 = box(2, seed=7)
    out.update({"c5_" + k: v for k, v in run(p5, 24, x5).items()}); out["c5_x0"] = x5
    print("config 5", time.time() - t0)
    pf = eo.load_problem(EX, overrides={"mhe_up": "filter", "N_mhe": 6})
    xf = np.vstack([pf.x0_p[None], box(1, seed=11)])
    out.update({"flt_" + k: v for k, v in run(pf, 16, xf).items()}); out["flt_x0"] = xf
    print("filter update", time.time() - t0)
    for pre in ("ship_", "c4_", "c5_", "flt_"):
        worst = max(float(out[pre + k].max()) for k in ("KKT_DYN", "KKT_SS", "KKT_MHE"))
        print(pre, "largest KKT residual", worst, "all solved", int(out[pre + "STATUS_DYN"].max()) == 0 and int(out[pre + "STATUS_SS"].max()) == 0)
    np.savez_compressed(os.path.join(HERE, "enmpc_reactor.npz"), **out)


def main_ekf():
    ex = os.path.join(ROOT, "mpc-code_amd", "examples", "reactor_enmpc_ekf.py")
    out = {}
    p = eo.load_problem(ex)
    x0 = np.vstack([p.x0_p[None], box(2, seed=5)])
    out.update({"ekf_" + k: v for k, v in run(p, p.Nsim, x0).items()}); out["ekf_x0"] = x0
    over = {"dmin": np.array([-0.05, -0.02]), "dmax": np.array([0.03, 0.05])}
    ps = eo.load_problem(ex, overrides=over)
    xs = box(2, seed=9)
    out.update({"sat_" + k: v for k, v in run(ps, 12, xs).items()}); out["sat_x0"] = xs; out["sat_dmin"], out["sat_dmax"] = over["dmin"], over["dmax"]
    for pre in ("ekf_", "sat_"):
        worst = max(float(out[pre + k].max()) for k in ("KKT_DYN", "KKT_SS"))
        print(pre, "largest KKT residual", worst, "all solved", int(out[pre + "STATUS_DYN"].max()) == 0 and int(out[pre + "STATUS_SS"].max()) == 0,
              "dhat in", float(out[pre + "D_HAT"].min()), float(out[pre + "D_HAT"].max()))
    np.savez_compressed(os.path.join(HERE, "enmpc_reactor_ekf.npz"), **out)


if __name__ == "__main__":
    main_ekf() if sys.argv[1:] == ["ekf"] else main()
