#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the build container; takes a few minutes).

The reference (CPCLAB-UNIPI/MPC-code) cannot produce vectors: CasADi/IPOPT are not installable here and
it ships none.  These fixtures therefore come from oracle/mpc_oracle.py - dense-KKT interior point followed
by an exact active-set polish - and are *self-certifying*: every solved OCP / target row carries its KKT
residual (stationarity, primal, complementarity, dual sign) of the stated QP, so a reader can re-verify
them with oracle.mpc_oracle.kkt_residual without trusting any solver.

  cstr_shipped.npz   the shipped Ex_LMPC_CSTR scenario, 100 closed-loop steps, one instance
  wb_shipped.npz     the shipped Ex_LMPC_WB scenario, 100 steps
  cstr_box.npz       24 instances from the benchmark's initial-state box, 12 steps each
Each file: per step the inputs of the OCP (XHAT_C, XS, US, D_HAT, U_PREV), its outputs (U, X_NEXT via
X_HAT of the next step), status words, KKT residuals, and the loop logs under the reference's names.
EXACT_DYN / EXACT_SS mark the rows whose active-set polish verified (complementarity exactly zero, primal and
dual feasibility checked): only those are exact optima.  A small KKT residual alone is not a certificate of a
small primal error when bounds are degenerate (s* = l* = 0): there the error of an interior-point answer goes
like the square root of its complementarity residual.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mpc_code_amd as m          # noqa: E402
import mpc_oracle as o            # noqa: E402


def run(p, nsteps, x0s):
    logs = []
    for x0 in x0s:
        logs.append(o.closed_loop(p, nsteps, x0_p=x0, x0_m=x0, ocp=o.ocp_solve_exact, target=o.target_solve_exact))
    return {k: np.stack([lg[k] for lg in logs], axis=1) for k in logs[0]}     # [step][instance][...]


def main():
    t0 = time.time()
    cstr = m.load_problem(m.example_path("cstr_lmpc.py"))
    wb = m.load_problem(m.example_path("wood_berry_lmpc.py"))
    out = run(cstr, 100, [cstr.x0_p])
    np.savez_compressed(os.path.join(HERE, "cstr_shipped.npz"), **out)
    print("cstr_shipped", time.time() - t0, np.bincount(out["STATUS_DYN"].ravel()), np.nanmax(np.where(out["STATUS_DYN"] == 0, out["KKT_DYN"], 0)))
    out = run(wb, 100, [wb.x0_p])
    np.savez_compressed(os.path.join(HERE, "wb_shipped.npz"), **out)
    print("wb_shipped", time.time() - t0, np.bincount(out["STATUS_DYN"].ravel()), np.nanmax(np.where(out["STATUS_DYN"] == 0, out["KKT_DYN"], 0)))
    rng = np.random.default_rng(20250614)
    x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(24, 3))
    out = run(cstr, 12, list(x0))
    out["X0"] = x0
    np.savez_compressed(os.path.join(HERE, "cstr_box.npz"), **out)
    print("cstr_box", time.time() - t0, np.bincount(out["STATUS_DYN"].ravel()), np.nanmax(np.where(out["STATUS_DYN"] == 0, out["KKT_DYN"], 0)))


if __name__ == "__main__":
    main()
