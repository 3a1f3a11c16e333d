#!/usr/bin/env python3
"""GPU probe of the non-linear path: statuses, SQP/IPM iteration counts, timing; parity with tests/golden/nmpc_cstr.npz if present."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import nmpc

p = m.load_problem(os.path.join(ROOT, "mpc-code_amd", "examples", "cstr_nmpc.py"))
s = nmpc.NmpcSolver(p)
print(s.build_info())
gp = os.path.join(ROOT, "tests", "golden", "nmpc_cstr.npz")
if os.path.exists(gp):
    g = np.load(gp)
    for mode, ms in (("rti", 1), ("sqp", 50)):
        x0 = g[mode + "_x0"]; ns = g[mode + "_U"].shape[0]
        r = nmpc.run_nmpc_closed_loop(p, x0, x0, nsteps=ns, solver=s, max_sqp=ms, sqp_tol=1e-9)
        for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
            e = np.abs(r[k] - g[f"{mode}_{k}"])
            print(mode, k, "max err", e.max(), "rel", (e / (1 + np.abs(g[f'{mode}_{k}']))).max(), "first bad step", int(np.argmax(e.max(axis=(1, 2)) > 1e-6)))
        print(mode, "status dyn", np.unique(r["STATUS_DYN"], return_counts=True), "golden", np.unique(g[mode + "_STATUS_DYN"], return_counts=True))
        print(mode, "status ss", np.unique(r["STATUS_SS"], return_counts=True), "sqp_dyn", r["SQP_DYN"].T.tolist()[0][:12], "golden", g[mode + "_SQP_DYN"].T.tolist()[0][:12])
        print(mode, "iters_dyn", r["ITERS_DYN"].T.tolist()[0][:12], "sqp_ss", r["SQP_SS"].T.tolist()[0][:12])
for B, ns, ms in ((256, 20, 1), (4096, 20, 1), (16384, 20, 1), (16384, 20, 10)):
    rng = np.random.default_rng(1)
    x0 = np.tile(p.x0_p, (B, 1)) + rng.uniform(-1, 1, size=(B, 3)) * np.array([0.02, 2.0, 0.02])
    t0 = time.time()
    r = nmpc.run_nmpc_closed_loop(p, x0, x0, nsteps=ns, solver=s, max_sqp=ms)
    wall = time.time() - t0
    ms_k = r["TIME_DYN"].sum() * 1e3
    print(f"B={B} steps={ns} max_sqp={ms}: kernel {ms_k:.1f} ms  {B * ns / ms_k * 1e3 / 1e6:.3f} M NMPC steps/s (wall {wall:.2f} s)  status dyn",
          dict(zip(*[a.tolist() for a in np.unique(r['STATUS_DYN'], return_counts=True)])), "ss", dict(zip(*[a.tolist() for a in np.unique(r['STATUS_SS'], return_counts=True)])),
          "mean ipm it", r["ITERS_DYN"].mean(), "mean sqp", r["SQP_DYN"].mean(), "finite", bool(np.isfinite(r["U"]).all()))
