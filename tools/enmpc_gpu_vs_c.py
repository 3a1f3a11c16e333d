#!/usr/bin/env python3
"""Development probe (GPU box): the economic closed loop on the HIP path (both launch styles, every lane count the horizons allow) against oracle/enmpc_oracle.c.
   tools/enmpc_gpu_vs_c.py [steps] [instances] [N] [N_mhe]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
warnings.simplefilter("ignore")
import mpc_code_amd as m
from mpc_code_amd import enmpc
import enmpc_oracle as eo, enmpc_oracle_c as ec
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 21
B = int(sys.argv[2]) if len(sys.argv) > 2 else 70
over = {"N": int(sys.argv[3]), "N_mhe": int(sys.argv[4])} if len(sys.argv) > 4 else None
path = m.example_path("reactor_enmpc.py")
p = m.load_problem(path, overrides=over)
rng = np.random.default_rng(20250614)
x0 = np.vstack([p.x0_p, rng.uniform([0.5, 0.0], [1.0, 0.5], size=(B - 1, 2))]) if B > 1 else p.x0_p[None]
c = ec.OracleEC(eo.load_problem(path, overrides=over)).closed_loop(nsteps, x0, nthreads=0)
s = enmpc.EnmpcSolver(p)
for kernel in (1, 2):
    t0 = time.time()
    r = enmpc.run_enmpc_closed_loop(p, x0, nsteps, solver=s, kernel=kernel)
    dv = {k: float(np.abs(r[k] - c[k]).max()) for k in ("U", "XS", "US", "X_ES", "Xp")}
    st = {k: int((r[k] != c[k]).sum()) for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE")}
    di = {k: (int(np.abs(r[k].astype(int) - c[k].astype(int)).max()), int((r[k] != c[k]).sum())) for k in ("ITERS_DYN", "ITERS_SS", "ITERS_MHE")}
    print(f"kernel {kernel}: {time.time() - t0:.2f} s, kernel {r['kernel_ms']:.1f} ms | max |dv|", {k: f"{v:.1e}" for k, v in dv.items()}, "| status words that differ", st, "| iterations (max diff, count)", di, flush=True)
print("mean iterations C:", {k: float(c[k].mean()) for k in ("ITERS_DYN", "ITERS_SS", "ITERS_MHE")}, "status max", {k: int(c[k].max()) for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE")})
s.close()
