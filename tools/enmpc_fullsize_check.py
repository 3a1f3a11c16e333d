#!/usr/bin/env python3
"""Diagnostic (GPU box): every instance-step of a BASELINE-size batch on the HIP path against oracle/enmpc_oracle.c on the host cores.
   tools/enmpc_fullsize_check.py [B] [steps] [N] [N_mhe] [threads of the C oracle, comma separated; default 64]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mpc_code_amd as m
from mpc_code_amd import enmpc
import enmpc_oracle as eo, enmpc_oracle_c as ec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
over = {"N": int(sys.argv[3]) if len(sys.argv) > 3 else 40, "N_mhe": int(sys.argv[4]) if len(sys.argv) > 4 else 10}
ex = m.example_path("reactor_enmpc.py")
x0 = np.random.default_rng(20250614).uniform([0.5, 0.0], [1.0, 0.5], size=(B, 2))
g = enmpc.run_enmpc_closed_loop(m.load_problem(ex, overrides=over), x0, K)
o = ec.OracleEC(eo.load_problem(ex, overrides=over), fast=True)
for th in [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "64").split(",")]:
    t0 = time.time(); c = o.closed_loop(K, x0, nthreads=th); print(f"C oracle on {th or o.max_threads()} threads: {B * K / (time.time() - t0):.0f} steps/s")
for k in ("U", "XS", "US", "X_ES", "Xp"):
    print(k, "max |gpu - c| =", float(np.abs(g[k] - c[k]).max()))
for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE", "ITERS_DYN", "ITERS_SS", "ITERS_MHE"):
    d = g[k].astype(int) - c[k].astype(int)
    print(k, "differ at", int((d != 0).sum()), "of", d.size, "instance-steps; largest difference", int(np.abs(d).max()), "steps:", np.unique(np.where(d != 0)[0]).tolist()[:12])
