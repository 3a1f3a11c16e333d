#!/usr/bin/env python3
"""Diagnostic: shader-clock ticks per part of an interior point iteration of the economic path's stage solver (library built with -DMPC_STAMPS), per kernel
of the split pipeline:   tools/enmpc_ipm_stamps.py [N] [N_mhe] [batch] [steps]"""
import ctypes as ct, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import enmpc, econcodegen
over = {"N": int(sys.argv[1]) if len(sys.argv) > 1 else 40, "N_mhe": int(sys.argv[2]) if len(sys.argv) > 2 else 10}
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
K = int(sys.argv[4]) if len(sys.argv) > 4 else 20
p = m.load_problem(m.example_path("reactor_enmpc.py"), overrides=over)
lib = econcodegen.build_enmpc_library(p, extra_flags=["-DMPC_STAMPS"])
s = enmpc.EnmpcSolver(p, lib_path=lib)
x0 = np.random.default_rng(1).uniform([0.5, 0.0], [1.0, 0.5], size=(B, 2))
s.alloc(B, K); s.set_state(x0); s.set_kernel(2); s.set_groups(1)
names = ["scaling, push", "linearisation", "scaling of the stage", "least-squares multipliers", "slacks, error, stop tests", "mu, barrier terms", "backward sweep(s)",
         "forward sweep", "step sizes, thresholds", "trial points", "acceptance tests", "second-order corrections", "filter, new iterate"]
buf = np.zeros(256 * 16, np.uint64)
u = np.tile(p.u0, (B, 1)); xp = x0.copy()
# the per-call seam launches one kernel per call: the stamps are read after each
def read():
    s.lib.enmpc_debug_ipm_stamps(buf.ctypes.data_as(ct.c_void_p), 256 * 16, 1)
    return buf.reshape(256, 16).astype(float)[:, :13].copy()
s.lib.enmpc_debug_ipm_stamps(None, 0, 1)
acc = {"mhe": np.zeros(13), "ocp": np.zeros(13)}; its = {"mhe": 0.0, "ocp": 0.0}
for k in range(K):
    xhat, dhat, xes, st, it = s.mhe_update(xp, u)
    acc["mhe"] += read().mean(axis=0); its["mhe"] += it.mean()
    xs, us, st, it = s.target_solve(dhat)
    read()
    u, xn, st, it = s.ocp_solve(xhat, dhat, xs, us)
    acc["ocp"] += read().mean(axis=0); its["ocp"] += it.mean()
    xp = s.plant_step(u, xp)
for ph in ("mhe", "ocp"):
    t = acc[ph] / K
    print(f"{ph}: mean iterations {its[ph] / K:.2f}; kticks per solve (mean of the first 256 waves), share")
    for n, v in zip(names, t):
        print(f"   {n:32s} {v / 1e3:9.1f}  {100 * v / t.sum():5.1f} %")
    print(f"   {'total':32s} {t.sum() / 1e3:9.1f}")
