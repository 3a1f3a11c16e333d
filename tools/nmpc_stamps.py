#!/usr/bin/env python3
"""Diagnostic: shader-clock ticks per phase of the non-linear kernels (library built with -DMPC_STAMPS):
   tools/nmpc_stamps.py <lib> [kernel 1|3|4] [batch]      (4: the wave-style launch of the split pipeline)"""
import ctypes as ct, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import nmpc
p = m.load_problem(m.example_path("cstr_nmpc.py"))
kern = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
s = nmpc.NmpcSolver(p, lib_path=sys.argv[1]); s.set_kernel(kern)
K = 12
x0 = p.x0_p * (1.0 + 0.02 * np.random.default_rng(1).uniform(-1, 1, size=(B, 3)))
s.alloc(B, K); s.set_schedule(p.schedules(K)); s.set_state(x0, x0)
names = (["est+target", "init", "factor+rhs", "forward(x2)", "predictor ew", "rhs", "corrector ew + test", "linearise"] if kern in (3, 4) else
         ["init sweep", "B1", "F1", "B2", "F2", "est+target", "linearise", "-"])
buf = np.zeros(64 * 8, np.uint64)
s.lib.nmpc_debug_stamps(None, 0, 1)
for k in range(K):
    s.run(k, 1, 1, 1e-9); s.sync()
    s.lib.nmpc_debug_stamps(buf.ctypes.data_as(ct.c_void_p), 64 * 8, 1)
    c = buf.reshape(64, 8).astype(float)
    w = np.argmax(c.sum(axis=1))
    print(f"step {k:2d} kernel {s.last_kernel_ms():.3f} ms | slowest of the first 64 waves, kticks: " + " ".join(f"{n}={c[w, i]/1e3:.0f}" for i, n in enumerate(names)) + f" | total {c[w].sum()/1e3:.0f}")
it = s.get_log("ITERS_DYN"); print("iters per step (max over batch):", it.max(axis=1))
