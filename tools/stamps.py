#!/usr/bin/env python3
"""Diagnostic: shader cycles per sweep of the OCP solver (needs a library built with -DMPC_STAMPS)."""
import ctypes as ct, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPC_AMD_LIB"] = sys.argv[1]
import mpc_code_amd as m
from mpc_code_amd import capi
p = m.load_problem(m.example_path("cstr_lmpc.py"))
s = capi.Solver(p)
B, K = 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 20
KM = 16      # after the K single-step launches: one launch of KM steps (steady state without per-launch cold misses)
x0 = np.random.default_rng(20250614).uniform([-0.5, -8, -5], [0.5, 8, 5], size=(B, 3))      # bench.py's box
s.loop_alloc(B, K + 40 + KM, capi.LOG_U); s.loop_set_schedule(p.schedules(K + 40 + KM)); s.loop_set_state(x0, x0)
buf = np.zeros(64 * 8, np.uint64)
s.lib.mpc_debug_stamps(None, 0, 1)
s.set_option("steps_per_launch", 1)
MODE = int(sys.argv[3]) if len(sys.argv) > 3 else 1
s.set_option("loop_kernel", MODE)
names = ["init", "B1", "F1", "B2", "F2"] if MODE == 1 else (["target", "init", "A", "B+C", "D+E", "F+G", "H+A", "est"] if MODE == 2 else
         ["est+target", "init", "factor+rhs", "forward(x2)", "predictor ew", "rhs", "corrector ew + test", "accept+plant"])
for k in range(K):
    s.loop_run(k, 1); s.loop_sync()
    s.lib.mpc_debug_stamps(buf.ctypes.data_as(ct.c_void_p), 64 * 8, 1)
    c = buf.reshape(64, 8)[:, :len(names)].astype(float)
    it = s.loop_get_log("ITERS_DYN")[k]
    ms, _ = s.last_kernel_ms()
    w = np.argmax(c.sum(axis=1))
    print(f"step {k:2d} iters max {it.max():2d} kernel {ms:.3f} ms | slowest wave kcycles: " + " ".join(f"{n}={c[w, i]/1e3:.0f}" for i, n in enumerate(names)) + f" | total {c[w].sum()/1e3:.0f}")

s.set_option("steps_per_launch", KM)
s.set_option("steps_per_launch", 40); s.loop_run(K, 40); s.loop_sync()   # transients of the set-point change pass
s.lib.mpc_debug_stamps(None, 0, 1); s.set_option("steps_per_launch", KM)
s.loop_run(K + 40, KM); s.loop_sync()
s.lib.mpc_debug_stamps(buf.ctypes.data_as(ct.c_void_p), 64 * 8, 1)
c = buf.reshape(64, 8)[:, :len(names)].astype(float) / KM
ms, _ = s.last_kernel_ms()
w = np.argmax(c.sum(axis=1))
print(f"one launch of {KM} steps: {ms/KM:.3f} ms per step | slowest wave kcycles per step: " + " ".join(f"{n}={c[w, i]/1e3:.0f}" for i, n in enumerate(names)) + f" | total {c[w].sum()/1e3:.0f}")
