#!/usr/bin/env python3
"""Development check of one closed-loop kernel on the CSTR benchmark problem: parity with oracle/mpc_oracle.c, then timing.

    MPC_AMD_LIB=build_diag/libmpc_one.so python3 tools/wv_check.py [loop_kernel=3]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mpc_code_amd as m                   # noqa: E402
from mpc_code_amd import capi              # noqa: E402
import oracle_c                            # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = m.load_problem(m.example_path("cstr_lmpc.py"))
B, K = 1001, 30
x0 = np.random.default_rng(7).uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3))
ref = oracle_c.OracleC(p).closed_loop(K, x0, x0)
s = capi.Solver(p); s.set_option("loop_kernel", mode); s.set_option("steps_per_launch", 7)
s.loop_alloc(B, K, capi.LOG_ALL); s.loop_set_schedule(p.schedules(K)); s.loop_set_state(x0, x0)
s.loop_run(0, K); s.loop_sync()
U = s.loop_get_log("U"); st = s.loop_get_log("STATUS_DYN"); it = s.loop_get_log("ITERS_DYN")
print(f"kernel={mode}: max|U-Uc| {np.nanmax(np.abs(U-ref['U'])):.2e}, nan {np.isnan(U).sum()}, status equal {np.mean(st==ref['STATUS_DYN']):.4f}, "
      f"iters equal {np.mean(it==ref['ITERS_DYN']):.4f}, mean iters {it.mean():.2f} (C {ref['ITERS_DYN'].mean():.2f})", flush=True)
for nm in ("X_HAT", "XS", "US", "Xp", "D_HAT", "YS"):
    d = np.abs(s.loop_get_log(nm) - ref[nm]).max()
    if d > 1e-7: print(f"    {nm}: max diff {d:.2e}")
s.close()
x0 = np.random.default_rng(20250614).uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(4096, 3))
for K in (20, 100):
    s = capi.Solver(p); s.set_option("loop_kernel", mode)
    s.loop_alloc(4096, K, capi.LOG_U); s.loop_set_schedule(p.schedules(K)); s.loop_set_state(x0, x0)
    s.loop_run(0, min(K, 2)); s.loop_sync(); s.loop_set_state(x0, x0)
    t0 = time.perf_counter(); s.loop_run(0, K); s.loop_sync(); dt = time.perf_counter() - t0
    print(f"B=4096 K={K}: {dt*1e3:.2f} ms, {4096*K/dt/1e6:.2f} M steps/s")
    s.close()
