#!/usr/bin/env python3
"""Closed loop with each loop kernel (1 = instance per lane, 2 = horizon-parallel, 3 = wave-autonomous) against oracle/mpc_oracle.c (development check)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mpc_code_amd as m                   # noqa: E402
from mpc_code_amd import capi              # noqa: E402
import oracle_c                            # noqa: E402

for ex, B, K in (("cstr_lmpc.py", 1000, 30), ("wood_berry_lmpc.py", 200, 20)):
    p = m.load_problem(m.example_path(ex))
    rng = np.random.default_rng(7)
    x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3)) if p.nx == 3 else 0.05 * rng.standard_normal((B, p.nx))
    ref = oracle_c.OracleC(p).closed_loop(K, x0, x0)
    for mode in (3, 2, 1):
        s = capi.Solver(p)
        try:
            s.set_option("loop_kernel", mode)
        except capi.MpcAmdError as e:
            print(f"{ex} kernel={mode}: not available ({e})"); s.close(); continue
        s.loop_alloc(B, K, capi.LOG_ALL)
        s.loop_set_schedule(p.schedules(K)); s.loop_set_state(x0, x0)
        t0 = time.perf_counter(); s.loop_run(0, K); s.loop_sync(); dt = time.perf_counter() - t0
        U = s.loop_get_log("U"); st = s.loop_get_log("STATUS_DYN"); it = s.loop_get_log("ITERS_DYN")
        print(f"{ex} kernel={mode}: {dt*1e3:.1f} ms, max|U-Uc| {np.nanmax(np.abs(U-ref['U'])):.2e}, nan {np.isnan(U).sum()}, status equal {np.mean(st==ref['STATUS_DYN']):.4f}, "
              f"iters equal {np.mean(it==ref['ITERS_DYN']):.4f}, mean iters {it.mean():.2f} (C {ref['ITERS_DYN'].mean():.2f})", flush=True)
        for nm in ("X_HAT", "XS", "US", "Xp", "D_HAT", "YS"):
            d = np.abs(s.loop_get_log(nm) - ref[nm]).max()
            if d > 1e-7: print(f"    {nm}: max diff {d:.2e}")
        s.close()
