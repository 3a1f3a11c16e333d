#!/usr/bin/env python3
"""Throughput against batch size on one GPU, every closed-loop kernel (kernel time by HIP events, median of the repeats):
linear benchmark problem (Ex_LMPC_CSTR, N = 50) for K = 20 and K = 100 steps from t = 0, Wood-Berry, and the non-linear workload
(Ex_NMPC, N = 30, one real-time iteration per step).  Writes gpurun_out/batch_sweep.json."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import capi, nmpc

out = {"lmpc_cstr": [], "lmpc_wb": [], "nmpc_cstr": []}
# (the non-linear workload first: after the linear sweeps' allocations and frees, round 4's first recordings of its 16384 row came out at 27 ms against 16.6 ms in a fresh process)
p = m.load_problem(m.example_path("cstr_nmpc.py"))
s = nmpc.NmpcSolver(p)
rng = np.random.default_rng(20250615)      # (starts of its own: continuing the linear part's stream, round 4's first recording drew a 16384 batch whose slowest wave took 27 ms against the 16.6 ms of every other draw)
for B in (1024, 4096, 16384, 65536, 131072):
    x0 = p.x0_p * (1.0 + 0.02 * rng.uniform(-1, 1, size=(B, 3)))
    K = 20
    s.alloc(B, K); s.set_schedule(p.schedules(K))
    ts = []
    for r in range(6):      # (the first run after a re-allocation or a change of launch style is discarded; the median of the rest - two were too few: one slow run of three made round 4's first recording of the 16384 row wrong)
        s.set_state(x0, x0); s.run(0, K, 1); s.sync(); ts.append(s.last_kernel_ms())
    ms = float(np.median(ts[1:]))
    out["nmpc_cstr"].append(dict(B=B, K=K, kernel_ms=ms, msteps_per_s=B * K / ms / 1e3, runs_ms=[round(t, 2) for t in ts], kernel=s.get_kernel()))
    print("nmpc", out["nmpc_cstr"][-1], flush=True)
s.close()
rng = np.random.default_rng(20250614)
for ex, key, Bs in (("cstr_lmpc.py", "lmpc_cstr", (256, 1024, 4096, 16384, 65536)), ("wood_berry_lmpc.py", "lmpc_wb", (1024, 4096, 16384))):
    p = m.load_problem(m.example_path(ex))
    s = capi.Solver(p)
    for B in Bs:
        x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3)) if p.nx == 3 else 0.05 * rng.standard_normal((B, p.nx))
        for K in (20, 100):
            for lk in (3, 2, 1):
                if lk == 1 and B * K > 4096 * 100 * 4:
                    continue
                try:
                    s.set_option("loop_kernel", lk)
                except capi.MpcAmdError:
                    continue
                s.set_option("steps_per_launch", K)
                s.loop_alloc(B, K, capi.LOG_U); s.loop_set_schedule(p.schedules(K))
                ts = []
                for r in range(4):
                    s.loop_set_state(x0, x0); s.loop_run(0, K); s.loop_sync(); ts.append(s.last_kernel_ms()[0])
                ms = float(np.median(ts[1:]))
                out[key].append(dict(B=B, K=K, loop_kernel=lk, kernel_ms=ms, msteps_per_s=B * K / ms / 1e3))
                print(key, out[key][-1], flush=True)
    s.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "batch_sweep.json"), "w"), indent=1)
