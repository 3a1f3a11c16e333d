#!/usr/bin/env python3
"""Throughput of the three ways to drive the same closed loop: resident state (bench.py), host buffers
per run (mpc_closed_loop: one upload + one download), host buffers per solver call (the literal three-call drop-in)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import capi
from mpc_code_amd.driver import run_closed_loop
p = m.load_problem(m.example_path("cstr_lmpc.py"))
B, K = 4096, 100
x0 = np.random.default_rng(20250614).uniform([-0.5, -8, -5], [0.5, 8, 5], size=(B, 3))
s = capi.Solver(p)
s.loop_alloc(B, K, capi.LOG_U); s.loop_set_schedule(p.schedules(K)); s.loop_set_state(x0, x0)
t0 = time.perf_counter(); s.loop_run(0, K); s.loop_sync(); t_res = time.perf_counter() - t0
t0 = time.perf_counter(); out = run_closed_loop(p, x0, x0, K, solver=s, fused=True); t_host = time.perf_counter() - t0
t0 = time.perf_counter(); out2 = run_closed_loop(p, x0, x0, 20, solver=s, fused=False); t_step = (time.perf_counter() - t0) * K / 20
print(f"resident state          : {B*K/t_res:12.0f} steps/s")
print(f"host buffers, fused run : {B*K/t_host:12.0f} steps/s (upload, {K} launches, all logs downloaded)")
print(f"host buffers, per call  : {B*K/t_step:12.0f} steps/s (3 C-ABI calls per step, cold-started OCPs)")

# ---- the non-linear and the economic path: resident state (what bench.py times) against the host-buffer call of their drivers (alloc + upload of the start
#      states + the same launches + download of every log the reference's result arrays need), bench.py's workloads at 20 steps
import warnings
warnings.simplefilter("ignore")
sys.path.insert(0, ROOT)
import bench as _bench
from mpc_code_amd import enmpc, nmpc
K = 20
for name in ("enmpc", "mhe"):
    cfg = _bench.ENMPC_CONFIGS[name]
    q = m.load_problem(m.example_path("reactor_enmpc.py"), overrides=cfg["over"])
    B = cfg["batch"]
    x0 = np.random.default_rng(20250614).uniform([0.5, 0.0], [1.0, 0.5], size=(B, 2))
    es = enmpc.EnmpcSolver(q)
    es.alloc(B, K); es.set_state(x0); es.run(0, K); es.sync()                                        # warm-up (first launch, clocks)
    t_res = t_host = 1e9
    for _ in range(5):                                                                               # (the best of five: clocks ramp up over the first runs)
        es.set_state(x0); t0 = time.perf_counter(); es.run(0, K); es.sync(); t_res = min(t_res, time.perf_counter() - t0)
    for _ in range(5):
        t0 = time.perf_counter(); out = enmpc.run_enmpc_closed_loop(q, x0, K, solver=es); t_host = min(t_host, time.perf_counter() - t0)
    mb = sum(v.nbytes for v in out.values() if isinstance(v, np.ndarray)) / 1e6
    print(f"{name:5s} resident state     : {B*K/t_res:12.0f} steps/s")
    print(f"{name:5s} host buffers, a run: {B*K/t_host:12.0f} steps/s (alloc, upload, {K} steps, {mb:.0f} MB of result arrays downloaded / derived)")
    es.close()
q = m.load_problem(m.example_path("cstr_nmpc.py"))
B = 16384
x0 = q.x0_p * (1.0 + 0.02 * np.random.default_rng(20250614).uniform(-1.0, 1.0, size=(B, 3)))
ns = nmpc.NmpcSolver(q)
ns.alloc(B, K); ns.set_schedule(q.schedules(K)); ns.set_state(x0, x0); ns.run(0, K, 1); ns.sync()
t_res = t_host = 1e9
for _ in range(5):
    ns.set_state(x0, x0); t0 = time.perf_counter(); ns.run(0, K, 1); ns.sync(); t_res = min(t_res, time.perf_counter() - t0)
for _ in range(5):
    t0 = time.perf_counter(); out = nmpc.run_nmpc_closed_loop(q, x0, x0, K, solver=ns); t_host = min(t_host, time.perf_counter() - t0)
mb = sum(v.nbytes for v in out.values() if isinstance(v, np.ndarray)) / 1e6
print(f"nmpc  resident state     : {B*K/t_res:12.0f} steps/s")
print(f"nmpc  host buffers, a run: {B*K/t_host:12.0f} steps/s (alloc, upload, {K} steps, {mb:.0f} MB of result arrays downloaded / derived)")
