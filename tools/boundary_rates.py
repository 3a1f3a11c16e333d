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
