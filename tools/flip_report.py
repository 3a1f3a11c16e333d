#!/usr/bin/env python3
"""Diagnostic (GPU box): which instance-steps of the randomised closed loops (tests/test_gpu_fuzz.py) and of the N = 64 loop get a status word
different from the C restatement's, and why.   tools/flip_report.py"""
import copy, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpc_code_amd as m
from mpc_code_amd import capi
from mpc_code_amd.driver import run_closed_loop
import oracle_c
from test_gpu_fuzz import random_problem, CASES
from conftest import bench_x0

def report(tag, p, g, c):
    same_d = g["STATUS_DYN"] == c["STATUS_DYN"]; same_s = g["STATUS_SS"] == c["STATUS_SS"]
    bad = np.where(~(same_d.all(axis=0) & same_s.all(axis=0)))[0]
    print(tag, "instances with a different status word:", len(bad), "of", same_d.shape[1])
    for b in bad[:6]:
        k = int(np.argmin(same_d[:, b] & same_s[:, b]))
        dh = g["D_HAT"][k, b] if p.nd else np.zeros(0)
        y0 = p.C @ g["X_HAT"][k, b] + p.fy_const + (p.Cd @ dh if p.nd else 0.0)
        with np.errstate(invalid="ignore"):
            margin = min(np.abs(y0 - p.ymin).min(), np.abs(y0 - p.ymax).min())
        print("   inst", b, "step", k, "dyn gpu/c", int(g["STATUS_DYN"][k, b]), int(c["STATUS_DYN"][k, b]), "ss gpu/c", int(g["STATUS_SS"][k, b]), int(c["STATUS_SS"][k, b]),
              "iters gpu/c", int(g["ITERS_DYN"][k, b]), int(c["ITERS_DYN"][k, b]), "stage-0 output margin %.2e" % margin, "max |dU| before: %.1e" % (np.abs(g["U"][:k, b] - c["U"][:k, b]).max() if k else 0.0))

for seed, nx, nu, ny, du in CASES[::2]:
    p = random_problem(seed, nx, nu, ny, du)
    rng = np.random.default_rng(seed + 11)
    B, K = 96, 15
    scale = np.where(np.isfinite(p.xmax), p.xmax, 3.0)
    x0 = rng.uniform(-0.5, 0.5, (B, nx)) * scale
    c = oracle_c.OracleC(p).closed_loop(K, x0, x0)
    for lk in (1, 2, 3):
        s = capi.Solver(p); s.set_option("loop_kernel", lk)
        g = run_closed_loop(p, x0, x0, K, solver=s)
        report(f"seed {seed} kernel {lk}", p, g, c)
        s.close()
cstr = m.load_problem(m.example_path("cstr_lmpc.py"))
q = copy.copy(cstr); q.N = 64
x0 = bench_x0(40, 77)
c = oracle_c.OracleC(q).closed_loop(5, x0, x0)
for lk in (1, 2, 3):
    s = capi.Solver(q); s.set_option("loop_kernel", lk)
    g = run_closed_loop(q, x0, x0, 5, solver=s)
    report(f"cstr N=64 kernel {lk}", q, g, c)
    s.close()
