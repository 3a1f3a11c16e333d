#!/usr/bin/env python3
"""Measurement: the headline workload (LMPC-CSTR, N = 50, 20 steps from a cold start) with one, two and four instances per wave of the wave-autonomous kernel (option
"wave_instances"; the kernel holds 512 registers in every form: one wave per SIMD), over the batch size.   tools/wave_instances.py > profiles/rNN_wave_instances.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import capi
from mpc_code_amd.driver import run_closed_loop
p = m.load_problem(m.example_path("cstr_lmpc.py"))
s = capi.Solver(p)
s.set_option("steps_per_launch", 20)
print("# LMPC-CSTR, 20 closed-loop steps from t = 0, one launch; M steps/s (best of five), waves launched")
for B in (1024, 2048, 4096, 8192, 16384):
    x0 = np.random.default_rng(20250614).uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3))
    row, ref = [], None
    for ni in (1, 2, 4):
        s.set_option("wave_instances", ni)
        best = None
        for _ in range(5):
            r = run_closed_loop(p, x0, x0, 20, solver=s)
            ms, _n = s.last_kernel_ms()
            best = ms if best is None else min(best, ms)
        same = True if ref is None else bool(np.array_equal(ref["STATUS_DYN"], r["STATUS_DYN"]) and np.abs(ref["U"] - r["U"]).max() < 1e-9)
        ref = ref or r
        row.append(f"{ni} per wave: {B * 20 / best / 1e3:6.2f} M ({(B + ni - 1) // ni} waves{'' if same else ', DIFFERS'})")
    print(f"batch {B:6d}:  " + "   ".join(row), flush=True)
s.close()
