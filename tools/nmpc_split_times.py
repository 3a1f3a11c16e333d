#!/usr/bin/env python3
"""Split pipeline of the non-linear path: wall time, device time (first to last event) and the wave-style launches' share, with and
without the per-launch events:  tools/nmpc_split_times.py [batch] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import nmpc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = m.load_problem(m.example_path("cstr_nmpc.py"))
s = nmpc.NmpcSolver(p)
x0 = p.x0_p * (1.0 + 0.02 * np.random.default_rng(20250614).uniform(-1, 1, size=(B, 3)))
s.alloc(B, K); s.set_schedule(p.schedules(K))
for timed in (False, True, False):
    s.time_kernels(timed)
    wall, dev, wv = [], [], []
    for r in range(12):
        s.set_state(x0, x0); s.sync()
        t0 = time.perf_counter(); s.run(0, K, 1); s.sync(); wall.append(time.perf_counter() - t0)
        dev.append(s.last_kernel_ms()); wv.append(s.wave_kernel_ms()[0])
    print(f"per-launch events {'on ' if timed else 'off'}: wall {np.median(wall)*1e3:.3f} ms, device {np.median(dev):.3f} ms, wave-style launches {np.median(wv):.3f} ms "
          f"-> {B*K/np.median(wall)/1e6:.2f} M steps/s   (wall min {np.min(wall)*1e3:.2f} max {np.max(wall)*1e3:.2f})")
