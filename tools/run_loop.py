#!/usr/bin/env python3
"""Run the fused closed-loop kernel a few times (profiling target for rocprofv3).

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/run_loop.py --batch 4096 --steps 10
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m                   # noqa: E402
from mpc_code_amd import capi              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--example", default="cstr_lmpc.py")
ap.add_argument("--steps-per-launch", type=int, default=0)
ap.add_argument("--loop-kernel", type=int, default=0, help="0 auto, 1 instance per lane, 2 horizon-parallel")
ap.add_argument("--warmup", type=int, default=0, help="untimed steps first (the first launch of a kernel pays for code upload and scratch allocation), then the state is reset")
a = ap.parse_args()
p = m.load_problem(m.example_path(a.example))
rng = np.random.default_rng(20250614)
if p.nx == 3:
    x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(a.batch, 3))
else:
    x0 = 0.05 * rng.standard_normal((a.batch, p.nx))
s = capi.Solver(p)
if a.steps_per_launch > 0:
    s.set_option("steps_per_launch", a.steps_per_launch)
s.set_option("loop_kernel", a.loop_kernel)
s.loop_alloc(a.batch, a.steps, capi.LOG_U)
s.loop_set_schedule(p.schedules(a.steps))
s.loop_set_state(x0, x0)
if a.warmup > 0:
    s.loop_run(0, min(a.warmup, a.steps)); s.loop_sync()
    s.loop_set_state(x0, x0)
t0 = time.perf_counter()
s.loop_run(0, a.steps)
s.loop_sync()
dt = time.perf_counter() - t0
ms, n = s.last_kernel_ms()
it = s.loop_get_log("ITERS_DYN"); st = s.loop_get_log("STATUS_DYN")
print(f"kernel={a.loop_kernel} B={a.batch} steps={a.steps}: wall {dt*1e3:.2f} ms, kernels {ms:.2f} ms / {n} launches = {ms/n:.3f} ms per launch, "
      f"{a.batch*a.steps/dt:.0f} steps/s, iters mean {it[st!=2].mean():.2f} max {it.max()}, infeasible {np.mean(st==2):.3f}")
s.close()
