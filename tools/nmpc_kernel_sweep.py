#!/usr/bin/env python3
"""Non-linear path: closed-loop steps/s of each kernel against the batch size (20 steps from t = 0, Ex_NMPC, N = 30):
   tools/nmpc_kernel_sweep.py [out.json]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import nmpc
p = m.load_problem(m.example_path("cstr_nmpc.py"))
s = nmpc.NmpcSolver(p)
K = 20
rows = []
for B in (1024, 4096, 16384, 32768, 65536, 131072):
    x0 = p.x0_p * (1.0 + 0.02 * np.random.default_rng(20250614).uniform(-1, 1, size=(B, 3)))
    s.alloc(B, K); s.set_schedule(p.schedules(K))
    row = {"batch": B}
    for kern in (1, 3, 4):
        s.set_kernel(kern)
        wall = []
        for r in range(4):
            s.set_state(x0, x0); s.sync()
            t0 = time.perf_counter(); s.run(0, K, 1); s.sync(); wall.append(time.perf_counter() - t0)
        row[f"kernel{kern}_ms"] = float(np.median(wall) * 1e3); row[f"kernel{kern}_Msteps_s"] = B * K / float(np.median(wall)) / 1e6
    s.set_kernel(0); row["auto"] = s.get_kernel()
    rows.append(row)
    print(row, flush=True)
if len(sys.argv) > 1:
    json.dump({"workload": "Ex_NMPC, N = 30, one real-time SQP iteration per step, 20 closed-loop steps from t = 0", "kernels": {"1": "instance per lane (helper waves up to 2 workgroups per CU)", "3": "wave-autonomous, all steps in one launch", "4": "split pipeline"}, "rows": rows}, open(sys.argv[1], "w"), indent=1)
