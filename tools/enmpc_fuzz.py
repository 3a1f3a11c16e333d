#!/usr/bin/env python3
"""Diagnostic (GPU box): randomised reactor models (constants, sampling time, horizons, estimator update) on both launch styles against
oracle/enmpc_oracle.c - the loop of tests/test_enmpc.py::test_gpu_randomised_reactor_models_follow_the_c_restatement over many seeds.
   tools/enmpc_fuzz.py [first seed] [count]"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.simplefilter("ignore")
import mpc_code_amd as m
from mpc_code_amd import enmpc
import enmpc_oracle as eo, enmpc_oracle_c as ec
from enmpc_cases import draw
EX = m.example_path("reactor_enmpc.py")
s0, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 5), (int(sys.argv[2]) if len(sys.argv) > 2 else 16)
bad = 0
for seed in range(s0, s0 + n):
    over, x0 = draw(seed)
    c = ec.OracleEC(eo.load_problem(EX, overrides=over)).closed_loop(12, x0, nthreads=0)
    p = m.load_problem(EX, overrides=over)
    s = enmpc.EnmpcSolver(p)
    msg = []
    for kernel in (1, 2):
        r = enmpc.run_enmpc_closed_loop(p, x0, 12, solver=s, kernel=kernel)
        dv = max(float(np.abs(r[k] - c[k]).max()) for k in ("U", "XS", "US", "X_ES", "Xp"))
        st = all(np.array_equal(r[k], c[k]) for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE"))
        di = max(int(np.abs(r[k].astype(int) - c[k].astype(int)).max()) for k in ("ITERS_DYN", "ITERS_SS", "ITERS_MHE"))
        ok = dv < 1e-7 and st and di <= 4
        bad += not ok
        where = [f"{k[7:]} step {a} instance {b}: {int(c[k][a, b])} here, {int(r[k][a, b])} on the GPU" for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE") for a, b in zip(*np.nonzero(r[k] != c[k]))]
        msg.append(f"k{kernel}: |dv| {dv:.1e} status {'=' if st else 'DIFFER (' + ', '.join(where) + ')'} iters +-{di}{'' if ok else '  <-- FAIL'}")
    s.close()
    print(seed, {k: (round(v, 3) if isinstance(v, float) else (np.round(v, 3).tolist() if isinstance(v, (list, np.ndarray)) else v)) for k, v in over.items()}, "| status max", int(c["STATUS_DYN"].max()), int(c["STATUS_SS"].max()), int(c["STATUS_MHE"].max()), "|", " ; ".join(msg), flush=True)
print("failures:", bad)
