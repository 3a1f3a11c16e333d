#!/usr/bin/env python3
"""Development probe (GPU box): one randomised reactor model of tests/enmpc_cases.py, every launch style, against oracle/enmpc_oracle.c - where (step, NLP) they part.
   ENMPC_NO_SELFTEST=1 tools/enmpc_fuzz_probe.py seed [steps]"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.simplefilter("ignore")
import mpc_code_amd as m
from mpc_code_amd import enmpc
import enmpc_oracle as eo, enmpc_oracle_c as ec
from enmpc_cases import draw
seed, nsteps = int(sys.argv[1]), (int(sys.argv[2]) if len(sys.argv) > 2 else 3)
over, x0 = draw(seed)
if len(sys.argv) > 3:
    over.update({"N": int(sys.argv[3]), "N_mhe": int(sys.argv[4])})
path = m.example_path("reactor_enmpc.py")
p = m.load_problem(path, overrides=over)
c = ec.OracleEC(eo.load_problem(path, overrides=over)).closed_loop(nsteps, x0, nthreads=0)
print("seed", seed, over, flush=True)
s = enmpc.EnmpcSolver(p)
np.set_printoptions(linewidth=220, precision=3)
for kernel in (1, 2, 64, 32, 16):
    try:
        r = enmpc.run_enmpc_closed_loop(p, x0, nsteps, solver=s, kernel=kernel)
    except Exception as e:
        print("kernel", kernel, "refused:", str(e)[:100]); continue
    print("kernel", kernel)
    for k in ("X_ES", "XS", "US", "U"):
        print("   ", k, "max |diff| per step", np.abs(r[k] - c[k]).reshape(nsteps, -1).max(axis=1))
    for k in ("ITERS_MHE", "ITERS_SS", "ITERS_DYN", "STATUS_MHE", "STATUS_SS", "STATUS_DYN"):
        print("   ", k, "gpu", r[k].reshape(nsteps, -1)[:2].tolist(), "c", c[k].reshape(nsteps, -1)[:2].tolist())
s.close()
