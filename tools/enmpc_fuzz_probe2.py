#!/usr/bin/env python3
"""Development probe (GPU box): the first step of a randomised model through the per-call seam, estimator with 64 and with 32 / 16 lanes per instance; what the target sees.
   ENMPC_NO_SELFTEST=1 tools/enmpc_fuzz_probe2.py seed"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.simplefilter("ignore")
import mpc_code_amd as m
from mpc_code_amd import enmpc
from mpc_code_amd.enmpc import _rows
from enmpc_cases import draw
seed = int(sys.argv[1])
over, x0 = draw(seed)
p = m.load_problem(m.example_path("reactor_enmpc.py"), overrides=over)
print("seed", seed, over, flush=True)
np.set_printoptions(linewidth=220, precision=6)
B = len(x0)
s = enmpc.EnmpcSolver(p)
res = {}
for kern in (64, 32, 16, 64):
    s.set_kernel(kern)
    s.alloc(B, 1); s.set_state(x0)
    u = _rows(p.u0, B, p.nu)
    xhat, dhat, xes, st_m, it_m = s.mhe_update(x0, u)
    xs, us, st_s, it_s = s.target_solve(dhat)
    xs2, us2, st_s2, it_s2 = s.target_solve(dhat)
    print("kernel", kern, "\n xhat", xhat.ravel(), "\n dhat", dhat.ravel(), "\n xes", xes.ravel(), "\n it_m", it_m, "\n xs", xs.ravel(), "us", us.ravel(), "it_s", it_s, st_s, "\n again: xs", xs2.ravel(), "us", us2.ravel(), "it_s", it_s2, flush=True)
    # the target alone on a fresh handle state (no estimator launch before it)
    s.alloc(B, 1); s.set_state(x0)
    xs3, us3, st_s3, it_s3 = s.target_solve(dhat)
    print(" target without an estimator launch before: xs", xs3.ravel(), "us", us3.ravel(), "it_s", it_s3, flush=True)
s.close()
