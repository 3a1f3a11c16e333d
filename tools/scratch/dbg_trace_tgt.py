import sys, os, warnings
sys.path.insert(0, "/root/repo")
import numpy as np
import mpc_code_amd as pkg
from mpc_code_amd import enmpc, econcodegen
EX = pkg.example_path("reactor_enmpc.py")
over = {"xmin": np.array([0.8, 0.8]), "N": 12}
x0 = np.array([[0.9, 0.1], [0.6, 0.3], [0.7, 0.2]])
warnings.simplefilter("ignore")
p = pkg.load_problem(EX, overrides=over)
lib = econcodegen.build_enmpc_library(p, extra_flags=["-DEC_TRACE_TGT"])
s = enmpc.EnmpcSolver(p, lib_path=lib)
r = enmpc.run_enmpc_closed_loop(p, x0, 1, solver=s, kernel=2)
print("SS", r["STATUS_SS"].T.tolist(), r["ITERS_SS"].T.tolist())
