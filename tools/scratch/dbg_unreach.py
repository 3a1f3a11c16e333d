import sys, os, warnings
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import mpc_code_amd as pkg
from mpc_code_amd import enmpc
import enmpc_oracle as eo, enmpc_oracle_c as ec
EX = pkg.example_path("reactor_enmpc.py")
over = {"xmin": np.array([0.8, 0.8]), "N": 12}
x0 = np.array([[0.9, 0.1], [0.6, 0.3], [0.7, 0.2]])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    p = pkg.load_problem(EX, overrides=over)
    c = ec.OracleEC(eo.load_problem(EX, overrides=over)).closed_loop(5, x0, nthreads=3)
print("C   SS", c["STATUS_SS"].T.tolist(), c["ITERS_SS"].T.tolist())
s = enmpc.EnmpcSolver(p)
for kernel in (1, 2, 1, 2):
    r = enmpc.run_enmpc_closed_loop(p, x0, 5, solver=s, kernel=kernel)
    print("k", kernel, "SS", r["STATUS_SS"].T.tolist(), r["ITERS_SS"].T.tolist(), "DYN", r["STATUS_DYN"].T.tolist(), r["ITERS_DYN"].T.tolist(), "maxdiff XS", np.abs(r["XS"]-c["XS"]).max())
print("C  DYN", c["STATUS_DYN"].T.tolist(), c["ITERS_DYN"].T.tolist())
