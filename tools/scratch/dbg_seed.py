import sys, os, warnings
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import mpc_code_amd as pkg
from mpc_code_amd import enmpc, econcodegen
from enmpc_cases import draw, FUZZ_STEPS
seed = int(sys.argv[1]); build_only = len(sys.argv) > 2
EX = pkg.example_path("reactor_enmpc.py")
over, x0 = draw(seed)
warnings.simplefilter("ignore")
p = pkg.load_problem(EX, overrides=over)
if build_only:
    print(econcodegen.build_enmpc_library(p)); sys.exit(0)
import enmpc_oracle as eo, enmpc_oracle_c as ec
q = eo.load_problem(EX, overrides=over)
c = ec.OracleEC(q).closed_loop(FUZZ_STEPS, x0, nthreads=0)
s = enmpc.EnmpcSolver(p)
print(over)
for kernel in (1, 2):
    r = enmpc.run_enmpc_closed_loop(p, x0, FUZZ_STEPS, solver=s, kernel=kernel)
    for k in ("ITERS_DYN", "ITERS_SS", "ITERS_MHE"):
        d = r[k].astype(int) - c[k].astype(int)
        idx = np.argwhere(d != 0)
        print("kernel", kernel, k, "mismatches", len(idx), [(int(a), int(b), int(r[k][a, b]), int(c[k][a, b])) for a, b in idx[:12]])
    for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE"):
        print("   ", k, "equal", np.array_equal(r[k], c[k]), "max", int(r[k].max()))
    for k in ("U", "XS", "US", "X_ES", "Xp"):
        print("   ", k, float(np.abs(r[k] - c[k]).max()))
