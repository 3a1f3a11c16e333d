import os, sys, copy, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpc_code_amd as m
from mpc_code_amd import capi
from mpc_code_amd.driver import run_closed_loop
from conftest import bench_x0
cstr = m.load_problem(m.example_path("cstr_lmpc.py"))
pc = copy.copy(cstr); pc.Dumin = np.array([-0.5, -1.0]); pc.Dumax = np.array([0.5, 1.0])
x0 = bench_x0(64, 8) * [1.0, 0.3, 0.6]
out = {}
for lk in (1, 2, 3):
    s = capi.Solver(pc); s.set_option("loop_kernel", lk)
    g = run_closed_loop(pc, x0, x0, 6, solver=s); s.close()
    out[lk] = g
    print(lk, "iters step0", g["ITERS_DYN"][0][:16], "status", np.bincount(g["STATUS_DYN"].ravel(), minlength=3), "max iters", g["ITERS_DYN"].max())
for lk in (2, 3):
    print(lk, "max |U - U1|", np.abs(out[lk]["U"] - out[1]["U"]).max(), "iters eq", (out[lk]["ITERS_DYN"] == out[1]["ITERS_DYN"]).mean())
