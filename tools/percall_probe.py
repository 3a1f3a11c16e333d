import os, sys, time, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mpc_code_amd as m
from mpc_code_amd import capi
from mpc_code_amd.driver import run_closed_loop
from conftest import bench_x0
import oracle_c
p = m.load_problem(m.example_path("cstr_lmpc.py"))
B, K = 512, 30
x0 = bench_x0(B, 3)
s = capi.Solver(p)
f = run_closed_loop(p, x0, x0, K, solver=s, fused=True)
for ws in (False, True):
    t0 = time.time(); g = run_closed_loop(p, x0, x0, K, solver=s, fused=False, warm_start=ws); dt = time.time() - t0
    print("warm" if ws else "cold", "status eq", (g["STATUS_DYN"] == f["STATUS_DYN"]).mean(), "max|U-Ufused|", np.abs(g["U"] - f["U"]).max(), "mean iters", g["ITERS_DYN"].mean(), "(fused", f["ITERS_DYN"].mean(), ")", "%.0f steps/s" % (B * K / dt))
# per call vs C oracle, lane vs wave
rng = np.random.default_rng(1)
xh = bench_x0(1000, 5); xs = np.zeros((1000, 3)); us = np.zeros((1000, 2)); d = 0.02 * rng.standard_normal((1000, 3)); up = rng.uniform(-1, 1, (1000, 2))
c = oracle_c.OracleC(p).ocp_solve(xh, xs, us, d, up, want_w=True)
for ok in (1, 3):
    s.set_option("ocp_kernel", ok); s.set_option("ocp_warm_start", 0)
    g = s.ocp_solve(xh, xs, us, d, up, want_w=True)
    good = c["status"] == 0
    print("ocp_kernel", ok, "status eq", (g["status"] == c["status"]).mean(), "u0 err", np.abs(g["u0"] - c["u0"])[good].max(), "w err", np.abs(g["w"] - c["w"])[good].max(), "iters eq", (g["iters"] == c["iters"]).mean(), "res", np.abs(g["res"][good]).max(axis=0))
B = 4096; x0 = bench_x0(B); K = 100
for ws in (False, True):
    t0 = time.time(); g = run_closed_loop(p, x0, x0, K, solver=s, fused=False, warm_start=ws); dt = time.time() - t0
    print("three calls per step, B=4096 K=100", "warm" if ws else "cold", "%.3f M steps/s" % (B * K / dt / 1e6), "mean iters", g["ITERS_DYN"].mean())
s.close()
