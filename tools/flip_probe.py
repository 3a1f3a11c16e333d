import os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpc_code_amd as m
from mpc_code_amd import capi
from mpc_code_amd.driver import run_closed_loop
import oracle_c
GOLD = os.path.join(ROOT, "tests", "golden")
cstr = m.load_problem(m.example_path("cstr_lmpc.py")); wb = m.load_problem(m.example_path("wood_berry_lmpc.py"))
for p, name in ((cstr, "cstr_shipped"), (wb, "wb_shipped")):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    for lk in (1, 2, 3):
        s = capi.Solver(p)
        try: s.set_option("loop_kernel", lk)
        except capi.MpcAmdError: s.close(); continue
        r = run_closed_loop(p, nsteps=100, solver=s)
        same = ((r["STATUS_DYN"] == 2) == (g["STATUS_DYN"] == 2)).all(axis=1)
        upto = int(np.argmin(same)) if not same.all() else 100
        print(name, lk, "upto", upto, "flips at", np.where(~same)[0][:10], "maxdiff U all", np.abs(r["U"] - g["U"]).max(), "upto", np.abs(r["U"][:upto] - g["U"][:upto]).max())
        s.close()
# full size: every instance vs C oracle, 100 steps
B, K = 4096, 100
x0 = np.random.default_rng(20250614).uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3))
c = oracle_c.OracleC(cstr).closed_loop(K, x0, x0)
for lk in (3, 2, 1):
    s = capi.Solver(cstr); s.set_option("loop_kernel", lk)
    g = run_closed_loop(cstr, x0, x0, K, solver=s); s.close()
    same = g["STATUS_DYN"] == c["STATUS_DYN"]
    good = same.all(axis=0)
    first = np.where(~good, np.argmin(same, axis=0), K)     # first flipped step per instance
    err = np.abs(g["U"] - c["U"]).max(axis=2)                 # [K,B]
    upto_err = max(err[:first[b], b].max() if first[b] > 0 else 0.0 for b in range(B))
    print("kernel", lk, "status eq frac", same.mean(), "instances all equal", good.mean(), "n flipped", (~good).sum(), "max err before first flip", upto_err,
          "max err on good", err[:, good].max(), "ss eq", (g["STATUS_SS"] == c["STATUS_SS"]).mean())
    for b in np.where(~good)[0][:6]:
        k = first[b]
        print("   inst", b, "first flip at step", k, "gpu", g["STATUS_DYN"][k, b], "c", c["STATUS_DYN"][k, b], "xhat", g["X_HAT"][k, b], "c xhat", c["X_HAT"][k, b])
