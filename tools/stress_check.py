import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mpc_code_amd as m
from mpc_code_amd import capi
import oracle_c
for ex, B, K in (("cstr_lmpc.py", int(os.environ.get("STRESS_B", 4096)), 100), ("wood_berry_lmpc.py", int(os.environ.get("STRESS_B", 4096)) // 2, 100)):
    p = m.load_problem(m.example_path(ex))
    for seed in tuple(int(v) for v in os.environ.get("STRESS_SEEDS", "1,2,3").split(",")):
        rng = np.random.default_rng(seed)
        x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3)) if p.nx == 3 else 0.05 * rng.standard_normal((B, p.nx))
        ref = oracle_c.OracleC(p).closed_loop(K, x0, x0)
        for mode in (3, 2, 1):
            s = capi.Solver(p)
            try:
                s.set_option("loop_kernel", mode)
            except capi.MpcAmdError:
                s.close(); continue
            s.loop_alloc(B, K, capi.LOG_ALL); s.loop_set_schedule(p.schedules(K)); s.loop_set_state(x0, x0)
            s.loop_run(0, K); s.loop_sync()
            U = s.loop_get_log("U"); st = s.loop_get_log("STATUS_DYN"); ss = s.loop_get_log("STATUS_SS")
            same = (st == ref["STATUS_DYN"]).all(axis=0)
            print(f"{ex} seed {seed} kernel {mode}: status eq {np.mean(st == ref['STATUS_DYN']):.5f} ss eq {np.mean(ss == ref['STATUS_SS']):.5f} instances with all statuses equal {same.mean():.4f}, "
                  f"max|U-Uc| on those {np.abs(U - ref['U'])[:, same].max():.2e}, nan {np.isnan(U).sum()}, maxiter {np.sum(st == 1)}", flush=True)
            s.close()
