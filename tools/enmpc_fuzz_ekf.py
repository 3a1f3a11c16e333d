#!/usr/bin/env python3
"""Diagnostic (GPU box): randomised reactor models with the extended Kalman filter in the estimator's place (examples/reactor_enmpc_ekf.py) on both launch styles against
oracle/enmpc_oracle.c.   tools/enmpc_fuzz_ekf.py [first seed] [count] [build: compile the libraries here, without a GPU]"""
import sys, os, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.simplefilter("ignore")
import numpy as np
import mpc_code_amd as m
from mpc_code_amd import enmpc, econcodegen
from enmpc_cases import draw
EX = m.example_path("reactor_enmpc_ekf.py")
seeds = range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2]))
build_only = len(sys.argv) > 3
def prob(seed):
    over, x0 = draw(seed)
    over = {k: v for k, v in over.items() if k not in ("N_mhe", "mhe_up")}
    return over, x0
if build_only:
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        for r in ex.map(lambda s: econcodegen.build_enmpc_library(m.load_problem(EX, overrides=prob(s)[0])), seeds): print(os.path.basename(r))
    sys.exit(0)
import enmpc_oracle as eo, enmpc_oracle_c as ec
bad = 0
for seed in seeds:
    over, x0 = prob(seed)
    c = ec.OracleEC(eo.load_problem(EX, overrides=over)).closed_loop(12, x0, nthreads=0)
    p = m.load_problem(EX, overrides=over)
    s = enmpc.EnmpcSolver(p)
    msg = []
    for kernel in (1, 2):
        r = enmpc.run_enmpc_closed_loop(p, x0, 12, solver=s, kernel=kernel)
        dv = max(float(np.abs(r[k] - c[k]).max()) for k in ("U", "XS", "US", "X_ES", "Xp"))
        st = all(np.array_equal(r[k], c[k]) for k in ("STATUS_DYN", "STATUS_SS"))
        di = max(int(np.abs(r[k].astype(int) - c[k].astype(int)).max()) for k in ("ITERS_DYN", "ITERS_SS"))
        ok = dv < 2e-6 and st and di <= 4; bad += not ok
        msg.append(f"k{kernel}: |dv| {dv:.1e} status {'=' if st else 'DIFFER'} iters +-{di}{'' if ok else '  <-- FAIL'}")
    s.close()
    print(seed, {k: (np.round(v, 3).tolist() if not isinstance(v, (str, int)) else v) for k, v in over.items()}, "| status max", int(c["STATUS_DYN"].max()), int(c["STATUS_SS"].max()), "|", " ; ".join(msg), flush=True)
print("failures:", bad)
