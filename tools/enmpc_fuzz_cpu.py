#!/usr/bin/env python3
"""Diagnostic (CPU): the randomised reactor models of tools/enmpc_fuzz.py on the C restatement alone (and, with --numpy, on the NumPy
restatement beside it): status words, iteration counts, how the solves ended.
   tools/enmpc_fuzz_cpu.py [first seed] [count] [--numpy]"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.simplefilter("ignore")
import enmpc_oracle as eo, enmpc_oracle_c as ec
from enmpc_cases import draw
EX = os.path.join(ROOT, "mpc-code_amd", "examples", "reactor_enmpc.py")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    s0, n = (int(args[0]) if len(args) > 0 else 5), (int(args[1]) if len(args) > 1 else 32)
    nsteps = 12
    tot = {"st2": 0, "over60": 0, "maxit": 0}
    for seed in range(s0, s0 + n):
        over, x0 = draw(seed)
        p = eo.load_problem(EX, overrides=over)
        c = ec.OracleEC(p).closed_loop(nsteps, x0, nthreads=8)
        st = {k: [int((c["STATUS_" + k] == v).sum()) for v in (1, 2)] for k in ("DYN", "SS", "MHE")}
        mi = {k: int(c["ITERS_" + k].max()) for k in ("DYN", "SS", "MHE")}
        tot["st2"] += sum(v[1] for v in st.values()); tot["maxit"] += sum(v[0] for v in st.values()); tot["over60"] += sum(int((c["ITERS_" + k] > 60).sum()) for k in ("DYN", "SS", "MHE"))
        msg = ""
        if "--numpy" in sys.argv:
            dv, di = 0.0, 0
            for b in range(2):
                r = eo.closed_loop(p, 6, x0_p=x0[b])
                dv = max(dv, max(float(np.abs(r[k] - c[k][:6, b]).max()) for k in ("U", "XS", "US", "X_ES", "Xp")))
                di = max(di, max(int(np.abs(r[k].astype(int) - c[k][:6, b].astype(int)).max()) for k in ("ITERS_DYN", "ITERS_SS", "ITERS_MHE")))
                assert all(np.array_equal(r[k], c[k][:6, b]) for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE")), (seed, b)
            msg = f" | numpy: |dv| {dv:.1e} iters +-{di}"
        print(seed, {k: (round(v, 3) if isinstance(v, float) else (np.round(v, 3).tolist() if isinstance(v, (list, np.ndarray)) else v)) for k, v in over.items()},
              "| status 1/2 counts", st, "| max iters", mi, msg, flush=True)
    print("totals:", tot)
