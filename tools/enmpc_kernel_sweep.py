#!/usr/bin/env python3
"""Measurement: the economic closed loop's two launch styles over the batch size (one MI355X): tools/enmpc_kernel_sweep.py [out.json]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import enmpc
res = []
for name, over in (("enmpc N=40", {"N": 40}), ("mhe N_mhe=20", {"N_mhe": 20})):
    p = m.load_problem(m.example_path("reactor_enmpc.py"), overrides=over)
    s = enmpc.EnmpcSolver(p)
    for B in (256, 1024, 4096, 16384, 65536):
        x0 = np.random.default_rng(1).uniform([0.5, 0.0], [1.0, 0.5], size=(B, 2))
        ref = None
        for kern in (1, 64, 2):
            for rep in range(2):
                r = enmpc.run_enmpc_closed_loop(p, x0, 20, solver=s, kernel=kern)
            same = True if ref is None else bool(np.array_equal(ref["U"], r["U"]) and np.array_equal(ref["ITERS_MHE"], r["ITERS_MHE"]) and np.array_equal(ref["X_ES"], r["X_ES"]))
            ref = ref or r
            if kern == 64 and B <= 1024:
                continue
            apart = None if same else dict(max_abs_dU=float(np.abs(ref["U"] - r["U"]).max()), iteration_counts_that_differ=int((ref["ITERS_DYN"] != r["ITERS_DYN"]).sum() + (ref["ITERS_MHE"] != r["ITERS_MHE"]).sum()))
            res.append(dict(config=name, batch=B, kernel=kern, ms=r["kernel_ms"], msteps_per_s=B * 20 / r["kernel_ms"] / 1e3, same_as_kernel_1=same, apart_from_kernel_1=apart))
            print(res[-1], flush=True)
    s.close()
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1)
