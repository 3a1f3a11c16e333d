#!/usr/bin/env python3
"""Diagnostic: shader-clock ticks per phase of the economic kernel (library built with -DMPC_STAMPS):
   tools/enmpc_stamps.py [N] [N_mhe] [batch<=4096]"""
import ctypes as ct, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import enmpc, econcodegen
over = {"N": int(sys.argv[1]) if len(sys.argv) > 1 else 40, "N_mhe": int(sys.argv[2]) if len(sys.argv) > 2 else 10}
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
p = m.load_problem(m.example_path("reactor_enmpc.py"), overrides=over)
lib = econcodegen.build_enmpc_library(p, extra_flags=["-DMPC_STAMPS"])
s = enmpc.EnmpcSolver(p, lib_path=lib)
K = 14
x0 = np.random.default_rng(1).uniform([0.5, 0.0], [1.0, 0.5], size=(B, 2))
s.alloc(B, K); s.set_state(x0)
names = ["mhe nlp", "kalman+smooth", "target nlp", "ocp nlp", "logs+plant"]
buf = np.zeros(64 * 8, np.uint64)
s.lib.enmpc_debug_stamps(None, 0, 1)
for k in range(K):
    s.run(k, 1); s.sync()
    s.lib.enmpc_debug_stamps(buf.ctypes.data_as(ct.c_void_p), 64 * 8, 1)
    c = buf.reshape(64, 8).astype(float)[:, :5]
    w = np.argmax(c.sum(axis=1))
    it = {n: s.get_log(n)[k] for n in ("ITERS_MHE", "ITERS_SS", "ITERS_DYN")}
    print(f"step {k:2d} kernel {s.last_kernel_ms():.3f} ms | slowest of the first 64 waves, kticks: " + " ".join(f"{n}={c[w, i]/1e3:.0f}" for i, n in enumerate(names))
          + f" | total {c[w].sum()/1e3:.0f} | mean iters mhe/ss/dyn {it['ITERS_MHE'].mean():.1f}/{it['ITERS_SS'].mean():.1f}/{it['ITERS_DYN'].mean():.1f}")
