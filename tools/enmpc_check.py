"""Development probe (GPU box): the economic closed loop on the HIP path against oracle/enmpc_oracle.py, per step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mpc_code_amd as m
from mpc_code_amd import enmpc
import enmpc_oracle as eo

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 21
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
over = {}
if len(sys.argv) > 3:
    over = {"N": int(sys.argv[3]), "N_mhe": int(sys.argv[4])}
path = m.example_path("reactor_enmpc.py")
p = m.load_problem(path, overrides=over or None)
rng = np.random.default_rng(20250614)
x0 = np.vstack([p.x0_p, rng.uniform([0.5, 0.0], [1.0, 0.5], size=(B - 1, 2))]) if B > 1 else p.x0_p[None]
t0 = time.time()
r = enmpc.run_enmpc_closed_loop(p, x0, nsteps)
print(f"GPU: {time.time() - t0:.2f} s wall, kernel {r['kernel_ms']:.1f} ms")
q = eo.load_problem(path, overrides=over or None)
for b in range(min(B, 3)):
    o = eo.closed_loop(q, nsteps, x0_p=x0[b])
    for k in ("U", "XS", "US", "X_ES", "X_HAT", "Xp"):
        print(b, k, "max diff", float(np.abs(r[k][:, b] - o[k]).max()))
    for k in ("ITERS_DYN", "ITERS_SS", "ITERS_MHE", "STATUS_DYN", "STATUS_SS"):
        print(b, k, "gpu", r[k][:, b].tolist(), "oracle", o[k].tolist())
