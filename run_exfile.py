#!/usr/bin/env python3
"""The reference's `python MPC_code.py` with an Ex-file of your choice, for a batch of instances on the MI355X:

    python run_exfile.py /path/to/Ex_LMPC_CSTR.py --batch 4096 --spread 0.05 --nsteps 100 --out run.npz
    python run_exfile.py /path/to/Ex_ENMPC.py -o N=40 -o N_mhe=20 --batch 1024

Loads the file unmodified (mpc-code_amd/exfile.py), runs the closed loop of MPC_code.py:485-827 through the HIP path the problem belongs to
(mpc_code_amd.run_example) and stores the reference's result arrays (U, X_HAT, XS, US, Xp, ... : [nsteps, batch, dim]) in an .npz; prints a
summary line per array.  `--batch B --spread s`: B instances whose plant and model start states are the file's own times (1 + s U(-1, 1))
(s = 0: B copies).  `-o NAME=VALUE` overrides a variable of the file (a Python literal), e.g. the horizons of the BASELINE configurations."""
import argparse
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("exfile")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--spread", type=float, default=0.0, help="relative spread of the start states over the batch")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--nsteps", type=int, default=None, help="default: the file's Nsim")
    ap.add_argument("--max-sqp", type=int, default=None, help="non-linear tracking path: SQP iterations per OCP (default 1: real-time iteration)")
    ap.add_argument("-o", "--override", action="append", default=[], metavar="NAME=VALUE")
    ap.add_argument("--out", default=None, help="write the result arrays to this .npz")
    ap.add_argument("--load-only", action="store_true", help="load and classify the file, then stop (no GPU needed)")
    ap.add_argument("--plots", default=None, metavar="DIR", help="write the reference's figures (State / Input / Output / Disturbance Estimate, MPC_code.py:897-935) of one instance as PDFs under DIR")
    ap.add_argument("--instance", type=int, default=0, help="the instance --plots draws")
    a = ap.parse_args(argv)
    import mpc_code_amd as m
    over = {}
    for item in a.override:
        k, _, v = item.partition("=")
        over[k.strip()] = ast.literal_eval(v)
    p = m.load_problem(a.exfile, overrides=over or None)
    print(f"{os.path.basename(a.exfile)}: {type(p).__name__}, nx={p.nx} nu={p.nu} ny={p.ny} nd={p.nd} N={p.N}" + (f" N_mhe={p.N_mhe}" if hasattr(p, "N_mhe") else ""))
    if a.load_only:
        return 0
    rng = np.random.default_rng(a.seed)
    x0p = np.tile(np.asarray(p.x0_p, dtype=float), (a.batch, 1)) * (1.0 + a.spread * rng.uniform(-1, 1, size=(a.batch, len(p.x0_p))))
    kw = {}
    if isinstance(p, m.EconomicMPCProblem):
        out = m.run_example(p, x0_p=x0p, nsteps=a.nsteps)
    else:
        x0m = np.tile(np.asarray(p.x0_m, dtype=float), (a.batch, 1)) * (1.0 + a.spread * rng.uniform(-1, 1, size=(a.batch, len(p.x0_m))))
        if a.max_sqp is not None and isinstance(p, m.NonlinearMPCProblem):
            kw["max_sqp"] = a.max_sqp
        out = m.run_example(p, x0_p=x0p, x0_m=x0m, nsteps=a.nsteps, **kw)
    for k, v in out.items():
        v = np.asarray(v)
        if v.ndim >= 2 and v.dtype.kind == "f":
            print(f"  {k:10s} {str(v.shape):18s} last step, instance 0: {np.array2string(v[-1, 0], precision=6)}")
        elif v.ndim >= 2:
            print(f"  {k:10s} {str(v.shape):18s} values {np.unique(v).tolist()[:8]}")
    if a.plots:
        from mpc_code_amd.plots import make_plots
        files = make_plots(out, float(p.h), a.plots, instance=a.instance)
        print(f"wrote {len(files)} figures under {a.plots}")
    if a.out:
        np.savez_compressed(a.out, **{k: np.asarray(v) for k, v in out.items()})
        print("wrote", a.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
