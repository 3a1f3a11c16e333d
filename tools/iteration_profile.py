#!/usr/bin/env python3
"""Where the headline loop's interior-point iterations go: the benchmark's workload (configs[1]: 20 steps, the seeded x0 box of bench.py) through the C restatement,
   iterations per step over the instances and over the waves of four instances the tile kernel runs (a wave iterates until its slowest instance is done).
   CPU only (oracle/): tools/iteration_profile.py [instances] > profiles/rNN_iteration_profile.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mpc_code_amd as m, oracle_c

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    p = m.load_problem(m.example_path("cstr_lmpc.py"))
    rng = np.random.default_rng(20250614)
    x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3))
    t = time.time(); r = oracle_c.OracleC(p).closed_loop(20, x0, x0)
    it = r["ITERS_DYN"]                                     # [step][instance]
    w = it.reshape(20, B // 4, 4).max(axis=2)
    np.set_printoptions(linewidth=200)
    print(f"cstr_lmpc, 20 steps, {B} instances, C restatement ({time.time() - t:.1f} s); 1 = the convergence test of a warm start that passes it, 0 = an OCP found infeasible before the first iteration (the input is held)")
    print("mean iterations per step over the instances :", np.round(it.mean(axis=1), 2))
    print("largest                                      :", it.max(axis=1))
    print("mean per step over waves of four (max of 4)  :", np.round(w.mean(axis=1), 2))
    print(f"per instance, summed over the steps          : mean {it.sum(axis=0).mean():.1f}, largest {it.sum(axis=0).max()}")
    print(f"per wave, summed over the steps              : mean {w.sum(axis=0).mean():.1f}, largest {w.sum(axis=0).max()}")
    big = np.argsort(-w.mean(axis=1))[:4]
    print(f"the four steps {sorted(big.tolist())} hold {w.mean(axis=1)[big].sum():.1f} of the waves' {w.mean(axis=1).sum():.1f} iterations")
    for k in sorted(big.tolist()):
        h = np.bincount(it[k], minlength=it[k].max() + 1)
        print(f"  step {k:2d}: iterations -> instances", {i: int(c) for i, c in enumerate(h) if c})
