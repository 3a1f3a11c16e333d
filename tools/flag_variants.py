#!/usr/bin/env python3
"""Measurement: compiler-flag variants of the linear (headline dimension set) and the non-linear (Ex_NMPC) library on their benchmark workloads, one MI355X.
   tools/flag_variants.py build        (here: compiles csrc/jit/variant_<name>_{amd,nmpc}.so)
   tools/flag_variants.py [out.json]   (GPU box: 4096 x 20 steps of LMPC-CSTR, 16384 x 20 of Ex_NMPC; best of five; results compared with the base build's)"""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import capi, nlcodegen
CSRC = capi.CSRC
VARIANTS = [("base", []), ("machine_licm_off", ["-mllvm", "-disable-machine-licm"]), ("no_vgpr_to_agpr_spills", ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]),
            ("both", ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"])]
lib = lambda name, kind: os.path.join(CSRC, "jit", f"variant_{name}_{kind}.so")


def build():
    from concurrent.futures import ThreadPoolExecutor
    pn = m.load_problem(m.example_path("cstr_nmpc.py"))
    hdr = nlcodegen.build_nmpc_library(pn)[:-3] + "_model.hpp"
    jobs = []
    for name, fl in VARIANTS:
        jobs.append(["/opt/rocm/bin/hipcc", *capi.HIPCC_FLAGS, *fl, "-DMPC_DIM_LIST(X)=X(3,2,3,3,3,0,0)", "-o", lib(name, "amd"), os.path.join(CSRC, "mpc_amd.hip")])
        jobs.append(["/opt/rocm/bin/hipcc", *nlcodegen.NMPC_FLAGS, *fl, f'-DMPC_NL_MODEL_HEADER="{hdr}"', "-o", lib(name, "nmpc"), os.path.join(CSRC, "mpc_nmpc.hip")])
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(lambda c: subprocess.check_call(c, cwd=CSRC), jobs))
    print("built", [os.path.basename(j[j.index("-o") + 1]) for j in jobs])


def run_linear(name):      # child process: capi takes the library from MPC_AMD_LIB
    from mpc_code_amd.driver import run_closed_loop
    p = m.load_problem(m.example_path("cstr_lmpc.py"))
    x0 = np.random.default_rng(20250614).uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(4096, 3))
    s = capi.Solver(p)
    s.set_option("steps_per_launch", 20)
    best = None
    for _ in range(6):
        r = run_closed_loop(p, x0, x0, 20, solver=s)
        ms, _n = s.last_kernel_ms()
        best = ms if best is None else min(best, ms)
    print(json.dumps(dict(ms=best, u_sum=float(np.abs(r["U"]).sum()), iters=int(r["ITERS_DYN"].sum()))))


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        build(); sys.exit(0)
    if sys.argv[1:2] == ["linear"]:
        run_linear(sys.argv[2]); sys.exit(0)
    from mpc_code_amd import nmpc
    res, ref = [], {}
    for name, fl in VARIANTS:
        o = subprocess.run([sys.executable, __file__, "linear", name], capture_output=True, text=True, env=dict(os.environ, MPC_AMD_LIB=lib(name, "amd")))
        try:
            d = json.loads(o.stdout.strip().split("\n")[-1])
        except Exception:
            d = dict(error=(o.stderr or o.stdout)[-300:])
        ref.setdefault("lin", d)
        res.append(dict(library="linear (3,2,3,3,3,0,0), LMPC-CSTR 4096 x 20", variant=name, flags=fl, **d, msteps_per_s=(4096 * 20 / d["ms"] / 1e3 if "ms" in d else None),
                        same_as_base=("ms" in d and d["u_sum"] == ref["lin"].get("u_sum") and d["iters"] == ref["lin"].get("iters"))))
        print(res[-1], flush=True)
    pn = m.load_problem(m.example_path("cstr_nmpc.py"))
    x0 = None
    for name, fl in VARIANTS:
        try:
            s = nmpc.NmpcSolver(pn, lib_path=lib(name, "nmpc"))
        except Exception as e:
            res.append(dict(library="non-linear", variant=name, flags=fl, error=str(e)[:300])); print(res[-1], flush=True); continue
        B = 16384
        rng = np.random.default_rng(20250614)
        xp = np.tile(pn.x0_p, (B, 1)) * (1.0 + 0.01 * rng.uniform(-1, 1, size=(B, pn.nxp)))
        best = None
        for _ in range(6):
            r = nmpc.run_nmpc_closed_loop(pn, xp, xp[:, :pn.nx], 20, solver=s)
            km = s.last_kernel_ms(); best = km if best is None else min(best, km)
        d = dict(ms=best, u_sum=float(np.abs(r["U"]).sum()), iters=int(r["ITERS_DYN"].sum()))
        ref.setdefault("nl", d)
        res.append(dict(library="non-linear (Ex_NMPC), 16384 x 20", variant=name, flags=fl, **d, msteps_per_s=B * 20 / best / 1e3,
                        same_as_base=(d["u_sum"] == ref["nl"]["u_sum"] and d["iters"] == ref["nl"]["iters"])))
        print(res[-1], flush=True)
        s.close()
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)
