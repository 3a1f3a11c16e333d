#!/usr/bin/env python3
"""Diagnostic (round 3): do the linear and the non-linear wave kernels survive build perturbations?  The economic library did not while the
broadcasts of its recursions lived in scalar registers (tools/enmpc_bcast_matrix.py).  Builds the CSTR dimension set of libmpc_amd and the
library of cstr_nmpc under the same six perturbations and runs, per build and in a process of its own, a closed loop on every kernel against
the C restatement (linear) / the golden real-time-iteration loop (non-linear): '.' right, 'X' wrong, 'T' no answer within 60 s.
   build here:  tools/build_perturbation_check.py build        run on the GPU box:  tools/build_perturbation_check.py
Recorded: profiles/r03_build_perturbations.txt."""
import os, subprocess, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
CSRC = os.path.join(ROOT, "mpc-code_amd", "csrc")
OUT = os.path.join(CSRC, "jit")
PERT = {"plain": [], "O2": ["-O2"], "noagpr": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"], "nopost": ["-mllvm", "-enable-post-misched=0"], "nosink": ["-mllvm", "-disable-machine-sink"],
        "nosched": ["-mllvm", "-enable-misched=0"]}
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def lib(kind, name):
    return os.path.join(OUT, f"libpert_{kind}_{name}.so")


def build():
    warnings.simplefilter("ignore")
    import mpc_code_amd as m
    from mpc_code_amd import nlcodegen
    p = m.load_problem(m.example_path("cstr_nmpc.py"))
    hdr = os.path.join(OUT, "libpert_nmpc_model.hpp")
    open(hdr, "w").write(nlcodegen.emit_model_header(p))
    jobs = []
    for name, fl in PERT.items():
        jobs.append(subprocess.Popen(BASE + fl + ["-DMPC_DIM_LIST(X)=X(3,2,3,3,3,0,0)", "-o", lib("lin", name), os.path.join(CSRC, "mpc_amd.hip")], cwd=CSRC, stderr=subprocess.DEVNULL))
        jobs.append(subprocess.Popen(BASE + fl + [f'-DMPC_NL_MODEL_HEADER="{hdr}"', "-o", lib("nmpc", name), os.path.join(CSRC, "mpc_nmpc.hip")], cwd=CSRC, stderr=subprocess.DEVNULL))
        if len(jobs) >= 6:
            for j in jobs: j.wait()
            jobs = []
    for j in jobs: j.wait()


def child(kind, name):
    warnings.simplefilter("ignore")
    res = ""
    if kind == "lin":
        os.environ["MPC_AMD_LIB"] = lib("lin", name)
        import mpc_code_amd as m
        from mpc_code_amd import capi
        from mpc_code_amd.driver import run_closed_loop
        import oracle_c
        p = m.load_problem(m.example_path("cstr_lmpc.py"))
        x0 = np.random.default_rng(20250614).uniform([-0.5, -8, -5], [0.5, 8, 5], size=(300, 3))
        c = oracle_c.OracleC(p).closed_loop(20, x0, x0)
        for lk in (1, 2, 3):
            s = capi.Solver(p); s.set_option("loop_kernel", lk)
            r = run_closed_loop(p, x0, x0, 20, solver=s)
            res += "." if np.array_equal(r["STATUS_DYN"], c["STATUS_DYN"]) and np.abs(r["U"] - c["U"]).max() < 1e-7 else "X"
            s.close()
    else:
        import mpc_code_amd as m
        from mpc_code_amd import nmpc
        p = m.load_problem(m.example_path("cstr_nmpc.py"))
        g = np.load(os.path.join(ROOT, "tests", "golden", "nmpc_cstr.npz"))
        s = nmpc.NmpcSolver(p, lib_path=lib("nmpc", name))
        for kern in (1, 3, 4):
            s.set_kernel(kern)
            r = nmpc.run_nmpc_closed_loop(p, g["rti_x0"], g["rti_x0"], nsteps=20, solver=s, max_sqp=1)
            e = float(np.max(np.abs(r["U"] - g["rti_U"][:20]) / (1.0 + np.abs(g["rti_U"][:20]))))
            res += "." if np.array_equal(r["STATUS_DYN"], g["rti_STATUS_DYN"][:20]) and e < 1e-6 else "X"
        s.close()
    print(res)


if sys.argv[1:] == ["build"]:
    build(); sys.exit(0)
if len(sys.argv) == 3:
    child(sys.argv[1], sys.argv[2]); sys.exit(0)
for kind, what in (("lin", "linear: kernels lane / horizon-parallel / wave-autonomous "), ("nmpc", "non-linear: kernels lane / wave-autonomous / split    ")):
    out = []
    for name in PERT:
        try:
            o = subprocess.run([sys.executable, __file__, kind, name], capture_output=True, text=True, timeout=60).stdout.strip().split("\n")[-1]
        except subprocess.TimeoutExpired:
            o = "T"
        out.append(f"{name}:{o}")
    print(what, "  ".join(out), flush=True)
