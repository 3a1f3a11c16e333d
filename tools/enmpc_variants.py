#!/usr/bin/env python3
"""Measurement: build variants of the economic kernels on the two economic workloads (one MI355X).  Each variant is a per-model library of its own (extra compiler flags enter
the library's hash), passes the create-time self-test or is reported as refused, and has to reproduce the base build's loop bit for bit.
   tools/enmpc_variants.py build      (here, no GPU: compile the variants ahead)
   tools/enmpc_variants.py [out.json] (GPU box)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_code_amd as m
from mpc_code_amd import enmpc, econcodegen

VARIANTS = [("base", []),
            ("broadcasts left in scalar registers", ["-DEC_BCAST_IN_SGPRS"]),
            ("OCP kernel compiled for one wave per SIMD (512 registers)", ["-DEC_OCP_WAVES=1"]),
            ("both", ["-DEC_BCAST_IN_SGPRS", "-DEC_OCP_WAVES=1"]),
            # the OCP kernel's scratch frame (204 B per lane) holds loop invariants the compiler hoisted out of the iteration loop - the polynomial coefficients of exp / log among
            # them - and then had no registers for: without machine LICM the frame is empty (tools/kernel_resources.py)
            ("machine LICM off", ["-mllvm", "-disable-machine-licm"]),      # (the product's build since this measurement: econcodegen.ENMPC_FLAGS)
            ("loop invariants sunk back where they would spill", ["-mllvm", "-sink-insts-to-avoid-spills"]),
            # the OCP's sweeps over the lanes (mpc_enmpc.hpp:ric_backward_scan, ric_forward): parallel scans in the product's build since round 5 (third record: the backward
            # scan as the variant, + 11 %); these build the recursions back in - values part by rounding
            ("both sweeps as recursions (the build before the scans)", ["-DEC_SWEEP_SERIAL"]),
            ("forward sweep as a recursion, backward matrix sweep as a scan", ["-DEC_FWD_SERIAL"]),
            ("scans for stage states up to four too (the estimator, an OCP with user rows)", ["-DEC_SWEEP_SCAN_MAXNS=4"])]
WORK = [("enmpc N=40, 16384 instances", {"N": 40}, 16384), ("mhe N_mhe=20, 4096 instances", {"N_mhe": 20}, 4096)]

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        from concurrent.futures import ThreadPoolExecutor
        p = m.load_problem(m.example_path("reactor_enmpc.py"))
        with ThreadPoolExecutor(4) as ex:
            for name, lib in zip(VARIANTS, ex.map(lambda v: econcodegen.build_enmpc_library(p, extra_flags=v[1]), VARIANTS)):
                print(name[0], os.path.basename(lib), flush=True)
        sys.exit(0)
    res = []
    for wname, over, B in WORK:
        p = m.load_problem(m.example_path("reactor_enmpc.py"), overrides=over)
        x0 = np.random.default_rng(1).uniform([0.5, 0.0], [1.0, 0.5], size=(B, 2))
        ref = None
        for vname, flags in VARIANTS:
            lib = econcodegen.build_enmpc_library(p, extra_flags=flags)
            try:
                s = enmpc.EnmpcSolver(p, lib_path=lib)
            except Exception as e:
                res.append(dict(workload=wname, variant=vname, flags=flags, refused=str(e)[:300])); print(res[-1], flush=True); continue
            s.time_kernels(True)
            best = None
            for rep in range(3):
                r = enmpc.run_enmpc_closed_loop(p, x0, 20, solver=s, kernel=2, groups=1)
                best = r["kernel_ms"] if best is None else min(best, r["kernel_ms"])
            ph = s.phase_ms()
            s.time_kernels(False)
            for rep in range(3):
                r2 = enmpc.run_enmpc_closed_loop(p, x0, 20, solver=s, kernel=2)
            same = True if ref is None else bool(all(np.array_equal(ref[k], r[k]) for k in ("U", "X_ES", "XS", "ITERS_DYN", "ITERS_MHE", "ITERS_SS", "STATUS_DYN")))
            apart = None if ref is None or same else dict(max_abs_dU=float(np.abs(ref["U"] - r["U"]).max()), ocp_solves_with_other_iteration_count=int((ref["ITERS_DYN"] != r["ITERS_DYN"]).sum()),
                                                          largest_iteration_difference=int(np.abs(ref["ITERS_DYN"].astype(int) - r["ITERS_DYN"]).max()), solves=int(r["ITERS_DYN"].size),
                                                          status_words_that_differ=int((ref["STATUS_DYN"] != r["STATUS_DYN"]).sum()))
            ref = ref or r
            res.append(dict(workload=wname, variant=vname, flags=flags, one_stream_ms=best, msteps_per_s_one_stream=B * 20 / best / 1e3, stream_groups_ms=r2["kernel_ms"],
                            msteps_per_s=B * 20 / r2["kernel_ms"] / 1e3, phase_ms_estimator_target_ocp=ph[0], launches=ph[1], same_as_base=same, apart_from_base=apart))
            print(res[-1], flush=True)
            s.close()
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)
