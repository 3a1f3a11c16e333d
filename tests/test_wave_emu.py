"""The economic path's KERNEL SOURCE on the CPU: ``mpc-code_amd/csrc/mpc_enmpc.hip`` compiled with g++ against the wave emulator of ``tests/wave_emu`` (64 host
fibers in lockstep stand for a wavefront; DPP moves, v_readlane, ds_bpermute and votes are restated from their lane patterns) and driven through the product's own
ctypes binding.  What the ``-m gpu`` tests of ``test_enmpc.py`` assert on the MI355X is asserted here lane by lane without one: the emulated library is handed to the
very same test functions in place of the hipcc-built one.  Test infrastructure: the product's loader never builds or opens the emulated library."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "wave_emu"))
import build as emu_build      # noqa: E402
import test_enmpc as te        # noqa: E402


@pytest.fixture()
def emulated(monkeypatch):
    """every EnmpcSolver built inside the test gets the emulated library"""
    from mpc_code_amd import econcodegen
    monkeypatch.setattr(econcodegen, "build_enmpc_library", lambda p, *a, **k: emu_build.build(p))
    yield


def test_wave_primitives_follow_the_dpp_lane_patterns(tmp_path):
    """the emulator's reductions and shifts against plain NumPy on a wave of random values (sum: the DPP tree's association of the additions)"""
    import ctypes as ct
    import subprocess
    src = tmp_path / "prim.cpp"
    src.write_text(r'''
#include "wave_emu.hpp"
extern "C" void run(const double *in, double *out) {   // out[op][lane]
    emu::launch(dim3(1), dim3(64), [&]() {
        const int l = threadIdx.x; const double v = in[l];
        out[0 * 64 + l] = mpc::wave_sum(v); out[1 * 64 + l] = mpc::wave_max(v); out[2 * 64 + l] = mpc::half_sum(v); out[3 * 64 + l] = mpc::half_max(v);
        out[4 * 64 + l] = mpc::wave_up1(-1.0, v); out[5 * 64 + l] = mpc::wave_dn1(-2.0, v); out[6 * 64 + l] = mpc::lane_of(v, 17); out[7 * 64 + l] = __shfl(v, 63 - l);
        out[8 * 64 + l] = (double)__any(l == 40); out[9 * 64 + l] = (double)__all(l < 64); out[10 * 64 + l] = (double)(__ballot(l & 1) == 0xAAAAAAAAAAAAAAAAull);
    });
}''')
    lib = tmp_path / "prim.so"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-w", "-I", os.path.join(HERE, "wave_emu", "include"), "-I", os.path.join(HERE, "wave_emu"),
                           "-I", os.path.join(te.ROOT, "mpc-code_amd", "csrc"), "-o", str(lib), str(src)])
    f = ct.CDLL(str(lib)).run
    v = np.random.default_rng(0).normal(size=64); out = np.zeros((11, 64))
    f(v.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p))
    rows = v.reshape(4, 16)
    def row_sum(r):      # quad swaps, half-row mirror, row mirror: a butterfly
        r = r + r.reshape(-1, 2)[:, ::-1].ravel(); r = r + r.reshape(-1, 4)[:, [2, 3, 0, 1]].ravel(); r = r + r.reshape(-1, 8)[:, ::-1].ravel(); r = r + r[::-1]
        return r[15]
    rs = [row_sum(r) for r in rows]
    assert np.all(out[0] == (rs[3] + rs[2]) + (rs[1] + rs[0])) and abs(out[0, 0] - v.sum()) < 1e-13
    assert np.all(out[1] == v.max())
    assert np.all(out[2, :32] == rs[1] + rs[0]) and np.all(out[2, 32:] == rs[3] + rs[2])
    assert np.all(out[3, :32] == v[:32].max()) and np.all(out[3, 32:] == v[32:].max())
    assert out[4, 0] == -1.0 and np.array_equal(out[4, 1:], v[:-1]) and out[5, 63] == -2.0 and np.array_equal(out[5, :-1], v[1:])
    assert np.all(out[6] == v[17]) and np.array_equal(out[7], v[::-1]) and np.all(out[8:] == 1.0)


@pytest.mark.parametrize("kernel", [1, 2])
def test_kernel_source_follows_the_golden_loop_of_the_shipped_example(emulated, pkg, kernel):
    te.test_gpu_shipped_example_follows_the_golden_loop.__wrapped__(pkg, np.load(te.GOLD), kernel) if hasattr(te.test_gpu_shipped_example_follows_the_golden_loop, "__wrapped__") \
        else te.test_gpu_shipped_example_follows_the_golden_loop(pkg, np.load(te.GOLD), kernel)


def test_kernel_source_follows_the_golden_loops_of_the_baseline_horizons(emulated, pkg):
    te.test_gpu_baseline_config_horizons_follow_the_golden_loops(pkg, np.load(te.GOLD), 2)


@pytest.mark.parametrize("kernel", [2, 64])
def test_kernel_source_filter_update_of_the_arrival_cost(emulated, pkg, kernel):
    te.test_gpu_filter_update_of_the_arrival_cost_follows_the_golden_loop(pkg, np.load(te.GOLD), kernel)


@pytest.mark.parametrize("pre", ["ekf_", "sat_"])
def test_kernel_source_extended_kalman_filter(emulated, pkg, pre):
    te.test_gpu_extended_kalman_filter_follows_the_golden_loops_and_the_c_restatement(pkg, pre, B=7)      # (the GPU takes 200 starts)


def test_kernel_source_estimator_with_bounded_state_noise(emulated, pkg):
    te.test_gpu_estimator_with_bounded_state_noise_follows_the_c_restatement(pkg, {}, B=7)      # (the GPU takes 70 starts)


@pytest.mark.parametrize("over,what", [({"xmin": np.array([0.8, 0.8]), "N": 12}, "ocp"), ({"xmin_ss": np.array([0.8, 0.8]), "N": 12}, "target")])
def test_kernel_source_unreachable_boxes_take_the_hold_branches(emulated, pkg, over, what):
    te.test_gpu_unreachable_boxes_take_the_hold_branches(pkg, over, what)


@pytest.mark.parametrize("over,nsteps", [({"N": 2, "N_mhe": 2}, 8), ({"N": 33, "N_mhe": 17}, 22), ({"Sol_itmax": 4}, 6)])
def test_kernel_source_edge_horizons_and_iteration_limits(emulated, pkg, over, nsteps):
    te.test_gpu_edge_horizons_and_iteration_limits_follow_the_c_restatement(pkg, over, nsteps)


def test_kernel_source_launch_styles_and_lane_counts_agree(emulated, pkg):
    """one launch for all steps, split pipeline with 64 / 32 / 16 lanes per instance, a ragged batch: the same loop"""
    from mpc_code_amd import enmpc
    p = pkg.load_problem(te.EX, overrides={"N": 14, "N_mhe": 6})
    x0 = np.random.default_rng(3).uniform([0.5, 0.0], [1.0, 0.5], size=(7, 2))
    s = enmpc.EnmpcSolver(p)
    try:
        r = {k: enmpc.run_enmpc_closed_loop(p, x0, 9, solver=s, kernel=k) for k in (1, 64, 32, 16)}
    finally:
        s.close()
    for k in (64, 32, 16):
        for nm in ("U", "X_ES", "XS"):
            assert np.abs(r[k][nm] - r[1][nm]).max() < 1e-9, (k, nm)
        for nm in ("ITERS_DYN", "ITERS_SS", "ITERS_MHE", "STATUS_DYN", "STATUS_SS", "STATUS_MHE"):
            assert np.array_equal(r[k][nm], r[1][nm]), (k, nm)


def test_kernel_source_restoration_phase_of_the_ocp(emulated, pkg):
    """The two cold OCPs of BASELINE configs[3]'s per-GPU share (instances 6907 and 9079 of 16384, N = 40) whose line search runs out of step lengths at an infeasible
    point: IPOPT enters its restoration phase there.  The kernels' rare path (ipm_stage MODE 1 / 2: the restoration problem as a stage problem with the defects as
    inputs, enmpc_ocp_resto_kernel behind the OCP launch; the one-launch kernel's non-inlined redo) against the C restatement (dense, orc_dense.h:ipm_restore):
    one restoration iteration, 25 iterations in all, the same point - and the restatement without the phase (EORC_RESTO=0: round 4's kernels) holds the input."""
    import enmpc_oracle_c as ec
    from mpc_code_amd import enmpc
    x0 = np.array([[0.82282676, 0.01523584], [0.73551569, 0.01499672], [0.9, 0.2]])
    over = {"N": 40}
    p = pkg.load_problem(te.EX, overrides=over)
    q = te.eo.load_problem(te.EX, overrides=over)
    c = ec.OracleEC(q).closed_loop(3, x0, nthreads=3)
    assert c["ITERS_DYN"][0].tolist() == [25, 25, 23] and int(c["STATUS_DYN"].max()) == 0
    os.environ["EORC_RESTO"] = "0"
    try:
        c0 = ec.OracleEC(q).closed_loop(1, x0, nthreads=3)
    finally:
        del os.environ["EORC_RESTO"]
    assert c0["STATUS_DYN"][0].tolist() == [2, 2, 0] and c0["ITERS_DYN"][0].tolist() == [8, 8, 23]
    s = enmpc.EnmpcSolver(p)
    try:
        for kernel in (1, 2):
            r = enmpc.run_enmpc_closed_loop(p, x0, 3, solver=s, kernel=kernel)
            for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE", "ITERS_DYN", "ITERS_SS", "ITERS_MHE"):
                assert np.array_equal(r[k], c[k]), (kernel, k, r[k].T.tolist(), c[k].T.tolist())
            for k in ("U", "XS", "US", "X_ES", "Xp"):
                assert np.abs(r[k] - c[k]).max() < te.TOL_U, (kernel, k)
    finally:
        s.close()


def test_kernel_source_sweeps_as_scans_and_as_recursions_agree(pkg):
    """The OCP's sweeps over the lanes are parallel scans in the product's build (mpc_enmpc.hpp:ric_backward_scan, ric_forward) with the recursions as the fallback of a wave in
    which a stage lacks curvature of its own; -DEC_SWEEP_SERIAL builds the recursions alone.  Both builds of the kernel source on the same loop: every status word and iteration
    count equal, values to rounding - and both on the C restatement's."""
    import enmpc_oracle_c as ec
    from mpc_code_amd import enmpc
    over = {"N": 33, "N_mhe": 6}
    p = pkg.load_problem(te.EX, overrides=over)
    x0 = np.random.default_rng(5).uniform([0.5, 0.0], [1.0, 0.5], size=(5, 2))
    c = ec.OracleEC(te.eo.load_problem(te.EX, overrides=over)).closed_loop(7, x0, nthreads=3)
    res = {}
    for name, flags in (("scans", ()), ("recursions", ("-DEC_SWEEP_SERIAL",))):
        s = enmpc.EnmpcSolver(p, lib_path=emu_build.build(p, extra_flags=flags))
        try:
            res[name] = enmpc.run_enmpc_closed_loop(p, x0, 7, solver=s, kernel=2)
        finally:
            s.close()
    for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE", "ITERS_DYN", "ITERS_SS", "ITERS_MHE"):
        assert np.array_equal(res["scans"][k], res["recursions"][k]) and np.array_equal(res["scans"][k], c[k]), k
    for k in ("U", "XS", "US", "X_ES", "Xp"):
        assert np.abs(res["scans"][k] - res["recursions"][k]).max() < 1e-11 and np.abs(res["scans"][k] - c[k]).max() < te.TOL_U, k
    assert not np.array_equal(res["scans"]["U"], res["recursions"]["U"])      # (two forms of the sweep: not the same bits)


def test_kernel_source_estimator_restoration_on_an_infeasible_window(emulated, pkg):
    """An estimator NLP that cannot be met - state noise boxed to 1e-4 where the measurements need a hundred times that - takes the estimator's restoration phase
    (enmpc_mhe_resto_kernel, the one-launch kernel's redo) to its end: the kernels report the failure (status 2: the loop keeps the predicted state) on every instance and step,
    in both launch styles with the same numbers, as the C restatement does (status 1 or 2 - its restoration follows other iterates: the kernels eliminate the output noise through
    its linear defining row, the restatements keep it as variables, so their infeasibility measures differ while that row is unmet); nothing non-finite leaves
    either."""
    import enmpc_oracle_c as ec
    from mpc_code_amd import enmpc
    over = {"wmin": [-1e-4] * 4, "wmax": [1e-4] * 4, "N_mhe": 8, "xmin": np.array([0.7, 0.1]), "xmax": np.array([1.0, 0.5]), "dmin": np.array([-0.01, -0.01]), "dmax": np.array([0.01, 0.01])}
    x0 = np.array([[0.9, 0.2], [0.6, 0.4], [0.75, 0.05]])
    p = pkg.load_problem(te.EX, overrides=over)
    q = te.eo.load_problem(te.EX, overrides=over)
    c = ec.OracleEC(q).closed_loop(3, x0, nthreads=3)
    assert int(c["STATUS_MHE"].min()) >= 1 and np.isfinite(c["X_ES"]).all()
    s = enmpc.EnmpcSolver(p)
    try:
        ref = None
        for kernel in (2, 1):
            r = enmpc.run_enmpc_closed_loop(p, x0, 3, solver=s, kernel=kernel)
            assert int(r["STATUS_MHE"].min()) == 2 and int(r["ITERS_MHE"].max()) < 100 and np.isfinite(r["X_ES"]).all() and np.isfinite(r["U"]).all(), kernel
            if ref is not None:
                for k in ("U", "X_ES", "ITERS_MHE", "ITERS_DYN", "STATUS_DYN"):
                    assert np.array_equal(r[k], ref[k]), k
            ref = r
    finally:
        s.close()


def test_kernel_source_user_inequality_rows(emulated, pkg):
    te.test_gpu_user_inequality_rows_follow_the_oracle(pkg, 3, 6)


@pytest.mark.parametrize("seed", [7, 24])
def test_kernel_source_randomised_reactor_models(emulated, pkg, seed):
    te.test_gpu_randomised_reactor_models_follow_the_c_restatement(pkg, seed)
