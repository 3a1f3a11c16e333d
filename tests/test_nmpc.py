"""Non-linear path (BASELINE config 3): tracer, Ex-file loading, oracle against its committed
vectors, and - on the GPU - libmpc_nmpc_<model>.so against them.

tests/golden/nmpc_cstr.npz comes from oracle/nmpc_oracle.py (tests/golden/make_nmpc_golden.py); its converged rows carry the
residuals of the NLP's own KKT conditions.  Parity against a reference run is unpinned: CasADi/IPOPT are absent here.
"""
import ctypes as ct
import os
import re

import numpy as np
import pytest

from conftest import REF, ROOT, gpu_available

GOLD = os.path.join(ROOT, "tests", "golden", "nmpc_cstr.npz")


@pytest.fixture(scope="module")
def nl(pkg):
    return pkg.load_problem(pkg.example_path("cstr_nmpc.py"))


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def oracle_problem(name, overrides=None):
    """The checker's own reading of an example (oracle/nmpc_oracle.py:load_problem -> oracle/exnum.py): the oracle never sees the product's
    loader, stand-ins or tracer."""
    import warnings
    import nmpc_oracle as no
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return no.load_problem(name if os.path.isabs(name) else os.path.join(ROOT, "mpc-code_amd", "examples", name), overrides)


@pytest.fixture(scope="module")
def onl():
    return oracle_problem("cstr_nmpc.py")


def sample(p, n=6, seed=0):
    rng = np.random.default_rng(seed)
    x = p.x0_m + rng.normal(size=(n, p.nx)) * [0.02, 3.0, 0.02]
    u = p.u0 + rng.normal(size=(n, p.nu)) * [2.0, 0.02]
    d = p.dhat0 + rng.normal(size=(n, p.nd)) * [0.01, 0.01]
    return x, u, d


# ------------------------------------------------------------------------------------------------- host logic (CPU)
def test_example_is_classified_as_nonlinear(pkg, nl):
    from mpc_code_amd.nlproblem import NonlinearMPCProblem
    assert isinstance(nl, NonlinearMPCProblem)
    assert (nl.nx, nl.nu, nl.ny, nl.nd, nl.nxp, nl.N, nl.h, nl.Mx, nl.Nsim) == (3, 2, 2, 2, 3, 30, 0.2, 10, 201)
    assert nl.ycols == [0, 2] and nl.nw == 3 * 31 + 2 * 30
    assert np.array_equal(nl.schedules(3)["ysp"], np.tile([0.874317, 0.6528], (3, 1)))


def test_traced_model_equals_the_user_functions(nl):
    """The DAG evaluated with NumPy gives what the Ex-file's own Python functions give on floats."""
    import math
    from mpc_code_amd import symtrace as st
    x, u, d = sample(nl)
    for i in range(x.shape[0]):
        vals = nl._vals(x=x[i], u=u[i], d=d[i], t=0.3)
        got = np.array([float(v) for v in st.evaluate(nl.f, vals)])
        # the same balance equations written out directly
        area = math.pi * 0.219 ** 2; k = 7.2e10 * math.exp(-8750 / 350) * math.exp(-8750 * (1 / x[i, 1] - 1 / 350)) * x[i, 0]
        ref = np.array([d[i, 1] * (1.0 - x[i, 0]) / (area * x[i, 2]) - k,
                        d[i, 1] * (350 - x[i, 1]) / (area * x[i, 2]) + 5.0e4 / (1000.0 * 0.239) * k + 2 * (915.6 * 60 / 1000) / (0.219 * 1000.0 * 0.239) * (u[i, 0] - x[i, 1]),
                        (d[i, 1] - u[i, 1]) / area])
        assert np.allclose(got, ref, rtol=1e-13, atol=0)


def test_symbolic_jacobians_against_central_differences(nl):
    from mpc_code_amd import symtrace as st
    x, u, d = sample(nl, 3, seed=1)
    for i in range(3):
        f = lambda xx, uu, dd: np.array([float(v) for v in st.evaluate(nl.f, nl._vals(x=xx, u=uu, d=dd, t=0.0))])
        J = np.array([[float(st.evaluate([e], nl._vals(x=x[i], u=u[i], d=d[i], t=0.0))[0]) for e in row] for row in nl.f_x])
        for j in range(nl.nx):
            hstep = 1e-6 * max(1.0, abs(x[i, j])); e = np.zeros(nl.nx); e[j] = hstep
            fd = (f(x[i] + e, u[i], d[i]) - f(x[i] - e, u[i], d[i])) / (2 * hstep)
            assert np.allclose(J[:, j], fd, rtol=1e-6, atol=1e-7)
        Jd = np.array([[float(st.evaluate([e], nl._vals(x=x[i], u=u[i], d=d[i], t=0.0))[0]) for e in row] for row in nl.f_d])
        for j in range(nl.nd):
            hstep = 1e-7; e = np.zeros(nl.nd); e[j] = hstep
            fd = (f(x[i], u[i], d[i] + e) - f(x[i], u[i], d[i] - e)) / (2 * hstep)
            assert np.allclose(Jd[:, j], fd, rtol=1e-6, atol=1e-7)


def test_if_else_is_traced_as_a_select(nl):
    """The plant's feed flow steps at t = 5, 15, 25 (Ex_NMPC.py:55): the traced plant follows the schedule."""
    x = nl.x0_p[None]; u = nl.u0[None]
    a, b, c = (nl.plant_step(x, u, t)[0, 2] for t in (1.0, 8.0, 20.0))
    assert abs(a - x[0, 2]) < 1e-12 and b > a + 1e-3 and c < a - 1e-3      # level: steady, filling (0.15 in), draining (0.08 in)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_same_problem_as_the_reference_example(pkg, nl, onl):
    """The reference's Ex_NMPC.py loads unmodified and defines the same problem as examples/cstr_nmpc.py."""
    ref = pkg.load_problem(os.path.join(REF, "Ex_NMPC.py"), overrides={"N": 30})
    for k in ("Q", "R", "Qss", "Rss", "umin", "umax", "xmin", "xmax", "ymin", "ymax", "dmin", "dmax", "Q_kf", "R_kf", "P0", "x0_p", "x0_m",
              "u0", "dhat0", "umin_ss", "umax_ss", "xmin_ss", "xmax_ss", "ymin_ss", "ymax_ss"):
        assert np.array_equal(getattr(ref, k), getattr(nl, k)), k
    assert (ref.nx, ref.nu, ref.ny, ref.nd, ref.nxp, ref.h, ref.Mx, ref.Nsim, ref.ycols) == (nl.nx, nl.nu, nl.ny, nl.nd, nl.nxp, nl.h, nl.Mx, nl.Nsim, nl.ycols)
    import nmpc_oracle as no
    x, u, d = sample(nl, 4, seed=2)
    for i in range(4):
        oref = oracle_problem(os.path.join(REF, "Ex_NMPC.py"), {"N": 30})
        assert np.allclose(no.model_fx(oref, x[i], u[i], d[i]), no.model_fx(onl, x[i], u[i], d[i]), rtol=1e-13, atol=0)
        for t in (0.0, 7.0, 30.0):
            assert np.allclose(no.plant_fx(oref, x[i], u[i], t), no.plant_fx(onl, x[i], u[i], t), rtol=1e-13, atol=0)
    for k in ("Q", "R", "Qss", "Rss", "umin", "umax", "xmin", "xmax", "ymin", "ymax", "dmin", "dmax", "Q_kf", "R_kf", "P0", "x0_p", "x0_m", "u0", "dhat0"):
        assert np.array_equal(getattr(onl, k), getattr(nl, k)), k      # two independent loaders, one problem


def test_unsupported_nonlinear_features_are_refused(pkg, tmp_path):
    src = open(pkg.example_path("cstr_nmpc.py")).read()
    f = tmp_path / "ex_slacks.py"; f.write_text(src.replace("slacks = False", "slacks = True"))
    with pytest.raises(pkg.UnsupportedProblem):
        pkg.load_problem(str(f))
    f = tmp_path / "ex_lin.py"; f.write_text(src.replace('offree = "nl"', 'offree = "no"'))
    with pytest.raises(pkg.UnsupportedProblem):
        pkg.load_problem(str(f))


# ------------------------------------------------------------------------------------------------- oracle (CPU)
def test_oracle_reproduces_its_golden_rows(nl, onl, gold):
    import nmpc_oracle as no
    r = no.closed_loop(onl, 3, x0_p=gold["rti_x0"][1], x0_m=gold["rti_x0"][1], max_sqp=1)
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
        assert np.allclose(r[k], gold["rti_" + k][:3, 1], rtol=1e-10, atol=1e-10), k


def test_c_restatement_follows_the_golden_loops(onl, gold):
    """oracle/nmpc_oracle.c (hand-written CSTR, complex-step Jacobians, QPs by the null-space interior point method) against the vectors of
    the NumPy oracle (finite-difference Jacobians, dense Mehrotra + exact active-set polish): real-time iteration over 40 steps through the
    feed-flow change, and SQP iterated to the KKT point."""
    import nmpc_oracle_c as nc
    o = nc.OracleNC(onl)
    r = o.closed_loop(40, gold["rti_x0"], max_sqp=1)
    assert np.array_equal(r["STATUS_DYN"], gold["rti_STATUS_DYN"]) and np.array_equal(r["STATUS_SS"], gold["rti_STATUS_SS"])
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
        assert np.max(np.abs(r[k] - gold["rti_" + k]) / (1 + np.abs(gold["rti_" + k]))) < 1e-8, k
    ns = gold["sqp_U"].shape[0]
    r = o.closed_loop(ns, gold["rti_x0"][:gold["sqp_U"].shape[1]], max_sqp=50)
    assert np.max(np.abs(r["U"] - gold["sqp_U"]) / (1 + np.abs(gold["sqp_U"]))) < 1e-8 and int(r["STATUS_DYN"].max()) == 0


def test_restatements_agree_with_white_noise_on_the_measurement(nl, onl):
    """R_wn of the shipped example (Ex_NMPC.py:108; MPC_code.py:537-541: y_k += sqrtm(R_wn) N(0, I), unseeded there): with the same draws handed to both, the NumPy
    and the C restatement give the same loop; the noise is in the loop (the estimate moves) and the loader keeps the covariance."""
    import nmpc_oracle as no
    import nmpc_oracle_c as nc
    assert nl.R_wn is not None and np.array_equal(nl.R_wn, 1e-7 * np.eye(2))
    K = 8
    v = np.random.default_rng(11).standard_normal((K, 1, 2)) * np.sqrt(1e-7)
    a = no.closed_loop(onl, K, max_sqp=1, v_wn=v[:, 0])
    o = nc.OracleNC(onl)
    c, c0 = o.closed_loop(K, onl.x0_p[None], max_sqp=1, v_wn=v), o.closed_loop(K, onl.x0_p[None], max_sqp=1)
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
        assert np.max(np.abs(np.asarray(a[k]).reshape(c[k][:, 0].shape) - c[k][:, 0]) / (1 + np.abs(c[k][:, 0]))) < 1e-7, k
    assert np.abs(c["D_HAT"] - c0["D_HAT"]).max() > 1e-6


def test_golden_converged_rows_satisfy_the_nlp_kkt_conditions(nl, onl, gold):
    """Re-verify the certificate without trusting any solver: dynamics defect, stationarity and bounds of the NLP
    (Control_Calc.py:20-260) at the stored trajectories."""
    import nmpc_oracle as no
    assert np.all(gold["sqp_STATUS_DYN"] == 0)
    assert gold["sqp_KKT_DEFECT"].max() < 1e-10 and gold["sqp_KKT_STAT"].max() < 1e-8 and gold["sqp_KKT_VIOL"].max() < 1e-10
    k, b = 0, 1
    w = gold["sqp_W"][k, b]
    c = no.kkt_nlp(onl, w, w[:nl.nx], gold["sqp_XS"][k, b], gold["sqp_US"][k, b], gold["sqp_D_HAT"][k, b], 0.0)
    assert c["defect"] < 1e-10 and c["stationarity"] < 1e-8 and c["bound_violation"] < 1e-10


# ------------------------------------------------------------------------------------------------- generated code and the C-ABI (CPU)
@pytest.fixture(scope="module")
def nmpc_lib(nl):
    from mpc_code_amd import nlcodegen
    return nlcodegen.build_nmpc_library(nl)


def test_generated_header_is_straight_line_code(nl):
    from mpc_code_amd import nlcodegen
    text = nlcodegen.emit_model_header(nl)
    for fn in ("f(", "f_jac(", "h(", "h_jac(", "fp(", "hp("):
        assert "void " + fn in text
    assert "NX = 3, NU = 2, NY = 2, ND = 2, NXP = 3, MX = 10" in text
    assert text.count("exp(") <= 8          # shared sub-expressions: one Arrhenius term per function, not one per use
    assert "?" in text                      # the feed-flow schedule became selects


def test_nmpc_library_exports_the_header(nmpc_lib):
    from mpc_code_amd import nmpc
    hdr = open(os.path.join(ROOT, "include", "mpc_nmpc.h")).read()
    declared = set(re.findall(r"\b(nmpc_[a-z_]+)\s*\(", hdr))
    assert declared == set(nmpc.NMPC_EXPORTS), declared ^ set(nmpc.NMPC_EXPORTS)
    lib = ct.CDLL(nmpc_lib)
    for s in declared:
        assert hasattr(lib, s), s
    lib.nmpc_build_info.restype = ct.c_char_p
    assert lib.nmpc_build_info().decode() == "gfx950;nmpc;dims=3/2/2/2/3;mx=10"


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU error path")
def test_nmpc_create_fails_loudly_without_a_gpu(nl, nmpc_lib):
    from mpc_code_amd import nmpc
    from mpc_code_amd.capi import MpcAmdError
    with pytest.raises(MpcAmdError, match="no HIP device"):
        nmpc.NmpcSolver(nl, lib_path=nmpc_lib)


# ------------------------------------------------------------------------------------------------- GPU parity
@pytest.fixture(scope="module")
def solver(nl):
    from mpc_code_amd import nmpc
    s = nmpc.NmpcSolver(nl)
    yield s
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [1, 3, 4])
@pytest.mark.parametrize("mode,max_sqp", [("rti", 1), ("sqp", 50)])
def test_gpu_closed_loop_equals_the_golden_vectors(nl, gold, solver, mode, max_sqp, kernel):
    """Real-time iteration over 40 steps (feed-flow change at step 25 included) and converged SQP over 8 steps.
    Tolerance: 1e-9 relative to 1 + |value| per step would be the solver tolerance; errors carry through the closed
    loop, so 2e-7 over the whole run (measured: 3e-9 / 2e-10)."""
    from mpc_code_amd import nmpc
    x0 = gold[mode + "_x0"]; ns = gold[mode + "_U"].shape[0]
    solver.set_kernel(kernel)          # 1: one instance per lane; 3: wave-autonomous (lane = stage, QP on the matrix cores); 4: split pipeline (per step a lane-style and a wave-style launch)
    assert solver.get_kernel() == kernel
    r = nmpc.run_nmpc_closed_loop(nl, x0, x0, nsteps=ns, solver=solver, max_sqp=max_sqp, sqp_tol=1e-9)
    solver.set_kernel(0)
    assert np.array_equal(r["STATUS_DYN"], gold[mode + "_STATUS_DYN"]) and np.array_equal(r["STATUS_SS"], gold[mode + "_STATUS_SS"])
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT", "Yp"):
        g = gold[f"{mode}_{k}"]
        assert np.max(np.abs(r[k] - g) / (1.0 + np.abs(g))) < 2e-7, k
    if mode == "rti":
        assert np.all(r["SQP_DYN"] == 1)


@pytest.mark.gpu
def test_gpu_full_size_batch_properties(nl, onl, gold, solver):
    """BASELINE configs[3]: B = 16384, N = 30.  Instance 0 is the golden instance; the rest start in the box of SURVEY.md 8d cfg 3.
    Properties: every status solved, inputs and predicted states within their bounds, results independent of the position in
    the batch (bit for bit), the golden instance reproduced."""
    from mpc_code_amd import nmpc
    B, ns = 16384, 30
    rng = np.random.default_rng(7)
    x0 = np.tile(nl.x0_p, (B, 1)); x0[1:] *= 1.0 + 0.02 * rng.uniform(-1, 1, size=(B - 1, 3))      # SURVEY.md 8d cfg 3
    r = nmpc.run_nmpc_closed_loop(nl, x0, x0, nsteps=ns, solver=solver, max_sqp=1)
    assert np.all(r["STATUS_DYN"] == 0) and np.all(r["STATUS_SS"] == 0)
    assert all(np.isfinite(r[k]).all() for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"))
    assert np.all(r["U"] >= nl.umin - 1e-7) and np.all(r["U"] <= nl.umax + 1e-7)
    assert np.all(r["XS"] >= nl.xmin_ss - 1e-7) and np.all(r["XS"] <= nl.xmax_ss + 1e-7)
    g = gold["rti_U"][:ns, 0]
    assert np.max(np.abs(r["U"][:, 0] - g) / (1 + np.abs(g))) < 2e-7
    # three instances drawn from the batch, re-run by the oracle (NumPy restatement, finite-difference Jacobians, dense QPs) from
    # their own initial states: the batch's trajectories are the oracle's
    import nmpc_oracle as no
    for b in rng.choice(B, 3, replace=False):
        o = no.closed_loop(onl, 6, x0_p=x0[b], x0_m=x0[b], max_sqp=1)
        for k in ("U", "X_HAT", "XS", "Xp", "D_HAT"):
            assert np.max(np.abs(r[k][:6, b] - o[k]) / (1 + np.abs(o[k]))) < 1e-6, (int(b), k)
        assert np.array_equal(r["STATUS_DYN"][:6, b], o["STATUS_DYN"])
    # a whole slice of the batch - 2048 instances, all 30 steps, the feed-flow step at t = 5 included - against the C restatement on the host
    # cores (hand-written CSTR, complex-step Jacobians, null-space QP solves): values of every instance-step and every status word
    import nmpc_oracle_c as nc
    cb = 2048
    c = nc.OracleNC(onl).closed_loop(ns, x0[:cb], max_sqp=1, nthreads=64)
    assert np.array_equal(r["STATUS_DYN"][:, :cb], c["STATUS_DYN"]) and np.array_equal(r["STATUS_SS"][:, :cb], c["STATUS_SS"])
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
        e = np.max(np.abs(r[k][:, :cb] - c[k]) / (1 + np.abs(c[k])))
        assert e < 1e-6, (k, e)
    perm = rng.permutation(B)
    r2 = nmpc.run_nmpc_closed_loop(nl, x0[perm], x0[perm], nsteps=ns, solver=solver, max_sqp=1)
    for k in ("U", "X_HAT", "Xp", "D_HAT"):
        assert np.array_equal(r2[k], r[k][:, perm]), k
    # the other kernels on a part of the batch: the same closed loops
    assert solver.get_kernel() == 4                    # 16384 instances: the split pipeline
    solver.set_kernel(3)
    for kern in (3, 1):
        solver.set_kernel(kern)
        r3 = nmpc.run_nmpc_closed_loop(nl, x0[:4096], x0[:4096], nsteps=ns, solver=solver, max_sqp=1)
        assert np.array_equal(r3["STATUS_DYN"], r["STATUS_DYN"][:, :4096]), kern
        for k in ("U", "X_HAT", "Xp", "D_HAT", "XS"):
            assert np.max(np.abs(r3[k] - r[k][:, :4096]) / (1 + np.abs(r3[k]))) < 1e-7, (kern, k)
    solver.set_kernel(0)
    # the controlled loop contracts: 30 steps on, the level of every instance follows the same response to the feed-flow step
    # at t = 5, whatever its start in the box
    assert np.ptp(r["Xp"][-1, :, 2]) < 1e-3 and np.ptp(x0[:, 2]) > 0.02


@pytest.mark.gpu
def test_gpu_horizon_beyond_32_takes_the_unpaired_wave_kernels(pkg):
    """N = 40: the wave-style kernels hold one instance per set of lanes (N <= 32: two side by side).  A ragged batch, real-time
    iteration and SQP to the KKT point: the three kernels give the same closed loops, and the oracle's for one instance."""
    from mpc_code_amd import nmpc
    p40 = pkg.load_problem(pkg.example_path("cstr_nmpc.py"), overrides={"N": 40})
    assert p40.N == 40
    s = nmpc.NmpcSolver(p40)
    B = 301
    rng = np.random.default_rng(40)
    x0 = np.tile(p40.x0_p, (B, 1)); x0[1:] *= 1.0 + 0.02 * rng.uniform(-1, 1, size=(B - 1, 3))
    try:
        for max_sqp, ns in ((1, 8), (30, 3)):
            res = {}
            for kern in (1, 3, 4):
                s.set_kernel(kern)
                res[kern] = nmpc.run_nmpc_closed_loop(p40, x0, x0, nsteps=ns, solver=s, max_sqp=max_sqp, sqp_tol=1e-9)
            assert np.all(res[1]["STATUS_DYN"] == 0)
            for kern in (3, 4):
                assert np.array_equal(res[kern]["STATUS_DYN"], res[1]["STATUS_DYN"]) and np.array_equal(res[kern]["STATUS_SS"], res[1]["STATUS_SS"]), (max_sqp, kern)
                for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
                    assert np.max(np.abs(res[kern][k] - res[1][k]) / (1 + np.abs(res[1][k]))) < 1e-7, (max_sqp, kern, k)
            if max_sqp == 1:
                import nmpc_oracle as no
                o = no.closed_loop(oracle_problem("cstr_nmpc.py", {"N": 40}), 3, x0_p=x0[5], x0_m=x0[5], max_sqp=1)
                for k in ("U", "X_HAT", "XS", "Xp", "D_HAT"):
                    assert np.max(np.abs(res[4][k][:3, 5] - o[k]) / (1 + np.abs(o[k]))) < 1e-6, k
    finally:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,B,spread", [(1, 1003, 0.02), (6, 2050, 0.05)])      # (6: instance 449 took a NaN over from its lane neighbour 448 at step 1 before the sweeps dropped foreign values)
def test_gpu_kernels_agree_under_mismatch_holds_and_divergence(nl, solver, seed, B, spread):
    """Plant and model start apart, in a wide box, ragged batch: many steps are held (infeasible), some instances run into the iteration
    limit or diverge (NaN).  Every instance the lane kernel keeps finite and never holds must come out the same from the wave-style
    kernels - in particular next to a diverged neighbour (two instances share the lanes of a wave there: nothing may cross over)."""
    from mpc_code_amd import nmpc
    rng = np.random.default_rng(seed)
    x0 = nl.x0_p * (1.0 + spread * rng.uniform(-1, 1, size=(B, 3)))
    xm = nl.x0_p * (1.0 + spread * rng.uniform(-1, 1, size=(B, 3)))
    res = {}
    for kern in (1, 3, 4):
        solver.set_kernel(kern)
        res[kern] = nmpc.run_nmpc_closed_loop(nl, x0, xm, nsteps=6, solver=solver, max_sqp=1)
    solver.set_kernel(0)
    st = res[1]["STATUS_DYN"]
    with np.errstate(invalid="ignore"):
        ok = np.isfinite(res[1]["Xp"]).all(axis=(0, 2)) & (st == 0).all(axis=0)
    assert ok.sum() > B // 3 and (st == 2).sum() > B               # the scenario holds what it promises
    for kern in (3, 4):
        assert np.array_equal(res[kern]["STATUS_DYN"][:, ok], st[:, ok]), kern
        for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
            a, b = res[kern][k][:, ok], res[1][k][:, ok]
            assert np.isfinite(a).all(), (kern, k)
            assert np.max(np.abs(a - b) / (1 + np.abs(b))) < 1e-6, (kern, k)


@pytest.mark.gpu
def test_gpu_held_steps_equal_the_oracle(nl, onl, solver):
    """A level below its lower output bound: the OCP is infeasible from the measured state, the previous input is held and the model
    propagates the estimate (MPC_code.py:798-805) - every kernel against the oracle run of the same start, next to a healthy instance."""
    from mpc_code_amd import nmpc
    import nmpc_oracle as no
    x0 = np.array([[0.874317, 325.0, 0.47], [0.9, 330.0, 0.51]])
    o = [no.closed_loop(onl, 5, x0_p=x, x0_m=x, max_sqp=1) for x in x0]
    assert np.all(o[0]["STATUS_DYN"] == 2) and np.all(o[1]["STATUS_DYN"] == 0)
    for kern in (1, 3, 4):
        solver.set_kernel(kern)
        r = nmpc.run_nmpc_closed_loop(nl, x0, x0, nsteps=5, solver=solver, max_sqp=1)
        for b in range(2):
            assert np.array_equal(r["STATUS_DYN"][:, b], o[b]["STATUS_DYN"]) and np.array_equal(r["STATUS_SS"][:, b], o[b]["STATUS_SS"]), (kern, b)
            for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
                assert np.max(np.abs(r[k][:, b] - o[b][k]) / (1 + np.abs(o[b][k]))) < 1e-7, (kern, b, k)
    solver.set_kernel(0)


@pytest.mark.gpu
def test_gpu_launch_chunks_continue_the_same_loop(nl, solver):
    """Two launches of 5 steps continue the resident state exactly as one launch of 10."""
    B = 128
    rng = np.random.default_rng(3)
    x0 = np.tile(nl.x0_p, (B, 1)) + rng.uniform(-1, 1, size=(B, 3)) * [0.02, 2.0, 0.02]
    sched = nl.schedules(10)
    for kernel in (1, 3, 4):
        solver.set_kernel(kernel)
        solver.alloc(B, 10); solver.set_state(x0, x0); solver.set_schedule(sched)
        solver.run(0, 10, 1, 1e-9); solver.sync()
        a = solver.get_log("U").copy()
        solver.alloc(B, 10); solver.set_state(x0, x0); solver.set_schedule(sched)
        solver.run(0, 5, 1, 1e-9); solver.run(5, 5, 1, 1e-9); solver.sync()
        assert np.array_equal(solver.get_log("U"), a), kernel
    solver.set_kernel(0)


@pytest.mark.gpu
@pytest.mark.parametrize("max_sqp", [1, 20])
def test_gpu_the_three_solver_calls_of_a_step_reproduce_the_fused_loop(nl, solver, max_sqp):
    """The per-call seam of include/mpc_nmpc.h - nmpc_ekf_update, nmpc_target_solve, nmpc_ocp_solve: the reference's defEstimator(..., 'ekf'),
    solver_ss(...), solver(...) of one step (MPC_code.py:577-650, :704-709, :776-781) for the whole batch, caller-owned host arrays - with the device's
    plant in between is the fused closed loop of the instance-per-lane kernel BIT FOR BIT through the feed-flow step: values, status words, SQP and
    interior-point iteration counts; with a plant of the caller's (the Ex-file's, NumPy Runge-Kutta) the same loop to rounding."""
    from mpc_code_amd import nmpc
    B, ns = 70, 30
    rng = np.random.default_rng(4)
    x0 = np.tile(nl.x0_p, (B, 1)) * (1.0 + 0.02 * rng.uniform(-1, 1, size=(B, 3)))
    solver.set_kernel(1)
    a = nmpc.run_nmpc_closed_loop(nl, x0, x0, nsteps=ns, solver=solver, max_sqp=max_sqp, sqp_tol=1e-9)
    b = nmpc.run_nmpc_stepwise(nl, x0, x0, nsteps=ns, solver=solver, max_sqp=max_sqp, sqp_tol=1e-9)
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT", "STATUS_DYN", "STATUS_SS", "ITERS_DYN", "SQP_DYN", "SQP_SS"):
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(a[k].astype(float) - b[k].astype(float)).max()))
    c = nmpc.run_nmpc_stepwise(nl, x0[:8], x0[:8], nsteps=ns, solver=solver, max_sqp=max_sqp, sqp_tol=1e-9, plant=lambda x, u, t: nl.plant_step(x, u, t))
    assert np.max(np.abs(c["U"] - a["U"][:, :8]) / (1 + np.abs(c["U"]))) < 1e-8 and np.array_equal(c["STATUS_DYN"], a["STATUS_DYN"][:, :8])
    solver.set_kernel(0)


@pytest.mark.gpu
def test_gpu_white_noise_on_the_measurement_through_the_per_call_seam(nl, onl, solver):
    """The shipped example's R_wn (Ex_NMPC.py:108; MPC_code.py:537-541): in the loop through the per-call seam the measurement is the caller's, and
    run_nmpc_stepwise(noise_seed=...) adds sqrtm(R_wn) N(0, I) to it - the C restatement, handed the same draws, gives the same loop."""
    import nmpc_oracle_c as nc
    from mpc_code_amd import nmpc
    B, ns = 40, 25
    x0, xm = np.tile(nl.x0_p, (B, 1)), np.tile(nl.x0_m, (B, 1))      # the shipped start, forty noise histories
    r = nmpc.run_nmpc_stepwise(nl, x0, xm, nsteps=ns, solver=solver, max_sqp=1, sqp_tol=1e-9, noise_seed=7)
    r0 = nmpc.run_nmpc_stepwise(nl, x0, xm, nsteps=ns, solver=solver, max_sqp=1, sqp_tol=1e-9)
    assert int(r["STATUS_DYN"].max()) == 0 and np.isfinite(r["U"]).all() and np.abs(r["U"][:, 0] - r["U"][:, 1]).max() > 0      # the histories differ
    assert r["V_WN"].shape == (ns, B, 2) and abs(r["V_WN"].std() - np.sqrt(1e-7)) < 0.1 * np.sqrt(1e-7) and np.abs(r["D_HAT"] - r0["D_HAT"]).max() > 1e-6
    c = nc.OracleNC(onl).closed_loop(ns, x0, xm, max_sqp=1, nthreads=0, v_wn=r["V_WN"])
    assert np.array_equal(r["STATUS_DYN"], c["STATUS_DYN"]) and np.array_equal(r["STATUS_SS"], c["STATUS_SS"])
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
        assert np.max(np.abs(r[k] - c[k]) / (1 + np.abs(c[k]))) < 1e-6, (k, float(np.max(np.abs(r[k] - c[k]) / (1 + np.abs(c[k])))))
    # ... and in the RESIDENT loop (nmpc_set_noise: the draws live on the device, one row per step): the instance-per-lane kernel is the seam's loop to the bit, the
    # wave-autonomous kernel and the split pipeline follow the C restatement; switched off again, the loop is the deterministic one
    for kernel in (1, 3, 4):
        solver.set_kernel(kernel)
        a = nmpc.run_nmpc_closed_loop(nl, x0, xm, nsteps=ns, solver=solver, max_sqp=1, sqp_tol=1e-9, noise_seed=7)
        assert np.array_equal(a["V_WN"], r["V_WN"])
        for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
            if kernel == 1:
                assert np.array_equal(a[k], r[k]), k
            assert np.max(np.abs(a[k] - c[k]) / (1 + np.abs(c[k]))) < 1e-6, (kernel, k)
        assert np.array_equal(a["STATUS_DYN"], c["STATUS_DYN"])
    solver.set_kernel(1)
    b = nmpc.run_nmpc_closed_loop(nl, x0, xm, nsteps=ns, solver=solver, max_sqp=1, sqp_tol=1e-9)
    assert np.array_equal(b["U"], r0["U"])
    solver.set_kernel(0)


@pytest.mark.gpu
def test_gpu_error_paths(nl, solver):
    from mpc_code_amd.capi import MpcAmdError
    solver.alloc(4, 3)
    with pytest.raises(MpcAmdError, match="nmpc_set_state first"):
        solver.set_schedule(nl.schedules(3)); solver.run(0, 3)
    solver.set_state(np.tile(nl.x0_p, (4, 1)), np.tile(nl.x0_m, (4, 1)))
    with pytest.raises(MpcAmdError, match="outside the schedule"):
        solver.run(0, 4)
    with pytest.raises(MpcAmdError, match="max_sqp"):
        solver.run(0, 3, 0)


# ================================================================================================= discrete-time example (Ex_NMPC_dis.py)
QGOLD = os.path.join(ROOT, "tests", "golden", "nmpc_quadtank.npz")


@pytest.fixture(scope="module")
def qt(pkg):
    return pkg.load_problem(pkg.example_path("quadtank_nmpc_dis.py"))


@pytest.fixture(scope="module")
def qt_mild(pkg):
    from nmpc_cases import quadtank_mild_setpoints
    return pkg.load_problem(pkg.example_path("quadtank_nmpc_dis.py"), overrides={"defSP": quadtank_mild_setpoints})


def test_discrete_example_is_classified(qt):
    """User_fxm_Dis / User_fxp_Dis (the sampled map itself, Utilities.py:84-87,186-198), offree = 'lin' (+ Bd d, + Cd d), lue with K,
    cost and bounds on the input moves (S, Dumin/Dumax), Sss, User_vfin = dx' 100 dx, plant disturbance schedule def_pxp."""
    assert (qt.nx, qt.nu, qt.ny, qt.nd, qt.nxp, qt.N, qt.h) == (6, 2, 2, 2, 6, 20, 5.0)
    assert qt.discrete and qt.plant_discrete and qt.offree == "lin" and qt.estimator == "lue" and qt.DUForm and qt.DUssForm
    assert np.array_equal(qt.Dumin, [-50, -50]) and qt.ycols == [2, 3] and np.array_equal(qt.Pf, 200.0 * np.eye(6))
    assert np.array_equal(qt.K, np.vstack([np.zeros((6, 2)), np.eye(2)])) and np.array_equal(qt.Cd, np.eye(2))
    s = qt.schedules(12)
    assert np.array_equal(s["pxp"][3], [0, 0, 0.5, 0, 0, 0]) and s["ysp"][10, 1] == 12.1883 and s["ysp"][11, 1] == 6.0      # t = 55 > 50


def test_traced_discrete_model_equals_the_user_function(qt):
    """Tracing (SymMat: SX(n, 1) zeros, slice copies, element assignment, ** 0.5, nested if_else) against the same function run on
    numbers; the shipped start is a steady state of the sampled map; levels outside [0, 20] are clamped as the reference clamps them."""
    import nmpc_oracle as no
    from mpc_code_amd import symtrace as st
    oqt = oracle_problem("quadtank_nmpc_dis.py")
    rng = np.random.default_rng(0)
    for x in [qt.x0_m + rng.normal(size=6) * [3, 3, 2, 2, 0.5, 0.5] for _ in range(3)] + [np.array([50, 50, 25.0, -1.0, 0.2, 19.99])]:
        u = qt.u0 + rng.normal(size=2) * 5; d = rng.normal(size=2) * 0.1
        tr = np.array([float(v) for v in st.evaluate(qt.f, qt._vals(x=x, u=u, d=d, t=0.0))])
        assert np.allclose(tr, no.model_fx(oqt, x, u, d), rtol=1e-14, atol=1e-14)
    assert np.abs(no.model_fx(oqt, qt.x0_m, qt.u0, np.zeros(2)) - qt.x0_m).max() < 1e-5
    J = np.array([[float(st.evaluate([e], qt._vals(x=qt.x0_m, u=qt.u0, d=np.zeros(2), t=0.0))[0]) for e in row] for row in qt.f_x])
    A, B, G, _ = no.linearize(oqt, qt.x0_m, qt.u0, np.zeros(2))
    assert np.allclose(J, A, rtol=1e-6, atol=1e-8) and np.allclose(J[:2], 0.0) and np.allclose(G, qt.Bd)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_same_problem_as_the_reference_discrete_example(pkg, qt):
    ref = pkg.load_problem(os.path.join(REF, "Ex_NMPC_dis.py"), overrides={"N": 20})
    for k in ("Q", "R", "Qss", "Rss", "umin", "umax", "xmin", "xmax", "ymin", "ymax", "Dumin", "Dumax", "K", "Pf", "Bd", "Cd", "x0_p", "x0_m", "u0", "dhat0"):
        assert np.array_equal(getattr(ref, k), getattr(qt, k)), k
    assert (ref.estimator, ref.DUForm, ref.DUssForm, ref.ycols, ref.Nsim, ref.discrete) == (qt.estimator, qt.DUForm, qt.DUssForm, qt.ycols, qt.Nsim, qt.discrete)
    import nmpc_oracle as no
    rng = np.random.default_rng(1)
    for x in [qt.x0_m + rng.normal(size=6) * [3, 3, 2, 2, 0.5, 0.5] for _ in range(3)] + [np.array([50, 50, 25.0, -1.0, 0.2, 19.99]), np.array([50, 50, 19.9, 0.01, 22, 0.0])]:
        u = qt.u0 + rng.normal(size=2) * 5
        oref, oqt = oracle_problem(os.path.join(REF, "Ex_NMPC_dis.py"), {"N": 20}), oracle_problem("quadtank_nmpc_dis.py")
        assert np.allclose(no.model_fx(oref, x, u, np.zeros(2)), no.model_fx(oqt, x, u, np.zeros(2)), rtol=1e-13, atol=1e-13)       # clamping included
        assert np.allclose(no.plant_fx(oref, x, u, 100.0), no.plant_fx(oqt, x, u, 100.0), rtol=1e-13, atol=1e-13)
    for t in (10, 60, 1500, 2500, 3500, 4500, 6000):
        assert all(np.array_equal(a_, b_) for a_, b_ in zip(ref.defSP(t), qt.defSP(t))) and np.array_equal(ref.def_pxp(t)[0], qt.def_pxp(t)[0])


def test_discrete_oracle_reproduces_its_golden_rows(qt_mild):
    import nmpc_oracle as no
    g = np.load(QGOLD)
    from nmpc_cases import quadtank_mild_setpoints
    r = no.closed_loop(oracle_problem("quadtank_nmpc_dis.py", {"defSP": quadtank_mild_setpoints}), 2, x0_p=g["rti_x0"][1], x0_m=g["rti_x0"][1], max_sqp=1)
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
        assert np.allclose(r[k], g["rti_" + k][:2, 1], rtol=1e-10, atol=1e-10), k
    assert np.all(g["sqp_STATUS_DYN"] == 0) and np.nanmax(g["sqp_KKT_DEFECT"]) < 1e-8 and np.nanmax(g["sqp_KKT_STAT"]) < 1e-7 and np.nanmax(g["sqp_KKT_VIOL"]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("mode,max_sqp", [("rti", 1), ("sqp", 50)])
def test_gpu_discrete_example_equals_the_golden_vectors(qt_mild, mode, max_sqp):
    """Discrete user model, fixed-gain observer, stage form with the input move as input (cost S and bounds Dumin/Dumax on it, the u
    bounds on the carried input), terminal weight; real-time iteration over 16 steps with a set-point change, and 9 steps iterated
    to the NLP's KKT point (measured 3e-8 / 1e-11 relative)."""
    from mpc_code_amd import nmpc
    g = np.load(QGOLD)
    x0 = g[mode + "_x0"]; ns = g[mode + "_U"].shape[0]
    s = nmpc.NmpcSolver(qt_mild)
    assert s.build_info().endswith("discrete;duv")
    r = nmpc.run_nmpc_closed_loop(qt_mild, x0, x0, nsteps=ns, solver=s, max_sqp=max_sqp, sqp_tol=1e-9)
    s.close()
    assert np.array_equal(r["STATUS_DYN"], g[mode + "_STATUS_DYN"]) and np.array_equal(r["STATUS_SS"], g[mode + "_STATUS_SS"])
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT", "Yp"):
        gg = g[f"{mode}_{k}"]
        assert np.max(np.abs(r[k] - gg) / (1.0 + np.abs(gg))) < 1e-6, k
    assert np.array_equal(r["SQP_DYN"], g[mode + "_SQP_DYN"])          # the SQP takes the same number of iterations as the dense restatement


@pytest.mark.gpu
def test_gpu_discrete_example_shipped_schedule_properties(qt):
    """The shipped schedule (level 6 asked of tank 2 at t > 50: an upper tank runs empty in the prediction) for a batch: inputs and
    input moves within their bounds, levels within [0, 20], finite everywhere, every status solved or iteration limit."""
    from mpc_code_amd import nmpc
    B, ns = 1024, 24
    rng = np.random.default_rng(5)
    x0 = np.tile(qt.x0_p, (B, 1)); x0[1:, 2:] *= 1.0 + 0.05 * rng.uniform(-1, 1, size=(B - 1, 4))
    r = nmpc.run_nmpc_closed_loop(qt, x0, x0, nsteps=ns, max_sqp=1)
    assert all(np.isfinite(r[k]).all() for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"))
    assert np.isin(r["STATUS_DYN"], (0, 1)).all() and (r["STATUS_DYN"] == 0).mean() > 0.95
    U = r["U"]; dU = np.diff(np.concatenate([np.tile(qt.u0, (1, B, 1)), U]), axis=0)
    assert (U >= qt.umin - 1e-6).all() and (U <= qt.umax + 1e-6).all() and (dU >= qt.Dumin - 1e-6).all() and (dU <= qt.Dumax + 1e-6).all()
    assert (r["Xp"][:, :, 2:] > -1e-9).all() and (r["Xp"][:, :, 2:] < 20.0).all()
    assert (r["Xp"][-1, :, 3] < r["Xp"][10, :, 3] - 0.5).all()          # tank 2 is being drained towards its new set point


# ================================================================================================= two states, one input (examples/reactor_nmpc.py)
RGOLD = os.path.join(ROOT, "tests", "golden", "nmpc_reactor.npz")


@pytest.fixture(scope="module")
def rx(pkg):
    return pkg.load_problem(pkg.example_path("reactor_nmpc.py"))


def test_reactor_example_is_classified_and_its_model_is_the_reference_reactor(pkg, rx):
    """nx = 2, nu = 1: the smallest stage (the wave-style kernels take it: state <= 4, nu <= 2); continuous model, EKF, disturbances
    inside the model.  With the disturbances at zero the model is the reactor of the reference's Ex_ENMPC.py (:42-49,64-65)."""
    from mpc_code_amd import nlcodegen
    assert (rx.nx, rx.nu, rx.ny, rx.nd, rx.nxp, rx.N, rx.h) == (2, 1, 2, 2, 2, 25, 2.0)
    assert not rx.discrete and rx.offree == "nl" and rx.estimator == "ekf" and not rx.DUForm
    assert "#define MPC_NL_WAVE_FITS 1" in nlcodegen.emit_model_header(rx)
    from mpc_code_amd import symtrace as st
    x, u = np.array([0.4, 0.55]), np.array([0.7])
    want = [0.7 * (1.0 - 0.4) - 1.0 * 0.4, -0.7 * 0.55 + 1.0 * 0.4 - 0.05 * 0.55]
    assert np.allclose([float(v) for v in st.evaluate(rx.f, rx._vals(x=x, u=u, d=np.zeros(2), t=0.0))], want, rtol=1e-14)
    d = np.array([0.1, -0.05])      # the two disturbances: rate constant of the first reaction, dilution rate
    want_d = [0.65 * (1.0 - 0.4) - 1.1 * 0.4, -0.65 * 0.55 + 1.1 * 0.4 - 0.05 * 0.55]
    assert np.allclose([float(v) for v in st.evaluate(rx.f, rx._vals(x=x, u=u, d=d, t=0.0))], want_d, rtol=1e-14)
    if os.path.isdir(REF):          # the same balances as the reference's file, traced from its own function
        ref = pkg.load_exfile(os.path.join(REF, "Ex_ENMPC.py"))
        fx = ref["User_fxm_Cont"](st.symvec("x", 2), st.symvec("u", 1), st.symvec("d", 2), st.Sym.var("t"), None)
        vals = {"x[0]": 0.4, "x[1]": 0.55, "u[0]": 0.7, "d[0]": 0.0, "d[1]": 0.0, "t": 0.0}
        assert np.allclose([float(v) for v in st.evaluate(list(fx), vals)], want, rtol=1e-14)


def test_reactor_oracle_reproduces_its_golden_rows(rx):
    import nmpc_oracle as no
    g = np.load(RGOLD)
    r = no.closed_loop(oracle_problem("reactor_nmpc.py"), 3, x0_p=g["rti_x0"][2], x0_m=g["rti_xm"][2], max_sqp=1)
    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
        assert np.allclose(r[k], g["rti_" + k][:3, 2], rtol=1e-10, atol=1e-10), k
    assert np.all(g["rti_STATUS_DYN"] == 0) and np.all(g["sqp_STATUS_DYN"] == 0)
    assert np.nanmax(g["sqp_KKT_DEFECT"]) < 1e-10 and np.nanmax(g["sqp_KKT_STAT"]) < 1e-9 and np.nanmax(g["sqp_KKT_VIOL"]) < 1e-12
    # offset-free: after the transient the product concentration sits on its set point although the plant's rate constant is not the model's
    assert abs(g["rti_Xp"][20, 0, 1] - 0.56) < 1e-3 and abs(g["rti_Xp"][29, 0, 1] - 0.48) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [1, 3, 4])
@pytest.mark.parametrize("mode,max_sqp", [("rti", 1), ("sqp", 50)])
def test_gpu_reactor_equals_the_golden_vectors(rx, mode, max_sqp, kernel):
    """One input (the single-column forms of the solver's input block) on the time-varying tables, plant and model starting apart,
    a set-point change inside the run: every kernel against the oracle's vectors."""
    from mpc_code_amd import nmpc
    g = np.load(RGOLD)
    s = nmpc.NmpcSolver(rx)
    try:
        s.set_kernel(kernel)
        ns = g[mode + "_U"].shape[0]
        r = nmpc.run_nmpc_closed_loop(rx, g[mode + "_x0"], g[mode + "_xm"], nsteps=ns, solver=s, max_sqp=max_sqp, sqp_tol=1e-9)
        assert np.array_equal(r["STATUS_DYN"], g[mode + "_STATUS_DYN"]) and np.array_equal(r["STATUS_SS"], g[mode + "_STATUS_SS"])
        for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
            gk = g[f"{mode}_{k}"]
            assert np.max(np.abs(r[k] - gk) / (1.0 + np.abs(gk))) < 2e-7, k
    finally:
        s.close()


@pytest.mark.gpu
def test_gpu_reactor_kernels_agree_on_a_ragged_batch(rx):
    from mpc_code_amd import nmpc
    B, ns = 1030, 24
    rng = np.random.default_rng(8)
    x0 = rx.x0_p + 0.05 * rng.uniform(-1, 1, size=(B, 2)); xm = x0 + 0.02 * rng.uniform(-1, 1, size=(B, 2))
    s = nmpc.NmpcSolver(rx)
    try:
        res = {}
        for kern in (1, 3, 4):
            s.set_kernel(kern)
            res[kern] = nmpc.run_nmpc_closed_loop(rx, x0, xm, nsteps=ns, solver=s, max_sqp=1)
        assert np.all(res[1]["STATUS_DYN"] == 0) and np.all(res[1]["U"] >= rx.umin - 1e-9) and np.all(res[1]["U"] <= rx.umax + 1e-9)
        for kern in (3, 4):
            assert np.array_equal(res[kern]["STATUS_DYN"], res[1]["STATUS_DYN"])
            for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
                assert np.max(np.abs(res[kern][k] - res[1][k]) / (1 + np.abs(res[1][k]))) < 1e-7, (kern, k)
    finally:
        s.close()


@pytest.mark.gpu
def test_gpu_reactor_saturated_disturbance_and_sqp_limit_equal_the_oracle(pkg):
    """The disturbance estimate clipped to [dmin, dmax] (MPC_code.py:655-668; the clipped value is the next step's prior) and SQP
    stopped by its iteration limit (status 1, accepted like IPOPT's, :786): every kernel against oracle runs of the same starts."""
    from mpc_code_amd import nmpc
    import nmpc_oracle as no
    p = pkg.load_problem(pkg.example_path("reactor_nmpc.py"), overrides={"dmin": -0.05 * np.ones((2, 1)), "dmax": 0.05 * np.ones((2, 1))})
    x0 = np.array([[0.45, 0.50], [0.47, 0.53]]); xm = np.array([[0.45, 0.50], [0.44, 0.50]])
    s = nmpc.NmpcSolver(p)
    try:
        for max_sqp, ns in ((1, 12), (3, 4)):
            op = oracle_problem("reactor_nmpc.py", {"dmin": -0.05 * np.ones((2, 1)), "dmax": 0.05 * np.ones((2, 1))})
            o = [no.closed_loop(op, ns, x0_p=a, x0_m=b, max_sqp=max_sqp, sqp_tol=1e-9) for a, b in zip(x0, xm)]
            if max_sqp == 1:
                assert o[0]["D_HAT"][-1, 0] == 0.05
            else:
                assert o[0]["STATUS_DYN"][0] == 1
            for kern in (1, 3, 4):
                s.set_kernel(kern)
                r = nmpc.run_nmpc_closed_loop(p, x0, xm, nsteps=ns, solver=s, max_sqp=max_sqp, sqp_tol=1e-9)
                for b in range(2):
                    assert np.array_equal(r["STATUS_DYN"][:, b], o[b]["STATUS_DYN"]) and np.array_equal(r["SQP_DYN"][:, b], o[b]["SQP_DYN"]), (max_sqp, kern, b)
                    for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
                        assert np.max(np.abs(r[k][:, b] - o[b][k]) / (1 + np.abs(o[b][k]))) < 1e-7, (max_sqp, kern, b, k)
    finally:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ex,N", [("reactor_nmpc.py", 2), ("reactor_nmpc.py", 32), ("reactor_nmpc.py", 33), ("reactor_nmpc.py", 64), ("cstr_nmpc.py", 32), ("cstr_nmpc.py", 3)])
def test_gpu_edge_horizons_kernels_agree(pkg, ex, N):
    """The shortest horizons, both sides of the paired / unpaired switch of the wave-style kernels (N = 32: the last block of one
    instance sits in the lane next to the first block of its neighbour) and the longest they take."""
    from mpc_code_amd import nmpc
    p = pkg.load_problem(pkg.example_path(ex), overrides={"N": N})
    s = nmpc.NmpcSolver(p)
    B = 203
    rng = np.random.default_rng(N)
    x0 = p.x0_p * (1.0 + 0.02 * rng.uniform(-1, 1, size=(B, p.nx))); xm = x0 * (1.0 + 0.005 * rng.uniform(-1, 1, size=(B, p.nx)))
    try:
        res = {}
        for kern in (1, 3, 4):
            s.set_kernel(kern)
            res[kern] = nmpc.run_nmpc_closed_loop(p, x0, xm, nsteps=6, solver=s, max_sqp=1)
        ok = (res[1]["STATUS_DYN"] == 0).all(axis=0)
        assert ok.sum() > B // 2
        for kern in (3, 4):
            assert np.array_equal(res[kern]["STATUS_DYN"], res[1]["STATUS_DYN"]) and np.array_equal(res[kern]["STATUS_SS"], res[1]["STATUS_SS"]), kern
            for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
                assert np.max(np.abs(res[kern][k][:, ok] - res[1][k][:, ok]) / (1 + np.abs(res[1][k][:, ok]))) < 1e-8, (kern, k)
    finally:
        s.close()


@pytest.mark.gpu
def test_gpu_horizon_beyond_64_runs_on_the_lane_kernel_only(pkg):
    from mpc_code_amd import nmpc
    from mpc_code_amd.capi import MpcAmdError
    p = pkg.load_problem(pkg.example_path("reactor_nmpc.py"), overrides={"N": 70})
    s = nmpc.NmpcSolver(p)
    try:
        x0 = np.tile(p.x0_p, (5, 1))
        s.alloc(5, 3); s.set_schedule(p.schedules(3)); s.set_state(x0, x0)
        assert s.get_kernel() == 1
        s.run(0, 3); s.sync()
        assert np.all(s.get_log("STATUS_DYN")[:3] == 0)
        for kern in (3, 4):
            with pytest.raises(MpcAmdError, match="N <= 64"):
                s.set_kernel(kern)
        assert s.get_kernel() == 1
    finally:
        s.close()


@pytest.mark.gpu
def test_gpu_discrete_example_plant_and_model_apart_equals_the_oracle(qt_mild):
    """The quadruple tank (input-move form, Luenberger observer, lane kernel) with the model's levels started away from the plant's:
    the first OCP's guess is the model state before the measurement update (MPC_code.py:740-756)."""
    from mpc_code_amd import nmpc
    import nmpc_oracle as no
    rng = np.random.default_rng(4)
    x0 = np.tile(qt_mild.x0_p, (2, 1)); xm = x0.copy(); xm[:, 2:] += rng.uniform(-1, 1, (2, 4)) * [0.5, 0.5, 0.2, 0.2]
    s = nmpc.NmpcSolver(qt_mild)
    try:
        from nmpc_cases import quadtank_mild_setpoints
        oqm = oracle_problem("quadtank_nmpc_dis.py", {"defSP": quadtank_mild_setpoints})
        o = [no.closed_loop(oqm, 4, x0_p=a, x0_m=b, max_sqp=1) for a, b in zip(x0, xm)]
        r = nmpc.run_nmpc_closed_loop(qt_mild, x0, xm, nsteps=4, solver=s, max_sqp=1)
        for b in range(2):
            assert np.array_equal(r["STATUS_DYN"][:, b], o[b]["STATUS_DYN"])
            for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
                assert np.max(np.abs(r[k][:, b] - o[b][k]) / (1 + np.abs(o[b][k]))) < 1e-7, (b, k)
    finally:
        s.close()


def _reactor_pxp(t):
    return [np.array([0.002, -0.003])] if 4 <= t <= 12 else [np.zeros(2)]


def _reactor_pyp(t):
    return [np.array([0.0, 0.004])] if t >= 8 else [np.zeros(2)]


@pytest.mark.gpu
def test_gpu_reactor_plant_and_measurement_disturbance_schedules_equal_the_oracle(pkg):
    """def_pxp / def_pyp (MPC_code.py:512-515: added to the plant's next state and to the measurement) on every kernel."""
    from mpc_code_amd import nmpc
    import nmpc_oracle as no
    p = pkg.load_problem(pkg.example_path("reactor_nmpc.py"), overrides={"def_pxp": _reactor_pxp, "def_pyp": _reactor_pyp})
    assert np.array_equal(p.schedules(4)["pxp"][2], [0.002, -0.003]) and np.array_equal(p.schedules(6)["pyp"][4], [0.0, 0.004])
    x0 = np.array([[0.45, 0.50], [0.47, 0.52]]); xm = np.array([[0.45, 0.50], [0.46, 0.51]])
    op = oracle_problem("reactor_nmpc.py", {"def_pxp": _reactor_pxp, "def_pyp": _reactor_pyp})
    o = [no.closed_loop(op, 8, x0_p=a, x0_m=b, max_sqp=1) for a, b in zip(x0, xm)]
    s = nmpc.NmpcSolver(p)
    try:
        for kern in (1, 3, 4):
            s.set_kernel(kern)
            r = nmpc.run_nmpc_closed_loop(p, x0, xm, nsteps=8, solver=s, max_sqp=1)
            for b in range(2):
                assert np.array_equal(r["STATUS_DYN"][:, b], o[b]["STATUS_DYN"])
                for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT", "Yp"):
                    assert np.max(np.abs(r[k][:, b] - o[b][k]) / (1 + np.abs(o[b][k]))) < 1e-7, (kern, b, k)
    finally:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 3, 10])
def test_gpu_short_horizons_with_several_sqp_iterations_follow_the_c_restatement(pkg, N):
    """Several SQP iterations per OCP at horizons so short that the wave kernels' exchange area is larger than the lane kernel's slab of
    linearisations (N * NLIN < 64 (nu + nx): the buffer both share used to be sized for the slab only, round-2 advisor finding): every
    kernel against the C restatement - statuses, SQP iteration counts, values - and the buffers behind the shared one intact (a second run
    on the same handle gives the same numbers)."""
    import nmpc_oracle_c as nc
    from mpc_code_amd import nmpc
    p = pkg.load_problem(pkg.example_path("cstr_nmpc.py"), overrides={"N": N})
    q = oracle_problem("cstr_nmpc.py", {"N": N})
    B = 37
    rng = np.random.default_rng(100 + N)
    x0 = p.x0_p * (1.0 + 0.02 * rng.uniform(-1, 1, size=(B, p.nx)))
    c = nc.OracleNC(q).closed_loop(5, x0, x0, max_sqp=3, nthreads=8)
    s = nmpc.NmpcSolver(p)
    try:
        for kern in (1, 3, 4):
            s.set_kernel(kern)
            r = nmpc.run_nmpc_closed_loop(p, x0, x0, nsteps=5, solver=s, max_sqp=3)
            r2 = nmpc.run_nmpc_closed_loop(p, x0, x0, nsteps=5, solver=s, max_sqp=3)
            assert np.array_equal(r["U"], r2["U"]) and np.array_equal(r["STATUS_DYN"], r2["STATUS_DYN"]), kern
            # (held steps equal; 'converged within three SQP iterations' - a step norm against 1e-9 - may fall either way at the threshold)
            assert np.array_equal(r["STATUS_DYN"] == 2, c["STATUS_DYN"] == 2) and np.array_equal(r["STATUS_SS"], c["STATUS_SS"]), (kern, N)
            assert (r["STATUS_DYN"] != c["STATUS_DYN"]).mean() < 0.1, (kern, N, int((r["STATUS_DYN"] != c["STATUS_DYN"]).sum()))
            ok = (c["STATUS_DYN"] != 2).all(axis=0)      # (status 1 = SQP iteration limit: accepted; a held step ends the comparison of that instance)
            assert ok.sum() > B // 2
            for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT"):
                assert np.max(np.abs(r[k][:, ok] - c[k][:, ok]) / (1 + np.abs(c[k][:, ok]))) < 1e-6, (kern, N, k)
    finally:
        s.close()


@pytest.mark.gpu
def test_gpu_stream_groups_of_the_split_pipeline_do_not_change_the_loop(pkg):
    """nmpc_set_groups: the batch in one, two or three parts on streams of their own (ragged last part) - the same loop bit for bit."""
    from mpc_code_amd import nmpc
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = pkg.load_problem(pkg.example_path("cstr_nmpc.py"), overrides={"N": 30})
    B = 9001
    rng = np.random.default_rng(9)
    x0 = p.x0_p * (1.0 + 0.02 * rng.uniform(-1, 1, size=(B, p.nx)))
    s = nmpc.NmpcSolver(p)
    try:
        s.set_kernel(4)
        res = {}
        for G in (1, 2, 3, 0):
            s.set_groups(G)
            res[G] = nmpc.run_nmpc_closed_loop(p, x0, x0, nsteps=5, solver=s, max_sqp=1)
        for G in (2, 3, 0):
            for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT", "STATUS_DYN", "STATUS_SS", "ITERS_DYN"):
                assert np.array_equal(res[G][k], res[1][k]), (G, k)
        with pytest.raises(Exception):
            s.set_groups(4)
    finally:
        s.close()
