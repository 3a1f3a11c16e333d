"""Affine user inequality rows of the OCP on the linear path (the reference's `User_g_ineq`, Control_Calc.py:94-100,132-147; MPC_code.py:306-314): the loader reads the rows
off the Ex-file's function (affine in x, u, y, d; anything else is refused), the dense statement of oracle/mpc_oracle.py:ocp_qp carries them as rows of `G`, the product as one
more stage state per row, w_{k+1} = Gx x_k + Gu u_k + const <= 0 (csrc/mpc_amd.hip:build_problem) - on every loop kernel, the rows being ordinary bounded stage states."""
import numpy as np
import pytest

import mpc_oracle as mo

ROWS_EX = "cstr_lmpc_rows.py"


@pytest.fixture(scope="module")
def rows(pkg):
    return pkg.load_problem(pkg.example_path(ROWS_EX))


def test_loader_reads_affine_rows_and_refuses_the_rest(pkg, rows):
    from mpc_code_amd.problem import UnsupportedProblem
    assert rows.n_user_rows == 2
    assert np.allclose(rows.Gu, [[1.0, 0.5], [-0.02, 0.0]]) and np.allclose(rows.g0, [-5.0, -0.2])
    assert np.allclose(rows.Gx, np.vstack([np.zeros(3), rows.C[0]]))      # y_0 = C_0 x (+ Cd d + fy_const: zero in this row) is substituted
    base = pkg.example_path("cstr_lmpc.py")
    with pytest.raises(UnsupportedProblem, match="affine"):
        pkg.load_problem(base, overrides={"User_g_ineq": lambda x, u, y, d, t, px, py: u[0] * x[0] - 1.0})
    with pytest.raises(UnsupportedProblem, match="affine"):
        pkg.load_problem(base, overrides={"User_g_ineq": lambda x, u, y, d, t, px, py: u[0] + t - 1.0})
    with pytest.raises(UnsupportedProblem):
        pkg.load_problem(base, overrides={"User_h_eq": lambda x, u, y, d, t, px, py: u[0] - 1.0})
    with pytest.raises(UnsupportedProblem):
        pkg.load_problem(pkg.example_path(ROWS_EX), overrides={"TermCons": True})


def test_dense_statement_holds_the_rows(rows, pkg):
    """the oracle's optimum satisfies the rows at every stage, some of them with equality, and differs from the problem without rows"""
    plain = pkg.load_problem(pkg.example_path("cstr_lmpc.py"))
    xhat = np.array([0.2416, -0.6318, 3.0]); xs = np.array([0.2, 4.9176, 0.0]); us = np.array([1.6374, 0.0]); dhat = np.array([0.1752, -1.0389, 0.0]); up = np.zeros(2)      # (the shipped scenario at its fourth step)
    o = mo.ocp_solve_exact(rows, xhat, xs, us, dhat, up, tol=1e-9)
    o0 = mo.ocp_solve_exact(plain, xhat, xs, us, dhat, up, tol=1e-9)
    assert o["status"] == 0 and o0["status"] == 0
    n, m, N = rows.nx, rows.nu, rows.N
    W = o["w"][: (n + m) * N].reshape(N, n + m)
    g = W[:, :n] @ rows.Gx.T + W[:, n:] @ rows.Gu.T + rows.g0
    assert g.max() < 1e-7 and (np.abs(g) < 1e-7).sum() >= 1 and np.abs(o["u0"] - o0["u0"]).max() > 1e-3


@pytest.mark.gpu
def test_gpu_ocp_with_user_rows_matches_the_dense_statement(pkg, rows):
    from mpc_code_amd import capi
    rng = np.random.default_rng(11)
    B = 40
    xhat = np.array([0.2416, -0.6318, 3.0]) + rng.uniform(-1.0, 1.0, size=(B, 3)) * np.array([0.05, 1.0, 0.5]); dhat = np.array([0.1752, -1.0389, 0.0]) + 0.02 * rng.normal(size=(B, 3))
    xs = np.array([0.2, 4.9176, 0.0]) + 0.05 * rng.normal(size=(B, 3)); us = np.array([1.6374, 0.0]) + 0.1 * rng.normal(size=(B, 2)); up = np.zeros((B, 2))      # (around the shipped scenario's fourth step, where the rows bind)
    s = capi.Solver(rows)
    try:
        n_active = 0
        for kern in (1, 3):      # the lane solver and the wave-autonomous one
            s.set_option("ocp_kernel", kern)
            r = s.ocp_solve(xhat, xs, us, dhat, up, want_w=True)
            for b in range(0, B, 5):
                o = mo.ocp_solve_exact(rows, xhat[b], xs[b], us[b], dhat[b], up[b], tol=1e-9)
                if o["status"] != 0:
                    assert r["status"][b] == 2, (kern, b); continue
                assert r["status"][b] == 0, (kern, b)
                if o["exact"]:      # a certified optimum (active set verified, KKT residual at rounding): BASELINE's bound on u*
                    assert np.abs(r["u0"][b] - o["u0"]).max() < 1e-6 and np.abs(r["x1"][b] - o["x1"]).max() < 3e-6, (kern, b, np.abs(r["u0"][b] - o["u0"]).max())
                else:               # the polish did not verify (a row active with a zero multiplier): the oracle's point is its interior point method's, some 1e-3 off along the
                    H, g, E, e, G, lo, hi = mo.ocp_qp(rows, xhat[b], xs[b], us[b], dhat[b], up[b])      # flat direction - the kernel's point has to be feasible and at least as good
                    w = r["w"][b]
                    cost = lambda v: 0.5 * v @ H @ v + g @ v
                    assert np.abs(E @ w - e).max() < 1e-9 and (G @ w - hi).max() < 1e-9 and (lo - G @ w).max() < 1e-9, (kern, b)
                    assert cost(w) <= cost(o["w"]) + 1e-9 * abs(cost(o["w"])) and np.abs(r["u0"][b] - o["u0"]).max() < 5e-3, (kern, b, cost(w) - cost(o["w"]))
                n_active += int(np.abs(rows.Gx @ xhat[b] + rows.Gu @ r["u0"][b] + rows.Gd @ dhat[b] + rows.g0).min() < 1e-6)
        assert n_active >= 2      # (rows that bind at stage 0 on some of the instances)
    finally:
        s.close()


@pytest.mark.gpu
def test_gpu_closed_loop_with_user_rows_follows_the_oracle_on_every_loop_kernel(pkg, rows):
    """the shipped scenario with the rows (examples/cstr_lmpc_rows.py): both bind along the loop; lane, horizon-parallel and wave-autonomous kernels against the dense oracle's loop"""
    from mpc_code_amd import capi, driver
    ns = 9
    o = mo.closed_loop(rows, ns, tol=1e-9, ocp=mo.ocp_solve_exact)      # (every OCP from the fourth step on to a certified optimum; the first three are infeasible, as the shipped scenario's)
    assert all(o["EXACT_DYN"][3:])
    U = np.array(o["U"])
    assert np.abs(U[3:5, 0] + 0.5 * U[3:5, 1] - 5.0).max() < 1e-6      # the first row binds at steps 3 and 4
    s = capi.Solver(rows)
    try:
        for lk in (1, 2, 3):
            s.set_option("loop_kernel", lk)
            r = driver.run_closed_loop(rows, nsteps=ns, solver=s)
            assert r["STATUS_DYN"][:, 0].tolist() == list(o["STATUS_DYN"]), lk
            for k, tol in (("U", 5e-6), ("X_HAT", 5e-6), ("XS", 2e-5), ("US", 2e-5)):      # (the targets: the oracle's loop solves them by its interior point method without the polish)
                assert np.abs(r[k][:, 0] - np.array(o[k])).max() < tol, (lk, k, np.abs(r[k][:, 0] - np.array(o[k])).max())
            assert (r["U"][:, 0, 0] + 0.5 * r["U"][:, 0, 1] - 5.0).max() < 1e-6
    finally:
        s.close()
