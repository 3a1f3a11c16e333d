"""Pin the oracle: known answers, KKT certificates, independent solvers, golden vectors."""
import os

import numpy as np
import pytest

import mpc_oracle as o
import riccati_np as rn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lqr_known_answer(cstr):
    """No active bound, P = DARE  =>  u0* = us + K (xhat - xs) for any N."""
    K = o.lqr_gain(cstr)
    assert np.allclose(K, [[1.457547e-2, -3.157928e-4, 7.022363e-4], [-1.439759e-4, -1.051276e-5, 2.850066]], rtol=2e-6)
    rng = np.random.default_rng(1)
    for _ in range(3):
        xs, us, d = np.zeros(3), np.zeros(2), np.zeros(3)
        xh = rng.uniform(-0.05, 0.05, 3) * [1, 10, 1]
        r = o.ocp_solve(cstr, xh, xs, us, d, us)
        assert r["status"] == 0 and o.kkt_max(r["res"]) < 1e-9
        assert np.allclose(r["u0"], us + K @ (xh - xs), atol=1e-9)


def test_dense_ipm_against_scipy_trust_constr(cstr):
    """Independent third opinion on a short-horizon instance with active bounds."""
    from scipy.optimize import Bounds, LinearConstraint, minimize
    import copy
    p = copy.copy(cstr); p.N = 6
    xh, xs, us, d = np.array([0.3, 6.0, 2.0]), np.zeros(3), np.zeros(2), np.zeros(3)
    H, g, E, e, G, lo, hi = o.ocp_qp(p, xh, xs, us, d, us)
    r = o.ocp_solve_exact(p, xh, xs, us, d, us)
    assert r["status"] == 0 and r["exact"] and o.kkt_max(r["res"]) < 1e-10
    sol = minimize(lambda w: 0.5 * w @ H @ w + g @ w, r["w"] * 0, jac=lambda w: H @ w + g, hess=lambda w: H, method="trust-constr",
                   constraints=[LinearConstraint(E, e, e), LinearConstraint(G, lo, hi)], options=dict(gtol=1e-10, xtol=1e-12, maxiter=3000))
    assert np.abs(sol.x[3:5] - r["u0"]).max() < 1e-5
    assert (0.5 * r["w"] @ H @ r["w"] + g @ r["w"]) <= sol.fun + 1e-9


def test_shipped_initial_state_is_infeasible(cstr):
    """xhat = [3,3,3]: x_1[1] >= 12.3 > 10 whatever u; HiGHS agrees."""
    xh = np.full(3, 3.0)
    H, g, E, e, G, lo, hi = o.ocp_qp(cstr, xh, xh, np.zeros(2), np.zeros(3), np.zeros(2))
    assert not o.lp_feasible(E, e, G, lo, hi)
    assert o.ocp_solve(cstr, xh, xh, np.zeros(2), np.zeros(3), np.zeros(2))["status"] == 2
    sd = rn.stage_data(cstr)
    r = rn.rpdip_solve(sd, rn.instance_data(cstr, sd, xh, xh, np.zeros(2), np.zeros(3), np.zeros(2)))
    assert r["status"][0] == 2 and r["iters"][0] < 20


def test_infeasibility_labels_match_highs(cstr):
    rng = np.random.default_rng(7)
    x0 = 3 + rng.uniform(-1, 1, size=(60, 3))
    z3, z2 = np.zeros((60, 3)), np.zeros((60, 2))
    sd = rn.stage_data(cstr)
    r = rn.rpdip_solve(sd, rn.instance_data(cstr, sd, x0, z3, z2, z3, z2))
    for i in range(60):
        H, g, E, e, G, lo, hi = o.ocp_qp(cstr, x0[i], z3[i], z2[i], z3[i], z2[i])
        assert (r["status"][i] != 2) == o.lp_feasible(E, e, G, lo, hi), i
    assert (r["status"] == 1).sum() == 0 and r["iters"].max() < 40


@pytest.mark.parametrize("name", ["cstr_shipped", "wb_shipped", "cstr_box"])
def test_golden_vectors_certify_themselves(name, cstr, wb):
    """Re-derive the KKT residual of every solved golden OCP from the stated QP (no solver involved)."""
    p = wb if name.startswith("wb") else cstr
    g = np.load(os.path.join(GOLD, name + ".npz"))
    nst, nin = g["U"].shape[:2]
    checked = 0
    for k in range(0, nst, max(1, nst // 12)):
        for i in range(min(nin, 4)):
            if g["STATUS_DYN"][k, i] != 0 or not g["EXACT_DYN"][k, i]:
                continue
            r = o.ocp_solve_exact(p, g["XHAT_C"][k, i], g["XS"][k, i], g["US"][k, i], g["D_HAT"][k, i], g["U_PREV"][k, i])
            assert r["status"] == 0 and o.kkt_max(r["res"]) < 1e-9
            assert np.abs(r["u0"] - g["U"][k, i]).max() < 1e-9
            checked += 1
    assert checked >= 8


@pytest.mark.parametrize("name", ["cstr_shipped", "wb_shipped", "cstr_box"])
def test_riccati_restatement_reproduces_golden_ocps(name, cstr, wb):
    """The structure-exploiting algorithm (NumPy statement) hits the exact optimum of every certified golden OCP."""
    p = wb if name.startswith("wb") else cstr
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sh = g["U"].shape[:2]
    flat = lambda a: a.reshape((sh[0] * sh[1],) + a.shape[2:])
    sd = rn.stage_data(p)
    r = rn.rpdip_solve(sd, rn.instance_data(p, sd, flat(g["XHAT_C"]), flat(g["XS"]), flat(g["US"]), flat(g["D_HAT"]), flat(g["U_PREV"])))
    st, exact = flat(g["STATUS_DYN"]), flat(g["EXACT_DYN"]).astype(bool)
    assert np.array_equal(r["status"] == 2, st == 2)
    ok = (st == 0) & exact
    err = np.abs(r["u0"] - flat(g["U"]))[ok].max(axis=1)
    assert ok.sum() > 0.8 * (st == 0).sum()
    # tolerance: an interior-point answer is within ~s of the optimum for active bounds but only
    # ~sqrt(s*lambda) for degenerate ones, which dominate the shipped CSTR run once the target sits on a bound
    lim = dict(cstr_shipped=(1e-6, 5e-7, 1e-8), wb_shipped=(1e-7, 1e-8, 1e-9), cstr_box=(1e-7, 1e-8, 1e-9))[name]
    assert err.max() < lim[0] and np.quantile(err, 0.9) < lim[1] and np.median(err) < lim[2], (err.max(), np.quantile(err, 0.9), np.median(err))
    assert (r["status"][ok] == 0).all()


def test_target_restatement_matches_dense_exact(cstr, wb):
    rng = np.random.default_rng(3)
    for p in (cstr, wb):
        td = rn.target_data(p)
        d = rng.uniform(-0.3, 0.3, size=(12, p.nd)) * (10 if p is cstr else 1)
        ysp = np.array([0.2, 0, 0]) if p is cstr else np.array([1.0, -1.0])
        usp, xsp, usprev = np.zeros(p.nu), np.zeros(p.nx), np.zeros((12, p.nu))
        r = rn.target_solve(p, td, usp, ysp, xsp, d, usprev)
        for i in range(12):
            q = o.target_solve_exact(p, usp, ysp, xsp, d[i], usprev[i])
            assert q["status"] == r["status"][i]
            if q["status"] == 0 and q["exact"]:
                assert np.abs(q["xs"] - r["xs"][i]).max() < 1e-6 and np.abs(q["us"] - r["us"][i]).max() < 1e-6


def test_kalman_matches_textbook_form(cstr):
    rng = np.random.default_rng(5)
    xi = rng.normal(size=6); Pm = cstr.P0 + 1e-3 * np.eye(6); y = rng.normal(size=3)
    yhat = cstr.C @ xi[:3] + cstr.Cd @ xi[3:]
    a, b = o.kalman(cstr, xi, Pm, y, yhat)
    Aa, Ca = cstr.aug_estimator_matrices()
    K = Pm @ Ca.T @ np.linalg.inv(Ca @ Pm @ Ca.T + cstr.R_kf)
    assert np.allclose(a, xi + K @ (y - yhat)) and np.allclose(b, Aa @ (Pm - K @ Ca @ Pm) @ Aa.T + cstr.Q_kf)
    a2, b2 = rn.kalman_batch(cstr, xi[None], Pm[None], y[None], yhat[None])
    assert np.allclose(a2[0], a) and np.allclose(b2[0], b)


def test_general_output_rows_against_the_dense_statement(dint_yrow, xp_nlplant):
    """Output rows that touch several states (Control_Calc.py:130,150-151,229-230) are carried by the Riccati statement as
    extra stage states; the dense statement has them as plain inequality rows.  Same optimum, with the row active."""
    p = dint_yrow
    sd = rn.stage_data(p)
    assert sd["n"] == 3 and list(sd["yg"]) == [0]
    rng = np.random.default_rng(3)
    B = 24
    xh = np.column_stack([rng.uniform(-0.3, 0.3, B), rng.uniform(-0.5, 0.5, B)]); xs = np.tile([1.0, 0.0], (B, 1)); us = np.zeros((B, 1))
    d = rng.uniform(-0.1, 0.1, (B, 1)); up = np.zeros((B, 1))
    r = rn.rpdip_solve(sd, rn.instance_data(p, sd, xh, xs, us, d, up))
    n_active = 0
    for b in range(B):
        e = o.ocp_solve_exact(p, xh[b], xs[b], us[b], d[b], up[b])
        assert (e["status"] == 2) == (r["status"][b] == 2)
        if e["status"] != 0:
            continue
        assert np.abs(e["u0"] - r["u0"][b]).max() < 1e-6
        y = r["z"][b, 1:p.N, :2] @ p.C[0] + p.fy_const[0] + p.Cd[0, 0] * d[b, 0] if "z" in r else None
        if y is not None:
            assert y.max() <= p.ymax[0] + 1e-7 and y.min() >= p.ymin[0] - 1e-7
            assert np.abs(r["z"][b, :, 2] - r["z"][b, :, :2] @ p.C[0]).max() < 1e-9       # the extra state is C_i x along the whole horizon
            n_active += int(y.max() > p.ymax[0] - 1e-5)
    assert n_active >= 3                                    # the set-point xs = (1, 0) lies beyond y <= 0.7: the row binds
    # a stage-0 violation of the row is an infeasible problem (the row constrains a given quantity there)
    bad = rn.instance_data(p, sd, np.array([[0.5, 0.4]]), xs[:1], us[:1], np.zeros((1, 1)), up[:1])
    assert not bad["ok0"][0]
    # the reference example with such a row
    q = xp_nlplant
    sq = rn.stage_data(q)
    assert sq["n"] == 4 + 2 + 1 and list(sq["yg"]) == [0]
    xh = q.x0_m + rng.normal(size=(6, 4)) * [0.02, 1.0, 0.02, 0.3]; dq = rng.normal(size=(6, 2)) * [0.5, 0.01]
    ysp, usp, xsp = q.defSP(0.0)
    t = rn.target_solve(q, rn.target_data(q), np.tile(usp, (6, 1)), np.tile(ysp, (6, 1)), np.tile(xsp, (6, 1)), dq, np.tile(q.u0, (6, 1)))
    uq = np.tile(q.u0, (6, 1))
    rq = rn.rpdip_solve(sq, rn.instance_data(q, sq, xh, t["xs"], t["us"], dq, uq))
    for b in range(6):
        e = o.ocp_solve_exact(q, xh[b], t["xs"][b], t["us"][b], dq[b], uq[b])
        if e["status"] == 0 and rq["status"][b] == 0:
            assert np.abs(e["u0"] - rq["u0"][b]).max() < 1e-6
    assert (rq["status"] == 0).sum() >= 4


def test_ipopt_vectors_if_present(cstr, wb):
    """True CasADi/IPOPT vectors (tools/make_ipopt_vectors.py, written only where casadi is importable): the oracle's exact
    optimum is within IPOPT's own accuracy (tol 1e-8, bound_relax_factor 1e-8: ~1e-6 on u*) of the reference solver's answer,
    and the feasibility labels agree."""
    found = False
    for name, p in (("cstr", cstr), ("wb", wb)):
        f = os.path.join(GOLD, f"ipopt_{name}.npz")
        if not os.path.exists(f):
            continue
        found = True
        g = np.load(f, allow_pickle=True)
        for b in range(len(g["XHAT"])):
            r = o.ocp_solve_exact(p, g["XHAT"][b], g["XS"][b], g["US"][b], g["DHAT"][b], g["U_PREV"][b])
            infeasible = str(g["STATUS"][b]) == "Infeasible_Problem_Detected"
            assert (r["status"] == 2) == infeasible, (name, b)
            if not infeasible:
                assert np.abs(r["u0"] - g["U"][b]).max() < 1e-6 and np.abs(r["x1"] - g["XNEXT"][b]).max() < 1e-6, (name, b)
    if not found:
        pytest.skip("no IPOPT vectors in tests/golden (casadi was not importable where the fixtures were made): parity is pinned by "
                    "the KKT certificates of make_golden.py instead")


def test_du_bounds_rows_of_the_dense_statement(cstr):
    """g2 rows (Control_Calc.py:163-169,241-243): with bounds on u_k - u_{k-1} the exact optimum respects them (also against
    u_prev at k = 0), they bind, and the optimum without them violates them - i.e. the rows are there and they matter."""
    import copy
    p = copy.copy(cstr); p.Dumin = np.array([-0.5, -1.0]); p.Dumax = np.array([0.5, 1.0])
    xh = np.array([0.2, -1.5, 3.0]); xs = np.zeros(3); us = np.zeros(2); d = np.zeros(3); up = np.array([0.3, -0.2])
    r = o.ocp_solve_exact(p, xh, xs, us, d, up)
    assert r["status"] == 0 and r["exact"] and o.kkt_max(r["res"]) < 1e-9
    nxu = p.nx + p.nu
    U = np.array([r["w"][nxu * k + p.nx:nxu * (k + 1)] for k in range(p.N)])
    dU = np.diff(np.vstack([up, U]), axis=0)
    assert (dU >= p.Dumin - 1e-9).all() and (dU <= p.Dumax + 1e-9).all()
    assert (np.abs(dU - p.Dumax) < 1e-8).sum() + (np.abs(dU - p.Dumin) < 1e-8).sum() >= 3
    r0 = o.ocp_solve_exact(cstr, xh, xs, us, d, up)
    U0 = np.array([r0["w"][nxu * k + p.nx:nxu * (k + 1)] for k in range(p.N)])
    assert np.abs(np.diff(np.vstack([up, U0]), axis=0)).max() > 1.5


def test_terminal_equality_rows(pkg):
    """TermCons: g.append(X[N] - xs), Control_Calc.py:193-198 - the dense statement carries nx more equality rows, its optimum ends on
    xs, and an unreachable xs is reported infeasible."""
    import mpc_oracle as o
    p = pkg.load_problem(pkg.example_path("cstr_lmpc.py"), overrides={"N": 6, "TermCons": True})
    q = pkg.load_problem(pkg.example_path("cstr_lmpc.py"), overrides={"N": 6})
    xs, us = np.array([0.02, -0.3, 0.2]), np.zeros(2)
    t = o.target_solve(p, np.zeros(2), np.array([0.1, 0, 0.2]), np.zeros(3), np.zeros(3), np.zeros(2)); xs, us = t["xs"], t["us"]
    Ep = o.ocp_qp(p, np.zeros(3), xs, us, np.zeros(3), np.zeros(2))[2]; Eq = o.ocp_qp(q, np.zeros(3), xs, us, np.zeros(3), np.zeros(2))[2]
    assert Ep.shape[0] == Eq.shape[0] + 3
    r = o.ocp_solve(p, np.array([0.1, 1.0, 1.0]), xs, us, np.zeros(3), np.zeros(2))
    assert r["status"] == 0 and np.abs(r["w"][-3:] - xs).max() < 1e-10
    far = o.ocp_solve(p, np.array([0.5, 8.0, -5.0]), xs, us, np.zeros(3), np.zeros(2))
    assert far["status"] == 2 and o.ocp_solve(q, np.array([0.5, 8.0, -5.0]), xs, us, np.zeros(3), np.zeros(2))["status"] != 2


def test_model_parameters_enter_the_dense_statements(pkg):
    """par_xmk[:, k] in the dynamics rows and par_ymk[:, k] in the output rows (Control_Calc.py:43-57,130,161); p_x_k / p_y_k in the
    target equalities (Target_Calc.py:75-81)."""
    import mpc_oracle as o
    p = pkg.load_problem(pkg.example_path("cstr_lmpc.py"), overrides={"N": 5})
    rng = np.random.default_rng(0)
    px, py = rng.normal(size=(5, 3)) * 0.1, rng.normal(size=(5, 3)) * 0.1
    a = o.ocp_qp(p, np.zeros(3), np.zeros(3), np.zeros(2), np.zeros(3), np.zeros(2))
    b = o.ocp_qp(p, np.zeros(3), np.zeros(3), np.zeros(2), np.zeros(3), np.zeros(2), px=px, py=py)
    n = p.nx
    for k in range(5):
        assert np.allclose(a[3][n * (k + 1):n * (k + 2)] - b[3][n * (k + 1):n * (k + 2)], px[k])
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4]) and not np.allclose(a[5], b[5])
    r = o.ocp_solve(p, np.array([0.1, 1, 1.0]), np.zeros(3), np.zeros(2), np.zeros(3), np.zeros(2), px=px, py=py)
    w = r["w"]
    for k in range(5):      # the returned trajectory obeys the parameterised model
        xk, uk, xn = w[5 * k:5 * k + 3], w[5 * k + 3:5 * k + 5], w[5 * (k + 1):5 * (k + 1) + 3]
        assert np.allclose(xn, o.model_fx(p, xk, uk, np.zeros(3), px[k]), atol=1e-10)
    t = o.target_solve(p, np.zeros(2), np.array([0.1, 0, 0.2]), np.zeros(3), np.zeros(3), np.zeros(2), px0=px[0], py0=py[0])
    assert np.allclose(t["xs"], o.model_fx(p, t["xs"], t["us"], np.zeros(3), px[0]), atol=1e-10)
    assert np.allclose(t["ys"], o.model_fy(p, t["xs"], np.zeros(3), py[0]), atol=1e-10)
