"""The C restatement (oracle/mpc_oracle.c) against the NumPy statements and the golden vectors."""
import os

import numpy as np
import pytest

import riccati_np as rn
from conftest import bench_x0

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("which", ["cstr", "wb"])
def test_ocp_matches_numpy_statement(which, cstr, wb, oracle_c):
    p = cstr if which == "cstr" else wb
    rng = np.random.default_rng(11)
    B = 96
    if which == "cstr":
        xh = bench_x0(B); xs = rng.uniform(-0.2, 0.2, (B, 3)); us = rng.uniform(-1, 1, (B, 2)); d = rng.uniform(-0.1, 0.1, (B, 3))
    else:
        xh = rng.normal(0, 0.3, (B, 4)); xs = rng.normal(0, 0.1, (B, 4)); us = rng.uniform(-0.3, 0.3, (B, 2)); d = rng.normal(0, 0.1, (B, 2))
    up = rng.uniform(-0.4, 0.4, (B, p.nu))
    c = oracle_c.OracleC(p).ocp_solve(xh, xs, us, d, up, want_w=True)
    sd = rn.stage_data(p)
    n = rn.rpdip_solve(sd, rn.instance_data(p, sd, xh, xs, us, d, up))
    assert np.array_equal(c["status"], n["status"])
    ok = c["status"] != 2
    assert (c["iters"] == n["iters"])[ok].mean() > 0.9       # the count flips by one when a test value sits on a threshold
    assert np.abs(c["u0"] - n["u0"])[ok].max() < 1e-8          # identical but for summation order: a rare +-1 iteration
    assert np.abs(c["x1"] - n["z1"][:, :p.nx])[ok].max() < 1e-8
    # w is in opt_dyn's order [x0,u0,...,xN] (Control_Calc.py:31-37) and obeys the model
    w = c["w"][ok]; nxu = p.nx + p.nu
    assert np.allclose(w[:, :p.nx], xh[ok]) and np.allclose(w[:, p.nx:nxu], c["u0"][ok])
    cx = p.fx_const + d[ok] @ p.Bd.T
    for k in range(p.N):
        x, u, xn = w[:, k * nxu:k * nxu + p.nx], w[:, k * nxu + p.nx:(k + 1) * nxu], w[:, (k + 1) * nxu:(k + 1) * nxu + p.nx]
        assert np.abs(x @ p.A.T + u @ p.B.T + cx - xn).max() < 1e-9


def test_target_and_kalman_match_numpy(cstr, wb, oracle_c):
    rng = np.random.default_rng(12)
    for p in (cstr, wb):
        oc = oracle_c.OracleC(p)
        B = 64
        d = rng.uniform(-0.3, 0.3, (B, p.nd)) * (10 if p is cstr else 1)
        ysp = np.array([0.2, 0, 0]) if p is cstr else np.array([1.0, -1.0])
        usprev = rng.uniform(-0.2, 0.2, (B, p.nu))
        c = oc.target_solve(np.zeros(p.nu), ysp, np.zeros(p.nx), d, usprev)
        n = rn.target_solve(p, rn.target_data(p), np.zeros(p.nu), ysp, np.zeros(p.nx), d, usprev)
        assert np.array_equal(c["status"], n["status"])
        ok = c["status"] == 0
        assert np.abs(c["xs"] - n["xs"])[ok].max() < 1e-8 and np.abs(c["us"] - n["us"])[ok].max() < 1e-8
    xi = rng.normal(size=(32, 6)); Pm = np.broadcast_to(cstr.P0 + 1e-3 * np.eye(6), (32, 6, 6)).copy(); y = rng.normal(size=(32, 3))
    yhat = xi[:, :3] @ cstr.C.T + xi[:, 3:] @ cstr.Cd.T
    a, b = oracle_c.OracleC(cstr).kf_update(y, yhat, xi, Pm)
    a2, b2 = rn.kalman_batch(cstr, xi, Pm, y, yhat)
    assert np.abs(a - a2).max() < 1e-12 and np.abs(b.reshape(32, 6, 6) - b2).max() < 1e-12


@pytest.mark.parametrize("name", ["cstr_shipped", "wb_shipped", "cstr_box"])
def test_closed_loop_reproduces_golden_trajectories(name, cstr, wb, oracle_c):
    """Whole closed loop (MPC_code.py:485-827 order, hold rules :714-718,804-805) against the exact golden run."""
    p = wb if name.startswith("wb") else cstr
    g = np.load(os.path.join(GOLD, name + ".npz"))
    nst = g["U"].shape[0]
    x0 = g["Xp"][0]
    L = oracle_c.OracleC(p).closed_loop(nst, x0, x0)
    same = (L["STATUS_DYN"] == 2) == (g["STATUS_DYN"] == 2)
    assert same.mean() > 0.97            # borderline feasibility (|violation| ~ 1e-8) may flip: DESIGN.md section 5
    first_flip = np.argmin(same.all(axis=1)) if not same.all() else nst
    upto = max(first_flip, 1)
    for k in ("U", "XS", "US", "X_HAT", "Xp", "D_HAT"):
        assert np.abs(L[k][:upto] - g[k][:upto]).max() < 5e-6, k
    assert upto >= min(nst, 20)


def test_closed_loop_matches_numpy_batch(cstr, oracle_c):
    x0 = bench_x0(48)
    c = oracle_c.OracleC(cstr).closed_loop(20, x0, x0)
    n = rn.closed_loop_batch(cstr, 20, x0, x0)
    assert np.array_equal(c["STATUS_DYN"], n["STATUS_DYN"]) and np.array_equal(c["STATUS_SS"], n["STATUS_SS"])
    for k in ("U", "XS", "US", "X_HAT", "Xp", "D_HAT", "YS"):
        assert np.abs(c[k] - n[k]).max() < 1e-6, k


@pytest.mark.parametrize("which", ["cstr", "wb"])
def test_warm_starts_do_not_move_the_answer(which, cstr, wb, oracle_c):
    """The closed-loop warm starts (OCP and target problem) change iteration counts, not the trajectory: the warm and the
    cold loop both converge to the same optima within the solver tolerances; numpy and C agree on the iteration counts."""
    p = cstr if which == "cstr" else wb
    x0 = bench_x0(24, 5) if p is cstr else 0.05 * np.random.default_rng(5).standard_normal((24, p.nx))
    oc = oracle_c.OracleC(p)
    warm, cold = oc.closed_loop(40, x0, x0, warm_start=True), oc.closed_loop(40, x0, x0, warm_start=False)
    assert np.array_equal(warm["STATUS_SS"], cold["STATUS_SS"]) and (warm["STATUS_DYN"] == cold["STATUS_DYN"]).mean() > 0.995
    good = (warm["STATUS_DYN"] == cold["STATUS_DYN"]).all(axis=0)
    for k in ("U", "XS", "US", "X_HAT"):
        assert np.abs(warm[k] - cold[k])[:, good].max() < 2e-6, k
    assert warm["ITERS_SS"][10:].mean() < 0.7 * cold["ITERS_SS"][10:].mean()          # the target warm start pays
    assert warm["ITERS_DYN"][10:].mean() < 0.7 * cold["ITERS_DYN"][10:].mean()        # so does the OCP's
    n = rn.closed_loop_batch(p, 40, x0, x0)
    assert np.array_equal(n["ITERS_SS"], warm["ITERS_SS"])


def test_nonlinear_plant_example_controller_path(nlplant, oracle_c):
    """Ex_LMPC_nlplant: linear controller (Delta-u cost, state bounds, operating-point offsets) around an open-loop unstable
    CSTR (|eig A| = 1.75 per step): C and NumPy statements agree on OCPs of that problem, and on the first closed-loop steps
    with the non-linear plant integrated on the host (later the loop limit-cycles between input bounds and amplifies
    rounding differences by about 3x per step, so a long comparison is meaningless)."""
    p = nlplant
    rng = np.random.default_rng(5)
    B = 40
    xh = p.x0_m + rng.normal(size=(B, 3)) * [2e-3, 0.3, 2e-3]; xs = p.x0_m + rng.normal(size=(B, 3)) * [1e-3, 0.1, 1e-3]
    us = np.tile(p.u0, (B, 1)); d = rng.normal(size=(B, 2)) * 0.01; up = p.u0 + rng.normal(size=(B, 2)) * [0.5, 0.005]
    c = oracle_c.OracleC(p).ocp_solve(xh, xs, us, d, up)
    sd = rn.stage_data(p); n = rn.rpdip_solve(sd, rn.instance_data(p, sd, xh, xs, us, d, up))
    assert np.array_equal(c["status"], n["status"]) and (c["status"] == 0).all()
    assert np.abs(c["u0"] - n["u0"]).max() < 1e-6                     # |u| = 300: 3e-9 relative
    x0 = p.x0_p + rng.normal(size=(6, 3)) * [2e-4, 0.02, 2e-4]
    a = rn.closed_loop_batch(p, 8, x0, x0)
    assert (a["STATUS_DYN"] == 0).all() and np.isfinite(a["Xp"]).all()
    assert np.abs(a["Xp"][1] - p.plant_step(x0, a["U"][0], 0.0, np.zeros(3))).max() == 0.0      # the loop uses the RK4 plant


def test_general_output_rows_match_numpy(dint_yrow, xp_nlplant, oracle_c):
    """Output rows carried as extra stage states (oracle/riccati_np.py:stage_data): C restatement against NumPy, per solve and
    in the closed loop."""
    p = dint_yrow
    rng = np.random.default_rng(8)
    B = 80
    xh = np.column_stack([rng.uniform(-0.3, 0.3, B), rng.uniform(-0.5, 0.5, B)]); xs = np.tile([1.0, 0.0], (B, 1)); us = np.zeros((B, 1))
    d = rng.uniform(-0.1, 0.1, (B, 1)); up = rng.uniform(-0.5, 0.5, (B, 1))
    sd = rn.stage_data(p)
    n = rn.rpdip_solve(sd, rn.instance_data(p, sd, xh, xs, us, d, up))
    c = oracle_c.OracleC(p).ocp_solve(xh, xs, us, d, up)
    assert np.array_equal(c["status"], n["status"]) and (c["status"] == 2).any() and (c["status"] == 0).sum() > 40
    ok = c["status"] == 0
    assert np.abs(c["u0"] - n["u0"])[ok].max() < 1e-8 and np.abs(c["x1"] - n["z1"][:, :2])[ok].max() < 1e-8
    x0 = rng.uniform(-0.2, 0.2, (20, 2))
    cl = oracle_c.OracleC(p).closed_loop(25, x0, x0)
    nl = rn.closed_loop_batch(p, 25, x0, x0)
    assert np.array_equal(cl["STATUS_DYN"], nl["STATUS_DYN"]) and np.abs(cl["U"] - nl["U"]).max() < 1e-7
    # the reference example with one general row (model state 4, plant state 3): per solve
    q = xp_nlplant
    xh = q.x0_m + rng.normal(size=(B, 4)) * [0.02, 1.0, 0.02, 0.3]; dq = rng.normal(size=(B, 2)) * [0.5, 0.01]
    xs = q.x0_m + rng.normal(size=(B, 4)) * [0.005, 0.3, 0.005, 0.1]; us = q.u0 + rng.normal(size=(B, 2)) * [0.5, 0.005]; uq = np.tile(q.u0, (B, 1))
    sq = rn.stage_data(q)
    n = rn.rpdip_solve(sq, rn.instance_data(q, sq, xh, xs, us, dq, uq))
    c = oracle_c.OracleC(q).ocp_solve(xh, xs, us, dq, uq)
    assert np.array_equal(c["status"], n["status"])
    ok = c["status"] == 0
    assert ok.sum() > B // 2 and np.abs(c["u0"] - n["u0"])[ok].max() < 1e-7 and np.abs(c["x1"] - n["z1"][:, :4])[ok].max() < 1e-7
