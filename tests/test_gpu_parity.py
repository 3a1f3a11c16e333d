"""Parity of the HIP path (through the C-ABI) with the oracle, on a real MI355X.

Tolerances: both sides stop at the same KKT tolerances, so on identical inputs they agree
far tighter than either agrees with the exact optimum; we require 1e-7 on u* against the C restatement per
solve; against the certified exact optimum of the golden vectors the limits are those an interior-point method can
meet (1e-7 where bounds are non-degenerate, looser on the degenerate steady state of the shipped CSTR run).
"""
import os

import numpy as np
import pytest

import riccati_np as rn
from conftest import bench_x0

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_PORT = 1e-7
TOL_EXACT = 1e-6


def _rand_inputs(p, B, seed):
    rng = np.random.default_rng(seed)
    if p.nx == 3:
        xh = bench_x0(B, seed); xs = rng.uniform(-0.2, 0.2, (B, 3)); us = rng.uniform(-1, 1, (B, 2)); d = rng.uniform(-0.1, 0.1, (B, 3))
    else:
        xh = rng.normal(0, 0.3, (B, p.nx)); xs = rng.normal(0, 0.1, (B, p.nx)); us = rng.uniform(-0.3, 0.3, (B, p.nu)); d = rng.normal(0, 0.1, (B, p.nd))
    return xh, xs, us, d, rng.uniform(-0.4, 0.4, (B, p.nu))


@pytest.mark.parametrize("which,B", [("cstr", 1), ("cstr", 65), ("cstr", 1000), ("wb", 130)])
def test_ocp_matches_c_restatement(which, B, cstr, wb, oracle_c, solver_factory):
    p = cstr if which == "cstr" else wb
    xh, xs, us, d, up = _rand_inputs(p, B, 100 + B)
    g = solver_factory(p).ocp_solve(xh, xs, us, d, up, want_w=True)
    c = oracle_c.OracleC(p).ocp_solve(xh, xs, us, d, up, want_w=True)
    assert np.array_equal(g["status"], c["status"])
    ok = c["status"] != 2
    assert np.abs(g["u0"] - c["u0"])[ok].max() < TOL_PORT
    assert np.abs(g["x1"] - c["x1"])[ok].max() < TOL_PORT
    assert np.nanmax(np.abs(g["w"] - c["w"])[ok]) < 1e-6
    assert (g["iters"] == c["iters"])[ok].mean() > 0.9
    assert np.isnan(g["u0"][~ok]).all()                       # outputs of infeasible instances are left untouched
    assert (g["res"][ok][:, 1] < 1e-8).all()                  # bound residual of the returned point


@pytest.mark.parametrize("name", ["cstr_shipped", "wb_shipped", "cstr_box"])
def test_ocp_hits_certified_exact_optimum(name, cstr, wb, solver_factory):
    p = wb if name.startswith("wb") else cstr
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sh = g["U"].shape[:2]
    flat = lambda a: a.reshape((sh[0] * sh[1],) + a.shape[2:])
    r = solver_factory(p).ocp_solve(flat(g["XHAT_C"]), flat(g["XS"]), flat(g["US"]), flat(g["D_HAT"]), flat(g["U_PREV"]))
    st, exact = flat(g["STATUS_DYN"]), flat(g["EXACT_DYN"]).astype(bool)
    assert np.array_equal(r["status"] == 2, st == 2)
    ok = (st == 0) & exact
    err = np.abs(r["u0"] - flat(g["U"]))[ok].max(axis=1)
    # same limits as tests/test_oracle.py::test_riccati_restatement_reproduces_golden_ocps
    lim = dict(cstr_shipped=(1e-6, 5e-7, 1e-8), wb_shipped=(1e-7, 1e-8, 1e-9), cstr_box=(1e-7, 1e-8, 1e-9))[name]
    assert err.max() < lim[0] and np.quantile(err, 0.9) < lim[1] and np.median(err) < lim[2], (err.max(), np.quantile(err, 0.9), np.median(err))


def test_lqr_known_answer_on_gpu(cstr, solver_factory):
    """No active bound => u0* = us + K (xhat - xs), exactly computable."""
    import mpc_oracle as o
    K = o.lqr_gain(cstr)
    rng = np.random.default_rng(5)
    B = 200
    xs = rng.uniform(-0.01, 0.01, (B, 3)); us = rng.uniform(-0.1, 0.1, (B, 2))
    xh = xs + rng.uniform(-0.03, 0.03, (B, 3)) * [1, 10, 1]
    # (xs, us) must be a model steady state for d: choose d accordingly (Target_Calc.py:75-77)
    d = xs - xs @ cstr.A.T - us @ cstr.B.T
    r = solver_factory(cstr).ocp_solve(xh, xs, us, d, us)
    assert (r["status"] == 0).all()
    assert np.abs(r["u0"] - (us + (xh - xs) @ K.T)).max() < 1e-8


@pytest.mark.parametrize("which", ["cstr", "wb"])
def test_target_and_estimator_match_c_restatement(which, cstr, wb, oracle_c, solver_factory):
    p = cstr if which == "cstr" else wb
    s, oc = solver_factory(p), oracle_c.OracleC(p)
    rng = np.random.default_rng(21)
    B = 333
    d = rng.uniform(-0.3, 0.3, (B, p.nd)) * (10 if p is cstr else 1)
    ysp = np.array([0.2, 0, 0]) if p is cstr else np.array([1.0, -1.0])
    usprev = rng.uniform(-0.2, 0.2, (B, p.nu))
    g = s.target_solve(np.zeros(p.nu), ysp, np.zeros(p.nx), d, usprev)
    c = oc.target_solve(np.zeros(p.nu), ysp, np.zeros(p.nx), d, usprev)
    assert np.array_equal(g["status"], c["status"])
    ok = c["status"] == 0
    for k in ("xs", "us", "ys"):
        assert np.abs(g[k] - c[k])[ok].max() < TOL_PORT, k
    ne = p.nx + p.nd
    xi = rng.normal(size=(B, ne)); y = rng.normal(size=(B, p.ny))
    Pm = np.broadcast_to(1e-3 * np.eye(ne), (B, ne, ne)).copy() if p.estimator == "kal" else None
    yhat = xi[:, :p.nx] @ p.C.T + xi[:, p.nx:] @ p.Cd.T + p.fy_const
    a, b = s.kf_update(y, xi, Pm)
    a2, b2 = oc.kf_update(y, yhat, xi, Pm)
    assert np.abs(a - a2).max() < 1e-12
    if Pm is not None:
        assert np.abs(b - b2.reshape(B, ne, ne)).max() < 1e-12


def assert_same_closed_loop(g, c, p, tol, keys=("U", "XS", "US", "X_HAT", "Xp", "D_HAT", "YS"), max_flipped=0.005):
    """Two closed loops of the same instances: equal status words and values within tol.  An instance whose feasibility label
    differs at some step is compared up to that step only, and the difference has to be *explained*: one of the two labels is
    'infeasible' and the stage-0 feasibility test is borderline (an output row of the given state next to its bound,
    Control_Calc.py:128-151 / SURVEY.md App. C; judged on the logged prior estimate, hence the loose 1e-3) - anything else fails.  At most max_flipped of the instances may be borderline."""
    K, B = g["STATUS_DYN"].shape
    same = g["STATUS_DYN"] == c["STATUS_DYN"]
    first = np.where(same.all(axis=0), K, np.argmin(same, axis=0))          # first step with a different label, per instance
    flipped = np.where(first < K)[0]
    assert len(flipped) <= max_flipped * B, f"{len(flipped)} of {B} instances change their feasibility label"
    for b in flipped:
        k = first[b]
        assert {int(g["STATUS_DYN"][k, b]), int(c["STATUS_DYN"][k, b])} & {2}, (b, k, "labels differ but neither is 'infeasible'")
        dh = g["D_HAT"][k, b] if "D_HAT" in g and p.nd else np.zeros(0)
        y0 = p.C @ g["X_HAT"][k, b] + p.fy_const + (p.Cd @ dh if p.nd else 0.0)      # prior estimate; the test uses the corrected one
        margin = min(np.abs(y0 - p.ymin).min(), np.abs(y0 - p.ymax).min())
        assert margin < 1e-3, (b, k, "label flip away from every output bound", margin)
    step = np.arange(K)[:, None]
    mask = step < first[None, :]                                                  # compare everything before the first flip
    for key in keys:
        if key in g and key in c and g[key].size:
            d = np.abs(g[key] - c[key]).max(axis=2)
            assert d[mask].max() < tol, (key, d[mask].max())
    assert np.array_equal(g["STATUS_SS"][mask], c["STATUS_SS"][mask])
    return len(flipped)


LOOP_KERNELS = [pytest.param(1, id="lane"), pytest.param(2, id="horizon"), pytest.param(3, id="wave")]      # the closed-loop kernels (mpc_set_option "loop_kernel")


@pytest.mark.parametrize("lk", LOOP_KERNELS)
@pytest.mark.parametrize("which,B,nst", [("cstr", 257, 30), ("wb", 64, 40)])
def test_fused_closed_loop_matches_c_restatement(which, B, nst, lk, cstr, wb, oracle_c, solver_factory):
    from mpc_code_amd.driver import run_closed_loop
    p = cstr if which == "cstr" else wb
    x0 = bench_x0(B, 9) if p is cstr else 0.05 * np.random.default_rng(9).standard_normal((B, p.nx))
    g = run_closed_loop(p, x0, x0, nst, solver=solver_factory(p, lk), fused=True)
    c = oracle_c.OracleC(p).closed_loop(nst, x0, x0)
    assert_same_closed_loop(g, c, p, 1e-6)


@pytest.mark.parametrize("which", ["cstr", "wb"])
@pytest.mark.parametrize("N", [2, 3, 5, 64])
def test_edge_horizons_match_c_restatement(N, which, pkg, oracle_c, solver_factory):
    """The shortest horizons (N = 2: the transposing buffer of the wave kernel is smaller than the estimator's exchange area, which
    once overwrote the instance data behind it) and the longest the on-chip kernels take; 65 goes to the lane kernel."""
    from mpc_code_amd.driver import run_closed_loop
    from mpc_code_amd.capi import MpcAmdError
    p = pkg.load_problem(pkg.example_path("cstr_lmpc.py" if which == "cstr" else "wood_berry_lmpc.py"), overrides={"N": N})
    rng = np.random.default_rng(N)
    B = 70
    x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3)) if which == "cstr" else 0.05 * rng.standard_normal((B, p.nx))
    c = oracle_c.OracleC(p).closed_loop(8, x0, x0)
    for lk in (3, 2, 1, 0):
        g = run_closed_loop(p, x0, x0, 8, solver=solver_factory(p, lk), fused=True)
        assert np.array_equal(g["STATUS_DYN"], c["STATUS_DYN"]) and np.array_equal(g["STATUS_SS"], c["STATUS_SS"]), lk
        for k in ("U", "XS", "US", "X_HAT", "Xp", "D_HAT"):
            assert np.max(np.abs(g[k] - c[k])) < 1e-7, (lk, k)
    if N == 64:
        q = pkg.load_problem(pkg.example_path("cstr_lmpc.py" if which == "cstr" else "wood_berry_lmpc.py"), overrides={"N": 65})
        from mpc_code_amd import capi
        s = capi.Solver(q, device=0)
        try:
            for lk in (3, 2):
                with pytest.raises(MpcAmdError, match="N <= 64|horizon"):
                    s.set_option("loop_kernel", lk)
        finally:
            s.close()
        g = run_closed_loop(q, x0, x0, 4, solver=solver_factory(q, 0), fused=True)
        c = oracle_c.OracleC(q).closed_loop(4, x0, x0)
        assert np.array_equal(g["STATUS_DYN"], c["STATUS_DYN"]) and np.max(np.abs(g["U"] - c["U"])) < 1e-7


@pytest.mark.parametrize("which", ["cstr", "wb"])
@pytest.mark.parametrize("lk", LOOP_KERNELS)
def test_plant_and_model_starting_apart(lk, which, cstr, wb, oracle_c, solver_factory):
    """x0_p != x0_m (MPC_code.py:442-476 take them from different Ex-file entries): the estimator has an innovation from the first
    step on.  Every loop kernel against the C restatement; the restatement against the dense oracle on two instances."""
    from mpc_code_amd.driver import run_closed_loop
    import mpc_oracle as mo
    p = cstr if which == "cstr" else wb
    rng = np.random.default_rng(3)
    B = 96
    x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3)) if which == "cstr" else 0.05 * rng.standard_normal((B, p.nx))
    xm = x0 + rng.uniform(-1, 1, size=x0.shape) * ([0.05, 1.0, 0.5] if which == "cstr" else 0.02)
    g = run_closed_loop(p, x0, xm, 12, solver=solver_factory(p, lk), fused=True)
    c = oracle_c.OracleC(p).closed_loop(12, x0, xm)
    assert np.array_equal(g["STATUS_DYN"], c["STATUS_DYN"]) and np.array_equal(g["STATUS_SS"], c["STATUS_SS"])
    for k in ("U", "XS", "US", "X_HAT", "Xp", "D_HAT"):
        assert np.max(np.abs(g[k] - c[k])) < 1e-7, k
    if lk == 3:
        for b in (0, 1):
            o = mo.closed_loop(p, 6, x0_p=x0[b], x0_m=xm[b])
            assert np.array_equal(np.asarray(o["STATUS_DYN"]).ravel(), c["STATUS_DYN"][:6, b])
            for k in ("U", "XS", "X_HAT", "D_HAT"):
                assert np.max(np.abs(np.asarray(o[k]) - c[k][:6, b])) < 2e-6, (b, k)


@pytest.mark.parametrize("lk", LOOP_KERNELS)
def test_stepwise_calls_equal_fused_kernel(lk, cstr, solver_factory):
    """Calling the three solvers per step through the C-ABI (the literal drop-in for MPC_code.py:704,776 and
    Estimator.py) gives the same closed loop as the fused kernel."""
    from mpc_code_amd.driver import run_closed_loop
    s = solver_factory(cstr, lk)
    x0 = bench_x0(96, 4)
    a = run_closed_loop(cstr, x0, x0, 12, solver=s, fused=True)
    b = run_closed_loop(cstr, x0, x0, 12, solver=s, fused=False)
    assert np.array_equal(a["STATUS_DYN"], b["STATUS_DYN"])
    for k in ("U", "XS", "US", "X_HAT", "Y_HAT", "Xp", "Yp", "D_HAT"):      # the reference's result arrays, MPC_code.py:877-895
        assert np.abs(a[k] - b[k]).max() < 5e-6, k
    assert a["TIME_DYN"].shape == (12,) and b["TIME_SS"].shape == (12,) and (b["TIME_DYN"] > 0).all()
    sched = cstr.schedules(12)
    assert np.abs(a["Yp"] - (a["Xp"] @ cstr.Cp.T + sched["pyp"][:, None, :])).max() == 0.0
    assert b["ITERS_DYN"][6:].mean() > 1.5 * a["ITERS_DYN"][6:].mean()      # the warm start is doing its job
    # ... and the per-call entry point honours one too (option "ocp_warm_start": previous optimum and multipliers stay in the handle,
    # MPC_code.py:740-764): same closed loop, the fused loop's iteration counts
    c = run_closed_loop(cstr, x0, x0, 12, solver=s, fused=False, warm_start=True)
    assert np.array_equal(a["STATUS_DYN"], c["STATUS_DYN"])
    for k in ("U", "XS", "US", "X_HAT", "Xp", "D_HAT"):
        assert np.abs(a[k] - c[k]).max() < 5e-6, k
    assert abs(c["ITERS_DYN"].mean() - a["ITERS_DYN"].mean()) < 0.05 * a["ITERS_DYN"].mean()
    s.set_option("ocp_warm_start", 0)


def test_per_call_solver_kernels_and_explicit_guess(cstr, oracle_c, solver_factory):
    """mpc_ocp_solve on the wave-autonomous solver (default where it exists) and on the lane kernel give the C restatement's
    answers; with "ocp_warm_start" a caller's guess x0= (the reference hands IPOPT the shifted previous optimum) is read: same
    optimum, fewer iterations; a guess that is not finite means none."""
    p = cstr
    s, oc = solver_factory(p), oracle_c.OracleC(p)
    B = 300
    xh, xs, us, d, up = _rand_inputs(p, B, 21)
    c = oc.ocp_solve(xh, xs, us, d, up, want_w=True)
    ok = c["status"] == 0
    assert s.get_option("ocp_kernel") == 3
    res = {}
    for kern in (3, 1):
        s.set_option("ocp_kernel", kern)
        g = s.ocp_solve(xh, xs, us, d, up, want_w=True)
        assert np.array_equal(g["status"], c["status"]) and np.abs(g["u0"] - c["u0"])[ok].max() < TOL_PORT and np.abs(g["w"] - c["w"])[ok].max() < 1e-6, kern
        assert (np.abs(g["res"][ok, 0]) < 1e-5).all() and (g["res"][ok, 1] < 1e-8).all()
        res[kern] = g
    s.set_option("ocp_kernel", 0); s.set_option("ocp_warm_start", 1)
    cold = s.ocp_solve(xh, xs, us, d, up, want_w=True)                       # first call of the batch: nothing to start from
    assert np.array_equal(cold["iters"], res[3]["iters"])
    # the "next step" of every instance: state moved along the optimum, guess = optimum shifted by one stage (MPC_code.py:764)
    nxu = p.nx + p.nu
    xh2 = np.where(ok[:, None], cold["x1"], xh); up2 = np.where(ok[:, None], cold["u0"], up)
    guess = np.hstack([np.nan_to_num(cold["w"][:, nxu:]), us, xs])
    warm = s.ocp_solve(xh2, xs, us, d, up2, w_guess=guess)
    ref = oc.ocp_solve(xh2, xs, us, d, up2)
    ok2 = ok & (ref["status"] == 0)
    assert np.array_equal(warm["status"][ok], ref["status"][ok]) and np.abs(warm["u0"] - ref["u0"])[ok2].max() < TOL_PORT
    assert warm["iters"][ok2].mean() < 0.5 * ref["iters"][ok2].mean()
    nog = s.ocp_solve(xh2, xs, us, d, up2, w_guess=np.full((B, p.nw), np.nan))    # no guess: the handle's own previous optimum, shifted
    assert np.array_equal(nog["status"][ok], ref["status"][ok]) and np.abs(nog["u0"] - ref["u0"])[ok2].max() < TOL_PORT
    s.set_option("ocp_warm_start", 0)


@pytest.mark.parametrize("lk", LOOP_KERNELS)
def test_shipped_scenarios_follow_the_golden_closed_loop(lk, cstr, wb, solver_factory):
    from mpc_code_amd.driver import run_closed_loop
    for p, name in ((cstr, "cstr_shipped"), (wb, "wb_shipped")):
        g = np.load(os.path.join(GOLD, name + ".npz"))
        r = run_closed_loop(p, nsteps=100, solver=solver_factory(p, lk))
        # all 100 steps: the same feasibility labels as the exact golden loop, and its values (measured: 5.7e-8 / 3e-9)
        assert np.array_equal(r["STATUS_DYN"] == 2, g["STATUS_DYN"] == 2)
        assert np.abs(r["U"] - g["U"]).max() < 1e-6
        assert np.abs(r["X_HAT"] - g["X_HAT"]).max() < 1e-6
    # the shipped CSTR run starts infeasible: u is held at u0 = 0 for steps 0-2
    r = run_closed_loop(cstr, nsteps=4, solver=solver_factory(cstr, lk))
    assert (r["STATUS_DYN"][:3, 0] == 2).all() and r["STATUS_DYN"][3, 0] == 0 and np.all(r["U"][:3] == 0.0)


def test_full_size_properties(cstr, solver_factory):
    """BASELINE.json configs[1] size (4096): size-independent properties of every returned trajectory."""
    p = cstr
    s = solver_factory(p)
    B = 4096
    xh, xs, us, d, up = _rand_inputs(p, B, 77)
    r = s.ocp_solve(xh, xs, us, d, up, want_w=True)
    ok = r["status"] == 0
    assert ok.mean() > 0.9 and (r["status"] != 1).all()
    w = r["w"][ok]; nxu = p.nx + p.nu
    X = np.stack([w[:, k * nxu:k * nxu + p.nx] for k in range(p.N + 1)], 1); U = np.stack([w[:, k * nxu + p.nx:(k + 1) * nxu] for k in range(p.N)], 1)
    cx = d[ok] @ p.Bd.T
    assert np.abs(X[:, :-1] @ p.A.T + U @ p.B.T + cx[:, None] - X[:, 1:]).max() < 1e-10        # the model holds
    assert (U >= p.umin - 1e-9).all() and (U <= p.umax + 1e-9).all()                             # bounds hold
    assert (X[:, 1:] >= p.xmin - 1e-9).all() and (X[:, 1:] <= p.xmax + 1e-9).all()
    assert np.array_equal(X[:, 0], xh[ok]) and np.array_equal(U[:, 0], r["u0"][ok]) and np.array_equal(X[:, 1], r["x1"][ok])
    r2 = s.ocp_solve(xh, xs, us, d, up)
    assert np.array_equal(r2["u0"][ok], r["u0"][ok])                                             # deterministic
    # optimality: no feasible perturbation of the input sequence lowers the cost (first-order, sampled)
    sd = rn.stage_data(p)
    def cost(Xt, Ut):
        dx = Xt - xs[ok][:, None]; du = Ut - us[ok][:, None]
        return 0.5 * (np.einsum("bki,ij,bkj->b", dx[:, :-1], p.Q, dx[:, :-1]) + np.einsum("bki,ij,bkj->b", du, p.R, du)
                      + np.einsum("bi,ij,bj->b", dx[:, -1], p.P, dx[:, -1]))
    base = cost(X, U)
    rng = np.random.default_rng(0)
    Up = np.clip(U + 1e-3 * rng.standard_normal(U.shape), p.umin, p.umax)
    Xp = np.empty_like(X); Xp[:, 0] = X[:, 0]
    for k in range(p.N):
        Xp[:, k + 1] = Xp[:, k] @ p.A.T + Up[:, k] @ p.B.T + cx
    feas = ((Xp[:, 1:] >= p.xmin) & (Xp[:, 1:] <= p.xmax)).all(axis=(1, 2))
    assert feas.sum() > 100 and (cost(Xp, Up)[feas] >= base[feas] - 1e-9).all()


def test_error_paths_are_loud(cstr, solver_factory, pkg):
    import ctypes as ct
    from mpc_code_amd import capi
    s = solver_factory(cstr)
    one = np.zeros((1, 3)); two = np.zeros((1, 2)); st = np.zeros(1, np.int32)
    dp = lambda a: a.ctypes.data_as(ct.POINTER(ct.c_double))
    rc = s.lib.mpc_ocp_solve(s.h, 1, dp(one), dp(one), dp(two), None, dp(two), None, None, None, dp(two), dp(one),
                             st.ctypes.data_as(ct.POINTER(ct.c_int32)), None, None)
    assert rc != 0 and b"dhat is required" in s.lib.mpc_last_error()
    s.set_model_offsets(4, np.zeros(3), np.zeros(3))             # this step's p_x_k, p_y_k for a batch of 4 ...
    with pytest.raises(capi.MpcAmdError, match="batch of 4"):
        s.target_solve(np.zeros(2), np.zeros(3), np.zeros(3), np.zeros((1, 3)), np.zeros(2))      # ... do not fit a batch of 1
    s.set_model_offsets(1, None, None)
    import copy
    q = copy.copy(cstr); q.nd = 2; q.Bd = cstr.Bd[:, :2]; q.Cd = cstr.Cd[:, :2]; q.dhat0 = np.zeros(2); q.estimator = "none"
    with pytest.raises(capi.MpcAmdError) as e:
        capi.Solver(q, jit=False)                               # (with jit=True the library of this dimension set would be built)
    assert "no kernel compiled" in str(e.value)
    with pytest.raises(capi.MpcAmdError):
        s.loop_run(0, 1)                                       # before mpc_loop_alloc
    with pytest.raises(capi.MpcAmdError):
        s.set_option("loop_kernel", 4)
    with pytest.raises(capi.MpcAmdError):
        s.set_option("no_such_option", 1)
    # the resident loop refuses to run on a state nobody supplied, and a fresh Kalman filter needs its covariance
    s.loop_alloc(4, 2, capi.LOG_NONE); s.loop_set_schedule(cstr.schedules(2))
    with pytest.raises(capi.MpcAmdError):
        s.loop_run(0, 1)
    x4, u4 = np.zeros((4, 3)), np.zeros((4, 2))
    rc = s.lib.mpc_loop_set_state(s.h, dp(x4), dp(x4), dp(x4), None, dp(u4), dp(x4), dp(u4))
    assert rc != 0 and b"covariance" in s.lib.mpc_last_error()
    rc = s.lib.mpc_loop_set_state(s.h, dp(x4), dp(x4), dp(x4), dp(np.zeros((4, 6, 6))), None, dp(x4), dp(u4))
    assert rc != 0 and b"needs x_p, xhat, u" in s.lib.mpc_last_error()


def test_loop_kernel_choice(cstr, wb, solver_factory):
    """mpc_loop_run picks the wave-autonomous kernel whenever it exists for the problem (stage state <= 8 as 2 x 2 tiles, nu <= 2,
    N <= 64: CSTR and Wood-Berry alike), the lane kernel for long horizons; steps_per_launch defaults to 50."""
    import copy
    from mpc_code_amd import capi
    s = solver_factory(cstr)
    assert s.get_option("steps_per_launch") == 50
    s.loop_alloc(100, 2, capi.LOG_NONE); assert s.get_option("loop_kernel") == 3
    s.loop_alloc(20000, 2, capi.LOG_NONE); assert s.get_option("loop_kernel") == 3
    sw = solver_factory(wb)
    sw.loop_alloc(100, 2, capi.LOG_NONE); assert sw.get_option("loop_kernel") == 3      # stage state 6: 2 x 2 tiles
    sw.loop_alloc(20000, 2, capi.LOG_NONE); assert sw.get_option("loop_kernel") == 3
    s.set_option("loop_kernel", 2); assert s.get_option("loop_kernel") == 2
    s.set_option("loop_kernel", 0)
    q = copy.copy(cstr); q.N = 70                               # N > 64 does not fit a wave: lane kernel, and 2 is refused
    t = capi.Solver(q)
    try:
        t.loop_alloc(100, 2, capi.LOG_NONE); assert t.get_option("loop_kernel") == 1
        with pytest.raises(capi.MpcAmdError):
            t.set_option("loop_kernel", 2)
        with pytest.raises(capi.MpcAmdError):
            t.set_option("loop_kernel", 3)
        x0 = bench_x0(100, 3)
        t.loop_set_schedule(q.schedules(2)); t.loop_set_state(x0, x0); t.loop_run(0, 2); t.loop_sync()
    finally:
        t.close()


@pytest.mark.parametrize("lk", LOOP_KERNELS)
def test_launch_granularity_and_convenience_api_do_not_change_results(lk, cstr, solver_factory):
    """steps_per_launch (one launch per step vs. all steps in one launch) and mpc_closed_loop (host buffers only)
    advance the same closed loop; ragged batch (not a multiple of the wave size)."""
    import ctypes as ct
    from mpc_code_amd import capi
    s = solver_factory(cstr, lk)
    B, K = 131, 9
    x0 = bench_x0(B, 11)
    sched = cstr.schedules(K)
    runs = []
    for spl in (1, 4, 9):
        s.set_option("steps_per_launch", spl)
        s.loop_alloc(B, K, capi.LOG_ALL); s.loop_set_state(x0, x0); s.loop_set_schedule(sched)
        s.loop_run(0, K); s.loop_sync()
        runs.append((s.loop_get_log("U"), s.loop_get_log("STATUS_DYN"), s.loop_get_state()))
    s.set_option("steps_per_launch", 1)
    for U, st, fin in runs[1:]:
        assert np.array_equal(U, runs[0][0]) and np.array_equal(st, runs[0][1])
        for k in ("x_p", "xhat", "dhat", "P", "u", "xs", "us"):
            assert np.array_equal(fin[k], runs[0][2][k]), k
    # two halves of the schedule, resumed from resident state, equal one run
    s.loop_alloc(B, K, capi.LOG_ALL); s.loop_set_state(x0, x0); s.loop_set_schedule(sched)
    s.loop_run(0, 4); s.loop_run(4, K - 4); s.loop_sync()
    assert np.array_equal(s.loop_get_log("U"), runs[0][0])
    # mpc_closed_loop: everything through host buffers
    dp = lambda a: a.ctypes.data_as(ct.POINTER(ct.c_double))
    ne = cstr.nx + cstr.nd
    xp, xh = x0.copy(), x0.copy(); dh = np.zeros((B, cstr.nd)); Pk = np.broadcast_to(cstr.P0, (B, ne, ne)).copy()
    u = np.zeros((B, cstr.nu)); xs = x0.copy(); us = u.copy(); Ulog = np.zeros((K, B, cstr.nu))
    rc = s.lib.mpc_closed_loop(s.h, B, K, dp(xp), dp(xh), dp(dh), dp(Pk), dp(u), dp(xs), dp(us), dp(sched["ysp"]), dp(sched["usp"]),
                               dp(sched["xsp"]), dp(sched["pxp"]), dp(sched["pyp"]), dp(Ulog))
    assert rc == 0, s.lib.mpc_last_error()
    assert np.array_equal(Ulog, runs[0][0]) and np.array_equal(u, runs[0][2]["u"]) and np.array_equal(xp, runs[0][2]["x_p"])


def test_short_horizon_and_double_integrator(pkg, oracle_c, solver_factory):
    """Smallest compiled dimension set (nx=2, nu=1) with N=2..5: edge of the sweeps' first/last-block handling."""
    import copy
    from mpc_code_amd.problem import LinearMPCProblem
    import scipy.linalg as scla
    A = np.array([[1.0, 0.1], [0.0, 1.0]]); Bm = np.array([[0.005], [0.1]]); C = np.array([[1.0, 0.0]])
    Q = np.diag([1.0, 0.1]); R = np.array([[0.01]])
    for N, du in ((2, False), (3, True), (5, False)):
        P = scla.solve_discrete_are(A, Bm, Q, R)
        inf = np.inf
        p = LinearMPCProblem(nx=2, nu=1, ny=1, nd=1, nxp=2, N=N, h=0.1, Nsim=10, A=A, B=Bm, C=C, Bd=np.array([[0.0], [0.1]]), Cd=np.zeros((1, 1)),
                             fx_const=np.zeros(2), fy_const=np.zeros(1), Ap=A, Bp=Bm, Cp=C, Q=Q, R=R, DUForm=du, P=P,
                             Qss=np.eye(1), Rss=np.zeros((1, 1)), DUssForm=False, umin=np.array([-1.0]), umax=np.array([1.0]),
                             xmin=np.array([-inf, -0.5]), xmax=np.array([2.0, inf]), ymin=np.array([-inf]), ymax=np.array([inf]), y_bounded=False,
                             umin_ss=np.array([-1.0]), umax_ss=np.array([1.0]), xmin_ss=np.array([-inf, -0.5]), xmax_ss=np.array([2.0, inf]),
                             ymin_ss=np.array([-inf]), ymax_ss=np.array([inf]), estimator="kalss", K=np.array([[0.5], [0.1], [0.2]]),
                             x0_p=np.zeros(2), x0_m=np.zeros(2), u0=np.zeros(1), dhat0=np.zeros(1))
        rng = np.random.default_rng(N)
        B = 70
        xh = rng.uniform(-1, 1, (B, 2)); xs = np.tile([0.5, 0.0], (B, 1)); us = np.zeros((B, 1)); d = rng.uniform(-0.1, 0.1, (B, 1)); up = rng.uniform(-1, 1, (B, 1))
        g = solver_factory(p).ocp_solve(xh, xs, us, d, up, want_w=True)
        c = oracle_c.OracleC(p).ocp_solve(xh, xs, us, d, up, want_w=True)
        assert np.array_equal(g["status"], c["status"])
        ok = c["status"] != 2
        assert ok.sum() > 10 and np.abs(g["u0"] - c["u0"])[ok].max() < TOL_PORT and np.nanmax(np.abs(g["w"] - c["w"])[ok]) < 1e-6
        # the closed loop on the same tiny horizon, both kernels (masked bounds, fixed-gain estimator, ragged batch)
        from mpc_code_amd.driver import run_closed_loop
        x0 = rng.uniform(-0.4, 0.4, (37, 2))
        cl = oracle_c.OracleC(p).closed_loop(8, x0, x0)
        for lk in (1, 2, 3):
            gl = run_closed_loop(p, x0, x0, 8, solver=solver_factory(p, lk))
            assert np.array_equal(gl["STATUS_DYN"], cl["STATUS_DYN"]), (N, lk)
            assert np.abs(gl["U"] - cl["U"]).max() < TOL_PORT and np.abs(gl["X_HAT"] - cl["X_HAT"]).max() < TOL_PORT, (N, lk)


@pytest.mark.parametrize("lk", LOOP_KERNELS)
def test_full_size_closed_loop(lk, cstr, oracle_c, solver_factory):
    """The benchmark workload itself (BASELINE.json configs[1]: 4096 instances, 100 steps from t = 0): EVERY instance and step
    against the C restatement - all 409 600 status words and all trajectories (measured: no label differs, 5e-9) - and
    invariants on all of them."""
    from mpc_code_amd.driver import run_closed_loop
    p = cstr
    B, K = 4096, 100
    x0 = bench_x0(B)
    g = run_closed_loop(p, x0, x0, K, solver=solver_factory(p, lk))
    c = oracle_c.OracleC(p).closed_loop(K, x0, x0)
    assert assert_same_closed_loop(g, c, p, 1e-7, max_flipped=0.001) <= 4
    # invariants on all 4096: inputs inside their box; the plant log obeys the plant; held steps keep u
    assert (g["U"] >= p.umin - 1e-9).all() and (g["U"] <= p.umax + 1e-9).all()
    sched = p.schedules(K)
    assert np.abs(g["Xp"][1:] - (g["Xp"][:-1] @ p.Ap.T + g["U"][:-1] @ p.Bp.T + sched["pxp"][:-1, None, :])).max() < 1e-12
    held = g["STATUS_DYN"][1:] == 2
    assert np.array_equal(g["U"][1:][held], g["U"][:-1][held])
    assert (g["STATUS_DYN"] != 1).all() and (g["STATUS_SS"] == 0).all()
    # the warm start pays: after the transients the slowest instance needs at most 2 iterations
    assert g["ITERS_DYN"][8:15].max() <= 2


@pytest.mark.parametrize("lk", LOOP_KERNELS)
def test_edge_batches_horizon_64_and_iteration_limit(lk, cstr, oracle_c, solver_factory):
    """Batch sizes around the workgroup granularity (1, 15, 17 instances), the longest horizon one wave holds (N = 64), and a
    run that hits the iteration limit (status 1 is accepted like any other, MPC_code.py:786): same closed loop as the C restatement."""
    import copy
    from mpc_code_amd.driver import run_closed_loop
    for B in (1, 15, 17):
        x0 = bench_x0(B, 40 + B)
        g = run_closed_loop(cstr, x0, x0, 6, solver=solver_factory(cstr, lk))
        c = oracle_c.OracleC(cstr).closed_loop(6, x0, x0)
        assert np.array_equal(g["STATUS_DYN"], c["STATUS_DYN"]) and np.abs(g["U"] - c["U"]).max() < TOL_PORT, B
    q = copy.copy(cstr); q.N = 64
    x0 = bench_x0(40, 77)
    g = run_closed_loop(q, x0, x0, 5, solver=solver_factory(q, lk))
    c = oracle_c.OracleC(q).closed_loop(5, x0, x0)
    assert assert_same_closed_loop(g, c, q, TOL_PORT, max_flipped=0.0) == 0      # no label differs (tools/flip_report.py); a flip would have to be explained
    r = copy.copy(cstr); r.max_iter = 4
    g = run_closed_loop(r, x0, x0, 3, solver=solver_factory(r, lk))
    c = oracle_c.OracleC(r).closed_loop(3, x0, x0)
    assert (c["STATUS_DYN"] == 1).any() and np.array_equal(g["STATUS_DYN"], c["STATUS_DYN"])
    assert np.abs(g["U"] - c["U"]).max() < 1e-6 and g["ITERS_DYN"].max() <= 4


def test_nonlinear_plant_example_through_the_three_calls(nlplant, oracle_c, solver_factory):
    """Ex_LMPC_nlplant (reference example with a non-linear plant): estimator, target and OCP through the C-ABI, plant on the
    host (driver.run_closed_loop(fused=False)); per-call parity with the C restatement and the first closed-loop steps
    against the NumPy loop (the loop amplifies differences about 3x per step: open-loop unstable CSTR)."""
    from mpc_code_amd.driver import run_closed_loop
    p = nlplant
    rng = np.random.default_rng(5)
    B = 64
    xh = p.x0_m + rng.normal(size=(B, 3)) * [2e-3, 0.3, 2e-3]; xs = p.x0_m + rng.normal(size=(B, 3)) * [1e-3, 0.1, 1e-3]
    us = np.tile(p.u0, (B, 1)); d = rng.normal(size=(B, 2)) * 0.01; up = p.u0 + rng.normal(size=(B, 2)) * [0.5, 0.005]
    s, oc = solver_factory(p), oracle_c.OracleC(p)
    g, c = s.ocp_solve(xh, xs, us, d, up), oc.ocp_solve(xh, xs, us, d, up)
    assert np.array_equal(g["status"], c["status"]) and np.abs(g["u0"] - c["u0"]).max() < 1e-6 and np.abs(g["x1"] - c["x1"]).max() < 1e-6
    t, tc = s.target_solve(np.array([299.963, 0.1]), np.array([0.5, 0.659]), np.zeros(3), d, up), oc.target_solve(np.array([299.963, 0.1]), np.array([0.5, 0.659]), np.zeros(3), d, up)
    assert np.array_equal(t["status"], tc["status"]) and np.abs(t["xs"] - tc["xs"]).max() < 1e-8 and np.abs(t["us"] - tc["us"]).max() < 1e-8
    x0 = p.x0_p + rng.normal(size=(12, 3)) * [2e-4, 0.02, 2e-4]
    gl = run_closed_loop(p, x0, x0, 8, solver=s, fused=False)
    nl = rn.closed_loop_batch(p, 8, x0, x0, warm_start=False)
    assert np.array_equal(gl["STATUS_DYN"], nl["STATUS_DYN"]) and (gl["STATUS_DYN"] == 0).all()
    assert np.abs(gl["U"] - nl["U"]).max() < 1e-5 and np.abs(gl["Xp"] - nl["Xp"]).max() < 1e-5


def test_both_kernels_leave_the_same_resident_state(cstr, solver_factory):
    """mpc_loop_get_state means the same after either kernel: in particular P is the prior covariance of the next step,
    although the horizon-parallel kernel computes the filter's covariance half one step ahead."""
    from mpc_code_amd import capi
    B, K = 50, 7
    x0 = bench_x0(B, 123)
    fin = []
    for lk in (1, 2, 3):
        s = solver_factory(cstr, lk)
        s.loop_alloc(B, K, capi.LOG_U); s.loop_set_schedule(cstr.schedules(K)); s.loop_set_state(x0, x0)
        s.loop_run(0, 3); s.loop_run(3, K - 3); s.loop_sync()
        fin.append(s.loop_get_state())
    for other in fin[1:]:
        for k in ("x_p", "xhat", "dhat", "u", "xs", "us"):
            assert np.abs(fin[0][k] - other[k]).max() < 1e-6, k
    assert np.array_equal(fin[0]["P"], fin[1]["P"])                     # same operations on the same numbers
    # the wave-autonomous kernel forms (I - K C) P with P C' transposed and A (P_corr A') instead of (A P_corr) A': rounding only
    assert np.abs(fin[0]["P"] - fin[2]["P"]).max() <= 1e-12 * np.abs(fin[0]["P"]).max()


def test_general_output_rows(dint_yrow, xp_nlplant, oracle_c, solver_factory):
    """Bounded output rows that touch several states (Control_Calc.py:130,150-151,229-230): one extra stage state each
    (mpc_amd.hip:build_problem).  Double integrator with y = x0 + x1 bounded, row active: per solve and in the fused closed
    loop of both kernels; then Ex_LMPCxp_nlplant (model state 4, plant state 3, ylin, one general row) through the three
    calls with the non-linear plant on the host."""
    from mpc_code_amd.driver import run_closed_loop
    p = dint_yrow
    rng = np.random.default_rng(8)
    B = 200
    xh = np.column_stack([rng.uniform(-0.3, 0.3, B), rng.uniform(-0.5, 0.5, B)]); xs = np.tile([1.0, 0.0], (B, 1)); us = np.zeros((B, 1))
    d = rng.uniform(-0.1, 0.1, (B, 1)); up = rng.uniform(-0.5, 0.5, (B, 1))
    s, oc = solver_factory(p), oracle_c.OracleC(p)
    assert "2/1/1/1/2/0/1" in s.build_info()
    g, c = s.ocp_solve(xh, xs, us, d, up, want_w=True), oc.ocp_solve(xh, xs, us, d, up, want_w=True)
    assert np.array_equal(g["status"], c["status"]) and (c["status"] == 2).any()
    ok = c["status"] == 0
    assert ok.sum() > 100 and np.abs(g["u0"] - c["u0"])[ok].max() < TOL_PORT and np.abs(g["x1"] - c["x1"])[ok].max() < TOL_PORT
    w = g["w"][ok]; nxu = 3
    y = np.stack([w[:, k * nxu:k * nxu + 2] @ p.C[0] for k in range(1, p.N)], axis=1) + p.fy_const[0] + p.Cd[0, 0] * d[ok]
    assert y.max() <= p.ymax[0] + 1e-7 and y.min() >= p.ymin[0] - 1e-7 and (y.max(axis=1) > p.ymax[0] - 1e-5).sum() > 20      # the row binds
    x0 = rng.uniform(-0.2, 0.2, (150, 2))
    cl = oc.closed_loop(25, x0, x0)
    for lk in (1, 2, 3):
        gl = run_closed_loop(p, x0, x0, 25, solver=solver_factory(p, lk))
        assert np.array_equal(gl["STATUS_DYN"], cl["STATUS_DYN"]), lk
        assert np.abs(gl["U"] - cl["U"]).max() < TOL_PORT and np.abs(gl["X_HAT"] - cl["X_HAT"]).max() < TOL_PORT, lk
    # the reference example
    q = xp_nlplant
    B = 96
    xh = q.x0_m + rng.normal(size=(B, 4)) * [0.02, 1.0, 0.02, 0.3]; dq = rng.normal(size=(B, 2)) * [0.5, 0.01]
    xs = q.x0_m + rng.normal(size=(B, 4)) * [0.005, 0.3, 0.005, 0.1]; us = q.u0 + rng.normal(size=(B, 2)) * [0.5, 0.005]; uq = np.tile(q.u0, (B, 1))
    s, oc = solver_factory(q), oracle_c.OracleC(q)
    g, c = s.ocp_solve(xh, xs, us, dq, uq), oc.ocp_solve(xh, xs, us, dq, uq)
    assert np.array_equal(g["status"], c["status"])
    ok = c["status"] == 0
    assert ok.sum() > B // 2 and np.abs(g["u0"] - c["u0"])[ok].max() < 1e-6 and np.abs(g["x1"] - c["x1"])[ok].max() < 1e-6
    ysp, usp, xsp = q.defSP(0.0)
    t, tc = s.target_solve(usp, ysp, xsp, dq, uq), oc.target_solve(usp, ysp, xsp, dq, uq)
    assert np.array_equal(t["status"], tc["status"]) and np.abs(t["xs"] - tc["xs"]).max() < 1e-8 and np.abs(t["us"] - tc["us"]).max() < 1e-8
    x0p = q.x0_p + rng.normal(size=(12, 3)) * [2e-4, 0.02, 2e-4]; x0m = np.hstack([x0p, np.zeros((12, 1))])
    gl = run_closed_loop(q, x0p, x0m, 8, solver=s, fused=False)
    nl = rn.closed_loop_batch(q, 8, x0p, x0m, warm_start=False)
    assert np.array_equal(gl["STATUS_DYN"], nl["STATUS_DYN"]) and (gl["STATUS_DYN"] == 0).all()
    assert np.abs(gl["U"] - nl["U"]).max() < 1e-5 and np.abs(gl["Xp"] - nl["Xp"]).max() < 1e-5


def test_kernel_variants_of_the_bound_sets(cstr, oracle_c, solver_factory):
    """The three kernel variants mpc_lin_create chooses from (bounds on everything: no masks; inputs only: no state rows at
    all; anything else: masks), each through the fused loop of both kernels - the matrix-core factorisation reads its barrier
    weights per variant (no sigma_z rows when only the inputs are bounded)."""
    import copy
    from mpc_code_amd.driver import run_closed_loop
    inf = np.inf
    x0 = bench_x0(150, 11) * [1.0, 0.5, 0.5]
    variants = {"all finite": {}, "inputs only": dict(xmin=np.full(3, -inf), xmax=np.full(3, inf), ymin=np.full(3, -inf), ymax=np.full(3, inf), y_bounded=False),
                "one-sided": dict(xmin=np.array([-inf, -8.0, -inf]), xmax=np.array([10.0, inf, 10.0]), ymin=np.full(3, -inf), ymax=np.array([inf, 10.0, inf]))}
    for name, over in variants.items():
        p = copy.copy(cstr)
        for k, v in over.items():
            setattr(p, k, v)
        cl = oracle_c.OracleC(p).closed_loop(25, x0, x0)
        assert (cl["STATUS_DYN"] == 0).mean() > 0.9, name
        for lk in (1, 2, 3):
            gl = run_closed_loop(p, x0, x0, 25, solver=solver_factory(p, lk))
            assert np.array_equal(gl["STATUS_DYN"], cl["STATUS_DYN"]), (name, lk)
            assert np.abs(gl["U"] - cl["U"]).max() < TOL_PORT and np.abs(gl["X_HAT"] - cl["X_HAT"]).max() < TOL_PORT, (name, lk)


def test_other_dimensions_through_the_jit_library(five_state, oracle_c, solver_factory):
    """A dimension set the default library does not carry (nx = 5, nu = 2: stage blocks of 2 x 2 tiles on the matrix cores):
    capi.Solver builds the library of exactly that set (csrc/jit/, prebuilt in the build container so that it ships) and the
    three closed-loop kernels agree with the C restatement."""
    from mpc_code_amd import capi
    from mpc_code_amd.driver import run_closed_loop
    p = five_state
    assert "5/2/2/2/5/0/0" not in capi.load_library().mpc_build_info().decode()
    rng = np.random.default_rng(12)
    x0 = rng.uniform(-1.0, 1.0, (203, 5))
    c = oracle_c.OracleC(p).closed_loop(20, x0, x0)
    assert (c["STATUS_DYN"] == 0).mean() > 0.9 and c["ITERS_DYN"].max() > 3
    for lk in (1, 2, 3):
        s = solver_factory(p, lk)
        assert "5/2/2/2/5/0/0" in s.build_info()
        g = run_closed_loop(p, x0, x0, 20, solver=s)
        assert_same_closed_loop(g, c, p, TOL_PORT)


def test_du_bounds_against_the_dense_statement(cstr, wb, solver_factory):
    """Bounds on u_k - u_{k-1} (g2 rows, Control_Calc.py:163-169,241-243) run in the stage form with input v = u_k - u_{k-1} and
    state [x; u_prev] (mpc_amd.hip:build_problem).  Per call and in the closed loop of each kernel against the dense statement
    of opt_dyn with the rows as plain inequalities (oracle/mpc_oracle.py), cost on u - us (CSTR) and on u_k - u_{k-1} (Wood-Berry)."""
    import copy
    import mpc_oracle as o
    from mpc_code_amd.driver import run_closed_loop
    rng = np.random.default_rng(31)
    pc = copy.copy(cstr); pc.Dumin = np.array([-0.5, -1.0]); pc.Dumax = np.array([0.5, 1.0])
    pw = copy.copy(wb); pw.Dumin = np.array([-0.02, -0.05]); pw.Dumax = np.array([0.03, 0.05])
    for p, B in ((pc, 24), (pw, 12)):
        if p is pc:
            xh = bench_x0(B, 5) * [1.0, 0.3, 0.6]; xs = np.zeros((B, 3)); us = np.zeros((B, 2))
            d = 0.02 * rng.standard_normal((B, 3)); up = rng.uniform(-1.0, 1.0, (B, 2))
        else:
            xh = 0.3 * rng.standard_normal((B, 4)); xs = 0.1 * rng.standard_normal((B, 4)); us = 0.05 * rng.standard_normal((B, 2))
            d = 0.05 * rng.standard_normal((B, 2)); up = rng.uniform(-0.3, 0.3, (B, 2))
        s = solver_factory(p)
        g = s.ocp_solve(xh, xs, us, d, up, want_w=True)
        nxu = p.nx + p.nu
        bound = 0
        for b in range(B):
            r = o.ocp_solve_exact(p, xh[b], xs[b], us[b], d[b], up[b])
            assert g["status"][b] == r["status"], (p.name, b)
            if r["status"] != 0:
                continue
            assert np.abs(g["u0"][b] - r["u0"]).max() < 1e-6 and np.abs(g["x1"][b] - r["x1"]).max() < 1e-6, (p.name, b)
            assert np.abs(g["w"][b] - r["w"]).max() < 1e-5, (p.name, b)                    # the whole trajectory, in opt_dyn's order
            U = np.array([g["w"][b][nxu * k + p.nx:nxu * (k + 1)] for k in range(p.N)])
            dU = np.diff(np.vstack([up[b], U]), axis=0)
            assert (dU >= p.Dumin - 1e-7).all() and (dU <= p.Dumax + 1e-7).all()
            bound += int(((np.abs(dU - p.Dumax) < 1e-6) | (np.abs(dU - p.Dumin) < 1e-6)).any())
        assert bound >= B // 3, (p.name, bound)                                          # the rows bind
    # closed loop, every kernel, against the dense closed loop (exact optimum per step)
    x0 = bench_x0(3, 8) * [1.0, 0.3, 0.6]
    ref = [o.closed_loop(pc, 10, x0_p=x, x0_m=x, ocp=o.ocp_solve_exact, target=o.target_solve_exact) for x in x0]
    for lk in (1, 2, 3):
        gl = run_closed_loop(pc, x0, x0, 10, solver=solver_factory(pc, lk))
        for b, r in enumerate(ref):
            assert np.array_equal(gl["STATUS_DYN"][:, b], r["STATUS_DYN"]), (lk, b)
            assert np.abs(gl["U"][:, b] - r["U"]).max() < 5e-6 and np.abs(gl["X_HAT"][:, b] - r["X_HAT"]).max() < 5e-6, (lk, b)
            dU = np.diff(np.vstack([pc.u0, gl["U"][:, b]]), axis=0)
            assert (dU >= pc.Dumin - 1e-7).all() and (dU <= pc.Dumax + 1e-7).all()


def test_terminal_equality(pkg, solver_factory):
    """TermCons (Control_Calc.py:193-198): x_N = xs.  Every kernel against the dense statement with the equality rows
    (oracle/mpc_oracle.py:ocp_qp): per call (lane and wave solver) and in the fused closed loop (three kernels); statuses include
    'unreachable' (status 2, hold rule) at the short horizons.  The equality is carried by the terminal weight (mpc_amd.hip:
    build_problem), which leaves a miss of |multiplier| / 1e12; the lane solver and the wave-autonomous one then aim the terminal reference
    off by the miss and solve again (mpc_device.hpp:term_aim, mpc_amd.hip:wv_term_aim): x_N = xs to rounding and u to 1e-7 at every horizon,
    also where the multipliers are of order 1e6 (N = 6, 3).  The horizon-parallel kernel carries the weight alone: 1e-6 at N = 20, 1e-4 at N = 6."""
    import mpc_oracle as o
    from mpc_code_amd import capi
    from mpc_code_amd.driver import run_closed_loop
    for N, tol_wave in ((20, 1e-6), (6, 1e-4), (3, None)):
        p = pkg.load_problem(pkg.example_path("cstr_lmpc.py"), overrides={"N": N, "TermCons": True})
        assert p.TermCons
        rng = np.random.default_rng(5)
        B = 40
        xh = rng.uniform([-0.5, -8, -5], [0.5, 8, 5], size=(B, 3)); d = rng.normal(size=(B, 3)) * 0.02; up = np.zeros((B, 2))
        xs = np.zeros((B, 3)); us = np.zeros((B, 2))
        for b in range(B):
            t = o.target_solve(p, np.zeros(2), np.array([0.1, 0, 0.2]), np.zeros(3), d[b], up[b]); xs[b], us[b] = t["xs"], t["us"]
        ref = [o.ocp_solve(p, xh[b], xs[b], us[b], d[b], up[b]) for b in range(B)]
        rst = np.array([r["status"] for r in ref])
        assert ((rst == 0).sum() >= 30) == (N > 3) and ((rst == 2).sum() >= 5) == (N <= 6) and (rst == 0).sum() >= 5
        s = solver_factory(p)
        assert s.get_option("ocp_kernel") == 3 and s.get_option("loop_kernel") == 3      # an exact one by default
        for ok in (1, 3):
            tol = 1e-7
            s = solver_factory(p); s.set_option("ocp_kernel", ok)
            g = s.ocp_solve(xh, xs, us, d, up, want_w=True)
            assert np.array_equal(g["status"], rst), (N, ok)
            good = rst == 0
            assert max(np.abs(g["u0"][b] - ref[b]["u0"]).max() for b in np.flatnonzero(good)) < tol, (N, ok, max(np.abs(g["u0"][b] - ref[b]["u0"]).max() for b in np.flatnonzero(good)))
            assert np.abs(g["w"][good][:, -3:] - xs[good]).max() < 1e-10, (N, ok, np.abs(g["w"][good][:, -3:] - xs[good]).max())          # the terminal state sits on xs
        x0 = rng.uniform([-0.3, -4, -3], [0.3, 4, 3], size=(6, 3))
        cl = [o.closed_loop(p, 6, x0_p=x, x0_m=x) for x in x0]
        U = np.stack([c["U"] for c in cl], axis=1); ST = np.stack([c["STATUS_DYN"] for c in cl], axis=1)
        assert (ST == 2).any() and ((ST == 0).sum() > 20) == (N > 3)
        for lk in (0, 1, 2, 3):
            tol = 1e-7 if lk != 2 else tol_wave      # (the horizon-parallel kernel carries the weight alone)
            if tol is None:
                continue
            r = run_closed_loop(p, x0, x0, 6, solver=solver_factory(p, lk))
            assert np.array_equal(r["STATUS_DYN"], ST), (N, lk)
            # (at the short horizons the loop is ill-conditioned - multipliers of order 1e6: a difference of 4e-12 at step 0 is 4e-9 at step 1 and
            # 1e-5 at step 5 whatever the solver - so the tight comparison is for the first two steps)
            assert np.abs(r["U"][:2] - U[:2]).max() < tol and np.abs(r["U"] - U).max() < max(tol, 1e-3), (N, lk, np.abs(r["U"] - U).max(axis=(1, 2)).tolist())


def _with_model_params(pkg, N=20, **extra):
    def def_px(t):
        return [np.array([0.02 * np.sin(0.3 * t), 0.15 * np.cos(0.2 * t), 0.05 * np.sin(0.1 * t + 1.0)])]

    def def_py(t):
        return [np.array([0.03 * np.cos(0.25 * t), 0.4 * np.sin(0.15 * t), 0.0])]
    ov = {"N": N, "def_px": def_px, "def_py": def_py}; ov.update(extra)
    return pkg.load_problem(pkg.example_path("cstr_lmpc.py"), overrides=ov)


def test_model_parameters_over_the_horizon(pkg, solver_factory):
    """def_px / def_py (MPC_code.py:492-510): px_k in the dynamics, py_k in the output rows of every stage, p_x_k / p_y_k in the
    estimator's predicted output, the target equalities and the plant.  mpc_ocp_solve(px, py) and mpc_set_model_offsets against the
    dense statements with par_xmk / par_ymk (oracle/mpc_oracle.py), per call and over a closed loop of the three calls."""
    import mpc_oracle as o
    from mpc_code_amd.driver import run_closed_loop
    p = _with_model_params(pkg)
    assert p.has_model_params
    rng = np.random.default_rng(11)
    B = 32
    pxh, pyh = p.horizon_params(3.0)
    assert np.allclose(pxh[4], p.def_px(3.0 + 4)[0]) and pxh.shape == (20, 3) and pyh.shape == (20, 3)      # t_k + i, not t_k + i h
    xh = rng.uniform([-0.4, -6, -4], [0.4, 6, 4], size=(B, 3)); d = rng.normal(size=(B, 3)) * 0.02; up = np.zeros((B, 2))
    s = solver_factory(p)
    # target with p_x_k, p_y_k
    s.set_model_offsets(B, pxh[0], pyh[0])
    t = s.target_solve(np.zeros(2), np.array([0.1, 0, 0.2]), np.zeros(3), d, up)
    tr = [o.target_solve_exact(p, np.zeros(2), np.array([0.1, 0, 0.2]), np.zeros(3), d[b], up[b], px0=pxh[0], py0=pyh[0]) for b in range(B)]
    assert np.array_equal(t["status"], [r["status"] for r in tr]) and all(r["exact"] for r in tr)
    # against the exact optimum (active-set polish verified); the bounds of this target are weakly active: 1e-6
    assert max(np.abs(t["xs"][b] - tr[b]["xs"]).max() for b in range(B)) < 1e-6 and max(np.abs(t["us"][b] - tr[b]["us"]).max() for b in range(B)) < 1e-6
    t0 = s.target_solve(np.zeros(2), np.array([0.1, 0, 0.2]), np.zeros(3), d, up)
    s.set_model_offsets(B, None, None)
    t1 = s.target_solve(np.zeros(2), np.array([0.1, 0, 0.2]), np.zeros(3), d, up)
    assert np.array_equal(t0["xs"], t["xs"]) and np.abs(t1["xs"] - t["xs"]).max() > 1e-3            # the offsets matter, and clear
    # OCP with the horizon values, per instance different (shifted in time)
    PX = np.stack([p.horizon_params(0.7 * b)[0] for b in range(B)]); PY = np.stack([p.horizon_params(0.7 * b)[1] for b in range(B)])
    g = s.ocp_solve(xh, t["xs"], t["us"], d, up, want_w=True, px=PX, py=PY)
    ref = [o.ocp_solve_exact(p, xh[b], t["xs"][b], t["us"][b], d[b], up[b], px=PX[b], py=PY[b]) for b in range(B)]
    rst = np.array([r["status"] for r in ref])
    assert np.array_equal(g["status"], rst) and (rst == 0).sum() > B // 2
    good = np.flatnonzero(rst == 0)
    assert max(np.abs(g["u0"][b] - ref[b]["u0"]).max() for b in good) < 1e-6 and max(np.abs(g["x1"][b] - ref[b]["x1"]).max() for b in good) < 1e-6
    assert max(np.abs(g["w"][b] - ref[b]["w"]).max() for b in good) < 1e-5                        # the whole trajectory
    plain = s.ocp_solve(xh, t["xs"], t["us"], d, up)
    assert np.abs(plain["u0"][good] - g["u0"][good]).max() > 1e-3                                   # the parameters matter
    # an output row moved by py_k binds somewhere: y_k = C x_k + py_k within [ymin, ymax] at every stage
    w = g["w"][good].reshape(len(good), -1)
    X = np.stack([w[:, k * 5:k * 5 + 3] for k in range(1, p.N)], axis=1)
    Y = X @ p.C.T + p.fy_const + (d[good] @ p.Cd.T)[:, None, :] + PY[good][:, 1:p.N]
    assert (Y <= p.ymax + 1e-7).all() and (Y >= p.ymin - 1e-7).all()
    # closed loop of the three calls
    x0 = rng.uniform([-0.3, -4, -3], [0.3, 4, 3], size=(5, 3))
    cl = [o.closed_loop(p, 8, x0_p=x, x0_m=x, ocp=o.ocp_solve_exact, target=o.target_solve_exact) for x in x0]
    r = run_closed_loop(p, x0, x0, 8, solver=s, fused=False)
    for k in ("U", "X_HAT", "XS", "US", "Xp", "Yp", "D_HAT"):
        ref_k = np.stack([c[k] for c in cl], axis=1)
        assert np.abs(r[k] - ref_k).max() < 1e-6, k
    assert np.array_equal(r["STATUS_DYN"], np.stack([c["STATUS_DYN"] for c in cl], axis=1))
    # the same loop fused (mpc_loop_set_model_schedule: all steps in one launch): the call-by-call numbers, hence the dense statements'
    f = run_closed_loop(p, x0, x0, 8, solver=s, fused=True)
    for k in ("U", "X_HAT", "XS", "US", "Xp", "Yp", "Y_HAT", "YS", "D_HAT"):
        assert np.abs(f[k] - r[k]).max() < 1e-8, (k, np.abs(f[k] - r[k]).max())      # (step 0 equal to the bit; then the host's measurement sums in another order)
        assert np.array_equal(f[k][0], r[k][0]) or k in ("Yp", "Y_HAT", "YS"), k
        if k in cl[0]:
            assert np.abs(f[k] - np.stack([c[k] for c in cl], axis=1)).max() < 1e-6, k
    for k in ("STATUS_DYN", "STATUS_SS", "ITERS_DYN", "ITERS_SS"):
        assert np.array_equal(f[k][0], r[k][0]), (k, f[k].tolist(), r[k].tolist())
        # (later steps: the two loops' states differ by 1e-10, and an iteration count may sit on a threshold of the algorithm)
        assert np.array_equal(f[k], r[k]) or (k.startswith("ITERS") and np.abs(f[k] - r[k]).max() <= 3 and (f[k] != r[k]).mean() < 0.2), (k, f[k].tolist(), r[k].tolist())
    # a ragged batch over two launches, one of the two schedules only, and back to a loop without them on the same handle
    q = _with_model_params(pkg, def_py=None)
    assert q.def_py is None and q.has_model_params
    x1 = rng.uniform([-0.3, -4, -3], [0.3, 4, 3], size=(70, 3))
    sq = solver_factory(q)
    a = run_closed_loop(q, x1, x1, 6, solver=sq, fused=False)
    sq.set_option("steps_per_launch", 4)
    b = run_closed_loop(q, x1, x1, 6, solver=sq, fused=True)
    assert np.abs(a["U"] - b["U"]).max() < 1e-8 and np.array_equal(a["STATUS_DYN"], b["STATUS_DYN"]) and np.array_equal(a["ITERS_DYN"][0], b["ITERS_DYN"][0])
    plain = pkg.load_problem(pkg.example_path("cstr_lmpc.py"), overrides={"N": 20})
    sp = solver_factory(plain)
    c0 = run_closed_loop(plain, x1, x1, 6, solver=sp)
    sp.loop_alloc(len(x1), 6, 2); sp.loop_set_state(x1, x1); sp.loop_set_schedule(plain.schedules(6))
    sp.loop_set_model_schedule(np.zeros((6, 20, 3)), None)      # zero parameters through the scheduled kernel: the plain loop's numbers
    sp.loop_run(0, 6); sp.loop_sync()
    assert np.abs(sp.loop_get_log("U") - c0["U"]).max() < 1e-7 and np.array_equal(sp.loop_get_log("STATUS_DYN"), c0["STATUS_DYN"])


def test_fused_closed_loop_with_the_user_plant(nlplant, xp_nlplant):
    """Ex_LMPC_nlplant / Ex_LMPCxp_nlplant: linear controller, the Ex-file's own plant function as the process.  The function is
    traced and compiled into the problem's own library (capi.Solver, nlcodegen.emit_plant_header; mpc_amd.hip:plant_next integrates it
    with Mx RK4 steps), so the whole loop runs in one kernel; it must equal the call-by-call mode with the plant on the host, which
    tests above pin to the oracle.  Both models are open-loop unstable (|A^32| > 1e4): the library then picks the instance-per-lane
    kernel and cold starts, and differences between two runs grow about threefold per step, hence 12 steps."""
    from mpc_code_amd import capi
    from mpc_code_amd.driver import run_closed_loop
    rng = np.random.default_rng(3)
    for p, tol_other in ((nlplant, 2e-3), (xp_nlplant, 1e-4)):
        B, K = 48, 12
        x0p = p.x0_p + rng.normal(size=(B, p.nxp)) * [2e-4, 0.02, 2e-4]
        x0m = np.hstack([x0p, np.zeros((B, p.nx - p.nxp))]) if p.nx > p.nxp else x0p
        s = capi.Solver(p)
        assert s.fused_plant and s.build_info().endswith(";nlplant") and s.get_option("loop_kernel") == 1
        ref = run_closed_loop(p, x0p, x0m, K, solver=s, fused=False)
        r = run_closed_loop(p, x0p, x0m, K, solver=s, fused=True)
        assert np.array_equal(r["STATUS_DYN"], ref["STATUS_DYN"]) and np.array_equal(r["STATUS_SS"], ref["STATUS_SS"])
        for k in ("U", "Xp", "X_HAT", "XS", "D_HAT"):
            assert np.abs(r[k] - ref[k]).max() < 1e-4, k
        assert np.abs(r["Yp"] - ref["Yp"]).max() < 1e-4
        for lk in (2, 3):          # the other kernels run it too (looser: their recursions are scans with powers of an unstable A)
            s.set_option("loop_kernel", lk)
            r2 = run_closed_loop(p, x0p, x0m, K, solver=s, fused=True)
            assert np.array_equal(r2["STATUS_DYN"], ref["STATUS_DYN"]) and np.abs(r2["U"] - ref["U"]).max() < tol_other, lk
        s.close()
    s = capi.Solver(nlplant, lib_path=capi.LIB_PATH)            # the default library has no plant function: the plant stays on the host
    assert not s.fused_plant
    with pytest.raises(ValueError, match="no compiled plant"):
        run_closed_loop(nlplant, nlplant.x0_p[None], nlplant.x0_m[None], 3, solver=s, fused=True)
    s.close()


def test_instances_per_wave_follow_the_batch_size_and_do_not_change_the_loop(cstr, wb, oracle_c, solver_factory):
    """The wave-autonomous kernel takes one, two or four instances per wave by the size of the batch (a wave is alone on its SIMD, so
    a small batch spreads over more SIMDs): an instance's arithmetic does not depend on its neighbours - the same closed loop bit for
    bit whatever the packing, and the C restatement's loop; ragged batches leave instance slots of the last wave empty."""
    from mpc_code_amd.driver import run_closed_loop
    for p, B, K in ((cstr, 203, 12), (wb, 61, 14)):
        x0 = bench_x0(B, 5) if p is cstr else np.zeros((B, p.nx)) + np.random.default_rng(3).normal(size=(B, p.nx)) * 0.05
        runs = {}
        for ni in (0, 1, 2, 4):
            s = solver_factory(p, 3)
            s.set_option("wave_instances", ni)
            runs[ni] = run_closed_loop(p, x0, x0, K, solver=s)
        for ni in (1, 2, 4):
            for k in ("U", "XS", "US", "X_HAT", "STATUS_DYN", "STATUS_SS", "ITERS_DYN", "ITERS_SS"):
                assert np.array_equal(runs[ni][k], runs[0][k]), (p.name, ni, k)
        c = oracle_c.OracleC(p).closed_loop(K, x0, x0)
        assert assert_same_closed_loop(runs[0], c, p, TOL_PORT, max_flipped=0.0) == 0
    s = solver_factory(cstr, 3)
    with pytest.raises(Exception):
        s.set_option("wave_instances", 3)
