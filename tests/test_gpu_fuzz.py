"""Randomised problems through the C-ABI against the oracle: the same kernels on data nobody tuned them for.

Dimension sets of the default library only (no build on the box): 3/2/3/3/3 (the benchmark's), 2/1/1/1/2 with and without the cost on
input moves.  Per problem: random stable or mildly unstable A, random B, SPD weights, DARE terminal cost, random horizon, random
boxes on inputs and on some states / outputs.  Per call against the dense statement's exact optimum (verified active-set polish),
closed loop against the C restatement on every loop kernel."""
import numpy as np
import pytest
import scipy.linalg as scla

pytestmark = pytest.mark.gpu


def random_problem(seed, nx, nu, ny, du):
    from mpc_code_amd.problem import LinearMPCProblem
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((nx, nx)); A *= rng.uniform(0.6, 1.04) / np.abs(np.linalg.eigvals(A)).max()
    B = rng.standard_normal((nx, nu))
    C = np.eye(ny, nx)
    nd = ny
    Bd = 0.3 * rng.standard_normal((nx, nd)); Cd = np.zeros((ny, nd))
    Q = np.diag(rng.uniform(0.05, 2.0, nx)); R = np.diag(rng.uniform(0.05, 1.0, nu))
    P = scla.solve_discrete_are(A, B, Q, R); P = 0.5 * (P + P.T)
    inf = np.inf
    umax = rng.uniform(0.5, 2.0, nu); umin = -rng.uniform(0.5, 2.0, nu)
    xmax = np.where(rng.random(nx) < 0.6, rng.uniform(1.0, 4.0, nx), inf); xmin = np.where(rng.random(nx) < 0.6, -rng.uniform(1.0, 4.0, nx), -inf)
    ymax = np.where(rng.random(ny) < 0.5, rng.uniform(1.0, 3.0, ny), inf); ymin = np.where(rng.random(ny) < 0.5, -rng.uniform(1.0, 3.0, ny), -inf)
    y_bounded = bool(np.isfinite(ymax).any() or np.isfinite(ymin).any())
    Ca = np.hstack([C, Cd]); Aa = np.eye(nx + nd); Aa[:nx, :nx] = A; Aa[:nx, nx:] = Bd
    Pe = scla.solve_discrete_are(Aa.T, Ca.T, np.eye(nx + nd) * 0.1, np.eye(ny) * 0.1)
    K = Pe @ Ca.T @ np.linalg.inv(Ca @ Pe @ Ca.T + 0.1 * np.eye(ny))
    return LinearMPCProblem(nx=nx, nu=nu, ny=ny, nd=nd, nxp=nx, N=int(rng.integers(6, 41)), h=1.0, Nsim=10, A=A, B=B, C=C, Bd=Bd, Cd=Cd,
                            fx_const=np.zeros(nx), fy_const=np.zeros(ny), Ap=A, Bp=B, Cp=C, Q=Q, R=R, DUForm=du, P=P,
                            Qss=np.eye(ny), Rss=np.zeros((nu, nu)), DUssForm=False, umin=umin, umax=umax, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax,
                            y_bounded=y_bounded, umin_ss=umin, umax_ss=umax, xmin_ss=xmin, xmax_ss=xmax, ymin_ss=np.full(ny, -inf), ymax_ss=np.full(ny, inf),
                            estimator="kalss", K=K, x0_p=np.zeros(nx), x0_m=np.zeros(nx), u0=np.zeros(nu), dhat0=np.zeros(nd))


CASES = [(s, 3, 2, 3, False) for s in range(101, 107)] + [(s, 2, 1, 1, False) for s in range(201, 205)] + [(s, 2, 1, 1, True) for s in range(301, 305)]


@pytest.mark.parametrize("seed,nx,nu,ny,du", CASES)
def test_random_problem_per_call_against_the_exact_optimum(seed, nx, nu, ny, du, solver_factory):
    import mpc_oracle as o
    p = random_problem(seed, nx, nu, ny, du)
    rng = np.random.default_rng(seed + 7)
    B = 24
    scale = np.where(np.isfinite(p.xmax), p.xmax, 3.0)
    xh = rng.uniform(-0.8, 0.8, (B, nx)) * scale; d = 0.1 * rng.standard_normal((B, p.nd)); up = rng.uniform(-0.3, 0.3, (B, nu))
    xs = 0.1 * rng.standard_normal((B, nx)); us = 0.1 * rng.standard_normal((B, nu))
    ref = [o.ocp_solve_exact(p, xh[b], xs[b], us[b], d[b], up[b]) for b in range(B)]
    rst = np.array([r["status"] for r in ref])
    for ok in (1, 3):
        s = solver_factory(p); s.set_option("ocp_kernel", ok)
        g = s.ocp_solve(xh, xs, us, d, up)
        # feasibility labels: equal, except where the dense statement itself sits on the fence (its interior point stalls: status 1)
        clear = rst != 1
        assert np.array_equal(g["status"][clear] == 2, rst[clear] == 2), (seed, ok, g["status"], rst)
        good = [b for b in range(B) if rst[b] == 0 and ref[b]["exact"] and g["status"][b] == 0]
        assert len(good) >= (B // 4 if (rst == 0).sum() >= B // 2 else 0)
        if good:
            err = max(np.abs(g["u0"][b] - ref[b]["u0"]).max() for b in good)
            assert err < 2e-6, (seed, ok, err)


@pytest.mark.parametrize("seed,nx,nu,ny,du", CASES[::2])
def test_random_problem_closed_loop_against_the_c_restatement(seed, nx, nu, ny, du, solver_factory, oracle_c):
    from mpc_code_amd.driver import run_closed_loop
    p = random_problem(seed, nx, nu, ny, du)
    rng = np.random.default_rng(seed + 11)
    B, K = 96, 15
    scale = np.where(np.isfinite(p.xmax), p.xmax, 3.0)
    x0 = rng.uniform(-0.5, 0.5, (B, nx)) * scale
    ref = oracle_c.OracleC(p).closed_loop(K, x0, x0)
    for lk in (1, 2, 3):
        r = run_closed_loop(p, x0, x0, K, solver=solver_factory(p, lk))
        # every status word of every instance-step equal (tools/flip_report.py: no label differs on any of these problems), then values
        assert np.array_equal(r["STATUS_DYN"], ref["STATUS_DYN"]) and np.array_equal(r["STATUS_SS"], ref["STATUS_SS"]), (seed, lk)
        assert np.abs(r["U"] - ref["U"]).max() < 1e-6, (seed, lk)
