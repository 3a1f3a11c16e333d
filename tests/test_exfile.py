"""Ex-file surface: the loader runs Ex-style files without CasADi and yields the reference's problem."""
import os

import numpy as np
import pytest

from conftest import REF

FIELDS = ["A", "B", "C", "Bd", "Cd", "Ap", "Bp", "Cp", "Q", "R", "P", "Qss", "Rss", "umin", "umax", "xmin", "xmax",
          "ymin", "ymax", "x0_p", "x0_m", "u0"]


def test_cstr_dimensions_and_flags(cstr):
    # reference MPC_code.py:31-52 on Ex_LMPC_CSTR: nw = 253; kal = True -> time-varying Kalman filter
    assert (cstr.nx, cstr.nu, cstr.ny, cstr.nd, cstr.N, cstr.nw) == (3, 2, 3, 3, 50, 253)
    assert cstr.estimator == "kal" and not cstr.DUForm and cstr.y_bounded and cstr.max_iter == 100
    # DARE terminal weight: the same SciPy call as Utilities.py:409; SURVEY.md 8a7 quotes its spectrum
    ev = np.linalg.eigvalsh(cstr.P)
    assert np.allclose(ev, [1.49963e-05, 1.064095, 5.323557], rtol=1e-5)
    assert np.allclose(cstr.A.T @ cstr.P @ cstr.A - cstr.P - cstr.A.T @ cstr.P @ cstr.B @ np.linalg.solve(
        cstr.R + cstr.B.T @ cstr.P @ cstr.B, cstr.B.T @ cstr.P @ cstr.A) + cstr.Q, 0, atol=1e-9)


def test_wb_is_delta_u_form(wb):
    # Ex_LMPC_WB gives S and no R -> DUForm (MPC_code.py:237-239), DARE with R <- S (:253-255); lue -> fixed gain
    assert (wb.nx, wb.nu, wb.ny, wb.nd, wb.nw) == (4, 2, 2, 2, 304)
    assert wb.DUForm and wb.estimator == "kalss" and not wb.y_bounded
    assert np.array_equal(wb.K, np.vstack([np.zeros((4, 2)), np.eye(2)]))
    assert np.all(np.isinf(wb.xmin)) and np.allclose(wb.umax, 0.5)


def test_schedules_follow_callbacks(cstr):
    s = cstr.schedules(30)
    assert np.array_equal(s["ysp"][15], [0.2, 0, 0]) and np.array_equal(s["ysp"][16], [0, 0, 0.1])   # Ex_LMPC_CSTR.py:119-141
    assert np.array_equal(s["pxp"][20], [0.1, 0, 0]) and np.array_equal(s["pxp"][21], [0, 0, 0])     # :40-57
    assert np.array_equal(s["pyp"][7], [0.1, 0.1, 0])                                               # :59-79


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("ours,theirs", [("cstr_lmpc.py", "Ex_LMPC_CSTR.py"), ("wood_berry_lmpc.py", "Ex_LMPC_WB.py")])
def test_unmodified_reference_examples_load_to_the_same_problem(pkg, ours, theirs):
    a = pkg.load_problem(pkg.example_path(ours))
    b = pkg.load_problem(os.path.join(REF, theirs))
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    for t in (0, 10, 11, 15, 16, 20, 21, 99):
        for x, y in zip(a.defSP(t), b.defSP(t)):
            assert np.array_equal(np.ravel(x), np.ravel(y))
    assert a.estimator == b.estimator and a.DUForm == b.DUForm and a.N == b.N == 50


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", ["Ex_NMPC.py", "Ex_ENMPC.py", "Ex_NMPC_dis.py", "Ex_LMPC_nlplant.py"])
def test_examples_outside_the_linear_path_are_refused_loudly(pkg, name):
    with pytest.raises(pkg.UnsupportedProblem):
        pkg.load_problem(os.path.join(REF, name))


def test_overrides_apply_after_the_file(pkg):
    p = pkg.load_problem(pkg.example_path("cstr_lmpc.py"), overrides={"N": 30})
    assert p.N == 30 and p.nw == 3 * 31 + 2 * 30
