"""The shipped CSTR scenario (cstr_lmpc.py) with two affine user inequality rows in the OCP (the reference's ``User_g_ineq``, Control_Calc.py:94-100,132-147):
the two inputs may not both be large, ``u_0 + 0.5 u_1 <= 5``, and the first output is tied to the first input, ``y_0 - 0.02 u_0 <= 0.2``."""
import os
import runpy

globals().update({k: v for k, v in runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cstr_lmpc.py")).items() if not k.startswith("__")})


def User_g_ineq(x, u, y, d, t, px, py):
    return vertcat(u[0] + 0.5 * u[1] - 5.0, y[0] - 0.02 * u[0] - 0.2)      # noqa: F821 (vertcat: the Ex-file surface)
