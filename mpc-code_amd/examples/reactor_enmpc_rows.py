"""The economic reactor problem (examples/reactor_enmpc.py = the reference's Ex_ENMPC.py) with USER INEQUALITY ROWS in the OCP: ``User_g_ineq(x, u, y, d, t, px, py) <= 0``
at every stage of the horizon (reference Control_Calc.py:94-100,132-147; MPC_code.py:306-314) - one affine row in input and output, one non-linear one:

    u + 0.5 y_A <= 1.2        the feed rate is limited by what the outlet still holds of A
    u c_A       <= 0.40       the molar flow of unconverted A leaving the reactor

The economic optimum of the unconstrained problem (u = 1.043, c_A = 0.415) violates both: the loop settles on the rows.
"""
import os as _os

exec(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "reactor_enmpc.py")).read())      # the example's data and functions


def User_g_ineq(x, u, y, d, t, px, py):
    return vertcat(u[0] + 0.5 * y[0] - 1.2, u[0] * x[0] - 0.40)
