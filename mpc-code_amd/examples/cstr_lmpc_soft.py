"""The CSTR benchmark problem with SOFT output bounds (`slacks = True`, reference Control_Calc.py:39-40,186-192,228-239; Default_Values.py:128-131; `Ws`
MPC_code.py:55-57): one slack vector Sl = [sl_ub; sl_lb] >= 0, shared by all stages, widens the output box of every stage and pays Sl' Ws Sl in every stage's cost.

With hard bounds the shipped scenario starts infeasible: from x0 = [3, 3, 3] the bound x2 <= 10 cannot be kept and the reference holds the
input for three steps.  Here the state bounds are dropped, the same box is asked of the outputs (C = I) and softened: every OCP of the run is feasible, the first ones
pay for the excursion with their slacks.
"""
import os as _os

exec(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "cstr_lmpc.py")).read())      # the benchmark problem's data

xmin = None
xmax = None
slacks = True
Ws = 100.0 * np.eye(2 * 3)
