"""Economic NMPC of an isothermal reactor (A -> B -> C) with a moving-horizon estimator.

Written for this project in the Ex-file surface of CPCLAB-UNIPI/MPC-code (names documented in its User_Guide.pdf ch. 3).  It poses
the same problem as the reference's ``Ex_ENMPC.py`` - same balances and rate constants (``:35-49``, ``:64-65``), sampling time,
horizons and simulation length (``:18-22``, ``:126``), full state measurement with a linear output-disturbance model (``:33``,
``:96-98``), starting points (``:101-103``), estimator data (``:125-177``), bounds (``:184-191``), profit functions
(``:194-233``), terminal weight (``:236-252``) and iteration limit (``:255``) - so that the GPU tests and the benchmark have the
BASELINE configurations' problem on a box that has no ``/root/reference``; ``tests/test_enmpc.py`` checks, where the reference is
present, that the two files give the same numbers.  The controller maximises the profit rate of the product:
``dilution * (price_A * feed - price_B * c_B)`` is the *cost* per unit time, integrated over every sampling interval.
"""
from casadi import *
import numpy as np
from Utilities import *

Nsim, N, h = 21, 25, 2.0

xp = SX.sym("xp", 2)     # plant state: concentrations of A and B [kmol/m^3]
x = SX.sym("x", 2)
u = SX.sym("u", 1)       # dilution rate F / V [1/min]
y = SX.sym("y", 2)
d = SX.sym("d", 2)

StateFeedback = True     # both concentrations are measured

FEED_CONC, VOLUME = 1.0, 1.0
K1, K2 = 1.0, 0.05
PRICE_A, PRICE_B = 1.0, 4.0
Mx = 10


def _balances(c_a, c_b, flow):
    return vertcat(flow * (FEED_CONC - c_a) / VOLUME - K1 * c_a, -flow * c_b / VOLUME + K1 * c_a - K2 * c_b)


def User_fxp_Cont(xp, t, u, pxp, pxmp):
    return _balances(xp[0], xp[1], u[0])


def User_fxm_Cont(x, u, d, t, px):
    return _balances(x[0], x[1], u[0])


offree = "lin"           # output disturbance: y = x + d
Bd = np.zeros((2, 2))
Cd = np.eye(2)

x0_p = np.array([0.9, 0.1])
x0_m = np.array([1.2, 0.5])
u0 = np.array([0.0])

# extended Kalman filter on [x; d] (the other position of the reference example's estimator switch, Ex_ENMPC.py:109-123): the state is trusted, the
# disturbance is what the filter moves
ekf = True
Q_kf = np.diag([1.0e-8, 1.0e-8, 1.0, 1.0])
R_kf = 1.0e-8 * np.eye(2)
P0 = 1.0e-8 * np.eye(4)


umin = [0.0]
umax = [2.0]
xmin = np.array([0.0, 0.0])
xmax = np.array([1.0, 1.0])


def _profit_rate_cost(u, y):
    return u[0] * (PRICE_A * FEED_CONC - PRICE_B * y[1])


def User_fssobj(x, u, y, xsp, usp, ysp):
    return _profit_rate_cost(u, y)


def User_fobj_Cont(x, u, y, xs, us, ys):
    return _profit_rate_cost(u, y)


def User_vfin(x, xs):
    e = x - xs
    return mtimes(e.T, mtimes(2000, e))


Sol_itmax = 200
ContForm = True
