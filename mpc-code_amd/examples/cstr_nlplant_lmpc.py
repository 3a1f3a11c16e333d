"""Linear offset-free MPC (cost on input moves) of a non-linear CSTR plant.

Written for this project in the Ex-file surface of CPCLAB-UNIPI/MPC-code (names documented in its
User_Guide.pdf ch. 3).  Numeric data are those of the reference's ``Ex_LMPC_nlplant.py`` (reactor parameters
``:56-67``, linearised model and its operating point ``:85-91``, disturbance model ``:94-96``, KF tuning
``:104-115``, set-point schedule ``:135-144``, bounds ``:148-154``, weights ``:157-162``) so that both files define
the same problem; tests/test_exfile.py checks that when the reference tree is present.  The controller (estimator,
target, OCP) is linear around ``xlin, ulin``; only the simulated plant is non-linear: its right-hand side is
integrated by RK4 with ``Mx`` sub-steps per sampling interval (reference ``Utilities.py:58-82``), on the host.
"""
from casadi import *
import math
import numpy as np
import scipy.linalg as scla

Nsim, N, h = 200, 50, 0.2

xp = SX.sym("xp", 3)     # plant state: concentration [kmol/m^3], temperature [K], level [m]
x = SX.sym("x", 3)
u = SX.sym("u", 2)       # coolant temperature [K], outlet flow [m^3/min]
y = SX.sym("y", 2)
d = SX.sym("d", 2)

# ---- plant: mass and energy balance of a jacketed tank with a first-order exothermic reaction
FEED_FLOW, FEED_TEMP, FEED_CONC = 0.1, 350, 1.0
TANK_RADIUS = 0.219
K_ARRHENIUS, E_OVER_R = 7.2e10, 8750
HEAT_TRANSFER = 915.6 * 60 / 1000
DENSITY, HEAT_CAPACITY, REACTION_HEAT = 1000.0, 0.239, -5.0e4
Mx = 10


def User_fxp_Cont(x, t, u, pxp, pxmp):
    conc, temp, level = x[0], x[1], x[2]
    coolant, outflow = u[0], u[1]
    area = math.pi * TANK_RADIUS ** 2
    rate_at_feed_temp = K_ARRHENIUS * exp(-E_OVER_R / FEED_TEMP)
    rate = rate_at_feed_temp * exp(-E_OVER_R * (1.0 / temp - 1.0 / FEED_TEMP)) * conc
    d_conc = FEED_FLOW * (FEED_CONC - conc) / (area * level) - rate
    d_temp = (FEED_FLOW * (FEED_TEMP - temp) / (area * level) - REACTION_HEAT / (DENSITY * HEAT_CAPACITY) * rate
              + 2 * HEAT_TRANSFER / (TANK_RADIUS * DENSITY * HEAT_CAPACITY) * (coolant - temp))
    d_level = (FEED_FLOW - outflow) / area
    return vertcat(d_conc, d_temp, d_level)


Cp = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])

# ---- controller model: linearisation around (xlin, ulin), sampled with h
A = np.array([[0.51448, -0.00917517, -0.117995], [53.6817, 2.15004, -3.77725], [0.0, 0.0, 1]])
B = np.array([[-0.0017669, 0.0864569], [0.639423, 1.60696], [0.0, -1.32737]])
C = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
xlin = np.array([0.5, 350, 0.659])
ulin = np.array([300, 0.1])

offree = "lin"
Bd = B.copy()
Cd = np.zeros((2, 2))

x0_p = np.array([0.5, 350, 0.659])
x0_m = np.array([0.5, 350, 0.659])
u0 = np.array([300, 0.1])

kal = True
Q_kf = scla.block_diag(1.0e-5 * np.eye(3), np.eye(2))
R_kf = 1.0e-4 * np.eye(2)
P0 = 1e-3 * Q_kf


def defSP(t):
    xsp = np.zeros(3)
    usp = np.array([299.963, 0.1])
    if t < 20:
        ysp = np.array([0.5, 0.659])
    elif t < 40:
        ysp = np.array([0.51, 0.659])
    else:
        ysp = np.array([0.50, 0.659])
    return [ysp, usp, xsp]


umin = np.array([295, 0.00])
umax = np.array([305, 0.25])
xmin = np.array([0.0, 320, 0.45])
xmax = np.array([1.0, 375, 0.75])

Qss = np.array([[10.0, 0.0], [0.0, 0.01]])
Rss = np.zeros((2, 2))
Q = np.array([[10.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
S = np.array([[0.1, 0.0], [0.0, 0.1]])
