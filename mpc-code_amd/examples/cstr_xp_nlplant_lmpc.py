"""Linear offset-free MPC of the non-linear CSTR with a controller model that has one state more than the plant.

Written for this project in the Ex-file surface of CPCLAB-UNIPI/MPC-code (names documented in its
User_Guide.pdf ch. 3).  Numeric data are those of the reference's ``Ex_LMPCxp_nlplant.py`` (reactor parameters
``:59-70``, linearised model ``:88-101`` with its operating point ``:103-105``, disturbance model ``:108-110``, KF
tuning ``:120-125``, set-point schedule ``:145-152``, bounds ``:155-162``, weights ``:165-171``) so that both files
define the same problem; tests/test_exfile.py checks that when the reference tree is present.

What this example adds to ``cstr_nlplant_lmpc.py``: the model carries a fourth state, a first-order lag of the
coolant temperature move that leaks into the first output, so ``nx = 4 != nxp = 3``; the output map has an offset
``ylin``; and the outputs are bounded.  The first output row, ``y0 = x0 + 0.001 x3``, touches two states: it is not
a box on a state but a general output row of the OCP (``Control_Calc.py:130,150-151,229-230``).
"""
from casadi import *
import math
import numpy as np
import scipy.linalg as scla

Nsim, N, h = 200, 50, 0.2

xp = SX.sym("xp", 3)     # plant state: concentration [kmol/m^3], temperature [K], level [m]
x = SX.sym("x", 4)       # model state: the three above and the coolant lag
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

# ---- controller model: the linearisation of cstr_nlplant_lmpc.py, extended by the lag state
A_REACTOR = np.array([[0.51448, -0.00917517, -0.117995], [53.6817, 2.15004, -3.77725], [0.0, 0.0, 1]])
B_REACTOR = np.array([[-0.0017669, 0.0864569], [0.639423, 1.60696], [0.0, -1.32737]])
C_REACTOR = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
LAG_POLE = 0.01
A = scla.block_diag(A_REACTOR, LAG_POLE)
B = np.vstack([B_REACTOR, [[1.0 - LAG_POLE, 0.0]]])
C = np.hstack([C_REACTOR, (LAG_POLE / 10.0) * np.array([[1.0], [0.0]])])
xlin = np.array([0.5, 350, 0.659, 0.0])
ulin = np.array([300, 0.1])
ylin = np.array([0.5, 0.659])

offree = "lin"
Bd = B.copy()
Cd = np.zeros((2, 2))

x0_p = np.array([0.5, 350, 0.659])
x0_m = np.array([0.5, 350, 0.659, 0.0])
u0 = np.array([300, 0.1])

kal = True
Q_kf = scla.block_diag(1.0e-2 * np.eye(4), np.eye(2))
R_kf = 1.0e-2 * np.eye(2)
P0 = Q_kf.copy()


def defSP(t):
    xsp = np.zeros(4)
    usp = np.array([300.0, 0.1])
    ysp = np.array([0.5, 0.659]) if t < 20 else np.array([0.51, 0.659])
    return [ysp, usp, xsp]


umin = np.array([295, 0.00])
umax = np.array([305, 0.25])
xmin = np.array([0.0, 300, 0.45, -1.0])
xmax = np.array([1.0, 375, 0.75, 1.0])
ymin = np.array([0.0, 0.0])
ymax = np.array([1.0, 1.0])

Qss = np.eye(2)
Rss = np.zeros((2, 2))
Q = np.diag([1.0, 1.0, 1.0, 0.1])
S = 0.10 * np.eye(2)
