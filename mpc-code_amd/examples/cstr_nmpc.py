"""Non-linear offset-free MPC of a CSTR whose feed flow changes stepwise and is estimated as a disturbance.

Written for this project in the Ex-file surface of CPCLAB-UNIPI/MPC-code (names documented in its
User_Guide.pdf ch. 3).  Numeric data are those of the reference's ``Ex_NMPC.py`` (reactor parameters ``:56-67``
and ``:131-142``, feed-flow schedule ``:55``, EKF tuning ``:181-200``, set points ``:217-221``, bounds ``:226-240``,
weights ``:245-250``) so that both files define the same problem; tests/test_nmpc.py checks that when the reference
tree is present.  Model and plant are the same balance equations; in the model the feed flow is the second
disturbance state (``offree = "nl"``), in the plant it follows a piecewise-constant schedule in time.  The horizon
is BASELINE.json configs[3]'s N = 30 (the reference file ships N = 50; ``load_problem(..., overrides={"N": ..})``
changes it).
"""
from casadi import *
import math
import numpy as np
import scipy.linalg as scla

Nsim, N, h = 201, 30, 0.2

xp = SX.sym("xp", 3)     # plant state: concentration [kmol/m^3], temperature [K], level [m]
x = SX.sym("x", 3)
u = SX.sym("u", 2)       # coolant temperature [K], outlet flow [m^3/min]
y = SX.sym("y", 2)
d = SX.sym("d", 2)       # d[1] is the model's feed flow

FEED_TEMP, FEED_CONC = 350, 1.0
TANK_RADIUS = 0.219
K_ARRHENIUS, E_OVER_R = 7.2e10, 8750
HEAT_TRANSFER = 915.6 * 60 / 1000
DENSITY, HEAT_CAPACITY, REACTION_HEAT = 1000.0, 0.239, -5.0e4
Mx = 10


def _balances(conc, temp, level, coolant, outflow, feed_flow):
    area = math.pi * TANK_RADIUS ** 2
    rate_at_feed_temp = K_ARRHENIUS * exp(-E_OVER_R / FEED_TEMP)
    rate = rate_at_feed_temp * exp(-E_OVER_R * (1.0 / temp - 1.0 / FEED_TEMP)) * conc
    d_conc = feed_flow * (FEED_CONC - conc) / (area * level) - rate
    d_temp = (feed_flow * (FEED_TEMP - temp) / (area * level) - REACTION_HEAT / (DENSITY * HEAT_CAPACITY) * rate
              + 2 * HEAT_TRANSFER / (TANK_RADIUS * DENSITY * HEAT_CAPACITY) * (coolant - temp))
    d_level = (feed_flow - outflow) / area
    return vertcat(d_conc, d_temp, d_level)


def User_fxp_Cont(x, t, u, pxp, pxmp):
    feed_flow = if_else(t <= 5, 0.1, if_else(t <= 15, 0.15, if_else(t <= 25, 0.08, 0.1)))
    return _balances(x[0], x[1], x[2], u[0], u[1], feed_flow)


def User_fyp(x, u, t, pyp, pymp):
    return vertcat(x[0], x[2])


def User_fxm_Cont(x, u, d, t, px):
    return _balances(x[0], x[1], x[2], u[0], u[1], d[1])


def User_fym(x, u, d, t, py):
    return vertcat(x[0], x[2])


R_wn = 1e-7 * np.eye(2)
offree = "nl"

x0_p = np.array([0.874317, 325, 0.6528])
x0_m = np.array([0.874317, 325, 0.6528])
u0 = np.array([300.157, 0.1])
dhat0 = np.array([0, 0.1])

ekf = True
Q_kf = scla.block_diag(1.0e-5 * np.eye(3), np.eye(2))
R_kf = 1.0e-4 * np.eye(2)
P0 = np.ones((5, 5))


def defSP(t):
    xsp = np.array([0.0, 0.0, 0.0])
    ysp = np.array([0.874317, 0.6528])
    usp = np.array([300.157, 0.1])
    return [ysp, usp, xsp]


umin = np.array([295, 0.00])
umax = np.array([305, 0.25])
xmin = np.array([0.0, 315, 0.50])
xmax = np.array([1.0, 375, 0.75])
ymin = np.array([0.0, 0.5])
ymax = np.array([1.0, 1.0])
dmin = -100 * np.ones((2, 1))
dmax = 100 * np.ones((2, 1))

Qss = np.diag([10.0, 1.0])
Rss = np.zeros((2, 2))
Q = np.eye(3)
R = 0.1 * np.eye(2)

slacks = False
