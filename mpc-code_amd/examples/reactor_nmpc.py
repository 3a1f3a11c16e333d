"""Non-linear offset-free tracking MPC of an isothermal reactor with the consecutive reactions A -> B -> C.

Written for this project in the Ex-file surface of CPCLAB-UNIPI/MPC-code (names documented in its User_Guide.pdf ch. 3).
The balance equations and their parameters are those of the reactor in the reference's ``Ex_ENMPC.py`` (``:42-49``, ``:64-65``,
sampling time and horizon ``:18-22``, bounds ``:184-187``); the reference optimises an economic cost there (out of this project's
scope so far), this file tracks a set point of the product concentration instead: quadratic stage cost, extended Kalman filter with two
disturbance states inside the model (``offree = "nl"``: a correction of the first rate constant and one of the dilution rate).
Two states and ONE input: the smallest stage the non-linear kernels meet (the other examples have two inputs), with a plant whose
first reaction is faster than the model believes, so that the disturbance estimate has something to do.
"""
from casadi import *
import numpy as np
import scipy.linalg as scla

Nsim, N, h = 60, 25, 2.0

xp = SX.sym("xp", 2)     # plant state: concentrations of A and B [kmol/m^3]
x = SX.sym("x", 2)
u = SX.sym("u", 1)       # dilution rate F / V [1/min]
y = SX.sym("y", 2)
d = SX.sym("d", 2)

FEED_CONC = 1.0
K1, K2 = 1.0, 0.05
K1_PLANT = 1.1           # the plant's first reaction is 10 % faster than the model believes
Mx = 10


def _balances(c_a, c_b, dilution, k1):
    return vertcat(dilution * (FEED_CONC - c_a) - k1 * c_a, -dilution * c_b + k1 * c_a - K2 * c_b)


def User_fxp_Cont(x, t, u, pxp, pxmp):
    return _balances(x[0], x[1], u[0], K1_PLANT)


def User_fyp(x, u, t, pyp, pymp):
    return vertcat(x[0], x[1])


def User_fxm_Cont(x, u, d, t, px):
    return _balances(x[0], x[1], u[0] + d[1], K1 + d[0])


def User_fym(x, u, d, t, py):
    return vertcat(x[0], x[1])


offree = "nl"

x0_p = np.array([0.45, 0.50])
x0_m = np.array([0.45, 0.50])
u0 = np.array([0.8])
dhat0 = np.array([0.0, 0.0])

ekf = True
Q_kf = scla.block_diag(1.0e-6 * np.eye(2), 1.0e-2 * np.eye(2))
R_kf = 1.0e-4 * np.eye(2)
P0 = 1.0e-2 * np.eye(4)


def defSP(t):
    xsp = np.array([0.0, 0.0])
    ysp = np.array([0.0, 0.56]) if t <= 40 else np.array([0.0, 0.48])     # product concentration; the first output carries no weight
    usp = np.array([0.0])
    return [ysp, usp, xsp]


umin = np.array([0.05])
umax = np.array([2.0])
xmin = np.array([0.0, 0.0])
xmax = np.array([1.0, 1.0])

dmin = -0.5 * np.ones((2, 1))
dmax = 0.5 * np.ones((2, 1))

Qss = np.diag([0.0, 10.0])
Rss = np.zeros((1, 1))
Q = np.diag([0.1, 1.0])
R = 0.05 * np.eye(1)

slacks = False
