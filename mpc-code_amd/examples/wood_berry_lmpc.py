"""Wood-Berry distillation column, 2x2, Delta-u cost, plant/model mismatch (BASELINE.json configs[0]).

Ex-file surface of CPCLAB-UNIPI/MPC-code; numeric data as in the reference's ``Ex_LMPC_WB.py``
(plant ``:34-36``, model ``:40-44``, output-disturbance model ``:47-49``, Luenberger gain ``:66-70``,
set-point step ``:77-99``, input bounds ``:103-104``, weights ``:115-121``).
"""
from casadi import *
import numpy as np

Nsim, N, h = 100, 50, 1

xp = SX.sym("xp", 4)
x = SX.sym("x", 4)
u = SX.sym("u", 2)
y = SX.sym("y", 2)
d = SX.sym("d", 2)

_poles = np.array([0.8871, 0.8324, 0.9092, 0.8703])
Ap = np.diag(_poles)
Bp = np.array([[1.0, 0.0], [1.0, 0.0], [0.0, 1.0], [0.0, 2.0]])
Cp = np.array([[1.4447, 0.0, -1.7169, 0.0], [0.0, 1.1064, 0.0, -1.2579]])

A = np.diag(_poles) + 2 * np.diag([0.01, -0.01, -0.01, 0.01])
B = Bp.copy()
C = Cp.copy()

offree = "lin"
Bd = np.zeros((4, 2))
Cd = np.eye(2)

x0_p = np.zeros((4, 1))
x0_m = np.zeros((4, 1))
u0 = np.zeros((2, 1))

lue = True
K = np.vstack((np.zeros((4, 2)), np.eye(2)))


def defSP(t):
    xsp = np.zeros(4)
    usp = np.zeros(2)
    ysp = np.zeros(2) if t <= 10 else np.array([1.0, -1.0])
    return [ysp, usp, xsp]


umin = -0.5 * np.ones((2, 1))
umax = 0.5 * np.ones((2, 1))

Qss = np.diag([1, 1])
Rss = np.zeros((2, 2))
Q = C.T @ np.diag([1, 1]) @ C
S = np.diag([10, 20])
