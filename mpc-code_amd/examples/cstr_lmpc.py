"""Linear offset-free MPC of a 3-state CSTR - the benchmark problem (BASELINE.json configs[1]).

Written for this project in the Ex-file surface of CPCLAB-UNIPI/MPC-code (names documented in
its User_Guide.pdf ch. 3).  Numeric data are those of the reference's ``Ex_LMPC_CSTR.py``
(model ``:83-85``, disturbance model ``:88-90``, KF tuning ``:101-112``, schedule ``:119-141``,
bounds ``:145-154``, weights ``:157-162``, plant disturbances ``:40-79``) so that both files
define the same problem; tests/test_exfile.py checks that when the reference tree is present.
"""
from casadi import *
import numpy as np
import scipy.linalg as scla

Nsim, N, h = 100, 50, 1

xp = SX.sym("xp", 3)
x = SX.sym("x", 3)
u = SX.sym("u", 2)
y = SX.sym("y", 3)
d = SX.sym("d", 3)

A = np.array([[0.2511, -3.368 * 1e-03, -7.056 * 1e-04],
              [11.06, 0.3296, -2.545],
              [0.0, 0.0, 1.0]])
B = np.array([[-5.426 * 1e-03, 1.53 * 1e-05],
              [1.297, 0.1218],
              [0.0, -6.592 * 1e-02]])
C = np.eye(3)
Ap, Bp, Cp = A.copy(), B.copy(), C.copy()

offree = "lin"
Bd = np.eye(3)
Cd = np.zeros((3, 3))

x0_p = 3.0 * np.ones((3, 1))
x0_m = 3.0 * np.ones((3, 1))
u0 = np.zeros((2, 1))

kal = True
Q_kf = scla.block_diag(1.0e-7 * np.eye(3), np.eye(3))
R_kf = 1.0e-7 * np.eye(3)
P0 = 1.0e-8 * np.eye(6)


def def_pxp(t):
    return [np.array([0.1, 0.0, 0.0]) if t <= 20 else np.zeros(3)]


def def_pyp(t):
    return [np.array([0.1, 0.1, 0.0])]


def defSP(t):
    xsp = np.zeros(3)
    usp = np.zeros(2)
    ysp = np.array([0.2, 0.0, 0.0]) if t <= 15 else np.array([0.0, 0.0, 0.1])
    return [ysp, usp, xsp]


umin = -10.0 * np.ones((2, 1))
umax = 10.0 * np.ones((2, 1))
xmin = np.array([-10.0, -8.0, -10.0])
xmax = 10.0 * np.ones((3, 1))
ymin = np.array([-10.0, -8.0, -10.0])
ymax = 10.0 * np.ones(3)

Qss = np.diag([20.0, 0.0, 1.0])
Rss = np.zeros((2, 2))
Q = np.diag([1.0, 0.0, 1.0])
R = 0.1 * np.eye(2)
