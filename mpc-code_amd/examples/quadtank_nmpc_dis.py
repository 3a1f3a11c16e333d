"""Non-linear MPC of a quadruple-tank process with a discrete-time user model: four tank levels fed by two valves whose outputs are
states of their own, output-disturbance model, Luenberger observer, cost and bounds on the input moves, terminal weight.

Written for this project in the Ex-file surface of CPCLAB-UNIPI/MPC-code (names documented in its User_Guide.pdf ch. 3).  Numeric
data are those of the reference's ``Ex_NMPC_dis.py`` (tank constants ``:47-63``, sampled model by five RK4 sub-steps ``:84-101``,
plant disturbance schedule ``:156-164``, observer gain ``:316-326``, set points ``:329-372``, bounds ``:376-388``, weights
``:392-401``, terminal weight ``:404-410``) so that both files define the same problem; tests/test_nmpc.py checks that when the
reference tree is present.  Here the horizon is N = 20 (the reference file ships N = 50).

The model state is ``[valve 1, valve 2, h1, h2, h3, h4]``: the first two states are the valve commands of the previous step
(``x+ = u``), the tanks are integrated over one sampling interval with the inputs held; levels are clamped to [0, 20] inside the
balance equations.  ``offree = "lin"`` with ``Bd = 0, Cd = I``: the two disturbances act on the measured levels.
"""
from casadi import *
import numpy as np

Nsim, N, h = 1000, 20, 5.0

xp = SX.sym("xp", 6)
x = SX.sym("x", 6)
u = SX.sym("u", 2)
y = SX.sym("y", 2)
d = SX.sym("d", 2)

GRAVITY = 981.0
OUTLET = (0.071, 0.057, 0.071, 0.057)      # outlet cross-sections a1..a4 [cm^2]
AREA = (28.0, 32.0, 28.0, 32.0)            # tank cross-sections A1..A4 [cm^2]
SPLIT = (0.7, 0.6)                         # share of each valve's flow that goes to the lower tank
LEVEL_MAX = 20.0
RK_STEPS = 5
FLOW_GAIN = tuple((OUTLET[i] + OUTLET[3 - i]) * (2.0 * GRAVITY * LEVEL_MAX) ** 0.5 / 100.0 for i in (0, 1))      # flow per percent of valve opening


def _clamped(level):
    lv = SX(4, 1)
    for i in range(4):
        lv[i] = if_else(level[i] < 0, 0., if_else(level[i] > LEVEL_MAX, LEVEL_MAX, level[i]))
    return lv


def _tank_balances(level, valve):
    lv = _clamped(level)
    out = [OUTLET[i] * (2.0 * GRAVITY * lv[i]) ** 0.5 for i in range(4)]
    f = SX(4, 1)
    f[0] = (-out[0] + out[2] + SPLIT[0] * FLOW_GAIN[0] * valve[0]) / AREA[0]
    f[1] = (-out[1] + out[3] + SPLIT[1] * FLOW_GAIN[1] * valve[1]) / AREA[1]
    f[2] = (-out[2] + (1.0 - SPLIT[1]) * FLOW_GAIN[1] * valve[1]) / AREA[2]
    f[3] = (-out[3] + (1.0 - SPLIT[0]) * FLOW_GAIN[0] * valve[0]) / AREA[3]
    return f


def _sampled(state, valve):
    nxt = SX(6, 1)
    nxt[0:2] = valve
    level = state[2:6]
    dt = h / RK_STEPS
    for _ in range(RK_STEPS):
        level = _clamped(level)      # the reference clamps the argument of its balance function in place: the sub-step starts from it
        k1 = _tank_balances(level, valve)
        k2 = _tank_balances(level + dt / 2.0 * k1, valve)
        k3 = _tank_balances(level + dt / 2.0 * k2, valve)
        k4 = _tank_balances(level + dt * k3, valve)
        level = level + (dt / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    nxt[2:6] = level
    return nxt


def User_fxp_Dis(x, t, u, pxp, pxmp):
    return _sampled(x, u)


def User_fxm_Dis(x, u, d, t, px):
    return _sampled(x, u)


def User_fyp(x, u, t, pyp, pymp):
    return vertcat(x[2], x[3])


def User_fym(x, u, d, t, px):
    return vertcat(x[2], x[3])


def def_pxp(t):
    if t <= 2250:
        return [np.array([0., 0., 0.5, 0., 0., 0.])]
    if t <= 4000:
        return [np.array([0., 0., 0., 0.5, 0., 0.])]
    return [np.zeros(6)]


offree = "lin"
Bd = np.zeros((6, 2))
Cd = np.eye(2)

x0_p = np.array([39.5794, 38.1492, 11.9996, 12.1883, 1.51364, 1.42194])
x0_m = np.array([39.5794, 38.1492, 11.9996, 12.1883, 1.51364, 1.42194])
u0 = np.array([39.5794, 38.1492])

lue = True
K = np.vstack([np.zeros((6, 2)), np.eye(2)])

_SCHEDULE = (      # (until t, ysp, xsp)
    (50, [11.9996, 12.1883], [50.0, 50.0, 10.0, 10.0, 2.0, 2.0]),
    (1000, [11.9996, 6.0], [60.0, 50.0, 12.0, 8.0, 2.0, 2.0]),
    (2000, [6.0, 6.0], [60.0, 40.0, 12.0, 8.0, 2.0, 2.0]),
    (3000, [12.0, 12.0], [40.0, 40.0, 8.0, 8.0, 2.0, 2.0]),
    (4000, [8.0, 12.0], [40.0, 60.0, 8.0, 12.0, 2.0, 2.0]),
    (5000, [10.0, 10.0], [50.0, 50.0, 10.0, 10.0, 2.0, 2.0]),
)


def defSP(t):
    usp = np.array([39.5185, 38.1743])
    for until, ys, xs_ in _SCHEDULE:
        if t <= until:
            return [np.array(ys), usp, np.array(xs_)]
    return [np.array([8.0, 12.0]), usp, np.array([40.0, 40.0, 8.0, 12.0, 2.0, 2.0])]


umin = np.array([0.0, 0.0])
umax = np.array([100.0, 100.0])
xmin = np.zeros((6, 1))
xmax = np.array([100.0, 100.0, 20.0, 20.0, 20.0, 20.0])
ymin = np.array([0.0, 0.0])
ymax = np.array([20.0, 20.0])
Dumin = np.array([-50.0, -50.0])
Dumax = np.array([50.0, 50.0])

Qss = np.eye(2)
Sss = np.zeros((2, 2))
Q = np.diag([1e3, 1e3, 1.0, 1.0, 1e-6, 1e-6])
S = np.diag([10.0, 10.0])


def User_vfin(x, xs):
    return mtimes(x.T, mtimes(100.0, x))
