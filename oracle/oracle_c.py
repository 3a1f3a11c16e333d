"""ORACLE (test infrastructure, never shipped): ctypes access to ``libmpc_oracle.so`` (mpc_oracle.c).

Build with ``make -C oracle``.  Used by tests/ and by bench.py's ``cpu_baseline`` leg only.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmpc_oracle.so")
_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int32)


class _Problem(ct.Structure):
    _fields_ = [(k, ct.c_int32) for k in ("nx", "nu", "ny", "nd", "nxp", "N", "du_form", "duss_form", "y_bounded",
                                          "estimator", "max_iter")] + \
               [(k, _dp) for k in ("A", "B", "C", "Bd", "Cd", "fx_const", "fy_const", "Ap", "Bp", "Cp",
                                   "Q", "R", "P", "Qss", "Rss", "umin", "umax", "xmin", "xmax", "ymin", "ymax",
                                   "umin_ss", "umax_ss", "xmin_ss", "xmax_ss", "ymin_ss", "ymax_ss",
                                   "dmin", "dmax", "Q_kf", "R_kf", "K")]


def build(force=False):
    src = os.path.join(HERE, "mpc_oracle.c")
    if force or not os.path.exists(LIB) or (os.path.exists(src) and os.path.getmtime(LIB) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", HERE, "-s", "libmpc_oracle.so"])
    return LIB


def build_fast():
    """-O3 -march=native build for the CPU baseline timing; always rebuilt on the host that will run it."""
    subprocess.check_call(["make", "-C", HERE, "-s", "-B", "libmpc_oracle_fast.so"])
    return os.path.join(HERE, "libmpc_oracle_fast.so")


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp)


class OracleC:
    """Holds the C view of a LinearMPCProblem-like object (duck-typed) and calls the C restatement."""

    def __init__(self, p, lib_path=None):
        self.lib = ct.CDLL(lib_path or build())
        self.p = p
        self._keep = {}
        s = _Problem()
        s.nx, s.nu, s.ny, s.nd, s.nxp, s.N = p.nx, p.nu, p.ny, p.nd, p.nxp, p.N
        s.du_form, s.duss_form, s.y_bounded = int(p.DUForm), int(p.DUssForm), int(p.y_bounded)
        s.estimator = {"none": 0, "kal": 1, "kalss": 2}[p.estimator]
        s.max_iter = int(p.max_iter)
        for k in ("A", "B", "C", "Bd", "Cd", "fx_const", "fy_const", "Ap", "Bp", "Cp", "Q", "R", "P", "Qss", "Rss",
                  "umin", "umax", "xmin", "xmax", "ymin", "ymax", "umin_ss", "umax_ss", "xmin_ss", "xmax_ss",
                  "ymin_ss", "ymax_ss", "dmin", "dmax", "Q_kf", "R_kf", "K"):
            v = getattr(p, k)
            if v is None:
                setattr(s, k, None)
            else:
                a = _c(v)
                if a.size == 0:
                    a = np.zeros(1)
                self._keep[k] = a
                setattr(s, k, _ptr(a))
        self.s = s
        self.lib.orc_max_threads.restype = ct.c_int

    def max_threads(self):
        return int(self.lib.orc_max_threads())

    def ocp_solve(self, xhat, xs, us, dhat, u_prev, want_w=False):
        p = self.p
        xhat, xs, us, dhat, u_prev = (_c(np.atleast_2d(a)) for a in (xhat, xs, us, dhat, u_prev))
        B = xhat.shape[0]
        u0 = np.full((B, p.nu), np.nan); x1 = np.full((B, p.nx), np.nan)
        st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); res = np.zeros((B, 3))
        w = np.full((B, p.nw), np.nan) if want_w else None
        rc = self.lib.orc_ocp_solve(ct.byref(self.s), B, _ptr(xhat), _ptr(xs), _ptr(us), _ptr(dhat), _ptr(u_prev),
                                    _ptr(u0), _ptr(x1), st.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _ptr(res),
                                    _ptr(w) if want_w else None)
        if rc != 0:
            raise RuntimeError(f"orc_ocp_solve failed: {rc}")
        return dict(u0=u0, x1=x1, status=st, iters=it, res=res, w=w)

    def target_solve(self, usp, ysp, xsp, dhat, us_prev):
        p = self.p
        dhat = _c(np.atleast_2d(dhat)); B = dhat.shape[0]
        usp = _c(np.broadcast_to(usp, (B, p.nu))); ysp = _c(np.broadcast_to(ysp, (B, p.ny)))
        xsp = _c(np.broadcast_to(xsp, (B, p.nx))); us_prev = _c(np.broadcast_to(us_prev, (B, p.nu)))
        xs = np.zeros((B, p.nx)); us = np.zeros((B, p.nu)); ys = np.zeros((B, p.ny))
        st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
        rc = self.lib.orc_target_solve(ct.byref(self.s), B, _ptr(usp), _ptr(ysp), _ptr(xsp), _ptr(dhat), _ptr(us_prev),
                                       _ptr(xs), _ptr(us), _ptr(ys), st.ctypes.data_as(_ip), it.ctypes.data_as(_ip))
        if rc != 0:
            raise RuntimeError(f"orc_target_solve failed: {rc}")
        return dict(xs=xs, us=us, ys=ys, status=st, iters=it)

    def kf_update(self, y, yhat, xi, P):
        y, yhat = _c(np.atleast_2d(y)), _c(np.atleast_2d(yhat))
        xi = _c(np.atleast_2d(xi)).copy(); B = xi.shape[0]
        P = _c(P).copy() if P is not None else np.zeros(1)
        rc = self.lib.orc_kf_update(ct.byref(self.s), B, _ptr(y), _ptr(yhat), _ptr(xi), _ptr(P))
        if rc != 0:
            raise RuntimeError(f"orc_kf_update failed: {rc}")
        return xi, P

    def closed_loop(self, nsteps, x0_p, x0_m, sched=None, nthreads=0, logs=True, u0=None, dhat0=None, P0=None, warm_start=True):
        p = self.p
        x = _c(np.atleast_2d(x0_p)).copy(); xhat = _c(np.atleast_2d(x0_m)).copy(); B = x.shape[0]
        sched = p.schedules(nsteps) if sched is None else sched
        u = _c(np.broadcast_to(p.u0 if u0 is None else u0, (B, p.nu))).copy()
        dhat = _c(np.broadcast_to(p.dhat0 if dhat0 is None else dhat0, (B, p.nd))).copy()
        ne = p.nx + p.nd
        Pk = _c(np.broadcast_to(p.P0 if P0 is None else P0, (B, ne, ne))).copy() if p.estimator == "kal" else np.zeros(1)
        xs = xhat.copy(); us = u.copy()                       # MPC_code.py:682-684
        L = {}
        if logs:
            for k, d in (("U", p.nu), ("X_HAT", p.nx), ("XS", p.nx), ("US", p.nu), ("YS", p.ny), ("Xp", p.nxp), ("D_HAT", p.nd)):
                L[k] = np.zeros((nsteps, B, d))
            for k in ("STATUS_DYN", "STATUS_SS", "ITERS_DYN", "ITERS_SS"):
                L[k] = np.zeros((nsteps, B), np.int32)
        f = lambda k: _ptr(L[k]) if logs else None
        g = lambda k: L[k].ctypes.data_as(_ip) if logs else None
        sc = {k: _c(v) for k, v in sched.items()}
        rc = self.lib.orc_closed_loop(ct.byref(self.s), B, nsteps, _ptr(x), _ptr(xhat), _ptr(dhat), _ptr(Pk), _ptr(u),
                                      _ptr(xs), _ptr(us), _ptr(sc["ysp"]), _ptr(sc["usp"]), _ptr(sc["xsp"]),
                                      _ptr(sc["pxp"]), _ptr(sc["pyp"]), f("U"), f("X_HAT"), f("XS"), f("US"), f("YS"),
                                      f("Xp"), f("D_HAT"), g("STATUS_DYN"), g("STATUS_SS"), g("ITERS_DYN"), g("ITERS_SS"),
                                      int(nthreads), int(bool(warm_start)))
        if rc != 0:
            raise RuntimeError(f"orc_closed_loop failed: {rc}")
        L.update(final=dict(x=x, xhat=xhat, dhat=dhat, P=Pk, u=u, xs=xs, us=us))
        return L
