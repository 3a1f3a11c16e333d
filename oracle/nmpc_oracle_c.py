"""ORACLE (test infrastructure, never shipped): ctypes access to ``libnmpc_oracle.so`` (nmpc_oracle.c), the C restatement of the
non-linear tracking loop of Ex_NMPC.py - used by tests/ (thousands of closed loops re-run on the host cores) and by bench.py's
``cpu_baseline`` leg only.

The C file writes the CSTR of Ex_NMPC.py out by hand (constants included: in the reference they are locals of the function body);
``OracleNC`` CHECKS those functions against the Ex-file's own Python functions at random points before it computes anything, and
refuses a problem whose structure is not this example's (outputs = states 0 and 2, quadratic costs, EKF, no input-move form).
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int32)
NX, NU, NY, ND, NE, NV = 3, 2, 2, 2, 5, 7


class _NProb(ct.Structure):
    _fields_ = ([(k, ct.c_int32) for k in ("N", "Mx", "max_iter", "has_dsat")] + [("h", ct.c_double)]
                + [("Q", ct.c_double * 9), ("R", ct.c_double * 4), ("Qss", ct.c_double * 4), ("Rss", ct.c_double * 4),
                   ("umin", ct.c_double * NU), ("umax", ct.c_double * NU), ("xmin", ct.c_double * NX), ("xmax", ct.c_double * NX), ("ymin", ct.c_double * NY), ("ymax", ct.c_double * NY),
                   ("tlo", ct.c_double * NV), ("thi", ct.c_double * NV), ("dmin", ct.c_double * ND), ("dmax", ct.c_double * ND),
                   ("Qkf", ct.c_double * (NE * NE)), ("Rkf", ct.c_double * (NY * NY)), ("P0", ct.c_double * (NE * NE)), ("x0m", ct.c_double * NX), ("u0", ct.c_double * NU), ("dhat0", ct.c_double * ND)])


def build(fast=False):
    name = "libnmpc_oracle_fast.so" if fast else "libnmpc_oracle.so"
    lib = os.path.join(HERE, name)
    srcs = [os.path.join(HERE, f) for f in ("nmpc_oracle.c", "orc_dense.h")]
    if fast or not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if fast else []) + [name])
    return lib


class OracleNC:
    def __init__(self, p, fast=False):
        """p: the oracle's own reading of the example (nmpc_oracle.load_problem)"""
        assert (p.nx, p.nu, p.ny, p.nd, p.nxp) == (3, 2, 2, 2, 3) and p.estimator == "ekf" and not p.discrete and p.offree == "nl", "the C restatement is written for the example family of Ex_NMPC.py"
        assert not p.DUForm and not p.DUssForm and p.Dumin is None and not np.any(p.Pf), "input-move forms / terminal weights are not restated in C"
        self.p = p
        self.lib = ct.CDLL(build(fast))
        s = _NProb()
        s.N, s.Mx, s.max_iter, s.has_dsat, s.h = p.N, p.Mx, p.max_iter, int(p.dmin is not None), p.h
        fill = lambda field, v: field.__setitem__(slice(0, len(field)), [float(a) for a in np.ravel(v)])
        for k in ("Q", "R", "Qss", "Rss", "umin", "umax", "xmin", "xmax", "ymin", "ymax", "P0", "u0", "dhat0"):
            fill(getattr(s, k), getattr(p, k))
        fill(s.Qkf, p.Q_kf); fill(s.Rkf, p.R_kf); fill(s.x0m, p.x0_m)
        fill(s.tlo, np.concatenate([p.xmin_ss, p.umin_ss, p.ymin_ss])); fill(s.thi, np.concatenate([p.xmax_ss, p.umax_ss, p.ymax_ss]))
        fill(s.dmin, p.dmin if p.dmin is not None else [-np.inf] * ND); fill(s.dmax, p.dmax if p.dmax is not None else [np.inf] * ND)
        self.s = s
        self._check_functions()

    def _check_functions(self):
        p, rng = self.p, np.random.default_rng(0)
        out = np.zeros(2 * NX)
        for t in (0.0, 4.9, 5.1, 14.0, 20.0, 30.0):
            x = p.x0_m * (1 + 0.05 * rng.standard_normal(3)); u = p.u0 * (1 + 0.02 * rng.standard_normal(2)); d = np.array([0.3, 0.1 + 0.02 * rng.standard_normal()])
            self.lib.norc_functions(x.ctypes.data_as(_dp), u.ctypes.data_as(_dp), d.ctypes.data_as(_dp), ct.c_double(t), out.ctypes.data_as(_dp))
            f = np.asarray(p.funcs["User_fxm_Cont"](x, u, d, t, np.zeros(3)), dtype=float)
            fp = np.asarray(p.funcs["User_fxp_Cont"](x, t, u, np.zeros(3), np.zeros(3)), dtype=float)
            assert np.allclose(out[:3], f, rtol=1e-12, atol=1e-13) and np.allclose(out[3:], fp, rtol=1e-12, atol=1e-13), "the C restatement's functions are not this example's"
            for fn, args in (("User_fym", (x, u, d, t, np.zeros(2))), ("User_fyp", (x, u, t, np.zeros(2), np.zeros(2)))):
                assert np.array_equal(np.asarray(p.funcs[fn](*args), dtype=float), x[[0, 2]]), "outputs are not states 0 and 2"

    def max_threads(self):
        return int(self.lib.norc_max_threads())

    def closed_loop(self, nsteps, x0_p, x0_m=None, max_sqp=1, sqp_tol=1e-9, nthreads=0, logs=True, v_wn=None):
        p = self.p
        x0 = np.ascontiguousarray(np.atleast_2d(x0_p), dtype=np.float64)
        xm = x0 if x0_m is None else np.ascontiguousarray(np.atleast_2d(x0_m), dtype=np.float64)
        B = len(x0)
        sch = p.schedules(nsteps)
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        ysp, usp, pxp, pyp = c(sch["ysp"]), c(sch["usp"]), c(sch["pxp"]), c(sch["pyp"])
        dl = {k: np.zeros((nsteps, B, d)) for k, d in (("U", NU), ("X_HAT", NX), ("XS", NX), ("US", NU), ("Xp", NX), ("D_HAT", ND))} if logs else {}
        il = {k: np.zeros((nsteps, B), dtype=np.int32) for k in ("STATUS_DYN", "STATUS_SS", "SQP_DYN")} if logs else {}
        ptr = lambda a, t=_dp: a.ctypes.data_as(t) if a is not None else None
        rc = self.lib.norc_closed_loop(ct.byref(self.s), B, int(nsteps), ptr(x0), ptr(xm), ptr(ysp), ptr(usp), ptr(pxp), ptr(pyp), int(max_sqp), ct.c_double(sqp_tol),
                                       *[ptr(dl.get(k)) for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT")], *[ptr(il.get(k), _ip) for k in ("STATUS_DYN", "STATUS_SS", "SQP_DYN")], int(nthreads),
                                       ptr(None if v_wn is None else c(np.asarray(v_wn).reshape(nsteps, B, NY))))
        assert rc == 0
        return {**dl, **il}
