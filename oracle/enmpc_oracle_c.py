"""ORACLE (test infrastructure, never shipped): ctypes access to ``libenmpc_oracle.so`` (enmpc_oracle.c), the C restatement of the economic
loop - used by tests/ (whole batches re-run on the host cores) and by bench.py's ``cpu_baseline`` leg only.

The C file writes the example's functions out by hand; ``OracleEC`` takes their parameters from the Ex-file's namespace and CHECKS the
hand-written functions against the Ex-file's own Python functions at random points before it computes anything - an example that is not
of this family (two-state reactor of Ex_ENMPC.py, profit cost, quadratic terminal and estimator costs) is refused.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

import enmpc_oracle as eo

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int32)
NX, NU, NY, ND, NE, NW, NV = 2, 1, 2, 2, 4, 4, 5


class _EProb(ct.Structure):
    _fields_ = ([(k, ct.c_int32) for k in ("N", "N_mhe", "Mx", "quad", "max_iter", "has_dsat", "mhe_filter")] + [(k, ct.c_double) for k in ("h", "tol", "tol_mhe")]
                + [("par", ct.c_double * 7), ("umin", ct.c_double * NU), ("umax", ct.c_double * NU), ("xmin", ct.c_double * NX), ("xmax", ct.c_double * NX),
                   ("tlo", ct.c_double * NV), ("thi", ct.c_double * NV), ("elo", ct.c_double * NE), ("ehi", ct.c_double * NE), ("dmin", ct.c_double * ND), ("dmax", ct.c_double * ND),
                   ("Bd", ct.c_double * (NX * ND)), ("Cd", ct.c_double * (NY * ND)), ("G", ct.c_double * (NE * NW)), ("P0", ct.c_double * (NE * NE)),
                   ("x0m", ct.c_double * NX), ("u0", ct.c_double * NU), ("est_ekf", ct.c_int32), ("pad_", ct.c_int32), ("Qkf", ct.c_double * (NE * NE)), ("Rkf", ct.c_double * (NY * NY)),
                   ("wlo", ct.c_double * NW), ("whi", ct.c_double * NW)])


def build(fast=False):
    name = "libenmpc_oracle_fast.so" if fast else "libenmpc_oracle.so"
    lib, src = os.path.join(HERE, name), os.path.join(HERE, "enmpc_oracle.c")
    if fast or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if fast else []) + [name])
    return lib


def _params(p):
    """cA0, V, k1, k2, alfa, beta from the Ex-file's constants (either spelling), the terminal weight from User_vfin itself"""
    ns = p.ns
    names = [("cA0", "FEED_CONC"), ("V", "VOLUME"), ("k1", "K1"), ("k2", "K2"), ("alfa", "PRICE_A"), ("beta", "PRICE_B")]
    par = [float(ns[a] if a in ns else ns[b]) for a, b in names]
    par.append(float(np.real(p.vfin(np.array([1.0, 0.0]), np.zeros(2)))) if p.vfin is not None else 0.0)
    return par


class OracleEC:
    def __init__(self, p, fast=False):
        assert (p.nx, p.nu, p.ny, p.nd, p.nxp) == (2, 1, 2, 2, 2) and ((p.mhe and p.n_w == 4) or p.ekf), "the C restatement is written for the example family of Ex_ENMPC.py"
        self.p = p
        self.lib = ct.CDLL(build(fast))
        s = _EProb()
        s.N, s.N_mhe, s.Mx, s.quad, s.max_iter, s.has_dsat = p.N, (p.N_mhe if p.mhe else 2), p.Mx, p.quad_steps, p.max_iter, int(p.dmin is not None)
        s.est_ekf = int(not p.mhe)
        s.mhe_filter = int(getattr(p, "mhe_up", "smooth") == "filter")
        s.h, s.tol, s.tol_mhe = p.h, 1e-8, 1e-10
        fill = lambda field, v: field.__setitem__(slice(0, len(field)), [float(a) for a in np.ravel(v)])
        fill(s.par, _params(p))
        for k in ("umin", "umax", "xmin", "xmax", "Bd", "Cd", "P0"):
            fill(getattr(s, k), getattr(p, k))
        fill(s.G, p.G_mhe if p.mhe else np.eye(NE)); fill(s.x0m, p.x0_m); fill(s.u0, p.u0)
        if not p.mhe:
            fill(s.Qkf, p.Q_kf); fill(s.Rkf, p.R_kf)
        fill(s.tlo, np.concatenate([p.xmin_ss, p.umin_ss, p.ymin_ss])); fill(s.thi, np.concatenate([p.xmax_ss, p.umax_ss, p.ymax_ss]))
        fill(s.wlo, [-np.inf] * NW); fill(s.whi, [np.inf] * NW)
        if p.mhe:
            fill(s.elo, p.xmin_mhe); fill(s.ehi, p.xmax_mhe); fill(s.wlo, p.wmin); fill(s.whi, p.wmax)
        fill(s.dmin, p.dmin if p.dmin is not None else [-np.inf] * ND); fill(s.dmax, p.dmax if p.dmax is not None else [np.inf] * ND)
        self.s = s
        self._check_functions()

    def _check_functions(self):
        """the hand-written C functions against the Ex-file's own, at random points"""
        p, rng = self.p, np.random.default_rng(0)
        out = np.zeros(5)
        for _ in range(8):
            x = rng.uniform(0, 1, 2); u = rng.uniform(0, 2, 1); d = rng.uniform(-0.2, 0.2, 2); xs = rng.uniform(0, 1, 2); wv = rng.standard_normal(6)
            self.lib.eorc_functions(ct.byref(self.s), x.ctypes.data_as(_dp), ct.c_double(u[0]), d.ctypes.data_as(_dp), xs.ctypes.data_as(_dp), wv.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
            f = np.asarray(p.fxm(x, u, d, 0.0, np.zeros(2)), dtype=float)
            fp = np.asarray(p.fxp(x, 0.0, u, np.zeros(2), np.zeros(2)), dtype=float)
            fm = np.asarray(p.fx_mhe(x, u, d, 0.0, np.zeros(2), wv[:4]), dtype=float) if p.mhe else f
            y = x + p.Cd @ d
            ref = [f[0], f[1], float(p.fobj(x, u, y, xs, u, y)), float(np.real(p.vfin(x, xs))) if p.vfin is not None else 0.0, float(p.fobj_mhe(wv[:4], wv[4:], 0.0)) if p.mhe else 0.5 * float(wv @ wv)]
            assert np.allclose(out, ref, rtol=1e-13, atol=1e-13), "the C restatement's functions are not this example's"
            assert np.allclose(fp, f, rtol=1e-13) and np.allclose(fm, f, rtol=1e-13), "plant / estimator model differ from the model: not restated in C"
            assert abs(float(p.fssobj(xs, u, y, None, None, None)) - float(p.fobj(x, u, y, xs, u, y))) < 1e-13, "target cost is not the stage cost rate"

    def max_threads(self):
        return int(self.lib.eorc_max_threads())

    def closed_loop(self, nsteps, x0_p, x_bar=None, nthreads=0, logs=True, v_wn=None, w_wn=None):
        x0 = np.ascontiguousarray(np.atleast_2d(x0_p), dtype=np.float64)
        B = len(x0)
        xb = None if x_bar is None else np.ascontiguousarray(np.broadcast_to(x_bar, (B, NE)), dtype=np.float64)
        dl = {k: np.zeros((nsteps, B, d)) for k, d in (("U", NU), ("XS", NX), ("US", NU), ("X_ES", NE), ("Xp", NX))} if logs else {}
        il = {k: np.zeros((nsteps, B), dtype=np.int32) for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE", "ITERS_DYN", "ITERS_SS", "ITERS_MHE")} if logs else {}
        ptr = lambda a, t=_dp: a.ctypes.data_as(t) if a is not None else None
        rc = self.lib.eorc_closed_loop(ct.byref(self.s), B, int(nsteps), ptr(x0), ptr(xb), *[ptr(dl.get(k)) for k in ("U", "XS", "US", "X_ES", "Xp")],
                                       *[ptr(il.get(k), _ip) for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE", "ITERS_DYN", "ITERS_SS", "ITERS_MHE")], int(nthreads),
                                       ptr(None if v_wn is None else np.ascontiguousarray(np.asarray(v_wn, dtype=np.float64).reshape(nsteps, B, NY))),
                                       ptr(None if w_wn is None else np.ascontiguousarray(np.asarray(w_wn, dtype=np.float64).reshape(nsteps, B, NX))))
        assert rc == 0
        return {**dl, **il}
