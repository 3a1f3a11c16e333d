"""ctypes binding of include/mpc_nmpc.h and the batched non-linear closed loop.

``NmpcSolver`` owns one per-model library (built by :mod:`nlcodegen` from the traced Ex-file functions) and a handle on it;
``run_nmpc_closed_loop`` is the loop body of the reference's ``MPC_code.py:485-827`` with a non-linear model, for B instances
that differ in their initial state - one kernel launch, everything resident in HBM.  There is no CPU path: without a GPU
``NmpcSolver`` raises.
"""
from __future__ import annotations

import ctypes as ct
from typing import Dict, Optional

import numpy as np

from . import nlcodegen
from .capi import MpcAmdError

NMPC_EXPORTS = ("nmpc_create", "nmpc_destroy", "nmpc_last_error", "nmpc_build_info", "nmpc_alloc", "nmpc_set_state", "nmpc_set_schedule",
                "nmpc_run", "nmpc_sync", "nmpc_get_log", "nmpc_last_kernel_ms", "nmpc_set_kernel", "nmpc_set_groups", "nmpc_get_kernel", "nmpc_time_kernels",
                "nmpc_wave_kernel_ms", "nmpc_ekf_update", "nmpc_target_solve", "nmpc_ocp_solve", "nmpc_plant_step", "nmpc_set_noise")

_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int32)


class _NDesc(ct.Structure):
    _fields_ = ([(k, ct.c_int32) for k in ("nx", "nu", "ny", "nd", "nxp", "N", "max_iter", "device")] + [("h", ct.c_double)]
                + [(k, _dp) for k in ("Q", "R", "Qss", "Rss", "umin", "umax", "xmin", "xmax", "ymin", "ymax", "umin_ss", "umax_ss", "xmin_ss",
                                      "xmax_ss", "ymin_ss", "ymax_ss", "dmin", "dmax", "Q_kf", "R_kf")] + [("ycols", _ip)]
                + [(k, _dp) for k in ("Pf", "Cd", "K")] + [(k, ct.c_int32) for k in ("estimator", "du_form", "duss_form")] + [(k, _dp) for k in ("Dumin", "Dumax")])


_libs: Dict[str, ct.CDLL] = {}


def load_nmpc_library(path: str) -> ct.CDLL:
    if path in _libs:
        return _libs[path]
    lib = ct.CDLL(path)
    vp = ct.c_void_p
    lib.nmpc_create.argtypes = [ct.POINTER(_NDesc), ct.POINTER(vp)]; lib.nmpc_create.restype = ct.c_int
    lib.nmpc_destroy.argtypes = [vp]; lib.nmpc_destroy.restype = None
    lib.nmpc_last_error.restype = ct.c_char_p
    lib.nmpc_build_info.restype = ct.c_char_p
    lib.nmpc_alloc.argtypes = [vp, ct.c_int32, ct.c_int32]
    lib.nmpc_set_state.argtypes = [vp] + [_dp] * 7
    lib.nmpc_set_schedule.argtypes = [vp, ct.c_int32, _dp, _dp, _dp, _dp]
    lib.nmpc_set_noise.argtypes = [vp, ct.c_int32, _dp]
    lib.nmpc_run.argtypes = [vp, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_double]
    lib.nmpc_sync.argtypes = [vp]
    lib.nmpc_set_kernel.argtypes = [vp, ct.c_int32]
    lib.nmpc_set_groups.argtypes = [vp, ct.c_int32]
    lib.nmpc_get_kernel.argtypes = [vp]
    lib.nmpc_get_log.argtypes = [vp, ct.c_char_p, vp]
    lib.nmpc_last_kernel_ms.argtypes = [vp]; lib.nmpc_last_kernel_ms.restype = ct.c_float
    lib.nmpc_time_kernels.argtypes = [vp, ct.c_int32]
    lib.nmpc_wave_kernel_ms.argtypes = [vp, ct.POINTER(ct.c_float), _ip]
    lib.nmpc_ekf_update.argtypes = [vp] + [_dp] * 5
    lib.nmpc_target_solve.argtypes = [vp] + [_dp] * 5 + [_ip] * 2
    lib.nmpc_ocp_solve.argtypes = [vp] + [_dp] * 5 + [ct.c_int32, ct.c_double] + [_dp] * 2 + [_ip] * 3
    lib.nmpc_plant_step.argtypes = [vp] + [_dp] * 3
    _libs[path] = lib
    return lib


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _rows(v, B, d):
    a = np.asarray(v, dtype=np.float64)
    return np.ascontiguousarray(np.broadcast_to(a.reshape(-1, d) if a.ndim > 1 else a, (B, d)))


class NmpcSolver:
    """A non-linear problem (:class:`NonlinearMPCProblem`) resident on one GPU."""

    LOGS = {"U": "nu", "X_HAT": "nx", "XS": "nx", "US": "nu", "Xp": "nxp", "D_HAT": "nd"}
    ILOGS = ("STATUS_DYN", "STATUS_SS", "ITERS_DYN", "SQP_DYN", "SQP_SS")

    def __init__(self, problem, device: int = 0, lib_path: Optional[str] = None):
        self.p = p = problem
        self.lib = load_nmpc_library(lib_path or nlcodegen.build_nmpc_library(p))
        self._keep = {}
        d = _NDesc()
        d.nx, d.nu, d.ny, d.nd, d.nxp, d.N, d.max_iter, d.device, d.h = p.nx, p.nu, p.ny, p.nd, p.nxp, p.N, int(p.max_iter), int(device), float(p.h)
        d.estimator, d.du_form, d.duss_form = int(p.estimator == "lue"), int(bool(p.DUForm)), int(bool(p.DUssForm))
        for k in ("Q", "R", "Qss", "Rss", "umin", "umax", "xmin", "xmax", "ymin", "ymax", "umin_ss", "umax_ss", "xmin_ss", "xmax_ss", "ymin_ss",
                  "ymax_ss", "dmin", "dmax", "Q_kf", "R_kf", "Pf", "Cd", "K", "Dumin", "Dumax"):
            v = getattr(p, k, None)
            if v is None:
                setattr(d, k, None)
            else:
                a = _c(v)
                self._keep[k] = a
                setattr(d, k, a.ctypes.data_as(_dp))
        yc = _c(p.ycols, np.int32); self._keep["ycols"] = yc; d.ycols = yc.ctypes.data_as(_ip)
        self.h = ct.c_void_p()
        rc = self.lib.nmpc_create(ct.byref(d), ct.byref(self.h))
        if rc != 0:
            self.h = None
            raise MpcAmdError(f"nmpc_create failed ({rc}): {self.lib.nmpc_last_error().decode()}")
        self.B = self.steps = 0

    def _chk(self, rc, what):
        if rc != 0:
            raise MpcAmdError(f"{what} failed ({rc}): {self.lib.nmpc_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.nmpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_info(self) -> str:
        return self.lib.nmpc_build_info().decode()

    def alloc(self, B: int, max_steps: int):
        self._chk(self.lib.nmpc_alloc(self.h, int(B), int(max_steps)), "nmpc_alloc")
        self.B, self.steps = int(B), 0

    def set_state(self, x0_p, x0_m, dhat=None, P=None, u=None, xs=None, us=None):
        """Defaults are the reference's start-up values: dhat0, P0, u0, and xs = x0_m, us = u0 (MPC_code.py:446-476)."""
        p, B = self.p, self.B
        ne = p.nx + p.nd
        a = [_rows(x0_p, B, p.nxp), _rows(x0_m, B, p.nx), _rows(p.dhat0 if dhat is None else dhat, B, max(p.nd, 1))[:, :max(p.nd, 1)],
             np.ascontiguousarray(np.broadcast_to(np.asarray(p.P0 if P is None else P, dtype=np.float64).reshape(-1, ne * ne), (B, ne * ne))),
             _rows(p.u0 if u is None else u, B, p.nu)]
        a.append(a[1].copy() if xs is None else _rows(xs, B, p.nx))
        a.append(a[4].copy() if us is None else _rows(us, B, p.nu))
        self._chk(self.lib.nmpc_set_state(self.h, *[v.ctypes.data_as(_dp) for v in a]), "nmpc_set_state")

    def set_schedule(self, sched: Dict[str, np.ndarray]):
        ysp, usp = _c(sched["ysp"]), _c(sched["usp"])
        pxp = _c(sched["pxp"]) if sched.get("pxp") is not None else None; pyp = _c(sched["pyp"]) if sched.get("pyp") is not None else None
        self._chk(self.lib.nmpc_set_schedule(self.h, ysp.shape[0], ysp.ctypes.data_as(_dp), usp.ctypes.data_as(_dp),
                                             None if pxp is None else pxp.ctypes.data_as(_dp), None if pyp is None else pyp.ctypes.data_as(_dp)), "nmpc_set_schedule")
        self.steps = ysp.shape[0]

    def set_noise(self, v: Optional[np.ndarray]):
        """white noise on the measurements of the resident loop, v [nsteps, B, ny] (MPC_code.py:537-541), or None"""
        if v is None:
            self._chk(self.lib.nmpc_set_noise(self.h, 0, None), "nmpc_set_noise")
            return
        v = _c(np.asarray(v, dtype=np.float64))
        if v.ndim != 3 or v.shape[1:] != (self.B, self.p.ny):
            raise ValueError(f"noise: [nsteps, {self.B}, {self.p.ny}] expected, got {v.shape}")
        self._chk(self.lib.nmpc_set_noise(self.h, v.shape[0], v.ctypes.data_as(_dp)), "nmpc_set_noise")

    # ---- per-call seam: the reference's three solver calls of a step (include/mpc_nmpc.h) --------------------------------------------------
    def ekf_update(self, y, u_prev, xhat, dhat, P):
        """defEstimator(..., 'ekf' | 'lue') for the batch (MPC_code.py:577-650); returns the posterior ``xhat, dhat, P``"""
        p, B = self.p, self.B
        ne = p.nx + p.nd
        y, u = _rows(y, B, p.ny), _rows(u_prev, B, p.nu)
        xh, dh = _rows(xhat, B, p.nx).copy(), _rows(dhat, B, max(p.nd, 1)).copy()
        Pk = np.ascontiguousarray(np.broadcast_to(np.asarray(P, dtype=np.float64).reshape(-1, ne * ne), (B, ne * ne))).copy()
        self._chk(self.lib.nmpc_ekf_update(self.h, *[v.ctypes.data_as(_dp) for v in (y, u, xh, dh, Pk)]), "nmpc_ekf_update")
        return xh, dh, Pk

    def target_solve(self, dhat, ysp, usp, xs, us):
        """solver_ss(...) for the batch (MPC_code.py:693-718): the targets of the step before in, this step's out; returns ``xs, us, status, sqp``"""
        p, B = self.p, self.B
        dh, xs, us = _rows(dhat, B, max(p.nd, 1)), _rows(xs, B, p.nx).copy(), _rows(us, B, p.nu).copy()
        ysp, usp = _c(ysp).reshape(p.ny), _c(usp).reshape(p.nu)
        st, sq = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        self._chk(self.lib.nmpc_target_solve(self.h, dh.ctypes.data_as(_dp), ysp.ctypes.data_as(_dp), usp.ctypes.data_as(_dp), xs.ctypes.data_as(_dp), us.ctypes.data_as(_dp),
                                             st.ctypes.data_as(_ip), sq.ctypes.data_as(_ip)), "nmpc_target_solve")
        return xs, us, st, sq

    def ocp_solve(self, xhat, dhat, xs, us, u_prev, max_sqp: int = 1, sqp_tol: float = 1e-9):
        """solver(...) for the batch (MPC_code.py:733-805); returns ``u, xhat_next, status, iters, sqp``"""
        p, B = self.p, self.B
        a = [_rows(xhat, B, p.nx), _rows(dhat, B, max(p.nd, 1)), _rows(xs, B, p.nx), _rows(us, B, p.nu), _rows(u_prev, B, p.nu)]
        u, xn = np.empty((B, p.nu)), np.empty((B, p.nx))
        st, it, sq = (np.empty(B, dtype=np.int32) for _ in range(3))
        self._chk(self.lib.nmpc_ocp_solve(self.h, *[v.ctypes.data_as(_dp) for v in a], int(max_sqp), float(sqp_tol), u.ctypes.data_as(_dp), xn.ctypes.data_as(_dp),
                                          st.ctypes.data_as(_ip), it.ctypes.data_as(_ip), sq.ctypes.data_as(_ip)), "nmpc_ocp_solve")
        return u, xn, st, it, sq

    def plant_step(self, u, x_p, pxp=None):
        """Fx_p for the batch on the device (Utilities.py:21-100) + the plant's disturbance of this step; returns the next plant state"""
        p, B = self.p, self.B
        uu, x = _rows(u, B, p.nu), _rows(x_p, B, p.nxp).copy()
        px = None if pxp is None else _c(pxp).reshape(p.nxp)
        self._chk(self.lib.nmpc_plant_step(self.h, uu.ctypes.data_as(_dp), x.ctypes.data_as(_dp), None if px is None else px.ctypes.data_as(_dp)), "nmpc_plant_step")
        return x

    def run(self, k0: int, nsteps: int, max_sqp: int = 1, sqp_tol: float = 1e-9):
        self._chk(self.lib.nmpc_run(self.h, int(k0), int(nsteps), int(max_sqp), float(sqp_tol)), "nmpc_run")

    def sync(self):
        self._chk(self.lib.nmpc_sync(self.h), "nmpc_sync")

    def set_kernel(self, kernel: int):
        """0 auto, 1 one instance per lane, 3 wave-autonomous, 4 split pipeline (the last two: model state <= 4, nu <= 2, N <= 64)."""
        self._chk(self.lib.nmpc_set_kernel(self.h, int(kernel)), "nmpc_set_kernel")

    def set_groups(self, groups: int):
        """split pipeline: groups of the batch on HIP streams of their own (0: by batch size, 1..3)"""
        self._chk(self.lib.nmpc_set_groups(self.h, int(groups)), "nmpc_set_groups")

    def get_kernel(self) -> int:
        return int(self.lib.nmpc_get_kernel(self.h))

    def last_kernel_ms(self) -> float:
        return float(self.lib.nmpc_last_kernel_ms(self.h))

    def time_kernels(self, on: bool = True) -> None:
        """Split pipeline: bracket every wave-style launch of the following runs with its own HIP events."""
        self._chk(self.lib.nmpc_time_kernels(self.h, 1 if on else 0), "nmpc_time_kernels")

    def wave_kernel_ms(self):
        """(sum of the wave-style launches' durations in ms, their count) of the last run; (0, 0) unless time_kernels is on and the
        split pipeline ran."""
        ms, n = ct.c_float(0.0), ct.c_int32(0)
        self._chk(self.lib.nmpc_wave_kernel_ms(self.h, ct.byref(ms), ct.byref(n)), "nmpc_wave_kernel_ms")
        return float(ms.value), int(n.value)

    def get_log(self, name: str) -> np.ndarray:
        if name in self.LOGS:
            out = np.empty((self.steps, self.B, getattr(self.p, self.LOGS[name])))
        elif name in self.ILOGS:
            out = np.empty((self.steps, self.B), dtype=np.int32)
        else:
            raise KeyError(name)
        if out.size:
            self._chk(self.lib.nmpc_get_log(self.h, name.encode(), out.ctypes.data_as(ct.c_void_p)), "nmpc_get_log")
        return out


def measurement_noise(problem, nsteps: int, B: int, seed: int) -> np.ndarray:
    """The draws ``sqrtm(R_wn) N(0, I)`` of MPC_code.py:537-541 for nsteps x B measurements from ``numpy.random.default_rng(seed)`` (step by step, instance by
    instance: the same numbers in the resident loop and in the loop through the per-call seam)."""
    if getattr(problem, "R_wn", None) is None:
        raise ValueError("noise_seed: the example defines no R_wn")
    ev, evec = np.linalg.eigh(0.5 * (problem.R_wn + problem.R_wn.T))      # the symmetric square root scipy.linalg.sqrtm returns for a covariance
    Rv, rng = (evec * np.sqrt(np.maximum(ev, 0.0))) @ evec.T, np.random.default_rng(seed)
    return np.stack([rng.standard_normal((B, problem.ny)) @ Rv.T for _ in range(nsteps)])


def run_nmpc_stepwise(problem, x0_p=None, x0_m=None, nsteps: Optional[int] = None, solver: Optional[NmpcSolver] = None, max_sqp: int = 1, sqp_tol: float = 1e-9,
                      device: int = 0, plant=None, noise_seed: Optional[int] = None) -> Dict[str, np.ndarray]:
    """The reference's loop body call by call (MPC_code.py:485-827): per step the measurement (the Ex-file's plant output on the host), ``ekf_update``
    (defEstimator), ``target_solve`` (solver_ss), ``ocp_solve`` (solver) and the plant - ``plant(x_p [B, nxp], u [B, nu], t) -> x_p+`` of the caller, or the
    device's.  With the device's plant: :func:`run_nmpc_closed_loop` on the instance-per-lane kernel to the bit (when the plant output is exact on the host).
    ``noise_seed``: the white noise of the example's ``R_wn`` on every measurement, ``y_k += sqrtm(R_wn) N(0, I)`` (MPC_code.py:538-541; unseeded there), one draw per
    step and instance from ``numpy.random.default_rng(noise_seed)``; the draws come back as ``V_WN`` [nsteps, B, ny]."""
    p = problem
    nsteps = p.Nsim if nsteps is None else int(nsteps)
    x_p = (p.x0_p[None] if x0_p is None else np.atleast_2d(np.asarray(x0_p, dtype=np.float64))).copy()
    x0_m = p.x0_m[None] if x0_m is None else np.atleast_2d(x0_m)
    B = x_p.shape[0]
    own = solver is None
    s = NmpcSolver(p, device=device) if own else solver
    try:
        s.alloc(B, 1)
        s.set_state(x_p, x0_m)
        sch = p.schedules(nsteps)
        ne = p.nx + p.nd
        xhat, dhat = _rows(x0_m, B, p.nx).copy(), _rows(p.dhat0, B, max(p.nd, 1)).copy()
        P = np.ascontiguousarray(np.broadcast_to(np.asarray(p.P0, dtype=np.float64).reshape(-1, ne * ne), (B, ne * ne))).copy()
        u, xs, us = _rows(p.u0, B, p.nu).copy(), xhat.copy(), _rows(p.u0, B, p.nu).copy()
        keys = ("U", "X_HAT", "XS", "US", "Xp", "D_HAT", "STATUS_DYN", "STATUS_SS", "ITERS_DYN", "SQP_DYN", "SQP_SS")
        out = {k: [] for k in keys}
        vn = measurement_noise(p, nsteps, B, noise_seed) if noise_seed is not None else None
        for k in range(nsteps):
            t = k * p.h
            out["Xp"].append(x_p.copy()); out["X_HAT"].append(xhat.copy())
            y = p.plant_output(x_p, u, t) + sch["pyp"][k]                    # MPC_code.py:531-534
            if vn is not None:                                               # :537-541
                y = y + vn[k]
            xhat, dhat, P = s.ekf_update(y, u, xhat, dhat, P)
            xs, us, st_s, sq_s = s.target_solve(dhat, sch["ysp"][k], sch["usp"][k], xs, us)
            u, xhat, st_d, it_d, sq_d = s.ocp_solve(xhat, dhat, xs, us, u, max_sqp, sqp_tol)
            x_p = s.plant_step(u, x_p, sch["pxp"][k]) if plant is None else np.asarray(plant(x_p, u, t), dtype=np.float64) + sch["pxp"][k]
            for kk, v in (("U", u), ("XS", xs), ("US", us), ("D_HAT", dhat[:, :p.nd]), ("STATUS_DYN", st_d), ("STATUS_SS", st_s), ("ITERS_DYN", it_d), ("SQP_DYN", sq_d), ("SQP_SS", sq_s)):
                out[kk].append(np.array(v))
        res = {k: np.stack(v) for k, v in out.items() if not (k == "D_HAT" and p.nd == 0)}
        if vn is not None:
            res["V_WN"] = vn
        return res
    finally:
        if own:
            s.close()


def run_nmpc_closed_loop(problem, x0_p=None, x0_m=None, nsteps: Optional[int] = None, solver: Optional[NmpcSolver] = None,
                         max_sqp: int = 1, sqp_tol: float = 1e-9, device: int = 0, noise_seed: Optional[int] = None) -> Dict[str, np.ndarray]:
    """``noise_seed``: white noise of the example's ``R_wn`` on every measurement (:func:`measurement_noise`; returned as ``V_WN``); None: the deterministic loop.
    ``max_sqp = 1``: one real-time iteration per step (one linearisation along the shifted previous trajectory, one QP);
    larger: iterate each OCP to the NLP's KKT point, what the reference's IPOPT call returns (``MPC_code.py:775-783``).
    Result arrays carry the reference's names (``MPC_code.py:877-895``), shaped [nsteps, B, dim]."""
    p = problem
    nsteps = p.Nsim if nsteps is None else int(nsteps)
    x0_p = p.x0_p[None] if x0_p is None else np.atleast_2d(x0_p)
    x0_m = p.x0_m[None] if x0_m is None else np.atleast_2d(x0_m)
    B = x0_p.shape[0]
    own = solver is None
    s = NmpcSolver(p, device=device) if own else solver
    try:
        s.alloc(B, nsteps)
        s.set_state(x0_p, x0_m)
        s.set_schedule(p.schedules(nsteps))
        vn = measurement_noise(p, nsteps, B, noise_seed) if noise_seed is not None else None
        if vn is not None:      # (a caller-owned solver keeps the noise its caller installed)
            s.set_noise(vn)
        s.run(0, nsteps, max_sqp, sqp_tol)
        s.sync()
        out = {k: s.get_log(k) for k in list(s.LOGS) + list(s.ILOGS) if not (k == "D_HAT" and p.nd == 0)}
        if vn is not None:
            out["V_WN"] = vn
        ms = s.last_kernel_ms()
        out["TIME_DYN"] = np.full(nsteps, ms * 1e-3 / nsteps); out["TIME_SS"] = np.zeros(nsteps)
        u_prev = np.concatenate([_rows(p.u0, B, p.nu)[None], out["U"][:-1]]) if nsteps else out["U"]
        t = (np.arange(nsteps) * p.h)[:, None]
        out["Yp"] = p.plant_output(out["Xp"], u_prev, t) + p.schedules(nsteps)["pyp"][:, None, :]      # MPC_code.py:531-534
        d_now = out["D_HAT"] if p.nd else np.zeros((nsteps, B, 0))
        d_prior = np.concatenate([_rows(p.dhat0, B, p.nd)[None], d_now[:-1]]) if p.nd else d_now       # dhat is carried unchanged between steps (:655-668)
        out["Y_HAT"] = p.model_output(out["X_HAT"], u_prev, d_prior, t)                                 # :524
        out["YS"] = p.model_output(out["XS"], out["US"], d_now, t)                                      # :730
    finally:
        if own:
            s.close()
        elif noise_seed is not None:      # (also after an exception)
            s.set_noise(None)
    return out
