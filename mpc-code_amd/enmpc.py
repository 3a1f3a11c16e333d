"""ctypes binding of include/mpc_enmpc.h and the batched economic closed loop.

``EnmpcSolver`` owns one per-model library (built by :mod:`econcodegen` from the traced Ex-file functions) and a handle on it;
``run_enmpc_closed_loop`` is the loop body of the reference's ``MPC_code.py:485-827`` for an economic example with a moving-horizon
estimator, for B instances that differ in their initial plant state - everything resident in HBM.  There is no CPU path: without a
GPU ``EnmpcSolver`` raises.
"""
from __future__ import annotations

import ctypes as ct
from typing import Dict, Optional

import numpy as np

from . import econcodegen
from .capi import MpcAmdError

ENMPC_EXPORTS = ("enmpc_create", "enmpc_destroy", "enmpc_last_error", "enmpc_build_info", "enmpc_alloc", "enmpc_set_state", "enmpc_run",
                 "enmpc_sync", "enmpc_get_log", "enmpc_last_kernel_ms", "enmpc_set_kernel", "enmpc_get_kernel", "enmpc_time_kernels",
                 "enmpc_phase_ms", "enmpc_set_groups", "enmpc_comm_unique_id", "enmpc_comm_init", "enmpc_comm_destroy", "enmpc_comm_rank",
                 "enmpc_comm_allgather", "enmpc_comm_allreduce_max", "enmpc_comm_barrier", "enmpc_allgather_log", "enmpc_mhe_update",
                 "enmpc_target_solve", "enmpc_ocp_solve", "enmpc_plant_step", "enmpc_set_noise")

_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int32)


class _EDesc(ct.Structure):
    _fields_ = ([(k, ct.c_int32) for k in ("nx", "nu", "ny", "nd", "nxp", "nw", "N", "N_mhe", "max_iter", "quad_steps", "device", "mhe_update", "estimator")]
                + [(k, ct.c_double) for k in ("h", "tol", "tol_mhe")]
                + [(k, _dp) for k in ("umin", "umax", "xmin", "xmax", "umin_ss", "umax_ss", "xmin_ss", "xmax_ss", "ymin_ss", "ymax_ss", "xmin_mhe", "xmax_mhe",
                                      "dmin", "dmax", "Bd", "Cd", "G_mhe", "P0", "x0_m", "u0", "Q_kf", "R_kf", "wmin", "wmax")])


_libs: Dict[str, ct.CDLL] = {}


def load_enmpc_library(path: str) -> ct.CDLL:
    if path in _libs:
        return _libs[path]
    lib = ct.CDLL(path)
    vp = ct.c_void_p
    lib.enmpc_create.argtypes = [ct.POINTER(_EDesc), ct.POINTER(vp)]; lib.enmpc_create.restype = ct.c_int
    lib.enmpc_destroy.argtypes = [vp]; lib.enmpc_destroy.restype = None
    lib.enmpc_last_error.restype = ct.c_char_p
    lib.enmpc_build_info.restype = ct.c_char_p
    lib.enmpc_alloc.argtypes = [vp, ct.c_int32, ct.c_int32]
    lib.enmpc_set_state.argtypes = [vp] + [_dp] * 5
    lib.enmpc_run.argtypes = [vp, ct.c_int32, ct.c_int32]
    lib.enmpc_sync.argtypes = [vp]
    lib.enmpc_get_log.argtypes = [vp, ct.c_char_p, vp]
    lib.enmpc_last_kernel_ms.argtypes = [vp]; lib.enmpc_last_kernel_ms.restype = ct.c_float
    lib.enmpc_set_kernel.argtypes = [vp, ct.c_int32]
    lib.enmpc_set_groups.argtypes = [vp, ct.c_int32]
    lib.enmpc_get_kernel.argtypes = [vp]
    lib.enmpc_time_kernels.argtypes = [vp, ct.c_int32]
    lib.enmpc_phase_ms.argtypes = [vp, ct.POINTER(ct.c_float), ct.POINTER(ct.c_int32)]
    lib.enmpc_mhe_update.argtypes = [vp] + [_dp] * 5 + [_ip] * 2
    lib.enmpc_set_noise.argtypes = [vp, ct.c_int32, _dp, _dp]
    lib.enmpc_comm_unique_id.argtypes = [ct.c_char_p]
    lib.enmpc_comm_init.argtypes = [vp, ct.c_int32, ct.c_int32, ct.c_char_p]
    lib.enmpc_comm_destroy.argtypes = [vp]
    lib.enmpc_comm_rank.argtypes = [vp, _ip, _ip]
    lib.enmpc_comm_allgather.argtypes = [vp, vp, ct.c_size_t, vp]
    lib.enmpc_comm_allreduce_max.argtypes = [vp, _dp, ct.c_int32]
    lib.enmpc_comm_barrier.argtypes = [vp]
    lib.enmpc_allgather_log.argtypes = [vp, ct.c_char_p, ct.c_int32, ct.c_int32, _dp]
    lib.enmpc_target_solve.argtypes = [vp] + [_dp] * 3 + [_ip] * 2
    lib.enmpc_ocp_solve.argtypes = [vp] + [_dp] * 6 + [_ip] * 2
    lib.enmpc_plant_step.argtypes = [vp] + [_dp] * 2
    _libs[path] = lib
    return lib


def _c(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _rows(v, B, d):
    a = np.asarray(v, dtype=np.float64)
    return np.ascontiguousarray(np.broadcast_to(a.reshape(-1, d) if a.ndim > 1 else a, (B, d)))


class EnmpcSolver:
    """An economic problem (:class:`EconomicMPCProblem`) resident on one GPU."""

    LOGS = {"U": "nu", "X_HAT": "nx", "XS": "nx", "US": "nu", "Xp": "nxp", "D_HAT": "nd", "X_ES": None}
    ILOGS = ("STATUS_DYN", "STATUS_SS", "STATUS_MHE", "ITERS_DYN", "ITERS_SS", "ITERS_MHE")

    def __init__(self, problem, device: int = 0, lib_path: Optional[str] = None, tol: float = 1e-8, tol_mhe: float = 1e-10):
        self.p = p = problem
        self.lib = load_enmpc_library(lib_path or econcodegen.build_enmpc_library(p))
        self._keep = {}
        d = _EDesc()
        d.nx, d.nu, d.ny, d.nd, d.nxp, d.nw, d.N, d.N_mhe = p.nx, p.nu, p.ny, p.nd, p.nxp, p.n_w, p.N, p.N_mhe
        d.max_iter, d.quad_steps, d.device, d.h, d.tol, d.tol_mhe = int(p.max_iter), int(p.quad_steps), int(device), float(p.h), float(tol), float(tol_mhe)
        d.mhe_update = {"smooth": 0, "filter": 1}[p.mhe_up]
        d.estimator = {"mhe": 0, "ekf": 1}[getattr(p, "estimator", "mhe")]
        for k in ("umin", "umax", "xmin", "xmax", "umin_ss", "umax_ss", "xmin_ss", "xmax_ss", "ymin_ss", "ymax_ss", "xmin_mhe", "xmax_mhe", "dmin", "dmax",
                  "Bd", "Cd", "G_mhe", "P0", "x0_m", "u0", "Q_kf", "R_kf", "wmin", "wmax"):
            v = getattr(p, k, None)
            if v is None:
                setattr(d, k, None)
            else:
                a = _c(v)
                self._keep[k] = a
                setattr(d, k, a.ctypes.data_as(_dp))
        self.h = ct.c_void_p()
        rc = self.lib.enmpc_create(ct.byref(d), ct.byref(self.h))
        if rc != 0:
            self.h = None
            raise MpcAmdError(f"enmpc_create failed ({rc}): {self.lib.enmpc_last_error().decode()}")
        self.B = self.steps = 0

    def _chk(self, rc, what):
        if rc != 0:
            raise MpcAmdError(f"{what} failed ({rc}): {self.lib.enmpc_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.enmpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_info(self) -> str:
        return self.lib.enmpc_build_info().decode()

    def alloc(self, B: int, max_steps: int):
        self._chk(self.lib.enmpc_alloc(self.h, int(B), int(max_steps)), "enmpc_alloc")
        self.B, self.steps = int(B), 0

    def set_state(self, x_p, xhat=None, dhat=None, u=None, x_bar=None):
        p, B = self.p, self.B
        xp = _rows(x_p, B, p.nxp)
        xh = _rows(p.x0_m if xhat is None else xhat, B, p.nx)
        dh = _rows(np.zeros(p.nd) if dhat is None else dhat, B, p.nd)
        uu = _rows(p.u0 if u is None else u, B, p.nu)
        xb = _rows(p.x_bar if x_bar is None else x_bar, B, p.nx + p.nd)
        self._chk(self.lib.enmpc_set_state(self.h, *[a.ctypes.data_as(_dp) for a in (xp, xh, dh, uu, xb)]), "enmpc_set_state")
        self.steps = 0

    def run(self, k0: int, nsteps: int):
        self._chk(self.lib.enmpc_run(self.h, int(k0), int(nsteps)), "enmpc_run")
        self.steps = max(self.steps, k0 + nsteps)

    def sync(self):
        self._chk(self.lib.enmpc_sync(self.h), "enmpc_sync")

    def set_kernel(self, kernel: int):
        """0 auto, 1 one launch for all steps, 2 split pipeline (one launch per phase and step)"""
        self._chk(self.lib.enmpc_set_kernel(self.h, int(kernel)), "enmpc_set_kernel")

    def set_groups(self, groups: int):
        """split pipeline: groups of the batch, each on its own stream (0: by batch size)"""
        self._chk(self.lib.enmpc_set_groups(self.h, int(groups)), "enmpc_set_groups")

    def get_kernel(self) -> int:
        return int(self.lib.enmpc_get_kernel(self.h))

    def time_kernels(self, on: bool = True):
        self._chk(self.lib.enmpc_time_kernels(self.h, int(bool(on))), "enmpc_time_kernels")

    def phase_ms(self):
        """(estimator, target, OCP + plant) device milliseconds of the last run and the launches of each (split pipeline, timing on)"""
        ms, n = (ct.c_float * 3)(), ct.c_int32(0)
        self._chk(self.lib.enmpc_phase_ms(self.h, ms, ct.byref(n)), "enmpc_phase_ms")
        return [float(v) for v in ms], int(n.value)

    def last_kernel_ms(self) -> float:
        return float(self.lib.enmpc_last_kernel_ms(self.h))

    # ---- multi-GPU: the job's communicator (RCCL inside the library; mpc-code_amd/shard.py:RcclComm drives these) -----------------------------
    def comm_unique_id(self) -> bytes:
        buf = ct.create_string_buffer(128)
        self._chk(self.lib.enmpc_comm_unique_id(buf), "enmpc_comm_unique_id")
        return buf.raw

    def comm_init(self, rank: int, world: int, uid: bytes):
        self._chk(self.lib.enmpc_comm_init(self.h, int(rank), int(world), uid), "enmpc_comm_init")

    def comm_rank(self):
        r, w = ct.c_int32(0), ct.c_int32(1)
        self._chk(self.lib.enmpc_comm_rank(self.h, ct.byref(r), ct.byref(w)), "enmpc_comm_rank")
        return int(r.value), int(w.value)

    def comm_allgather(self, array: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(array)
        out = np.empty((self.comm_rank()[1],) + a.shape, dtype=a.dtype)
        self._chk(self.lib.enmpc_comm_allgather(self.h, a.ctypes.data_as(ct.c_void_p), a.nbytes, out.ctypes.data_as(ct.c_void_p)), "enmpc_comm_allgather")
        return out

    def comm_allreduce_max(self, values) -> np.ndarray:
        a = np.ascontiguousarray(np.asarray(values, dtype=np.float64)).copy()
        self._chk(self.lib.enmpc_comm_allreduce_max(self.h, a.ctypes.data_as(_dp), a.size), "enmpc_comm_allreduce_max")
        return a

    def comm_barrier(self):
        self._chk(self.lib.enmpc_comm_barrier(self.h), "enmpc_comm_barrier")

    def allgather_log(self, name: str, k0: int, nsteps: int, to_host: bool = True):
        """steps [k0, k0 + nsteps) of a float64 log of every rank: one RCCL all-gather, device to device; ``to_host``: also return [world, nsteps, B, dim]"""
        d = self.p.nx + self.p.nd if name == "X_ES" else getattr(self.p, self.LOGS[name])
        out = np.empty((self.comm_rank()[1], nsteps, self.B, d)) if to_host else None
        self._chk(self.lib.enmpc_allgather_log(self.h, name.encode(), int(k0), int(nsteps), None if out is None else out.ctypes.data_as(_dp)), "enmpc_allgather_log")
        return out

    # ---- per-call seam: the reference's three solver calls of a step (include/mpc_enmpc.h) -------------------------------------------------
    def set_noise(self, v=None, w=None):
        """white noise of the resident loop: v [nsteps, B, ny] on the measurements, w [nsteps, B, nxp] on the plant state after its step (MPC_code.py:537-541, 822-827); None: none"""
        arrs = [None if a is None else _c(np.asarray(a, dtype=np.float64)) for a in (v, w)]
        for a, d in zip(arrs, (self.p.ny, self.p.nxp)):
            if a is not None and (a.ndim != 3 or a.shape[1:] != (self.B, d)):
                raise ValueError(f"noise: [nsteps, {self.B}, {d}] expected, got {a.shape}")
        n = next((a.shape[0] for a in arrs if a is not None), 0)
        if any(a is not None and a.shape[0] != n for a in arrs):
            raise ValueError("noise: v and w cover the same number of steps")
        self._chk(self.lib.enmpc_set_noise(self.h, n, *[None if a is None else a.ctypes.data_as(_dp) for a in arrs]), "enmpc_set_noise")

    def mhe_update(self, y, u_prev):
        """defEstimator(..., 'mhe') for the batch (MPC_code.py:577-650): measurement ``y [B, ny]``, the input applied over the step before ``[B, nu]``;
        returns ``xhat, dhat, xes, status, iters``"""
        p, B = self.p, self.B
        y, u = _rows(y, B, p.ny), _rows(u_prev, B, p.nu)
        xh, dh, xe = np.empty((B, p.nx)), np.empty((B, p.nd)), np.empty((B, p.nx + p.nd))
        st, it = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        self._chk(self.lib.enmpc_mhe_update(self.h, y.ctypes.data_as(_dp), u.ctypes.data_as(_dp), xh.ctypes.data_as(_dp), dh.ctypes.data_as(_dp), xe.ctypes.data_as(_dp),
                                            st.ctypes.data_as(_ip), it.ctypes.data_as(_ip)), "enmpc_mhe_update")
        return xh, dh, xe, st, it

    def target_solve(self, dhat):
        """solver_ss(...) for the batch (MPC_code.py:693-718); returns ``xs, us, status, iters``"""
        p, B = self.p, self.B
        dh = _rows(dhat, B, p.nd)
        xs, us = np.empty((B, p.nx)), np.empty((B, p.nu))
        st, it = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        self._chk(self.lib.enmpc_target_solve(self.h, dh.ctypes.data_as(_dp), xs.ctypes.data_as(_dp), us.ctypes.data_as(_dp), st.ctypes.data_as(_ip), it.ctypes.data_as(_ip)), "enmpc_target_solve")
        return xs, us, st, it

    def ocp_solve(self, xhat, dhat, xs, us):
        """solver(...) for the batch (MPC_code.py:733-805); returns ``u, xhat_next, status, iters``"""
        p, B = self.p, self.B
        a = [_rows(xhat, B, p.nx), _rows(dhat, B, p.nd), _rows(xs, B, p.nx), _rows(us, B, p.nu)]
        u, xn = np.empty((B, p.nu)), np.empty((B, p.nx))
        st, it = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        self._chk(self.lib.enmpc_ocp_solve(self.h, *[v.ctypes.data_as(_dp) for v in a], u.ctypes.data_as(_dp), xn.ctypes.data_as(_dp), st.ctypes.data_as(_ip), it.ctypes.data_as(_ip)), "enmpc_ocp_solve")
        return u, xn, st, it

    def plant_step(self, u, x_p):
        """Fx_p for the batch on the device (Utilities.py:58-82); returns the next plant state"""
        p, B = self.p, self.B
        uu, x = _rows(u, B, p.nu), _rows(x_p, B, p.nxp).copy()
        self._chk(self.lib.enmpc_plant_step(self.h, uu.ctypes.data_as(_dp), x.ctypes.data_as(_dp)), "enmpc_plant_step")
        return x

    def get_log(self, name: str) -> np.ndarray:
        p = self.p
        if name in self.LOGS:
            d = p.nx + p.nd if name == "X_ES" else getattr(p, self.LOGS[name])
            out = np.empty((self.steps, self.B, d), dtype=np.float64)
        elif name in self.ILOGS:
            out = np.empty((self.steps, self.B), dtype=np.int32)
        else:
            raise KeyError(name)
        self._chk(self.lib.enmpc_get_log(self.h, name.encode(), out.ctypes.data_as(ct.c_void_p)), "enmpc_get_log")
        return out


def loop_noise(problem, nsteps: int, B: int, seed: int):
    """The draws of the reference's white noises for nsteps x B instances from ``numpy.random.default_rng(seed)``: ``V_WN`` = sqrtm(R_wn) N(0, I) on the measurement
    (MPC_code.py:537-541) and ``W_WN`` = G_wn sqrtm(Q_wn) N(0, I) on the plant state after its step (:822-827) - None where the example does not define them.  Per step
    first the measurement's draws, then the state's."""
    def root(S):
        ev, evec = np.linalg.eigh(0.5 * (S + S.T))      # the symmetric square root scipy.linalg.sqrtm returns for a covariance
        return (evec * np.sqrt(np.maximum(ev, 0.0))) @ evec.T
    p, rng = problem, np.random.default_rng(seed)
    if getattr(p, "R_wn", None) is None and getattr(p, "G_wn", None) is None:
        raise ValueError("noise_seed: the example defines neither R_wn nor G_wn / Q_wn")
    Rv = None if p.R_wn is None else root(p.R_wn)
    Gw = None if p.G_wn is None else p.G_wn @ root(p.Q_wn)
    V, W = [], []
    for _ in range(nsteps):
        if Rv is not None:
            V.append(rng.standard_normal((B, p.ny)) @ Rv.T)
        if Gw is not None:
            W.append(rng.standard_normal((B, p.nxp)) @ Gw.T)
    return (np.stack(V) if V else None), (np.stack(W) if W else None)


def run_enmpc_stepwise(problem, x0_p, nsteps: Optional[int] = None, device: int = 0, solver: Optional[EnmpcSolver] = None, plant=None, noise_seed: Optional[int] = None):
    """The reference's loop body call by call (MPC_code.py:485-827): per step the measurement, ``mhe_update`` (defEstimator), ``target_solve`` (solver_ss),
    ``ocp_solve`` (solver) and the plant - ``plant(x_p [B, nxp], u [B, nu]) -> x_p+`` of the caller, or the device's.  Same result arrays as
    :func:`run_enmpc_closed_loop`; with the device's plant, the same numbers to the bit.  ``noise_seed``: the example's white noises on the measurement and on the plant
    state (:func:`loop_noise`; returned as ``V_WN`` / ``W_WN``) - the measurement and the plant are the caller's side of the seam."""
    p = problem
    nsteps = p.Nsim if nsteps is None else int(nsteps)
    x_p = np.atleast_2d(np.asarray(x0_p, dtype=np.float64)).copy()
    B = len(x_p)
    s = solver or EnmpcSolver(p, device=device)
    try:
        s.alloc(B, 1)
        s.set_state(x_p)
        u = _rows(p.u0, B, p.nu)
        xhat = _rows(p.x0_m, B, p.nx)
        out = {k: [] for k in ("U", "X_HAT", "XS", "US", "Xp", "D_HAT", "X_ES", "STATUS_DYN", "STATUS_SS", "STATUS_MHE", "ITERS_DYN", "ITERS_SS", "ITERS_MHE")}
        vn, wn = loop_noise(p, nsteps, B, noise_seed) if noise_seed is not None else (None, None)
        for k_ in range(nsteps):
            out["Xp"].append(x_p.copy()); out["X_HAT"].append(xhat.copy())
            y = x_p                                        # Fy_p with StateFeedback (Utilities.py:84-86)
            if vn is not None:
                y = y + vn[k_]                             # MPC_code.py:537-541
            xhat, dhat, xes, st_m, it_m = s.mhe_update(y, u)
            xs, us, st_s, it_s = s.target_solve(dhat)
            u, xhat, st_d, it_d = s.ocp_solve(xhat, dhat, xs, us)
            x_p = s.plant_step(u, x_p) if plant is None else np.asarray(plant(x_p, u), dtype=np.float64)
            if wn is not None:
                x_p = x_p + wn[k_]                         # :822-827
            for k, v in (("U", u), ("XS", xs), ("US", us), ("D_HAT", dhat), ("X_ES", xes), ("STATUS_DYN", st_d), ("STATUS_SS", st_s), ("STATUS_MHE", st_m),
                         ("ITERS_DYN", it_d), ("ITERS_SS", it_s), ("ITERS_MHE", it_m)):
                out[k].append(np.array(v))
        res = {k: np.stack(v) for k, v in out.items()}
        if vn is not None:
            res["V_WN"] = vn
        if wn is not None:
            res["W_WN"] = wn
        return res
    finally:
        if solver is None:
            s.close()


def run_enmpc_closed_loop(problem, x0_p, nsteps: Optional[int] = None, device: int = 0, steps_per_launch: int = 0, solver: Optional[EnmpcSolver] = None,
                          kernel: Optional[int] = None, groups: Optional[int] = None, noise_seed: Optional[int] = None):
    """The closed loop of the reference for B instances (rows of ``x0_p``); model state, input and the estimator's prior start from the
    Ex-file's ``x0_m``, ``u0``, ``x_bar``.  Returns the reference's result arrays ``[nsteps, B, dim]`` plus status / iteration words.
    ``noise_seed``: the example's white noises on measurement and plant state (:func:`loop_noise`; on the device: ``enmpc_set_noise``), returned as ``V_WN`` / ``W_WN``."""
    p = problem
    nsteps = p.Nsim if nsteps is None else int(nsteps)
    x0_p = np.atleast_2d(np.asarray(x0_p, dtype=np.float64))
    s = solver or EnmpcSolver(p, device=device)
    try:
        s.alloc(len(x0_p), nsteps)
        if kernel is not None:
            s.set_kernel(kernel)
        if groups is not None:
            s.set_groups(groups)
        s.set_state(x0_p)
        vn, wn = loop_noise(p, nsteps, len(x0_p), noise_seed) if noise_seed is not None else (None, None)
        if noise_seed is not None:      # (a caller-owned solver keeps the noise its caller installed)
            s.set_noise(vn, wn)
        spl = steps_per_launch if steps_per_launch > 0 else nsteps
        for k0 in range(0, nsteps, spl):
            s.run(k0, min(spl, nsteps - k0))
        s.sync()
        out = {k: s.get_log(k) for k in list(s.LOGS) + list(s.ILOGS)}
        out["kernel_ms"] = s.last_kernel_ms()
        # the reference's remaining result arrays (MPC_code.py:877-895), from the logs: with StateFeedback the measurement is the plant state
        # (Utilities.py:84-86), yhat_k = Fy_model(xhat_k, dhat_k) with the disturbance estimate of the step before (:524), ys_k = Fy_model(xs_k, dhat_k) (:730)
        B = len(x0_p)
        d_prior = np.zeros((nsteps, B, p.nd)); d_prior[1:] = out["D_HAT"][:-1]
        out["Yp"] = out["Xp"].copy() if vn is None else out["Xp"] + vn      # (the measurement: StateFeedback, + its white noise)
        if vn is not None:
            out["V_WN"] = vn
        if wn is not None:
            out["W_WN"] = wn
        out["Y_HAT"] = out["X_HAT"] + d_prior @ p.Cd.T
        out["YS"] = out["XS"] + out["D_HAT"] @ p.Cd.T
        out["TIME_DYN"] = np.full(nsteps, out["kernel_ms"] * 1e-3 / nsteps); out["TIME_SS"] = np.zeros(nsteps)      # (one device time for the whole step)
        return out
    finally:
        if solver is None:
            s.close()
        elif noise_seed is not None:      # (also after an exception: a caller-owned solver does not keep this run's draws)
            s.set_noise(None, None)
