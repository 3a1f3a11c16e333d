"""ctypes binding of ``include/mpc_amd.h`` - the only way Python reaches the HIP solver.

``Solver(problem)`` is the counterpart of the reference's solver construction
(``opt_ss`` + ``opt_dyn`` + estimator set-up, reference ``MPC_code.py:300,331-363``); its methods are
the per-step calls of the loop (``solver_ss(...)`` ``:704-709``, ``solver(...)`` ``:776-781``,
``defEstimator(...)`` ``:577-650``) over a batch of instances.

There is no fallback: if ``libmpc_amd.so`` is missing or no HIP device is usable, construction
raises :class:`MpcAmdError`.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess
from typing import Optional

import numpy as np

from . import PKG_DIR

CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("MPC_AMD_LIB") or os.path.join(CSRC, "libmpc_amd.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]

STATUS_SOLVED, STATUS_MAXITER, STATUS_INFEASIBLE = 0, 1, 2
LOG_NONE, LOG_U, LOG_ALL = 0, 1, 2
_EST = {"none": 0, "kal": 1, "kalss": 2}

_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int32)


class MpcAmdError(RuntimeError):
    pass


class _Desc(ct.Structure):
    _fields_ = [(k, ct.c_int32) for k in ("nx", "nu", "ny", "nd", "nxp", "N", "du_form", "duss_form", "y_bounded",
                                          "estimator", "max_iter", "device")] + \
               [(k, _dp) for k in ("A", "B", "C", "Bd", "Cd", "fx_const", "fy_const", "Ap", "Bp", "Cp",
                                   "Q", "R", "P", "Qss", "Rss", "umin", "umax", "xmin", "xmax", "ymin", "ymax",
                                   "umin_ss", "umax_ss", "xmin_ss", "xmax_ss", "ymin_ss", "ymax_ss",
                                   "dmin", "dmax", "Q_kf", "R_kf", "K", "Dumin", "Dumax")] + [("term_cons", ct.c_int32), ("nl_plant", ct.c_int32), ("h_sample", ct.c_double),
                                                                                             ("slacks", ct.c_int32), ("Ws", _dp),
                                                                                             ("n_user_rows", ct.c_int32), ("Gx", _dp), ("Gu", _dp), ("Gd", _dp), ("g0", _dp)]


def jit_library_path(dims) -> str:
    return os.path.join(CSRC, "jit", "libmpc_amd_" + "_".join(str(int(v)) for v in dims) + ".so")


def plant_library_path(dims, header_text: str) -> str:
    import hashlib
    return os.path.join(CSRC, "jit", "libmpc_amd_" + "_".join(str(int(v)) for v in dims) + "_plant_" + hashlib.sha256(header_text.encode()).hexdigest()[:12] + ".so")


def build_library(force: bool = False, verbose: bool = False, dims=None, plant_header: Optional[str] = None) -> str:
    """Compile ``csrc/mpc_amd.hip`` for gfx950 in-tree (hipcc cross-compiles without a GPU).

    ``dims = (nx, nu, ny, nd, nxp, du_form, general_output_rows)``: a library holding the kernels of exactly that
    dimension set (every kernel is a template on the problem dimensions; the default library carries the sets of the
    shipped examples, ``mpc_build_info()``), written to ``csrc/jit/`` and reused while the sources are unchanged."""
    srcs = [os.path.join(CSRC, f) for f in ("mpc_amd.hip", "mpc_device.hpp", "mpc_sym.hpp", "mpc_tp.hpp", "mpc_wave.hpp")] + \
           [os.path.join(os.path.dirname(PKG_DIR), "include", "mpc_amd.h")]
    out = LIB_PATH if dims is None else jit_library_path(dims)
    hdr = None
    if plant_header is not None:       # the fused closed loop with the Ex-file's own plant function: one library per (dimension set, plant)
        assert dims is not None
        out = plant_library_path(dims, plant_header)
        hdr = out[:-3] + "_model.hpp"
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs if os.path.exists(s)):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = list(HIPCC_FLAGS)
    if dims is not None:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        flags.append("-DMPC_DIM_LIST(X)=X(" + ",".join(str(int(v)) for v in dims) + ")")
    if hdr is not None:
        with open(hdr, "w") as fh:
            fh.write(plant_header)
        flags.append(f'-DMPC_NL_PLANT_HEADER="{hdr}"')
    tmp = out + f".{os.getpid()}.tmp"
    if dims is None:
        # the default library: two objects compiled side by side (the C-ABI with the BASELINE dimension sets | the kernels of the other sets,
        # csrc/mpc_amd.hip:MPC_PART2), then linked - about half the wall time of one translation unit
        from concurrent.futures import ThreadPoolExecutor
        cflags = [f for f in flags if f != "-shared"]
        objs = [tmp + ".a.o", tmp + ".b.o"]
        cmds = [[hipcc] + cflags + ["-DMPC_HAVE_PART2", "-c", "-o", objs[0], srcs[0]], [hipcc] + cflags + ["-DMPC_PART2", "-c", "-o", objs[1], srcs[0]]]
        if verbose:
            for c in cmds:
                print(" ".join(c))
        try:
            with ThreadPoolExecutor(max_workers=2) as pool:
                for f in [pool.submit(subprocess.check_call, c, cwd=CSRC) for c in cmds]:
                    f.result()
            subprocess.check_call([hipcc] + flags + ["-o", tmp] + objs, cwd=CSRC)
        finally:
            for o in objs:
                if os.path.exists(o):
                    os.remove(o)
        os.replace(tmp, out)
        return out
    cmd = [hipcc] + flags + ["-o", tmp, srcs[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(tmp, out)
    return out


_lib = None
_libs = {}


def load_library(path: Optional[str] = None) -> ct.CDLL:
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise MpcAmdError(f"{path} not found: build it with mpc_code_amd.capi.build_library() (needs hipcc); "
                          "there is no CPU fallback")
    lib = ct.CDLL(path)
    lib.mpc_last_error.restype = ct.c_char_p
    lib.mpc_build_info.restype = ct.c_char_p
    lib.mpc_last_kernel_ms.restype = ct.c_float
    lib.mpc_last_kernel_ms.argtypes = [ct.c_void_p, _ip]
    lib.mpc_stream.restype = ct.c_void_p
    lib.mpc_stream.argtypes = [ct.c_void_p]
    lib.mpc_dev_ptr.restype = ct.c_void_p
    lib.mpc_dev_ptr.argtypes = [ct.c_void_p, ct.c_char_p, ct.POINTER(ct.c_int64)]
    lib.mpc_destroy.restype = None
    lib.mpc_destroy.argtypes = [ct.c_void_p]
    lib.mpc_lin_create.argtypes = [ct.POINTER(_Desc), ct.POINTER(ct.c_void_p)]
    lib.mpc_set_option.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_double]
    lib.mpc_get_option.argtypes = [ct.c_void_p, ct.c_char_p, ct.POINTER(ct.c_double)]
    lib.mpc_pack_u.argtypes = [ct.c_void_p, ct.c_void_p]
    lib.mpc_pack_log.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int32, ct.c_int32, ct.c_void_p]
    lib.mpc_loop_alloc.argtypes = [ct.c_void_p, ct.c_int32, ct.c_int32, ct.c_int32]
    lib.mpc_loop_run.argtypes = [ct.c_void_p, ct.c_int32, ct.c_int32]
    lib.mpc_loop_sync.argtypes = [ct.c_void_p]
    lib.mpc_loop_get_log.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_void_p]
    lib.mpc_loop_set_state.argtypes = [ct.c_void_p] + [_dp] * 7
    lib.mpc_loop_get_state.argtypes = [ct.c_void_p] + [_dp] * 7
    lib.mpc_loop_set_schedule.argtypes = [ct.c_void_p, ct.c_int32] + [_dp] * 5
    lib.mpc_loop_set_model_schedule.argtypes = [ct.c_void_p, ct.c_int32] + [_dp] * 2
    lib.mpc_ocp_solve.argtypes = [ct.c_void_p, ct.c_int32] + [_dp] * 7 + [_dp, _dp, _dp, _ip, _ip, _dp]
    lib.mpc_target_solve.argtypes = [ct.c_void_p, ct.c_int32] + [_dp] * 5 + [_dp, _dp, _dp, _ip, _ip]
    lib.mpc_kf_update.argtypes = [ct.c_void_p, ct.c_int32, _dp, _dp, _dp]
    lib.mpc_get_slacks.argtypes = [ct.c_void_p, ct.c_int32, _dp]
    lib.mpc_set_model_offsets.argtypes = [ct.c_void_p, ct.c_int32, _dp, _dp]
    lib.mpc_closed_loop.argtypes = [ct.c_void_p, ct.c_int32, ct.c_int32] + [_dp] * 13
    lib.mpc_comm_unique_id.argtypes = [ct.c_char_p]
    lib.mpc_comm_init.argtypes = [ct.c_void_p, ct.c_int32, ct.c_int32, ct.c_char_p]
    lib.mpc_comm_destroy.argtypes = [ct.c_void_p]
    lib.mpc_comm_rank.argtypes = [ct.c_void_p, _ip, _ip]
    lib.mpc_comm_allgather.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_void_p]
    lib.mpc_comm_allreduce_max.argtypes = [ct.c_void_p, _dp, ct.c_int32]
    lib.mpc_comm_barrier.argtypes = [ct.c_void_p]
    lib.mpc_allgather_u.argtypes = [ct.c_void_p, _dp]
    lib.mpc_allgather_log.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int32, ct.c_int32, _dp]
    if path == LIB_PATH:
        _lib = lib
    _libs[path] = lib
    return lib


EXPORTS = ("mpc_lin_create", "mpc_destroy", "mpc_last_error", "mpc_ocp_solve", "mpc_get_slacks", "mpc_target_solve", "mpc_kf_update", "mpc_set_model_offsets",
           "mpc_loop_alloc", "mpc_loop_set_state", "mpc_loop_get_state", "mpc_loop_set_schedule", "mpc_loop_set_model_schedule", "mpc_loop_run",
           "mpc_loop_sync", "mpc_loop_get_log", "mpc_closed_loop", "mpc_last_kernel_ms", "mpc_stream", "mpc_dev_ptr",
           "mpc_pack_u", "mpc_pack_log", "mpc_set_option", "mpc_get_option", "mpc_build_info",
           "mpc_comm_unique_id", "mpc_comm_init", "mpc_comm_destroy", "mpc_comm_rank", "mpc_comm_allgather",
           "mpc_comm_allreduce_max", "mpc_comm_barrier", "mpc_allgather_u", "mpc_allgather_log")


def _c(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = np.ascontiguousarray(np.broadcast_to(a, shape))
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


class Solver:
    """A problem resident on one GPU.  ``problem`` is a :class:`LinearMPCProblem`."""

    def __init__(self, problem, device: int = 0, lib_path: Optional[str] = None, jit: bool = True):
        self.lib = load_library(lib_path)
        self.p = p = problem
        self._keep = {}
        d = _Desc()
        d.nx, d.nu, d.ny, d.nd, d.nxp, d.N = p.nx, p.nu, p.ny, p.nd, p.nxp, p.N
        d.du_form, d.duss_form, d.y_bounded = int(p.DUForm), int(p.DUssForm), int(p.y_bounded)
        d.estimator, d.max_iter, d.device = _EST[p.estimator], int(p.max_iter), int(device)
        d.term_cons = int(bool(getattr(p, "TermCons", False)))
        d.h_sample = float(p.h)
        d.slacks = int(bool(getattr(p, "slacks", False)))
        if d.slacks:
            self._keep["Ws"] = _c(p.Ws); d.Ws = _p(self._keep["Ws"])
        d.n_user_rows = int(getattr(p, "n_user_rows", 0))
        if d.n_user_rows:      # affine User_g_ineq rows (problem.py:_affine_user_rows)
            for k in ("Gx", "Gu", "Gd", "g0"):
                a = _c(getattr(p, k)); a = a if a.size else np.zeros(1)
                self._keep[k] = a; setattr(d, k, _p(a))
        self.fused_plant = False
        if lib_path is None and not p.plant_is_linear and jit and not os.environ.get("MPC_AMD_NO_JIT"):
            # the Ex-file's plant function, traced and compiled into a library of this problem's own (cached under csrc/jit/)
            from . import nlcodegen
            try:
                hdr = nlcodegen.emit_plant_header(p)
            except Exception:      # noqa: BLE001 - a plant the tracer cannot follow stays on the host (call-by-call mode)
                hdr = None
            if hdr is not None:
                ng = 0 if getattr(p, "slacks", False) else sum(1 for i in range(p.ny) if p.y_bounded and (np.isfinite(p.ymin[i]) or np.isfinite(p.ymax[i])) and np.count_nonzero(p.C[i]) != 1)
                ng += int(getattr(p, "n_user_rows", 0))
                dims = (p.nx, p.nu, p.ny, p.nd, p.nxp, int(p.DUForm or p.Dumin is not None or p.Dumax is not None), ng)
                self.lib = load_library(build_library(dims=dims, plant_header=hdr))
                d.nl_plant = 1
                self.fused_plant = True
        for k in ("A", "B", "C", "Bd", "Cd", "fx_const", "fy_const", "Ap", "Bp", "Cp", "Q", "R", "P", "Qss", "Rss",
                  "umin", "umax", "xmin", "xmax", "ymin", "ymax", "umin_ss", "umax_ss", "xmin_ss", "xmax_ss",
                  "ymin_ss", "ymax_ss", "dmin", "dmax", "Q_kf", "R_kf", "K", "Dumin", "Dumax"):
            v = getattr(p, k, None)
            if v is None:
                setattr(d, k, None)
            else:
                a = _c(v)
                if a.size == 0:
                    a = np.zeros(1)
                self._keep[k] = a
                setattr(d, k, _p(a))
        self.h = ct.c_void_p()
        rc = self.lib.mpc_lin_create(ct.byref(d), ct.byref(self.h))
        if rc == -5 and lib_path is None and jit and not os.environ.get("MPC_AMD_NO_JIT"):
            # no kernel compiled for these dimensions: build the library of exactly this set (about a minute of hipcc, then cached
            # under csrc/jit/) - the kernels are templates on the dimensions, any stage state <= 8 and nu <= 4 compiles
            import re
            m = re.search(r"nx=(\d+) nu=(\d+) ny=(\d+) nd=(\d+) nxp=(\d+) du_form=(\d+) general_output_rows=(\d+)", self.lib.mpc_last_error().decode())
            if m:
                self.lib = load_library(build_library(dims=tuple(int(v) for v in m.groups())))
                rc = self.lib.mpc_lin_create(ct.byref(d), ct.byref(self.h))
        if rc != 0:
            self.h = None
            raise MpcAmdError(f"mpc_lin_create failed ({rc}): {self.lib.mpc_last_error().decode()}")
        self._loop_B = 0
        self._loop_steps = 0

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc, what):
        if rc != 0:
            raise MpcAmdError(f"{what} failed ({rc}): {self.lib.mpc_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.mpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_info(self) -> str:
        return self.lib.mpc_build_info().decode()

    def set_option(self, name: str, value: float):
        self._chk(self.lib.mpc_set_option(self.h, name.encode(), float(value)), "mpc_set_option")

    def get_option(self, name: str) -> float:
        v = ct.c_double(0.0)
        self._chk(self.lib.mpc_get_option(self.h, name.encode(), ct.byref(v)), "mpc_get_option")
        return float(v.value)

    def last_kernel_ms(self):
        n = ct.c_int32(0)
        ms = self.lib.mpc_last_kernel_ms(self.h, ct.byref(n))
        return float(ms), int(n.value)

    def stream(self) -> int:
        return int(self.lib.mpc_stream(self.h) or 0)

    def dev_ptr(self, name: str):
        bp = ct.c_int64(0)
        ptr = self.lib.mpc_dev_ptr(self.h, name.encode(), ct.byref(bp))
        return (int(ptr) if ptr else 0), int(bp.value)

    def pack_u(self, dst_ptr: int):
        self._chk(self.lib.mpc_pack_u(self.h, ct.c_void_p(dst_ptr)), "mpc_pack_u")

    def pack_log(self, name: str, k0: int, nsteps: int, dst_ptr: int):
        self._chk(self.lib.mpc_pack_log(self.h, name.encode(), int(k0), int(nsteps), ct.c_void_p(dst_ptr)), "mpc_pack_log")

    # ------------------------------------------------------------------ multi-GPU (RCCL inside the library)
    def comm_unique_id(self) -> bytes:
        buf = ct.create_string_buffer(128)
        self._chk(self.lib.mpc_comm_unique_id(buf), "mpc_comm_unique_id")
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        self._chk(self.lib.mpc_comm_init(self.h, int(rank), int(world), unique_id), "mpc_comm_init")

    def comm_rank(self):
        r, w = ct.c_int32(0), ct.c_int32(1)
        self._chk(self.lib.mpc_comm_rank(self.h, ct.byref(r), ct.byref(w)), "mpc_comm_rank")
        return r.value, w.value

    def comm_barrier(self):
        self._chk(self.lib.mpc_comm_barrier(self.h), "mpc_comm_barrier")

    def comm_allreduce_max(self, values):
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self._chk(self.lib.mpc_comm_allreduce_max(self.h, _p(v), int(v.size)), "mpc_comm_allreduce_max")
        return v

    def comm_allgather(self, send: np.ndarray) -> np.ndarray:
        """Equal-sized host arrays of every rank -> [world, ...] on every rank."""
        send = np.ascontiguousarray(send)
        _, world = self.comm_rank()
        recv = np.empty((world,) + send.shape, dtype=send.dtype)
        self._chk(self.lib.mpc_comm_allgather(self.h, send.ctypes.data_as(ct.c_void_p), send.nbytes, recv.ctypes.data_as(ct.c_void_p)),
                  "mpc_comm_allgather")
        return recv

    def allgather_u(self, to_host: bool = True):
        """u* of the last step of every rank, [world, B, nu]; None when left on the device."""
        _, world = self.comm_rank()
        out = np.empty((world, self._loop_B, self.p.nu)) if to_host else None
        self._chk(self.lib.mpc_allgather_u(self.h, _p(out)), "mpc_allgather_u")
        return out

    def allgather_log(self, name: str, k0: int, nsteps: int, to_host: bool = True):
        """Steps [k0, k0+nsteps) of a float64 log of every rank, [world, nsteps, B, dim]; None when left on the device."""
        p = self.p
        dims = dict(U=p.nu, X_HAT=p.nx, XS=p.nx, US=p.nu, YS=p.ny, Xp=p.nxp, D_HAT=p.nd, SL=2 * p.ny)      # SL: the slacks of a problem with soft output constraints
        _, world = self.comm_rank()
        out = np.empty((world, int(nsteps), self._loop_B, dims[name])) if to_host else None
        self._chk(self.lib.mpc_allgather_log(self.h, name.encode(), int(k0), int(nsteps), _p(out)), "mpc_allgather_log")
        return out

    # ------------------------------------------------------------------ per-step calls
    def ocp_solve(self, xhat, xs, us, dhat, u_prev, want_w=False, w_guess=None, px=None, py=None):
        """``solver(...)`` of MPC_code.py:776-781 for a batch; returns dict(u0, x1, status, iters, res[, w]).

        ``px`` [B, N, nx] / ``py`` [B, N, ny] (or [N, .], shared): the model parameters over the horizon (``def_px`` / ``def_py``,
        MPC_code.py:492-497).

        ``w_guess`` [B, nw]: the reference's ``x0=`` (MPC_code.py:740-764); read only after ``set_option("ocp_warm_start", 1)``."""
        p = self.p
        xhat = _c(np.atleast_2d(xhat)); B = xhat.shape[0]
        xs, us = _c(xs, (B, p.nx)), _c(us, (B, p.nu))
        dhat, u_prev = _c(dhat, (B, p.nd)), _c(u_prev, (B, p.nu))
        u0 = np.full((B, p.nu), np.nan); x1 = np.full((B, p.nx), np.nan)
        st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); res = np.zeros((B, 3))
        w = np.full((B, p.nw), np.nan) if (want_w or w_guess is not None) else None
        if w_guess is not None:
            w[:] = np.broadcast_to(np.asarray(w_guess, dtype=np.float64), (B, p.nw))
        pxa = None if px is None else np.ascontiguousarray(np.broadcast_to(np.asarray(px, dtype=np.float64), (B, p.N, p.nx)))
        pya = None if py is None else np.ascontiguousarray(np.broadcast_to(np.asarray(py, dtype=np.float64), (B, p.N, p.ny)))
        self._chk(self.lib.mpc_ocp_solve(self.h, B, _p(xhat), _p(xs), _p(us), _p(dhat) if p.nd else None, _p(u_prev),
                                         _p(pxa), _p(pya), _p(w), _p(u0), _p(x1), _pi(st), _pi(it), _p(res)), "mpc_ocp_solve")
        out = dict(u0=u0, x1=x1, status=st, iters=it, res=res, w=w)
        if getattr(p, "slacks", False):      # sl_k = w_opt[nw-ns:nw], MPC_code.py:800
            sl = np.zeros((B, 2 * p.ny))
            self._chk(self.lib.mpc_get_slacks(self.h, B, _p(sl)), "mpc_get_slacks")
            out["sl"] = sl
        return out

    def target_solve(self, usp, ysp, xsp, dhat, us_prev):
        """``solver_ss(...)`` of MPC_code.py:704-709 for a batch; returns dict(xs, us, ys, status, iters)."""
        p = self.p
        dhat = _c(np.atleast_2d(dhat)); B = dhat.shape[0]
        usp, ysp, xsp, us_prev = _c(usp, (B, p.nu)), _c(ysp, (B, p.ny)), _c(xsp, (B, p.nx)), _c(us_prev, (B, p.nu))
        xs = np.zeros((B, p.nx)); us = np.zeros((B, p.nu)); ys = np.zeros((B, p.ny))
        st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
        self._chk(self.lib.mpc_target_solve(self.h, B, _p(usp), _p(ysp), _p(xsp), _p(dhat) if p.nd else None, _p(us_prev),
                                            _p(xs), _p(us), _p(ys), _pi(st), _pi(it)), "mpc_target_solve")
        return dict(xs=xs, us=us, ys=ys, status=st, iters=it)

    def set_model_offsets(self, B, px0=None, py0=None):
        """This step's ``p_x_k`` [B, nx] / ``p_y_k`` [B, ny] (MPC_code.py:500-501) for the following kf_update / target_solve calls;
        ``None`` clears."""
        p = self.p
        a = None if px0 is None else _c(px0, (B, p.nx)); b = None if py0 is None else _c(py0, (B, p.ny))
        self._chk(self.lib.mpc_set_model_offsets(self.h, int(B), _p(a), _p(b)), "mpc_set_model_offsets")

    def kf_update(self, y, xi, P=None):
        """``defEstimator(...)`` of MPC_code.py:577-650; returns (xi_corrected, P_plus)."""
        p = self.p
        xi = _c(np.atleast_2d(xi)).copy(); B = xi.shape[0]
        y = _c(y, (B, p.ny))
        ne = p.nx + p.nd
        Pk = _c(P, (B, ne, ne)).copy() if P is not None else None
        self._chk(self.lib.mpc_kf_update(self.h, B, _p(y), _p(xi), _p(Pk)), "mpc_kf_update")
        return xi, Pk

    # ------------------------------------------------------------------ resident closed loop
    def loop_alloc(self, B: int, max_steps: int, log_level: int = LOG_ALL):
        self._chk(self.lib.mpc_loop_alloc(self.h, int(B), int(max_steps), int(log_level)), "mpc_loop_alloc")
        self._loop_B, self._loop_steps, self._loop_log = int(B), int(max_steps), int(log_level)

    def loop_set_state(self, x_p, xhat, dhat=None, P=None, u=None, xs=None, us=None):
        p, B = self.p, self._loop_B
        ne = p.nx + p.nd
        x_p, xhat = _c(x_p, (B, p.nxp)), _c(xhat, (B, p.nx))
        dhat = _c(p.dhat0 if dhat is None else dhat, (B, p.nd)) if p.nd else None
        u = _c(p.u0 if u is None else u, (B, p.nu))
        xs = xhat.copy() if xs is None else _c(xs, (B, p.nx))       # MPC_code.py:682-684
        us = u.copy() if us is None else _c(us, (B, p.nu))
        Pk = None
        if p.estimator == "kal":
            Pk = _c(p.P0 if P is None else P, (B, ne, ne))
        self._chk(self.lib.mpc_loop_set_state(self.h, _p(x_p), _p(xhat), _p(dhat), _p(Pk), _p(u), _p(xs), _p(us)),
                  "mpc_loop_set_state")

    def loop_get_state(self):
        p, B = self.p, self._loop_B
        ne = p.nx + p.nd
        out = dict(x_p=np.zeros((B, p.nxp)), xhat=np.zeros((B, p.nx)), dhat=np.zeros((B, p.nd)) if p.nd else None,
                   P=np.zeros((B, ne, ne)) if p.estimator == "kal" else None, u=np.zeros((B, p.nu)),
                   xs=np.zeros((B, p.nx)), us=np.zeros((B, p.nu)))
        self._chk(self.lib.mpc_loop_get_state(self.h, *(_p(out[k]) for k in ("x_p", "xhat", "dhat", "P", "u", "xs", "us"))),
                  "mpc_loop_get_state")
        return out

    def loop_set_schedule(self, sched, nsteps=None):
        nsteps = len(sched["ysp"]) if nsteps is None else nsteps
        self._sched_keep = {k: _c(v) for k, v in sched.items()}
        s = self._sched_keep
        self._chk(self.lib.mpc_loop_set_schedule(self.h, int(nsteps), _p(s["ysp"]), _p(s["usp"]), _p(s.get("xsp")),
                                                 _p(s.get("pxp")), _p(s.get("pyp"))), "mpc_loop_set_schedule")
        self._sched_n = int(nsteps)

    def loop_set_model_schedule(self, px=None, py=None):
        """``def_px`` / ``def_py`` for every step of the fused loop: ``px`` [nsteps, N, nx] with ``px[k, i] = def_px(t_k + i)``, ``py`` [nsteps, N, ny]
        likewise (``MPC_code.py:492-497``); both ``None`` switch them off."""
        p = self.p
        pxa = None if px is None else np.ascontiguousarray(np.asarray(px, dtype=np.float64).reshape(-1, p.N, p.nx))
        pya = None if py is None else np.ascontiguousarray(np.asarray(py, dtype=np.float64).reshape(-1, p.N, p.ny))
        n = 0 if pxa is None and pya is None else len(pxa if pxa is not None else pya)
        self._msched_keep = (pxa, pya)
        self._chk(self.lib.mpc_loop_set_model_schedule(self.h, int(n), _p(pxa), _p(pya)), "mpc_loop_set_model_schedule")

    def loop_run(self, k0: int, nsteps: int):
        self._chk(self.lib.mpc_loop_run(self.h, int(k0), int(nsteps)), "mpc_loop_run")

    def loop_sync(self):
        self._chk(self.lib.mpc_loop_sync(self.h), "mpc_loop_sync")

    def loop_get_log(self, name: str):
        p, B, ns = self.p, self._loop_B, self._sched_n
        dims = dict(U=p.nu, X_HAT=p.nx, XS=p.nx, US=p.nu, YS=p.ny, Xp=p.nxp, D_HAT=p.nd, SL=2 * p.ny)      # SL: the slacks of a problem with soft output constraints
        if name in dims:
            out = np.zeros((ns, B, dims[name]))
        else:
            out = np.zeros((ns, B), np.int32)
        self._chk(self.lib.mpc_loop_get_log(self.h, name.encode(), out.ctypes.data_as(ct.c_void_p)), "mpc_loop_get_log")
        return out
