"""The closed loop of the reference's ``MPC_code.py:485-875`` over a batch of instances, on the GPU.

``run_closed_loop`` is what "python MPC_code.py" does after its set-up section, for B instances
that share the Ex-file and differ in their initial state; the result arrays carry the reference's
names (``MPC_code.py:877-895``): ``U, X_HAT, Y_HAT, XS, US, YS, Xp, Yp, D_HAT, TIME_SS, TIME_DYN`` plus the solver
status / iteration words per step.  ``Yp`` is the measurement ``Fy_p(x_k) + pyp`` (``:531-534``), ``Y_HAT`` the prediction
``Fy_model(xhat_k, dhat_k)`` of it from the prior estimate (``:524``).  ``TIME_SS`` / ``TIME_DYN`` are the reference's
wall-clock pair around the two solver calls (``:703-711``, ``:775-783``) in the call-by-call mode; the fused kernel cannot
separate them: there ``TIME_DYN`` is the kernel time of a whole step (HIP events / steps) and ``TIME_SS`` is zero.  Two modes:

* ``fused=True``  - ``mpc_loop_run``: one kernel launch advances all instances one or more steps
  (estimator, target, OCP, plant fused), state resident in HBM.
* ``fused=False`` - the reference's own call sequence, one C-ABI call per solver per step
  (``mpc_kf_update``, ``mpc_target_solve``, ``mpc_ocp_solve``) with the glue of ``MPC_code.py:524-816``
  in NumPy.  Slower (host round trips) but it is the literal drop-in for the three call sites.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import capi
from .shard import allgather_rows, shard_bounds


def _bcast(v, B, d):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64).reshape(-1, d) if np.ndim(v) > 1 else np.asarray(v, dtype=np.float64), (B, d)))


def run_closed_loop(problem, x0_p=None, x0_m=None, nsteps: Optional[int] = None, solver: Optional[capi.Solver] = None,
                    fused: bool = True, device: int = 0, gather: bool = False, total: Optional[int] = None, comm=None,
                    warm_start: bool = False) -> Dict[str, np.ndarray]:
    """``gather=True``: this rank ran its shard of a batch of ``total`` instances; every result array is all-gathered over the
    ranks of ``comm`` (mpc-code_amd/shard.py).  ``total`` is required then: a shard does not know the size of the whole.
    ``warm_start`` (call-by-call mode): hand ``mpc_ocp_solve`` the shifted previous optimum as the reference hands it to IPOPT
    (``MPC_code.py:740-764``) and let it keep its multipliers between calls (``ocp_warm_start``); the fused loop always does."""
    p = problem
    nsteps = p.Nsim if nsteps is None else int(nsteps)
    x0_p = p.x0_p[None] if x0_p is None else np.atleast_2d(x0_p)
    x0_m = p.x0_m[None] if x0_m is None else np.atleast_2d(x0_m)
    B = x0_p.shape[0]
    own = solver is None
    s = capi.Solver(p, device=device) if own else solver
    try:
        sched = p.schedules(nsteps)
        if fused and not p.plant_is_linear and not getattr(s, "fused_plant", False):
            raise ValueError("this solver's library has no compiled plant function (User_fxp_Cont): the plant is simulated on the host, "
                             "call run_closed_loop(..., fused=False)")
        if fused:
            s.loop_alloc(B, nsteps, capi.LOG_ALL)
            s.loop_set_state(x0_p, x0_m)
            s.loop_set_schedule(sched)
            py0 = np.zeros((nsteps, p.ny))
            if p.has_model_params:                                   # :492-510: def_px / def_py over the horizon, for every step of the loop
                hp = [p.horizon_params(k * p.h) for k in range(nsteps)]
                s.loop_set_model_schedule(np.stack([h_[0] for h_ in hp]) if p.def_px is not None else None,
                                          np.stack([h_[1] for h_ in hp]) if p.def_py is not None else None)
                py0 = np.stack([h_[1][0] for h_ in hp])              # p_y_k (zeros without def_py)
            else:
                s.loop_set_model_schedule(None, None)
            s.loop_run(0, nsteps)
            s.loop_sync()
            out = {k: s.loop_get_log(k) for k in ("U", "X_HAT", "XS", "US", "YS", "Xp", "D_HAT", "STATUS_DYN",
                                                  "STATUS_SS", "ITERS_DYN", "ITERS_SS") if not (k == "D_HAT" and p.nd == 0)}
            if getattr(p, "slacks", False):
                out["Sl"] = s.loop_get_log("SL")                     # :800,808-809 (csrc/mpc_amd.hip:loop_kernel_soft)
            out["Yp"] = out["Xp"] @ p.Cp.T + sched["pyp"][:nsteps, None, :] + py0[:, None, :]                 # :531-534 (p_ymp = p_y_k, :505-507)
            d_prior = np.zeros((nsteps, B, p.nd))
            if p.nd:                                                 # dhat is carried unchanged between steps (:655-668)
                d_prior[0] = _bcast(p.dhat0, B, p.nd); d_prior[1:] = out["D_HAT"][:-1]
            out["Y_HAT"] = out["X_HAT"] @ p.C.T + p.fy_const + (d_prior @ p.Cd.T if p.nd else 0.0) + py0[:, None, :]   # :524
            ms, _ = s.last_kernel_ms()
            out["TIME_DYN"] = np.full(nsteps, ms * 1e-3 / nsteps); out["TIME_SS"] = np.zeros(nsteps)
        else:
            out = _stepwise(p, s, x0_p, x0_m, nsteps, sched, warm_start=warm_start)
    finally:
        if own:
            s.close()
    if gather:
        if total is None:
            if comm is not None and comm.world > 1:
                raise ValueError("run_closed_loop(gather=True) over several ranks needs total= (the size of the whole batch)")
            total = B
        out = {k: (np.moveaxis(allgather_rows(np.moveaxis(v, 1, 0), total, comm), 0, 1) if np.ndim(v) >= 2 else v) for k, v in out.items()}
    return out


def _stepwise(p, s, x0_p, x0_m, nsteps, sched, warm_start=False):
    """The reference's call sequence, MPC_code.py:519-816, with each solver call going through the C-ABI."""
    B, n = x0_p.shape[0], p.nx
    x, xhat = x0_p.copy(), x0_m.copy()
    u = _bcast(p.u0, B, p.nu).copy(); dhat = _bcast(p.dhat0, B, p.nd).copy() if p.nd else np.zeros((B, 0))
    Pk = np.broadcast_to(p.P0, (B,) + p.P0.shape).copy() if p.estimator == "kal" else None
    us_k, xs_k = u.copy(), x0_m.copy()                                   # :682-684
    import time
    keys = ("U", "X_HAT", "Y_HAT", "XS", "US", "YS", "Xp", "Yp", "D_HAT", "STATUS_DYN", "STATUS_SS", "ITERS_DYN", "ITERS_SS", "TIME_SS", "TIME_DYN") + (("Sl",) if getattr(p, "slacks", False) else ())
    sl_k = np.zeros((B, 2 * p.ny))
    log = {k: [] for k in keys}
    s.set_option("ocp_warm_start", 1 if warm_start else 0)
    # The reference builds x0= for IPOPT itself (first guess (x0_m, u0) tiled :740-756, then the previous optimum shifted by one
    # stage :760-764).  mpc_ocp_solve keeps the previous optimum and its multipliers in the handle and shifts them on the device, so
    # nothing has to travel; an explicit guess can still be passed (capi.Solver.ocp_solve(w_guess=...)).
    pxh = pyh = None; px0 = np.zeros(p.nx); py0 = np.zeros(p.ny)
    for k in range(nsteps):
        if p.has_model_params:                                           # :492-510: p_xk, p_yk over the horizon; the plant gets p_x_k, p_y_k too
            pxh, pyh = p.horizon_params(k * p.h)
            px0, py0 = pxh[0], pyh[0]
            s.set_model_offsets(B, px0 if p.def_px is not None else None, py0 if p.def_py is not None else None)
        log["Xp"].append(x.copy()); log["X_HAT"].append(xhat.copy())      # :519-520
        log["Y_HAT"].append(xhat @ p.C.T + p.fy_const + (dhat @ p.Cd.T if p.nd else 0.0) + py0)   # :524
        y = x @ p.Cp.T + sched["pyp"][k] + py0                          # :534
        log["Yp"].append(y.copy())
        if p.estimator != "none":
            xi, Pk = s.kf_update(y, np.hstack([xhat, dhat]), Pk)          # :577-650
            xhat, dhat = xi[:, :n].copy(), xi[:, n:].copy()
            if p.dmin is not None and p.dmax is not None:                # both or neither (problem.py refuses one-sided)
                dhat = np.minimum(np.maximum(dhat, p.dmin), p.dmax)       # :660-665
        log["D_HAT"].append(dhat.copy())
        t0 = time.time()                                                                     # :703
        t = s.target_solve(sched["usp"][k], sched["ysp"][k], sched["xsp"][k], dhat, us_k)   # :704-709
        log["TIME_SS"].append(time.time() - t0)                                              # :711,729
        ok = (t["status"] != capi.STATUS_INFEASIBLE)[:, None]
        xs_k = np.where(ok, t["xs"], xs_k); us_k = np.where(ok, t["us"], us_k)               # :714-718
        log["XS"].append(xs_k.copy()); log["US"].append(us_k.copy())
        log["YS"].append(xs_k @ p.C.T + p.fy_const + (dhat @ p.Cd.T if p.nd else 0.0) + py0)      # :730
        t0 = time.time()                                                                     # :775
        o = s.ocp_solve(xhat, xs_k, us_k, dhat, u, px=pxh if p.def_px is not None else None, py=pyh if p.def_py is not None else None)   # :776-781
        log["TIME_DYN"].append(time.time() - t0)                                             # :783,810
        ok = (o["status"] != capi.STATUS_INFEASIBLE)[:, None]
        hold = xhat @ p.A.T + u @ p.B.T + p.fx_const + (dhat @ p.Bd.T if p.nd else 0.0) + px0      # :804-805
        u = np.where(ok, o["u0"], u); xhat = np.where(ok, o["x1"], hold)                     # :798-799
        log["U"].append(u.copy())
        if "sl" in o:                                                                        # :800,808-809: Sl.append(sl_k), the last accepted one
            sl_k = np.where(ok, o["sl"], sl_k); log["Sl"].append(sl_k.copy())
        log["STATUS_DYN"].append(o["status"]); log["STATUS_SS"].append(t["status"])
        log["ITERS_DYN"].append(o["iters"]); log["ITERS_SS"].append(t["iters"])
        x = p.plant_step(x, u, k * p.h, sched["pxp"][k] + (px0 if p.has_model_params else 0.0))                               # :813-816 (p_xmp = p_x_k, :502-505)
    return {k: np.array(v) for k, v in log.items()}
