#!/usr/bin/env python3
"""Headline benchmark: closed-loop MPC steps/s over a batch, LMPC-CSTR, N=50 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A *step* is one closed-loop step (estimator + target + OCP + plant, reference MPC_code.py:485-827)
of every instance of the per-GPU batch.  Workload at N=1: BASELINE.json configs[1] - the shipped
Ex_LMPC_CSTR problem, batch 4096 initial states drawn as BASELINE.md section 3 says.  For N>1 the driver
launches this file once per GPU under torch.distributed.run; each rank owns its own 4096 instances
(weak scaling), the only exchange is the all-gather of the optimal controls U at the end of the
timed region (RCCL).  PyTorch is used for rendezvous, barrier, the collective and device-wide
synchronisation only; the solver is libmpc_amd.so through ctypes.

The timed region holds exactly K steps from t = 0 with all inputs resident in HBM; the W warm-up
steps run before it from the same initial state, which is then restored (untimed).

Prints ONE JSON line (rank 0).  `roofline` prices the fused closed-loop kernel against HBM with the
algorithmic bytes of DESIGN.md section 6 (this path is latency / fp64-issue bound, not HBM bound);
`cpu_baseline` times oracle/mpc_oracle.c (a C port of the same algorithm - the reference's own
CasADi/IPOPT path cannot run here) on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KERNEL_NAMES = {1: "loop_kernel (one instance per lane)", 2: "loop_kernel_tp (horizon-parallel)"}
B_PER_GPU = 4096
SEED = 20250614
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def alg_bytes_per_step(p) -> int:
    """Algorithmic HBM bytes per closed-loop step per instance.

    State in/out as SURVEY 8d (920 B for the CSTR) plus the warm starts this solver carries between steps, in and
    out: for the OCP the inputs and the bound multipliers per stage (the reference carries the primal w,
    MPC_code.py:764), for the target problem its reduced optimum, multipliers and the QP vectors they belong to."""
    ne = p.nx + p.nd
    n_in = p.nxp + p.nx + p.nd + (ne * ne if p.estimator == "kal" else 0) + p.nu + (p.ny + p.nu + p.nx) + (p.nx + p.nu)
    n_out = p.nxp + p.nx + p.nd + (ne * ne if p.estimator == "kal" else 0) + p.nu + p.nx + p.nu + p.ny
    nbounded = p.nu + (p.nx if (np.isfinite(p.xmin).any() or np.isfinite(p.xmax).any() or p.y_bounded) else 0)
    warm_ocp = 2 * p.N * (p.nu + 2 * nbounded)
    warm_target = 2 * (2 * p.nu + 3 * (p.nx + p.nu + p.ny))
    return 8 * (n_in + n_out + warm_ocp + warm_target)


def measured_traffic():
    """HBM bytes per launch of the closed-loop kernel from the committed PMC summary (rocprofv3 cannot run inside
    this process); None if the summary is missing."""
    f = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    try:
        return float(json.load(open(f))["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(problem, x0, nsteps, target_seconds=10.0, max_seconds=40.0):
    """Time the oracle's C restatement on the host cores: the same closed loop (same instances, same steps, from
    t=0), repeated until about `target_seconds` of wall time have been spent (at least once, bounded above)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    oc = oracle_c.OracleC(problem)
    nthr = oc.max_threads()
    nb = len(x0)
    t0 = time.perf_counter(); oc.closed_loop(min(nsteps, 10), x0[:min(nb, 256)], x0[:min(nb, 256)], logs=False); t1 = time.perf_counter() - t0
    rate = min(nb, 256) * min(nsteps, 10) / max(t1, 1e-6)            # rough, only to bound the sample
    if nb * nsteps / rate > max_seconds:                             # slow host: shrink the sample, say so
        nb = max(64, int(max_seconds * rate / nsteps))
    reps, spent = 0, 0.0
    while reps == 0 or (spent < target_seconds and spent * (reps + 1) / reps < max_seconds):
        t0 = time.perf_counter(); oc.closed_loop(nsteps, x0[:nb], x0[:nb], logs=False); spent += time.perf_counter() - t0
        reps += 1
    return dict(value=reps * nb * nsteps / spent, unit="steps/s", cores=nthr, kind="port",
                sample=f"{nb} instances x {nsteps} closed-loop steps from t=0 of the same workload, {reps} repetitions, {spent:.1f} s wall: "
                       f"oracle/mpc_oracle.c (C port of the same Riccati-PDIP with the same warm start, gcc -O2 -fopenmp, {nthr} threads); "
                       "the reference's own CasADi/IPOPT path is not installable here")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="instances per GPU (default: BASELINE configs[1])")
    ap.add_argument("--steps-per-launch", type=int, default=0, help="closed-loop steps per kernel launch (0: library default)")
    ap.add_argument("--loop-kernel", type=int, default=0, help="0: library default (by batch size), 1: instance per lane, 2: horizon-parallel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even for one rank: exercises the N>1 code path on a 1-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    K, W, B = args.steps, args.warmup, args.batch
    use_dist = world > 1 or args.force_dist

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import mpc_code_amd as m
    from mpc_code_amd import capi
    prob = m.load_problem(m.example_path("cstr_lmpc.py"))
    if rank == 0:
        capi.build_library()          # no-op when the in-tree .so is current; one rank only (no concurrent hipcc)
    if use_dist:
        dist.barrier()
    solver = capi.Solver(prob, device=local_rank)
    if args.steps_per_launch > 0:
        solver.set_option("steps_per_launch", args.steps_per_launch)
    solver.set_option("loop_kernel", args.loop_kernel)

    # synthetic initial states: the same generator for the whole job, each rank takes its block
    rng = np.random.default_rng(SEED)
    x0_all = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B * world, 3))
    x0 = x0_all[rank * B:(rank + 1) * B]
    nsched = max(K, W, 1)
    sched = prob.schedules(nsched)
    solver.loop_alloc(B, nsched, capi.LOG_U)
    solver.loop_set_schedule(sched)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # gather buffers for U: [K][nu][Bpad] per rank
    _, bpad = solver.dev_ptr("U")
    send = torch.empty((K, prob.nu, bpad), dtype=torch.float64, device="cuda")
    recv = torch.empty((world,) + tuple(send.shape), dtype=torch.float64, device="cuda") if use_dist else None

    # warm-up (untimed), then restore the initial state
    solver.loop_set_state(x0, x0)
    if W > 0:
        solver.loop_run(0, W)
        solver.loop_sync()
    if use_dist:
        dist.all_gather_into_tensor(recv, send)
    solver.loop_set_state(x0, x0)

    barrier()
    t0 = time.perf_counter()
    solver.loop_run(0, K)
    solver.pack_log("U", 0, K, send.data_ptr())
    solver.loop_sync()
    if use_dist:
        dist.all_gather_into_tensor(recv, send)
    barrier()
    dt = time.perf_counter() - t0

    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kernel_ms, n_launch = solver.last_kernel_ms()
    loop_kernel = int(solver.get_option("loop_kernel"))

    if rank == 0:
        st = solver.loop_get_log("STATUS_DYN")[:K]
        it = solver.loop_get_log("ITERS_DYN")[:K]
        steps_total = B * world * K
        per_launch_s = kernel_ms * 1e-3 / max(n_launch, 1)
        steps_per_launch = B * K / max(n_launch, 1)
        ab = alg_bytes_per_step(prob)
        achieved = ab * steps_per_launch / per_launch_s / 1e9
        out = {
            "metric": "closed-loop MPC steps/sec over batch, LMPC-CSTR N=50",
            "value": steps_total / dt, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Ex_LMPC_CSTR (nx=3,nu=2,ny=3,nd=3), N=50, batch=%d per GPU, x0~U([-0.5,0.5]x[-8,8]x[-5,5]) seed %d, "
                                   "closed loop from t=0: Kalman filter + target QP + OCP (Riccati-PDIP) + plant per step" % (B, SEED),
                       "batch_per_gpu": B, "horizon": prob.N, "steps_per_launch": K / max(n_launch, 1), "loop_kernel": KERNEL_NAMES[loop_kernel],
                       "parallelism": "instances sharded over %d GPU(s), all-gather of U at the end" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (measured_traffic() if (B == B_PER_GPU and K == 100) else None),
                         "traffic_source": "profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/run_loop.py --batch 4096 --steps 100; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, per launch)",
                         "kernel": KERNEL_NAMES[loop_kernel], "launches": n_launch,
                         "avg_launch_ms": per_launch_s * 1e3, "alg_bytes_per_step": ab,
                         "note": "bound by single-wave instruction issue in the sequential recursions (Riccati factorisation of four instances per wave on the fp64 matrix cores, vector recursions on one wave per 16 instances), not by HBM (SURVEY.md 8d)"},
            "solver": {"mean_iters": float(it[st != 2].mean()) if (st != 2).any() else None, "max_iters": int(it.max()),
                       "frac_solved": float((st == 0).mean()), "frac_maxiter": float((st == 1).mean()),
                       "frac_infeasible_hold": float((st == 2).mean())},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, x0, K)
        print(json.dumps(out), flush=True)
    if use_dist and rank == 0:      # the gathered block of this rank must be what the kernel logged
        mine = recv[rank].cpu().numpy()[:, :, :B]
        assert np.array_equal(np.moveaxis(mine, 2, 1), solver.loop_get_log("U")[:K]), "all-gather of U corrupted the data"
    solver.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
