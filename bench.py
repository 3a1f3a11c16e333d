#!/usr/bin/env python3
"""Headline benchmark: closed-loop MPC steps/s over a batch, LMPC-CSTR, N=50 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A *step* is one closed-loop step (estimator + target + OCP + plant, reference MPC_code.py:485-827)
of every instance of the per-GPU batch.  Workload at N=1: BASELINE.json configs[1] - the shipped
Ex_LMPC_CSTR problem, batch 4096 initial states drawn as BASELINE.md section 3 says.  For N>1 the driver
launches this file once per GPU (torch.distributed.run is only the process launcher: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_PORT are read from the environment); each rank owns its own 4096 instances
(weak scaling), the only exchange is the all-gather of the optimal controls U at the end of the
timed region - RCCL inside libmpc_amd.so (mpc_allgather_log).  No PyTorch anywhere: the solver, the
collective, the barrier and the device synchronisation are all the library's, through ctypes.

Timing.  The K steps from t = 0 (inputs resident in HBM) are bracketed by barrier + device sync on both
sides, maximum over the ranks.  One such region lasts a few milliseconds, so it is repeated R times
from the restored initial state (restore untimed) until about a second of GPU time has been spent;
`value` and `ms_per_step` are the MEDIAN region, `repeats` = R.  The W warm-up steps run first.

Prints ONE JSON line (rank 0).  `roofline` prices the fused closed-loop kernel against HBM with the
algorithmic bytes of SURVEY.md section 8d (4 968 B per instance-step); `traffic` is the measured HBM-side
traffic per launch, scaled from the per-instance-step figure of the committed PMC summary of this kernel.
`cpu_baseline` times oracle/mpc_oracle.c (a C port of the same algorithm, -O3 -march=native -fopenmp
- the reference's own CasADi/IPOPT path cannot run here) on all host cores and on one.
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

KERNEL_NAMES = {1: "loop_kernel (one instance per lane)", 2: "loop_kernel_tp (horizon-parallel)", 3: "loop_kernel_wv (wave-autonomous)"}
B_PER_GPU = 4096
SEED = 20250614
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
PROFILE_ROUND = "r05"      # the round whose PMC summaries (profiles/<round>_*pmc_summary.json) the traffic figures come from
PMC_SUMMARY = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_pmc_summary.json")


def alg_bytes_per_step(p) -> int:
    """Algorithmic HBM bytes per closed-loop step per instance, SURVEY.md section 8d / BASELINE.md section 3: state in
    (x_p, xhat, dhat, P, u_prev, set-points, previous target) and out (x_p, xhat, dhat, P, u, xs, us, ys) plus the primal
    warm start w the reference carries between steps (MPC_code.py:764), in and out.  CSTR: 920 + 4048 = 4968 B."""
    ne = p.nx + p.nd
    pk = ne * ne if p.estimator == "kal" else 0
    n_in = p.nxp + p.nx + p.nd + pk + p.nu + (p.ny + p.nu + p.nx) + (p.nx + p.nu)
    n_out = p.nxp + p.nx + p.nd + pk + p.nu + p.nx + p.nu + p.ny
    return 8 * (n_in + n_out + 2 * p.nw)


def carried_bytes_per_step(p) -> int:
    """What THIS solver would move per instance-step if nothing stayed on chip: the state as above plus its own warm starts
    (inputs and bound multipliers per stage, reduced target optimum) - reported next to the section-8d figure."""
    ne = p.nx + p.nd
    pk = ne * ne if p.estimator == "kal" else 0
    n_in = p.nxp + p.nx + p.nd + pk + p.nu + (p.ny + p.nu + p.nx) + (p.nx + p.nu)
    n_out = p.nxp + p.nx + p.nd + pk + p.nu + p.nx + p.nu + p.ny
    nbounded = p.nu + (p.nx if (np.isfinite(p.xmin).any() or np.isfinite(p.xmax).any() or p.y_bounded) else 0)
    return 8 * (n_in + n_out + 2 * p.N * (p.nu + 2 * nbounded) + 2 * (2 * p.nu + 3 * (p.nx + p.nu + p.ny)))


def measured_traffic(kernel_name: str, instance_steps_per_launch: float, summary: str = PMC_SUMMARY):
    """HBM-side bytes per launch from the committed rocprofv3 PMC summary (the counters cannot be read from inside this
    process): the summary stores bytes per instance-step of the named kernel; a launch of this run moves that times its
    instance-steps.  Returns (bytes, source)."""
    name = os.path.relpath(summary, ROOT)
    try:
        d = json.load(open(summary))
    except Exception:
        return None, name + " missing"
    per = d.get("hbm_bytes_per_instance_step")
    src = (f"{name}: kernel {d.get('kernel')}, {per:.0f} B per instance-step = (2*FETCH_SIZE + WRITE_SIZE) KiB of "
           f"separate rocprofv3 --pmc passes of `{d.get('command')}` / its instance-steps; scaled to this run's {instance_steps_per_launch:.0f} instance-steps per launch")
    if d.get("kernel_short") and d["kernel_short"] not in kernel_name:
        src += f" (NOTE: summary is for {d['kernel_short']}, this run used {kernel_name})"
    # a figure measured on other kernel sources than the ones this run was built from is STALE: said so in the line (roofline.traffic_stale)
    stale = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_summary
        now = pmc_summary.sources_sha16(d.get("kernel_short") or kernel_name)
        stale = (d.get("sources_sha16") != now) if d.get("sources_sha16") else None
        src += f"; kernel sources then {d.get('sources_sha16', 'not recorded')}, now {now}"
    except Exception as e:      # noqa: BLE001
        src += f"; source fingerprint unavailable ({e})"
    measured_traffic.stale = stale
    return per * instance_steps_per_launch, src


ISSUE_PEAK_WINST = 256 * 4 * 2.4e9 / 4.0      # wave-instructions per second the chip issues in fp64 (or unpacked 32-bit) VALU: 256 CUs x 4 SIMDs, 2.4 GHz, a wave64 instruction
                                               # takes four cycles on a 16-lane SIMD (/opt/skills/guides/MI355X_MICROARCH.md)


def issue_roofline(summary: str, launch_s: float, scale: float = 1.0, alg_flops_per_launch=None):
    """What 'bound: issue' is priced against: the vector instructions the priced kernel issues per launch (SQ_INSTS_VALU of the committed counter summary - the counters cannot be
    read from inside this process -, scaled to this run's launch size) over the LIVE launch time, against the rate at which the chip's SIMDs issue them."""
    try:
        d = json.load(open(summary))
        valu = float(d["SQ_INSTS_VALU"]["mean_per_launch"]) * scale
    except Exception:
        return None
    out = {"valu_insts_per_launch": valu, "achieved_winst_per_s": valu / launch_s, "peak_winst_per_s": ISSUE_PEAK_WINST, "frac": valu / launch_s / ISSUE_PEAK_WINST,
           "wait_fraction": d.get("wait_fraction"), "source": os.path.relpath(summary, ROOT) + ": SQ_INSTS_VALU per launch of " + str(d.get("kernel"))}
    if alg_flops_per_launch:      # of the 64 lane-slots of every vector instruction issued, the share an algorithmic multiply-add occupies (2 flops each): what the mapping wastes in lanes
        out["useful_lane_fraction"] = 0.5 * float(alg_flops_per_launch) / (valu * 64.0)
    return out


def cpu_quota():
    """What the host gives this process: hardware threads it may run on (affinity) and the cgroup's CPU quota in cores (None: unlimited) - a thread curve that peaks far below the
    hardware thread count is usually the quota, not the code."""
    q = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(path).read().split()
            if path.endswith("cpu.max"):
                q = None if t[0] == "max" else float(t[0]) / float(t[1])
            else:
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                q = None if float(t[0]) <= 0 else float(t[0]) / per
            break
        except Exception:
            continue
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count()
    return {"affinity_threads": aff, "cgroup_quota_cores": q}


def cpu_baseline(problem, x0, nsteps, target_seconds=8.0, max_seconds=25.0):
    """Time the oracle's C restatement on the host cores: the same closed loop (same instances, same steps, from
    t=0), on all cores and on one, each a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    oc = oracle_c.OracleC(problem, lib_path=oracle_c.build_fast())
    nthr = oc.max_threads()

    def run(nb, threads, budget):
        reps, spent = 0, 0.0
        while reps == 0 or (spent < budget and spent * (reps + 1) / reps < max_seconds):
            t0 = time.perf_counter(); oc.closed_loop(nsteps, x0[:nb], x0[:nb], logs=False, nthreads=threads); spent += time.perf_counter() - t0
            reps += 1
        return reps * nb * nsteps / spent, reps, spent

    t0 = time.perf_counter(); oc.closed_loop(min(nsteps, 10), x0[:256], x0[:256], logs=False, nthreads=1); t1 = time.perf_counter() - t0
    rate1 = 256 * min(nsteps, 10) / max(t1, 1e-6)                       # one core, rough: only to size the samples
    nb1 = int(min(len(x0), max(16, 4.0 * rate1 / nsteps)))              # about 4 s on one core
    v1, r1, s1 = run(nb1, 1, 4.0)
    # Every thread count gets a sample that keeps each of its threads busy for a good part of a second: the benchmark batch (4096 instances) is 16 instances per thread
    # on a 256-thread host - a parallel region of some ten milliseconds, which measures thread wake-up and the spread of the cold starts' iteration counts, not the
    # solver (round 4: 'peaks at 32 of 256 threads').  Instances are independent: the sample is the batch tiled.  Threads are pinned (OMP_PROC_BIND / OMP_PLACES, set
    # before the OpenMP runtime starts: main()).
    tried = {}
    for th in sorted({nthr, max(1, nthr // 2), max(1, nthr // 4), max(1, nthr // 8)}, reverse=True):
        nb_th = int(max(len(x0), min(128 * th, max_seconds * rate1 * th * 0.25 / nsteps)))
        xs_ = np.ascontiguousarray(np.tile(x0, (-(-nb_th // len(x0)), 1))[:nb_th])
        reps, spent = 0, 0.0
        while reps == 0 or (spent < target_seconds / 2.0 and spent * (reps + 1) / reps < max_seconds):
            t0 = time.perf_counter(); oc.closed_loop(nsteps, xs_, xs_, logs=False, nthreads=th); spent += time.perf_counter() - t0      # explicit: orc_closed_loop's thread count is sticky (omp_set_num_threads)
            reps += 1
        tried[th] = (reps * nb_th * nsteps / spent, reps, spent, nb_th)
    nbn = tried[max(tried, key=lambda th: tried[th][0])][3]
    best = max(tried, key=lambda th: tried[th][0])
    vn, rn, sn = tried[best][:3]
    return dict(value=vn, unit="steps/s", cores=best, kind="port", single_core_value=v1, threads_tried={str(th): tried[th][0] for th in tried}, host=cpu_quota(),
                sample=f"{nbn} instances x {nsteps} closed-loop steps from t=0 of the same workload, {rn} repetitions, {sn:.1f} s wall on {best} threads, the fastest of "
                       f"{sorted(tried)} tried ({nthr} hardware threads) "
                       f"(single core: {nb1} instances, {r1} repetitions, {s1:.1f} s): oracle/mpc_oracle.c, a C port of the same Riccati-PDIP with the "
                       f"same warm start, gcc -O3 -march=native -fopenmp built on this host; the reference's own CasADi/IPOPT path is not installable here")


def launch_ranks(args):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it (WORLD_SIZE unset): start N copies of this command, one per
    GPU, with RANK / LOCAL_RANK / WORLD_SIZE and an explicit rendezvous file for the RCCL id (mpc-code_amd/shard.py), wait for them and
    leave with the worst exit code - this process never touches the GPU.  Under a launcher (torch.distributed.run sets WORLD_SIZE) the
    ranks already exist: then --gpus must agree with it, a line that says n_gpus != --gpus is never printed."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks")
        return
    if args.gpus <= 1:
        return
    import subprocess
    import tempfile
    fd, rdzv = tempfile.mkstemp(prefix="mpc_amd_rccl_", suffix=".id"); os.close(fd); os.unlink(rdzv)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MPC_AMD_RDZV_FILE=rdzv, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    try:
        os.unlink(rdzv)
    except OSError:
        pass
    sys.exit(rc)


def dry_run(args):
    """The launch path without a GPU: the ranks meet through the same file rendezvous that carries the RCCL id
    (shard.exchange_unique_id), every rank leaves a marker, rank 0 reports the ranks it saw."""
    from mpc_code_amd import shard
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    uid = bytes(range(128))
    if world > 1:
        got = shard.exchange_unique_id((lambda: uid) if rank == 0 else None, rank, world, timeout=60.0)
        assert got == uid
        base = shard.rendezvous_path()
        open(f"{base}.rank{rank}", "w").write(os.environ.get("LOCAL_RANK", ""))
        if rank == 0:
            t0 = time.time()
            while not all(os.path.exists(f"{base}.rank{r}") for r in range(world)):
                if time.time() - t0 > 60.0:
                    sys.exit("dry run: not every rank arrived")
                time.sleep(0.01)
            seen = sorted(int(open(f"{base}.rank{r}").read()) for r in range(world))
            for r in range(world):
                os.unlink(f"{base}.rank{r}")
            os.unlink(base)
    else:
        seen = [0]
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "gpus_arg": args.gpus, "local_ranks_seen": seen}), flush=True)


def other_configs(args):
    """The other BASELINE workloads (configs[2], [3], [4]) measured by the SAME command, so that every headline number is the driver's: each runs as
    a child process of its own (its own per-model library and handle; the same K and W; the GPU is free - this process has not touched it yet) and
    prints its own full bench line, of which the compact part is kept under ``other_configs``.  A child that fails fails the whole run."""
    import subprocess
    out = {}
    for name in ("nmpc", "enmpc", "mhe"):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--min-seconds", str(args.min_seconds), "--cpu-seconds", str(min(args.cpu_seconds, 2.5))] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
        t0 = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            sys.stderr.write(r.stderr[-4000:])
            sys.exit("bench.py: --config %s failed (exit code %d)" % (name, r.returncode))
        d = json.loads(lines[-1])
        rf = d["roofline"]
        out[name] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                     "batch_per_gpu": d["config"]["batch_per_gpu"], "repeats": d["config"]["repeats"], "workload": d["config"]["workload"],
                     "roofline": {"bound": rf["bound"], "priced_against": rf.get("priced_against"), "kernel": rf["kernel"].split(" (")[0], "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"],
                                  "frac": rf["frac"], "traffic": rf["traffic"], "avg_launch_ms": rf["avg_launch_ms"], "alg_bytes_per_step": rf["alg_bytes_per_step"],
                                  "instance_steps_per_launch": rf["instance_steps_per_launch"], "fp64_frac": (rf.get("fp64") or {}).get("frac"),
                                  "issue_frac": (rf.get("issue") or {}).get("frac"), "useful_lane_fraction": (rf.get("issue") or {}).get("useful_lane_fraction"),
                                  "traffic_stale": rf.get("traffic_stale")},
                     "solver": d.get("solver"), "cpu_baseline": ({k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "single_core_value", "sample")} if "cpu_baseline" in d else None),
                     "command": " ".join(["python", "bench.py"] + cmd[2:]), "wall_s": time.perf_counter() - t0}
    return out


ENMPC_CONFIGS = {
    "enmpc": dict(over={"N": 40}, batch=16384, metric="closed-loop economic NMPC steps/sec over batch, Ex_ENMPC N=40 (BASELINE configs[3])",
                  what="BASELINE configs[3]: N = 40 (ships 25), N_mhe = 10; 131072 instances over 8 GPUs = 16384 per GPU"),
    "mhe": dict(over={"N_mhe": 20}, batch=4096, metric="closed-loop MHE + economic NMPC steps/sec over batch, Ex_ENMPC N_mhe=20 (BASELINE configs[4])",
                what="BASELINE configs[4]: N_mhe = 20 (ships 10), N = 25; 32768 instances over 8 GPUs = 4096 per GPU"),
}
FP64_PEAK_TFLOPS = 78.6      # MI355X vector fp64, public spec [ext]


def enmpc_alg(p, it_dyn, it_ss, it_mhe, nw_mean):
    """Algorithmic bytes and flops per instance-step of the economic loop: bytes = the state a step carries in
    and out (plant / model state, targets, the estimator's prior, window and covariance lists, the OCP's primal warm start
    MPC_code.py:764); flops = interior-point iterations x stages x Runge-Kutta stage evaluations x the operations of the generated
    second-order sensitivity code + the Riccati recursions, from the iteration counts of the run."""
    ne = p.nx + p.nd
    # the estimator reads its whole state - prior, window, the three covariance lists of the smoothing update - and writes what a step changes: the prior,
    # ONE new entry of the window and ONE new entry of each list (the lists shift in place on chip); target and OCP read and write their state
    mhe_fix = p.nxp + p.nx + p.nd + p.nu + ne + 2 * ne * ne + p.n_w + p.ny
    mhe_in = mhe_fix + p.N_mhe * (p.ny + p.nu) + p.N_mhe * 3 * ne * ne
    mhe_out = mhe_fix + (p.ny + p.nu) + 3 * ne * ne
    st_tgt = p.nd + 2 * (p.nx + p.nu)
    st_ocp = p.nxp + p.nx + p.nd + p.nu + 2 * (p.nx + p.nu) + p.nw                                                            # + primal warm start
    npo, npm = p.nx + p.nu, p.nx
    rk = lambda rows, cols: 4 * (2 * rows * (1 + cols + cols * (cols + 1) // 2) * 3 + 12 * rows * cols)      # accumulate K, dK, d2K + sparse chain rule, per RK step
    f_ocp = it_dyn * p.N * (p.quad_steps * rk(p.nx + 1, npo) + 2 * (7 * p.nx ** 3 // 3 + 4 * p.nx ** 2 * p.nu + 2 * p.nx * p.nu ** 2) + 60 * (p.nx + p.nu))
    f_ss = it_ss * (p.Mx * rk(p.nx, npo) + 2 * (p.nx + p.nu + p.ny) ** 3)
    # the estimator's stage matrices A = [[Phi_x, Bd], [0, I]], B = G_mhe have their zeros and ones compiled in (csrc/mpc_enmpc.hip MheStage):
    # a general entry costs a multiply-add (2), a one an add (1), a zero nothing
    wgt = lambda m_: float(np.sum(np.where(m_ == 0.0, 0, np.where(m_ == 1.0, 1, 2))))
    wA = 2.0 * p.nx * p.nx + wgt(np.asarray(p.Bd).reshape(p.nx, -1)) + p.nd
    wB = wgt(np.asarray(p.G_mhe))
    nw_ = p.n_w
    ric_mhe = (ne * (wA + wB) + 2 * ne * ne) + (nw_ * wB + ne * wB + wB) + (ne * wA + wA) + nw_ ** 3 + (2 * nw_ * nw_ * ne + 2 * nw_ * nw_) \
        + (2 * ne * ne * nw_ + 2 * ne * nw_) + (wA + wB) + (wA + wB + 2 * nw_ * ne)      # PA PB pc | Quu Qux qu | Qxx qx | inverse | gains | P p | gradients | forward
    f_mhe = it_mhe * nw_mean * (p.Mx * rk(p.nx, npm) + ric_mhe + 60 * (ne + p.n_w))
    b = [8 * (mhe_in + mhe_out), 16 * st_tgt, 16 * st_ocp]      # in and out, 8 bytes each
    return [(b[0], float(f_mhe)), (b[1], float(f_ss)), (b[2], float(f_ocp)), (sum(b), float(f_ocp + f_ss + f_mhe))]      # estimator, target, OCP kernels; all in one


def main_enmpc(args):
    """BASELINE configs[3] / [4]: the economic example Ex_ENMPC.py (continuous-time cost quadrature, moving-horizon estimator), every NLP
    of every step solved to its KKT point; instances sharded over the GPUs, one process each (weak scaling), the controls all-gathered
    through RCCL (the communicator of libmpc_amd.so) inside the timed region."""
    cfg = ENMPC_CONFIGS[args.config]
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import mpc_code_amd as m
    from mpc_code_amd import capi, enmpc, shard
    K, W = args.steps, args.warmup
    B = args.batch if args.batch != B_PER_GPU else cfg["batch"]
    p = m.load_problem(m.example_path("reactor_enmpc.py"), overrides=cfg["over"])
    s = enmpc.EnmpcSolver(p, device=local_rank)      # raises without the GPU: no CPU fallback
    comm = None
    if world > 1 or args.force_dist:                 # the job's communicator: RCCL inside this model's own library (enmpc_comm_*), on this rank's GPU
        comm = shard.RcclComm(s, rank, world)
    rng = np.random.default_rng(SEED)
    x0 = rng.uniform([0.5, 0.0], [1.0, 0.5], size=(B * world, 2))[rank * B:(rank + 1) * B]
    ns = max(K, W, 1)
    s.alloc(B, ns)
    s.set_groups(args.groups)
    kern = s.get_kernel()
    if W > 0:
        s.set_state(x0); s.run(0, W); s.sync()
    times, kms, pms, spent = [], [], [], 0.0
    while True:
        s.set_state(x0)                              # untimed: t = 0 again (cold OCP, empty estimation window)
        s.sync()
        if comm is not None:
            comm.barrier()
        t0 = time.perf_counter()
        s.run(0, K)
        if comm is not None:
            s.allgather_log("U", 0, K, to_host=False)      # every rank's controls on every rank: one RCCL all-gather, device log to device memory
            comm.barrier()
        else:
            s.sync()
        dt = time.perf_counter() - t0
        if comm is not None:
            dt = comm.max(dt)
        times.append(dt); kms.append(s.last_kernel_ms()); spent += dt
        if (args.repeats > 0 and len(times) >= args.repeats) or (args.repeats == 0 and (spent >= args.min_seconds or len(times) >= 2000)):
            break
    dt = float(np.median(times))
    if comm is not None:
        allU = s.allgather_log("U", 0, K)
        assert np.array_equal(allU[rank], s.get_log("U")[:K]), "all-gather of U corrupted the data"
    if kern == 2:
        # the split pipeline's launches one by one: passes of the same K steps with HIP events around every launch (which puts the batch on one
        # stream - the timed regions above run it in groups on streams of their own, where launches overlap and have no duration of their own)
        s.time_kernels(True)
        for _ in range(max(1, min(3, len(times)))):
            s.set_state(x0); s.run(0, K); s.sync()
            pms.append(s.phase_ms()[0])
        s.time_kernels(False)
    if rank == 0:
        st = {k: s.get_log(k)[:K] for k in ("STATUS_DYN", "STATUS_SS", "STATUS_MHE", "ITERS_DYN", "ITERS_SS", "ITERS_MHE")}
        nw_mean = float(np.mean([min(k + 1, p.N_mhe) for k in range(K)]))
        parts = enmpc_alg(p, float(st["ITERS_DYN"].mean()), float(st["ITERS_SS"].mean()), float(st["ITERS_MHE"].mean()), nw_mean)
        if kern == 2:      # the dominant kernel of the split pipeline: one launch = one phase of one step of every instance
            share = np.mean(pms, axis=0)
            j = int(np.argmax(share))
            kname = ("enmpc_mhe_kernel", "enmpc_target_kernel", "enmpc_ocp_kernel")[j]
            kdesc = ("%s (split pipeline: per step one launch for the estimator, one for the target - lane = instance - and one for OCP + plant; estimator and OCP with one "
                     "wave per instance, lane = stage, Riccati recursion over the lanes; device time by phase: estimator %.0f %%, target %.0f %%, OCP + plant %.0f %%)"
                     % (kname, *(100.0 * share / share.sum())))
            ab, fl = parts[j]
            per_launch_s, units, launches = float(share[j]) / K * 1e-3, B, K * len(pms)
        else:
            kname, kdesc = "enmpc_loop_kernel", "enmpc_loop_kernel (one wave = one instance, lane = stage; every phase of all K steps in one launch)"
            ab, fl = parts[3]
            per_launch_s, units, launches = float(np.mean(kms)) * 1e-3, B * K, len(times)
        achieved = ab * units / per_launch_s / 1e9
        tflops = fl * units / per_launch_s / 1e12
        traffic, traffic_src = measured_traffic(kname, units, os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.json" % (PROFILE_ROUND, args.config)))
        out = {"metric": cfg["metric"], "value": B * world * K / dt, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "Ex_ENMPC (isothermal reactor, nx=2,nu=1,ny=2,nd=2; continuous-time economic cost integrated over every shooting interval, "
                                      "moving-horizon estimator with smoothing update), %s, x0_p~U([0.5,1]x[0,0.5]) seed %d, x0_m=[1.2,0.5], closed loop from t=0: MHE NLP + "
                                      "target NLP + OCP NLP (each to its KKT point, tol 1e-8 / 1e-10) + plant per step" % (cfg["what"], SEED),
                          "batch_per_gpu": B, "horizon": p.N, "mhe_horizon": p.N_mhe, "quad_steps": p.quad_steps, "steps_per_run": K, "kernel": kern, "stream_groups": args.groups, "repeats": len(times),
                          "parallelism": "instances sharded over %d GPU(s), one process each; RCCL all-gather of U inside the timed region" % world,
                          "rccl_ranks": (s.comm_rank()[1] if comm is not None else 1),
                          "timed_region_ms": {"median": dt * 1e3, "min": float(np.min(times)) * 1e3, "max": float(np.max(times)) * 1e3}},
               "roofline": {"bound": "issue", "priced_against": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": getattr(measured_traffic, "stale", None),
                            "kernel": kdesc, "launches": launches, "avg_launch_ms": per_launch_s * 1e3,
                            "issue": issue_roofline(os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.json" % (PROFILE_ROUND, args.config)), per_launch_s, units / float(cfg["batch"]), alg_flops_per_launch=fl * units),
                            "launch_timing": ("HIP events around every launch of %d separate passes of the same %d steps on one stream (the timed regions run the batch in "
                                              "groups on streams of their own)" % (len(pms), K)) if kern == 2 else "HIP events around the timed regions' launches",
                            "alg_bytes_per_step": ab, "instance_steps_per_launch": units, "device_ms_per_run": float(np.mean(kms)),
                            "fp64": {"achieved_tflops": tflops, "peak_tflops": FP64_PEAK_TFLOPS, "frac": tflops / FP64_PEAK_TFLOPS, "alg_flops_per_step": fl},
                            "note": "the path is bound by fp64 vector issue and dependent-instruction latency (Runge-Kutta sensitivities, Riccati recursion over the lanes), "
                                    "not by HBM: 'achieved' prices the state the priced kernel carries in and out per instance-step (DESIGN.md section 10), 'fp64' the "
                                    "algorithmic flops of the run's iteration counts against the vector fp64 peak"},
               "solver": {"frac_solved_dyn": float((st["STATUS_DYN"] == 0).mean()), "frac_solved_ss": float((st["STATUS_SS"] == 0).mean()), "frac_solved_mhe": float((st["STATUS_MHE"] == 0).mean()),
                          "mean_iters_dyn": float(st["ITERS_DYN"].mean()), "mean_iters_ss": float(st["ITERS_SS"].mean()), "mean_iters_mhe": float(st["ITERS_MHE"].mean())}}
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import enmpc_oracle as eo
            import enmpc_oracle_c as eoc
            oc = eoc.OracleEC(eo.load_problem(m.example_path("reactor_enmpc.py"), overrides=cfg["over"]), fast=True)      # -O3 -march=native, built on this host
            nthr = oc.max_threads()
            t0 = time.perf_counter(); oc.closed_loop(K, x0[:2], nthreads=1, logs=False); r1 = 2 * K / (time.perf_counter() - t0)      # one core, to size the samples
            tried, est = {}, None
            for th in sorted({nthr, max(1, nthr // 2), max(1, nthr // 4), max(1, nthr // 8)}):      # ascending: each sample sized by the best rate measured so far
                est = 0.6 * r1 * th if est is None else max(v[0] for v in tried.values())
                nb = int(min(B, max(th, args.cpu_seconds * est / K)))       # about --cpu-seconds (5 s) per thread count
                t0 = time.perf_counter(); oc.closed_loop(K, x0[:nb], nthreads=th, logs=False); tried[th] = (nb * K / (time.perf_counter() - t0), nb)
            best = max(tried, key=lambda th: tried[th][0])
            out["cpu_baseline"] = {"value": tried[best][0], "unit": "steps/s", "cores": best, "kind": "port", "single_core_value": r1,
                                   "threads_tried": {str(th): tried[th][0] for th in tried}, "host": cpu_quota(),
                                   "sample": "%d instances x %d closed-loop steps from t=0 of the same workload on %d threads, the fastest of %s tried (%d hardware threads): "
                                             "oracle/enmpc_oracle.c - the same three NLPs per step solved by the same outer interior point method with complex-step "
                                             "derivatives and DENSE null-space (QR + Cholesky) Newton steps, gcc -O3 -march=native -fopenmp built on this host - the CHECKER, which makes no use of the "
                                             "stage structure: a structure-exploiting host build (analytic sensitivities in place of complex steps: a third of the flops; Riccati solves in place of the dense null-space "
                                             "method: none of its n^3) is estimated five to ten times faster - the Runge-Kutta second-order sensitivities dominate either way; the GPU / CPU ratio of this line is no statement about either; "
                                             "the reference's CasADi/IPOPT/IDAS path is not installable here" % (tried[best][1], K, best, sorted(tried), nthr)}
        import ctypes
        sys.stdout.flush(); ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
    s.close()


def main_nmpc(args):
    """BASELINE configs[3]: the non-linear CSTR of Ex_NMPC.py with N = 30, EKF, batch 16384 (SURVEY.md 8d cfg 3), one GPU."""
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        sys.exit("bench.py --config nmpc runs on one GPU (instances are independent: launch one process per GPU for more)")
    import mpc_code_amd as m
    from mpc_code_amd import nmpc
    K, W = args.steps, args.warmup
    B = args.batch if args.batch != B_PER_GPU else 16384
    p = m.load_problem(m.example_path("cstr_nmpc.py"))
    s = nmpc.NmpcSolver(p, device=int(os.environ.get("LOCAL_RANK", "0")))      # raises without the GPU: no CPU fallback
    rng = np.random.default_rng(SEED)
    x0 = p.x0_p * (1.0 + 0.02 * rng.uniform(-1.0, 1.0, size=(B, 3)))
    ns = max(K, W, 1)
    s.alloc(B, ns); s.set_schedule(p.schedules(ns))
    s.set_state(x0, x0)
    kern = s.get_kernel()
    split = kern == 4
    s.set_groups(args.groups)
    if W > 0:
        s.run(0, W, args.max_sqp); s.sync()
    times, kms, wms, wn, spent = [], [], [], 0, 0.0
    while True:
        s.set_state(x0, x0)                      # untimed: t = 0 again
        s.sync()
        t0 = time.perf_counter()
        s.run(0, K, args.max_sqp); s.sync()
        dt = time.perf_counter() - t0
        times.append(dt); kms.append(s.last_kernel_ms()); spent += dt
        if (args.repeats > 0 and len(times) >= args.repeats) or (args.repeats == 0 and (spent >= args.min_seconds or len(times) >= 2000)):
            break
    dt = float(np.median(times))
    kms1 = kms
    if split:
        # the wave-style launches one by one: separate passes of the same K steps, every wave-style launch bracketed by its own HIP events (which puts the
        # batch on one stream - the timed regions above run it in groups on streams of their own, where launches overlap and have no duration of their own)
        s.time_kernels(True); kms1 = []
        for _ in range(max(1, min(3, len(times)))):
            s.set_state(x0, x0); s.run(0, K, args.max_sqp); s.sync()
            ms, nl = s.wave_kernel_ms(); wms.append(ms / max(nl, 1)); wn = nl; kms1.append(s.last_kernel_ms())
        s.time_kernels(False)
    st, sqp, it = s.get_log("STATUS_DYN")[:K], s.get_log("SQP_DYN")[:K], s.get_log("ITERS_DYN")[:K]
    ne = p.nx + p.nd
    state = p.nxp + p.nx + p.nd + ne * ne + p.nu + p.nx + p.nu
    ab = (2 * state + 2 * p.nw + p.ny + p.nu) * 8              # state in + out, shifted trajectory in + out, set points
    summary = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_nmpc_pmc_summary.json")
    if split:
        # the dominant kernel of the split pipeline: one launch = linearisation + QP of one step of every instance
        per_launch_s = float(np.mean(wms)) * 1e-3
        units = B
        kname = "nmpc_loop_kernel_wv<true>"
        kdesc = ("nmpc_loop_kernel_wv<true> (split pipeline, the wave-style launch of a step: one wave owns four instances, lane = stage - two "
                 "instances side by side for N <= 32 -, RK4 sensitivities into LDS, QP on the matrix cores; %.0f %% of the device time of a "
                 "run, the lane-style launch nmpc_step_lane_kernel - estimator, target, plant - is the rest)" % (100.0 * np.mean(wms) * wn / np.mean(kms1)))
        note = ("algorithmic bytes: resident state in and out + the shifted trajectory in and out + set points, per instance-step; one launch of "
                "this kernel advances every instance by one step.  The kernel is bound by dependent fp64 chains (RK4 sensitivities, Riccati "
                "recursions on one wave per SIMD), not by HBM: its measured traffic is the trajectories / multipliers (warm start rows) it loads "
                "and stores per launch and the exchange buffers")
    else:
        per_launch_s = float(np.mean(kms)) * 1e-3
        units = B * K
        kname = "nmpc_loop_kernel"
        kdesc = "nmpc_loop_kernel (one instance per lane, all steps in one launch; helper waves share the stage linearisations up to one workgroup per CU)"
        note = ("algorithmic bytes: resident state in and out + the shifted trajectory in and out + set points.  Measured traffic is two "
                "orders above them: this kernel is the instance-per-lane design, its Riccati workspace (20 KB per instance) and the "
                "linearisation slab (7 KB) stream through HBM in every sweep")
    achieved = ab * units / per_launch_s / 1e9
    traffic, traffic_src = measured_traffic(kname, units, summary)
    out = {"metric": "closed-loop NMPC steps/sec over batch, Ex_NMPC N=30 (BASELINE configs[3])", "value": B * K / dt, "unit": "steps/s", "n_gpus": 1,
           "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": "Ex_NMPC (nx=3,nu=2,ny=2,nd=2, RK4 Mx=10, EKF), N=30, batch=%d, x0=[0.874317,325,0.6528]*(1+0.02*U(-1,1)^3) seed %d, closed loop "
                                  "from t=0: EKF + target SQP + %s + plant per step" % (B, SEED, "one real-time SQP iteration" if args.max_sqp == 1 else "SQP (<= %d iterations)" % args.max_sqp),
                      "batch_per_gpu": B, "horizon": p.N, "steps_per_run": K, "kernel": kern, "stream_groups": args.groups, "max_sqp": args.max_sqp, "repeats": len(times),
                      "timed_region_ms": {"median": dt * 1e3, "min": float(np.min(times)) * 1e3, "max": float(np.max(times)) * 1e3},
                      "device_ms_per_run": float(np.mean(kms))},
           "roofline": {"bound": "issue", "priced_against": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": getattr(measured_traffic, "stale", None),
                        "kernel": kdesc, "launches": (wn * len(wms)) if split else len(times), "avg_launch_ms": per_launch_s * 1e3, "alg_bytes_per_step": ab,
                        "issue": issue_roofline(summary, per_launch_s, units / 16384.0) if split else None,
                        "launch_timing": ("HIP events around every wave-style launch of %d separate passes of the same %d steps on one stream (the timed regions run the batch in "
                                          "groups on streams of their own)" % (len(wms), K)) if split else "HIP events around the timed regions' launches",
                        "instance_steps_per_launch": units, "note": note},
           "solver": {"frac_solved": float((st == 0).mean()), "frac_maxiter": float((st == 1).mean()), "frac_infeasible_hold": float((st == 2).mean()),
                      "mean_sqp": float(sqp.mean()), "mean_ipm_iters_last_qp": float(it.mean())}}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import warnings
        import nmpc_oracle as no
        import nmpc_oracle_c as noc
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            oc = noc.OracleNC(no.load_problem(m.example_path("cstr_nmpc.py")), fast=True)      # the checker's own reading of the example; -O3 -march=native, built on this host
        nthr = oc.max_threads()
        t0 = time.perf_counter(); oc.closed_loop(K, x0[:2], max_sqp=args.max_sqp, nthreads=1, logs=False); r1 = 2 * K / (time.perf_counter() - t0)
        tried, est = {}, None
        for th in sorted({nthr, max(1, nthr // 2), max(1, nthr // 4), max(1, nthr // 8)}):      # ascending: each sample sized by the best rate measured so far
            est = 0.6 * r1 * th if est is None else max(v[0] for v in tried.values())
            nb = int(min(B, max(th, args.cpu_seconds * est / K)))
            t0 = time.perf_counter(); oc.closed_loop(K, x0[:nb], max_sqp=args.max_sqp, nthreads=th, logs=False); tried[th] = (nb * K / (time.perf_counter() - t0), nb)
        best = max(tried, key=lambda th: tried[th][0])
        out["cpu_baseline"] = {"value": tried[best][0], "unit": "steps/s", "cores": best, "kind": "port", "single_core_value": r1, "threads_tried": {str(th): tried[th][0] for th in tried},
                               "sample": "%d instances x %d closed-loop steps from t=0 of the same workload on %d threads, the fastest of %s tried (%d hardware threads): "
                                         "oracle/nmpc_oracle.c - EKF, target SQP and one SQP iteration of the OCP per step with complex-step Jacobians and dense null-space "
                                         "(QR + Cholesky) interior-point QP solves, gcc -O3 -march=native -fopenmp built on this host; the reference's CasADi/IPOPT path is "
                                         "not installable here" % (tried[best][1], K, best, sorted(tried), nthr)}
    print(json.dumps(out), flush=True)
    s.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="instances per GPU (default: BASELINE configs[1])")
    ap.add_argument("--total-batch", type=int, default=0, help="lmpc: instances of the WHOLE job, split evenly over the GPUs (strong scaling; BASELINE.md section 3 / the target: 65536 over 8 GPUs); "
                    "0: --batch instances per GPU (weak scaling, the default)")
    ap.add_argument("--steps-per-launch", type=int, default=0, help="closed-loop steps per kernel launch (0: library default)")
    ap.add_argument("--loop-kernel", type=int, default=0, help="0: library default, 1: instance per lane, 2: horizon-parallel, 3: wave-autonomous")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions (0: as many as fill --min-seconds)")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="GPU time to spend in timed regions (of this workload and of each of the other configs' child runs: "
                                                                   "twelve seconds of busy GPU in the default run, which a utilisation sampler can see between the CPU baselines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="create the RCCL communicator even for one rank: exercises the N>1 code path on a 1-GPU box")
    ap.add_argument("--config", default="lmpc", choices=["lmpc", "nmpc", "enmpc", "mhe"], help="lmpc: the metric workload (BASELINE configs[1]); nmpc: "
                    "configs[2], Ex_NMPC N=30, batch 16384, one real-time SQP iteration per step, one GPU; enmpc: configs[3], Ex_ENMPC N=40, batch "
                    "16384 per GPU (131072 over 8); mhe: configs[4], Ex_ENMPC with N_mhe=20, batch 4096 per GPU (32768 over 8)")
    ap.add_argument("--max-sqp", type=int, default=1, help="nmpc: SQP iterations per OCP (1 = real-time iteration)")
    ap.add_argument("--groups", type=int, default=0, help="nmpc / enmpc / mhe: groups of the batch on HIP streams of their own (0: the library's choice; 1: one stream, "
                    "as the profiles need it - overlapping launches have no duration of their own)")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="nmpc / enmpc / mhe: host seconds per thread count of the CPU baseline's bounded sample")
    ap.add_argument("--no-other-configs", action="store_true", help="lmpc on one GPU: do not measure BASELINE configs[2..4] next to the metric workload")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: every rank goes through the rendezvous only and rank 0 prints who was there (launcher test)")
    args = ap.parse_args()
    launch_ranks(args)                # --gpus N without a launcher: N child processes, one per GPU (before anything here touches the GPU)
    if args.dry_run:
        return dry_run(args)
    if args.config == "nmpc":
        return main_nmpc(args)
    if args.config in ("enmpc", "mhe"):
        return main_enmpc(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    K, W, B = args.steps, args.warmup, args.batch
    strong = args.total_batch > 0
    if strong:
        if args.total_batch % world:
            sys.exit(f"bench.py: --total-batch {args.total_batch} does not split evenly over {world} GPUs")
        B = args.total_batch // world
    use_dist = world > 1 or args.force_dist
    others = None
    if world == 1 and not args.force_dist and not args.no_other_configs and B == B_PER_GPU and not strong:
        others = other_configs(args)      # child processes, before this one touches the GPU

    import mpc_code_amd as m
    from mpc_code_amd import capi, shard
    prob = m.load_problem(m.example_path("cstr_lmpc.py"))
    if local_rank == 0:
        capi.build_library()          # no-op when the in-tree .so is current; one rank only (no concurrent hipcc)
    else:
        t0 = time.time()
        while not os.path.exists(capi.LIB_PATH) and time.time() - t0 < 900:
            time.sleep(0.5)
    solver = capi.Solver(prob, device=local_rank)      # raises if the HIP library or the GPU is missing: no CPU fallback
    comm = shard.RcclComm(solver, rank, world) if use_dist else None
    if args.steps_per_launch > 0:
        solver.set_option("steps_per_launch", args.steps_per_launch)
    solver.set_option("loop_kernel", args.loop_kernel)

    # synthetic initial states: the same generator for the whole job, each rank takes its block
    rng = np.random.default_rng(SEED)
    x0_all = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B * world, 3))
    x0 = x0_all[rank * B:(rank + 1) * B]
    nsched = max(K, W, 1)
    solver.loop_alloc(B, nsched, capi.LOG_U)
    solver.loop_set_schedule(prob.schedules(nsched))

    def barrier():
        solver.comm_barrier()          # every rank's stream drained (+ one all-reduce when there is a communicator)

    # warm-up (untimed), then restore the initial state
    solver.loop_set_state(x0, x0)
    if W > 0:
        solver.loop_run(0, W)
        solver.allgather_log("U", 0, W, to_host=False)
        solver.loop_sync()
    times, kernel_ms, n_launch = [], [], 1
    gpu_spent = 0.0
    while True:
        solver.loop_set_state(x0, x0)       # untimed: t = 0 again
        barrier()
        t0 = time.perf_counter()
        solver.loop_run(0, K)
        solver.allgather_log("U", 0, K, to_host=False)      # all ranks' controls, device to device (RCCL for N > 1)
        barrier()
        dt = time.perf_counter() - t0
        if comm is not None:
            dt = comm.max(dt)               # the slowest rank's region: every rank takes the same decision below
        times.append(dt)
        ms, n_launch = solver.last_kernel_ms()
        kernel_ms.append(ms)
        gpu_spent += dt
        if (args.repeats > 0 and len(times) >= args.repeats) or (args.repeats == 0 and (gpu_spent >= args.min_seconds or len(times) >= 2000)):
            break
    dt = float(np.median(times))
    loop_kernel = int(solver.get_option("loop_kernel"))

    if rank == 0:
        st = solver.loop_get_log("STATUS_DYN")[:K]
        it = solver.loop_get_log("ITERS_DYN")[:K]
        steps_total = B * world * K
        per_launch_s = float(np.mean(kernel_ms)) * 1e-3 / max(n_launch, 1)
        inst_steps_per_launch = B * K / max(n_launch, 1)
        ab = alg_bytes_per_step(prob)
        achieved = ab * inst_steps_per_launch / per_launch_s / 1e9
        traffic, traffic_src = measured_traffic(KERNEL_NAMES[loop_kernel], inst_steps_per_launch)
        out = {
            "metric": "closed-loop MPC steps/sec over batch, LMPC-CSTR N=50",
            "value": steps_total / dt, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Ex_LMPC_CSTR (nx=3,nu=2,ny=3,nd=3), N=50, batch=%d per GPU, x0~U([-0.5,0.5]x[-8,8]x[-5,5]) seed %d, "
                                   "closed loop from t=0: Kalman filter + target QP + OCP (Riccati-PDIP) + plant per step" % (B, SEED),
                       "batch_per_gpu": B, "total_batch": B * world, "horizon": prob.N, "steps_per_launch": K / max(n_launch, 1), "loop_kernel": KERNEL_NAMES[loop_kernel],
                       "parallelism": "instances sharded over %d GPU(s), one process each; RCCL all-gather of U (mpc_allgather_log) inside the timed region" % world,
                       "rccl_ranks": (solver.comm_rank()[1] if use_dist else 1),
                       "repeats": len(times), "timed_region_ms": {"median": dt * 1e3, "min": float(np.min(times)) * 1e3, "max": float(np.max(times)) * 1e3}},
            "roofline": {"bound": "issue", "priced_against": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": getattr(measured_traffic, "stale", None),
                         "kernel": KERNEL_NAMES[loop_kernel], "launches": n_launch,
                         "avg_launch_ms": per_launch_s * 1e3, "alg_bytes_per_step": ab, "carried_bytes_per_step": carried_bytes_per_step(prob),
                         "issue_bound": True,
                         "issue": issue_roofline(PMC_SUMMARY, per_launch_s, inst_steps_per_launch / 81920.0, alg_flops_per_launch=31.0e3 * float(it.mean()) * inst_steps_per_launch),
                         "fp64": (lambda fl: {"achieved_tflops": fl * inst_steps_per_launch / per_launch_s / 1e12, "peak_tflops": FP64_PEAK_TFLOPS,
                                              "frac": fl * inst_steps_per_launch / per_launch_s / 1e12 / FP64_PEAK_TFLOPS, "alg_flops_per_step": fl,
                                              "formula": "BASELINE.md section 3 / SURVEY.md 8d: 31 kflop per interior-point iteration x mean iterations per instance-step of this run"})(
                                 31.0e3 * float(it.mean())),
                         "note": "HIP events on the handle's stream around each timed region's launches, mean over the repeats. The path is bound by dependent "
                                 "fp64 issue of one wave per SIMD (Riccati recursion on the matrix cores, four instances per wave), not by HBM "
                                 "(SURVEY.md 8d): iterates and loop state stay in registers / LDS for a whole launch"},
            "solver": {"mean_iters": float(it[st != 2].mean()) if (st != 2).any() else None, "max_iters": int(it.max()),
                       "frac_solved": float((st == 0).mean()), "frac_maxiter": float((st == 1).mean()),
                       "frac_infeasible_hold": float((st == 2).mean())},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, x0, K)
        if others is not None:
            out["other_configs"] = others
        line = json.dumps(out)
    if use_dist:      # the gathered block of this rank must be what the kernel logged
        allU = solver.allgather_log("U", 0, K)
        assert np.array_equal(allU[rank], solver.loop_get_log("U")[:K]), "all-gather of U corrupted the data"
        comm.barrier()
    solver.close()
    if rank == 0:
        # the one JSON line is the last thing on stdout: whatever a native library left in C's stdio buffer goes out first
        import ctypes
        sys.stdout.flush(); ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == "__main__":
    # the CPU baseline's threads stay where they start (before any OpenMP runtime is loaded): without it a 256-thread host migrates them between the samples
    os.environ.setdefault("OMP_PROC_BIND", "spread"); os.environ.setdefault("OMP_PLACES", "threads")
    main()
