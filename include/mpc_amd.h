/*
 * mpc_amd.h - C-ABI of the MI355X-native batched linear MPC solver (libmpc_amd.so).
 *
 * Drop-in boundary for the hot path of CPCLAB-UNIPI/MPC-code.  The reference has no FFI; its only
 * seam is the CasADi nlpsol function object.  Each entry point below names the
 * reference call it replaces.  All reference line numbers are in /root/reference.
 *
 * Conventions
 *   - every array is caller-owned, contiguous float64 (int32 for status / iteration words);
 *   - host arrays are batch-major [B][k] (one row per instance), matrices row-major;
 *   - absent bounds are +-INFINITY; optional pointers may be NULL where stated;
 *   - every function returns 0 on success and a negative code on error, mpc_last_error() then holds a
 *     message (thread-local).  Nothing falls back to a CPU path: without a usable GPU the create call fails.
 *   - status words: 0 solved, 1 iteration limit (accepted by the reference, MPC_code.py:714,786),
 *     2 infeasible (== IPOPT 'Infeasible_Problem_Detected': the driver holds u / the target).
 *   - one host thread drives a handle; calls are blocking unless stated (mpc_loop_run is asynchronous).
 */
#ifndef MPC_AMD_H
#define MPC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_STATUS_SOLVED 0
#define MPC_STATUS_MAXITER 1
#define MPC_STATUS_INFEASIBLE 2

#define MPC_EST_NONE 0
#define MPC_EST_KALMAN 1 /* time-varying Kalman filter, Estimator.py:263-311 */
#define MPC_EST_FIXED_GAIN 2 /* steady-state KF / Luenberger, Estimator.py:231-261 */

typedef struct mpc_handle mpc_handle;

/*
 * Numeric problem: what the reference builds symbolically once per run -
 * defF_model (Utilities.py:102-245), defF_p (:21-100), defFss_obj (:267-321), defF_obj (:323-381),
 * defVfin (:383-420; P is the DARE solution the caller computes with the same SciPy call, :409),
 * opt_ss (Target_Calc.py:20-161), opt_dyn (Control_Calc.py:20-260), Kkalss (Estimator.py:103-229).
 */
typedef struct mpc_lin_desc {
    int32_t nx, nu, ny, nd, nxp; /* MPC_code.py:31-35 */
    int32_t N;                   /* horizon */
    int32_t du_form;             /* cost on u_k - u_{k-1} with weight R (the Ex-file's S), Control_Calc.py:163-166,180-181 */
    int32_t duss_form;           /* target cost on us - us_prev, Target_Calc.py:121-122 */
    int32_t y_bounded;           /* the g1 rows exist (yFree False), Control_Calc.py:60-63,150-151.  A bounded row of C with one
                                    non-zero entry is a box on that state; any other bounded row is carried as one more stage
                                    state (the compiled kernel set must have that many, see mpc_build_info) */
    int32_t estimator;           /* MPC_EST_* */
    int32_t max_iter;            /* ipopt.max_iter = Sol_itmax, MPC_code.py:262-263 */
    int32_t device;              /* HIP device ordinal */
    /* model x+ = A x + B u + Bd d + fx_const, y = C x + Cd d + fy_const (Utilities.py:135-155,208-244) */
    const double *A, *B, *C, *Bd, *Cd, *fx_const, *fy_const;
    /* plant xp+ = Ap xp + Bp u + pxp, y = Cp xp + pyp (Utilities.py:45-49,88-91) */
    const double *Ap, *Bp, *Cp;
    /* weights: stage Q [nx,nx], R [nu,nu], terminal P [nx,nx]; target Qss [ny,ny], Rss [nu,nu] */
    const double *Q, *R, *P, *Qss, *Rss;
    /* bounds of the dynamic problem (Control_Calc.py:213-252) and of the target problem (Target_Calc.py:127-134) */
    const double *umin, *umax, *xmin, *xmax, *ymin, *ymax;
    const double *umin_ss, *umax_ss, *xmin_ss, *xmax_ss, *ymin_ss, *ymax_ss;
    const double *dmin, *dmax; /* saturation of the disturbance estimate, MPC_code.py:660-665; NULL = none */
    /* estimator data: Q_kf,R_kf [nx+nd]^2,[ny]^2 for MPC_EST_KALMAN; K [nx+nd,ny] for MPC_EST_FIXED_GAIN */
    const double *Q_kf, *R_kf, *K;
    /* bounds on u_k - u_{k-1} (k = 0: u_0 - u_prev), the g2 rows of opt_dyn, Control_Calc.py:163-169,241-243; NULL = none.  The
     * problem then runs in the stage form with input v = u_k - u_{k-1} and state [x; u_prev] (kernel set du = 1) */
    const double *Dumin, *Dumax;
    /* terminal equality x_N = xs of opt_dyn (TermCons, Control_Calc.py:197-198); 0 = none.  P is ignored then (the terminal cost is
     * zero on the constraint).  Exact (x_N = xs to rounding) on the instance-per-lane and the wave-autonomous kernels; the horizon-parallel
     * kernel ("loop_kernel" = 2) carries it by a terminal weight alone: a miss of |multiplier| / 1e12 */
    int32_t term_cons;
    /* the simulated process is a user function (User_fxp_Cont, MPC_code.py:176-199) compiled into the library from the traced
     * Ex-file function (one library per plant; capi.Solver builds it); Ap, Bp are ignored then */
    int32_t nl_plant;
    double h_sample;             /* sampling interval h (the time a user plant integrates over); 0 = 1 */
    /* soft output constraints (`slacks = True`, Control_Calc.py:39-40,186-192,228-239; Default_Values.py:128): ONE slack vector Sl = [sl_ub (ny); sl_lb (ny)] >= 0
     * shared by all stages widens every stage's output rows, ymin - sl_lb <= y_k <= ymax + sl_ub (k = 0..N-1), and is penalised Sl' Ws Sl in every stage's cost;
     * Ws [2 ny][2 ny] (MPC_code.py:55-57).  Input and state bounds stay hard.  Solved by mpc_ocp_solve on the arrowhead solver (csrc/mpc_soft.hpp: one Riccati
     * factorisation with 1 + 2 ny right-hand sides, a dense Schur complement for Sl); the optimal slacks of the last call: mpc_get_slacks.  The resident loop
     * (mpc_loop_run) of such a problem is the instance-per-lane loop with this solver as its OCP; its log "SL" [step][B][2 ny] holds the last accepted slack vector
     * of every step (MPC_code.py:800, 808-809). */
    int32_t slacks;
    const double *Ws;
    /* Affine user inequality rows of the OCP (User_g_ineq, Control_Calc.py:94-100,132-147; MPC_code.py:306-314): for k = 0..N-1
     *     Gx x_k + Gu u_k + Gd dhat + g0 <= 0        (Gx [n][nx], Gu [n][nu], Gd [n][nd], g0 [n], row-major; y_k = C x_k + Cd dhat + fy_const substituted by the caller)
     * Each row is carried as one more stage state w_{k+1} = Gx x_k + Gu u_k + const with the box (-inf, 0] at k = 1..N (the reference's solver gives such a row a slack
     * variable of its own).  n_user_rows = 0: none.  Not together with slacks, term_cons or px / py. */
    int32_t n_user_rows;
    const double *Gx, *Gu, *Gd, *g0;
} mpc_lin_desc;

/* Replaces the construction nlpsol('solver','ipopt',...) of Control_Calc.py:256-258 and
 * Target_Calc.py:157-159 (and their bound vectors, :260 / :161).  Fails if no gfx950 device is usable,
 * if the dimensions have no compiled kernel, or if [A-I, B] is rank deficient. */
int mpc_lin_create(const mpc_lin_desc *desc, mpc_handle **out);
void mpc_destroy(mpc_handle *h);
const char *mpc_last_error(void);

/*
 * One solver(lbx,ubx,x0,p,lbg,ubg) call per instance, MPC_code.py:776-781, with the driver's glue:
 * x_0 fixed to xhat (:734), parameters par[0:13] = xhat,xs,us,dhat,u_prev (:772, Control_Calc.py:44-48),
 * read-out u* = w[nx:nx+nu], xhat+ = w[nx+nu:2nx+nu] (:798-799).
 *   px, py    time-varying model parameters over the horizon, [B][N][nx] / [B][N][ny] (def_px / def_py evaluated by the driver,
 *             MPC_code.py:492-497; Control_Calc.py:43-57,161,150): x_{k+1} = A x_k + B u_k + Bd d + px_k, y_k = C x_k + Cd d + py_k.
 *             NULL = zero.  Such a call runs on the instance-per-lane solver with per-stage data and starts cold
 *   w_inout   optional [B][nx*(N+1)+nu*N]: on return the primal optimum in opt_dyn's order [x0,u0,...,xN] (Control_Calc.py:31-37).
 *             With mpc_set_option("ocp_warm_start", 1) it is also read: the caller's guess x0= of the reference call (the shifted
 *             previous optimum, MPC_code.py:740-764) supplies the inputs of the starting point; the bound multipliers of the previous
 *             mpc_ocp_solve call of the same batch stay in the handle and are shifted by one stage whenever the
 *             problem data moved little.  A first element that is not finite means "no guess".  Default: cold start, content ignored.
 *   u_out [B][nu], xnext_out [B][nx]: untouched for instances with status 2
 *   kkt_res   optional [B][3]: stationarity, bound residual, mean complementarity at the returned point
 */
int mpc_ocp_solve(mpc_handle *h, int32_t B, const double *xhat, const double *xs, const double *us,
                  const double *dhat, const double *u_prev, const double *px, const double *py,
                  double *w_inout, double *u_out, double *xnext_out, int32_t *status, int32_t *iters,
                  double *kkt_res);

/* The optimal slack vectors Sl = w_opt[nw-ns:nw] of the last mpc_ocp_solve call of a problem with soft constraints (MPC_code.py:800), [B][2 ny] */
int mpc_get_slacks(mpc_handle *h, int32_t B, double *sl_out);

/* One solver_ss(...) call per instance, MPC_code.py:693-718 (par_ss = usp,ysp,xsp,dhat,us_prev, :693). */
int mpc_target_solve(mpc_handle *h, int32_t B, const double *usp, const double *ysp, const double *xsp,
                     const double *dhat, const double *us_prev, double *xs, double *us, double *ys,
                     int32_t *status, int32_t *iters);

/* This step's model parameters p_x_k = px[:,0], p_y_k = py[:,0] (MPC_code.py:500-501) for the following mpc_kf_update (predicted output,
 * :524) and mpc_target_solve (par_ss, :693) calls of a batch of B: [B][nx], [B][ny]; NULL clears one. */
int mpc_set_model_offsets(mpc_handle *h, int32_t B, const double *px0, const double *py0);

/* defEstimator(...) -> kalman() / kalss(), MPC_code.py:577-650, Estimator.py:231-311.
 * xi = [xhat; dhat] [B][nx+nd] in/out; P [B][(nx+nd)^2] in/out (ignored for MPC_EST_FIXED_GAIN, may be NULL).
 * The predicted output Fy_model(xhat, dhat) (MPC_code.py:524) is formed inside. */
int mpc_kf_update(mpc_handle *h, int32_t B, const double *y, double *xi_inout, double *P_inout);

/*
 * The closed loop MPC_code.py:485-827 for B instances that share the problem and the schedules and differ
 * in their state.  Step order: measure (:524-534) -> estimate (:577-650) -> target (:693-718) -> OCP
 * (:733-805) -> plant (:813-816).  State lives in HBM inside the handle between calls.
 *
 *   mpc_loop_set_state   upload [B][.] arrays: x_p [nxp], xhat [nx], dhat [nd], P [(nx+nd)^2] (NULL unless
 *                        MPC_EST_KALMAN), u [nu], xs [nx], us [nu] (first step: xs = x0_m, us = u0, :682-684)
 *   mpc_loop_set_schedule per-step values shared by the batch, [nsteps][.]: ysp,usp,xsp = defSP(t) (:677-680),
 *                        pxp = def_pxp(t), pyp = def_pyp(t) (:512-515)
 *   mpc_loop_run         advance steps k0 .. k0+nsteps-1 of the schedule; asynchronous on the handle's stream
 *   mpc_loop_sync        wait for it
 *   mpc_loop_get_state   download the state (any pointer may be NULL)
 *   mpc_loop_get_log     download one log, [nsteps][B][dim] float64 ("U","X_HAT","XS","US","YS","Xp","D_HAT")
 *                        or [nsteps][B] int32 ("STATUS_DYN","STATUS_SS","ITERS_DYN","ITERS_SS") - the
 *                        reference's result arrays, MPC_code.py:877-895
 */
#define MPC_LOG_NONE 0
#define MPC_LOG_U 1   /* only U and the status / iteration words */
#define MPC_LOG_ALL 2
int mpc_loop_alloc(mpc_handle *h, int32_t B, int32_t max_steps, int32_t log_level);
int mpc_loop_set_state(mpc_handle *h, const double *x_p, const double *xhat, const double *dhat,
                       const double *P, const double *u, const double *xs, const double *us);
int mpc_loop_get_state(mpc_handle *h, double *x_p, double *xhat, double *dhat, double *P, double *u,
                       double *xs, double *us);
int mpc_loop_set_schedule(mpc_handle *h, int32_t nsteps, const double *ysp, const double *usp,
                          const double *xsp, const double *pxp, const double *pyp);
/* Model parameters that vary over the horizon, for every step of the fused loop (def_px / def_py, MPC_code.py:492-497): px [nsteps][N][nx] with
 * px[k][i] = def_px(t_k + i) - the reference's indexing: time plus stage index -, py [nsteps][N][ny] likewise; either may be NULL, both NULL switch
 * them off.  Estimator, target, the stage-0 output test, the hold rule and the plant see px[k][0], py[k][0] (p_x_k, p_y_k: MPC_code.py:500-507).
 * With a schedule set mpc_loop_run uses the instance-per-lane loop with per-block stage data, target and OCP cold at every step: the same numbers
 * as the call-by-call sequence mpc_set_model_offsets / mpc_kf_update / mpc_target_solve / mpc_ocp_solve(px, py). */
int mpc_loop_set_model_schedule(mpc_handle *h, int32_t nsteps, const double *px, const double *py);
int mpc_loop_run(mpc_handle *h, int32_t k0, int32_t nsteps);
int mpc_loop_sync(mpc_handle *h);
int mpc_loop_get_log(mpc_handle *h, const char *name, void *out);

/* Convenience wrapper with host buffers only (set_state, set_schedule, run, sync, get_state):
 * the fused equivalent of calling the three solvers nsteps times from MPC_code.py's loop. */
int mpc_closed_loop(mpc_handle *h, int32_t B, int32_t nsteps, double *x_p, double *xhat, double *dhat,
                    double *P, double *u, double *xs, double *us, const double *ysp, const double *usp,
                    const double *xsp, const double *pxp, const double *pyp, double *U_log /* [nsteps][B][nu] or NULL */);

/*
 * Measurement and integration hooks.
 *   mpc_last_kernel_ms   HIP-event time (ms) of the kernels of the last mpc_loop_run / solve call, measured
 *                        on the handle's stream; n_launches receives the number of launches it covers
 *   mpc_stream           the hipStream_t the handle launches on (for callers that order their own work)
 *   mpc_dev_ptr          device address of a resident array: state "x_p","xhat","dhat","P","u","xs","us"
 *                        (layout [dim][Bpad], instance index fastest) or a log ("U", ... layout
 *                        [step][dim][Bpad]); *bpad receives the padded batch stride
 *   mpc_pack_u           gather u of the last step into a dense [B][nu] device buffer owned by the caller
 *                        (what a rank hands to the all-gather of u*)
 */
float mpc_last_kernel_ms(mpc_handle *h, int32_t *n_launches);
void *mpc_stream(mpc_handle *h);
void *mpc_dev_ptr(mpc_handle *h, const char *name, int64_t *bpad);
int mpc_pack_u(mpc_handle *h, void *dst_dev /* [B][nu] float64 */);
/* device-to-device copy of steps [k0,k0+nsteps) of a float64 log, layout [step][dim][Bpad], into dst_dev
 * (asynchronous on the handle's stream) - the send buffer of the end-of-run all-gather of U */
int mpc_pack_log(mpc_handle *h, const char *name, int32_t k0, int32_t nsteps, void *dst_dev);

/* ---- multi-GPU: one process per GPU of one node, RCCL over xGMI -------------------------------------------------
 * Replaces nothing in the reference (it has no parallelism): the batch shards over the ranks with no
 * data-path collective; what is exchanged are the optimal controls.  Rendezvous: rank 0 calls mpc_comm_unique_id and hands the
 * 128 bytes to the other ranks by any means (mpc-code_amd/shard.py: a file under /tmp keyed by the launcher's pid); every rank
 * then calls mpc_comm_init on its handle (= its GPU).  Without a communicator every call below behaves as world size 1.
 *   mpc_allgather_u      ncclAllGather of u*[B][nu] of the last step (what MPC_code.py:798 reads out of sol["x"], over all ranks)
 *   mpc_allgather_log    the same for steps [k0, k0+nsteps) of a float64 log ("U", ...), device to device from the log
 *   mpc_comm_allgather   all-gather of equal-sized host buffers (statuses, ragged shards padded by the caller)
 *   mpc_comm_barrier     every rank's stream drained + one all-reduce: the bracket of a timed region
 *   mpc_comm_allreduce_max  maximum over the ranks of host doubles (slowest rank's time)                                               */
#define MPC_COMM_ID_BYTES 128
int mpc_comm_unique_id(char *out128);
int mpc_comm_init(mpc_handle *h, int32_t rank, int32_t world, const char *id128);
int mpc_comm_destroy(mpc_handle *h);
int mpc_comm_rank(mpc_handle *h, int32_t *rank, int32_t *world);
int mpc_comm_allgather(mpc_handle *h, const void *send, size_t bytes, void *recv /* [world][bytes] */);
int mpc_comm_allreduce_max(mpc_handle *h, double *inout, int32_t n);
int mpc_comm_barrier(mpc_handle *h);
int mpc_allgather_u(mpc_handle *h, double *u_all /* host [world][B][nu] or NULL: result stays on the device, mpc_dev_ptr "coll_recv" */);
int mpc_allgather_log(mpc_handle *h, const char *name, int32_t k0, int32_t nsteps, double *out /* host [world][nsteps][B][dim] or NULL */);

/* Tunables; they never change results beyond rounding.
 *   "ocp_warm_start"    0 (default) / 1: mpc_ocp_solve warm-starts from the caller's guess and its own previous call (see above)
 *   "ocp_kernel"        0 = auto, 1 = one instance per lane, 3 = wave-autonomous solver (N <= 64, stage state <= 8, nu <= 2)
 * of the resident closed loop (mpc_loop_run):
 *   "steps_per_launch"  closed-loop steps per kernel launch (default 50; a launch starts with cold caches)
 *   "loop_kernel"       0 = choose by problem and batch size (default), 1 = one instance per lane, 2 = horizon-parallel
 *                       (eight waves share sixteen instances, block-parallel element-wise work; needs N <= 64),
 *                       3 = wave-autonomous (one wave owns four instances for the whole launch, iterates resident in registers,
 *                       every recursion on the fp64 matrix cores; needs N <= 64, stage state <= 4, nu <= 2; the default there) */
int mpc_set_option(mpc_handle *h, const char *name, double value);
/* Current value; for "loop_kernel" the kernel mpc_loop_run will use for the allocated batch (1, 2 or 3). */
int mpc_get_option(mpc_handle *h, const char *name, double *value);

/* Library self-description: "gfx950;dims=3/2/3/3/3/0,4/2/2/2/4/1;..." */
const char *mpc_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* MPC_AMD_H */
