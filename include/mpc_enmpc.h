/*
 * mpc_enmpc.h - C-ABI of the economic MPC path (libmpc_enmpc_<model>.so): SURVEY.md section 8f ranks 2 and 3, BASELINE configs 4, 5.
 *
 * The library is built per model: the Ex-file's User_fxm_Cont / User_fxp_Cont / User_fobj_Cont / User_fssobj / User_vfin /
 * User_fx_mhe_Cont / User_fobj_mhe are traced and emitted as device functions with their first and second derivatives
 * (mpc-code_amd/econcodegen.py), the kernel of csrc/mpc_enmpc.hip is compiled against them for gfx950.  It replaces, for a batch
 * of instances, the loop body of the reference for an economic example (MPC_code.py:485-827 with Ex_ENMPC.py):
 *   defEstimator(..., 'mhe') -> mhe()          Estimator.py:388-768 on the NLP of mhe_opt, Utilities.py:825-990; MPC_code.py:583-641
 *   solver_ss(...) on the NLP of opt_ss        Target_Calc.py:20-161 with User_fssobj; MPC_code.py:693-718
 *   solver(...) on the NLP of opt_dyn          Control_Calc.py:20-260 with ContForm (:102-111,153-158); MPC_code.py:733-805
 *   Fx_p / Fy_p                                Utilities.py:21-100; MPC_code.py:531-534,813-816
 * All three NLPs are solved to their KKT points by a primal-dual interior point method (the outer algorithm of the reference's
 * IPOPT); the shooting intervals of the OCP are integrated with quad_steps classical Runge-Kutta steps where
 * the reference calls SUNDIALS IDAS.
 * Conventions as mpc_amd.h: caller-owned contiguous float64 host arrays, one row per instance [B][dim]; status words 0 solved,
 * 1 iteration limit (accepted, as the reference accepts it), 2 failed (non-finite iterate: the hold rule, MPC_code.py:804-805);
 * 0 on success, negative code + enmpc_last_error() otherwise; no CPU path.
 */
#ifndef MPC_ENMPC_H
#define MPC_ENMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct enmpc_handle enmpc_handle;

typedef struct enmpc_desc {
    int32_t nx, nu, ny, nd, nxp, nw; /* must equal the dimensions the library was generated for (nw: noise vector of the estimator) */
    int32_t N;                       /* control horizon, 2..64 */
    int32_t N_mhe;                   /* estimation horizon, 2..63 */
    int32_t max_iter;                /* Sol_itmax: interior-point iterations per NLP */
    int32_t quad_steps;              /* Runge-Kutta steps per shooting interval of the OCP (cost quadrature included) */
    int32_t device;
    int32_t mhe_update;              /* update of the arrival cost, mhe_up (Estimator.py:626-736): 0 = 'smooth', 1 = 'filter' */
    int32_t estimator;               /* 0 = moving-horizon estimator (mhe = True), 1 = extended Kalman filter on [x; d] (ekf = True: the other position of the
                                        example's switch, Ex_ENMPC.py:109-123; Estimator.py:313-386 through MPC_code.py:640-650) */
    double h;                        /* sampling interval */
    double tol, tol_mhe;             /* optimality tolerances: IPOPT's default 1e-8, 1e-10 for the estimator (MPC_code.py:383) */
    const double *umin, *umax, *xmin, *xmax;                                   /* OCP boxes; +-INFINITY = absent */
    const double *umin_ss, *umax_ss, *xmin_ss, *xmax_ss, *ymin_ss, *ymax_ss;   /* target boxes */
    const double *xmin_mhe, *xmax_mhe;                                         /* [nx+nd] boxes of the estimator's states (MPC_code.py:397-402) */
    const double *dmin, *dmax;       /* saturation of dhat, or NULL */
    const double *Bd, *Cd;           /* [nx][nd], [ny][nd] of offree = 'lin' */
    const double *G_mhe;             /* [nx+nd][nw] */
    const double *P0;                /* [nx+nd]^2 initial arrival weight and Kalman covariance (MPC_code.py:421-422,455-458) */
    const double *x0_m, *u0;         /* the first guesses of target and OCP (MPC_code.py:696-700,740-756) */
    const double *Q_kf, *R_kf;       /* estimator = 1: [nx+nd]^2 process and [ny]^2 measurement noise covariances; P0 is then P(0|-1).  NULL with estimator = 0 */
    const double *wmin, *wmax;       /* [nw] bounds of the estimator's state noise (Utilities.py:881-884,974-977), +-INFINITY = absent; NULL = none.  A library generated
                                        for a problem without them (build info "wb=0") refuses finite ones */
} enmpc_desc;

int enmpc_create(const enmpc_desc *desc, enmpc_handle **out);
void enmpc_destroy(enmpc_handle *h);
const char *enmpc_last_error(void);
const char *enmpc_build_info(void);      /* "gfx950;enmpc;dims=nx/nu/ny/nd/nxp/nw;mx=.." */

/* resident closed loop */
int enmpc_alloc(enmpc_handle *h, int32_t B, int32_t max_steps);
/* plant state, model state, disturbance estimate, applied input, the estimator's prior mean x_bar [B][nx+nd]; resets the estimator
 * (window, covariances = P0) and the OCP's warm start; the first target guess is (x0_m, u0) */
int enmpc_set_state(enmpc_handle *h, const double *x_p, const double *xhat, const double *dhat, const double *u, const double *x_bar);
/* white noise of the resident loop: v [nsteps][B][ny] on the measurement of every step before the estimator (the reference's sqrtm(R_wn) N(0, I), MPC_code.py:537-541),
 * w [nsteps][B][nxp] on the plant state after its step (G_wn sqrtm(Q_wn) N(0, I), :822-827) - unseeded there, the caller's draws here (enmpc.py makes them from a seed).
 * Either may be NULL; both NULL: none.  After enmpc_alloc */
int enmpc_set_noise(enmpc_handle *h, int32_t nsteps, const double *v, const double *w);
/* steps [k0, k0+nsteps) of every instance in one launch; k0 must continue where the last run ended (0 after enmpc_set_state); asynchronous */
int enmpc_run(enmpc_handle *h, int32_t k0, int32_t nsteps);
int enmpc_sync(enmpc_handle *h);
/* which kernels enmpc_run launches: 1 = one launch for all steps (one wave per instance, lane = stage of the horizon, every phase in it:
 * the instance's state stays in registers); 2 = split pipeline, per step one launch for the estimator (wave = instance), one for the target
 * (lane = instance: the target problem is serial per instance) and one for OCP + plant (wave = instance); 0 = auto: 1 while one round of
 * waves holds the batch (four instances per CU: 1024 on an MI355X), 2 beyond.  In the split pipeline a wave holds 64 / SEG instances side
 * by side where a horizon fits SEG = 16 or 32 lanes (and the batch has waves to spare); kernel = 16 / 32 / 64 forces the split pipeline with
 * that SEG.  Same results either way.  enmpc_get_kernel: 1 or 2, the launch style in force */
int enmpc_set_kernel(enmpc_handle *h, int32_t kernel);
/* split pipeline: the batch in `groups` parts (multiples of 64 instances), each part's launches on a HIP stream of its own - the three launches of a
 * step depend on each other, the parts do not, so one part's target launch (a few dozen waves) runs beside another part's estimator or OCP launch.
 * 0 = by batch size (default), 1..8.  Results do not depend on it.  All streams are ordered after / before the handle's stream at the ends of
 * enmpc_run; with enmpc_time_kernels on, one part. */
int enmpc_set_groups(enmpc_handle *h, int32_t groups);
int enmpc_get_kernel(enmpc_handle *h);
/* logs [nsteps][B][dim] float64: "U","X_HAT","XS","US","Xp","D_HAT","X_ES" (the estimator's corrected [x; d]);
 * [nsteps][B] int32: "STATUS_DYN","STATUS_SS","STATUS_MHE","ITERS_DYN","ITERS_SS","ITERS_MHE" (interior-point iterations) */
int enmpc_get_log(enmpc_handle *h, const char *name, void *out);
float enmpc_last_kernel_ms(enmpc_handle *h);      /* device time of the last enmpc_run, HIP events on the library's stream */
/* split pipeline (kernel 2): with enmpc_time_kernels(h, 1) every launch of the following runs is bracketed by HIP events on the library's
 * stream; enmpc_phase_ms returns, for the last run, the summed device time of the estimator, target and OCP + plant kernels (ms3[0..2]) and
 * the number of launches of each (= its steps; 0 when kernel 1 ran or timing is off) */
int enmpc_time_kernels(enmpc_handle *h, int32_t on);
int enmpc_phase_ms(enmpc_handle *h, float *ms3, int32_t *launches);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI; as mpc_comm_* / mpc_allgather_log of mpc_amd.h ---------------------
 * The batch shards over the ranks with no data-path collective (instances are independent); the one exchange is the all-gather of the controls.
 * Rank 0 makes the 128-byte id (enmpc_comm_unique_id) and hands it to the others by any channel (mpc-code_amd/shard.py: a file); every rank then
 * calls enmpc_comm_init on its own handle.  Without a communicator every entry below degenerates to a copy. */
#define ENMPC_COMM_ID_BYTES 128
int enmpc_comm_unique_id(char *out128);
int enmpc_comm_init(enmpc_handle *h, int32_t rank, int32_t world, const char *id128);
int enmpc_comm_destroy(enmpc_handle *h);
int enmpc_comm_rank(enmpc_handle *h, int32_t *rank, int32_t *world);
/* host buffers, `bytes` per rank: rank r's block lands at recv + r * bytes (status words, ragged shards) */
int enmpc_comm_allgather(enmpc_handle *h, const void *send, size_t bytes, void *recv);
int enmpc_comm_allreduce_max(enmpc_handle *h, double *inout, int32_t n);
int enmpc_comm_barrier(enmpc_handle *h);      /* everything queued on every rank's stream has completed */
/* steps [k0, k0 + nsteps) of a float64 log ("U", ...) of every rank: ONE ncclAllGather, device to device, straight from the device log, asynchronous on
 * the handle's stream; out (host, optional) receives [world][nsteps][B][dim] */
int enmpc_allgather_log(enmpc_handle *h, const char *name, int32_t k0, int32_t nsteps, double *out);

/* ---- per-call seam: the reference's three solver calls of one closed-loop step, each for the whole batch -----------------------------
 * The reference's loop body calls, per step: defEstimator(..., 'mhe') (MPC_code.py:577-650), solver_ss(lbx, ubx, x0, p, lbg, ubg) (:704-709) and
 * solver(lbx, ubx, x0, p, lbg, ubg) (:776-781), with the plant in between (:531-534, :813-816).  These entries are those calls for B instances with
 * caller-owned host arrays [B][dim]; the same kernels as enmpc_run's split pipeline, one launch per call, blocking.  What the handle keeps
 * between the calls is what the reference's driver keeps: the estimator's window, prior and covariance lists (mhe()'s arguments), the targets of the
 * step before (xs_prev, us_prev: the tail of the shifted guess) and the OCP's last optimum (w_guess of :764).  enmpc_alloc + enmpc_set_state first
 * (they give the estimator's prior and the first guesses); then, per step and in this order, enmpc_mhe_update, enmpc_target_solve,
 * enmpc_ocp_solve - a step of the closed loop is exactly one call of each (tests/test_enmpc.py: the three calls + enmpc_plant_step reproduce
 * enmpc_run bit for bit).  status words as everywhere: 0 solved, 1 accepted without convergence, 2 the reference's hold rule. */
/* The estimator call of the step - the moving-horizon estimator or, with estimator = 1, the extended Kalman filter (whose predicted state x(k|k-1) is what
 * enmpc_ocp_solve left in the handle: the optimiser's next state, MPC_code.py:798-799, and whose covariance the handle keeps).
 * y [B][ny]: the measurement of this step (StateFeedback: the plant state); u_prev [B][nu]: the input applied over the step before (u0 at the first).
 * Out: xhat [B][nx], dhat [B][nd] (saturated by dmin / dmax when given, MPC_code.py:657-664), xes [B][nx+nd] the estimator's corrected [x; d] (or NULL) */
int enmpc_mhe_update(enmpc_handle *h, const double *y, const double *u_prev, double *xhat, double *dhat, double *xes, int32_t *status, int32_t *iters);
/* cold-started from (x0_m, u0) as the reference's (MPC_code.py:696-700); a failed solve (status 2) returns the targets of the step before (:714-718) */
int enmpc_target_solve(enmpc_handle *h, const double *dhat, double *xs, double *us, int32_t *status, int32_t *iters);
/* u_out [B][nu] = the optimal first input, xnext_out [B][nx] = the optimiser's next state (MPC_code.py:798-799); with status 2 the held input and
 * the model's propagation (:804-805) */
int enmpc_ocp_solve(enmpc_handle *h, const double *xhat, const double *dhat, const double *xs, const double *us, double *u_out, double *xnext_out,
                    int32_t *status, int32_t *iters);
/* the simulator's side, for callers without a plant of their own: x_p <- Fx_p(x_p, u) (Utilities.py:58-82), in place */
int enmpc_plant_step(enmpc_handle *h, const double *u, double *x_p);

#ifdef __cplusplus
}
#endif
#endif /* MPC_ENMPC_H */
