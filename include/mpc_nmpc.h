/*
 * mpc_nmpc.h - C-ABI of the non-linear MPC path (libmpc_nmpc_<model>.so): SURVEY.md section 8f rank 1, BASELINE config 3.
 *
 * The library is built per model: the Ex-file's User_fxm_Cont / User_fym / User_fxp_Cont / User_fyp are traced and emitted
 * as device functions (mpc-code_amd/nlcodegen.py), the kernels of csrc/mpc_nmpc.hip are compiled against them for gfx950.
 * It replaces, for a batch of instances, the loop body of the reference with a non-linear model (MPC_code.py:485-827):
 *   defEstimator(...) -> ekf()                 Estimator.py:313-386, MPC_code.py:577-650
 *   solver_ss(...) on the NLP of opt_ss        Target_Calc.py:20-161, MPC_code.py:693-718      (SQP on the linear target QP)
 *   solver(...) on the NLP of opt_dyn          Control_Calc.py:20-260, MPC_code.py:733-805     (SQP on the Riccati-PDIP solver:
 *                                              max_sqp = 1 is one real-time iteration per step, a larger value iterates to the
 *                                              NLP's KKT point)
 *   Fx_p / Fy_p                                Utilities.py:21-100, MPC_code.py:531-534,813-816
 * Conventions as mpc_amd.h: caller-owned contiguous float64 host arrays, one row per instance [B][dim]; status words 0 solved,
 * 1 iteration limit (accepted), 2 infeasible (hold rule); 0 on success, negative code + nmpc_last_error() otherwise; no CPU path.
 */
#ifndef MPC_NMPC_H
#define MPC_NMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nmpc_handle nmpc_handle;

typedef struct nmpc_desc {
    int32_t nx, nu, ny, nd, nxp; /* must equal the dimensions the library was generated for */
    int32_t N;                   /* horizon (<= 512) */
    int32_t max_iter;            /* interior-point iterations per QP (Sol_itmax) */
    int32_t device;
    double h;                    /* sampling interval */
    const double *Q, *R;         /* stage cost on x - xs, u - us (no terminal cost: Utilities.py:398-399) */
    const double *Qss, *Rss;     /* target cost */
    const double *umin, *umax, *xmin, *xmax, *ymin, *ymax;                         /* +-INFINITY = absent */
    const double *umin_ss, *umax_ss, *xmin_ss, *xmax_ss, *ymin_ss, *ymax_ss;
    const double *dmin, *dmax;   /* saturation of dhat, or NULL */
    const double *Q_kf, *R_kf;   /* EKF covariances [nx+nd]^2, [ny]^2 */
    const int32_t *ycols;        /* output row i is state ycols[i] (-1: not a single state; then it must be unbounded) */
    /* round-2 additions (discrete-time examples, Ex_NMPC_dis.py) */
    const double *Pf;            /* terminal weight 1/2 (x_N - xs)' Pf (x_N - xs) (User_vfin when it is such a form), or NULL */
    const double *Cd;            /* [ny][nd] of offree = 'lin' (y = h(x) + Cd d; the state part Bd d is inside the generated model), or NULL */
    const double *K;             /* [nx+nd][ny] fixed observer gain (lue), with estimator = 1 */
    int32_t estimator;           /* 0: extended Kalman filter (Q_kf, R_kf), 1: fixed gain xi+ = xi + K (y - yhat), Estimator.py:231-261 */
    int32_t du_form;             /* R weighs u_k - u_{k-1} (the Ex-file's S), Control_Calc.py:163-166,180-181 */
    int32_t duss_form;           /* Rss weighs us - us_prev (Sss), Target_Calc.py:121-122 */
    const double *Dumin, *Dumax; /* bounds on u_k - u_{k-1} (g2 rows), or NULL.  du_form or these need a library generated with DUV */
} nmpc_desc;

int nmpc_create(const nmpc_desc *desc, nmpc_handle **out);
void nmpc_destroy(nmpc_handle *h);
const char *nmpc_last_error(void);
const char *nmpc_build_info(void);      /* "gfx950;nmpc;dims=nx/nu/ny/nd/nxp;mx=.." */

/* resident closed loop, as mpc_loop_* of mpc_amd.h */
int nmpc_alloc(nmpc_handle *h, int32_t B, int32_t max_steps);
int nmpc_set_state(nmpc_handle *h, const double *x_p, const double *xhat, const double *dhat, const double *P, const double *u,
                   const double *xs, const double *us);
/* pxp [nsteps][nxp], pyp [nsteps][ny]: the plant's disturbances def_pxp(t), def_pyp(t) (MPC_code.py:512-515), or NULL */
int nmpc_set_schedule(nmpc_handle *h, int32_t nsteps, const double *ysp /* [nsteps][ny] */, const double *usp /* [nsteps][nu] */,
                      const double *pxp, const double *pyp);
/* white noise on the measurements of the resident loop: v [nsteps][B][ny] is added to y_k before the estimator - the reference's sqrtm(R_wn) N(0, I) of
 * MPC_code.py:537-541 (unseeded there; the draws are the caller's here: nmpc.py makes them from a seed).  NULL: none.  After nmpc_alloc */
int nmpc_set_noise(nmpc_handle *h, int32_t nsteps, const double *v);
/* steps [k0, k0+nsteps); asynchronous.  max_sqp SQP iterations per OCP (1 = real-time iteration), stopped early when the
 * trajectory moves less than sqp_tol */
int nmpc_run(nmpc_handle *h, int32_t k0, int32_t nsteps, int32_t max_sqp, double sqp_tol);
int nmpc_sync(nmpc_handle *h);
/* which closed-loop kernel nmpc_run launches: 1 = one instance per lane (any model), 3 = wave-autonomous (one wave owns four instances
 * for a launch, lane = stage, QP on the matrix cores), 4 = split pipeline (per step one lane-style launch for estimator / target /
 * plant and one wave-style launch for linearisation + QP); 3 and 4 need model state <= 4, nu <= 2, N <= 64 and no input-move form.
 * 0 = auto when the model fits: 3 while one round of its waves holds the batch (16 instances per CU: 4096 on an MI355X), 4 beyond;
 * else 1.  nmpc_get_kernel returns the one in force */
int nmpc_set_kernel(nmpc_handle *h, int32_t kernel);
/* split pipeline: the batch in `groups` parts (multiples of 64 instances) on HIP streams of their own, so that the lane-style launch of one part runs beside
 * the wave-style launches of the others; 0 = by batch size (default), 1..3.  Results do not depend on it; with nmpc_time_kernels on, one part. */
int nmpc_set_groups(nmpc_handle *h, int32_t groups);
int nmpc_get_kernel(nmpc_handle *h);
/* logs [nsteps][B][dim] float64: "U","X_HAT","XS","US","Xp","D_HAT"; [nsteps][B] int32: "STATUS_DYN","STATUS_SS","ITERS_DYN"
 * (interior-point iterations of the last QP),"SQP_DYN","SQP_SS" */
int nmpc_get_log(nmpc_handle *h, const char *name, void *out);
float nmpc_last_kernel_ms(nmpc_handle *h);      /* device time of the last nmpc_run (all its launches), HIP events on the library's stream */
/* split pipeline (kernel 4): with nmpc_time_kernels(h, 1) every wave-style launch (linearisation + QP, the dominant kernel) of the
 * following runs is bracketed by its own pair of HIP events; nmpc_wave_kernel_ms returns their sum and count for the last run
 * (0 launches when another kernel ran or timing is off) */
int nmpc_time_kernels(nmpc_handle *h, int32_t on);
int nmpc_wave_kernel_ms(nmpc_handle *h, float *total_ms, int32_t *launches);

/* ---- per-call seam: the reference's three solver calls of one closed-loop step, each for the whole batch -----------------------------
 * defEstimator(..., 'ekf' | 'lue') (MPC_code.py:577-650), solver_ss(lbx, ubx, x0, p, lbg, ubg) (:704-709), solver(lbx, ubx, x0, p, lbg, ubg) (:776-781), with
 * the plant in between (:531-534, :813-816): caller-owned host arrays [B][dim], blocking, one launch of the instance-per-lane kernel per call.  The
 * handle keeps between the calls what the reference's driver keeps - the last optimum, i.e. the shifted guess of :764 - and what this solver's warm
 * start compares against (estimate and targets before their updates).  nmpc_alloc + nmpc_set_state first; then per step, in this order,
 * nmpc_ekf_update, nmpc_target_solve, nmpc_ocp_solve: with nmpc_plant_step in between they reproduce nmpc_run (kernel 1) bit for bit
 * (tests/test_nmpc.py). */
/* y [B][ny] this step's measurement, u_prev [B][nu] the input applied over the step before; xhat [B][nx], dhat [B][nd], P [B][(nx+nd)^2]: the
 * estimator's state, prior in / posterior out (P is not used by the fixed-gain observer); dhat is saturated by dmin / dmax when given */
int nmpc_ekf_update(nmpc_handle *h, const double *y, const double *u_prev, double *xhat, double *dhat, double *P);
/* ysp [ny], usp [nu]: this step's set points; xs [B][nx], us [B][nu]: the targets of the step before in (first guess of the SQP and us_prev of the
 * input-move cost), this step's out - unchanged when status is 2 (MPC_code.py:714-718); sqp: SQP iterations taken */
int nmpc_target_solve(nmpc_handle *h, const double *dhat, const double *ysp, const double *usp, double *xs, double *us, int32_t *status, int32_t *sqp);
/* u_out [B][nu] the optimal first input, xnext_out [B][nx] the optimiser's next state (MPC_code.py:798-799); the held input and the model's
 * propagation with status 2 (:804-805); iters: interior-point iterations of the last QP */
int nmpc_ocp_solve(nmpc_handle *h, const double *xhat, const double *dhat, const double *xs, const double *us, const double *u_prev, int32_t max_sqp, double sqp_tol,
                   double *u_out, double *xnext_out, int32_t *status, int32_t *iters, int32_t *sqp);
/* the simulator's side, for callers without a plant of their own: x_p <- Fx_p(x_p, u) + pxp (pxp [nxp] or NULL), in place (Utilities.py:21-100) */
int nmpc_plant_step(nmpc_handle *h, const double *u, double *x_p, const double *pxp);

#ifdef __cplusplus
}
#endif
#endif /* MPC_NMPC_H */
