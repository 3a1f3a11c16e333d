/* ORACLE (test infrastructure, never shipped): the non-linear tracking MPC loop of Ex_NMPC.py, restated in plain C.
 *
 * PARITY UNPINNED against the reference's own solver (CasADi / IPOPT cannot run here, no vectors shipped - SURVEY.md 8c).
 * This file restates oracle/nmpc_oracle.py - model and plant by Mx Runge-Kutta steps (Utilities.py:157-183, :58-82), extended Kalman
 * filter (Estimator.py:313-386), the NLP of opt_ss by SQP on its linear QP (Target_Calc.py:20-161), the NLP of opt_dyn by SQP on the QP of
 * its own variable order (Control_Calc.py:20-260; one iteration per step = real-time iteration), the loop MPC_code.py:485-827 - for the
 * example family of Ex_NMPC.py (three-state CSTR, two inputs, feed flow as non-linearly entering disturbance), fast enough to re-run
 * thousands of closed loops on the host cores and to serve as the timed CPU baseline of bench.py --config nmpc.  What is its own:
 *   - the example's functions are written out by hand below (Ex_NMPC.py:114-150 model, :38-98 plant with its feed-flow schedule);
 *     oracle/nmpc_oracle_c.py checks them against the Ex-file's Python functions before anything is computed;
 *   - Jacobians are complex-step differences through the Runge-Kutta steps (the NumPy oracle: central differences; the product: symbolic);
 *   - every QP is solved by the interior point method of orc_dense.h with null-space (QR + Cholesky) Newton steps to 1e-10 - neither the
 *     NumPy oracle's dense Mehrotra + active-set polish nor the product's Riccati recursion.
 * Build: make -C oracle libnmpc_oracle.so.  Nothing under mpc-code_amd/ links, loads or calls this.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cplx;
enum { NX = 3, NU = 2, NY = 2, ND = 2, NE = NX + ND, NZ = NX + NU, NV = NX + NU + NY };
#define CS 1e-30
#define BOUND_RELAX 1e-8
#include "orc_dense.h"

typedef struct {
    int32_t N, Mx, max_iter, has_dsat;
    double h;
    double Q[NX][NX], R[NU][NU], Qss[NY][NY], Rss[NU][NU];
    double umin[NU], umax[NU], xmin[NX], xmax[NX], ymin[NY], ymax[NY], tlo[NV], thi[NV], dmin[ND], dmax[ND];
    double Qkf[NE][NE], Rkf[NY][NY], P0[NE][NE], x0m[NX], u0[NU], dhat0[ND];
} NProb;

/* ---- the example's functions, by hand (Ex_NMPC.py:129-147): checked against the Ex-file by the wrapper -------------------------------- */
static void cstr(const cplx *x, const cplx *u, cplx F0, cplx *dx)
{
    const double T0 = 350.0, c0 = 1.0, r = 0.219, k0 = 7.2e10, EoR = 8750.0, U0 = 915.6 * 60.0 / 1000.0, rho = 1000.0, Cp = 0.239, DH = -5.0e4, pi = 3.14159265358979323846;
    const double kT0 = k0 * exp(-EoR / T0), area = pi * r * r;
    const cplx rate = kT0 * cexp(-EoR * (1.0 / x[1] - 1.0 / T0)) * x[0];
    dx[0] = F0 * (c0 - x[0]) / (area * x[2]) - rate;
    dx[1] = F0 * (T0 - x[1]) / (area * x[2]) - DH / (rho * Cp) * rate + 2.0 * U0 / (r * rho * Cp) * (u[0] - x[1]);
    dx[2] = (F0 - u[1]) / area;
}
static double plant_feed(double t) { return t <= 5.0 ? 0.1 : (t <= 15.0 ? 0.15 : (t <= 25.0 ? 0.08 : 0.1)); }      /* Ex_NMPC.py:58-64 */

/* Mx classical Runge-Kutta steps; plant: the feed flow follows its schedule in time (time is carried along, Utilities.py:58-82) */
static void rk4(const NProb *P, const cplx *x0, const cplx *u, cplx F0, int plant, double t, cplx *xn)
{
    const double dt = P->h / P->Mx;
    cplx x[NX] = {x0[0], x0[1], x0[2]};
    for (int s = 0; s < P->Mx; s++) {
        const double ts = t + s * dt;
        cplx k[4][NX], xa[NX];
        for (int st = 0; st < 4; st++) {
            const double a = st == 0 ? 0.0 : (st == 3 ? 1.0 : 0.5);
            for (int i = 0; i < NX; i++) xa[i] = x[i] + (st ? a * dt * k[st - 1][i] : 0.0);
            cstr(xa, u, plant ? plant_feed(ts + a * dt) : F0, k[st]);
        }
        for (int i = 0; i < NX; i++) x[i] += dt / 6.0 * (k[0][i] + 2.0 * k[1][i] + 2.0 * k[2][i] + k[3][i]);
    }
    for (int i = 0; i < NX; i++) xn[i] = x[i];
}

/* F = Fx_model(x, u, d), A = dF/dx, B = dF/du, G = dF/dd (d[0] does not enter this model) by complex steps */
static void linearize(const NProb *P, const double *x, const double *u, const double *d, double *F, double (*A)[NX], double (*B)[NU], double (*G)[ND])
{
    for (int j = 0; j < NX + NU + ND; j++) {
        cplx xc[NX] = {x[0], x[1], x[2]}, uc[NU] = {u[0], u[1]}, dc[ND] = {d[0], d[1]}, o[NX];
        if (j < NX) xc[j] += I * CS; else if (j < NX + NU) uc[j - NX] += I * CS; else dc[j - NX - NU] += I * CS;
        rk4(P, xc, uc, dc[1], 0, 0.0, o);
        for (int r = 0; r < NX; r++) {
            const double v = cimag(o[r]) / CS;
            if (j < NX) A[r][j] = v; else if (j < NX + NU) B[r][j - NX] = v; else if (G) G[r][j - NX - NU] = v;
            F[r] = creal(o[r]);
        }
    }
}
static const int YCOL[NY] = {0, 2};      /* Fy_model = Fy_p = [x0, x2] (Ex_NMPC.py:88-98,152-175) */

/* ---- a QP  min 1/2 w'Hw + g'w,  E w = e,  lo <= w <= hi  through the interior point method (constant derivatives) ------------------- */
typedef struct { int n, m; const double *H, *g, *E, *e; } QpCtx;
static void qp_evalf(void *vctx, const double *w, const double *lam, int want_h, double *f, double *gf, double *gc, double *J, double *Hout)
{
    const QpCtx *c = (const QpCtx *)vctx;
    (void)lam;
    *f = 0.0;
    for (int i = 0; i < c->n; i++) { double s = c->g[i]; for (int j = 0; j < c->n; j++) s += c->H[i * c->n + j] * w[j]; gf[i] = s; *f += 0.5 * (s + c->g[i]) * w[i]; }
    for (int r = 0; r < c->m; r++) { double s = -c->e[r]; for (int j = 0; j < c->n; j++) s += c->E[r * c->n + j] * w[j]; gc[r] = s; }
    memcpy(J, c->E, sizeof(double) * c->m * c->n);
    if (want_h) memcpy(Hout, c->H, sizeof(double) * c->n * c->n);
}

static double qp_eq_violation(const QpCtx *c, const double *w)
{
    double v = 0.0;
    for (int r = 0; r < c->m; r++) { double s = -c->e[r]; for (int j = 0; j < c->n; j++) s += c->E[r * c->n + j] * w[j]; v = fmax(v, fabs(s)); }
    return v;
}

/* ---- target: SQP on the linear target QP in w = [xs, us, ys] (nmpc_oracle.py:target_solve) -------------------------------------------- */
static int target_solve(const NProb *P, const double *usp, const double *ysp, const double *d, double *xs, double *us, int *sqp_iters)
{
    const int n = NV, m = NX + NY;
    for (int it = 0; it < 30; it++) {
        double F[NX], A[NX][NX], B[NX][NU], H[NV * NV] = {0}, g[NV] = {0}, E[(NX + NY) * NV] = {0}, e[NX + NY] = {0}, w[NV];
        linearize(P, xs, us, d, F, A, B, NULL);
        for (int i = 0; i < NY; i++) for (int j = 0; j < NY; j++) { H[(NZ + i) * n + NZ + j] = P->Qss[i][j]; g[NZ + i] -= P->Qss[i][j] * ysp[j]; }
        for (int i = 0; i < NU; i++) for (int j = 0; j < NU; j++) { H[(NX + i) * n + NX + j] = P->Rss[i][j]; g[NX + i] -= P->Rss[i][j] * usp[j]; }
        for (int r = 0; r < NX; r++) {      /* (A - I) xs + B us = -(F - A xs - B us) */
            double c = F[r];
            for (int j = 0; j < NX; j++) { E[r * n + j] = A[r][j] - (r == j ? 1.0 : 0.0); c -= A[r][j] * xs[j]; }
            for (int j = 0; j < NU; j++) { E[r * n + NX + j] = B[r][j]; c -= B[r][j] * us[j]; }
            e[r] = -c;
        }
        for (int r = 0; r < NY; r++) { E[(NX + r) * n + YCOL[r]] = 1.0; E[(NX + r) * n + NZ + r] = -1.0; }
        for (int i = 0; i < NX; i++) w[i] = xs[i];
        for (int i = 0; i < NU; i++) w[NX + i] = us[i];
        for (int i = 0; i < NY; i++) w[NZ + i] = xs[YCOL[i]];
        QpCtx c = {n, m, H, g, E, e};
        int iters;
        int st = ipm_nullspace(n, m, qp_evalf, &c, w, P->tlo, P->thi, 1e-11, 200, &iters, NULL);
        if (st == ST_MAXITER && qp_eq_violation(&c, w) > 1e-6) st = ST_FAILED;      /* the iteration never became feasible: an infeasible QP */
        if (st != ST_SOLVED) { *sqp_iters = it; return st == ST_FAILED ? ST_FAILED : ST_MAXITER; }
        double step = 0.0;
        for (int i = 0; i < NX; i++) { step = fmax(step, fabs(w[i] - xs[i])); xs[i] = w[i]; }
        for (int i = 0; i < NU; i++) { step = fmax(step, fabs(w[NX + i] - us[i])); us[i] = w[NX + i]; }
        if (step < 1e-10) { *sqp_iters = it + 1; return ST_SOLVED; }
    }
    *sqp_iters = 30;
    return ST_MAXITER;
}

/* ---- OCP: SQP on the QP of opt_dyn's order without the given x0: w = [u0, x1, u1, ..., u_{N-1}, x_N] ------------------------------------ */
static int ocp_solve(const NProb *P, const double *xhat, const double *xs, const double *us, const double *d, double *wfull /* [x0,u0,x1,...,x_N], guess in / optimum out */,
                     int max_sqp, double sqp_tol, int *sqp_iters, double *u0, double *x1)
{
    const int N = P->N, n = NZ * N, m = NX * N;
    for (int r = 0; r < NY; r++) {      /* stage-0 output rows constrain a given quantity: a feasibility test (Control_Calc.py:128-151) */
        const double y0 = xhat[YCOL[r]];
        if (y0 < P->ymin[r] - BOUND_RELAX * fmax(1.0, fabs(P->ymin[r])) || y0 > P->ymax[r] + BOUND_RELAX * fmax(1.0, fabs(P->ymax[r]))) { *sqp_iters = 0; return ST_FAILED; }
    }
    const size_t mark_ = arena_mark();
    double *H = vec((size_t)n * n), *g = vec(n), *E = vec((size_t)m * n), *e = vec(m), *lo = vec(n), *hi = vec(n), *w = vec(n);
    for (int i = 0; i < NX; i++) wfull[i] = xhat[i];
    double step = INFINITY;
    int it = 0, status = ST_SOLVED;
    for (it = 0; it < max_sqp; it++) {
        memset(H, 0, sizeof(double) * n * n); memset(g, 0, sizeof(double) * n); memset(E, 0, sizeof(double) * m * n);
        for (int k = 0; k < N; k++) {
            const double *xk = wfull + NZ * k, *uk = wfull + NZ * k + NX;
            double F[NX], A[NX][NX], B[NX][NU];
            linearize(P, xk, uk, d, F, A, B, NULL);
            const int iu = NZ * k, ix = k > 0 ? NZ * (k - 1) + NU : -1, ixn = NZ * k + NU;
            for (int i = 0; i < NU; i++) for (int j = 0; j < NU; j++) { H[(iu + i) * n + iu + j] += P->R[i][j]; g[iu + i] -= P->R[i][j] * us[j]; }
            if (ix >= 0) for (int i = 0; i < NX; i++) for (int j = 0; j < NX; j++) { H[(ix + i) * n + ix + j] += P->Q[i][j]; g[ix + i] -= P->Q[i][j] * xs[j]; }
            for (int r = 0; r < NX; r++) {      /* A_k x_k + B_k u_k - x_{k+1} = -c_k; with k = 0 the known x_0 goes to the right-hand side */
                double c = F[r];
                for (int j = 0; j < NX; j++) c -= A[r][j] * xk[j];
                for (int j = 0; j < NU; j++) { c -= B[r][j] * uk[j]; E[(NX * k + r) * n + iu + j] = B[r][j]; }
                E[(NX * k + r) * n + ixn + r] = -1.0;
                double rhs = -c;
                for (int j = 0; j < NX; j++) { if (ix >= 0) E[(NX * k + r) * n + ix + j] = A[r][j]; else rhs -= A[r][j] * xhat[j]; }
                e[NX * k + r] = rhs;
            }
            for (int i = 0; i < NU; i++) { lo[iu + i] = P->umin[i]; hi[iu + i] = P->umax[i]; }
            for (int i = 0; i < NX; i++) { lo[ixn + i] = P->xmin[i]; hi[ixn + i] = P->xmax[i]; }
            if (k + 1 < N) for (int r = 0; r < NY; r++) { lo[ixn + YCOL[r]] = fmax(lo[ixn + YCOL[r]], P->ymin[r]); hi[ixn + YCOL[r]] = fmin(hi[ixn + YCOL[r]], P->ymax[r]); }      /* output rows k = 1..N-1: boxes on single states */
        }
        memcpy(w, wfull + NX, sizeof(double) * n);
        QpCtx c = {n, m, H, g, E, e};
        int iters;
        int st = ipm_nullspace(n, m, qp_evalf, &c, w, lo, hi, 1e-10, 300, &iters, NULL);
        if (st == ST_MAXITER && qp_eq_violation(&c, w) > 1e-6) st = ST_FAILED;      /* infeasible QP: the hold rule (MPC_code.py:804-805) */
        if (st == ST_FAILED) { status = ST_FAILED; break; }
        step = 0.0;
        for (int i = 0; i < n; i++) { step = fmax(step, fabs(w[i] - wfull[NX + i])); wfull[NX + i] = w[i]; }
        if (step < sqp_tol) { it++; break; }
    }
    *sqp_iters = it;
    if (status != ST_FAILED) {
        status = (step < sqp_tol || max_sqp == 1) ? ST_SOLVED : ST_MAXITER;
        for (int i = 0; i < NU; i++) u0[i] = wfull[NX + i];
        for (int i = 0; i < NX; i++) x1[i] = wfull[NZ + i];
    }
    arena_release(mark_);
    return status;
}

/* ---- extended Kalman filter (Estimator.py:313-386) -------------------------------------------------------------------------------------- */
static void ekf(const NProb *P, double *xi, double *Pk /* [NE][NE] */, const double *y, const double *u)
{
    double C[NY][NE] = {{0}}, PCt[NE][NY], S[NY * NY], Si[NY * NY], K[NE][NY], Pc[NE][NE];
    for (int r = 0; r < NY; r++) C[r][YCOL[r]] = 1.0;
    for (int i = 0; i < NE; i++) for (int r = 0; r < NY; r++) { double s = 0.0; for (int l = 0; l < NE; l++) s += Pk[i * NE + l] * C[r][l]; PCt[i][r] = s; }
    for (int r = 0; r < NY; r++) for (int q = 0; q < NY; q++) { double s = P->Rkf[r][q]; for (int l = 0; l < NE; l++) s += C[r][l] * PCt[l][q]; S[r * NY + q] = s; }
    inv_small(NY, S, Si);
    for (int i = 0; i < NE; i++) for (int r = 0; r < NY; r++) { double s = 0.0; for (int q = 0; q < NY; q++) s += PCt[i][q] * Si[q * NY + r]; K[i][r] = s; }
    for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = Pk[i * NE + j]; for (int r = 0; r < NY; r++) for (int l = 0; l < NE; l++) s -= K[i][r] * C[r][l] * Pk[l * NE + j]; Pc[i][j] = s; }
    double innov[NY];
    for (int r = 0; r < NY; r++) innov[r] = y[r] - xi[YCOL[r]];      /* y - Fy_model(prior) */
    for (int i = 0; i < NE; i++) { double s = 0.0; for (int r = 0; r < NY; r++) s += K[i][r] * innov[r]; xi[i] += s; }
    double F[NX], A[NX][NX], B[NX][NU], G[NX][ND], Aa[NE][NE] = {{0}}, T[NE][NE];
    linearize(P, xi, u, xi + NX, F, A, B, G);
    for (int i = 0; i < NX; i++) { for (int j = 0; j < NX; j++) Aa[i][j] = A[i][j]; for (int j = 0; j < ND; j++) Aa[i][NX + j] = G[i][j]; }
    for (int i = 0; i < ND; i++) Aa[NX + i][NX + i] = 1.0;
    for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = 0.0; for (int l = 0; l < NE; l++) s += Aa[i][l] * Pc[l][j]; T[i][j] = s; }
    for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = P->Qkf[i][j]; for (int l = 0; l < NE; l++) s += T[i][l] * Aa[j][l]; Pk[i * NE + j] = s; }
}

/* ---- closed loop of B instances (MPC_code.py:485-827); schedules [step][dim], logs [step][B][dim] -------------------------------------- */
int norc_closed_loop(const NProb *P, int B, int nsteps, const double *x0_p, const double *x0_m, const double *ysp, const double *usp, const double *pxp, const double *pyp,
                     int max_sqp, double sqp_tol, double *U, double *XHAT, double *XS, double *US, double *XP, double *DHAT,
                     int32_t *st_dyn, int32_t *st_ss, int32_t *sqp_dyn, int nthreads, const double *v_wn /* [nsteps][B][NY] white noise on the measurement (MPC_code.py:537-541), or NULL */)
{
    const int N = P->N;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        const size_t mark_b = arena_mark();
        double x[NX], xi[NE], u[NU], xs[NX], us[NU], Pk[NE * NE];
        for (int i = 0; i < NX; i++) { x[i] = x0_p[NX * b + i]; xi[i] = x0_m[NX * b + i]; xs[i] = xi[i]; }
        for (int i = 0; i < ND; i++) xi[NX + i] = P->dhat0[i];
        for (int i = 0; i < NU; i++) { u[i] = P->u0[i]; us[i] = u[i]; }
        for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) Pk[i * NE + j] = P->P0[i][j];
        double *w = vec(NZ * N + NX);
        for (int k = 0; k <= N; k++) { for (int i = 0; i < NX; i++) w[NZ * k + i] = xi[i]; if (k < N) for (int i = 0; i < NU; i++) w[NZ * k + NX + i] = u[i]; }      /* :740-756 */
        for (int k = 0; k < nsteps; k++) {
            const double t = k * P->h;
            const size_t o = (size_t)k * B + b;
            if (XP) for (int i = 0; i < NX; i++) XP[o * NX + i] = x[i];
            if (XHAT) for (int i = 0; i < NX; i++) XHAT[o * NX + i] = xi[i];
            double y[NY];
            for (int r = 0; r < NY; r++) y[r] = x[YCOL[r]] + pyp[NY * k + r] + (v_wn ? v_wn[o * NY + r] : 0.0);
            ekf(P, xi, Pk, y, u);
            if (P->has_dsat) for (int i = 0; i < ND; i++) xi[NX + i] = fmin(fmax(xi[NX + i], P->dmin[i]), P->dmax[i]);
            if (DHAT) for (int i = 0; i < ND; i++) DHAT[o * ND + i] = xi[NX + i];
            double xs_prev[NX], us_prev[NU], xst[NX], ust[NU];
            for (int i = 0; i < NX; i++) { xs_prev[i] = xs[i]; xst[i] = xs[i]; }
            for (int i = 0; i < NU; i++) { us_prev[i] = us[i]; ust[i] = us[i]; }
            int sqs, sqd;
            const int ss = target_solve(P, usp + NU * k, ysp + NY * k, xi + NX, xst, ust, &sqs);
            if (ss != ST_FAILED) { for (int i = 0; i < NX; i++) xs[i] = xst[i]; for (int i = 0; i < NU; i++) us[i] = ust[i]; }
            double u0[NU], x1[NX];
            const int sd = ocp_solve(P, xi, xs, us, xi + NX, w, max_sqp, sqp_tol, &sqd, u0, x1);
            if (sd != ST_FAILED) {
                for (int i = 0; i < NU; i++) u[i] = u0[i];
                for (int i = 0; i < NX; i++) xi[i] = x1[i];
                memmove(w, w + NZ, sizeof(double) * (NZ * (N - 1) + NX));      /* :764: shifted by one stage, previous target appended */
                for (int i = 0; i < NU; i++) w[NZ * (N - 1) + NX + i] = us_prev[i];
                for (int i = 0; i < NX; i++) w[NZ * N + i] = xs_prev[i];
            } else {
                cplx xc[NX] = {xi[0], xi[1], xi[2]}, uc[NU] = {u[0], u[1]}, xo[NX];
                rk4(P, xc, uc, xi[NX + 1], 0, 0.0, xo);
                for (int i = 0; i < NX; i++) xi[i] = creal(xo[i]);
            }
            if (U) for (int i = 0; i < NU; i++) U[o * NU + i] = u[i];
            if (XS) for (int i = 0; i < NX; i++) XS[o * NX + i] = xs[i];
            if (US) for (int i = 0; i < NU; i++) US[o * NU + i] = us[i];
            if (st_dyn) { st_dyn[o] = sd; st_ss[o] = ss; sqp_dyn[o] = sqd; }
            cplx xc[NX] = {x[0], x[1], x[2]}, uc[NU] = {u[0], u[1]}, xo[NX];
            rk4(P, xc, uc, 0.0, 1, t, xo);
            for (int i = 0; i < NX; i++) x[i] = creal(xo[i]) + pxp[NX * k + i];
        }
        arena_release(mark_b);
    }
    return 0;
}

/* the hand-written functions, for the wrapper's check against the Ex-file: model and plant right-hand sides */
void norc_functions(const double *x, const double *u, const double *d, double t, double *out)
{
    cplx xc[NX] = {x[0], x[1], x[2]}, uc[NU] = {u[0], u[1]}, dx[NX];
    cstr(xc, uc, d[1], dx);
    for (int i = 0; i < NX; i++) out[i] = creal(dx[i]);
    cstr(xc, uc, plant_feed(t), dx);
    for (int i = 0; i < NX; i++) out[NX + i] = creal(dx[i]);
}
int norc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
