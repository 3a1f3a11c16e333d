/* ORACLE (test infrastructure, never shipped): the economic NMPC loop with the moving-horizon estimator, restated in plain C.
 *
 * PARITY UNPINNED against the reference's own solver (CasADi / IPOPT / IDAS cannot run here, no vectors shipped - SURVEY.md 8c).
 * This file restates oracle/enmpc_oracle.py - the NLPs of opt_dyn with ContForm (Control_Calc.py:20-260), opt_ss with User_fssobj
 * (Target_Calc.py:20-161), mhe_opt (Utilities.py:825-990), the bookkeeping of mhe() (Estimator.py:388-768) and the loop
 * MPC_code.py:485-827 - for the example family of Ex_ENMPC.py (two-state reactor, one input, output disturbance on both states), fast
 * enough to re-run whole batches on the host cores and to serve as the timed CPU baseline of bench.py.  What is its own:
 *   - the example's functions are written out by hand below (balances Ex_ENMPC.py:42-49,64-65; profit :194-233; terminal weight :236-252;
 *     estimator cost :166-173) with their parameters handed in; oracle/enmpc_oracle_c.py checks them against the Ex-file's Python
 *     functions at random points before anything is computed;
 *   - derivatives are complex-step differences of those functions through the Runge-Kutta steps (second derivatives: central
 *     differences of complex-step gradients): nothing symbolic, nothing generated;
 *   - the Newton systems of the interior point method are solved by a NULL-SPACE method: Householder QR of the constraint Jacobian,
 *     Cholesky of the reduced Hessian (whose failure IS the inertia test) - neither the dense LU of the NumPy oracle nor the product's
 *     Riccati recursion.
 * The outer algorithm (IPOPT's, at the reference's options) is the one documented in enmpc_oracle.py:ipm_dense.
 * Build: make -C oracle libenmpc_oracle.so.  Nothing under mpc-code_amd/ links, loads or calls this.
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
enum { NX = 2, NU = 1, NY = 2, ND = 2, NE = 4, NW = 4, NZ = NX + NU, NV = NX + NU + NY };
#define CS 1e-30
#include "orc_dense.h"
/* EORC_RESTO=0 (a test switch): the OCP and the estimator without the restoration phase, as the kernels of round 4 were - a failed line search at an infeasible point then ends the solve */
static int resto_on(void) { const char *e = getenv("EORC_RESTO"); return !(e && e[0] == '0'); }


typedef struct {
    int32_t N, N_mhe, Mx, quad, max_iter, has_dsat, mhe_filter;      /* mhe_filter: mhe_up = 'filter' (else 'smooth') */
    double h, tol, tol_mhe;
    double par[7];               /* cA0, V, k1, k2, alfa, beta, terminal weight */
    double umin[NU], umax[NU], xmin[NX], xmax[NX], tlo[NV], thi[NV], elo[NE], ehi[NE], dmin[ND], dmax[ND];
    double Bd[NX][ND], Cd[NY][ND], G[NE][NW], P0[NE][NE], x0m[NX], u0[NU];
    int32_t est_ekf, pad_;       /* estimator: 0 the moving-horizon estimator, 1 the extended Kalman filter on [x; d] (Ex_ENMPC.py:109-123, mhe_mod = 'off') */
    double Qkf[NE][NE], Rkf[NY][NY];
    double wlo[NW], whi[NW];     /* bounds of the estimator's state noise (Utilities.py:881-884,974-977); +-INFINITY = absent */
} EProb;

/* ---- the example's functions (hand-written; checked against the Ex-file by the Python wrapper) --------------------------------------- */
static void balances(const double *p, const cplx *x, cplx u, cplx *dx)
{
    dx[0] = u * (p[0] - x[0]) / p[1] - p[2] * x[0];
    dx[1] = -u * x[1] / p[1] + p[2] * x[0] - p[3] * x[1];
}
static cplx profit_cost(const double *p, cplx u, cplx y2) { return u * (p[4] * p[0] - p[5] * y2); }      /* User_fobj_Cont = User_fssobj */
static cplx vfin(const double *p, const cplx *x, const double *xs) { return p[6] * ((x[0] - xs[0]) * (x[0] - xs[0]) + (x[1] - xs[1]) * (x[1] - xs[1])); }
static cplx cost_mhe(const cplx *w, const cplx *v) { cplx s = 0; for (int i = 0; i < NW; i++) s += w[i] * w[i]; for (int i = 0; i < NY; i++) s += v[i] * v[i]; return 0.5 * s; }

/* Mx (or quad) classical Runge-Kutta steps of the balances; with quad: the cost rate integrated along (q) */
static void rk4(const EProb *P, const cplx *x0, cplx u, const double *d, int steps, int with_cost, cplx *xn, cplx *q)
{
    const double dt = P->h / steps;
    cplx z[3] = {x0[0], x0[1], 0.0};
    const int n = with_cost ? 3 : 2;
    for (int s = 0; s < steps; s++) {
        cplx k[4][3], za[3];
        for (int st = 0; st < 4; st++) {
            const double a = st == 0 ? 0.0 : (st == 3 ? 1.0 : 0.5);
            for (int i = 0; i < n; i++) za[i] = z[i] + (st ? a * dt * k[st - 1][i] : 0.0);
            balances(P->par, za, u, k[st]);
            if (with_cost) k[st][2] = profit_cost(P->par, u, za[1] + P->Cd[1][0] * d[0] + P->Cd[1][1] * d[1]);      /* y = x + Cd d (StateFeedback) */
        }
        for (int i = 0; i < n; i++) z[i] += dt / 6.0 * (k[0][i] + 2.0 * k[1][i] + 2.0 * k[2][i] + k[3][i]);
    }
    xn[0] = z[0]; xn[1] = z[1];
    if (q) *q = z[2];
}

/* ---- OCP: variables [u0, x1, u1, x2, ..., u_{N-1}, x_N] (x0 is a parameter, MPC_code.py:734), equalities x_{k+1} - F_k = 0 ------------- */
typedef struct { const EProb *P; const double *xhat, *xs, *us, *d; } OcpCtx;

static void interval_vals(const EProb *P, const cplx *z /* x0,x1,u */, const double *d, cplx *out /* x0+,x1+,q */)
{
    cplx xn[2], q;
    rk4(P, z, z[2], d, P->quad, 1, xn, &q);
    out[0] = xn[0]; out[1] = xn[1]; out[2] = q;
}
/* value, Jacobian [3][3] and Hessian of (q + pi' F) [3][3] of one interval at z = (x, u) */
static void interval_derivs(const EProb *P, const double *z, const double *d, const double *pi, int want_h, double *val, double (*Jc)[NZ], double (*Hc)[NZ])
{
    cplx zc[NZ], o[3];
    for (int j = 0; j < NZ; j++) {
        for (int i = 0; i < NZ; i++) zc[i] = z[i];
        zc[j] += I * CS;
        interval_vals(P, zc, d, o);
        for (int r = 0; r < 3; r++) { Jc[r][j] = cimag(o[r]) / CS; val[r] = creal(o[r]); }
    }
    if (!want_h) return;
    for (int j = 0; j < NZ; j++) {
        const double dz = 1e-5 * fmax(1.0, fabs(z[j]));
        double gp[NZ], gm[NZ];
        for (int sgn = 0; sgn < 2; sgn++) {
            double *gg = sgn ? gm : gp;
            for (int c = 0; c < NZ; c++) {
                for (int i = 0; i < NZ; i++) zc[i] = z[i];
                zc[j] += sgn ? -dz : dz;
                zc[c] += I * CS;
                interval_vals(P, zc, d, o);
                gg[c] = (cimag(o[2]) + pi[0] * cimag(o[0]) + pi[1] * cimag(o[1])) / CS;
            }
        }
        for (int c = 0; c < NZ; c++) Hc[j][c] = (gp[c] - gm[c]) / (2.0 * dz);
    }
    for (int a = 0; a < NZ; a++) for (int b = 0; b < a; b++) { const double s = 0.5 * (Hc[a][b] + Hc[b][a]); Hc[a][b] = s; Hc[b][a] = s; }
}

static void ocp_evalf(void *vctx, const double *w, const double *lam, int want, double *f, double *gf, double *g, double *J, double *H)
{
    const OcpCtx *c = (const OcpCtx *)vctx;
    const EProb *P = c->P;
    const int N = P->N, n = NZ * N, m = NX * N;
    const int want_h = want == WANT_HESS;
    *f = 0.0;
    if (want == WANT_VALUES) {      /* a trial point of the line search: cost and constraint values only */
        for (int k = 0; k < N; k++) {
            const double *xk = k == 0 ? c->xhat : w + NZ * (k - 1) + 1;
            cplx z[NZ] = {xk[0], xk[1], w[NZ * k]}, o[3];
            interval_vals(P, z, c->d, o);
            *f += creal(o[2]);
            for (int r = 0; r < NX; r++) g[NX * k + r] = w[NZ * k + 1 + r] - creal(o[r]);
        }
        for (int r = 0; r < NX; r++) { const double e = w[NZ * (N - 1) + 1 + r] - c->xs[r]; *f += P->par[6] * e * e; }
        return;
    }
    memset(gf, 0, sizeof(double) * n); memset(J, 0, sizeof(double) * m * n);
    if (want_h) memset(H, 0, sizeof(double) * n * n);
    /* variable index of u_k: 3k; of x_{k+1}: 3k + 1, 3k + 2 */
    for (int k = 0; k < N; k++) {
        double z[NZ], val[3], Jc[3][NZ], Hc[NZ][NZ], pi[NX];
        const double *xk = k == 0 ? c->xhat : w + NZ * (k - 1) + 1;
        z[0] = xk[0]; z[1] = xk[1]; z[2] = w[NZ * k];
        pi[0] = -lam[NX * k]; pi[1] = -lam[NX * k + 1];      /* L = f + lam'(x+ - F): the Hessian of the interval is that of q - lam'F */
        interval_derivs(P, z, c->d, pi, want_h, val, Jc, Hc);
        *f += val[2];
        const int iu = NZ * k, ix = k > 0 ? NZ * (k - 1) + 1 : -1, ixn = NZ * k + 1;
        gf[iu] += Jc[2][2];
        if (ix >= 0) { gf[ix] += Jc[2][0]; gf[ix + 1] += Jc[2][1]; }
        for (int r = 0; r < NX; r++) {
            const int row = NX * k + r;
            g[row] = w[ixn + r] - val[r];
            J[row * n + ixn + r] = 1.0;
            J[row * n + iu] = -Jc[r][2];
            if (ix >= 0) { J[row * n + ix] = -Jc[r][0]; J[row * n + ix + 1] = -Jc[r][1]; }
        }
        if (want_h) {
            const int idx[NZ] = {ix, ix + 1, iu};
            for (int a = 0; a < NZ; a++) for (int b = 0; b < NZ; b++) if ((a == 2 || ix >= 0) && (b == 2 || ix >= 0)) H[idx[a] * n + idx[b]] += Hc[a][b];
        }
    }
    /* terminal cost User_vfin(x_N, xs): p6 |x_N - xs|^2 */
    const int ixN = NZ * (N - 1) + 1;
    for (int r = 0; r < NX; r++) {
        const double e = w[ixN + r] - c->xs[r];
        *f += P->par[6] * e * e; gf[ixN + r] += 2.0 * P->par[6] * e;
        if (want_h) H[(ixN + r) * n + ixN + r] += 2.0 * P->par[6];
    }
}

static int ocp_solve(const EProb *P, const double *xhat, const double *xs, const double *us, const double *d, double *w /* guess in, optimum out */, int *iters)
{
    const int N = P->N, n = NZ * N, m = NX * N;
    const size_t mark_ = arena_mark();
    double *lo = vec(n), *hi = vec(n);
    for (int k = 0; k < N; k++) { lo[NZ * k] = P->umin[0]; hi[NZ * k] = P->umax[0]; for (int r = 0; r < NX; r++) { lo[NZ * k + 1 + r] = P->xmin[r]; hi[NZ * k + 1 + r] = P->xmax[r]; } }
    OcpCtx c = {P, xhat, xs, us, d};
    const int st = ipm_ipopt(n, m, ocp_evalf, &c, w, lo, hi, P->tol, P->max_iter, resto_on(), iters, NULL, NULL);      /* (with the restoration phase, as every NLP of the reference's solver) */
    arena_release(mark_);
    return st;
}

/* ---- target: w = [xs, us, ys]; g = [Fx_model(xs, us, d) - xs; xs + Cd d - ys] ------------------------------------------------------ */
typedef struct { const EProb *P; const double *d; } TgtCtx;
static void model_map(const EProb *P, const cplx *z /* x0,x1,u */, const double *d, cplx *out)
{
    cplx xn[2];
    rk4(P, z, z[2], d, P->Mx, 0, xn, NULL);
    for (int i = 0; i < NX; i++) out[i] = xn[i] + P->Bd[i][0] * d[0] + P->Bd[i][1] * d[1];
}
static void tgt_evalf(void *vctx, const double *w, const double *lam, int want, double *f, double *gf, double *g, double *J, double *H)
{
    const TgtCtx *c = (const TgtCtx *)vctx;
    const EProb *P = c->P;
    const int n = NV, m = NX + NY;
    const int want_h = want == WANT_HESS;
    cplx zc[NZ], o[NX];
    if (want == WANT_VALUES) {
        for (int i = 0; i < NZ; i++) zc[i] = w[i];
        model_map(P, zc, c->d, o);
        for (int r = 0; r < NX; r++) { g[r] = creal(o[r]) - w[r]; g[NX + r] = w[r] + P->Cd[r][0] * c->d[0] + P->Cd[r][1] * c->d[1] - w[NZ + r]; }
        *f = w[NX] * (P->par[4] * P->par[0] - P->par[5] * w[NZ + 1]);
        return;
    }
    memset(gf, 0, sizeof(double) * n); memset(J, 0, sizeof(double) * m * n);
    if (want_h) memset(H, 0, sizeof(double) * n * n);
    double F[NX], A[NX][NZ];
    for (int j = 0; j < NZ; j++) {
        for (int i = 0; i < NZ; i++) zc[i] = w[i];
        zc[j] += I * CS;
        model_map(P, zc, c->d, o);
        for (int r = 0; r < NX; r++) { A[r][j] = cimag(o[r]) / CS; F[r] = creal(o[r]); }
    }
    for (int r = 0; r < NX; r++) {
        g[r] = F[r] - w[r];
        for (int j = 0; j < NZ; j++) J[r * n + j] = A[r][j] - (j == r ? 1.0 : 0.0);
        g[NX + r] = w[r] + P->Cd[r][0] * c->d[0] + P->Cd[r][1] * c->d[1] - w[NZ + r];
        J[(NX + r) * n + r] = 1.0; J[(NX + r) * n + NZ + r] = -1.0;
    }
    /* cost us (alfa cA0 - beta ys2): gradient and Hessian by hand (bilinear) */
    const double us = w[NX], y2 = w[NZ + 1];
    *f = us * (P->par[4] * P->par[0] - P->par[5] * y2);
    gf[NX] = P->par[4] * P->par[0] - P->par[5] * y2; gf[NZ + 1] = -P->par[5] * us;
    if (want_h) {
        H[NX * n + NZ + 1] += -P->par[5]; H[(NZ + 1) * n + NX] += -P->par[5];
        for (int j = 0; j < NZ; j++) {      /* + sum lam_r Hess Fx_r: central differences of complex-step gradients */
            const double dz = 1e-5 * fmax(1.0, fabs(w[j]));
            double gp[NZ], gm[NZ];
            for (int sgn = 0; sgn < 2; sgn++) {
                double *gg = sgn ? gm : gp;
                for (int cc = 0; cc < NZ; cc++) {
                    for (int i = 0; i < NZ; i++) zc[i] = w[i];
                    zc[j] += sgn ? -dz : dz; zc[cc] += I * CS;
                    model_map(P, zc, c->d, o);
                    gg[cc] = (lam[0] * cimag(o[0]) + lam[1] * cimag(o[1])) / CS;
                }
            }
            for (int cc = 0; cc < NZ; cc++) H[j * n + cc] += 0.5 * (gp[cc] - gm[cc]) / (2.0 * dz), H[cc * n + j] += 0.5 * (gp[cc] - gm[cc]) / (2.0 * dz);
        }
    }
}

/* ---- MHE in mhe_opt's own layout: w = [x0 v0 w0 | x1 v1 w1 | ... | x_N] (blocks of NE + NY + NW), g = [Fy(X_k) + V_k - Y_k; Fx_mhe - X_{k+1}] -- */
typedef struct { const EProb *P; int N; const double *U, *Y, *xbar, *Pinv; } MheCtx;
enum { NB = NE + NY + NW };
static void mhe_map(const EProb *P, const cplx *xi, double u, const cplx *wn, cplx *out)
{
    cplx xn[2];
    rk4(P, xi, u, NULL, P->Mx, 0, xn, NULL);
    for (int i = 0; i < NE; i++) {
        cplx v = i < NX ? xn[i] + P->Bd[i][0] * xi[NX] + P->Bd[i][1] * xi[NX + 1] : xi[i];
        for (int j = 0; j < NW; j++) v += P->G[i][j] * wn[j];
        out[i] = v;
    }
}
static void mhe_evalf(void *vctx, const double *w, const double *lam, int want, double *f, double *gf, double *g, double *J, double *H)
{
    const MheCtx *c = (const MheCtx *)vctx;
    const EProb *P = c->P;
    const int N = c->N, n = N * NB + NE, m = N * (NY + NE);
    const int want_h = want == WANT_HESS;
    *f = 0.0;
    if (want == WANT_VALUES) {
        for (int k = 0; k < N; k++) {
            const int o0 = NB * k, r0 = (NY + NE) * k;
            const double *X = w + o0, *V = w + o0 + NE, *W = w + o0 + NE + NY;
            for (int i = 0; i < NW; i++) *f += 0.5 * W[i] * W[i];
            for (int i = 0; i < NY; i++) *f += 0.5 * V[i] * V[i];
            for (int r = 0; r < NY; r++) g[r0 + r] = X[r] + P->Cd[r][0] * X[NX] + P->Cd[r][1] * X[NX + 1] + V[r] - c->Y[NY * k + r];
            cplx zc[NE + NW], oo[NE];
            for (int i = 0; i < NE; i++) zc[i] = X[i];
            for (int i = 0; i < NW; i++) zc[NE + i] = W[i];
            mhe_map(P, zc, c->U[k], zc + NE, oo);
            for (int r = 0; r < NE; r++) g[r0 + NY + r] = creal(oo[r]) - w[o0 + NB + r];
        }
        for (int i = 0; i < NE; i++) { double s = 0.0; for (int j = 0; j < NE; j++) s += c->Pinv[i * NE + j] * (w[j] - c->xbar[j]); *f += 0.5 * (w[i] - c->xbar[i]) * s; }
        return;
    }
    memset(gf, 0, sizeof(double) * n); memset(J, 0, sizeof(double) * m * n);
    if (want_h) memset(H, 0, sizeof(double) * n * n);
    for (int k = 0; k < N; k++) {
        const int o0 = NB * k, r0 = (NY + NE) * k;
        const double *X = w + o0, *V = w + o0 + NE, *W = w + o0 + NE + NY;
        for (int i = 0; i < NW; i++) { *f += 0.5 * W[i] * W[i]; gf[o0 + NE + NY + i] += W[i]; if (want_h) H[(o0 + NE + NY + i) * n + o0 + NE + NY + i] += 1.0; }
        for (int i = 0; i < NY; i++) { *f += 0.5 * V[i] * V[i]; gf[o0 + NE + i] += V[i]; if (want_h) H[(o0 + NE + i) * n + o0 + NE + i] += 1.0; }
        for (int r = 0; r < NY; r++) {      /* Fy_es = x + Cd d */
            g[r0 + r] = X[r] + P->Cd[r][0] * X[NX] + P->Cd[r][1] * X[NX + 1] + V[r] - c->Y[NY * k + r];
            J[(r0 + r) * n + o0 + r] = 1.0; J[(r0 + r) * n + o0 + NX] += P->Cd[r][0]; J[(r0 + r) * n + o0 + NX + 1] += P->Cd[r][1]; J[(r0 + r) * n + o0 + NE + r] = 1.0;
        }
        cplx zc[NE + NW], oo[NE];
        double Fv[NE];
        const int zi[NE + NW] = {o0, o0 + 1, o0 + 2, o0 + 3, o0 + NE + NY, o0 + NE + NY + 1, o0 + NE + NY + 2, o0 + NE + NY + 3};
        for (int j = 0; j < NE + NW; j++) {
            for (int i = 0; i < NE + NW; i++) zc[i] = w[zi[i]];
            zc[j] += I * CS;
            mhe_map(P, zc, c->U[k], zc + NE, oo);
            for (int r = 0; r < NE; r++) { J[(r0 + NY + r) * n + zi[j]] = cimag(oo[r]) / CS; Fv[r] = creal(oo[r]); }
        }
        for (int r = 0; r < NE; r++) { g[r0 + NY + r] = Fv[r] - w[o0 + NB + r]; J[(r0 + NY + r) * n + o0 + NB + r] = -1.0; }
        if (want_h) {      /* + sum lam Hess Fx_mhe (the output map is linear) */
            for (int j = 0; j < NX; j++) {      /* only the x-part is non-linear: the model is affine in d and w */
                const double dz = 1e-5 * fmax(1.0, fabs(w[o0 + j]));
                double gp[NX], gm[NX];
                for (int sgn = 0; sgn < 2; sgn++) {
                    double *gg = sgn ? gm : gp;
                    for (int cc = 0; cc < NX; cc++) {
                        for (int i = 0; i < NE + NW; i++) zc[i] = w[zi[i]];
                        zc[j] += sgn ? -dz : dz; zc[cc] += I * CS;
                        mhe_map(P, zc, c->U[k], zc + NE, oo);
                        double s = 0.0;
                        for (int r = 0; r < NE; r++) s += lam[r0 + NY + r] * cimag(oo[r]);
                        gg[cc] = s / CS;
                    }
                }
                for (int cc = 0; cc < NX; cc++) { const double hjc = 0.5 * (gp[cc] - gm[cc]) / (2.0 * dz); H[(o0 + j) * n + o0 + cc] += hjc; H[(o0 + cc) * n + o0 + j] += hjc; }
            }
        }
    }
    for (int i = 0; i < NE; i++) {      /* arrival cost (Utilities.py:944-945) */
        double s = 0.0;
        for (int j = 0; j < NE; j++) { s += c->Pinv[i * NE + j] * (w[j] - c->xbar[j]); if (want_h) H[i * n + j] += 0.5 * (c->Pinv[i * NE + j] + c->Pinv[j * NE + i]); }
        gf[i] += s; *f += 0.5 * (w[i] - c->xbar[i]) * s;
    }
}

/* ---- what mhe() carries from call to call ------------------------------------------------------------------------------------------- */
typedef struct {
    double U[64], Y[64 * NY], wk[NW], vk[NY], xbar[NE], Pk[NE * NE], Pkal[NE * NE], bA[64][NE * NE], bP[64][NE * NE], bPc[64][NE * NE];
    double XL[64][NE], WL[64][NW];      /* the lists of x(k+1|k) and w_k the 'filter' update reads (:541-554; V is not needed: the cost's Hessian is constant) */
    int nU, nL;
} MheState;

static void mm4(const double *A, const double *B, double *C, int tb) { for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = 0.0; for (int l = 0; l < NE; l++) s += A[i * NE + l] * (tb ? B[j * NE + l] : B[l * NE + j]); C[i * NE + j] = s; } }

/* P(k|k) and P(k+1|k) of an extended Kalman step on the covariance Pin, linearised at (x, u, w) - Estimator.py:557-623 and :629-649 are
   this with different arguments; the Hessian of the estimator cost is the identity here, so Q_k = I, R_k = I, S_k = 0 (no cross terms) */
static void ekf_cov(const EProb *P, const double *Pin, const double *x, double u, const double *wk, double *Ak, double *Pc, double *Pn)
{
    double Ca[NY * NE], K[NE * NY], Sm[NY * NY], Si[NY * NY], T1[NE * NE], T2[NE * NE];
    cplx zc[NE + NW], oo[NE];
    for (int j = 0; j < NE; j++) {
        for (int i = 0; i < NE; i++) zc[i] = x[i];
        for (int i = 0; i < NW; i++) zc[NE + i] = wk[i];
        zc[j] += I * CS;
        mhe_map(P, zc, u, zc + NE, oo);
        for (int r = 0; r < NE; r++) Ak[r * NE + j] = cimag(oo[r]) / CS;
    }
    for (int r = 0; r < NY; r++) for (int i = 0; i < NE; i++) Ca[r * NE + i] = i < NX ? (r == i) : P->Cd[r][i - NX];
    for (int r = 0; r < NY; r++) for (int q = 0; q < NY; q++) { double s = r == q ? 1.0 : 0.0; for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) s += Ca[r * NE + i] * Pin[i * NE + j] * Ca[q * NE + j]; Sm[r * NY + q] = s; }
    inv_small(NY, Sm, Si);
    for (int i = 0; i < NE; i++) for (int r = 0; r < NY; r++) { double s = 0.0; for (int j = 0; j < NE; j++) for (int q = 0; q < NY; q++) s += Pin[i * NE + j] * Ca[q * NE + j] * Si[q * NY + r]; K[i * NY + r] = s; }
    for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = Pin[i * NE + j]; for (int r = 0; r < NY; r++) for (int l = 0; l < NE; l++) s -= K[i * NY + r] * Ca[r * NE + l] * Pin[l * NE + j]; Pc[i * NE + j] = s; }
    mm4(Ak, Pc, T1, 0); mm4(T1, Ak, T2, 1);      /* A Pc A' */
    for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = T2[i * NE + j]; for (int l = 0; l < NW; l++) s += P->G[i][l] * P->G[j][l]; Pn[i * NE + j] = s; }      /* + G Q G' */
}

/* ekf() of Estimator.py:313-386 with Fx_es = [Fx_model(x, u, d); d], Fy_es = x + Cd d (MPC_code.py:546-561): Pk = P(k|k-1) in, P(k+1|k) out; xes = [x; d](k|k-1) in,
   (k|k) out.  The state map is linearised at the corrected estimate (:371-379). */
static void ekf_step(const EProb *P, double *Pk, double *xes, const double *y, double u)
{
    double Ca[NY * NE], K[NE * NY], Sm[NY * NY], Si[NY * NY], Pc[NE * NE], Ak[NE * NE], T1[NE * NE], T2[NE * NE], e[NY];
    for (int r = 0; r < NY; r++) for (int i = 0; i < NE; i++) Ca[r * NE + i] = i < NX ? (r == i) : P->Cd[r][i - NX];
    for (int r = 0; r < NY; r++) { double yh = 0.0; for (int i = 0; i < NE; i++) yh += Ca[r * NE + i] * xes[i]; e[r] = y[r] - yh; }
    for (int r = 0; r < NY; r++) for (int q = 0; q < NY; q++) { double s = P->Rkf[r][q]; for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) s += Ca[r * NE + i] * Pk[i * NE + j] * Ca[q * NE + j]; Sm[r * NY + q] = s; }
    inv_small(NY, Sm, Si);
    for (int i = 0; i < NE; i++) for (int r = 0; r < NY; r++) { double s = 0.0; for (int j = 0; j < NE; j++) for (int q = 0; q < NY; q++) s += Pk[i * NE + j] * Ca[q * NE + j] * Si[q * NY + r]; K[i * NY + r] = s; }
    for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = Pk[i * NE + j]; for (int r = 0; r < NY; r++) for (int l = 0; l < NE; l++) s -= K[i * NY + r] * Ca[r * NE + l] * Pk[l * NE + j]; Pc[i * NE + j] = s; }
    for (int i = 0; i < NE; i++) { double s = xes[i]; for (int r = 0; r < NY; r++) s += K[i * NY + r] * e[r]; xes[i] = s; }
    for (int j = 0; j < NE; j++) {      /* A = d [Fx_model(x, u, d); d] / d [x; d] by complex steps */
        cplx zc[3] = {xes[0], xes[1], u}, dc[ND] = {xes[2], xes[3]}, oo[NX + 1];
        if (j < NX) zc[j] += I * CS; else dc[j - NX] += I * CS;
        double dr[ND] = {0.0, 0.0};
        model_map(P, zc, dr, oo);      /* the flow without its Bd d; that term is added here with the complex d */
        for (int r = 0; r < NX; r++) { cplx v = oo[r]; for (int l = 0; l < ND; l++) v += P->Bd[r][l] * dc[l]; Ak[r * NE + j] = cimag(v) / CS; }
        for (int r = 0; r < ND; r++) Ak[(NX + r) * NE + j] = cimag(dc[r]) / CS;
    }
    mm4(Ak, Pc, T1, 0); mm4(T1, Ak, T2, 1);
    for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) Pk[i * NE + j] = T2[i * NE + j] + P->Qkf[i][j];
}

static int mhe_step(const EProb *P, MheState *S, int ksim, const double *y, double u, double *xes, int *iters)
{
    const int Nm = P->N_mhe, N = ksim + 1 < Nm ? ksim + 1 : Nm;
    if (ksim >= Nm) { memmove(S->U, S->U + 1, sizeof(double) * (Nm - 1)); memmove(S->Y, S->Y + NY, sizeof(double) * NY * (Nm - 1)); }
    S->Y[NY * (N - 1)] = y[0]; S->Y[NY * (N - 1) + 1] = y[1];
    S->U[N - 1] = u; if (N >= 2) S->U[N - 2] = u;
    const int n = N * NB + NE;
    const size_t mark_ = arena_mark();
    double *w = vec(n), *lo = vec(n), *hi = vec(n);
    cplx xc[NE], wz[NW] = {0, 0, 0, 0}, xo[NE];
    for (int i = 0; i < NE; i++) xc[i] = S->xbar[i];
    for (int k = 0; k <= N; k++) {
        for (int i = 0; i < NE; i++) { w[NB * k + i] = creal(xc[i]); lo[NB * k + i] = P->elo[i]; hi[NB * k + i] = P->ehi[i]; }
        if (k < N) {
            for (int i = NE; i < NB; i++) { lo[NB * k + i] = i < NE + NY ? -INFINITY : P->wlo[i - NE - NY]; hi[NB * k + i] = i < NE + NY ? INFINITY : P->whi[i - NE - NY]; }
            mhe_map(P, xc, S->U[k], wz, xo);
            for (int i = 0; i < NE; i++) xc[i] = xo[i];
        }
    }
    double Pinv[NE * NE];
    int ok = inv_small(NE, S->Pk, Pinv);
    MheCtx c = {P, N, S->U, S->Y, S->xbar, Pinv};
    int st = ipm_ipopt(n, N * (NY + NE), mhe_evalf, &c, w, lo, hi, P->tol_mhe, P->max_iter, resto_on(), iters, NULL, NULL);
    if (!ok) st = ST_FAILED;
    const double *Xl = w + NB * (N - 1);
    for (int i = 0; i < NE; i++) xes[i] = Xl[i];
    for (int i = 0; i < NY; i++) S->vk[i] = Xl[NE + i];
    if (ksim != 0) for (int i = 0; i < NW; i++) S->wk[i] = Xl[NE + NY + i];
    const int idx = ksim < Nm - 1 ? ksim : Nm - 1;
    {   /* the lists of one-step predictions and process noises (:541-554) */
        const int il = ksim < Nm ? ksim : Nm - 1;
        if (ksim >= Nm) for (int i = 0; i + 1 < Nm; i++) { memcpy(S->XL[i], S->XL[i + 1], sizeof(S->XL[0])); memcpy(S->WL[i], S->WL[i + 1], sizeof(S->WL[0])); }
        for (int i = 0; i < NE; i++) S->XL[il][i] = w[NB * N + i];
        for (int i = 0; i < NW; i++) S->WL[il][i] = S->wk[i];
    }
    double Ak[NE * NE], Pc[NE * NE], Pn[NE * NE], T1[NE * NE], T2[NE * NE];
    if (P->mhe_filter) {      /* (:557-623 also run for 'filter' in the reference, into lists nothing reads afterwards: left out here) */
        if (ksim >= Nm - 1) {      /* :627-649: one Kalman step on the prior weight at the window's first entries; :740-748 */
            ekf_cov(P, S->Pk, S->XL[0], S->U[0], S->WL[0], Ak, Pc, Pn);
            memcpy(S->Pk, Pn, sizeof(Pn));
            for (int i = 0; i < NE; i++) S->xbar[i] = S->XL[0][i];
        }
        arena_release(mark_);
        return st;
    }
    /* Kalman quantities (Estimator.py:558-623) */
    ekf_cov(P, S->Pkal, xes, u, S->wk, Ak, Pc, Pn);
    memcpy(S->bA[idx], Ak, sizeof(Ak)); memcpy(S->bP[idx], S->Pkal, sizeof(Ak)); memcpy(S->bPc[idx], Pc, sizeof(Ak));
    memcpy(S->Pkal, Pn, sizeof(Pn));
    if (ksim >= Nm - 1) {      /* smoothing (:652-665) */
        double Pis[NE * NE], Pim[NE * NE], D[NE * NE], Gn[NE * NE];
        memcpy(Pis, S->bPc[Nm - 1], sizeof(Pis));
        for (int i2 = Nm - 2; i2 >= 1; i2--) {
            inv_small(NE, S->bP[i2 + 1], Pim);
            for (int i = 0; i < NE * NE; i++) D[i] = Pis[i] - S->bP[i2 + 1][i];
            mm4(S->bPc[i2], S->bA[i2], T1, 1); mm4(T1, Pim, Gn, 0);      /* Pc A' P^-1 */
            mm4(Gn, D, T1, 0); mm4(T1, Gn, T2, 1);
            for (int i = 0; i < NE * NE; i++) Pis[i] = S->bPc[i2][i] + T2[i];
        }
        memcpy(S->Pk, Pis, sizeof(Pis));
        for (int i = 0; i + 1 < Nm; i++) { memcpy(S->bA[i], S->bA[i + 1], sizeof(Ak)); memcpy(S->bP[i], S->bP[i + 1], sizeof(Ak)); memcpy(S->bPc[i], S->bPc[i + 1], sizeof(Ak)); }
        for (int i = 0; i < NE; i++) S->xbar[i] = w[NB + i];
    }
    arena_release(mark_);
    return st;
}

/* ---- closed loop of B instances (MPC_code.py:485-827); logs [step][B][dim] ----------------------------------------------------------- */
int eorc_closed_loop(const EProb *P, int B, int nsteps, const double *x0_p, const double *x_bar0, double *U, double *XS, double *US, double *XES, double *XP,
                     int32_t *st_dyn, int32_t *st_ss, int32_t *st_mhe, int32_t *it_dyn, int32_t *it_ss, int32_t *it_mhe, int nthreads,
                     const double *v_wn /* [nsteps][B][NY] white noise on the measurement (MPC_code.py:537-541), or NULL */, const double *w_wn /* [nsteps][B][NX] on the plant state (:822-827), or NULL */)
{
    const int N = P->N;
    if (N > 64 || P->N_mhe > 63) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        MheState *S = (MheState *)calloc(1, sizeof(MheState));
        double x[NX] = {x0_p[NX * b], x0_p[NX * b + 1]}, xh[NX] = {P->x0m[0], P->x0m[1]}, dh[ND] = {0, 0}, u = P->u0[0], xs[NX] = {P->x0m[0], P->x0m[1]}, us = P->u0[0];
        for (int i = 0; i < NE; i++) { S->xbar[i] = x_bar0 ? x_bar0[NE * b + i] : (i < NX ? P->x0m[i] : 0.0); for (int j = 0; j < NE; j++) S->Pk[i * NE + j] = S->Pkal[i * NE + j] = P->P0[i][j]; }
        const size_t mark_b = arena_mark();
        double *wopt = vec(NZ * N), *wg = vec(NZ * N), *wtry = vec(NZ * N);
        int have_w = 0, last_ok = 1;
        for (int k = 0; k < nsteps; k++) {
            const size_t o = (size_t)k * B + b;
            if (XP) { XP[o * NX] = x[0]; XP[o * NX + 1] = x[1]; }
            double xes[NE], ym[NY] = {x[0], x[1]};      /* y = x_p (StateFeedback plant output) + its white noise */
            if (v_wn) { ym[0] += v_wn[o * NY]; ym[1] += v_wn[o * NY + 1]; }
            int itm, its, itd;
            int sm = 0;
            if (P->est_ekf) { xes[0] = xh[0]; xes[1] = xh[1]; xes[2] = dh[0]; xes[3] = dh[1]; ekf_step(P, S->Pkal, xes, ym, u); itm = 0; }      /* (P_k lives where the other estimator keeps its filter covariance) */
            else sm = mhe_step(P, S, k, ym, u, xes, &itm);
            xh[0] = xes[0]; xh[1] = xes[1]; dh[0] = xes[2]; dh[1] = xes[3];
            if (P->has_dsat) for (int i = 0; i < ND; i++) dh[i] = fmin(fmax(dh[i], P->dmin[i]), P->dmax[i]);
            const double xs_prev[NX] = {xs[0], xs[1]}, us_prev = us;
            double v[NV] = {P->x0m[0], P->x0m[1], P->u0[0], P->x0m[0] + P->Cd[0][0] * dh[0] + P->Cd[0][1] * dh[1], P->x0m[1] + P->Cd[1][0] * dh[0] + P->Cd[1][1] * dh[1]};
            TgtCtx tc = {P, dh};
            const int ss = ipm_ipopt(NV, NX + NY, tgt_evalf, &tc, v, P->tlo, P->thi, P->tol, P->max_iter, 1, &its, NULL, NULL);      /* (the target: with the restoration phase, as in the product) */
            if (ss != ST_FAILED) { xs[0] = v[0]; xs[1] = v[1]; us = v[2]; }
            if (!have_w) for (int kk = 0; kk < N; kk++) { wg[NZ * kk] = P->u0[0]; wg[NZ * kk + 1] = P->x0m[0]; wg[NZ * kk + 2] = P->x0m[1]; }
            else if (last_ok) { memcpy(wg, wopt + NZ, sizeof(double) * NZ * (N - 1)); wg[NZ * (N - 1)] = us_prev; wg[NZ * (N - 1) + 1] = xs_prev[0]; wg[NZ * (N - 1) + 2] = xs_prev[1]; }
            memcpy(wtry, wg, sizeof(double) * NZ * N);
            const int sd = ocp_solve(P, xh, xs, &us, dh, wtry, &itd);
            last_ok = sd != ST_FAILED; have_w = 1;
            if (last_ok) { memcpy(wopt, wtry, sizeof(double) * NZ * N); u = wopt[0]; xh[0] = wopt[1]; xh[1] = wopt[2]; }
            else { cplx xc[2] = {xh[0], xh[1]}, xo[NX + 1]; cplx zz[3] = {xc[0], xc[1], u}; model_map(P, zz, dh, xo); xh[0] = creal(xo[0]); xh[1] = creal(xo[1]); }
            if (U) U[o] = u;
            if (XS) { XS[o * NX] = xs[0]; XS[o * NX + 1] = xs[1]; }
            if (US) US[o] = us;
            if (XES) for (int i = 0; i < NE; i++) XES[o * NE + i] = xes[i];
            if (st_dyn) { st_dyn[o] = sd; st_ss[o] = ss; st_mhe[o] = sm; it_dyn[o] = itd; it_ss[o] = its; it_mhe[o] = itm; }
            cplx xc[2] = {x[0], x[1]}, xo[2];
            rk4(P, xc, u, NULL, P->Mx, 0, xo, NULL);      /* plant: the same balances (Ex_ENMPC.py:42-49) */
            x[0] = creal(xo[0]); x[1] = creal(xo[1]);
            if (w_wn) { x[0] += w_wn[o * NX]; x[1] += w_wn[o * NX + 1]; }
        }
        arena_release(mark_b); free(S);
    }
    return 0;
}

/* the hand-written functions, for the wrapper's check against the Ex-file */
void eorc_functions(const EProb *P, const double *x, double u, const double *d, const double *xs, const double *wv, double *out)
{
    cplx xc[2] = {x[0], x[1]}, dx[2];
    balances(P->par, xc, u, dx);
    out[0] = creal(dx[0]); out[1] = creal(dx[1]);
    out[2] = creal(profit_cost(P->par, u, x[1] + P->Cd[1][0] * d[0] + P->Cd[1][1] * d[1]));
    out[3] = creal(vfin(P->par, xc, xs));
    cplx w[NW] = {wv[0], wv[1], wv[2], wv[3]}, v[NY] = {wv[4], wv[5]};
    out[4] = creal(cost_mhe(w, v));
}

int eorc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
