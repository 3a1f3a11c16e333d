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
enum { ST_SOLVED = 0, ST_MAXITER = 1, ST_FAILED = 2 };
#define CS 1e-30
#define KAPPA_PUSH 1e-2
#define MU_INIT 0.1
#define KAPPA_EPS 10.0
#define KAPPA_MU 0.2
#define THETA_MU 1.5
#define TAU_MIN 0.99
#define KAPPA_SIGMA 1e10
#define S_MAX 100.0
#define DELTA_FIRST 1e-4
#define DELTA_MAX 1e40

typedef struct {
    int32_t N, N_mhe, Mx, quad, max_iter, has_dsat;
    double h, tol, tol_mhe;
    double par[7];               /* cA0, V, k1, k2, alfa, beta, terminal weight */
    double umin[NU], umax[NU], xmin[NX], xmax[NX], tlo[NV], thi[NV], elo[NE], ehi[NE], dmin[ND], dmax[ND];
    double Bd[NX][ND], Cd[NY][ND], G[NE][NW], P0[NE][NE], x0m[NX], u0[NU];
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

/* ---- dense helpers ------------------------------------------------------------------------------------------------------------------ */
/* Work space: one arena per thread, used as a stack (mark / release).  (malloc per solve would put a third-of-a-megabyte block through
 * mmap / munmap every time: with hundreds of threads the kernel's address-space lock becomes the benchmark.) */
#define ARENA_DOUBLES (6u << 20)
static __thread double *arena = NULL;
static __thread size_t arena_top = 0;
static double *vec(size_t n)
{
    if (!arena) arena = (double *)malloc(sizeof(double) * ARENA_DOUBLES);
    if (n == 0) n = 1;
    if (!arena || arena_top + n > ARENA_DOUBLES) abort();
    double *p = arena + arena_top;
    arena_top += n;
    memset(p, 0, sizeof(double) * n);
    return p;
}
static size_t arena_mark(void) { return arena_top; }
static void arena_release(size_t mark) { arena_top = mark; }

/* Householder QR of A' (n x m, m <= n; Jt[i*m+j] = J[j][i]) in place; v's below the diagonal, beta in tau, R on and above */
static int qr_factor(int n, int m, double *A, double *tau)
{
    for (int k = 0; k < m; k++) {
        double nrm = 0.0;
        for (int i = k; i < n; i++) nrm += A[i * m + k] * A[i * m + k];
        nrm = sqrt(nrm);
        if (!(nrm > 0.0)) return 0;
        const double alpha = A[k * m + k] > 0 ? -nrm : nrm;
        const double v0 = A[k * m + k] - alpha;
        double vn = v0 * v0;
        for (int i = k + 1; i < n; i++) vn += A[i * m + k] * A[i * m + k];
        tau[k] = vn > 0.0 ? 2.0 / vn : 0.0;
        A[k * m + k] = v0;
        for (int j = k + 1; j < m; j++) {
            double s = 0.0;
            for (int i = k; i < n; i++) s += A[i * m + k] * A[i * m + j];
            s *= tau[k];
            for (int i = k; i < n; i++) A[i * m + j] -= s * A[i * m + k];
        }
        /* keep v in the column below the diagonal (and v0 on it), R's diagonal entry aside */
        tau[m + k] = alpha;
    }
    return 1;
}
/* y = Q' x or Q x for the factor above (x of length n, in place) */
static void qr_apply(int n, int m, const double *A, const double *tau, double *x, int transpose)
{
    for (int kk = 0; kk < m; kk++) {
        const int k = transpose ? kk : m - 1 - kk;
        double s = 0.0;
        for (int i = k; i < n; i++) s += A[i * m + k] * x[i];
        s *= tau[k];
        for (int i = k; i < n; i++) x[i] -= s * A[i * m + k];
    }
}
static double r_at(int m, const double *A, const double *tau, int i, int j) { return i == j ? tau[m + i] : A[i * m + j]; }      /* R (upper triangular), i <= j */

static int cholesky(int n, double *A)      /* lower factor in place; 0 when not positive definite */
{
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d); A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 1;
}
static void chol_solve(int n, const double *L, double *b)
{
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k]; b[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k]; b[i] = s / L[i * n + i]; }
}
static int inv_small(int n, const double *A, double *inv)      /* Gauss-Jordan with partial pivoting, n <= 8 */
{
    double a[64];
    for (int i = 0; i < n * n; i++) { a[i] = A[i]; inv[i] = 0.0; }
    for (int i = 0; i < n; i++) inv[i * n + i] = 1.0;
    for (int c = 0; c < n; c++) {
        int pv = c;
        for (int r = c + 1; r < n; r++) if (fabs(a[r * n + c]) > fabs(a[pv * n + c])) pv = r;
        if (fabs(a[pv * n + c]) < 1e-300) return 0;
        for (int j = 0; j < n; j++) { double t = a[c * n + j]; a[c * n + j] = a[pv * n + j]; a[pv * n + j] = t; t = inv[c * n + j]; inv[c * n + j] = inv[pv * n + j]; inv[pv * n + j] = t; }
        const double ip = 1.0 / a[c * n + c];
        for (int j = 0; j < n; j++) { a[c * n + j] *= ip; inv[c * n + j] *= ip; }
        for (int r = 0; r < n; r++) if (r != c) { const double f = a[r * n + c]; for (int j = 0; j < n; j++) { a[r * n + j] -= f * a[c * n + j]; inv[r * n + j] -= f * inv[c * n + j]; } }
    }
    return 1;
}

/* ---- the interior point method on  min f(w)  s.t.  g(w) = 0,  lo <= w <= hi  (enmpc_oracle.py:ipm_dense) ---------------------------- */
typedef void (*evalf_t)(void *ctx, const double *w, const double *lam, int want_h, double *f, double *gf, double *g, double *J, double *H);

static double push_in(double v, double lo, double hi)
{
    const int fl = isfinite(lo), fh = isfinite(hi);
    const double gap = (fl && fh) ? KAPPA_PUSH * (hi - lo) : INFINITY;
    if (fl) v = fmax(v, lo + fmin(KAPPA_PUSH * fmax(1.0, fabs(lo)), gap));
    if (fh) v = fmin(v, hi - fmin(KAPPA_PUSH * fmax(1.0, fabs(hi)), gap));
    return v;
}

/* n variables (none fixed: the caller has removed parameters), m equalities */
static int ipm_nullspace(int n, int m, evalf_t evalf, void *ctx, double *w, const double *lo, const double *hi, double tol, int max_iter, int *iters, double *lam_out)
{
    const size_t mark_ = arena_mark();
    double *zl = vec(n), *zh = vec(n), *lam = vec(m), *gf = vec(n), *g = vec(m), *J = vec((size_t)m * n), *H = vec((size_t)n * n), *Jt = vec((size_t)n * m), *tau = vec(2 * m),
           *sl = vec(n), *sh = vec(n), *Sig = vec(n), *gt = vec(n), *dw = vec(n), *lamn = vec(m), *py = vec(n), *tmp = vec(n), *Hr = vec((size_t)(n - m) * (n - m)), *rz = vec(n), *HZ = vec((size_t)n * (n - m)), *Zc = vec(n);
    int nb = 0, status = ST_MAXITER, it = 0;
    for (int i = 0; i < n; i++) { w[i] = push_in(w[i], lo[i], hi[i]); zl[i] = isfinite(lo[i]) ? 1.0 : 0.0; zh[i] = isfinite(hi[i]) ? 1.0 : 0.0; nb += (isfinite(lo[i]) ? 1 : 0) + (isfinite(hi[i]) ? 1 : 0); }
    double mu = MU_INIT, delta_last = 0.0, f;
    const int nz = n - m;
    for (it = 0;; it++) {
        evalf(ctx, w, lam, 1, &f, gf, g, J, H);
        double e_st = 0.0, e_c = 0.0, s_l = 0.0, s_z = 0.0, cmax = -INFINITY, cmin = INFINITY;
        int finite = 1;
        for (int i = 0; i < n; i++) {
            const int fl = isfinite(lo[i]), fh = isfinite(hi[i]);
            sl[i] = fl ? w[i] - lo[i] : 1.0; sh[i] = fh ? hi[i] - w[i] : 1.0;
            double r = gf[i] - zl[i] + zh[i];
            for (int j = 0; j < m; j++) r += J[j * n + i] * lam[j];
            e_st = fmax(e_st, fabs(r)); s_z += zl[i] + zh[i];
            finite = finite && isfinite(r) && isfinite(w[i]);
            if (fl) { cmax = fmax(cmax, sl[i] * zl[i]); cmin = fmin(cmin, sl[i] * zl[i]); }
            if (fh) { cmax = fmax(cmax, sh[i] * zh[i]); cmin = fmin(cmin, sh[i] * zh[i]); }
        }
        for (int j = 0; j < m; j++) { e_c = fmax(e_c, fabs(g[j])); s_l += fabs(lam[j]); finite = finite && isfinite(g[j]); }
        if (!finite) { status = ST_FAILED; break; }
        const double s_d = fmax(S_MAX, (s_l + s_z) / fmax(m + nb, 1.0)) / S_MAX, s_c = fmax(S_MAX, s_z / fmax(nb, 1.0)) / S_MAX;
#define ERR(m_) fmax(fmax(e_st / s_d, e_c), nb > 0 ? fmax(cmax - (m_), (m_) - cmin) / s_c : 0.0)
        if (ERR(0.0) <= tol) { status = ST_SOLVED; break; }
        if (it >= max_iter) break;
        while (mu > tol / 10.0 && ERR(mu) <= KAPPA_EPS * mu) mu = fmax(tol / 10.0, fmin(KAPPA_MU * mu, pow(mu, THETA_MU)));
#undef ERR
        const double tau_f = fmax(TAU_MIN, 1.0 - mu);
        for (int i = 0; i < n; i++) {
            const double il = isfinite(lo[i]) ? 1.0 / sl[i] : 0.0, ih = isfinite(hi[i]) ? 1.0 / sh[i] : 0.0;
            Sig[i] = zl[i] * il + zh[i] * ih; gt[i] = gf[i] - mu * il + mu * ih;
        }
        /* null-space method: J' = Q R;  dw = Y py + Z pz */
        for (int i = 0; i < n; i++) for (int j = 0; j < m; j++) Jt[i * m + j] = J[j * n + i];
        if (m > 0 && !qr_factor(n, m, Jt, tau)) { status = ST_FAILED; break; }
        /* range-space part: R' t = -g (forward substitution), py = Q [t; 0] */
        for (int i = 0; i < n; i++) py[i] = 0.0;
        for (int i = 0; i < m; i++) { double s = -g[i]; for (int k = 0; k < i; k++) s -= r_at(m, Jt, tau, k, i) * py[k]; py[i] = s / r_at(m, Jt, tau, i, i); }
        if (m > 0) qr_apply(n, m, Jt, tau, py, 0);
        double delta = 0.0;
        int failed = 0;
        for (;;) {
            /* HZ = (H + Sig + delta) Z column by column (Z = Q e_{m+c}), reduced Hessian Z' HZ */
            for (int c = 0; c < nz; c++) {
                for (int i = 0; i < n; i++) Zc[i] = 0.0;
                Zc[m + c] = 1.0;
                if (m > 0) qr_apply(n, m, Jt, tau, Zc, 0);
                for (int i = 0; i < n; i++) { double s = (Sig[i] + delta) * Zc[i]; for (int l = 0; l < n; l++) s += H[i * n + l] * Zc[l]; tmp[i] = s; }
                if (m > 0) qr_apply(n, m, Jt, tau, tmp, 1);
                for (int r = 0; r < nz; r++) HZ[r * nz + c] = tmp[m + r];      /* (Z' (H Z))[r][c] */
            }
            for (int r = 0; r < nz; r++) for (int c = 0; c < nz; c++) Hr[r * nz + c] = 0.5 * (HZ[r * nz + c] + HZ[c * nz + r]);
            if (nz == 0 || cholesky(nz, Hr)) break;
            delta = delta == 0.0 ? fmax(DELTA_FIRST, delta_last / 3.0) : delta * (delta_last == 0.0 ? 100.0 : 8.0);
            if (delta > DELTA_MAX) { failed = 1; break; }
        }
        if (failed) { status = ST_FAILED; break; }
        if (delta > 0.0) delta_last = delta;
        /* rz = -Z'((H + Sig + delta) py + gt) */
        for (int i = 0; i < n; i++) { double s = (Sig[i] + delta) * py[i] + gt[i]; for (int l = 0; l < n; l++) s += H[i * n + l] * py[l]; tmp[i] = s; }
        if (m > 0) qr_apply(n, m, Jt, tau, tmp, 1);
        for (int r = 0; r < nz; r++) rz[r] = -tmp[m + r];
        if (nz > 0) chol_solve(nz, Hr, rz);
        for (int i = 0; i < n; i++) dw[i] = 0.0;
        for (int r = 0; r < nz; r++) dw[m + r] = rz[r];
        if (m > 0) qr_apply(n, m, Jt, tau, dw, 0);
        for (int i = 0; i < n; i++) dw[i] += py[i];
        /* multipliers: R lam+ = -Y'((H + Sig + delta) dw + gt) */
        for (int i = 0; i < n; i++) { double s = (Sig[i] + delta) * dw[i] + gt[i]; for (int l = 0; l < n; l++) s += H[i * n + l] * dw[l]; tmp[i] = s; }
        if (m > 0) qr_apply(n, m, Jt, tau, tmp, 1);
        for (int i = m - 1; i >= 0; i--) { double s = -tmp[i]; for (int k = i + 1; k < m; k++) s -= r_at(m, Jt, tau, i, k) * lamn[k]; lamn[i] = s / r_at(m, Jt, tau, i, i); }
        double apr = 1.0, adu = 1.0;
        for (int i = 0; i < n; i++) {
            const int fl = isfinite(lo[i]), fh = isfinite(hi[i]);
            const double dzl = fl ? mu / sl[i] - zl[i] - zl[i] / sl[i] * dw[i] : 0.0, dzh = fh ? mu / sh[i] - zh[i] + zh[i] / sh[i] * dw[i] : 0.0;
            if (fl) { if (dw[i] < 0.0) apr = fmin(apr, -tau_f * sl[i] / dw[i]); if (dzl < 0.0) adu = fmin(adu, -tau_f * zl[i] / dzl); }
            if (fh) { if (-dw[i] < 0.0) apr = fmin(apr, -tau_f * sh[i] / (-dw[i])); if (dzh < 0.0) adu = fmin(adu, -tau_f * zh[i] / dzh); }
            Sig[i] = dzl; gt[i] = dzh;      /* (reused as storage) */
        }
        for (int i = 0; i < n; i++) {
            w[i] += apr * dw[i];
            zl[i] += adu * Sig[i]; zh[i] += adu * gt[i];
            if (isfinite(lo[i])) { const double s = w[i] - lo[i]; zl[i] = fmin(fmax(zl[i], mu / (KAPPA_SIGMA * s)), KAPPA_SIGMA * mu / s); }
            if (isfinite(hi[i])) { const double s = hi[i] - w[i]; zh[i] = fmin(fmax(zh[i], mu / (KAPPA_SIGMA * s)), KAPPA_SIGMA * mu / s); }
        }
        for (int j = 0; j < m; j++) lam[j] += apr * (lamn[j] - lam[j]);
    }
    *iters = it;
    if (lam_out) memcpy(lam_out, lam, sizeof(double) * m);
    arena_release(mark_);
    return status;
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

static void ocp_evalf(void *vctx, const double *w, const double *lam, int want_h, double *f, double *gf, double *g, double *J, double *H)
{
    const OcpCtx *c = (const OcpCtx *)vctx;
    const EProb *P = c->P;
    const int N = P->N, n = NZ * N, m = NX * N;
    memset(gf, 0, sizeof(double) * n); memset(J, 0, sizeof(double) * m * n);
    if (want_h) memset(H, 0, sizeof(double) * n * n);
    *f = 0.0;
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
    const int st = ipm_nullspace(n, m, ocp_evalf, &c, w, lo, hi, P->tol, P->max_iter, iters, NULL);
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
static void tgt_evalf(void *vctx, const double *w, const double *lam, int want_h, double *f, double *gf, double *g, double *J, double *H)
{
    const TgtCtx *c = (const TgtCtx *)vctx;
    const EProb *P = c->P;
    const int n = NV, m = NX + NY;
    memset(gf, 0, sizeof(double) * n); memset(J, 0, sizeof(double) * m * n);
    if (want_h) memset(H, 0, sizeof(double) * n * n);
    cplx zc[NZ], o[NX];
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
static void mhe_evalf(void *vctx, const double *w, const double *lam, int want_h, double *f, double *gf, double *g, double *J, double *H)
{
    const MheCtx *c = (const MheCtx *)vctx;
    const EProb *P = c->P;
    const int N = c->N, n = N * NB + NE, m = N * (NY + NE);
    memset(gf, 0, sizeof(double) * n); memset(J, 0, sizeof(double) * m * n);
    if (want_h) memset(H, 0, sizeof(double) * n * n);
    *f = 0.0;
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
    int nU, nL;
} MheState;

static void mm4(const double *A, const double *B, double *C, int tb) { for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = 0.0; for (int l = 0; l < NE; l++) s += A[i * NE + l] * (tb ? B[j * NE + l] : B[l * NE + j]); C[i * NE + j] = s; } }

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
            for (int i = NE; i < NB; i++) { lo[NB * k + i] = -INFINITY; hi[NB * k + i] = INFINITY; }
            mhe_map(P, xc, S->U[k], wz, xo);
            for (int i = 0; i < NE; i++) xc[i] = xo[i];
        }
    }
    double Pinv[NE * NE];
    int ok = inv_small(NE, S->Pk, Pinv);
    MheCtx c = {P, N, S->U, S->Y, S->xbar, Pinv};
    int st = ipm_nullspace(n, N * (NY + NE), mhe_evalf, &c, w, lo, hi, P->tol_mhe, P->max_iter, iters, NULL);
    if (!ok) st = ST_FAILED;
    const double *Xl = w + NB * (N - 1);
    for (int i = 0; i < NE; i++) xes[i] = Xl[i];
    for (int i = 0; i < NY; i++) S->vk[i] = Xl[NE + i];
    if (ksim != 0) for (int i = 0; i < NW; i++) S->wk[i] = Xl[NE + NY + i];
    /* Kalman quantities (Estimator.py:558-623); the Hessian of the estimator cost is the identity here, so Q_k = I, R_k = I, S_k = 0 */
    double Ak[NE * NE], Ca[NY * NE], K[NE * NY], Sm[NY * NY], Si[NY * NY], Pc[NE * NE], T1[NE * NE], T2[NE * NE];
    cplx zc[NE + NW], oo[NE];
    for (int j = 0; j < NE; j++) {
        for (int i = 0; i < NE; i++) zc[i] = xes[i];
        for (int i = 0; i < NW; i++) zc[NE + i] = S->wk[i];
        zc[j] += I * CS;
        mhe_map(P, zc, u, zc + NE, oo);
        for (int r = 0; r < NE; r++) Ak[r * NE + j] = cimag(oo[r]) / CS;
    }
    for (int r = 0; r < NY; r++) for (int i = 0; i < NE; i++) Ca[r * NE + i] = i < NX ? (r == i) : P->Cd[r][i - NX];
    for (int r = 0; r < NY; r++) for (int q = 0; q < NY; q++) { double s = r == q ? 1.0 : 0.0; for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) s += Ca[r * NE + i] * S->Pkal[i * NE + j] * Ca[q * NE + j]; Sm[r * NY + q] = s; }
    inv_small(NY, Sm, Si);
    for (int i = 0; i < NE; i++) for (int r = 0; r < NY; r++) { double s = 0.0; for (int j = 0; j < NE; j++) for (int q = 0; q < NY; q++) s += S->Pkal[i * NE + j] * Ca[q * NE + j] * Si[q * NY + r]; K[i * NY + r] = s; }
    for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = S->Pkal[i * NE + j]; for (int r = 0; r < NY; r++) for (int l = 0; l < NE; l++) s -= K[i * NY + r] * Ca[r * NE + l] * S->Pkal[l * NE + j]; Pc[i * NE + j] = s; }
    const int idx = ksim < Nm - 1 ? ksim : Nm - 1;
    memcpy(S->bA[idx], Ak, sizeof(Ak)); memcpy(S->bP[idx], S->Pkal, sizeof(Ak)); memcpy(S->bPc[idx], Pc, sizeof(Ak));
    mm4(Ak, Pc, T1, 0); mm4(T1, Ak, T2, 1);      /* A Pc A' */
    for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { double s = T2[i * NE + j]; for (int l = 0; l < NW; l++) s += P->G[i][l] * P->G[j][l]; S->Pkal[i * NE + j] = s; }      /* + G Q G' (S_k = 0: no cross terms) */
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
                     int32_t *st_dyn, int32_t *st_ss, int32_t *st_mhe, int32_t *it_dyn, int32_t *it_ss, int32_t *it_mhe, int nthreads)
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
            double xes[NE];
            int itm, its, itd;
            const int sm = mhe_step(P, S, k, x, u, xes, &itm);      /* y = x_p (StateFeedback plant output) */
            xh[0] = xes[0]; xh[1] = xes[1]; dh[0] = xes[2]; dh[1] = xes[3];
            if (P->has_dsat) for (int i = 0; i < ND; i++) dh[i] = fmin(fmax(dh[i], P->dmin[i]), P->dmax[i]);
            const double xs_prev[NX] = {xs[0], xs[1]}, us_prev = us;
            double v[NV] = {P->x0m[0], P->x0m[1], P->u0[0], P->x0m[0] + P->Cd[0][0] * dh[0] + P->Cd[0][1] * dh[1], P->x0m[1] + P->Cd[1][0] * dh[0] + P->Cd[1][1] * dh[1]};
            TgtCtx tc = {P, dh};
            const int ss = ipm_nullspace(NV, NX + NY, tgt_evalf, &tc, v, P->tlo, P->thi, P->tol, P->max_iter, &its, NULL);
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
