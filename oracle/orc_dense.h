/* ORACLE (test infrastructure, never shipped): dense linear algebra and the interior point method shared by the C restatements
 * (enmpc_oracle.c, nmpc_oracle.c).  Everything static: each restatement is one translation unit.
 *
 * ipm_ipopt: primal-dual interior point method on  min f(w)  s.t.  g(w) = 0,  lo <= w <= hi  - the algorithm documented in
 * enmpc_oracle.py:ipm_dense (the reference solver's, IPOPT at the reference's options: scaling, least-squares multipliers, monotone barrier
 * parameter, filter line search with second-order correction, safe slacks), Newton steps by a NULL-SPACE method: Householder QR of the constraint
 * Jacobian, Cholesky of the reduced Hessian (its failure is the inertia test).  Used by enmpc_oracle.c.
 * ipm_nullspace: the plain full-step form of it (no line search, no scaling) that nmpc_oracle.c solves its convex QPs with. */
#ifndef ORC_DENSE_H
#define ORC_DENSE_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ST_SOLVED = 0, ST_MAXITER = 1, ST_FAILED = 2 };
#define KAPPA_PUSH 1e-2
#define BOUND_RELAX_FACTOR 1e-8
#define MU_INIT 0.1
#define KAPPA_EPS 10.0
#define KAPPA_MU 0.2
#define THETA_MU 1.5
#define TAU_MIN 0.99
#define KAPPA_SIGMA 1e10
#define S_MAX 100.0
#define DELTA_FIRST 1e-4
#define DELTA_MAX 1e40

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
static int ipm_nullspace(int n, int m, evalf_t evalf, void *ctx, double *w, const double *lo_in, const double *hi_in, double tol, int max_iter, int *iters, double *lam_out)
{
    const size_t mark_ = arena_mark();
    const double *lo = lo_in, *hi = hi_in;
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


/* ---- the reference solver's algorithm (enmpc_oracle.py:ipm_dense, same constants, same order of decisions) ---------------------------- */
#define ORC_EPS 2.220446049250313e-16
#define SLACK_MOVE 1.81898940354585648e-12      /* eps^(3/4) */
#define SCALE_MAX_GRAD 100.0
#define SCALE_MIN 1e-8
#define Y_INIT_MAX 1e3
#define KAPPA_D 1e-5
#define GAMMA_THETA 1e-5
#define GAMMA_PHI 1e-8
#define S_THETA 1.1
#define S_PHI 2.3
#define ETA_PHI 1e-8
#define THETA_MAX_FACT 1e4
#define THETA_MIN_FACT 1e-4
#define ALPHA_MIN_FRAC 0.05
#define OBJ_MAX_INC 5.0
#define MAX_SOC 4
#define KAPPA_SOC 0.99
#define TINY_STEP_TOL (10.0 * ORC_EPS)
#define TINY_STEP_Y_TOL 1e-2
#define DUAL_INF_TOL 1.0
#define CONSTR_VIOL_TOL 1e-4
#define COMPL_INF_TOL 1e-4
#define ACC_TOL 1e-6
#define ACC_ITER 15
#define ACC_DUAL_INF_TOL 1e10
#define ACC_CONSTR_VIOL_TOL 1e-2
#define ACC_COMPL_INF_TOL 1e-2
#define FILTER_CAP 16
enum { WANT_VALUES = 0, WANT_GRAD = 1, WANT_HESS = 2 };
/* evalf2(ctx, w, lam, want, ...): want = WANT_VALUES: f and g only; WANT_GRAD: + gf, J; WANT_HESS: + H = Hessian of f + lam'g */
typedef void (*evalf2_t)(void *ctx, const double *w, const double *lam, int want, double *f, double *gf, double *g, double *J, double *H);

static inline int orc_le(double lhs, double rhs, double bas) { return lhs - rhs <= 10.0 * ORC_EPS * fabs(bas); }
/* slack of one bound with IPOPT's CalculateSafeSlack; a corrected slack moves *bound */
static inline double safe_slack(double w, double *bound, double z, double mu, int lower)
{
    double s = lower ? w - *bound : *bound - w;
    const double s_min = ORC_EPS * fmin(1.0, mu);
    if (s < s_min) {
        s = fmin(fmax(mu / z, s_min), fmax(s, 0.0) + SLACK_MOVE * fmax(1.0, fabs(*bound)));
        *bound = lower ? w - s : w + s;
    }
    return s;
}

typedef struct {
    int n, m, nz;
    double *Jt, *tau, *H, *Sig, *Hr, *gphi;      /* QR of J', Hessian, Sigma, Cholesky factor of the reduced Hessian (with delta), gradient of the barrier function */
    double delta;
    double *py, *tmp, *rz;
} NsSys;
/* Newton step of the primal-dual system for the constraint right-hand side c: dw and the new equality multipliers lamn */
static void ns_direction(const NsSys *S, const double *c, double *dw, double *lamn)
{
    const int n = S->n, m = S->m, nz = S->nz;
    double *py = S->py, *tmp = S->tmp, *rz = S->rz;
    for (int i = 0; i < n; i++) py[i] = 0.0;
    for (int i = 0; i < m; i++) { double s = -c[i]; for (int k = 0; k < i; k++) s -= r_at(m, S->Jt, S->tau, k, i) * py[k]; py[i] = s / r_at(m, S->Jt, S->tau, i, i); }
    if (m > 0) qr_apply(n, m, S->Jt, S->tau, py, 0);
    for (int i = 0; i < n; i++) { double s = (S->Sig[i] + S->delta) * py[i] + S->gphi[i]; for (int l = 0; l < n; l++) s += S->H[i * n + l] * py[l]; tmp[i] = s; }
    if (m > 0) qr_apply(n, m, S->Jt, S->tau, tmp, 1);
    for (int r = 0; r < nz; r++) rz[r] = -tmp[m + r];
    if (nz > 0) chol_solve(nz, S->Hr, rz);
    for (int i = 0; i < n; i++) dw[i] = 0.0;
    for (int r = 0; r < nz; r++) dw[m + r] = rz[r];
    if (m > 0) qr_apply(n, m, S->Jt, S->tau, dw, 0);
    for (int i = 0; i < n; i++) dw[i] += py[i];
    for (int i = 0; i < n; i++) { double s = (S->Sig[i] + S->delta) * dw[i] + S->gphi[i]; for (int l = 0; l < n; l++) s += S->H[i * n + l] * dw[l]; tmp[i] = s; }
    if (m > 0) qr_apply(n, m, S->Jt, S->tau, tmp, 1);
    for (int i = m - 1; i >= 0; i--) { double s = -tmp[i]; for (int k = i + 1; k < m; k++) s -= r_at(m, S->Jt, S->tau, i, k) * lamn[k]; lamn[i] = s / r_at(m, S->Jt, S->tau, i, i); }
}

typedef struct { int ls_steps, soc, tiny, filter_max, stop, resto, resto_iters; double df; } IpmInfo;
/* stop: 0 converged, 1 acceptable, 2 iteration limit, 3 tiny step, 4 line search failed at a feasible point, 5 restoration needed, 6 not finite, 7 no curvature,
   8 restored (the restoration phase's own test), 9 restoration converged to a feasible point, 10 locally infeasible, 11 restoration failed */
#define RESTO_RHO 1000.0
#define RESTO_KAPPA 0.9
#define RESTO_THETA_MAX_FACT 1e8
#define RESTO_BOUND_MULT_RESET 1e3
#define RESTO_FEAS_FACT 1e2

/* a problem as the iteration sees it: ev evaluates objective (possibly depending on mu: the restoration problem's does), constraints and derivatives */
typedef struct IpmProb {
    int n, m;
    void (*ev)(struct IpmProb *P, const double *w, const double *lam, double mu, int want, double *f, double *gf, double *g, double *J, double *H);
    int (*hook)(struct IpmProb *P, const double *w);      /* ends the solve when it returns 1 (the restoration phase's test), or NULL */
    void *ctx;
} IpmProb;

/* what the restoration phase's test needs of the iteration that called it */
typedef struct {
    IpmProb *P; int n, m; double mu, theta, phi; const double *lo, *hi, *zl, *zh; const unsigned char *fl, *fh; const double *filt_phi, *filt_th; int nfilt; double *gt, *gdum, *J, *H;
} OrigRef;

static double barrier_of(int n, const double *w, double f, const double *lo, const double *hi, const double *zl, const double *zh, const unsigned char *fl, const unsigned char *fh, double mu)
{
    double phi = f;
    for (int i = 0; i < n; i++) {
        double bl = lo[i], bh = hi[i];
        const double s1 = fl[i] ? safe_slack(w[i], &bl, zl[i], mu, 1) : 1.0, s2 = fh[i] ? safe_slack(w[i], &bh, zh[i], mu, 0) : 1.0;
        if (fl[i]) phi -= mu * log(s1);
        if (fh[i]) phi -= mu * log(s2);
        if (fl[i] && !fh[i]) phi += KAPPA_D * mu * s1;
        if (fh[i] && !fl[i]) phi += KAPPA_D * mu * s2;
    }
    return phi;
}
static int to_iterate(double theta, double phi, double theta_t, double phi_t, int from_resto)
{
    if (!from_resto && phi_t > phi) { const double bas = fabs(phi) > 10.0 ? log10(fabs(phi)) : 1.0; if (log10(phi_t - phi) > OBJ_MAX_INC + bas) return 0; }
    return orc_le(theta_t, (1.0 - GAMMA_THETA) * theta, theta) || orc_le(phi_t - phi, -GAMMA_PHI * theta, phi);
}
static int to_filter(const double *fp, const double *ft, int nf, double theta_t, double phi_t)
{
    for (int e = 0; e < nf; e++) if (orc_le(fp[e], phi_t, fp[e]) && orc_le(ft[e], theta_t, ft[e])) return 0;
    return 1;
}
static int resto_hook(IpmProb *R, const double *wb);
static int ipm_restore(IpmProb *P, OrigRef *O, const double *x_r, double *lo, double *hi, const double *zl, const double *zh, double mu, const double *c_r, double tol, int max_iter, int it0,
                       double *x_out, int *it_out, IpmInfo *inf);

/* the iteration from a given first iterate (w, zl, zh, lam, mu); lo / hi are this solve's own bounds (moved by the safe slack) */
static int ipm_core(IpmProb *P, double *w, double *lo, double *hi, double *zl, double *zh, double *lam, double mu, double tol, int max_iter, int it0, double theta_max_fact, int resto_ok,
                    int *it_out, IpmInfo *inf)
{
    const size_t mark_ = arena_mark();
    const int n = P->n, m = P->m, nz = n - m;
    double *lot = vec(n), *hit = vec(n);
    double *gf = vec(n), *g = vec(m), *J = vec((size_t)m * n), *H = vec((size_t)n * n), *Jt = vec((size_t)n * m), *tau = vec(2 * m + 2),
           *sl = vec(n), *sh = vec(n), *slt = vec(n), *sht = vec(n), *Sig = vec(n), *gphi = vec(n), *dw = vec(n), *ds = vec(n), *lamn = vec(m), *lamsoc = vec(m), *Hr = vec((size_t)(nz > 0 ? nz : 1) * (nz > 0 ? nz : 1)),
           *HZ = vec((size_t)n * (nz > 0 ? nz : 1)), *Zc = vec(n), *tmp = vec(n), *wt = vec(n), *gt = vec(m), *csoc = vec(m), *ctr = vec(m), *gdum = vec(n), *xr = vec(n);
    unsigned char *fl = (unsigned char *)vec((n + 7) / 8 + 1), *fh = (unsigned char *)vec((n + 7) / 8 + 1);
    NsSys S = {n, m, nz, Jt, tau, H, Sig, Hr, gphi, 0.0, vec(n), vec(n), vec(n)};
    int nb = 0, status = ST_MAXITER, it = it0;
    double f, ft;
    for (int i = 0; i < n; i++) { fl[i] = isfinite(lo[i]) ? 1 : 0; fh[i] = isfinite(hi[i]) ? 1 : 0; nb += fl[i] + fh[i]; }
    double tau_f = fmax(TAU_MIN, 1.0 - mu), delta_last = 0.0, theta_max = -1.0, theta_min = -1.0;
    double filt_phi[FILTER_CAP], filt_th[FILTER_CAP];
    int nfilt = 0, acc_count = 0, tiny_last = 0, tiny_flag = 0;
    const double mu_min = fmin(tol, COMPL_INF_TOL) / (KAPPA_EPS + 1.0);
    inf->stop = 2;
    for (;;) {
        P->ev(P, w, lam, mu, WANT_HESS, &f, gf, g, J, H);
        double e_st = 0.0, e_c = 0.0, s_l = 0.0, s_z = 0.0, cmax = -INFINITY, cmin = INFINITY, theta = 0.0;
        int finite = isfinite(f);
        for (int i = 0; i < n; i++) {
            sl[i] = fl[i] ? safe_slack(w[i], &lo[i], zl[i], mu, 1) : 1.0; sh[i] = fh[i] ? safe_slack(w[i], &hi[i], zh[i], mu, 0) : 1.0;
            double r = gf[i] - zl[i] + zh[i];
            for (int j = 0; j < m; j++) r += J[j * n + i] * lam[j];
            e_st = fmax(e_st, fabs(r)); s_z += zl[i] + zh[i];
            finite = finite && isfinite(r) && isfinite(w[i]) && isfinite(gf[i]);
            if (fl[i]) { cmax = fmax(cmax, sl[i] * zl[i]); cmin = fmin(cmin, sl[i] * zl[i]); }
            if (fh[i]) { cmax = fmax(cmax, sh[i] * zh[i]); cmin = fmin(cmin, sh[i] * zh[i]); }
        }
        for (int j = 0; j < m; j++) { e_c = fmax(e_c, fabs(g[j])); theta += fabs(g[j]); s_l += fabs(lam[j]); finite = finite && isfinite(g[j]); }
        if (!finite) { status = ST_FAILED; inf->stop = 6; break; }
        if (P->hook && P->hook(P, w)) { status = ST_SOLVED; inf->stop = 8; break; }
        const double s_d = fmax(S_MAX, (s_l + s_z) / fmax(m + nb, 1.0)) / S_MAX, s_c = fmax(S_MAX, s_z / fmax(nb, 1.0)) / S_MAX;
#define COMPL(m_) (nb > 0 ? fmax(cmax - (m_), (m_) - cmin) : 0.0)
#define ERR(m_) fmax(fmax(e_st / s_d, e_c), COMPL(m_) / s_c)
        const double e0 = ERR(0.0), c0 = COMPL(0.0);
        if (e0 <= tol && e_st <= DUAL_INF_TOL && e_c <= CONSTR_VIOL_TOL && c0 <= COMPL_INF_TOL) { status = ST_SOLVED; inf->stop = 0; break; }
        if (e0 <= ACC_TOL && e_st <= ACC_DUAL_INF_TOL && e_c <= ACC_CONSTR_VIOL_TOL && c0 <= ACC_COMPL_INF_TOL) {
            if (++acc_count >= ACC_ITER) { status = ST_SOLVED; inf->stop = 1; break; }
        } else acc_count = 0;
        if (it >= max_iter) { inf->stop = 2; break; }
        /* barrier parameter */
        int mu_changed = 0, stop_tiny = 0;
        while (ERR(mu) <= KAPPA_EPS * mu || tiny_flag) {
            const double new_mu = fmax(fmin(KAPPA_MU * mu, pow(mu, THETA_MU)), mu_min);
            if (new_mu == mu) { stop_tiny = tiny_flag; break; }
            mu = new_mu; mu_changed = 1; tiny_flag = 0;
        }
#undef ERR
#undef COMPL
        if (stop_tiny) { status = ST_MAXITER; inf->stop = 3; break; }
        tiny_flag = 0;
        if (mu_changed) {
            nfilt = 0; tau_f = fmax(TAU_MIN, 1.0 - mu);
            if (P->hook) P->ev(P, w, lam, mu, WANT_HESS, &f, gf, g, J, H);      /* (the restoration problem's objective changes with mu) */
        }
        /* search direction */
        double phi = f, gbd = 0.0;
        for (int i = 0; i < n; i++) {
            const double il = fl[i] ? 1.0 / sl[i] : 0.0, ih = fh[i] ? 1.0 / sh[i] : 0.0;
            const int ol = fl[i] && !fh[i], oh = fh[i] && !fl[i];
            Sig[i] = zl[i] * il + zh[i] * ih; gphi[i] = gf[i] - mu * il + mu * ih + KAPPA_D * mu * ((ol ? 1.0 : 0.0) - (oh ? 1.0 : 0.0));
            if (fl[i]) phi -= mu * log(sl[i]);
            if (fh[i]) phi -= mu * log(sh[i]);
            if (ol) phi += KAPPA_D * mu * sl[i];
            if (oh) phi += KAPPA_D * mu * sh[i];
        }
        for (int i = 0; i < n; i++) for (int j = 0; j < m; j++) Jt[i * m + j] = J[j * n + i];
        if (m > 0 && !qr_factor(n, m, Jt, tau)) { status = ST_FAILED; inf->stop = 7; break; }
        double delta = 0.0;
        int failed = 0;
        for (;;) {
            for (int c = 0; c < nz; c++) {
                for (int i = 0; i < n; i++) Zc[i] = 0.0;
                Zc[m + c] = 1.0;
                if (m > 0) qr_apply(n, m, Jt, tau, Zc, 0);
                for (int i = 0; i < n; i++) { double s = (Sig[i] + delta) * Zc[i]; for (int l = 0; l < n; l++) s += H[i * n + l] * Zc[l]; tmp[i] = s; }
                if (m > 0) qr_apply(n, m, Jt, tau, tmp, 1);
                for (int r = 0; r < nz; r++) HZ[r * nz + c] = tmp[m + r];
            }
            for (int r = 0; r < nz; r++) for (int c = 0; c < nz; c++) Hr[r * nz + c] = 0.5 * (HZ[r * nz + c] + HZ[c * nz + r]);
            if (nz == 0 || cholesky(nz, Hr)) break;
            delta = delta == 0.0 ? fmax(DELTA_FIRST, delta_last / 3.0) : delta * (delta_last == 0.0 ? 100.0 : 8.0);
            if (delta > DELTA_MAX) { failed = 1; break; }
        }
        if (failed) { status = ST_FAILED; inf->stop = 7; break; }
        if (delta > 0.0) delta_last = delta;
        S.delta = delta;
        ns_direction(&S, g, dw, lamn);
        double a_max = 1.0, dmaxrel = 0.0, dymax = 0.0;
        for (int i = 0; i < n; i++) {
            if (fl[i] && dw[i] < 0.0) a_max = fmin(a_max, -tau_f * sl[i] / dw[i]);
            if (fh[i] && -dw[i] < 0.0) a_max = fmin(a_max, -tau_f * sh[i] / (-dw[i]));
            gbd += gphi[i] * dw[i];
            dmaxrel = fmax(dmaxrel, fabs(dw[i]) / (1.0 + fabs(w[i])));
        }
        for (int j = 0; j < m; j++) dymax = fmax(dymax, fabs(lamn[j] - lam[j]));
        /* filter line search */
        double a_min = GAMMA_THETA;
        if (gbd < 0.0) {
            a_min = fmin(GAMMA_THETA, GAMMA_PHI * theta / (-gbd));
            if (theta <= theta_min) a_min = fmin(a_min, pow(theta, S_THETA) / pow(-gbd, S_PHI));
        }
        a_min *= ALPHA_MIN_FRAC;
        if (theta_max < 0.0) { theta_max = theta_max_fact * fmax(1.0, theta); theta_min = THETA_MIN_FACT * fmax(1.0, theta); }
        double theta_t = 0.0, phi_t = 0.0;
        int ok_t = 0;
#define TRIAL(alpha_, d_) do { \
            for (int i_ = 0; i_ < n; i_++) { wt[i_] = w[i_] + (alpha_) * (d_)[i_]; lot[i_] = lo[i_]; hit[i_] = hi[i_]; } \
            P->ev(P, wt, lam, mu, WANT_VALUES, &ft, gdum, gt, J, H); \
            ok_t = isfinite(ft); theta_t = 0.0; phi_t = ft; \
            for (int j_ = 0; j_ < m; j_++) { theta_t += fabs(gt[j_]); ok_t = ok_t && isfinite(gt[j_]); } \
            for (int i_ = 0; i_ < n; i_++) { \
                slt[i_] = fl[i_] ? safe_slack(wt[i_], &lot[i_], zl[i_], mu, 1) : 1.0; sht[i_] = fh[i_] ? safe_slack(wt[i_], &hit[i_], zh[i_], mu, 0) : 1.0; \
                if (fl[i_]) phi_t -= mu * log(slt[i_]); \
                if (fh[i_]) phi_t -= mu * log(sht[i_]); \
                if (fl[i_] && !fh[i_]) phi_t += KAPPA_D * mu * slt[i_]; \
                if (fh[i_] && !fl[i_]) phi_t += KAPPA_D * mu * sht[i_]; \
            } \
            if (!ok_t) { theta_t = INFINITY; phi_t = INFINITY; } \
        } while (0)
#define FTYPE(alpha_) ((theta == 0.0 && gbd > 0.0 && gbd < 100.0 * ORC_EPS) || (gbd < 0.0 && (alpha_) * pow(-gbd, S_PHI) > pow(theta, S_THETA)))
#define ARMIJO(alpha_) orc_le(phi_t - phi, ETA_PHI * (alpha_) * gbd, phi)
#define ACCEPTABLE(alpha_) (ok_t && !(theta_t > theta_max) && ((((alpha_) > 0.0 && FTYPE(alpha_) && theta <= theta_min) ? ARMIJO(alpha_) : to_iterate(theta, phi, theta_t, phi_t, 0)) && to_filter(filt_phi, filt_th, nfilt, theta_t, phi_t)))
#define AUGMENT() do { \
            const double e_phi = phi - GAMMA_PHI * theta, e_th = (1.0 - GAMMA_THETA) * theta; \
            int k2 = 0; \
            for (int e = 0; e < nfilt; e++) if (!(filt_phi[e] >= e_phi && filt_th[e] >= e_th)) { filt_phi[k2] = filt_phi[e]; filt_th[k2] = filt_th[e]; k2++; } \
            nfilt = k2; \
            if (nfilt >= FILTER_CAP) { filt_phi[nfilt - 1] = fmin(filt_phi[nfilt - 1], e_phi); filt_th[nfilt - 1] = fmin(filt_th[nfilt - 1], e_th); } \
            else { filt_phi[nfilt] = e_phi; filt_th[nfilt] = e_th; nfilt++; } \
            if (nfilt > inf->filter_max) inf->filter_max = nfilt; \
        } while (0)
        int accepted = 0, soc_taken = 0, restored = 0;
        double alpha = a_max, a_soc = a_max;
        int tiny = dmaxrel < TINY_STEP_TOL && theta <= 1e-4;
        if (tiny) {
            TRIAL(a_max, dw);
            if (ok_t) { accepted = 1; inf->tiny++; tiny_flag = tiny_last; tiny_last = dymax < TINY_STEP_Y_TOL; }
            else tiny = 0;
        }
        if (!tiny) {
            tiny_last = 0;
            int n_steps = 0;
            while (alpha > a_min || n_steps == 0) {
                TRIAL(alpha, dw);
                if (ACCEPTABLE(alpha)) { accepted = 1; break; }
                if (ok_t && n_steps == 0 && theta <= theta_t) {      /* second-order correction */
                    memcpy(csoc, g, sizeof(double) * m);
                    memcpy(ctr, gt, sizeof(double) * m);
                    double theta_old = 0.0, th_s = theta_t;
                    int cnt = 0;
                    a_soc = alpha;
                    while (cnt < MAX_SOC && !accepted && (cnt == 0 || th_s <= KAPPA_SOC * theta_old)) {
                        theta_old = th_s;
                        for (int j = 0; j < m; j++) csoc[j] = a_soc * csoc[j] + ctr[j];
                        ns_direction(&S, csoc, ds, lamsoc);
                        a_soc = 1.0;
                        for (int i = 0; i < n; i++) {
                            if (fl[i] && ds[i] < 0.0) a_soc = fmin(a_soc, -tau_f * sl[i] / ds[i]);
                            if (fh[i] && -ds[i] < 0.0) a_soc = fmin(a_soc, -tau_f * sh[i] / (-ds[i]));
                        }
                        TRIAL(a_soc, ds);
                        inf->soc++;
                        if (ACCEPTABLE(alpha)) { accepted = 1; soc_taken = 1; }      /* (the tests keep the original step length) */
                        else { cnt++; th_s = theta_t; memcpy(ctr, gt, sizeof(double) * m); if (!ok_t) break; }
                    }
                    if (accepted) break;
                }
                alpha *= 0.5;
                n_steps++;
            }
            inf->ls_steps += n_steps;
            if (!accepted) {      /* IPOPT's restoration phase */
                if (theta <= 1e-2 * tol) { status = ST_MAXITER; inf->stop = 4; break; }
                if (!resto_ok || P->hook) { status = ST_FAILED; inf->stop = 5; break; }      /* (not restated for this problem / none inside the restoration phase) */
                AUGMENT();
                inf->resto++;
                OrigRef O = {P, n, m, mu, theta, phi, lo, hi, zl, zh, fl, fh, filt_phi, filt_th, nfilt, gt, gdum, J, H};
                int it_r = it + 1;
                IpmInfo sub = {0, 0, 0, 0, 2, 0, 0, inf->df};
                const int rs = ipm_restore(P, &O, w, lo, hi, zl, zh, mu, g, tol, max_iter, it + 1, xr, &it_r, &sub);
                inf->resto_iters += it_r - (it + 1);
                it = it_r - 1;
                if (rs != 8) {
                    if (rs == 2) { status = ST_MAXITER; inf->stop = 2; }
                    else if (rs == 0 || rs == 1 || rs == 3) {      /* the restoration problem has a minimiser here: infeasible, or feasible and not acceptable */
                        P->ev(P, xr, lam, mu, WANT_VALUES, &ft, gdum, gt, J, H);
                        double cm = 0.0;
                        for (int j = 0; j < m; j++) cm = fmax(cm, fabs(gt[j]));
                        if (cm <= RESTO_FEAS_FACT * tol) { status = ST_MAXITER; inf->stop = 9; } else { status = ST_FAILED; inf->stop = 10; }
                    } else { status = ST_MAXITER; inf->stop = 11; }
                    break;
                }
                /* back from the restoration: bound multipliers as if the whole move had been one Newton step, reset to 1 beyond 1e3; equality multipliers zero */
                double adu = 1.0, zmax = 0.0;
                for (int i = 0; i < n; i++) {
                    lot[i] = lo[i]; hit[i] = hi[i];
                    slt[i] = fl[i] ? safe_slack(xr[i], &lot[i], zl[i], mu, 1) : 1.0; sht[i] = fh[i] ? safe_slack(xr[i], &hit[i], zh[i], mu, 0) : 1.0;
                    const double dzl = fl[i] ? mu / sl[i] - zl[i] - zl[i] / sl[i] * (slt[i] - sl[i]) : 0.0, dzh = fh[i] ? mu / sh[i] - zh[i] - zh[i] / sh[i] * (sht[i] - sh[i]) : 0.0;
                    if (fl[i] && dzl < 0.0) adu = fmin(adu, -tau_f * zl[i] / dzl);
                    if (fh[i] && dzh < 0.0) adu = fmin(adu, -tau_f * zh[i] / dzh);
                    Sig[i] = dzl; gphi[i] = dzh;
                }
                for (int i = 0; i < n; i++) { zl[i] += adu * Sig[i]; zh[i] += adu * gphi[i]; zmax = fmax(zmax, fmax(zl[i], zh[i])); }
                for (int i = 0; i < n; i++) {
                    if (zmax > RESTO_BOUND_MULT_RESET) { zl[i] = fl[i] ? 1.0 : 0.0; zh[i] = fh[i] ? 1.0 : 0.0; }
                    w[i] = xr[i]; lo[i] = lot[i]; hi[i] = hit[i];
                    if (fl[i]) zl[i] = fmin(fmax(zl[i], mu / (KAPPA_SIGMA * slt[i])), KAPPA_SIGMA * mu / slt[i]);
                    if (fh[i]) zh[i] = fmin(fmax(zh[i], mu / (KAPPA_SIGMA * sht[i])), KAPPA_SIGMA * mu / sht[i]);
                }
                for (int j = 0; j < m; j++) lam[j] = 0.0;
                restored = 1;
            } else if (!FTYPE(alpha) || !ARMIJO(alpha)) AUGMENT();      /* the filter grows unless the step was an Armijo step on the barrier function */
        }
#undef TRIAL
#undef FTYPE
#undef ARMIJO
#undef ACCEPTABLE
#undef AUGMENT
        if (restored) { it++; continue; }
        /* the accepted point (wt, slt, sht, lot, hit): multiplier steps of the direction that was taken */
        const double *dacc = soc_taken ? ds : dw, *lacc = soc_taken ? lamsoc : lamn;
        const double a_pr = soc_taken ? a_soc : alpha;
        double adu = 1.0;
        for (int i = 0; i < n; i++) {
            const double dzl = fl[i] ? mu / sl[i] - zl[i] - zl[i] / sl[i] * dacc[i] : 0.0, dzh = fh[i] ? mu / sh[i] - zh[i] + zh[i] / sh[i] * dacc[i] : 0.0;
            if (fl[i] && dzl < 0.0) adu = fmin(adu, -tau_f * zl[i] / dzl);
            if (fh[i] && dzh < 0.0) adu = fmin(adu, -tau_f * zh[i] / dzh);
            Sig[i] = dzl; gphi[i] = dzh;      /* (reused as storage) */
        }
        for (int i = 0; i < n; i++) {
            w[i] = wt[i]; lo[i] = lot[i]; hi[i] = hit[i];
            zl[i] += adu * Sig[i]; zh[i] += adu * gphi[i];
            if (fl[i]) zl[i] = fmin(fmax(zl[i], mu / (KAPPA_SIGMA * slt[i])), KAPPA_SIGMA * mu / slt[i]);
            if (fh[i]) zh[i] = fmin(fmax(zh[i], mu / (KAPPA_SIGMA * sht[i])), KAPPA_SIGMA * mu / sht[i]);
        }
        for (int j = 0; j < m; j++) lam[j] += a_pr * (lacc[j] - lam[j]);
        it++;
    }
    *it_out = it;
    arena_release(mark_);
    return status;
}

/* ---- the restoration phase: the same iteration on  min rho sum(n + p) + eta(mu) / 2 |D_R (x - x_R)|^2  s.t.  c(x) + n - p = 0, bounds, n, p >= 0 ---- */
typedef struct { IpmProb *P; OrigRef *O; const double *x_r, *d_r; int n, m; double *Ja, *Ha, *Hb, *ga, *zero; } RestoCtx;
static void resto_ev(IpmProb *R, const double *wb, const double *lam, double mu, int want, double *f, double *gf, double *g, double *J, double *H)
{
    RestoCtx *C = (RestoCtx *)R->ctx;
    const int n = C->n, m = C->m, nb = n + 2 * m;
    const double eta = sqrt(mu);
    double fo;
    C->P->ev(C->P, wb, lam, mu, want == WANT_VALUES ? WANT_VALUES : WANT_GRAD, &fo, C->ga, g, C->Ja, C->Ha);
    double s = 0.0, q = 0.0;
    for (int j = 0; j < m; j++) { s += wb[n + j] + wb[n + m + j]; g[j] += wb[n + j] - wb[n + m + j]; }
    for (int i = 0; i < n; i++) { const double e = C->d_r[i] * (wb[i] - C->x_r[i]); q += e * e; }
    *f = RESTO_RHO * s + 0.5 * eta * q;
    if (want == WANT_VALUES) return;
    for (int i = 0; i < n; i++) gf[i] = eta * C->d_r[i] * C->d_r[i] * (wb[i] - C->x_r[i]);
    for (int i = n; i < nb; i++) gf[i] = RESTO_RHO;
    memset(J, 0, sizeof(double) * m * nb);
    for (int j = 0; j < m; j++) { for (int i = 0; i < n; i++) J[j * nb + i] = C->Ja[j * n + i]; J[j * nb + n + j] = 1.0; J[j * nb + n + m + j] = -1.0; }
    if (want != WANT_HESS) return;
    memset(H, 0, sizeof(double) * nb * nb);
    C->P->ev(C->P, wb, lam, mu, WANT_HESS, &fo, C->ga, C->zero + m, C->Ja, C->Ha);        /* Hessian of (objective + lam'c) ... */
    C->P->ev(C->P, wb, C->zero, mu, WANT_HESS, &fo, C->ga, C->zero + m, C->Ja, C->Hb);    /* ... minus that of the objective alone: lam'c */
    for (int i = 0; i < m; i++) C->zero[m + i] = 0.0;
    for (int i = 0; i < n; i++) for (int l = 0; l < n; l++) H[i * nb + l] = C->Ha[i * n + l] - C->Hb[i * n + l] + (i == l ? eta * C->d_r[i] * C->d_r[i] : 0.0);
}
static int resto_hook(IpmProb *R, const double *wb)
{
    RestoCtx *C = (RestoCtx *)R->ctx;
    OrigRef *O = C->O;
    double ft, th = 0.0;
    O->P->ev(O->P, wb, C->zero, O->mu, WANT_VALUES, &ft, O->gdum, O->gt, O->J, O->H);
    int ok = isfinite(ft);
    for (int j = 0; j < O->m; j++) { th += fabs(O->gt[j]); ok = ok && isfinite(O->gt[j]); }
    if (!ok || th > RESTO_KAPPA * O->theta) return 0;
    const double ph = barrier_of(O->n, wb, ft, O->lo, O->hi, O->zl, O->zh, O->fl, O->fh, O->mu);
    return to_filter(O->filt_phi, O->filt_th, O->nfilt, th, ph) && to_iterate(O->theta, O->phi, th, ph, 1);
}
static int ipm_restore(IpmProb *P, OrigRef *O, const double *x_r, double *lo, double *hi, const double *zl, const double *zh, double mu, const double *c_r, double tol, int max_iter, int it0,
                       double *x_out, int *it_out, IpmInfo *inf)
{
    const size_t mark_ = arena_mark();
    const int n = P->n, m = P->m, nb = n + 2 * m;
    double *wb = vec(nb), *lob = vec(nb), *hib = vec(nb), *zlb = vec(nb), *zhb = vec(nb), *lam = vec(m), *d_r = vec(n);
    double mu_r = mu;
    for (int j = 0; j < m; j++) mu_r = fmax(mu_r, fabs(c_r[j]));
    for (int i = 0; i < n; i++) { wb[i] = x_r[i]; d_r[i] = 1.0 / fmax(1.0, fabs(x_r[i])); lob[i] = lo[i]; hib[i] = hi[i]; zlb[i] = isfinite(lo[i]) ? fmin(RESTO_RHO, zl[i]) : 0.0; zhb[i] = isfinite(hi[i]) ? fmin(RESTO_RHO, zh[i]) : 0.0; }
    for (int j = 0; j < m; j++) {
        const double a = mu_r / (2.0 * RESTO_RHO) - 0.5 * c_r[j], nn = a + sqrt(a * a + mu_r * c_r[j] / (2.0 * RESTO_RHO)), pp = c_r[j] + nn;
        wb[n + j] = nn; wb[n + m + j] = pp;
        lob[n + j] = 0.0; lob[n + m + j] = 0.0; hib[n + j] = INFINITY; hib[n + m + j] = INFINITY;
        zlb[n + j] = mu_r / nn; zlb[n + m + j] = mu_r / pp; zhb[n + j] = 0.0; zhb[n + m + j] = 0.0;
    }
    RestoCtx C = {P, O, x_r, d_r, n, m, vec((size_t)m * n), vec((size_t)n * n), vec((size_t)n * n), vec(n), vec(2 * m)};
    IpmProb R = {nb, m, resto_ev, resto_hook, &C};
    (void)ipm_core(&R, wb, lob, hib, zlb, zhb, lam, mu_r, tol, max_iter, it0, RESTO_THETA_MAX_FACT, 0, it_out, inf);
    for (int i = 0; i < n; i++) { x_out[i] = wb[i]; lo[i] = lob[i]; hi[i] = hib[i]; }      /* (bounds the restoration moved stay moved) */
    const int stop = inf->stop;
    arena_release(mark_);
    return stop;
}

/* ---- the problem as the callers pose it: scaling, first iterate, least-squares multipliers, then the iteration -------------------------------------------- */
typedef struct { evalf2_t evalf; void *ctx; double df; double *lams; int m; } MainCtx;
static void main_ev(IpmProb *P, const double *w, const double *lam, double mu, int want, double *f, double *gf, double *g, double *J, double *H)
{
    MainCtx *C = (MainCtx *)P->ctx;
    (void)mu;
    if (want == WANT_HESS) for (int j = 0; j < C->m; j++) C->lams[j] = lam[j] / C->df;      /* Hessian of df f + lam'g = df (Hessian of f + (lam / df)'g) */
    C->evalf(C->ctx, w, want == WANT_HESS ? C->lams : lam, want, f, gf, g, J, H);
    *f *= C->df;
    if (want != WANT_VALUES) for (int i = 0; i < P->n; i++) gf[i] *= C->df;
    if (want == WANT_HESS) for (size_t i = 0; i < (size_t)P->n * P->n; i++) H[i] *= C->df;
}

/* n variables (none fixed: the caller has removed parameters), m equalities; resto: whether a failed line search may enter the restoration phase */
static int ipm_ipopt(int n, int m, evalf2_t evalf, void *ctx, double *w, const double *lo_in, const double *hi_in, double tol, int max_iter, int resto, int *iters, double *lam_out, IpmInfo *info)
{
    const size_t mark_ = arena_mark();
    double *lo = vec(n), *hi = vec(n), *zl = vec(n), *zh = vec(n), *lam = vec(m), *lamn = vec(m), *gf = vec(n), *g = vec(m), *J = vec((size_t)m * n), *H = vec((size_t)n * n), *Jt = vec((size_t)n * m), *tau = vec(2 * m + 2), *tmp = vec(n);
    /* IPOPT relaxes every finite bound by bound_relax_factor max(1, |bound|) before it starts (OrigIpoptNLP::relax_bounds [ext], default 1e-8, left there by
       MPC_code.py:262-263) and projects the final point back into the caller's bounds (honor_original_bounds = yes, the default of the 3.12 series [ext]) */
    for (int i = 0; i < n; i++) {
        lo[i] = isfinite(lo_in[i]) ? lo_in[i] - BOUND_RELAX_FACTOR * fmax(1.0, fabs(lo_in[i])) : lo_in[i];
        hi[i] = isfinite(hi_in[i]) ? hi_in[i] + BOUND_RELAX_FACTOR * fmax(1.0, fabs(hi_in[i])) : hi_in[i];
    }
    IpmInfo inf = {0, 0, 0, 0, 2, 0, 0, 1.0};
    double f;
    /* scaling of the objective at the caller's point */
    evalf(ctx, w, lam, WANT_GRAD, &f, gf, g, J, H);
    double gmax = 0.0;
    for (int i = 0; i < n; i++) gmax = fmax(gmax, fabs(gf[i]));
    const double df = gmax > SCALE_MAX_GRAD ? fmax(SCALE_MAX_GRAD / gmax, SCALE_MIN) : 1.0;
    inf.df = df;
    for (int i = 0; i < n; i++) { w[i] = push_in(w[i], lo[i], hi[i]); zl[i] = isfinite(lo[i]) ? 1.0 : 0.0; zh[i] = isfinite(hi[i]) ? 1.0 : 0.0; }
    /* least-squares equality multipliers: min |gf - zl + zh + J'y|  (null-space form: R y = -Q1'(gf - zl + zh)) */
    if (m > 0 && m < n) {
        evalf(ctx, w, lam, WANT_GRAD, &f, gf, g, J, H);
        int ok = 1;
        for (int i = 0; i < n; i++) { tmp[i] = df * gf[i] - zl[i] + zh[i]; ok = ok && isfinite(tmp[i]); for (int j = 0; j < m; j++) Jt[i * m + j] = J[j * n + i]; }
        if (ok && qr_factor(n, m, Jt, tau)) {
            qr_apply(n, m, Jt, tau, tmp, 1);
            double ymax = 0.0;
            for (int i = m - 1; i >= 0; i--) { double s = -tmp[i]; for (int k = i + 1; k < m; k++) s -= r_at(m, Jt, tau, i, k) * lamn[k]; lamn[i] = s / r_at(m, Jt, tau, i, i); }
            for (int j = 0; j < m; j++) { ymax = fmax(ymax, fabs(lamn[j])); ok = ok && isfinite(lamn[j]); }
            if (ok && ymax <= Y_INIT_MAX) memcpy(lam, lamn, sizeof(double) * m);
        }
    }
    MainCtx C = {evalf, ctx, df, vec(m), m};
    IpmProb P = {n, m, main_ev, NULL, &C};
    const int status = ipm_core(&P, w, lo, hi, zl, zh, lam, MU_INIT, tol, max_iter, 0, THETA_MAX_FACT, resto, iters, &inf);
    for (int i = 0; i < n; i++) w[i] = fmin(fmax(w[i], lo_in[i]), hi_in[i]);      /* honor_original_bounds */
    if (lam_out) for (int j = 0; j < m; j++) lam_out[j] = lam[j] / df;
    if (info) *info = inf;
    arena_release(mark_);
    return status;
}

#endif /* ORC_DENSE_H */
