/* ORACLE (test infrastructure, never shipped): dense linear algebra and the interior point method shared by the C restatements
 * (enmpc_oracle.c, nmpc_oracle.c).  Everything static: each restatement is one translation unit.
 *
 * ipm_nullspace: primal-dual interior point method on  min f(w)  s.t.  g(w) = 0,  lo <= w <= hi  with the outer algorithm documented in
 * enmpc_oracle.py:ipm_dense (the reference solver's, IPOPT at the reference's options), Newton steps by a NULL-SPACE method: Householder
 * QR of the constraint Jacobian, Cholesky of the reduced Hessian (its failure is the inertia test). */
#ifndef ORC_DENSE_H
#define ORC_DENSE_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ST_SOLVED = 0, ST_MAXITER = 1, ST_FAILED = 2 };
#define KAPPA_PUSH 1e-2
#define MU_INIT 0.1
#define KAPPA_EPS 10.0
#define KAPPA_MU 0.2
#define THETA_MU 1.5
#define TAU_MIN 0.99
#define KAPPA_SIGMA 1e10
#define S_MAX 100.0
/* IPOPT's safe slack (IpIpoptCalculatedQuantities.cpp: CalculateSafeSlack, slack_move = eps^(3/4)) - a PROTOTYPE for the next round, off unless orc_safe_slack is set
   (the kernels do not have it): a slack below eps min(1, mu) becomes min(max(mu / z, eps min(1, mu)), max(s, 0) + slack_move max(1, |bound|)) and the
   bound of this solve moves by the difference (where the doubles at the bound can show it). */
#define SLACK_EPS 2.220446049250313e-16
#define SLACK_MOVE 1.81898940354585648e-12
static int orc_safe_slack = 0;
static inline double slack_of(double w, double *bound, double z, double mu, int lower)
{
    double s = lower ? w - *bound : *bound - w;
    const double s_min = SLACK_EPS * fmin(1.0, mu);
    if (orc_safe_slack && s < s_min) {
        s = fmin(fmax(mu / z, s_min), fmax(s, 0.0) + SLACK_MOVE * fmax(1.0, fabs(*bound)));
        *bound = lower ? w - s : w + s;
    }
    return s;
}
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
    double *lo = vec(n), *hi = vec(n);      /* (this solve's own bounds: the safe-slack prototype moves them) */
    memcpy(lo, lo_in, sizeof(double) * n); memcpy(hi, hi_in, sizeof(double) * n);
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
            sl[i] = fl ? slack_of(w[i], &lo[i], zl[i], mu, 1) : 1.0; sh[i] = fh ? slack_of(w[i], &hi[i], zh[i], mu, 0) : 1.0;
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
            if (isfinite(lo[i])) { const double s = slack_of(w[i], &lo[i], zl[i], mu, 1); zl[i] = fmin(fmax(zl[i], mu / (KAPPA_SIGMA * s)), KAPPA_SIGMA * mu / s); }
            if (isfinite(hi[i])) { const double s = slack_of(w[i], &hi[i], zh[i], mu, 0); zh[i] = fmin(fmax(zh[i], mu / (KAPPA_SIGMA * s)), KAPPA_SIGMA * mu / s); }
        }
        for (int j = 0; j < m; j++) lam[j] += apr * (lamn[j] - lam[j]);
    }
    *iters = it;
    if (lam_out) memcpy(lam_out, lam, sizeof(double) * m);
    arena_release(mark_);
    return status;
}

#endif /* ORC_DENSE_H */
