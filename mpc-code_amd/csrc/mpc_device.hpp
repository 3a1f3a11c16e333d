// Device-side algorithm of libmpc_amd.so: one MPC instance per lane ("instance-per-lane" mapping).
//
// Everything here is templated on the problem dimensions so that every small-matrix loop is fully
// unrolled into fp64 FMAs on registers; the only memory traffic is the per-instance workspace of the
// OCP (structure-of-arrays, instance index fastest => every wave access is a contiguous 512-byte row)
// and the problem constants, which are wave-uniform and therefore come through the scalar cache.
//
// Reference semantics (file:line in /root/reference):
//   rpdip_lane    solver(...) on the NLP of opt_dyn        Control_Calc.py:20-260 + MPC_code.py:733-805
//   target_lane   solver_ss(...) on the NLP of opt_ss      Target_Calc.py:20-161 + MPC_code.py:693-718
//   kalman_lane   kalman()                                 Estimator.py:263-311
// The numerical method (Mehrotra predictor-corrector, Riccati recursion in closed-loop form, constants
// below) is specified in DESIGN.md section 4; oracle/ holds independent restatements used by the tests only.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace mpc {

// ---- algorithm constants -----------------------------------------------------
constexpr double kMu0 = 1.0;          // initial complementarity product
constexpr double kSMin = 1.0;         // minimum initial slack
constexpr double kTau = 0.995;        // fraction to the boundary
constexpr double kTolStat = 1e-9;     // |grad_u L|_inf relative to its initial value (>= 1)
constexpr double kTolStatAcc = 1e-6;  // accepted after kStallMax stalled iterations
constexpr int kStallMax = 2;
constexpr double kTolFeas = 1e-9;     // bound residual
constexpr double kTolC = 1e-9;        // complementarity: min(s,l) <= kTolC ...
constexpr double kTolMu = 1e-12;      // ... or s*l <= kTolMu
constexpr double kMuFloor = 1e-13;    // centring target never below this
constexpr double kSFloor = 1e-11;     // ... nor below l*kSFloor
constexpr double kBoundRelax = 1e-8;  // relaxation of the stage-0 output rows
constexpr double kInfeasZ = 1e10;     // dual blow-up => infeasible

constexpr int kMaxN = 8, kMaxM = 4, kMaxY = 8, kMaxD = 8, kMaxV = kMaxN + kMaxM, kMaxC = kMaxN + kMaxM + kMaxY,
              kMaxE = kMaxN + kMaxD;

enum : int { kSolved = 0, kMaxIter = 1, kInfeasible = 2 };

// Problem constants as the kernels read them (one copy in HBM, wave-uniform loads).
struct DevProblem {
    int nx, nu, ny, nd, nxp, N, du_form, duss_form, y_bounded, estimator, max_iter, has_dsat;
    // stage form (z = x, or [x; u_prev] when du_form)
    double A[kMaxN][kMaxN], B[kMaxN][kMaxM], Q[kMaxN][kMaxN], M[kMaxN][kMaxM], R[kMaxM][kMaxM], Pf[kMaxN][kMaxN];
    double ulo[kMaxM], uhi[kMaxM], zlo_m[kMaxN], zhi_m[kMaxN], zlo_e[kMaxN], zhi_e[kMaxN];
    // model / plant in the reference's terms
    double Am[kMaxN][kMaxN], Bm[kMaxN][kMaxM], Cm[kMaxY][kMaxN], Bd[kMaxN][kMaxD], Cd[kMaxY][kMaxD], fxc[kMaxN], fyc[kMaxY];
    double Ap[kMaxN][kMaxN], Bp[kMaxN][kMaxM], Cp[kMaxY][kMaxN];
    double ymin[kMaxY], ymax[kMaxY], dmin[kMaxD], dmax[kMaxD];
    int ymap_idx[kMaxY]; double ymap_scale[kMaxY];
    // target problem in null-space coordinates
    double Ep[kMaxV][kMaxN], Zn[kMaxV][kMaxM], CZx[kMaxY][kMaxM], Hr[kMaxM][kMaxM], W[kMaxC][kMaxM], tlo[kMaxC], thi[kMaxC];
    double Qss[kMaxY][kMaxY], Rss[kMaxM][kMaxM];
    // estimator
    double Aa[kMaxE][kMaxE], Ca[kMaxY][kMaxE], Qkf[kMaxE][kMaxE], Rkf[kMaxY][kMaxY], Kfix[kMaxE][kMaxY];
};

#define MPC_UNROLL _Pragma("unroll")

__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ bool fin(double a) { return fabs(a) < 1.0e300; }
__device__ __forceinline__ double comp_measure(double s, double l)
{
    return dmin(dmin(s, l) * (1.0 / kTolC), s * l * (1.0 / kTolMu));
}

// ---- OCP workspace: rows of N*Bs doubles, addressed (row, block k, instance b) ----------------------
template <int NS, int NU>
struct WsLayout {
    static constexpr int NV = NS + NU;
    static constexpr int U = 0, Z = NU, SLO = NV, SHI = 2 * NV, LLO = 3 * NV, LHI = 4 * NV, PLO = 5 * NV, PHI = 6 * NV,
                         DU = 7 * NV, DZ = 7 * NV + NU, K = 8 * NV, LI = K + NU * NS, KFF = LI + NU * (NU + 1) / 2,
                         ROWS = KFF + NU;
};

struct Ws {
    double *base; size_t Bs; int N; int b;
    __device__ __forceinline__ double &at(int row, int k) const { return base[((size_t)row * N + k) * Bs + b]; }
};

// symmetric positive definite inverse, n x n, in place (full storage), by Cholesky; returns false if not PD
template <int n>
__device__ __forceinline__ bool spd_inverse(double (&a)[n][n])
{
    double c[n][n];
    bool ok = true;
    MPC_UNROLL for (int i = 0; i < n; i++) {
        MPC_UNROLL for (int j = 0; j <= i; j++) {
            double v = a[i][j];
            MPC_UNROLL for (int k = 0; k < j; k++) v -= c[i][k] * c[j][k];
            if (i == j) { ok = ok && (v > 0.0); c[i][i] = sqrt(v); }
            else c[i][j] = v / c[j][j];
        }
    }
    double ci[n];
    MPC_UNROLL for (int i = 0; i < n; i++) ci[i] = 1.0 / c[i][i];
    MPC_UNROLL for (int col = 0; col < n; col++) {
        double y[n];
        MPC_UNROLL for (int i = 0; i < n; i++) {
            double v = (i == col) ? 1.0 : 0.0;
            MPC_UNROLL for (int k = 0; k < i; k++) v -= c[i][k] * y[k];
            y[i] = v * ci[i];
        }
        MPC_UNROLL for (int i = n - 1; i >= 0; i--) {
            double v = y[i];
            MPC_UNROLL for (int k = i + 1; k < n; k++) v -= c[k][i] * a[k][col];
            a[i][col] = v * ci[i];
        }
    }
    return ok;
}

// Per-instance data of one OCP (registers)
template <int NS, int NU>
struct OcpInst {
    double z0[NS], zr[NS], ur[NU], c[NS], us[NU], zlo_m[NS], zhi_m[NS];
    bool ok0;
};

// xhat, xs [NX]; us, u_prev [NU]; dhat [ND]  ->  stage-form instance
template <int NX, int NU, int NY, int ND, bool DU>
__device__ __forceinline__ void build_inst(const DevProblem &P, const double (&xhat)[NX], const double (&xs)[NX],
                                           const double (&us)[NU], const double *dhat, const double (&u_prev)[NU],
                                           OcpInst<NX + (DU ? NU : 0), NU> &q)
{
    constexpr int NS = NX + (DU ? NU : 0);
    MPC_UNROLL for (int i = 0; i < NX; i++) {
        double c = P.fxc[i];
        MPC_UNROLL for (int j = 0; j < ND; j++) c += P.Bd[i][j] * dhat[j];
        q.c[i] = c; q.z0[i] = xhat[i]; q.zr[i] = xs[i];
    }
    MPC_UNROLL for (int i = 0; i < NU; i++) { q.us[i] = us[i]; q.ur[i] = DU ? 0.0 : us[i]; }
    if (DU) {
        MPC_UNROLL for (int i = 0; i < NU; i++) { q.z0[NX + i] = u_prev[i]; q.zr[NX + i] = 0.0; q.c[NX + i] = 0.0; }
    }
    MPC_UNROLL for (int i = 0; i < NS; i++) { q.zlo_m[i] = P.zlo_m[i]; q.zhi_m[i] = P.zhi_m[i]; }
    q.ok0 = true;
    if (P.y_bounded) {
        MPC_UNROLL for (int i = 0; i < NY; i++) {
            double e = P.fyc[i];
            MPC_UNROLL for (int j = 0; j < ND; j++) e += P.Cd[i][j] * dhat[j];
            double y0 = e;
            MPC_UNROLL for (int j = 0; j < NX; j++) y0 += P.Cm[i][j] * xhat[j];
            // stage-0 row (Control_Calc.py:128-151): constraint on a given quantity = feasibility test
            const double rl = kBoundRelax * dmax(1.0, fabs(P.ymin[i])), rh = kBoundRelax * dmax(1.0, fabs(P.ymax[i]));
            if (!(y0 >= P.ymin[i] - rl) || !(y0 <= P.ymax[i] + rh)) q.ok0 = false;
            const double sc = P.ymap_scale[i];
            const double a = (P.ymin[i] - e) / sc, b = (P.ymax[i] - e) / sc;
            const double lo = sc > 0 ? a : b, hi = sc > 0 ? b : a;
            const int idx = P.ymap_idx[i];
            MPC_UNROLL for (int j = 0; j < NX; j++)
                if (j == idx) { q.zlo_m[j] = dmax(q.zlo_m[j], lo); q.zhi_m[j] = dmin(q.zhi_m[j], hi); }
        }
    }
}

// --------------------------------------------------------------------------------------------------------
// RPDIP: Mehrotra predictor-corrector, Riccati KKT solves, one instance per lane.
// Four sweeps over the horizon per iteration:
//   B1 (backward)  apply the previous step, residuals + convergence data, factorisation, predictor rhs
//   F1 (forward)   predictor direction, step length, second-order products
//   B2 (backward)  corrector rhs
//   F2 (forward)   corrector direction, step length
// Returns the status; u0/z1 receive the first input / next state of the final iterate.
// --------------------------------------------------------------------------------------------------------
template <int NS, int NU, bool HASM>
__device__ int rpdip_lane(const DevProblem &P, const OcpInst<NS, NU> &q, const Ws &ws, int max_iter,
                          double (&u0)[NU], double (&z1)[NS], double (&res)[3], int &iters)
{
    using L = WsLayout<NS, NU>;
    constexpr int NV = NS + NU;
    const int N = ws.N;
    res[0] = res[1] = res[2] = 0.0;
    iters = 0;
    if (!q.ok0) return kInfeasible;

    // bounds: u box, z box (mid stages / last stage); fl/fh say which are finite
    double blo_u[NU], bhi_u[NU];
    bool fl_u[NU], fh_u[NU], fl_zm[NS], fh_zm[NS], fl_ze[NS], fh_ze[NS];
    double ncon = 0.0;
    MPC_UNROLL for (int i = 0; i < NU; i++) {
        blo_u[i] = P.ulo[i]; bhi_u[i] = P.uhi[i]; fl_u[i] = fin(blo_u[i]); fh_u[i] = fin(bhi_u[i]);
        ncon += (double)N * ((fl_u[i] ? 1 : 0) + (fh_u[i] ? 1 : 0));
    }
    MPC_UNROLL for (int i = 0; i < NS; i++) {
        fl_zm[i] = fin(q.zlo_m[i]); fh_zm[i] = fin(q.zhi_m[i]); fl_ze[i] = fin(P.zlo_e[i]); fh_ze[i] = fin(P.zhi_e[i]);
        ncon += (double)(N - 1) * ((fl_zm[i] ? 1 : 0) + (fh_zm[i] ? 1 : 0)) + (fl_ze[i] ? 1 : 0) + (fh_ze[i] ? 1 : 0);
    }
    const double inv_ncon = 1.0 / dmax(ncon, 1.0);

#define MPC_BOUNDS(k, i, lo, hi, fl, fh)                                                           \
    double lo, hi; bool fl, fh;                                                                    \
    if (i < NU) { lo = blo_u[i < NU ? i : 0]; hi = bhi_u[i < NU ? i : 0]; fl = fl_u[i < NU ? i : 0]; fh = fh_u[i < NU ? i : 0]; } \
    else if (k < N - 1) { lo = q.zlo_m[i >= NU ? i - NU : 0]; hi = q.zhi_m[i >= NU ? i - NU : 0]; fl = fl_zm[i >= NU ? i - NU : 0]; fh = fh_zm[i >= NU ? i - NU : 0]; } \
    else { lo = P.zlo_e[i >= NU ? i - NU : 0]; hi = P.zhi_e[i >= NU ? i - NU : 0]; fl = fl_ze[i >= NU ? i - NU : 0]; fh = fh_ze[i >= NU ? i - NU : 0]; } \
    if (!fl) lo = 0.0;                                                                             \
    if (!fh) hi = 0.0;

    // ---- initial point: u = us pushed inside its box, z simulated, slacks >= kSMin -------------------
    {
        double uinit[NU], z[NS];
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            const double lo = P.ulo[i], hi = P.uhi[i];
            double push, v = q.us[i];
            if (fl_u[i] && fh_u[i]) push = 0.1 * (hi - lo);
            else push = 0.1 * dmax(1.0, fabs(fl_u[i] ? lo : (fh_u[i] ? hi : 0.0)));
            if (fl_u[i]) v = dmax(v, lo + push);
            if (fh_u[i]) v = dmin(v, hi - push);
            uinit[i] = v;
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) z[i] = q.z0[i];
        for (int k = 0; k < N; k++) {
            double zn[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                double a = q.c[i];
                MPC_UNROLL for (int j = 0; j < NS; j++) a += P.A[i][j] * z[j];
                MPC_UNROLL for (int j = 0; j < NU; j++) a += P.B[i][j] * uinit[j];
                zn[i] = a;
            }
            MPC_UNROLL for (int i = 0; i < NS; i++) z[i] = zn[i];
            MPC_UNROLL for (int i = 0; i < NV; i++) {
                MPC_BOUNDS(k, i, lo, hi, fl, fh)
                const double v = i < NU ? uinit[i < NU ? i : 0] : z[i >= NU ? i - NU : 0];
                const double sl = fl ? dmax(v - lo, kSMin) : 1.0, sh = fh ? dmax(hi - v, kSMin) : 1.0;
                ws.at(L::SLO + i, k) = sl; ws.at(L::SHI + i, k) = sh;
                ws.at(L::LLO + i, k) = fl ? kMu0 / sl : 0.0; ws.at(L::LHI + i, k) = fh ? kMu0 / sh : 0.0;
                ws.at(L::PLO + i, k) = 0.0; ws.at(L::PHI + i, k) = 0.0;
                if (i < NU) { ws.at(L::U + i, k) = v; ws.at(L::DU + i, k) = 0.0; }
                else { ws.at(L::Z + (i - NU), k) = v; ws.at(L::DZ + (i - NU), k) = 0.0; }
            }
        }
    }

    double alpha = 0.0, sm = 0.0, gscale = 1.0;
    int stall = 0, status = kMaxIter;
    for (int it = 0;; it++) {
        // ======================= sweep B1 (backward) =================================================
        double mu_sum = 0.0, res_p = 0.0, res_s = 0.0, cres = 0.0, lmax = 0.0;
        double pi[NS], Pm[NS][NS], pcar[NS], unext_dev[NU];
        bool pd_ok = true;
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            pi[i] = 0.0; pcar[i] = 0.0;
            MPC_UNROLL for (int j = 0; j < NS; j++) Pm[i][j] = P.Pf[i][j];
        }
        MPC_UNROLL for (int i = 0; i < NU; i++) unext_dev[i] = 0.0;
        double ublk[NU], zblk[NS];
        for (int k = N - 1; k >= 0; k--) {
            double sig[NV], dlm[NV], haff[NV];
            MPC_UNROLL for (int i = 0; i < NU; i++) ublk[i] = ws.at(L::U + i, k);
            MPC_UNROLL for (int i = 0; i < NS; i++) zblk[i] = ws.at(L::Z + i, k);
            MPC_UNROLL for (int i = 0; i < NV; i++) {
                MPC_BOUNDS(k, i, lo, hi, fl, fh)
                double v = i < NU ? ublk[i < NU ? i : 0] : zblk[i >= NU ? i - NU : 0];
                const double dv = i < NU ? ws.at(L::DU + (i < NU ? i : 0), k) : ws.at(L::DZ + (i >= NU ? i - NU : 0), k);
                double sl = ws.at(L::SLO + i, k), sh = ws.at(L::SHI + i, k), ll = ws.at(L::LLO + i, k), lh = ws.at(L::LHI + i, k);
                if (alpha != 0.0) {   // apply the step of the previous iteration
                    const double plo = ws.at(L::PLO + i, k), phi = ws.at(L::PHI + i, k);
                    const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                    const double rch = fh ? sh * lh - dmax(sm, lh * kSFloor) + phi : 0.0;
                    const double rcl = fl ? sl * ll - dmax(sm, ll * kSFloor) + plo : 0.0;
                    const double dsh = fh ? -rh - dv : 0.0, dsl = fl ? rl + dv : 0.0;
                    const double dlh = fh ? (-rch - lh * dsh) / sh : 0.0, dll = fl ? (-rcl - ll * dsl) / sl : 0.0;
                    sl += alpha * dsl; sh += alpha * dsh; ll += alpha * dll; lh += alpha * dlh; v += alpha * dv;
                    ws.at(L::SLO + i, k) = sl; ws.at(L::SHI + i, k) = sh; ws.at(L::LLO + i, k) = ll; ws.at(L::LHI + i, k) = lh;
                    if (i < NU) { ublk[i < NU ? i : 0] = v; ws.at(L::U + (i < NU ? i : 0), k) = v; }
                    else { zblk[i >= NU ? i - NU : 0] = v; ws.at(L::Z + (i >= NU ? i - NU : 0), k) = v; }
                }
                const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                const double isl = 1.0 / sl, ish = 1.0 / sh;
                mu_sum += sl * ll + sh * lh;
                sig[i] = ll * isl + lh * ish;
                dlm[i] = lh - ll;
                haff[i] = lh * (rh * ish - 1.0) + ll * (rl * isl + 1.0);   // h for rc = s*l
                res_p = dmax(res_p, dmax(fabs(rl), fabs(rh)));
                cres = dmax(cres, dmax(comp_measure(sl, ll), comp_measure(sh, lh)));
                lmax = dmax(lmax, dmax(ll, lh));
            }
            // gradients of the current point (cost + bound multipliers)
            double dz1[NS], du[NU], gz1[NS], gu[NU];
            MPC_UNROLL for (int i = 0; i < NS; i++) dz1[i] = zblk[i] - q.zr[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = ublk[i] - q.ur[i];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                double a = dlm[NU + i];
                if (k == N - 1) { MPC_UNROLL for (int j = 0; j < NS; j++) a += P.Pf[i][j] * dz1[j]; }
                else {
                    MPC_UNROLL for (int j = 0; j < NS; j++) a += P.Q[i][j] * dz1[j];
                    if (HASM) { MPC_UNROLL for (int j = 0; j < NU; j++) a += P.M[i][j] * unext_dev[j]; }
                }
                gz1[i] = a;
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                double a = dlm[i];
                MPC_UNROLL for (int j = 0; j < NU; j++) a += P.R[i][j] * du[j];
                gu[i] = a;
            }
            if (HASM) {   // M'(z_k - zr): z_k is block k-1 (or z0)
                double zk[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    double zv = k > 0 ? ws.at(L::Z + i, k - 1) : q.z0[i];
                    if (k > 0 && alpha != 0.0) zv += alpha * ws.at(L::DZ + i, k - 1);   // not yet updated in memory
                    zk[i] = zv - q.zr[i];
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) gu[i] += P.M[j][i] * zk[j]; }
            }
            // adjoint pi_{k+1} = gz_{k+1} + A' pi_{k+2};  stationarity residual r_u,k = gu_k + B' pi_{k+1}
            {
                double pn[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = gz1[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += P.A[j][i] * pi[j]; pn[i] = a; }
                MPC_UNROLL for (int i = 0; i < NS; i++) pi[i] = pn[i];
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = gu[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += P.B[j][i] * pi[j]; res_s = dmax(res_s, fabs(a)); }
            }
            // Riccati: P_{k+1} completed with the barrier weights of z_{k+1}
            MPC_UNROLL for (int i = 0; i < NS; i++) Pm[i][i] += sig[NU + i];
            double PB[NS][NU], PA[NS][NS], Lam[NU][NU], Psi[NU][NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pm[i][l] * P.B[l][j]; PB[i][j] = a; }
                MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pm[i][l] * P.A[l][j]; PA[i][j] = a; }
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                MPC_UNROLL for (int j = 0; j < NU; j++) { double a = P.R[i][j] + (i == j ? sig[i] : 0.0); MPC_UNROLL for (int l = 0; l < NS; l++) a += P.B[l][i] * PB[l][j]; Lam[i][j] = a; }
                MPC_UNROLL for (int j = 0; j < NS; j++) { double a = HASM ? P.M[j][i] : 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += P.B[l][i] * PA[l][j]; Psi[i][j] = a; }
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (Lam[i][j] + Lam[j][i]); Lam[i][j] = a; Lam[j][i] = a; } }
            pd_ok = spd_inverse<NU>(Lam) && pd_ok;     // Lam now holds Li
            double Kk[NU][NS], Acl[NS][NS];
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NU; l++) a += Lam[i][l] * Psi[l][j]; Kk[i][j] = -a; ws.at(L::K + i * NS + j, k) = -a; } }
            {
                int c = 0;
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j <= i; j++) { ws.at(L::LI + c, k) = Lam[i][j]; c++; } }
            }
            // predictor rhs
            double pv[NS], qu[NU];
            MPC_UNROLL for (int i = 0; i < NS; i++) pv[i] = gz1[i] + haff[NU + i] + pcar[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) qu[i] = gu[i] + haff[i];
            {
                double psi[NU];
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = qu[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += P.B[j][i] * pv[j]; psi[i] = a; }
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) a += Lam[i][j] * psi[j]; ws.at(L::KFF + i, k) = -a; }
            }
            if (k > 0) {
                MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = P.A[i][j]; MPC_UNROLL for (int l = 0; l < NU; l++) a += P.B[i][l] * Kk[l][j]; Acl[i][j] = a; } }
                // closed-loop (Joseph) form: P_k = Q + Acl' P Acl + K' Rt K + M K + K' M'  (no cancellation)
                double T[NS][NS], RK[NU][NS], Pn[NS][NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pm[i][l] * Acl[l][j]; T[i][j] = a; } }
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = sig[i] * Kk[i][j]; MPC_UNROLL for (int l = 0; l < NU; l++) a += P.R[i][l] * Kk[l][j]; RK[i][j] = a; } }
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    MPC_UNROLL for (int j = 0; j < NS; j++) {
                        double a = P.Q[i][j];
                        MPC_UNROLL for (int l = 0; l < NS; l++) a += Acl[l][i] * T[l][j];
                        MPC_UNROLL for (int l = 0; l < NU; l++) a += Kk[l][i] * RK[l][j];
                        if (HASM) { MPC_UNROLL for (int l = 0; l < NU; l++) a += P.M[i][l] * Kk[l][j] + Kk[l][i] * P.M[j][l]; }
                        Pn[i][j] = a;
                    }
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Pm[i][j] = 0.5 * (Pn[i][j] + Pn[j][i]); }
                double pn[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    double a = 0.0;
                    MPC_UNROLL for (int j = 0; j < NS; j++) a += Acl[j][i] * pv[j];
                    MPC_UNROLL for (int j = 0; j < NU; j++) a += Kk[j][i] * qu[j];
                    pn[i] = a;
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) pcar[i] = pn[i];
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) unext_dev[i] = du[i];
        }
        // ---- convergence / failure tests at the current iterate ------------------------------------
        const double mu = mu_sum * inv_ncon;
        if (it == 0) gscale = dmax(1.0, res_s);
        res[0] = res_s; res[1] = res_p; res[2] = mu;
        iters = it;
        MPC_UNROLL for (int i = 0; i < NU; i++) u0[i] = ublk[i];
        MPC_UNROLL for (int i = 0; i < NS; i++) z1[i] = zblk[i];
        const bool ok_cp = (cres <= 1.0) && (res_p <= kTolFeas);
        stall = ok_cp ? stall + 1 : 0;
        if (ok_cp && (res_s <= kTolStat * gscale || (stall > kStallMax && res_s <= kTolStatAcc * gscale))) { status = kSolved; break; }
        if (lmax > kInfeasZ * gscale || !(fabs(mu) < 1.0e300) || !pd_ok) { status = kInfeasible; break; }
        if (it == max_iter) { status = kMaxIter; break; }

        // ======================= sweep F1 (forward): predictor ======================================
        double a_aff = 1.0, s1 = 0.0, s2 = 0.0;
        {
            double dz[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = 0.0;
            for (int k = 0; k < N; k++) {
                double ddu[NU], dzn[NS];
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = ws.at(L::KFF + i, k); MPC_UNROLL for (int j = 0; j < NS; j++) a += ws.at(L::K + i * NS + j, k) * dz[j]; ddu[i] = a; }
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) a += P.A[i][j] * dz[j]; MPC_UNROLL for (int j = 0; j < NU; j++) a += P.B[i][j] * ddu[j]; dzn[i] = a; }
                MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = dzn[i];
                MPC_UNROLL for (int i = 0; i < NV; i++) {
                    MPC_BOUNDS(k, i, lo, hi, fl, fh)
                    const double v = i < NU ? ws.at(L::U + (i < NU ? i : 0), k) : ws.at(L::Z + (i >= NU ? i - NU : 0), k);
                    const double dv = i < NU ? ddu[i < NU ? i : 0] : dz[i >= NU ? i - NU : 0];
                    const double sl = ws.at(L::SLO + i, k), sh = ws.at(L::SHI + i, k), ll = ws.at(L::LLO + i, k), lh = ws.at(L::LHI + i, k);
                    const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                    const double dsh = fh ? -rh - dv : 0.0, dsl = fl ? rl + dv : 0.0;
                    const double dlh = fh ? (-sh * lh - lh * dsh) / sh : 0.0, dll = fl ? (-sl * ll - ll * dsl) / sl : 0.0;
                    if (dsl < 0.0) a_aff = dmin(a_aff, -sl / dsl);
                    if (dsh < 0.0) a_aff = dmin(a_aff, -sh / dsh);
                    if (dll < 0.0) a_aff = dmin(a_aff, -ll / dll);
                    if (dlh < 0.0) a_aff = dmin(a_aff, -lh / dlh);
                    s1 += sl * dll + ll * dsl + sh * dlh + lh * dsh;
                    s2 += dsl * dll + dsh * dlh;
                    ws.at(L::PLO + i, k) = dsl * dll; ws.at(L::PHI + i, k) = dsh * dlh;
                }
            }
        }
        {
            const double mu_aff = (mu_sum + a_aff * s1 + a_aff * a_aff * s2) * inv_ncon;
            const double rat = mu > 0.0 ? mu_aff / mu : 0.0;
            sm = dmax(rat * rat * rat * mu, kMuFloor);
        }
        // ======================= sweep B2 (backward): corrector rhs ================================
        {
            double pc[NS], und[NU];
            MPC_UNROLL for (int i = 0; i < NS; i++) pc[i] = 0.0;
            MPC_UNROLL for (int i = 0; i < NU; i++) und[i] = 0.0;
            for (int k = N - 1; k >= 0; k--) {
                double hcc[NV], dlm[NV], ub[NU], zb[NS];
                MPC_UNROLL for (int i = 0; i < NU; i++) ub[i] = ws.at(L::U + i, k);
                MPC_UNROLL for (int i = 0; i < NS; i++) zb[i] = ws.at(L::Z + i, k);
                MPC_UNROLL for (int i = 0; i < NV; i++) {
                    MPC_BOUNDS(k, i, lo, hi, fl, fh)
                    const double v = i < NU ? ub[i < NU ? i : 0] : zb[i >= NU ? i - NU : 0];
                    const double sl = ws.at(L::SLO + i, k), sh = ws.at(L::SHI + i, k), ll = ws.at(L::LLO + i, k), lh = ws.at(L::LHI + i, k);
                    const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                    const double rch = fh ? sh * lh - dmax(sm, lh * kSFloor) + ws.at(L::PHI + i, k) : 0.0;
                    const double rcl = fl ? sl * ll - dmax(sm, ll * kSFloor) + ws.at(L::PLO + i, k) : 0.0;
                    hcc[i] = (-rch + lh * rh) / sh + (rcl + ll * rl) / sl;
                    dlm[i] = lh - ll;
                }
                double dz1[NS], du[NU], pv[NS], qu[NU];
                MPC_UNROLL for (int i = 0; i < NS; i++) dz1[i] = zb[i] - q.zr[i];
                MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = ub[i] - q.ur[i];
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    double a = dlm[NU + i] + hcc[NU + i] + pc[i];
                    if (k == N - 1) { MPC_UNROLL for (int j = 0; j < NS; j++) a += P.Pf[i][j] * dz1[j]; }
                    else {
                        MPC_UNROLL for (int j = 0; j < NS; j++) a += P.Q[i][j] * dz1[j];
                        if (HASM) { MPC_UNROLL for (int j = 0; j < NU; j++) a += P.M[i][j] * und[j]; }
                    }
                    pv[i] = a;
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) {
                    double a = dlm[i] + hcc[i];
                    MPC_UNROLL for (int j = 0; j < NU; j++) a += P.R[i][j] * du[j];
                    qu[i] = a;
                }
                if (HASM) {
                    double zk[NS];
                    MPC_UNROLL for (int i = 0; i < NS; i++) zk[i] = (k > 0 ? ws.at(L::Z + i, k - 1) : q.z0[i]) - q.zr[i];
                    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) qu[i] += P.M[j][i] * zk[j]; }
                }
                double Li[NU][NU], Kk[NU][NS];
                {
                    int c = 0;
                    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j <= i; j++) { const double a = ws.at(L::LI + c, k); Li[i][j] = a; Li[j][i] = a; c++; } }
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Kk[i][j] = ws.at(L::K + i * NS + j, k); }
                {
                    double psi[NU];
                    MPC_UNROLL for (int i = 0; i < NU; i++) { double a = qu[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += P.B[j][i] * pv[j]; psi[i] = a; }
                    MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) a += Li[i][j] * psi[j]; ws.at(L::KFF + i, k) = -a; }
                }
                if (k > 0) {
                    double pn[NS];
                    MPC_UNROLL for (int i = 0; i < NS; i++) {
                        double a = 0.0;
                        MPC_UNROLL for (int j = 0; j < NS; j++) {
                            double acl = P.A[j][i];
                            MPC_UNROLL for (int l = 0; l < NU; l++) acl += P.B[j][l] * Kk[l][i];
                            a += acl * pv[j];
                        }
                        MPC_UNROLL for (int j = 0; j < NU; j++) a += Kk[j][i] * qu[j];
                        pn[i] = a;
                    }
                    MPC_UNROLL for (int i = 0; i < NS; i++) pc[i] = pn[i];
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) und[i] = du[i];
            }
        }
        // ======================= sweep F2 (forward): corrector direction ============================
        double a_max = 1.0e300;
        {
            double dz[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = 0.0;
            for (int k = 0; k < N; k++) {
                double ddu[NU], dzn[NS];
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = ws.at(L::KFF + i, k); MPC_UNROLL for (int j = 0; j < NS; j++) a += ws.at(L::K + i * NS + j, k) * dz[j]; ddu[i] = a; }
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) a += P.A[i][j] * dz[j]; MPC_UNROLL for (int j = 0; j < NU; j++) a += P.B[i][j] * ddu[j]; dzn[i] = a; }
                MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = dzn[i];
                MPC_UNROLL for (int i = 0; i < NU; i++) ws.at(L::DU + i, k) = ddu[i];
                MPC_UNROLL for (int i = 0; i < NS; i++) ws.at(L::DZ + i, k) = dz[i];
                MPC_UNROLL for (int i = 0; i < NV; i++) {
                    MPC_BOUNDS(k, i, lo, hi, fl, fh)
                    const double v = i < NU ? ws.at(L::U + (i < NU ? i : 0), k) : ws.at(L::Z + (i >= NU ? i - NU : 0), k);
                    const double dv = i < NU ? ddu[i < NU ? i : 0] : dz[i >= NU ? i - NU : 0];
                    const double sl = ws.at(L::SLO + i, k), sh = ws.at(L::SHI + i, k), ll = ws.at(L::LLO + i, k), lh = ws.at(L::LHI + i, k);
                    const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                    const double rch = fh ? sh * lh - dmax(sm, lh * kSFloor) + ws.at(L::PHI + i, k) : 0.0;
                    const double rcl = fl ? sl * ll - dmax(sm, ll * kSFloor) + ws.at(L::PLO + i, k) : 0.0;
                    const double dsh = fh ? -rh - dv : 0.0, dsl = fl ? rl + dv : 0.0;
                    const double dlh = fh ? (-rch - lh * dsh) / sh : 0.0, dll = fl ? (-rcl - ll * dsl) / sl : 0.0;
                    if (dsl < 0.0) a_max = dmin(a_max, -sl / dsl);
                    if (dsh < 0.0) a_max = dmin(a_max, -sh / dsh);
                    if (dll < 0.0) a_max = dmin(a_max, -ll / dll);
                    if (dlh < 0.0) a_max = dmin(a_max, -lh / dlh);
                }
            }
        }
        alpha = dmin(1.0, kTau * dmin(a_max, 1.0));
    }
#undef MPC_BOUNDS
    return status;
}

// --------------------------------------------------------------------------------------------------------
// target problem, reduced to the null space of [A-I, B]; registers only
// --------------------------------------------------------------------------------------------------------
template <int NX, int NU, int NY, int ND>
__device__ int target_lane(const DevProblem &P, const double *usp, const double *ysp, const double *dhat,
                           const double (&us_prev)[NU], double (&xs)[NX], double (&us)[NU], double (&ys)[NY], int &iters)
{
    constexpr int NV = NX + NU, NC = NV + NY, NR = NU;
    double cx[NX], e[NY], vp[NV], yp[NY], gr[NR], w0[NC], y[NR];
    double s_lo[NC], s_hi[NC], l_lo[NC], l_hi[NC], lo[NC], hi[NC];
    bool fl[NC], fh[NC];
    MPC_UNROLL for (int i = 0; i < NX; i++) { double a = P.fxc[i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += P.Bd[i][j] * dhat[j]; cx[i] = a; }
    MPC_UNROLL for (int i = 0; i < NY; i++) { double a = P.fyc[i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += P.Cd[i][j] * dhat[j]; e[i] = a; }
    MPC_UNROLL for (int r = 0; r < NV; r++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NX; j++) a -= P.Ep[r][j] * cx[j]; vp[r] = a; }
    MPC_UNROLL for (int i = 0; i < NY; i++) { double a = e[i]; MPC_UNROLL for (int j = 0; j < NX; j++) a += P.Cm[i][j] * vp[j]; yp[i] = a; }
    MPC_UNROLL for (int c = 0; c < NR; c++) {
        double a = 0.0;
        MPC_UNROLL for (int i = 0; i < NY; i++) { double qi = 0.0; MPC_UNROLL for (int j = 0; j < NY; j++) qi += P.Qss[i][j] * (yp[j] - ysp[j]); a += qi * P.CZx[i][c]; }
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            double ri = 0.0;
            MPC_UNROLL for (int j = 0; j < NU; j++) ri += P.Rss[i][j] * (vp[NX + j] - (P.duss_form ? us_prev[j] : usp[j]));
            a += ri * P.Zn[NX + i][c];
        }
        gr[c] = a;
    }
    MPC_UNROLL for (int r = 0; r < NV; r++) w0[r] = vp[r];
    MPC_UNROLL for (int r = 0; r < NY; r++) w0[NV + r] = yp[r];
    double ncon = 0.0;
    MPC_UNROLL for (int r = 0; r < NC; r++) {
        fl[r] = fin(P.tlo[r]); fh[r] = fin(P.thi[r]); lo[r] = fl[r] ? P.tlo[r] : 0.0; hi[r] = fh[r] ? P.thi[r] : 0.0;
        ncon += (fl[r] ? 1 : 0) + (fh[r] ? 1 : 0);
    }
    const double inv_ncon = 1.0 / dmax(ncon, 1.0);
    {
        double Hi[NR][NR];
        MPC_UNROLL for (int i = 0; i < NR; i++) { MPC_UNROLL for (int j = 0; j < NR; j++) Hi[i][j] = P.Hr[i][j]; }
        spd_inverse<NR>(Hi);
        MPC_UNROLL for (int i = 0; i < NR; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NR; j++) a += Hi[i][j] * gr[j]; y[i] = -a; }
    }
    MPC_UNROLL for (int r = 0; r < NC; r++) {
        double v = w0[r]; MPC_UNROLL for (int c = 0; c < NR; c++) v += P.W[r][c] * y[c];
        s_lo[r] = fl[r] ? dmax(v - lo[r], kSMin) : 1.0; s_hi[r] = fh[r] ? dmax(hi[r] - v, kSMin) : 1.0;
        l_lo[r] = fl[r] ? kMu0 / s_lo[r] : 0.0; l_hi[r] = fh[r] ? kMu0 / s_hi[r] : 0.0;
    }
    double gscale = 1.0; int stall = 0, status = kMaxIter;
    MPC_UNROLL for (int c = 0; c < NR; c++) gscale = dmax(gscale, fabs(gr[c]));
    for (int it = 0;; it++) {
        double mu = 0.0, res_p = 0.0, res_s = 0.0, cres = 0.0, lmax = 0.0, grad[NR], sig[NC], r_lo[NC], r_hi[NC];
        MPC_UNROLL for (int r = 0; r < NC; r++) {
            double v = w0[r]; MPC_UNROLL for (int c = 0; c < NR; c++) v += P.W[r][c] * y[c];
            r_lo[r] = fl[r] ? v - s_lo[r] - lo[r] : 0.0; r_hi[r] = fh[r] ? v + s_hi[r] - hi[r] : 0.0;
            mu += s_lo[r] * l_lo[r] + s_hi[r] * l_hi[r];
            sig[r] = l_lo[r] / s_lo[r] + l_hi[r] / s_hi[r];
            res_p = dmax(res_p, dmax(fabs(r_lo[r]), fabs(r_hi[r])));
            cres = dmax(cres, dmax(comp_measure(s_lo[r], l_lo[r]), comp_measure(s_hi[r], l_hi[r])));
            lmax = dmax(lmax, dmax(l_lo[r], l_hi[r]));
        }
        mu *= inv_ncon;
        MPC_UNROLL for (int c = 0; c < NR; c++) {
            double a = gr[c]; MPC_UNROLL for (int j = 0; j < NR; j++) a += P.Hr[c][j] * y[j];
            MPC_UNROLL for (int r = 0; r < NC; r++) a += (l_hi[r] - l_lo[r]) * P.W[r][c];
            grad[c] = a; res_s = dmax(res_s, fabs(a));
        }
        iters = it;
        const bool ok_cp = (cres <= 1.0) && (res_p <= kTolFeas);
        stall = ok_cp ? stall + 1 : 0;
        if (ok_cp && (res_s <= kTolStat * gscale || (stall > kStallMax && res_s <= kTolStatAcc * gscale))) { status = kSolved; break; }
        if (lmax > kInfeasZ * gscale || !(fabs(mu) < 1.0e300)) { status = kInfeasible; break; }
        if (it == P.max_iter) { status = kMaxIter; break; }
        double Ht[NR][NR];
        MPC_UNROLL for (int i = 0; i < NR; i++) { MPC_UNROLL for (int j = 0; j < NR; j++) { double a = P.Hr[i][j]; MPC_UNROLL for (int r = 0; r < NC; r++) a += sig[r] * P.W[r][i] * P.W[r][j]; Ht[i][j] = a; } }
        if (!spd_inverse<NR>(Ht)) { status = kInfeasible; break; }
        double dy[NR], ds_lo[NC], ds_hi[NC], dl_lo[NC], dl_hi[NC];
        double sm = 0.0, alpha = 1.0;
        MPC_UNROLL for (int pass = 0; pass < 2; pass++) {
            double rc_lo[NC], rc_hi[NC], rhs[NR];
            MPC_UNROLL for (int r = 0; r < NC; r++) {
                if (pass == 0) { rc_lo[r] = fl[r] ? s_lo[r] * l_lo[r] : 0.0; rc_hi[r] = fh[r] ? s_hi[r] * l_hi[r] : 0.0; }
                else {
                    rc_lo[r] = fl[r] ? s_lo[r] * l_lo[r] - dmax(sm, l_lo[r] * kSFloor) + ds_lo[r] * dl_lo[r] : 0.0;
                    rc_hi[r] = fh[r] ? s_hi[r] * l_hi[r] - dmax(sm, l_hi[r] * kSFloor) + ds_hi[r] * dl_hi[r] : 0.0;
                }
            }
            MPC_UNROLL for (int c = 0; c < NR; c++) rhs[c] = grad[c];
            MPC_UNROLL for (int r = 0; r < NC; r++) {
                const double h = (-rc_hi[r] + l_hi[r] * r_hi[r]) / s_hi[r] + (rc_lo[r] + l_lo[r] * r_lo[r]) / s_lo[r];
                MPC_UNROLL for (int c = 0; c < NR; c++) rhs[c] += h * P.W[r][c];
            }
            MPC_UNROLL for (int i = 0; i < NR; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NR; j++) a += Ht[i][j] * rhs[j]; dy[i] = -a; }
            double amax = 1.0, s1 = 0.0;
            MPC_UNROLL for (int r = 0; r < NC; r++) {
                double dv = 0.0; MPC_UNROLL for (int c = 0; c < NR; c++) dv += P.W[r][c] * dy[c];
                ds_hi[r] = fh[r] ? -r_hi[r] - dv : 0.0; ds_lo[r] = fl[r] ? r_lo[r] + dv : 0.0;
                dl_hi[r] = fh[r] ? (-rc_hi[r] - l_hi[r] * ds_hi[r]) / s_hi[r] : 0.0;
                dl_lo[r] = fl[r] ? (-rc_lo[r] - l_lo[r] * ds_lo[r]) / s_lo[r] : 0.0;
                if (ds_lo[r] < 0) amax = dmin(amax, -s_lo[r] / ds_lo[r]);
                if (ds_hi[r] < 0) amax = dmin(amax, -s_hi[r] / ds_hi[r]);
                if (dl_lo[r] < 0) amax = dmin(amax, -l_lo[r] / dl_lo[r]);
                if (dl_hi[r] < 0) amax = dmin(amax, -l_hi[r] / dl_hi[r]);
            }
            if (pass == 0) {
                MPC_UNROLL for (int r = 0; r < NC; r++) s1 += (s_lo[r] + amax * ds_lo[r]) * (l_lo[r] + amax * dl_lo[r]) + (s_hi[r] + amax * ds_hi[r]) * (l_hi[r] + amax * dl_hi[r]);
                const double mu_aff = s1 * inv_ncon, rat = mu > 0.0 ? mu_aff / mu : 0.0;
                sm = dmax(rat * rat * rat * mu, kMuFloor);
            } else alpha = dmin(1.0, kTau * amax);
        }
        MPC_UNROLL for (int c = 0; c < NR; c++) y[c] += alpha * dy[c];
        MPC_UNROLL for (int r = 0; r < NC; r++) { s_lo[r] += alpha * ds_lo[r]; s_hi[r] += alpha * ds_hi[r]; l_lo[r] += alpha * dl_lo[r]; l_hi[r] += alpha * dl_hi[r]; }
    }
    MPC_UNROLL for (int r = 0; r < NV; r++) {
        double a = vp[r]; MPC_UNROLL for (int c = 0; c < NR; c++) a += P.Zn[r][c] * y[c];
        if (r < NX) xs[r < NX ? r : 0] = a; else us[r >= NX ? r - NX : 0] = a;
    }
    MPC_UNROLL for (int i = 0; i < NY; i++) { double a = e[i]; MPC_UNROLL for (int j = 0; j < NX; j++) a += P.Cm[i][j] * xs[j]; ys[i] = a; }
    return status;
}

// --------------------------------------------------------------------------------------------------------
// estimator: xi = [xhat; dhat], Pk row-major [NE][NE]; innov = y - yhat
// --------------------------------------------------------------------------------------------------------
template <int NE, int NY>
__device__ void kalman_lane(const DevProblem &P, double (&xi)[NE], double (&Pk)[NE][NE], const double (&innov)[NY])
{
    double PCt[NE][NY], S[NY][NY], K[NE][NY];
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += Pk[i][l] * P.Ca[j][l]; PCt[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = P.Rkf[i][j]; MPC_UNROLL for (int l = 0; l < NE; l++) a += P.Ca[i][l] * PCt[l][j]; S[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (S[i][j] + S[j][i]); S[i][j] = a; S[j][i] = a; } }
    spd_inverse<NY>(S);                                   // K = P C' S^-1   (Estimator.py:297)
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) a += PCt[i][l] * S[l][j]; K[i][j] = a; } }
    double CP[NY][NE], Pc[NE][NE], T[NE][NE];
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += P.Ca[i][l] * Pk[l][j]; CP[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = Pk[i][j]; MPC_UNROLL for (int l = 0; l < NY; l++) a -= K[i][l] * CP[l][j]; Pc[i][j] = a; } }   // :300
    MPC_UNROLL for (int i = 0; i < NE; i++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) a += K[i][l] * innov[l]; xi[i] += a; }                                              // :303-306
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += P.Aa[i][l] * Pc[l][j]; T[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = P.Qkf[i][j]; MPC_UNROLL for (int l = 0; l < NE; l++) a += T[i][l] * P.Aa[j][l]; Pk[i][j] = a; } }  // :309
}

}  // namespace mpc
