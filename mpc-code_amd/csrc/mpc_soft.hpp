// Soft output constraints of the linear OCP (reference Control_Calc.py:39-40,186-192,228-239; Default_Values.py:128; `slacks = True` in an Ex-file): ONE slack vector
//     Sl = [sl_ub (ny); sl_lb (ny)] >= 0
// shared by all stages widens every stage's output rows,  ymin - sl_lb <= C x_k + Cd d + .. <= ymax + sl_ub  (k = 0..N-1), and is penalised Sl' Ws Sl in EVERY stage's
// cost (N times).  The inputs' and states' boxes stay hard.  In stage form that is
//     min  sum_k stage cost(z_k, u_k) + terminal + N sig' Ws sig      z_{k+1} = A z_k + B u_k + c,  z_0 given
//          ulo <= u_k <= uhi,   zlo <= z_k <= zhi (k = 1..N),   ymin - sig_lb <= Cy z_k + cy <= ymax + sig_ub (k = 0..N-1),   sig >= 0
// - a stage-structured QP with a GLOBAL variable block sig: an ARROWHEAD Newton system.  It is solved here by the same Mehrotra predictor-corrector as the hard-constrained
// solver (mpc_device.hpp:rpdip_lane; same constants), the Newton system by ONE Riccati factorisation of the stage block with 1 + 2 ny right-hand
// sides (the Newton right-hand side and the columns that couple the stages to sig) and a dense 2 ny x 2 ny Schur complement for sig.
// One instance per lane; plain C++ (no wave intrinsics: every lane works alone), so the CPU test suite compiles this very file with g++ next to the dense oracle
// (tests/test_soft.py).  Workspace: `ws`, FIELDS doubles per block, element e of block k at ws[(k * FIELDS + e) * WST] (WST = 64 on the device: the lanes of a wave side by
// side; 1 on the host).
#pragma once
#include <stddef.h>
#ifndef MPC_UNROLL
#define MPC_UNROLL
#endif
#ifdef __HIPCC__
#define MPC_SOFT_FN __host__ __device__ inline
#else
#define MPC_SOFT_FN inline
#endif

namespace mpc {

template <int NS, int NU, int NY>
struct SoftProb {      // per-instance data in stage form (wave-uniform parts are the same for every lane)
    int N, max_iter;
    double A[NS][NS], B[NS][NU], Q[NS][NS], M[NS][NU], R[NU][NU], Pf[NS][NS];
    double c[NS], z0[NS], zr[NS], zrN[NS], ur[NU], us[NU];
    double ulo[NU], uhi[NU], zlo[NS], zhi[NS], zlo_e[NS], zhi_e[NS];      // hard boxes, +-inf = absent; _e: block N - 1 (z_N)
    double Cy[NY][NS], cy[NY], ymin[NY], ymax[NY];                         // soft rows (|bound| >= 1e11: that side is absent, as the reference's +-1e12)
    double Ws[2 * NY][2 * NY];
};

template <int NS, int NU, int NY>
struct SoftLayout {
    static constexpr int NSL = 2 * NY, NC = 2 * (NU + NS + NY), NCOL = 1 + NSL;
    // per block k = (u_k, z_{k+1}; the soft rows of y_k): u z | slack t | multiplier l | predictor products pp | K Li | kff, du, dz per right-hand side
    static constexpr int U = 0, Z = U + NU, T = Z + NS, L = T + NC, PP = L + NC, K = PP + NC, LI = K + NU * NS, KFF = LI + NU * NU, DU = KFF + NU * NCOL, DZ = DU + NU * NCOL,
                         FIELDS = DZ + NS * NCOL;
    // rows inside T / L / PP: lower and upper bound of the inputs, of the states z_{k+1}, of the outputs y_k
    static constexpr int RUL = 0, RUH = NU, RZL = 2 * NU, RZH = 2 * NU + NS, RYL = 2 * NU + 2 * NS, RYH = 2 * NU + 2 * NS + NY;
};

namespace soft_detail {
MPC_SOFT_FN double smax(double a, double b) { return a > b ? a : b; }
MPC_SOFT_FN double smin(double a, double b) { return a < b ? a : b; }
MPC_SOFT_FN bool sfin(double a) { return (a < 0 ? -a : a) < 1.0e300; }
// symmetric positive definite inverse by LDL' (n small); false when a pivot is not positive
template <int n> MPC_SOFT_FN bool spd_inverse(double (&a)[n][n])
{
    double l[n][n], d[n];
    bool ok = true;
    for (int j = 0; j < n; j++) {
        double dj = a[j][j];
        for (int k = 0; k < j; k++) dj -= l[j][k] * l[j][k] * d[k];
        ok = ok && dj > 0.0;
        d[j] = dj;
        for (int i = j + 1; i < n; i++) { double v = a[i][j]; for (int k = 0; k < j; k++) v -= l[i][k] * l[j][k] * d[k]; l[i][j] = v / dj; }
    }
    for (int col = 0; col < n; col++) {
        double y[n];
        for (int i = 0; i < n; i++) { double v = i == col ? 1.0 : 0.0; for (int k = 0; k < i; k++) v -= l[i][k] * y[k]; y[i] = v; }
        for (int i = n - 1; i >= 0; i--) { double v = y[i] / d[i]; for (int k = i + 1; k < n; k++) v -= l[k][i] * a[k][col]; a[i][col] = v; }
    }
    return ok;
}
}  // namespace soft_detail

// status: 0 solved, 1 iteration limit, 2 infeasible (the hard boxes cannot be met; with soft outputs alone the problem is always feasible).
// u0 / z1: first input and next state of the final iterate; sl: the optimal slack vector [sl_ub; sl_lb] (the reference's Sl, MPC_code.py:800);
// res: stationarity, bound residual, mean complementarity.
template <int NS, int NU, int NY, int WST>
MPC_SOFT_FN int soft_solve(const SoftProb<NS, NU, NY> &P, double *const ws, double (&u0)[NU], double (&z1)[NS], double (&sl)[2 * NY], double (&res)[3], int &iters)
{
    using namespace soft_detail;
    using LY = SoftLayout<NS, NU, NY>;
    constexpr int NSL = LY::NSL, NC = LY::NC, NCOL = LY::NCOL;
    constexpr double kMu0 = 1.0, kSMin = 1.0, kTau = 0.995, kTolStat = 1e-9, kTolStatAcc = 1e-6, kTolFeas = 1e-9, kTolC = 1e-9, kTolMu = 1e-14, kMuFloor = 1e-15, kSFloor = 1e-11,
                     kInfeasZ = 1e10;
    constexpr int kStallMax = 2;
    const int N = P.N;
    auto W = [&](int k, int e) -> double & { return ws[((size_t)k * LY::FIELDS + e) * WST]; };
    // which rows exist
    bool on[NC];
    for (int i = 0; i < NU; i++) { on[LY::RUL + i] = sfin(P.ulo[i]); on[LY::RUH + i] = sfin(P.uhi[i]); }
    for (int i = 0; i < NY; i++) { on[LY::RYL + i] = sfin(P.ymin[i]) && P.ymin[i] > -1e11; on[LY::RYH + i] = sfin(P.ymax[i]) && P.ymax[i] < 1e11; }
    auto zrow_on = [&](int k, int i, bool hi_) { const double b = hi_ ? (k == N - 1 ? P.zhi_e[i] : P.zhi[i]) : (k == N - 1 ? P.zlo_e[i] : P.zlo[i]); return sfin(b); };
    auto zbound = [&](int k, int i, bool hi_) { return hi_ ? (k == N - 1 ? P.zhi_e[i] : P.zhi[i]) : (k == N - 1 ? P.zlo_e[i] : P.zlo[i]); };
    double sig[NSL], tsg[NSL], lsg[NSL], ppsg[NSL];
    double Wsym[NSL][NSL];      // the cost N sig' Ws sig = 1/2 sig' (N (Ws + Ws')) sig
    for (int i = 0; i < NSL; i++) for (int j = 0; j < NSL; j++) Wsym[i][j] = (double)N * (P.Ws[i][j] + P.Ws[j][i]);
    // the value of row r of block k: t = value >= 0.  zk = z_k (the state the outputs of this block read), u, zn = z_{k+1}
    auto row_value = [&](int k, int r, const double *u, const double *zk, const double *zn) -> double {
        if (r < LY::RUH) return u[r] - P.ulo[r];
        if (r < LY::RZL) return P.uhi[r - LY::RUH] - u[r - LY::RUH];
        if (r < LY::RZH) return zn[r - LY::RZL] - zbound(k, r - LY::RZL, false);
        if (r < LY::RYL) return zbound(k, r - LY::RZH, true) - zn[r - LY::RZH];
        const int i = r < LY::RYH ? r - LY::RYL : r - LY::RYH;
        double y = P.cy[i];
        for (int j = 0; j < NS; j++) y += P.Cy[i][j] * zk[j];
        return r < LY::RYH ? y - P.ymin[i] + sig[NY + i] : P.ymax[i] + sig[i] - y;
    };
    auto row_on = [&](int k, int r) -> bool {
        if (r >= LY::RZL && r < LY::RYL) return zrow_on(k, r < LY::RZH ? r - LY::RZL : r - LY::RZH, r >= LY::RZH);
        return on[r];
    };
    // ---- initial point: inputs from us pushed inside their box, states simulated, slacks and multipliers as the hard-constrained solver's cold start --------------
    double ncon = 0.0;
    {
        double z[NS], u[NU];
        for (int i = 0; i < NS; i++) z[i] = P.z0[i];
        for (int i = 0; i < NU; i++) {
            const double lo = P.ulo[i], hi = P.uhi[i];
            const bool fl = sfin(lo), fh = sfin(hi);
            double push, v = P.us[i];
            if (fl && fh) push = 0.1 * (hi - lo); else { const double b_ = fl ? lo : (fh ? hi : 0.0); push = 0.1 * smax(1.0, b_ < 0 ? -b_ : b_); }
            if (fl) v = smax(v, lo + push);
            if (fh) v = smin(v, hi - push);
            u[i] = v;
        }
        for (int j = 0; j < NSL; j++) { sig[j] = 1.0; tsg[j] = 1.0; lsg[j] = kMu0; ppsg[j] = 0.0; ncon += 1.0; }
        for (int k = 0; k < N; k++) {
            double zn[NS];
            for (int i = 0; i < NS; i++) { double a = P.c[i]; for (int j = 0; j < NS; j++) a += P.A[i][j] * z[j]; for (int j = 0; j < NU; j++) a += P.B[i][j] * u[j]; zn[i] = a; }
            for (int i = 0; i < NU; i++) W(k, LY::U + i) = u[i];
            for (int i = 0; i < NS; i++) W(k, LY::Z + i) = zn[i];
            for (int r = 0; r < NC; r++) {
                const bool o = row_on(k, r);
                const double t = o ? smax(row_value(k, r, u, z, zn), kSMin) : 1.0;
                W(k, LY::T + r) = t; W(k, LY::L + r) = o ? kMu0 / t : 0.0; W(k, LY::PP + r) = 0.0;
                if (o) ncon += 1.0;
            }
            for (int i = 0; i < NS; i++) z[i] = zn[i];
        }
    }
    const double inv_ncon = 1.0 / ncon;
    double gscale = 1.0, alpha = 0.0;      // (alpha: the last step length)
    int stall = 0, status = 1;
    res[0] = res[1] = res[2] = 0.0;
    for (int it = 0;; it++) {
        iters = it;
        // ======== backward sweep: residuals, weights, Riccati factorisation, right-hand sides (Newton gradient + the NSL coupling columns) ========
        // h-term of a row for a complementarity residual rc: rc / t + w rp.  Predictor: rc = t l.
        double mu_sum = 0.0, res_p = 0.0, res_s = 0.0, cres = 0.0, lmax = 0.0;
        double Pm[NS][NS], pcar[NCOL][NS], und[NU];
        double gsig[NSL], Fh[NSL], S[NSL][NSL];      // gradient of the Lagrangian in sig, F'h, F'WF (the Schur complement's own part)
        bool pd_ok = true;
        for (int i = 0; i < NS; i++) { for (int j = 0; j < NS; j++) Pm[i][j] = P.Pf[i][j]; for (int c = 0; c < NCOL; c++) pcar[c][i] = 0.0; }
        for (int i = 0; i < NU; i++) und[i] = 0.0;
        for (int j = 0; j < NSL; j++) {
            double a = 0.0;
            for (int l = 0; l < NSL; l++) a += Wsym[j][l] * sig[l];
            const double rp = sig[j] - tsg[j];      // row sig_j - t = 0
            const double w = lsg[j] / tsg[j];
            gsig[j] = a - lsg[j]; Fh[j] = lsg[j] + w * rp;      // rc / t = l
            for (int l = 0; l < NSL; l++) S[j][l] = Wsym[j][l] + (j == l ? w : 0.0);
            mu_sum += tsg[j] * lsg[j];
            res_p = smax(res_p, rp < 0 ? -rp : rp);
            cres = smax(cres, smin(smin(tsg[j], lsg[j]) * (1.0 / kTolC), tsg[j] * lsg[j] * (1.0 / kTolMu)));
            lmax = smax(lmax, lsg[j]);
        }
        // output-row data of block k + 1 enter the state z_{k+1} of block k: carried from the iteration of the loop before
        double qy_next[NS], Qy_next[NS][NS], coly_next[NSL][NS];
        for (int i = 0; i < NS; i++) { qy_next[i] = 0.0; for (int j = 0; j < NS; j++) Qy_next[i][j] = 0.0; for (int c = 0; c < NSL; c++) coly_next[c][i] = 0.0; }
        for (int k = N - 1; k >= 0; k--) {
            double u[NU], zn[NS], zk[NS];
            for (int i = 0; i < NU; i++) u[i] = W(k, LY::U + i);
            for (int i = 0; i < NS; i++) { zn[i] = W(k, LY::Z + i); zk[i] = k > 0 ? W(k - 1, LY::Z + i) : P.z0[i]; }
            double wrow[NC], hrow[NC], dl[NC];
            for (int r = 0; r < NC; r++) {
                wrow[r] = 0.0; hrow[r] = 0.0; dl[r] = 0.0;
                if (!row_on(k, r)) continue;
                const double t = W(k, LY::T + r), l = W(k, LY::L + r);
                const double rp = row_value(k, r, u, zk, zn) - t;
                wrow[r] = l / t; hrow[r] = l + wrow[r] * rp; dl[r] = l;
                mu_sum += t * l;
                res_p = smax(res_p, rp < 0 ? -rp : rp);
                cres = smax(cres, smin(smin(t, l) * (1.0 / kTolC), t * l * (1.0 / kTolMu)));
                lmax = smax(lmax, l);
            }
            // soft rows of y_k: their part of the gradient / Hessian in z_k (goes to block k - 1), in sig, and the coupling columns
            double qy[NS], Qy[NS][NS], coly[NSL][NS];
            for (int i = 0; i < NS; i++) { qy[i] = 0.0; for (int j = 0; j < NS; j++) Qy[i][j] = 0.0; for (int c = 0; c < NSL; c++) coly[c][i] = 0.0; }
            for (int i = 0; i < NY; i++) {
                const int rl = LY::RYL + i, rh = LY::RYH + i;
                // lower row: e = +Cy_i, f = e_{NY+i};  upper row: e = -Cy_i, f = e_i
                const double gl = -dl[rl] + hrow[rl], gh = -dl[rh] + hrow[rh];      // (-l + h): gradient + h of the row's own direction
                for (int a = 0; a < NS; a++) {
                    qy[a] += P.Cy[i][a] * (gl - gh);
                    for (int b = 0; b < NS; b++) Qy[a][b] += P.Cy[i][a] * (wrow[rl] + wrow[rh]) * P.Cy[i][b];
                    coly[NY + i][a] += P.Cy[i][a] * wrow[rl];      // E'WF: lower row couples z_k to sig_lb,i with +w
                    coly[i][a] -= P.Cy[i][a] * wrow[rh];           //        upper row couples z_k to sig_ub,i with -w
                }
                gsig[NY + i] += -dl[rl]; gsig[i] += -dl[rh];
                Fh[NY + i] += hrow[rl]; Fh[i] += hrow[rh];
                S[NY + i][NY + i] += wrow[rl]; S[i][i] += wrow[rh];
            }
            // ---- Riccati step of block k: P_{k+1} completed with the weights of z_{k+1} (its boxes, the soft rows of y_{k+1}) ----
            for (int i = 0; i < NS; i++) { Pm[i][i] += wrow[LY::RZL + i] + wrow[LY::RZH + i]; for (int j = 0; j < NS; j++) Pm[i][j] += Qy_next[i][j]; }
            double PB[NS][NU], PA[NS][NS], Lam[NU][NU], Psi[NU][NS], Kk[NU][NS];
            for (int i = 0; i < NS; i++) {
                for (int j = 0; j < NU; j++) { double a = 0.0; for (int l = 0; l < NS; l++) a += Pm[i][l] * P.B[l][j]; PB[i][j] = a; }
                for (int j = 0; j < NS; j++) { double a = 0.0; for (int l = 0; l < NS; l++) a += Pm[i][l] * P.A[l][j]; PA[i][j] = a; }
            }
            double sigu[NU];
            for (int i = 0; i < NU; i++) sigu[i] = wrow[LY::RUL + i] + wrow[LY::RUH + i];
            for (int i = 0; i < NU; i++) {
                for (int j = 0; j <= i; j++) { double a = P.R[i][j] + (i == j ? sigu[i] : 0.0); for (int l = 0; l < NS; l++) a += P.B[l][i] * PB[l][j]; Lam[i][j] = a; Lam[j][i] = a; }
                for (int j = 0; j < NS; j++) { double a = P.M[j][i]; for (int l = 0; l < NS; l++) a += P.B[l][i] * PA[l][j]; Psi[i][j] = a; }
            }
            pd_ok = spd_inverse<NU>(Lam) && pd_ok;
            for (int i = 0; i < NU; i++) for (int j = 0; j < NS; j++) { double a = 0.0; for (int l = 0; l < NU; l++) a += Lam[i][l] * Psi[l][j]; Kk[i][j] = -a; }
            for (int i = 0; i < NU; i++) { for (int j = 0; j < NS; j++) W(k, LY::K + i * NS + j) = Kk[i][j]; for (int j = 0; j < NU; j++) W(k, LY::LI + i * NU + j) = Lam[i][j]; }
            // ---- gradients of the Lagrangian at the current point: in z_{k+1} (cost, its boxes, the rows of y_{k+1}) and u_k ----
            double gz[NS], gu[NU], hz[NS], hu[NU];
            for (int i = 0; i < NS; i++) {
                double a = -dl[LY::RZL + i] + dl[LY::RZH + i];
                if (k == N - 1) { for (int j = 0; j < NS; j++) a += P.Pf[i][j] * (zn[j] - P.zrN[j]); }
                else { for (int j = 0; j < NS; j++) a += P.Q[i][j] * (zn[j] - P.zr[j]); for (int j = 0; j < NU; j++) a += P.M[i][j] * und[j]; }
                gz[i] = a; hz[i] = hrow[LY::RZL + i] - hrow[LY::RZH + i];
            }
            for (int i = 0; i < NU; i++) {
                double a = -dl[LY::RUL + i] + dl[LY::RUH + i];
                for (int j = 0; j < NU; j++) a += P.R[i][j] * (u[j] - P.ur[j]);
                for (int j = 0; j < NS; j++) a += P.M[j][i] * (zk[j] - P.zr[j]);
                gu[i] = a; hu[i] = hrow[LY::RUL + i] - hrow[LY::RUH + i];
            }
            // right-hand sides: column 0 = gradient + h; columns 1.. = coupling columns of sig_j (E'WF)
            for (int c = 0; c < NCOL; c++) {
                double pv[NS], qu[NU], psi[NU], kff[NU];
                for (int i = 0; i < NS; i++) pv[i] = (c == 0 ? gz[i] + hz[i] + qy_next[i] : coly_next[c - 1][i]) + pcar[c][i];
                for (int i = 0; i < NU; i++) qu[i] = c == 0 ? gu[i] + hu[i] : 0.0;
                for (int i = 0; i < NU; i++) { double a = qu[i]; for (int j = 0; j < NS; j++) a += P.B[j][i] * pv[j]; psi[i] = a; }
                for (int i = 0; i < NU; i++) { double a = 0.0; for (int j = 0; j < NU; j++) a += Lam[i][j] * psi[j]; kff[i] = -a; W(k, LY::KFF + c * NU + i) = kff[i]; }
                for (int i = 0; i < NS; i++) { double a = 0.0; for (int j = 0; j < NS; j++) a += P.A[j][i] * pv[j]; for (int j = 0; j < NU; j++) a += Kk[j][i] * psi[j]; pcar[c][i] = a; }
            }
            if (k > 0) {      // P_k = Q + Acl' P Acl + K' Rt K + M K + K' M'  (closed-loop form: no cancellation of large barrier weights)
                double Acl[NS][NS], T[NS][NS], RK[NU][NS];
                for (int i = 0; i < NS; i++) for (int j = 0; j < NS; j++) { double a = P.A[i][j]; for (int l = 0; l < NU; l++) a += P.B[i][l] * Kk[l][j]; Acl[i][j] = a; }
                for (int i = 0; i < NS; i++) for (int j = 0; j < NS; j++) { double a = 0.0; for (int l = 0; l < NS; l++) a += Pm[i][l] * Acl[l][j]; T[i][j] = a; }
                for (int i = 0; i < NU; i++) for (int j = 0; j < NS; j++) { double a = sigu[i] * Kk[i][j]; for (int l = 0; l < NU; l++) a += P.R[i][l] * Kk[l][j]; RK[i][j] = a; }
                for (int i = 0; i < NS; i++) for (int j = 0; j <= i; j++) {
                    double a = P.Q[i][j];
                    for (int l = 0; l < NS; l++) a += Acl[l][i] * T[l][j];
                    for (int l = 0; l < NU; l++) a += Kk[l][i] * RK[l][j] + P.M[i][l] * Kk[l][j] + Kk[l][i] * P.M[j][l];
                    Pm[i][j] = a; Pm[j][i] = a;
                }
            }
            for (int i = 0; i < NU; i++) und[i] = u[i] - P.ur[i];
            for (int i = 0; i < NS; i++) { qy_next[i] = qy[i]; for (int j = 0; j < NS; j++) Qy_next[i][j] = Qy[i][j]; for (int c = 0; c < NSL; c++) coly_next[c][i] = coly[c][i]; }
        }
        // ======== forward sweeps: the 1 + NSL solutions dw^c (H~ dw^c = -rhs^c under the dynamics) ========
        for (int c = 0; c < NCOL; c++) {
            double dz[NS];
            for (int i = 0; i < NS; i++) dz[i] = 0.0;
            for (int k = 0; k < N; k++) {
                double du[NU], dzn[NS];
                for (int i = 0; i < NU; i++) { double a = W(k, LY::KFF + c * NU + i); for (int j = 0; j < NS; j++) a += W(k, LY::K + i * NS + j) * dz[j]; du[i] = a; W(k, LY::DU + c * NU + i) = a; }
                for (int i = 0; i < NS; i++) { double a = 0.0; for (int j = 0; j < NS; j++) a += P.A[i][j] * dz[j]; for (int j = 0; j < NU; j++) a += P.B[i][j] * du[j]; dzn[i] = a; }
                for (int i = 0; i < NS; i++) { dz[i] = dzn[i]; W(k, LY::DZ + c * NS + i) = dzn[i]; }
            }
        }
        // ======== stationarity of the current point: the Lagrangian's gradient in the reduced variables (u, sig), g_u + B' pi through the adjoint of the dynamics -
        // a sweep of its own (cheap next to the 1 + NSL solves) ========
        {
            double pia[NS];
            for (int i = 0; i < NS; i++) pia[i] = 0.0;
            double gsr[NSL];
            for (int j = 0; j < NSL; j++) { double a = 0.0; for (int l = 0; l < NSL; l++) a += Wsym[j][l] * sig[l]; gsr[j] = a - lsg[j]; }
            double undx[NU];
            for (int i = 0; i < NU; i++) undx[i] = 0.0;
            double cyl_next[NS];      // C'(-l_lo + l_hi) of the rows of y_{k+1}
            for (int i = 0; i < NS; i++) cyl_next[i] = 0.0;
            for (int k = N - 1; k >= 0; k--) {
                double u[NU], zn[NS], zk[NS];
                for (int i = 0; i < NU; i++) u[i] = W(k, LY::U + i);
                for (int i = 0; i < NS; i++) { zn[i] = W(k, LY::Z + i); zk[i] = k > 0 ? W(k - 1, LY::Z + i) : P.z0[i]; }
                auto lam = [&](int r) { return row_on(k, r) ? W(k, LY::L + r) : 0.0; };
                double gz[NS], pn[NS], cyl[NS];
                for (int i = 0; i < NS; i++) {
                    double a = -lam(LY::RZL + i) + lam(LY::RZH + i) + cyl_next[i];
                    if (k == N - 1) { for (int j = 0; j < NS; j++) a += P.Pf[i][j] * (zn[j] - P.zrN[j]); }
                    else { for (int j = 0; j < NS; j++) a += P.Q[i][j] * (zn[j] - P.zr[j]); for (int j = 0; j < NU; j++) a += P.M[i][j] * undx[j]; }
                    gz[i] = a;
                }
                for (int i = 0; i < NS; i++) { double a = gz[i]; for (int j = 0; j < NS; j++) a += P.A[j][i] * pia[j]; pn[i] = a; }
                for (int i = 0; i < NS; i++) pia[i] = pn[i];
                for (int i = 0; i < NU; i++) {
                    double a = -lam(LY::RUL + i) + lam(LY::RUH + i);
                    for (int j = 0; j < NU; j++) a += P.R[i][j] * (u[j] - P.ur[j]);
                    for (int j = 0; j < NS; j++) a += P.M[j][i] * (zk[j] - P.zr[j]) + P.B[j][i] * pia[j];
                    res_s = smax(res_s, a < 0 ? -a : a);
                }
                for (int i = 0; i < NS; i++) cyl[i] = 0.0;
                for (int i = 0; i < NY; i++) {
                    const double ll = lam(LY::RYL + i), lh = lam(LY::RYH + i);
                    for (int a = 0; a < NS; a++) cyl[a] += P.Cy[i][a] * (-ll + lh);
                    gsr[NY + i] -= ll; gsr[i] -= lh;
                }
                for (int i = 0; i < NS; i++) cyl_next[i] = cyl[i];
                for (int i = 0; i < NU; i++) undx[i] = u[i] - P.ur[i];
            }
            for (int j = 0; j < NSL; j++) res_s = smax(res_s, gsr[j] < 0 ? -gsr[j] : gsr[j]);
        }
        const double mu = mu_sum * inv_ncon;
        if (it == 0) gscale = smax(1.0, res_s);
        res[0] = res_s; res[1] = res_p; res[2] = mu;
        for (int i = 0; i < NU; i++) u0[i] = W(0, LY::U + i);
        for (int i = 0; i < NS; i++) z1[i] = W(0, LY::Z + i);
        for (int j = 0; j < NSL; j++) sl[j] = sig[j];
        const bool ok_cp = cres <= 1.0 && res_p <= kTolFeas;
        stall = ok_cp ? stall + 1 : 0;
        if (ok_cp && (res_s <= kTolStat * gscale || (stall > kStallMax && res_s <= kTolStatAcc * gscale))) { status = 0; break; }
        if (lmax > kInfeasZ * gscale || !(mu < 1.0e300) || !pd_ok) { status = 2; break; }
        if (it == P.max_iter) { status = 1; break; }
        // ======== Schur complement in sig: S = N (Ws + Ws') + F'WF - F'WE Y, with Y the solutions of the coupling columns ========
        // (F'WE y)_j for a stage direction y: sum over the soft rows of f_j w e'y: lower row i -> sig_{NY+i}: +w Cy_i dz_k; upper row i -> sig_i: -w Cy_i dz_k
        auto fwe = [&](int c, double (&out)[NSL]) {      // F'WE dw^c
            for (int j = 0; j < NSL; j++) out[j] = 0.0;
            for (int k = 1; k < N; k++) {      // (y_0 reads the given z_0: no direction)
                double dzk[NS];
                for (int i = 0; i < NS; i++) dzk[i] = W(k - 1, LY::DZ + c * NS + i);
                for (int i = 0; i < NY; i++) {
                    double cd = 0.0;
                    for (int a = 0; a < NS; a++) cd += P.Cy[i][a] * dzk[a];
                    if (row_on(k, LY::RYL + i)) out[NY + i] += W(k, LY::L + LY::RYL + i) / W(k, LY::T + LY::RYL + i) * cd;
                    if (row_on(k, LY::RYH + i)) out[i] -= W(k, LY::L + LY::RYH + i) / W(k, LY::T + LY::RYH + i) * cd;
                }
            }
        };
        double Sm[NSL][NSL];
        for (int j = 0; j < NSL; j++) for (int l = 0; l < NSL; l++) Sm[j][l] = S[j][l];
        for (int c = 1; c < NCOL; c++) { double col[NSL]; fwe(c, col); for (int j = 0; j < NSL; j++) Sm[j][c - 1] += col[j]; }      // dw = dw^0 + sum_c dw^c dsig_c (dw^c solves -col_c)
        for (int j = 0; j < NSL; j++) for (int l = 0; l < j; l++) { const double a = 0.5 * (Sm[j][l] + Sm[l][j]); Sm[j][l] = a; Sm[l][j] = a; }
        pd_ok = spd_inverse<NSL>(Sm) && pd_ok;
        // ---- one Newton solve for given complementarity residuals: rc = t l - target + pp  (predictor: target = 0, pp = 0) ----
        // The factorisation, the coupling columns and S are those of the sweep above; only the right-hand side of column 0 and of the sig block change with rc.
        // To keep ONE code path, column 0 is recomputed for the corrector by a backward + forward sweep of its own (vector work only).
        double dsig[NSL];
        auto solve_dir = [&](const double target, const bool with_pp) {
            if (with_pp || target != 0.0) {
                // column 0 again with h = (t l - target + pp) / t + w rp
                double pc[NS], undc[NU], qy_n[NS];
                for (int i = 0; i < NS; i++) { pc[i] = 0.0; qy_n[i] = 0.0; }
                for (int i = 0; i < NU; i++) undc[i] = 0.0;
                for (int j = 0; j < NSL; j++) {
                    const double rp = sig[j] - tsg[j], w = lsg[j] / tsg[j];
                    Fh[j] = (tsg[j] * lsg[j] - smax(target, lsg[j] * kSFloor) + (with_pp ? ppsg[j] : 0.0)) / tsg[j] + w * rp;
                }
                for (int k = N - 1; k >= 0; k--) {
                    double u[NU], zn[NS], zk[NS];
                    for (int i = 0; i < NU; i++) u[i] = W(k, LY::U + i);
                    for (int i = 0; i < NS; i++) { zn[i] = W(k, LY::Z + i); zk[i] = k > 0 ? W(k - 1, LY::Z + i) : P.z0[i]; }
                    double hrow[NC], dl[NC];
                    for (int r = 0; r < NC; r++) {
                        hrow[r] = 0.0; dl[r] = 0.0;
                        if (!row_on(k, r)) continue;
                        const double t = W(k, LY::T + r), l = W(k, LY::L + r), rp = row_value(k, r, u, zk, zn) - t;
                        hrow[r] = (t * l - smax(target, l * kSFloor) + (with_pp ? W(k, LY::PP + r) : 0.0)) / t + l / t * rp; dl[r] = l;
                    }
                    double qy[NS];
                    for (int i = 0; i < NS; i++) qy[i] = 0.0;
                    for (int i = 0; i < NY; i++) {
                        const int rl = LY::RYL + i, rh = LY::RYH + i;
                        for (int a = 0; a < NS; a++) qy[a] += P.Cy[i][a] * ((-dl[rl] + hrow[rl]) - (-dl[rh] + hrow[rh]));
                        Fh[NY + i] += hrow[rl]; Fh[i] += hrow[rh];
                    }
                    double pv[NS], qu[NU], psi[NU];
                    for (int i = 0; i < NS; i++) {
                        double a = -dl[LY::RZL + i] + dl[LY::RZH + i] + hrow[LY::RZL + i] - hrow[LY::RZH + i] + qy_n[i] + pc[i];
                        if (k == N - 1) { for (int j = 0; j < NS; j++) a += P.Pf[i][j] * (zn[j] - P.zrN[j]); }
                        else { for (int j = 0; j < NS; j++) a += P.Q[i][j] * (zn[j] - P.zr[j]); for (int j = 0; j < NU; j++) a += P.M[i][j] * undc[j]; }
                        pv[i] = a;
                    }
                    for (int i = 0; i < NU; i++) {
                        double a = -dl[LY::RUL + i] + dl[LY::RUH + i] + hrow[LY::RUL + i] - hrow[LY::RUH + i];
                        for (int j = 0; j < NU; j++) a += P.R[i][j] * (u[j] - P.ur[j]);
                        for (int j = 0; j < NS; j++) a += P.M[j][i] * (zk[j] - P.zr[j]);
                        qu[i] = a;
                    }
                    for (int i = 0; i < NU; i++) { double a = qu[i]; for (int j = 0; j < NS; j++) a += P.B[j][i] * pv[j]; psi[i] = a; }
                    for (int i = 0; i < NU; i++) { double a = 0.0; for (int j = 0; j < NU; j++) a += W(k, LY::LI + i * NU + j) * psi[j]; W(k, LY::KFF + i) = -a; }
                    for (int i = 0; i < NS; i++) { double a = 0.0; for (int j = 0; j < NS; j++) a += P.A[j][i] * pv[j]; for (int j = 0; j < NU; j++) a += W(k, LY::K + j * NS + i) * psi[j]; pc[i] = a; }
                    for (int i = 0; i < NU; i++) undc[i] = u[i] - P.ur[i];
                    for (int i = 0; i < NS; i++) qy_n[i] = qy[i];
                }
                double dz[NS];
                for (int i = 0; i < NS; i++) dz[i] = 0.0;
                for (int k = 0; k < N; k++) {
                    double du[NU], dzn[NS];
                    for (int i = 0; i < NU; i++) { double a = W(k, LY::KFF + i); for (int j = 0; j < NS; j++) a += W(k, LY::K + i * NS + j) * dz[j]; du[i] = a; W(k, LY::DU + i) = a; }
                    for (int i = 0; i < NS; i++) { double a = 0.0; for (int j = 0; j < NS; j++) a += P.A[i][j] * dz[j]; for (int j = 0; j < NU; j++) a += P.B[i][j] * du[j]; dzn[i] = a; }
                    for (int i = 0; i < NS; i++) { dz[i] = dzn[i]; W(k, LY::DZ + i) = dzn[i]; }
                }
            }
            // sig: Sm dsig = -(g_sig + F'h) - F'WE dw^0
            double rhs[NSL], col0[NSL];
            fwe(0, col0);
            for (int j = 0; j < NSL; j++) rhs[j] = -(gsig[j] + Fh[j]) - col0[j];
            for (int j = 0; j < NSL; j++) { double a = 0.0; for (int l = 0; l < NSL; l++) a += Sm[j][l] * rhs[l]; dsig[j] = a; }
        };
        // direction of a row's slack and multiplier for the combined stage direction dw = dw^0 + sum_c dw^c dsig_c
        auto dval = [&](int k, int r, const double *du, const double *dzk, const double *dzn) -> double {
            if (r < LY::RUH) return du[r];
            if (r < LY::RZL) return -du[r - LY::RUH];
            if (r < LY::RZH) return dzn[r - LY::RZL];
            if (r < LY::RYL) return -dzn[r - LY::RZH];
            const int i = r < LY::RYH ? r - LY::RYL : r - LY::RYH;
            double cd = 0.0;
            for (int a = 0; a < NS; a++) cd += P.Cy[i][a] * dzk[a];
            return r < LY::RYH ? cd + dsig[NY + i] : dsig[i] - cd;
        };
        // sweep over all rows with the current direction: calls f(k, r, t, l, dt, dl_) ; k = -1 for the sig rows (r = j)
        auto for_rows = [&](const double target, const bool with_pp, auto &&f) {
            double dzk[NS];
            for (int i = 0; i < NS; i++) dzk[i] = 0.0;
            double zk[NS];
            for (int i = 0; i < NS; i++) zk[i] = P.z0[i];
            for (int k = 0; k < N; k++) {
                double du[NU], dzn[NS], u[NU], zn[NS];
                for (int i = 0; i < NU; i++) { double a = W(k, LY::DU + i); for (int c = 1; c < NCOL; c++) a += W(k, LY::DU + c * NU + i) * dsig[c - 1]; du[i] = a; u[i] = W(k, LY::U + i); }
                for (int i = 0; i < NS; i++) { double a = W(k, LY::DZ + i); for (int c = 1; c < NCOL; c++) a += W(k, LY::DZ + c * NS + i) * dsig[c - 1]; dzn[i] = a; zn[i] = W(k, LY::Z + i); }
                for (int r = 0; r < NC; r++) {
                    if (!row_on(k, r)) continue;
                    const double t = W(k, LY::T + r), l = W(k, LY::L + r);
                    const double rp = row_value(k, r, u, zk, zn) - t;
                    const double dt = dval(k, r, du, dzk, dzn) + rp;
                    const double rc = t * l - (with_pp ? smax(target, l * kSFloor) : 0.0) + (with_pp ? W(k, LY::PP + r) : 0.0);
                    f(k, r, t, l, dt, (-rc - l * dt) / t);
                }
                for (int i = 0; i < NS; i++) { dzk[i] = dzn[i]; zk[i] = zn[i]; }
            }
            for (int j = 0; j < NSL; j++) {
                const double rp = sig[j] - tsg[j], dt = dsig[j] + rp;
                const double rc = tsg[j] * lsg[j] - (with_pp ? smax(target, lsg[j] * kSFloor) : 0.0) + (with_pp ? ppsg[j] : 0.0);
                f(-1, j, tsg[j], lsg[j], dt, (-rc - lsg[j] * dt) / tsg[j]);
            }
        };
        // ---- predictor ----
        solve_dir(0.0, false);
        double m_aff = 1.0, s1 = 0.0, s2 = 0.0;
        for_rows(0.0, false, [&](int k, int r, double t, double l, double dt, double dl_) {
            m_aff = smax(m_aff, smax(-dt / t, -dl_ / l));
            s1 += t * dl_ + l * dt; s2 += dt * dl_;
            if (k >= 0) W(k, LY::PP + r) = dt * dl_; else ppsg[r] = dt * dl_;
        });
        const double a_aff = 1.0 / m_aff;
        const double mu_aff = (mu_sum + a_aff * s1 + a_aff * a_aff * s2) * inv_ncon;
        const double rat = mu > 0.0 ? mu_aff / mu : 0.0;
        const double sm = smax(rat * rat * rat * mu, kMuFloor);
        // ---- corrector ----
        solve_dir(sm, true);
        double m_cc = kTau;
        for_rows(sm, true, [&](int, int, double t, double l, double dt, double dl_) { m_cc = smax(m_cc, smax(-dt / t, -dl_ / l)); });
        alpha = m_cc <= kTau ? 1.0 : kTau / m_cc;
        // ---- the step ----
        for_rows(sm, true, [&](int k, int r, double t, double l, double dt, double dl_) {
            if (k >= 0) { W(k, LY::T + r) = t + alpha * dt; W(k, LY::L + r) = l + alpha * dl_; } else { tsg[r] = t + alpha * dt; lsg[r] = l + alpha * dl_; }
        });
        for (int k = 0; k < N; k++) {
            for (int i = 0; i < NU; i++) { double a = W(k, LY::DU + i); for (int c = 1; c < NCOL; c++) a += W(k, LY::DU + c * NU + i) * dsig[c - 1]; W(k, LY::U + i) += alpha * a; }
            for (int i = 0; i < NS; i++) { double a = W(k, LY::DZ + i); for (int c = 1; c < NCOL; c++) a += W(k, LY::DZ + c * NS + i) * dsig[c - 1]; W(k, LY::Z + i) += alpha * a; }
        }
        for (int j = 0; j < NSL; j++) sig[j] += alpha * dsig[j];
    }
    return status;
}

}  // namespace mpc
