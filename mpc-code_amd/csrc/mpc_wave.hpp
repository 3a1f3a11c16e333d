// Wave-autonomous closed-loop solver: one wavefront owns four MPC instances from the first step of a launch to the last.
//
// The workgroup IS the wave (64 threads, no inter-wave barrier anywhere), so an instance never waits for anything but its
// three neighbours in the wave, and the batch spreads as B / 4 independent wavefronts over the 1024 SIMDs of the chip
// (batch 4096 = one wave per SIMD).  Between the first and the last step of a launch nothing but logs goes to HBM:
//   * the interior-point iterate of each of the four instances (slacks, multipliers, inputs, states: lane = block k of the
//     horizon) stays in registers for the whole launch - it is also the warm start of the next step, shifted by one lane;
//   * the closed-loop state (plant state, estimate, filter covariance, previous target, target warm start) stays in LDS;
//   * everything sequential in the horizon runs in *tile form* on the fp64 matrix cores: lane 16 r + 4 b + c holds element
//     (r, c) of a 4x4 tile of instance b, v_mfma_f64_4x4x4f64 multiplies the four instances' tiles at once (M'S + C).
//     Riccati factorisation with the predictor's right-hand-side recursion fused into it, then the direction (forward) and,
//     for the corrector, right-hand side (backward) recursions with one matrix-core product on the dependent chain per block:
//         p_k  = Acl_k'(h_z + p_{k+1}) + K_k' h_u          kff_k = -Lambda_k^-1 (h_u + B'(h_z + p_{k+1}))
//         dz+  = Acl_k dz + B kff_k                        du_k  = K_k dz + kff_k            (Acl = A + B K)
//     vectors are carried replicated over the four tile columns, which costs nothing and lets one masked ds_write store two
//     results.
// The two layouts (lane = block for the element-wise phases, tile for the recursions) meet in a transposing buffer in LDS,
// T[row][instance][65]: sigma -> K | Lambda^-1, g + h -> du | dz, kff; 16 rows x 4 x 65 doubles = 33 KB per wave, four waves
// per CU.
//
// The arithmetic per block is that of rpdip_lane / tp_solve: same constants, same tests, same warm
// start; sums over the horizon are taken in tree order and the tile products sum over a zero-padded inner dimension.
// Reference semantics: solver(...) on opt_dyn's NLP, Control_Calc.py:20-260 + MPC_code.py:733-805.
#pragma once
#include "mpc_device.hpp"
#include "mpc_tp.hpp"

namespace mpc {

template <int NS, int NU, int NC>
struct WvCfg {
    static constexpr int NI = 4;                                       // instances per wave = tiles per matrix-core product
    static constexpr int NV = NS + NU, NKF = NU * NS, NLI = NU * (NU + 1) / 2;
    // rows of the transposing buffer, per (instance, block):
    //   RA: sigma (element-wise -> Riccati), overwritten by K | Lambda^-1 (kept for the corrector)
    //   RG: gu + hu | gz + hz (-> rhs recursion), then du | dz (direction ->); the same again for the corrector
    //   RK: kff
    static constexpr int RA = 0, RA_SZ = (NC > NKF + NLI ? NC : NKF + NLI);
    static constexpr int RG = RA_SZ, RK = RG + NV, ROWS = RK + NU;
    static constexpr int LD = 65;                                      // odd: the tile view (row, instance) -> distinct banks
    static constexpr int T_DOUBLES = ROWS * NI * LD;
    static constexpr int QN = 5 * NS + 2 * NU + 1;                     // z0 zr c zlo zhi | ur us | ws_delta
    static constexpr int ROWS_WS = NU + 2 * NC;                        // warm start kept in HBM between launches: u | l_lo | l_hi
    static constexpr int OUT = NU + NS;                                // first input / next state of the final iterate
    __host__ __device__ static constexpr size_t lds_doubles(int keep_per_inst) { return (size_t)T_DOUBLES + NI * QN + NI * OUT + NI * keep_per_inst; }
};

template <int NS, int NU, int NC>
struct WvIter { double sl[NC], sh[NC], ll[NC], lh[NC], u[NU], z[NS]; };      // one block of one instance (lane = block)

struct WvInst { double mu, mu_sum, sm, inv_ncon, gscale; int stall, iters, status; bool on, warm; };

enum : int { kWvOk0 = 1, kWvWarm = 2, kWvValid = 4 };

// sums / maxima over the 16 lanes of a DPP row (= one instance of the target problem's constraint rows); result in every lane
__device__ __forceinline__ double row16_sum(double v)
{
    v += dpp_move<0xB1, 0xF>(0.0, v);
    v += dpp_move<0x4E, 0xF>(0.0, v);
    v += dpp_move<0x141, 0xF>(0.0, v);
    v += dpp_move<0x140, 0xF>(0.0, v);
    return v;
}
__device__ __forceinline__ double row16_max(double v)
{
    v = dmax(v, dpp_move<0xB1, 0xF>(v, v));
    v = dmax(v, dpp_move<0x4E, 0xF>(v, v));
    v = dmax(v, dpp_move<0x141, 0xF>(v, v));
    v = dmax(v, dpp_move<0x140, 0xF>(v, v));
    return v;
}

// Solves the four OCPs of this wave.  T: transposing buffer; q: instance data [4][QN] (z0 zr c zlo zhi | ur us | delta) and
// iflag[4] (kWv*) written by the caller; X: the resident iterates - on entry the previous step's final iterate (used when
// kWvWarm), on return this step's.  S[j].status / iters: verdicts.
template <int NS, int NU, bool HASM, int NC, bool MASKED>
__device__ __forceinline__ void wv_solve(const DevProblem &P, double *__restrict__ T, const double *__restrict__ q, const int *__restrict__ iflag,
                                         WvIter<NS, NU, NC> (&X)[4], WvInst (&S)[4], int max_iter)
{
    using Cfg = WvCfg<NS, NU, NC>;
    using Iter = WvIter<NS, NU, NC>;
    constexpr int NV = Cfg::NV, NKF = Cfg::NKF, NLI = Cfg::NLI, RA = Cfg::RA, RG = Cfg::RG, RK = Cfg::RK, NI = Cfg::NI, LD = Cfg::LD;
    static_assert(NS <= 4 && NU <= 2, "the stage has to fit one 4x4 tile");
    const int lane = threadIdx.x, N = P.N;
    const int k = lane;
    const bool blk_on = k < N, last = k == N - 1;
    auto tk = [&](int row, int inst) -> double & { return T[(row * NI + inst) * LD + k]; };      // lane = block view

    struct Bnd { double lo[NC], hi[NC]; bool fl[NC], fh[NC]; };
    auto bounds = [&](int j, Bnd &Bd) {
        const double *qd = q + j * Cfg::QN;
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            const double zlm = i >= NU ? qd[3 * NS + (i >= NU ? i - NU : 0)] : 0.0, zhm = i >= NU ? qd[4 * NS + (i >= NU ? i - NU : 0)] : 0.0;
            const double lm = i < NU ? P.ulo[i < NU ? i : 0] : zlm, hm = i < NU ? P.uhi[i < NU ? i : 0] : zhm;
            const double le = i < NU ? P.ulo[i < NU ? i : 0] : P.zlo_e[i >= NU ? i - NU : 0], he = i < NU ? P.uhi[i < NU ? i : 0] : P.zhi_e[i >= NU ? i - NU : 0];
            const double lo = last ? le : lm, hi = last ? he : hm;
            Bd.fl[i] = MASKED ? fin(lo) : true; Bd.fh[i] = MASKED ? fin(hi) : true;
            Bd.lo[i] = Bd.fl[i] ? lo : 0.0; Bd.hi[i] = Bd.fh[i] ? hi : 0.0;
        }
    };
    // cost gradient of the current point for this lane's block: gu (NU), gz (NS), with the bound multipliers
    auto gradient = [&](int j, const Iter &Xj, double (&gu)[NU], double (&gz)[NS]) {
        const double *qd = q + j * Cfg::QN;
        double dz1[NS], du[NU];
        MPC_UNROLL for (int i = 0; i < NS; i++) dz1[i] = Xj.z[i] - qd[NS + i];
        MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = Xj.u[i] - qd[5 * NS + i];
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            double a = (NU + i < NC) ? Xj.lh[NU + i < NC ? NU + i : 0] - Xj.ll[NU + i < NC ? NU + i : 0] : 0.0;
            MPC_UNROLL for (int l = 0; l < NS; l++) a += (last ? P.Pf[i][l] : P.Q[i][l]) * dz1[l];
            gz[i] = a;
        }
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            double a = Xj.lh[i] - Xj.ll[i];
            MPC_UNROLL for (int l = 0; l < NU; l++) a += P.R[i][l] * du[l];
            gu[i] = a;
        }
        if (HASM) {     // cross terms of the Delta-u form: M (u_{k+1} - ur) into gz (k < N-1), M'(z_k - zr) into gu
            double un[NU], zp[NS];
            MPC_UNROLL for (int i = 0; i < NU; i++) un[i] = __shfl_down(du[i], 1, 64);
            MPC_UNROLL for (int i = 0; i < NS; i++) { const double t = __shfl_up(dz1[i], 1, 64); zp[i] = k > 0 ? t : qd[i] - qd[NS + i]; }
            if (!last) { MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int l = 0; l < NU; l++) gz[i] += P.M[i][l] * un[l]; } }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int l = 0; l < NS; l++) gu[i] += P.M[l][i] * zp[l]; }
        }
    };
    // Residuals, barrier weights, gradients of the iterate -> LDS; then the convergence test of this iterate: the stationarity
    // residual needs the costates pi_k = gz_k + A' pi_{k+1}, a linear recursion with a constant matrix, taken as a parallel
    // scan over the lanes (log2(64) steps with A^(2^e)).
    auto phase_a = [&](int j, WvInst &Sj, const Iter &Xj, int it) {
        Bnd Bd; bounds(j, Bd);
        double mu_p = 0.0, resp_p = 0.0, cres_p = 0.0, lmax_p = 0.0, hb[NC];
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
            const double rh = Bd.fh[i] ? v + Xj.sh[i] - Bd.hi[i] : 0.0, rl = Bd.fl[i] ? v - Xj.sl[i] - Bd.lo[i] : 0.0;
            const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
            mu_p += Xj.sl[i] * Xj.ll[i] + Xj.sh[i] * Xj.lh[i];
            tk(RA + i, j) = Xj.ll[i] * isl + Xj.lh[i] * ish;
            hb[i] = Xj.lh[i] * (rh * ish - 1.0) + Xj.ll[i] * (rl * isl + 1.0);
            resp_p = dmax(resp_p, dmax(fabs(rl), fabs(rh)));
            cres_p = dmax(cres_p, dmax(comp_measure(Xj.sl[i], Xj.ll[i]), comp_measure(Xj.sh[i], Xj.lh[i])));
            lmax_p = dmax(lmax_p, dmax(Xj.ll[i], Xj.lh[i]));
        }
        double gu[NU], gz[NS], pi[NS];
        gradient(j, Xj, gu, gz);
        MPC_UNROLL for (int i = 0; i < NU; i++) tk(RG + i, j) = gu[i] + hb[i];
        MPC_UNROLL for (int i = 0; i < NS; i++) { tk(RG + NU + i, j) = gz[i] + (NU + i < NC ? hb[NU + i < NC ? NU + i : 0] : 0.0); pi[i] = blk_on ? gz[i] : 0.0; }
        MPC_UNROLL for (int e = 0; e < 6; e++) {
            const int d = 1 << e;
            if (d < N) {
                double t[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) { const double v = __shfl_down(pi[i], d, 64); t[i] = (k + d < N) ? v : 0.0; }
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = pi[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += P.Apow[e][l][i] * t[l]; pi[i] = a; }
            }
        }
        double rs_p = 0.0;
        MPC_UNROLL for (int i = 0; i < NU; i++) { double a = gu[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += P.B[l][i] * pi[l]; rs_p = dmax(rs_p, fabs(a)); }
        Sj.mu_sum = wave_sum(blk_on ? mu_p : 0.0);
        const double res_p = wave_max(blk_on ? resp_p : 0.0), cres = wave_max(blk_on ? cres_p : 0.0), lmax = wave_max(blk_on ? lmax_p : 0.0);
        const double res_s = wave_max(blk_on ? rs_p : 0.0);
        Sj.mu = Sj.mu_sum * Sj.inv_ncon;
        if (it == 0) Sj.gscale = dmax(1.0, res_s);
        const bool ok_cp = (cres <= 1.0) && (res_p <= kTolFeas);
        Sj.stall = ok_cp ? Sj.stall + 1 : 0;
        int verdict = -1;
        if (ok_cp && (res_s <= kTolStat * Sj.gscale || (Sj.stall > kStallMax && res_s <= kTolStatAcc * Sj.gscale))) verdict = kSolved;
        else if (lmax > kInfeasZ * Sj.gscale || !(fabs(Sj.mu) < 1.0e300)) verdict = kInfeasible;
        else if (it == max_iter) verdict = kMaxIter;
        if (verdict >= 0) { Sj.on = false; Sj.status = verdict; Sj.iters = it; }
    };

    // ---- instance constants and the initial point (cold: us pushed inside the box; warm: previous iterate shifted one stage)
    MPC_UNROLL for (int j = 0; j < NI; j++) {
        WvInst &Sj = S[j];
        Iter &Xj = X[j];
        const double *qd = q + j * Cfg::QN;
        const int myflag = __builtin_amdgcn_readfirstlane(iflag[j]);
        Sj.on = (myflag & kWvValid) && (myflag & kWvOk0);
        Sj.warm = (myflag & kWvWarm) != 0;
        Sj.mu = 0.0; Sj.mu_sum = 0.0; Sj.sm = 0.0; Sj.gscale = 1.0; Sj.stall = 0; Sj.iters = 0;
        Sj.status = ((myflag & kWvValid) && !(myflag & kWvOk0)) ? kInfeasible : kMaxIter;
        double ncon = 0.0;
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            const double zlm = i >= NU ? uni(qd[3 * NS + (i >= NU ? i - NU : 0)]) : 0.0, zhm = i >= NU ? uni(qd[4 * NS + (i >= NU ? i - NU : 0)]) : 0.0;
            const double lm = i < NU ? P.ulo[i < NU ? i : 0] : zlm, hm = i < NU ? P.uhi[i < NU ? i : 0] : zhm;
            const double le = i < NU ? P.ulo[i < NU ? i : 0] : P.zlo_e[i >= NU ? i - NU : 0], he = i < NU ? P.uhi[i < NU ? i : 0] : P.zhi_e[i >= NU ? i - NU : 0];
            const bool flm = MASKED ? fin(lm) : true, fhm = MASKED ? fin(hm) : true, fle = MASKED ? fin(le) : true, fhe = MASKED ? fin(he) : true;
            ncon += (double)(N - 1) * ((flm ? 1 : 0) + (fhm ? 1 : 0)) + (fle ? 1 : 0) + (fhe ? 1 : 0);
        }
        Sj.inv_ncon = 1.0 / dmax(ncon, 1.0);
        if (Sj.on) {
            const bool rep = k >= N - 1;       // shift by one stage, the last block repeats
            double ll0[NC], lh0[NC];
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                const double ulo = P.ulo[i], uhi = P.uhi[i];
                const bool f_lo = fin(ulo), f_hi = fin(uhi);
                double v;
                if (Sj.warm) {
                    const double t = __shfl_down(Xj.u[i], 1, 64);
                    v = rep ? Xj.u[i] : t;
                    if (f_lo) v = dmax(v, ulo);
                    if (f_hi) v = dmin(v, uhi);
                } else {
                    const double us = qd[5 * NS + NU + i];
                    double push;
                    if (f_lo && f_hi) push = 0.1 * (uhi - ulo);
                    else push = 0.1 * dmax(1.0, fabs(f_lo ? ulo : (f_hi ? uhi : 0.0)));
                    v = us;
                    if (f_lo) v = dmax(v, ulo + push);
                    if (f_hi) v = dmin(v, uhi - push);
                }
                Xj.u[i] = v;
            }
            MPC_UNROLL for (int i = 0; i < NC; i++) {
                if (Sj.warm) {
                    const double tl = __shfl_down(Xj.ll[i], 1, 64), th = __shfl_down(Xj.lh[i], 1, 64);
                    ll0[i] = rep ? Xj.ll[i] : tl; lh0[i] = rep ? Xj.lh[i] : th;
                } else { ll0[i] = 0.0; lh0[i] = 0.0; }
            }
            // states of the initial point by forward simulation z_{k+1} = A z_k + B u_k + c: a linear recursion with a constant
            // matrix, taken as a scan over the lanes with A^(2^e) (lane k ends up with z_{k+1})
            {
                double xk[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    double a = qd[2 * NS + i];
                    MPC_UNROLL for (int l = 0; l < NU; l++) a += P.B[i][l] * Xj.u[l];
                    if (k == 0) { MPC_UNROLL for (int l = 0; l < NS; l++) a += P.A[i][l] * qd[l]; }
                    xk[i] = blk_on ? a : 0.0;
                }
                MPC_UNROLL for (int e = 0; e < 6; e++) {
                    const int d = 1 << e;
                    if (d < N) {
                        double t[NS];
                        MPC_UNROLL for (int i = 0; i < NS; i++) { const double v = __shfl_up(xk[i], d, 64); t[i] = k >= d ? v : 0.0; }
                        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = xk[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += P.Apow[e][i][l] * t[l]; xk[i] = a; }
                    }
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) Xj.z[i] = xk[i];
            }
            Bnd Bd; bounds(j, Bd);
            const double ws_delta = uni(qd[5 * NS + 2 * NU]);
            const double ws_smin = dmin(dmax(kWsKappa * ws_delta, kWsSMinLo), kWsSMinHi), ws_mu = kWsMuFactor * ws_smin * ws_smin;
            const double smin = Sj.warm ? ws_smin : kSMin;
            MPC_UNROLL for (int i = 0; i < NC; i++) {
                const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                Xj.sl[i] = Bd.fl[i] ? dmax(v - Bd.lo[i], smin) : 1.0; Xj.sh[i] = Bd.fh[i] ? dmax(Bd.hi[i] - v, smin) : 1.0;
                const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                const double llo = Sj.warm ? dmax(ll0[i], ws_mu * isl) : kMu0 * isl, lhi = Sj.warm ? dmax(lh0[i], ws_mu * ish) : kMu0 * ish;
                Xj.ll[i] = Bd.fl[i] ? llo : 0.0; Xj.lh[i] = Bd.fh[i] ? lhi : 0.0;
            }
            phase_a(j, Sj, Xj, 0);
        }
    }

    // ---- tile view ------------------------------------------------------------------------------------------------------
    // lane 16 r + 4 b + c holds element (r, c) of the tile of instance b; mm(M, S, C) = M'S + C on all four tiles at once
    const int tr = lane >> 4, tb = (lane >> 2) & 3, tc = lane & 3;
    const bool in_ss = tr < NS && tc < NS, in_su = tr < NS && tc < NU, in_us = tr < NU && tc < NS, in_uu = tr < NU && tc < NU;
    auto mm = [](double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); };
    double *const Tt = T + tb * LD;                                        // this lane's instance
    auto tt = [&](int row, int kk) -> double & { return Tt[row * (NI * LD) + kk]; };
    // model tiles (constants of the problem)
    const double Ar = in_ss ? P.A[tr][tc] : 0.0, Atr = in_ss ? P.A[tc][tr] : 0.0, Br = in_su ? P.B[tr][tc] : 0.0, Btr = in_us ? P.B[tc][tr] : 0.0;
    // vectors live in column-replicated tiles: rows of hu / hz / kff for this lane (row RG as a harmless dummy) and their 0 / 1 weights
    const int r_hu = tr < NU ? RG + tr : RG, r_hz = tr < NS ? RG + NU + tr : RG, r_kf = tr < NU ? RK + tr : RG;
    const double w_u = tr < NU ? 1.0 : 0.0, w_z = tr < NS ? 1.0 : 0.0;
    // K (rows < NU, columns < NS), K' and Lambda^-1 (symmetric, rows / columns < NU) as tiles
    const int r_k = in_us ? RA + tr * NS + tc : RA, r_kt = in_su ? RA + tc * NS + tr : RA;
    const int li_i = tr > tc ? tr : tc, li_j = tr > tc ? tc : tr;
    const int r_li = in_uu ? RA + NKF + li_i * (li_i + 1) / 2 + li_j : RA;
    const double w_k = in_us ? 1.0 : 0.0, w_kt = in_su ? 1.0 : 0.0, w_li = in_uu ? 1.0 : 0.0;
    bool pd_all = true;         // every Lambda of this lane's instance was positive definite so far

    // backward: Riccati factorisation (sigma -> K, Lambda^-1) with the right-hand-side recursion of the predictor behind it
    auto tile_factor = [&](bool t_on) {
        const double Rr = in_uu ? P.R[tr][tc] : 0.0, Qr = in_ss ? P.Q[tr][tc] : 0.0, Mtr = (HASM && in_us) ? P.M[tc][tr] : 0.0;
        // rows 0 / 1 of Lambda = R~ + B'PB broadcast over all tile rows, straight from PB: (B E_i)' PB + E_i' R~
        const double BE0 = tr < NS ? P.B[tr][0] : 0.0, BE1 = (NU > 1 && tr < NS) ? P.B[tr][NU > 1 ? 1 : 0] : 0.0;
        const double RE0 = tc < NU ? P.R[0][tc] : 0.0, RE1 = (NU > 1 && tc < NU) ? P.R[NU > 1 ? 1 : 0][tc] : 0.0;
        const double Ir = tr == tc ? 1.0 : 0.0;
        // 0 / 1 weights: sigma_z[r] on the diagonal of P, sigma_u[r] on the diagonal of R~, sigma_u[c] in column c of the broadcast rows
        const bool dz_on = tr == tc && tr < NS && NU + tr < NC, du_on = tr == tc && tr < NU, db_on = tc < NU;
        const double mz = dz_on ? 1.0 : 0.0, mu_ = du_on ? 1.0 : 0.0, mb0 = tc == 0 ? 1.0 : 0.0, mb1 = (NU > 1 && tc == 1) ? 1.0 : 0.0;
        const int rz = dz_on ? RA + NU + tr : RA, ru = du_on ? RA + tr : RA, rb = db_on ? RA + tc : RA;
        // K goes to LDS from the lanes that hold it (rows < NU of the tile), Lambda^-1 from rows 2..3, which compute it as well: one
        // store per block; kff (valid in rows < NU of every column) leaves from column 0
        const bool st_k = in_us, st_l = tr >= 2 && tc < NU && tc <= tr - 2 && tr - 2 < NU;
        const int st_row = st_k ? RA + tr * NS + tc : (st_l ? RA + NKF + (tr - 2) * (tr - 1) / 2 + tc : RA);
        const bool st_on = t_on && (st_k || st_l), st_f = t_on && tr < NU && tc == 0;
        double Pm = in_ss ? P.Pf[tr][tc] : 0.0, PC = 0.0;
        bool pd_ok = true;
        double szn = tt(rz, N - 1), sun = tt(ru, N - 1), sbn = tt(rb, N - 1), hun = tt(r_hu, N - 1), hzn = tt(r_hz, N - 1);
        for (int kk = N - 1; kk >= 0; kk--) {
            const double sz = szn, su = sun, sb = sbn, HU = hun * w_u, HZ = hzn * w_z;
            const int kn = kk > 0 ? kk - 1 : 0;      // next block's data now, they arrive while this block computes
            szn = tt(rz, kn); sun = tt(ru, kn); sbn = tt(rb, kn); hun = tt(r_hu, kn); hzn = tt(r_hz, kn);
            Pm = __builtin_fma(sz, mz, Pm);
            const double PA = mm(Pm, Ar, 0.0), PB = mm(Pm, Br, 0.0), BtP = mm(Br, Pm, 0.0);
            const double Rs = __builtin_fma(su, mu_, Rr);
            const double Psi = mm(Br, PA, Mtr);                             // M' + B'PA
            // Lambda^-1 (NU <= 2): every lane gets the numbers it is made of (columns via the quad), then forms the element (r mod 2, c)
            double Lall;
            const double X0 = mm(BE0, PB, __builtin_fma(sb, mb0, RE0));
            if (NU == 1) {
                const double a = dpp_move<0x00, 0xF>(X0, X0);
                pd_ok = pd_ok && (a > 0.0);
                Lall = tc == 0 ? frcp(a) : 0.0;
            } else {
                const double X1 = mm(BE1, PB, __builtin_fma(sb, mb1, RE1));
                const double a = dpp_move<0x00, 0xF>(X0, X0), off = dpp_move<0x55, 0xF>(X0, X0), d = dpp_move<0x55, 0xF>(X1, X1);
                const double det = a * d - off * off, rdet = frcp(det);
                pd_ok = pd_ok && (a > 0.0) && (det > 0.0);
                Lall = tc < 2 ? ((tr & 1) == tc ? (tc == 0 ? d : a) * rdet : -off * rdet) : 0.0;
            }
            const double Li = tr < NU ? Lall : 0.0;
            const double Kk = mm(-Li, Psi, 0.0);      // K = -Lambda^-1 Psi
            const double Acl = mm(Btr, Kk, Ar);       // A + B K
            // right-hand side of the predictor for this block (off the chain of the matrix recursion)
            const double PV = HZ + PC;
            const double PSIv = mm(Br, PV, HU);       // hu + B'(hz + p+)
            const double KFF = mm(-Li, PSIv, 0.0);    // -Lambda^-1 psi, in every column
            if (st_on) tt(st_row, kk) = st_k ? Kk : Lall;
            if (st_f) tt(r_kf, kk) = KFF;
            if (kk > 0) {       // closed-loop (Joseph) form: Q + Acl' P Acl + K' R~ K (+ M K + K' M')
                const double RK_ = mm(Rs, Kk, 0.0), Tm = mm(BtP, Kk, PA);      // R~ K,  P Acl = PA + PB K
                double Pn = mm(Acl, Tm, mm(Kk, RK_, Qr));
                if (HASM) { const double MK = mm(Mtr, Kk, 0.0); Pn = mm(MK, Ir, Pn + MK); }
                Pm = Pn;      // symmetric up to rounding; the recursion does not amplify the difference
                PC = mm(Acl, PV, mm(Kk, HU, 0.0));      // Acl'(hz + p+) + K' hu
            }
        }
        pd_all = pd_all && (pd_ok || !t_on);
    };
    // backward: right-hand-side recursion of the corrector: h (RG rows), K, Lambda^-1 -> kff
    auto tile_rhs = [&](bool t_on) {
        const bool st_on = t_on && tr < NU && tc == 0;
        double PC = 0.0;
        double hun = tt(r_hu, N - 1), hzn = tt(r_hz, N - 1), kn_ = tt(r_k, N - 1), lin = tt(r_li, N - 1);
        for (int kk = N - 1; kk >= 0; kk--) {
            const double HU = hun * w_u, HZ = hzn * w_z, Kk = kn_ * w_k, mLi = -lin * w_li;
            const int kn = kk > 0 ? kk - 1 : 0;
            hun = tt(r_hu, kn); hzn = tt(r_hz, kn); kn_ = tt(r_k, kn); lin = tt(r_li, kn);
            const double PV = HZ + PC;
            const double KFF = mm(mLi, mm(Br, PV, HU), 0.0);
            if (st_on) tt(r_kf, kk) = KFF;
            if (kk > 0) {
                const double Acl = mm(Btr, Kk, Ar);
                PC = mm(Acl, PC, mm(Acl, HZ, mm(Kk, HU, 0.0)));
            }
        }
    };
    // forward: Newton direction: K, kff -> du | dz
    auto tile_forward = [&](bool t_on) {
        const bool st_u = tr < NU && tc == 0, st_z = tr < NS && tc == 1;
        const int st_row = st_u ? RG + tr : (st_z ? RG + NU + tr : RG);
        const bool st_on = t_on && (st_u || st_z);
        double DZ = 0.0;
        double kn_ = tt(r_k, 0), ktn = tt(r_kt, 0), kfn = tt(r_kf, 0);
        for (int kk = 0; kk < N; kk++) {
            const double Kk = kn_ * w_k, KkT = ktn * w_kt, KFF = kfn * w_u;
            const int kx = kk + 1 < N ? kk + 1 : kk;
            kn_ = tt(r_k, kx); ktn = tt(r_kt, kx); kfn = tt(r_kf, kx);
            const double DU = mm(KkT, DZ, KFF);                       // K dz + kff
            const double AclT = mm(Kk, Btr, Atr);                     // (A + B K)'
            const double DZn = mm(AclT, DZ, mm(Btr, KFF, 0.0));       // Acl dz + B kff
            if (st_on) tt(st_row, kk) = st_u ? DU : DZn;
            DZ = DZn;
        }
    };

    double dvp[NI][NC];      // predictor direction of each instance's bounded variables, kept for the corrector's second-order terms
    for (int it = 0;; it++) {
        if (!(S[0].on || S[1].on || S[2].on || S[3].on)) break;      // wave-uniform: every instance has its verdict
        const bool t_on = tb == 0 ? S[0].on : (tb == 1 ? S[1].on : (tb == 2 ? S[2].on : S[3].on));
        __syncthreads();
        tile_factor(t_on);
        __syncthreads();
        tile_forward(t_on);
        __syncthreads();
        {
            const unsigned long long bad = __ballot(!pd_all);      // a Lambda lost definiteness: the instance stops as infeasible
            MPC_UNROLL for (int j = 0; j < NI; j++) {
                if (S[j].on && (bad & (0x000F000F000F000FULL << (4 * j)))) { S[j].on = false; S[j].status = kInfeasible; S[j].iters = it; }
            }
        }
        // ================= element-wise: predictor step length, centring, corrector rhs -> LDS ===========================
        MPC_UNROLL for (int j = 0; j < NI; j++) {
            WvInst &Sj = S[j];
            if (Sj.on) {
                const Iter &Xj = X[j];
                Bnd Bd; bounds(j, Bd);
                double maff_p = 1.0, s1_p = 0.0, s2_p = 0.0, pl[NC], ph[NC];
                MPC_UNROLL for (int i = 0; i < NC; i++) dvp[j][i] = tk(RG + i, j);      // du | dz of the predictor
                MPC_UNROLL for (int i = 0; i < NC; i++) {
                    const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                    const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                    const double rh = Bd.fh[i] ? v + Xj.sh[i] - Bd.hi[i] : 0.0, rl = Bd.fl[i] ? v - Xj.sl[i] - Bd.lo[i] : 0.0;
                    const double dsh = Bd.fh[i] ? -rh - dvp[j][i] : 0.0, dsl = Bd.fl[i] ? rl + dvp[j][i] : 0.0;
                    const double qh = dsh * ish, ql = dsl * isl;
                    const double dlh = Bd.fh[i] ? -Xj.lh[i] - Xj.lh[i] * qh : 0.0, dll = Bd.fl[i] ? -Xj.ll[i] - Xj.ll[i] * ql : 0.0;
                    maff_p = dmax(maff_p, dmax(-ql, -qh));
                    if (Bd.fl[i]) maff_p = dmax(maff_p, 1.0 + ql);
                    if (Bd.fh[i]) maff_p = dmax(maff_p, 1.0 + qh);
                    s1_p += Xj.sl[i] * dll + Xj.ll[i] * dsl + Xj.sh[i] * dlh + Xj.lh[i] * dsh;
                    s2_p += dsl * dll + dsh * dlh;
                    pl[i] = dsl * dll; ph[i] = dsh * dlh;
                }
                const double m_aff = wave_max(blk_on ? maff_p : 1.0), s1 = wave_sum(blk_on ? s1_p : 0.0), s2 = wave_sum(blk_on ? s2_p : 0.0);
                const double a_aff = frcp(m_aff);
                const double mu_aff = (Sj.mu_sum + a_aff * s1 + a_aff * a_aff * s2) * Sj.inv_ncon;
                const double rat = Sj.mu > 0.0 ? mu_aff * frcp(Sj.mu) : 0.0;
                Sj.sm = dmax(rat * rat * rat * Sj.mu, kMuFloor);
                double gu[NU], gz[NS], hc[NV];
                gradient(j, Xj, gu, gz);
                MPC_UNROLL for (int i = NC; i < NV; i++) hc[i] = 0.0;
                MPC_UNROLL for (int i = 0; i < NC; i++) {
                    const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                    const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                    const double rh = Bd.fh[i] ? v + Xj.sh[i] - Bd.hi[i] : 0.0, rl = Bd.fl[i] ? v - Xj.sl[i] - Bd.lo[i] : 0.0;
                    const double rch = Bd.fh[i] ? Xj.sh[i] * Xj.lh[i] - dmax(Sj.sm, Xj.lh[i] * kSFloor) + ph[i] : 0.0;
                    const double rcl = Bd.fl[i] ? Xj.sl[i] * Xj.ll[i] - dmax(Sj.sm, Xj.ll[i] * kSFloor) + pl[i] : 0.0;
                    hc[i] = (-rch + Xj.lh[i] * rh) * ish + (rcl + Xj.ll[i] * rl) * isl;
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) tk(RG + i, j) = gu[i] + hc[i];
                MPC_UNROLL for (int i = 0; i < NS; i++) tk(RG + NU + i, j) = gz[i] + hc[NU + i];
            }
        }
        const bool t_on2 = tb == 0 ? S[0].on : (tb == 1 ? S[1].on : (tb == 2 ? S[2].on : S[3].on));
        __syncthreads();
        tile_rhs(t_on2);
        __syncthreads();
        tile_forward(t_on2);
        __syncthreads();
        // ================= element-wise: corrector step length, step; then the next iterate's residuals / gradients =====
        MPC_UNROLL for (int j = 0; j < NI; j++) {
            WvInst &Sj = S[j];
            if (Sj.on) {
                Iter &Xj = X[j];
                Bnd Bd; bounds(j, Bd);
                double dvzj[NV];
                MPC_UNROLL for (int i = 0; i < NV; i++) dvzj[i] = tk(RG + i, j);
                double mcc_p = kTau;
                double dsl[NC], dsh[NC], dll[NC], dlh[NC];
                MPC_UNROLL for (int i = 0; i < NC; i++) {
                    const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                    const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                    const double rh = Bd.fh[i] ? v + Xj.sh[i] - Bd.hi[i] : 0.0, rl = Bd.fl[i] ? v - Xj.sl[i] - Bd.lo[i] : 0.0;
                    // second-order products of the predictor direction (recomputed, not stored)
                    const double ash = Bd.fh[i] ? -rh - dvp[j][i] : 0.0, asl = Bd.fl[i] ? rl + dvp[j][i] : 0.0;
                    const double alh = Bd.fh[i] ? -Xj.lh[i] - Xj.lh[i] * (ash * ish) : 0.0, all_ = Bd.fl[i] ? -Xj.ll[i] - Xj.ll[i] * (asl * isl) : 0.0;
                    const double rch = Bd.fh[i] ? Xj.sh[i] * Xj.lh[i] - dmax(Sj.sm, Xj.lh[i] * kSFloor) + ash * alh : 0.0;
                    const double rcl = Bd.fl[i] ? Xj.sl[i] * Xj.ll[i] - dmax(Sj.sm, Xj.ll[i] * kSFloor) + asl * all_ : 0.0;
                    dsh[i] = Bd.fh[i] ? -rh - dvzj[i] : 0.0; dsl[i] = Bd.fl[i] ? rl + dvzj[i] : 0.0;
                    dlh[i] = Bd.fh[i] ? (-rch - Xj.lh[i] * dsh[i]) * ish : 0.0; dll[i] = Bd.fl[i] ? (-rcl - Xj.ll[i] * dsl[i]) * isl : 0.0;
                    mcc_p = dmax(mcc_p, dmax(-dsl[i] * isl, -dsh[i] * ish));
                    if (Bd.fl[i]) mcc_p = dmax(mcc_p, -dll[i] * frcp_approx(Xj.ll[i]));
                    if (Bd.fh[i]) mcc_p = dmax(mcc_p, -dlh[i] * frcp_approx(Xj.lh[i]));
                }
                const double m_cc = wave_max(blk_on ? mcc_p : kTau);
                const double alpha = m_cc <= kTau ? 1.0 : kTau * frcp(m_cc);
                MPC_UNROLL for (int i = 0; i < NC; i++) { Xj.sl[i] += alpha * dsl[i]; Xj.sh[i] += alpha * dsh[i]; Xj.ll[i] += alpha * dll[i]; Xj.lh[i] += alpha * dlh[i]; }
                MPC_UNROLL for (int i = 0; i < NU; i++) Xj.u[i] += alpha * dvzj[i];
                MPC_UNROLL for (int i = 0; i < NS; i++) Xj.z[i] += alpha * dvzj[NU + i];
                phase_a(j, Sj, Xj, it + 1);
            }
        }
    }
}

}  // namespace mpc
