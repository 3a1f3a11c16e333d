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

template <int NS, int NU, int NC, int NI_>
struct WvCfg {
    static constexpr int NI = NI_;                                     // instances per wave (<= 4 = tiles per matrix-core product)
    static constexpr int NV = NS + NU, NKF = NU * NS, NLI = NU * (NU + 1) / 2;
    // rows of the transposing buffer, per (instance, block):
    //   RA: sigma (element-wise -> Riccati), overwritten by K | Lambda^-1 (kept for the corrector)
    //   RG: gu + hu | gz + hz (-> rhs recursion), then du | dz (direction ->); the same again for the corrector
    //   RK: kff
    //   RZ: zeros, written once in the kernel's prologue
    static constexpr int RA = 0, RA_SZ = (NC > NKF + NLI ? NC : NKF + NLI);
    static constexpr int RG = RA_SZ, RK = RG + NV, RZ = RK + NU, ROWS = RZ + 1;      // RZ: a row of zeros (tile lanes without a datum read it)
    static constexpr int LD = 65;                                      // odd: the tile view (row, instance) -> distinct banks
    static constexpr int GUARD = 8;                                    // cells in front of T: run-on reads of the backward loops, stores of idle lanes
    static constexpr int T_DOUBLES = GUARD + ROWS * NI * LD;
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
template <int NS, int NU, bool HASM, int NC, bool MASKED, int NI, class PT>
__device__ __forceinline__ void wv_solve(const PT &P, double *__restrict__ T, const double *__restrict__ q, const int *__restrict__ iflag,
                                         WvIter<NS, NU, NC> (&X)[NI], WvInst (&S)[NI], int max_iter)
{
    using Cfg = WvCfg<NS, NU, NC, NI>;
    using Iter = WvIter<NS, NU, NC>;
    constexpr int NV = Cfg::NV, NKF = Cfg::NKF, NLI = Cfg::NLI, RA = Cfg::RA, RG = Cfg::RG, RK = Cfg::RK, LD = Cfg::LD;
    static_assert(NS <= 4 && NU <= 2 && NI >= 1 && NI <= 4, "the stage has to fit one 4x4 tile; at most four tiles per product");
    const int lane = threadIdx.x, N = P.N;
    const int k = lane;
    const bool blk_on = k < N, last = k == N - 1;
    auto tk = [&](int row, int inst) -> double & { return T[(row * NI + inst) * LD + k]; };      // lane = block view
    MPC_STAMP_INIT

    // the problem pointer made opaque: scalar loads of the constants stay inside the phase that asks for them (hoisted out of the
    // iteration loop they occupy - and spill - scalar registers for the whole solve)
    auto launder = [](const PT &Pin) -> const PT & { const PT *pp = &Pin; asm volatile("" : "+s"(pp)); return *pp; };
    struct Bnd { double lo[NC], hi[NC]; bool fl[NC], fh[NC]; };
    auto bounds = [&](const PT &Pl, int j, Bnd &Bd) {
        const double *qd = q + j * Cfg::QN;
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            const double zlm = i >= NU ? qd[3 * NS + (i >= NU ? i - NU : 0)] : 0.0, zhm = i >= NU ? qd[4 * NS + (i >= NU ? i - NU : 0)] : 0.0;
            const double lm = i < NU ? Pl.ulo[i < NU ? i : 0] : zlm, hm = i < NU ? Pl.uhi[i < NU ? i : 0] : zhm;
            const double le = i < NU ? Pl.ulo[i < NU ? i : 0] : Pl.zlo_e[i >= NU ? i - NU : 0], he = i < NU ? Pl.uhi[i < NU ? i : 0] : Pl.zhi_e[i >= NU ? i - NU : 0];
            const double lo = last ? le : lm, hi = last ? he : hm;
            Bd.fl[i] = MASKED ? fin(lo) : true; Bd.fh[i] = MASKED ? fin(hi) : true;
            Bd.lo[i] = Bd.fl[i] ? lo : 0.0; Bd.hi[i] = Bd.fh[i] ? hi : 0.0;
        }
    };
    // cost gradient of the current point for this lane's block: gu (NU), gz (NS), with the bound multipliers
    auto gradient = [&](const PT &Pl, int j, const Iter &Xj, double (&gu)[NU], double (&gz)[NS]) {
        const double *qd = q + j * Cfg::QN;
        double dz1[NS], du[NU];
        MPC_UNROLL for (int i = 0; i < NS; i++) dz1[i] = Xj.z[i] - qd[NS + i];
        MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = Xj.u[i] - qd[5 * NS + i];
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            double a = (NU + i < NC) ? Xj.lh[NU + i < NC ? NU + i : 0] - Xj.ll[NU + i < NC ? NU + i : 0] : 0.0;
            MPC_UNROLL for (int l = 0; l < NS; l++) a += (last ? Pl.Pf[i][l] : Pl.Q[i][l]) * dz1[l];
            gz[i] = a;
        }
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            double a = Xj.lh[i] - Xj.ll[i];
            MPC_UNROLL for (int l = 0; l < NU; l++) a += Pl.R[i][l] * du[l];
            gu[i] = a;
        }
        if (HASM) {     // cross terms of the Delta-u form: M (u_{k+1} - ur) into gz (k < N-1), M'(z_k - zr) into gu
            double un[NU], zp[NS];
            MPC_UNROLL for (int i = 0; i < NU; i++) un[i] = __shfl_down(du[i], 1, 64);
            MPC_UNROLL for (int i = 0; i < NS; i++) { const double t = __shfl_up(dz1[i], 1, 64); zp[i] = k > 0 ? t : qd[i] - qd[NS + i]; }
            if (!last) { MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int l = 0; l < NU; l++) gz[i] += Pl.M[i][l] * un[l]; } }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int l = 0; l < NS; l++) gu[i] += Pl.M[l][i] * zp[l]; }
        }
    };
    // Residuals, barrier weights, gradients of the iterate -> LDS; then the convergence test of this iterate: the stationarity
    // residual needs the costates pi_k = gz_k + A' pi_{k+1}, a linear recursion with a constant matrix, taken as a parallel
    // scan over the lanes (log2(64) steps with A^(2^e)).
    auto phase_a = [&](const PT &Pl, int j, WvInst &Sj, const Iter &Xj, int it) {
        Bnd Bd; bounds(Pl, j, Bd);
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
        gradient(Pl, j, Xj, gu, gz);
        MPC_UNROLL for (int i = 0; i < NU; i++) tk(RG + i, j) = gu[i] + hb[i];
        MPC_UNROLL for (int i = 0; i < NS; i++) { tk(RG + NU + i, j) = gz[i] + (NU + i < NC ? hb[NU + i < NC ? NU + i : 0] : 0.0); pi[i] = blk_on ? gz[i] : 0.0; }
        MPC_UNROLL for (int e = 0; e < 6; e++) {
            const int d = 1 << e;
            if (d < N) {
                double t[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) { const double v = __shfl_down(pi[i], d, 64); t[i] = (k + d < N) ? v : 0.0; }
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = pi[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pl.Apow[e][l][i] * t[l]; pi[i] = a; }
            }
        }
        double rs_p = 0.0;
        MPC_UNROLL for (int i = 0; i < NU; i++) { double a = gu[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pl.B[l][i] * pi[l]; rs_p = dmax(rs_p, fabs(a)); }
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
        const PT &Pl = P;
        const double *qd = q + j * Cfg::QN;
        const int myflag = __builtin_amdgcn_readfirstlane(iflag[j]);
        Sj.on = (myflag & kWvValid) && (myflag & kWvOk0);
        Sj.warm = (myflag & kWvWarm) != 0;
        Sj.mu = 0.0; Sj.mu_sum = 0.0; Sj.sm = 0.0; Sj.gscale = 1.0; Sj.stall = 0; Sj.iters = 0;
        Sj.status = ((myflag & kWvValid) && !(myflag & kWvOk0)) ? kInfeasible : kMaxIter;
        double ncon = 0.0;
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            const double zlm = i >= NU ? uni(qd[3 * NS + (i >= NU ? i - NU : 0)]) : 0.0, zhm = i >= NU ? uni(qd[4 * NS + (i >= NU ? i - NU : 0)]) : 0.0;
            const double lm = i < NU ? Pl.ulo[i < NU ? i : 0] : zlm, hm = i < NU ? Pl.uhi[i < NU ? i : 0] : zhm;
            const double le = i < NU ? Pl.ulo[i < NU ? i : 0] : Pl.zlo_e[i >= NU ? i - NU : 0], he = i < NU ? Pl.uhi[i < NU ? i : 0] : Pl.zhi_e[i >= NU ? i - NU : 0];
            const bool flm = MASKED ? fin(lm) : true, fhm = MASKED ? fin(hm) : true, fle = MASKED ? fin(le) : true, fhe = MASKED ? fin(he) : true;
            ncon += (double)(N - 1) * ((flm ? 1 : 0) + (fhm ? 1 : 0)) + (fle ? 1 : 0) + (fhe ? 1 : 0);
        }
        Sj.inv_ncon = 1.0 / dmax(ncon, 1.0);
        if (Sj.on) {
            const bool rep = k >= N - 1;       // shift by one stage, the last block repeats
            double ll0[NC], lh0[NC];
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                const double ulo = Pl.ulo[i], uhi = Pl.uhi[i];
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
                    MPC_UNROLL for (int l = 0; l < NU; l++) a += Pl.B[i][l] * Xj.u[l];
                    if (k == 0) { MPC_UNROLL for (int l = 0; l < NS; l++) a += Pl.A[i][l] * qd[l]; }
                    xk[i] = blk_on ? a : 0.0;
                }
                MPC_UNROLL for (int e = 0; e < 6; e++) {
                    const int d = 1 << e;
                    if (d < N) {
                        double t[NS];
                        MPC_UNROLL for (int i = 0; i < NS; i++) { const double v = __shfl_up(xk[i], d, 64); t[i] = k >= d ? v : 0.0; }
                        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = xk[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pl.Apow[e][i][l] * t[l]; xk[i] = a; }
                    }
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) Xj.z[i] = xk[i];
            }
            Bnd Bd; bounds(Pl, j, Bd);
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
            phase_a(Pl, j, Sj, Xj, 0);
        }
    }

    MPC_TSTAMP(1);
    // ---- tile view ------------------------------------------------------------------------------------------------------
    // lane 16 r + 4 b + c holds element (r, c) of the tile of instance b; mm(M, S, C) = M'S + C on all four tiles at once.
    // One wave alone on a SIMD issues one instruction every four cycles whatever its kind, so these loops are written for the
    // lowest instruction count per block:
    //   * a lane that has no use for a quantity reads the zero row RZ instead (never written after the kernel's prologue): no masks;
    //   * a lane that has nothing to store stores to the guard cells in front of T: no exec masking;
    //   * the loops run in groups of PD blocks with the LDS reads of the next group in flight (the last group's look-ahead stays
    //     inside the LDS allocation - guard cells in front, the instance data behind - and is dropped), then the N mod PD blocks left;
    //   * per field one address register stepped once per group, the blocks of a group at immediate offsets.
    auto mm = [](double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); };
    constexpr int RZ = Cfg::RZ;
    // Everything a pass needs per lane is rebuilt at its start from a lane index the compiler cannot see through: as loop
    // invariants of the iteration loop these ~70 registers would stay live across the element-wise phases (and spill there).
    struct TileCtx {
        int tr, tc; bool live, in_ss, in_su, in_us, in_uu;
        double *Tt, *trash;
        double Ar, Atr, Br, Btr;
        double *p_hu, *p_hz, *p_kf, *p_k, *p_kt, *p_li;
    };
    auto tile_ctx = [&](const PT &Pl) {
        int lo = threadIdx.x;
        asm volatile("" : "+v"(lo));
        TileCtx c;
        c.tr = lo >> 4; c.tc = lo & 3;
        const int tbq = (lo >> 2) & 3;
        c.live = tbq < NI;                                                 // tiles >= NI carry zeros (their lanes read RZ, store to the guard)
        const int tb = c.live ? tbq : 0;
        const int tr = c.tr, tc = c.tc;
        c.in_ss = tr < NS && tc < NS; c.in_su = tr < NS && tc < NU; c.in_us = tr < NU && tc < NS; c.in_uu = tr < NU && tc < NU;
        c.Tt = T + tb * LD; c.trash = T - Cfg::GUARD;
        c.Ar = c.in_ss ? Pl.A[tr][tc] : 0.0; c.Atr = c.in_ss ? Pl.A[tc][tr] : 0.0; c.Br = c.in_su ? Pl.B[tr][tc] : 0.0; c.Btr = c.in_us ? Pl.B[tc][tr] : 0.0;
        auto trow = [&](int row) -> double * { return c.Tt + (c.live ? row : RZ) * (NI * LD); };
        // vectors live in column-replicated tiles: hu / hz / kff rows of this lane
        c.p_hu = trow(tr < NU ? RG + tr : RZ); c.p_hz = trow(tr < NS ? RG + NU + tr : RZ); c.p_kf = trow(tr < NU ? RK + tr : RZ);
        // K (rows < NU, columns < NS), K' and -Lambda^-1 (symmetric, rows / columns < NU) as tiles
        const int li_i = tr > tc ? tr : tc, li_j = tr > tc ? tc : tr;
        c.p_k = trow(c.in_us ? RA + tr * NS + tc : RZ); c.p_kt = trow(c.in_su ? RA + tc * NS + tr : RZ);
        c.p_li = trow(c.in_uu ? RA + NKF + li_i * (li_i + 1) / 2 + li_j : RZ);
        return c;
    };
    double pd_min = 1.0;        // smallest pivot / determinant of this lane's instance so far: <= 0 means a Lambda lost definiteness
    constexpr int PD = 4;       // blocks per group

    // backward: Riccati factorisation (sigma -> K, -Lambda^-1) with the right-hand-side recursion of the predictor behind it
    auto tile_factor = [&]() {
        const PT &Pl = launder(P);
        const TileCtx c = tile_ctx(Pl);
        const int tr = c.tr, tc = c.tc; const bool live = c.live, in_ss = c.in_ss, in_us = c.in_us, in_uu = c.in_uu;
        const double Ar = c.Ar, Br = c.Br, Btr = c.Btr; double *const trash = c.trash, *const p_hu = c.p_hu, *const p_hz = c.p_hz, *const p_kf = c.p_kf;
        auto trow = [&](int row) -> double * { return c.Tt + (live ? row : RZ) * (NI * LD); };
        const double Rr = in_uu ? Pl.R[tr][tc] : 0.0, Qr = in_ss ? Pl.Q[tr][tc] : 0.0, Mtr = (HASM && in_us) ? Pl.M[tc][tr] : 0.0;
        // rows 0 / 1 of Lambda = R~ + B'PB broadcast over all tile rows, straight from PB: (B E_i)' PB + E_i' R~
        const double BE0 = tr < NS ? Pl.B[tr][0] : 0.0, BE1 = (NU > 1 && tr < NS) ? Pl.B[tr][NU > 1 ? 1 : 0] : 0.0;
        const double RE0 = tc < NU ? Pl.R[0][tc] : 0.0, RE1 = (NU > 1 && tc < NU) ? Pl.R[NU > 1 ? 1 : 0][tc] : 0.0;
        const double Ir = tr == tc ? 1.0 : 0.0;
        // barrier weights: sigma_z[r] on the diagonal of P, sigma_u[r] on the diagonal of R~, sigma_u[i] at (., i) of broadcast row i
        const bool dz_on = tr == tc && tr < NS && NU + tr < NC, du_on = tr == tc && tr < NU;
        const double *q_sz = trow(dz_on ? RA + NU + tr : RZ) + (N - 1), *q_su = trow(du_on ? RA + tr : RZ) + (N - 1);
        const double *q_s0 = trow(tc == 0 ? RA : RZ) + (N - 1), *q_s1 = trow((NU > 1 && tc == 1) ? RA + 1 : RZ) + (N - 1);
        const double *q_hu = p_hu + (N - 1), *q_hz = p_hz + (N - 1);
        // this lane's element of -adj(Lambda) as a combination of a = L00, d = L11, off = L01 (2 x 2), element (r mod 2, c):
        //   (0,0): -d   (1,1): -a   (0,1), (1,0): +off;  -Lambda^-1 = that / det.  NU = 1: -1 in column 0.
        const int ri = tr & 1;
        const double c_a = (NU > 1 && tc == 1 && ri == 1) ? -1.0 : 0.0, c_d = (NU > 1 && tc == 0 && ri == 0) ? -1.0 : 0.0;
        const double c_o = (NU > 1 && tc < 2 && ri != tc) ? 1.0 : 0.0, c_1 = (NU == 1 && tc == 0) ? -1.0 : 0.0;
        const double w_top = tr < NU ? 1.0 : 0.0;
        const double k_a = c_a * w_top, k_d = c_d * w_top, k_o = c_o * w_top, k_1 = c_1 * w_top;      // the same, rows < NU only (the K product)
        // K goes to LDS from the lanes that hold it (rows < NU of the tile), -Lambda^-1 from rows 2..3, which compute it as well, kff
        // (valid in rows < NU of every column) from column 3 when the state leaves it free: one store per block
        const bool st_k = in_us, st_l = tr >= 2 && tc < NU && tc <= tr - 2 && tr - 2 < NU, st_f3 = NS < 4 && tr < NU && tc == 3;
        const bool st_any = live && (st_k || st_l || st_f3), st_f = live && NS >= 4 && tr < NU && tc == 0;
        double *q_st = st_any ? trow(st_k ? RA + tr * NS + tc : (st_l ? RA + NKF + (tr - 2) * (tr - 1) / 2 + tc : RK + (tr < NU ? tr : 0))) + (N - 1) : trash + PD;
        double *q_sf = st_f ? p_kf + (N - 1) : trash + PD;
        const int stp = st_any ? PD : 0, stpf = st_f ? PD : 0;
        double Pm = in_ss ? Pl.Pf[tr][tc] : 0.0, PC = 0.0;
        double f[PD][6];
        MPC_UNROLL for (int d = 0; d < PD; d++) { f[d][0] = q_sz[-d]; f[d][1] = q_su[-d]; f[d][2] = q_s0[-d]; f[d][3] = NU > 1 ? q_s1[-d] : 0.0; f[d][4] = q_hu[-d]; f[d][5] = q_hz[-d]; }
        auto block = [&](int d, bool more) {
            const double sz = f[d][0], su = f[d][1], s0 = f[d][2], s1 = f[d][3], HU = f[d][4], HZ = f[d][5];
            if (more) { f[d][0] = q_sz[-PD - d]; f[d][1] = q_su[-PD - d]; f[d][2] = q_s0[-PD - d]; if (NU > 1) f[d][3] = q_s1[-PD - d]; f[d][4] = q_hu[-PD - d]; f[d][5] = q_hz[-PD - d]; }
            Pm += sz;
            const double PB = mm(Pm, Br, 0.0), PA = mm(Pm, Ar, 0.0), BtP = mm(Br, Pm, 0.0);
            const double X0 = mm(BE0, PB, RE0 + s0), X1 = NU > 1 ? mm(BE1, PB, RE1 + s1) : 0.0;
            const double Psi = mm(Br, PA, Mtr);                             // M' + B'PA
            const double Rs = Rr + su;
            // every lane gets the numbers Lambda is made of (columns via the quad)
            const double a = dpp_move<0x00, 0xF>(X0, X0);
            double adjm, adjk, det;
            if (NU == 1) { det = a; adjm = c_1; adjk = k_1; pd_min = dmin(pd_min, a); }
            else {
                const double off = dpp_move<0x55, 0xF>(X0, X0), dd = dpp_move<0x55, 0xF>(X1, X1);
                det = __builtin_fma(a, dd, -(off * off));
                adjk = __builtin_fma(k_d, dd, __builtin_fma(k_a, a, k_o * off));
                adjm = __builtin_fma(c_d, dd, __builtin_fma(c_a, a, c_o * off));
                pd_min = dmin(pd_min, dmin(a, det));
            }
            // K = -Lambda^-1 Psi = (-adj Psi) / det: the product runs while the reciprocal is refined
            const double Kraw = mm(adjk, Psi, 0.0);
            const double rdet = frcp(det);
            const double Kk = Kraw * rdet, mL = adjm * rdet, mLi = adjk * rdet;
            // right-hand side of the predictor for this block (off the chain of the matrix recursion)
            const double PV = HZ + PC;
            const double KFF = mm(mLi, mm(Br, PV, HU), 0.0);                // -Lambda^-1 (hu + B'(hz + p+)), in every column
            q_st[-d] = st_k ? Kk : (st_l ? mL : KFF);
            if (NS >= 4) q_sf[-d] = KFF;
            // closed-loop (Joseph) form: Q + Acl' P Acl + K' R~ K (+ M K + K' M'); wasted (and harmless) for block 0
            const double Acl = mm(Btr, Kk, Ar), RK_ = mm(Rs, Kk, 0.0), Tm = mm(BtP, Kk, PA);      // A + B K,  R~ K,  P Acl = PA + PB K
            double Pn = mm(Acl, Tm, 0.0) + mm(Kk, RK_, Qr);
            if (HASM) { const double MK = mm(Mtr, Kk, 0.0); Pn = mm(MK, Ir, Pn + MK); }
            Pm = Pn;      // symmetric up to rounding; the recursion does not amplify the difference
            PC = mm(Acl, PV, mm(Kk, HU, 0.0));      // Acl'(hz + p+) + K' hu
        };
        const int G = N / PD, rem = N - G * PD;
        for (int g = 0; g < G; g++) {       // full groups: the reads of the next group (or of the remainder) are in flight
            MPC_UNROLL for (int d = 0; d < PD; d++) block(d, true);
            q_sz -= PD; q_su -= PD; q_s0 -= PD; q_s1 -= PD; q_hu -= PD; q_hz -= PD; q_st -= stp; q_sf -= stpf;
        }
        MPC_UNROLL for (int d = 0; d < PD - 1; d++) { if (d < rem) block(d, false); }
    };
    // backward: right-hand-side recursion of the corrector: h (RG rows), K, -Lambda^-1 -> kff
    auto tile_rhs = [&]() {
        const PT &Pl = launder(P);
        const TileCtx c = tile_ctx(Pl);
        const int tr = c.tr, tc = c.tc; const bool live = c.live;
        const double Ar = c.Ar, Br = c.Br, Btr = c.Btr; double *const trash = c.trash, *const p_hu = c.p_hu, *const p_hz = c.p_hz, *const p_kf = c.p_kf, *const p_k = c.p_k, *const p_li = c.p_li;
        const bool st_any = live && tr < NU && tc == 0;
        const double *q_hu = p_hu + (N - 1), *q_hz = p_hz + (N - 1), *q_k = p_k + (N - 1), *q_li = p_li + (N - 1);
        double *q_st = st_any ? p_kf + (N - 1) : trash + PD;
        const int stp = st_any ? PD : 0;
        double PC = 0.0;
        double f[PD][4];
        MPC_UNROLL for (int d = 0; d < PD; d++) { f[d][0] = q_hu[-d]; f[d][1] = q_hz[-d]; f[d][2] = q_k[-d]; f[d][3] = q_li[-d]; }
        auto block = [&](int d, bool more) {
            const double HU = f[d][0], HZ = f[d][1], Kk = f[d][2], mLi = f[d][3];
            if (more) { f[d][0] = q_hu[-PD - d]; f[d][1] = q_hz[-PD - d]; f[d][2] = q_k[-PD - d]; f[d][3] = q_li[-PD - d]; }
            q_st[-d] = mm(mLi, mm(Br, HZ + PC, HU), 0.0);
            const double Acl = mm(Btr, Kk, Ar);
            PC = mm(Acl, PC, mm(Acl, HZ, mm(Kk, HU, 0.0)));
        };
        const int G = N / PD, rem = N - G * PD;
        for (int g = 0; g < G; g++) {
            MPC_UNROLL for (int d = 0; d < PD; d++) block(d, true);
            q_hu -= PD; q_hz -= PD; q_k -= PD; q_li -= PD; q_st -= stp;
        }
        MPC_UNROLL for (int d = 0; d < PD - 1; d++) { if (d < rem) block(d, false); }
    };
    // forward: Newton direction: K, kff -> du | dz
    auto tile_forward = [&]() {
        const PT &Pl = launder(P);
        const TileCtx c = tile_ctx(Pl);
        const int tr = c.tr, tc = c.tc; const bool live = c.live;
        const double Atr = c.Atr, Btr = c.Btr; double *const trash = c.trash, *const p_kf = c.p_kf, *const p_k = c.p_k, *const p_kt = c.p_kt;
        auto trow = [&](int row) -> double * { return c.Tt + (live ? row : RZ) * (NI * LD); };
        const bool st_u = tr < NU && tc == 0, st_z = tr < NS && tc == 1;
        const double *q_k = p_k, *q_kt = p_kt, *q_kf = p_kf;
        const bool st_any = live && (st_u || st_z);
        double *q_st = st_any ? trow(st_u ? RG + tr : RG + NU + (tr < NS ? tr : 0)) : trash;
        const int stp = st_any ? PD : 0;
        double DZ = 0.0;
        double f[PD][3];
        MPC_UNROLL for (int d = 0; d < PD; d++) { f[d][0] = q_k[d]; f[d][1] = q_kt[d]; f[d][2] = q_kf[d]; }
        auto block = [&](int d, bool more) {
            const double Kk = f[d][0], KkT = f[d][1], KFF = f[d][2];
            if (more) { f[d][0] = q_k[PD + d]; f[d][1] = q_kt[PD + d]; f[d][2] = q_kf[PD + d]; }
            const double DU = mm(KkT, DZ, KFF);                       // K dz + kff
            const double AclT = mm(Kk, Btr, Atr);                     // (A + B K)'
            const double DZn = mm(AclT, DZ, mm(Btr, KFF, 0.0));       // Acl dz + B kff
            q_st[d] = st_u ? DU : DZn;
            DZ = DZn;
        };
        const int G = N / PD, rem = N - G * PD;
        for (int g = 0; g < G; g++) {
            MPC_UNROLL for (int d = 0; d < PD; d++) block(d, true);
            q_k += PD; q_kt += PD; q_kf += PD; q_st += stp;
        }
        MPC_UNROLL for (int d = 0; d < PD - 1; d++) { if (d < rem) block(d, false); }
    };

    double dvp[NI][NC];      // predictor direction of each instance's bounded variables, kept for the corrector's second-order terms
    for (int it = 0;; it++) {
        bool any_on = false;
        MPC_UNROLL for (int j = 0; j < NI; j++) any_on = any_on || S[j].on;
        if (!any_on) break;      // wave-uniform: every instance has its verdict
        __syncthreads();
        pd_min = 1.0;
        tile_factor();
        __syncthreads();
        MPC_TSTAMP(2);
        tile_forward();
        __syncthreads();
        MPC_TSTAMP(3);
        {
            const unsigned long long bad = __ballot(((lane >> 2) & 3) < NI && !(pd_min > 0.0));      // a Lambda lost definiteness: the instance stops as infeasible
            MPC_UNROLL for (int j = 0; j < NI; j++) {
                if (S[j].on && (bad & (0x000F000F000F000FULL << (4 * j)))) { S[j].on = false; S[j].status = kInfeasible; S[j].iters = it; }
            }
        }
        // ================= element-wise: predictor step length, centring, corrector rhs -> LDS ===========================
        MPC_UNROLL for (int j = 0; j < NI; j++) {
            WvInst &Sj = S[j];
            if (Sj.on) {
                const Iter &Xj = X[j];
                const PT &Pl = P;
                Bnd Bd; bounds(Pl, j, Bd);
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
                gradient(Pl, j, Xj, gu, gz);
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
        __syncthreads();
        MPC_TSTAMP(4);
        tile_rhs();
        __syncthreads();
        MPC_TSTAMP(5);
        tile_forward();
        __syncthreads();
        MPC_TSTAMP(3);
        // ================= element-wise: corrector step length, step; then the next iterate's residuals / gradients =====
        MPC_UNROLL for (int j = 0; j < NI; j++) {
            WvInst &Sj = S[j];
            if (Sj.on) {
                Iter &Xj = X[j];
                const PT &Pl = P;
                Bnd Bd; bounds(Pl, j, Bd);
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
                phase_a(Pl, j, Sj, Xj, it + 1);
            }
        }
        MPC_TSTAMP(6);
    }
}

}  // namespace mpc
