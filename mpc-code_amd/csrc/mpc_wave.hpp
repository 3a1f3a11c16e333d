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
    // row stride: the horizon + 1, made odd: the tile view (row, instance) then hits distinct banks; 65 for the longest horizon
    __host__ __device__ static constexpr int ld(int N) { return (N + 1) | 1; }
    static constexpr int GUARD = 8;                                    // cells in front of T: look-ahead reads of the backward loops, stores of idle lanes
    __host__ __device__ static constexpr int t_doubles(int N) { return GUARD + ROWS * NI * ld(N); }
    // time-varying stage data (wv_solve<.., LTV>: one SQP iteration of the non-linear path): per (instance, block) the matrices
    // A_k (NS x NS), B_k (NS x NU) and the affine term c_k as rows behind the others (35 rows for ns = 3, nu = 2: four workgroups per CU at N = 30)
    static constexpr int RL_A = ROWS, RL_B = RL_A + NS * NS, RL_C = RL_B + NS * NU, ROWS_LTV = RL_C + NS;
    __host__ __device__ static constexpr int t_doubles_ltv(int N) { return GUARD + ROWS_LTV * NI * ld(N); }
    static constexpr int QZN = 5 * NS + 2 * NU + 1;                    // reference of the terminal cost: zr, or (terminal equality) zr aimed off by the measured miss
    static constexpr int QN = QZN + NS;                                // z0 zr c zlo zhi | ur us | ws_delta | zrN
    static constexpr int ROWS_WS = NU + 2 * NC;                        // warm start kept in HBM between launches: u | l_lo | l_hi
    static constexpr int OUT = NU + NS + 1;                            // first input / next state of the final iterate, terminal miss
    __host__ __device__ static constexpr size_t lds_doubles(int keep_per_inst, int N) { return (size_t)t_doubles(N) + NI * QN + NI * OUT + NI * keep_per_inst; }
};

template <int NS, int NU, int NC>
struct WvIter { double sl[NC], sh[NC], ll[NC], lh[NC], u[NU], z[NS]; };      // one block of one instance (lane = block)

// A double parked in two accumulation registers (AGPRs).  The resident iterates of a wave are 240 registers that are touched three
// times per interior-point iteration; left to the register allocator about a third of them ends up in scratch (a memory round trip
// per touch).  Parked explicitly they occupy the AGPR file, which nothing else here needs (the matrix-core results stay in VGPRs),
// and moving one instance's block in or out is 50 + 50 one-cycle-issue moves per phase.
struct AReg2 { int lo, hi; };
__device__ __forceinline__ void a_put(AReg2 &r, double v)
{
    union { double d; int i[2]; } u; u.d = v;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(r.lo) : "v"(u.i[0]));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(r.hi) : "v"(u.i[1]));
}
__device__ __forceinline__ double a_get(const AReg2 &r)
{
    union { double d; int i[2]; } u;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(u.i[0]) : "a"(r.lo));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(u.i[1]) : "a"(r.hi));
    return u.d;
}
template <int NS, int NU, int NC>
struct WvIterA {
    AReg2 sl[NC], sh[NC], ll[NC], lh[NC], u[NU], z[NS];
    __device__ __forceinline__ void get(WvIter<NS, NU, NC> &X) const
    {
        MPC_UNROLL for (int i = 0; i < NC; i++) { X.sl[i] = a_get(sl[i]); X.sh[i] = a_get(sh[i]); X.ll[i] = a_get(ll[i]); X.lh[i] = a_get(lh[i]); }
        MPC_UNROLL for (int i = 0; i < NU; i++) X.u[i] = a_get(u[i]);
        MPC_UNROLL for (int i = 0; i < NS; i++) X.z[i] = a_get(z[i]);
    }
    __device__ __forceinline__ void put(const WvIter<NS, NU, NC> &X)
    {
        MPC_UNROLL for (int i = 0; i < NC; i++) { a_put(sl[i], X.sl[i]); a_put(sh[i], X.sh[i]); a_put(ll[i], X.ll[i]); a_put(lh[i], X.lh[i]); }
        MPC_UNROLL for (int i = 0; i < NU; i++) a_put(u[i], X.u[i]);
        MPC_UNROLL for (int i = 0; i < NS; i++) a_put(z[i], X.z[i]);
    }
};

struct WvInst { double mu, mu_sum, sm, inv_ncon, gscale, res_s, res_p; int stall, iters, status; bool on, warm, keep_u; };

enum : int { kWvOk0 = 1, kWvWarm = 2, kWvValid = 4, kWvKeepU = 8, kWvNoShift = 16 };      // NoShift: warm start from the iterate as it is (an SQP iteration of the same step)      // KeepU: warm start whose inputs are a caller's guess for THIS problem (not to be shifted)

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
// LTV: the stage matrices and affine terms come per (instance, block) from the rows RL_A / RL_B / RL_C of T (written by the caller:
// the QP of one SQP iteration, x+ = A_k x + B_k u + c_k) instead of P.A / P.B / q's c.  The recursions that are scans with powers of
// a constant A (initial simulation, costates of the stationarity test) then run as sequential sweeps over the lanes.
// PAIR (horizons up to 32): the element-wise phases take two instances at once - lanes 0-31 the blocks of instance 2 j, lanes 32-63
// those of instance 2 j + 1 - so X and S have NI / 2 entries, every "per instance" quantity of an entry is per lane (the same in the
// lanes of a half: reductions run per half, branches on an instance's flags diverge by whole halves), and the caller reads verdicts
// at a lane of the half.  The tile passes are the same; the sums are taken in the same tree order.
template <int NS, int NU, bool HASM, int NC, bool MASKED, int NI, class PT, bool LTV = false, bool PAIR = false>
__device__ __forceinline__ void wv_solve(const PT &P, double *T, const double *q, const int *iflag,
                                         WvIterA<NS, NU, NC> (&X)[PAIR ? NI / 2 : NI], WvInst (&S)[PAIR ? NI / 2 : NI], int max_iter)
{
    static_assert(!LTV || (NS <= 4 && !HASM), "time-varying stage matrices: one 4 x 4 tile per state matrix, no cross term");
    using Cfg = WvCfg<NS, NU, NC, NI>;
    using Iter = WvIter<NS, NU, NC>;
    constexpr int NV = Cfg::NV, NKF = Cfg::NKF, NLI = Cfg::NLI, RA = Cfg::RA, RG = Cfg::RG, RK = Cfg::RK;
    static_assert(NS <= 8 && NU <= 2 && NI >= 1 && NI <= 4, "stage state <= 8 (2 x 2 tiles of 4 x 4), nu <= 2 (closed-form inverse of Lambda); at most four instances per product");
    static_assert(!PAIR || NI % 2 == 0, "pairs of instances");
    constexpr int NJ = PAIR ? NI / 2 : NI;      // register sets: instances, or pairs of instances
    const int lane = threadIdx.x, N = P.N, LD = Cfg::ld(N);
    const int k = PAIR ? (lane & 31) : lane;
    const int hsel = PAIR ? (lane >> 5) : 0;
    auto ji = [&](int j) -> int { return PAIR ? 2 * j + hsel : j; };      // the instance this lane works on in register set j
    const bool blk_on = k < N, last = k == N - 1;
    auto tk = [&](int row, int j) -> double & { return T[(row * NI + ji(j)) * LD + k]; };      // lane = block view
    auto un = [](double v) -> double { return PAIR ? v : uni(v); };
    auto rsum = [](double v) -> double { return PAIR ? half_sum(v) : wave_sum(v); };
    auto rmax = [](double v) -> double { return PAIR ? half_max(v) : wave_max(v); };
    MPC_STAMP_INIT

    // the problem pointer made opaque: scalar loads of the constants stay inside the phase that asks for them (hoisted out of the
    // iteration loop they occupy - and spill - scalar registers for the whole solve)
    auto launder = [](const PT &Pin) -> const PT & { const PT *pp = &Pin; asm volatile("" : "+s"(pp)); return *pp; };
    struct Bnd { double lo[NC], hi[NC]; bool fl[NC], fh[NC]; };
    auto bounds = [&](const PT &Pl, int j, Bnd &Bd) {
        const double *qd = q + ji(j) * Cfg::QN;
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
        const double *qd = q + ji(j) * Cfg::QN;
        double dz1[NS], du[NU];
        const double *qz = qd + (last ? Cfg::QZN : NS);      // (the last block's cost is the terminal one, with a reference of its own)
        MPC_UNROLL for (int i = 0; i < NS; i++) dz1[i] = Xj.z[i] - qz[i];
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
            MPC_UNROLL for (int i = 0; i < NU; i++) un[i] = wave_dn1(du[i], du[i]);      // (neighbour lanes on DPP wave shifts: no LDS round trip)
            MPC_UNROLL for (int i = 0; i < NS; i++) { const double t = wave_up1(dz1[i], dz1[i]); zp[i] = k > 0 ? t : qd[i] - qd[NS + i]; }
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
            if (blk_on) tk(RA + i, j) = Xj.ll[i] * isl + Xj.lh[i] * ish;      // lanes beyond the horizon would land in the next row (stride N + 1)
            hb[i] = Xj.lh[i] * (rh * ish - 1.0) + Xj.ll[i] * (rl * isl + 1.0);
            resp_p = dmax(resp_p, dmax(fabs(rl), fabs(rh)));
            cres_p = dmax(cres_p, dmax(comp_measure(Xj.sl[i], Xj.ll[i]), comp_measure(Xj.sh[i], Xj.lh[i])));
            lmax_p = dmax(lmax_p, dmax(Xj.ll[i], Xj.lh[i]));
        }
        double gu[NU], gz[NS], pi[NS];
        gradient(Pl, j, Xj, gu, gz);
        if (blk_on) {
            MPC_UNROLL for (int i = 0; i < NU; i++) tk(RG + i, j) = gu[i] + hb[i];
            MPC_UNROLL for (int i = 0; i < NS; i++) tk(RG + NU + i, j) = gz[i] + (NU + i < NC ? hb[NU + i < NC ? NU + i : 0] : 0.0);
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) pi[i] = blk_on ? gz[i] : 0.0;
        Sj.mu_sum = rsum(blk_on ? mu_p : 0.0);
        const double res_p = rmax(blk_on ? resp_p : 0.0), cres = rmax(blk_on ? cres_p : 0.0), lmax = rmax(blk_on ? lmax_p : 0.0);
        const bool ok_cp = (cres <= 1.0) && (res_p <= kTolFeas);
        double rs_p = 0.0;
        if (LTV) {
            // pi_{k+1} = gz_{k+1} + A_{k+1}' pi_{k+2} (lane k holds block k = (u_k, z_{k+1}): its costate passes through the NEXT block's A):
            // a sweep from the last lane down.  Every lane below the last block recomputes from its successor in every round - lanes whose
            // successor is final only reproduce their value - so the rounds carry no lane mask.  The costates serve the stationarity
            // test alone: they are taken when it can decide (complementarity and feasibility in tolerance) and for the scale at it = 0.
            if (it == 0 || ok_cp) {
                double An[NS][NS], g0[NS];
                const int kn = k + 1 < N ? k + 1 : (N > 0 ? N - 1 : 0);
                const bool upd = k + 1 < N;      // (the other lanes: A = 0, their value stays gz resp. 0)
                MPC_UNROLL for (int i = 0; i < NS; i++) { g0[i] = pi[i]; MPC_UNROLL for (int l = 0; l < NS; l++) { const double v = T[((Cfg::RL_A + i * NS + l) * NI + ji(j)) * LD + kn]; An[i][l] = upd ? v : 0.0; } }
                for (int s = N - 2; s >= 0; s--) {
                    double t[NS];
                    // (lane 63 reads itself: its A is 0.  PAIR: the lanes without a successor block sit next to the other instance's
                    // lanes - what arrives there is dropped, not multiplied by zero: a NaN of a diverged neighbour must not cross over)
                    MPC_UNROLL for (int i = 0; i < NS; i++) { const double v = wave_dn1(pi[i], pi[i]); t[i] = (PAIR && !upd) ? 0.0 : v; }
                    MPC_UNROLL for (int i = 0; i < NS; i++) { double a = g0[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += An[l][i] * t[l]; pi[i] = a; }
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) {
                    double a = gu[i];
                    MPC_UNROLL for (int l = 0; l < NS; l++) a += (blk_on ? tk(Cfg::RL_B + l * NU + i, j) : 0.0) * pi[l];
                    rs_p = dmax(rs_p, fabs(a));
                }
            }
        } else if (it == 0 || ok_cp) {      // (the costates serve the stationarity test alone: taken when it can decide, and for the scale at it = 0)
            MPC_UNROLL for (int e = 0; e < 6; e++) {
                const int d = 1 << e;
                if (d < N) {
                    double t[NS];
                    MPC_UNROLL for (int i = 0; i < NS; i++) { const double v = e == 0 ? wave_dn1(pi[i], pi[i]) : __shfl_down(pi[i], d, 64); t[i] = (k + d < N) ? v : 0.0; }
                    MPC_UNROLL for (int i = 0; i < NS; i++) { double a = pi[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pl.Apow[e][l][i] * t[l]; pi[i] = a; }
                }
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) { double a = gu[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pl.B[l][i] * pi[l]; rs_p = dmax(rs_p, fabs(a)); }
        }
        double res_s = Sj.res_s;
        if (it == 0 || ok_cp) res_s = rmax(blk_on ? rs_p : 0.0);
        Sj.mu = Sj.mu_sum * Sj.inv_ncon; Sj.res_s = res_s; Sj.res_p = res_p;
        if (it == 0) Sj.gscale = dmax(1.0, P.term_cons ? dmin(res_s, P.term_gcap) : res_s);      // mpc_device.hpp:rpdip_lane
        Sj.stall = ok_cp ? Sj.stall + 1 : 0;
        int verdict = -1;
        if (ok_cp && (res_s <= kTolStat * Sj.gscale + P.term_floor || (Sj.stall > kStallMax && res_s <= kTolStatAcc * Sj.gscale + P.term_floor))) verdict = kSolved;
        else if (lmax > kInfeasZ * Sj.gscale || !(fabs(Sj.mu) < 1.0e300)) verdict = kInfeasible;
        else if (it == max_iter) verdict = kMaxIter;
        if (verdict >= 0) { Sj.on = false; Sj.status = verdict; Sj.iters = it; }
    };

    // ---- instance constants and the initial point (cold: us pushed inside the box; warm: previous iterate shifted one stage)
    MPC_UNROLL for (int j = 0; j < NJ; j++) {
        WvInst &Sj = S[j];
        Iter Xj;
        const PT &Pl = P;
        const double *qd = q + ji(j) * Cfg::QN;
        const int myflag = PAIR ? iflag[ji(j)] : __builtin_amdgcn_readfirstlane(iflag[j]);
        Sj.on = (myflag & kWvValid) && (myflag & kWvOk0);
        Sj.warm = (myflag & kWvWarm) != 0;
        Sj.mu = 0.0; Sj.mu_sum = 0.0; Sj.sm = 0.0; Sj.gscale = 1.0; Sj.stall = 0; Sj.iters = 0; Sj.res_s = 0.0; Sj.res_p = 0.0;
        Sj.keep_u = (myflag & kWvKeepU) != 0;
        Sj.status = ((myflag & kWvValid) && !(myflag & kWvOk0)) ? kInfeasible : kMaxIter;
        double ncon = 0.0;
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            const double zlm = i >= NU ? un(qd[3 * NS + (i >= NU ? i - NU : 0)]) : 0.0, zhm = i >= NU ? un(qd[4 * NS + (i >= NU ? i - NU : 0)]) : 0.0;
            const double lm = i < NU ? Pl.ulo[i < NU ? i : 0] : zlm, hm = i < NU ? Pl.uhi[i < NU ? i : 0] : zhm;
            const double le = i < NU ? Pl.ulo[i < NU ? i : 0] : Pl.zlo_e[i >= NU ? i - NU : 0], he = i < NU ? Pl.uhi[i < NU ? i : 0] : Pl.zhi_e[i >= NU ? i - NU : 0];
            const bool flm = MASKED ? fin(lm) : true, fhm = MASKED ? fin(hm) : true, fle = MASKED ? fin(le) : true, fhe = MASKED ? fin(he) : true;
            ncon += (double)(N - 1) * ((flm ? 1 : 0) + (fhm ? 1 : 0)) + (fle ? 1 : 0) + (fhe ? 1 : 0);
        }
        Sj.inv_ncon = 1.0 / dmax(ncon, 1.0);
        if (Sj.on) {
            if (Sj.warm) {      // the previous step's inputs and multipliers
                MPC_UNROLL for (int i = 0; i < NU; i++) Xj.u[i] = a_get(X[j].u[i]);
                MPC_UNROLL for (int i = 0; i < NC; i++) { Xj.ll[i] = a_get(X[j].ll[i]); Xj.lh[i] = a_get(X[j].lh[i]); }
            }
            const bool rep = k >= N - 1 || (myflag & kWvNoShift) != 0;       // shift by one stage, the last block repeats (NoShift: every block stays)
            double ll0[NC], lh0[NC];
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                const double ulo = Pl.ulo[i], uhi = Pl.uhi[i];
                const bool f_lo = fin(ulo), f_hi = fin(uhi);
                double v;
                if (Sj.warm) {
                    const double t = wave_dn1(Xj.u[i], Xj.u[i]);
                    v = (rep || Sj.keep_u) ? Xj.u[i] : t;
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
                    const double tl = wave_dn1(Xj.ll[i], Xj.ll[i]), th = wave_dn1(Xj.lh[i], Xj.lh[i]);
                    ll0[i] = rep ? Xj.ll[i] : tl; lh0[i] = rep ? Xj.lh[i] : th;
                } else { ll0[i] = 0.0; lh0[i] = 0.0; }
            }
            // states of the initial point by forward simulation z_{k+1} = A z_k + B u_k + c: a linear recursion with a constant
            // matrix, taken as a scan over the lanes with A^(2^e) (lane k ends up with z_{k+1})
            if (LTV) {      // z_{k+1} = A_k z_k + B_k u_k + c_k with this lane's own matrices: a sweep from lane 0 up
                double Ak[NS][NS], bk[NS], xk[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    double a = blk_on ? tk(Cfg::RL_C + i, j) : 0.0;
                    MPC_UNROLL for (int l = 0; l < NU; l++) a += (blk_on ? tk(Cfg::RL_B + i * NU + l, j) : 0.0) * Xj.u[l];
                    bk[i] = a;
                    MPC_UNROLL for (int l = 0; l < NS; l++) Ak[i][l] = blk_on ? tk(Cfg::RL_A + i * NS + l, j) : 0.0;
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = bk[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += Ak[i][l] * qd[l]; xk[i] = k == 0 ? a : 0.0; }
                // every lane above 0 recomputes from its predecessor in every round (no per-round mask); lane 0 and the lanes beyond the
                // horizon have A = 0 and keep their value
                double x00[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) x00[i] = xk[i];
                if (k == 0 || !blk_on) { MPC_UNROLL for (int i = 0; i < NS; i++) { bk[i] = x00[i]; MPC_UNROLL for (int l = 0; l < NS; l++) Ak[i][l] = 0.0; } }
                for (int s = 1; s < N; s++) {
                    double t[NS];
                    MPC_UNROLL for (int i = 0; i < NS; i++) { const double v = wave_up1(xk[i], xk[i]); t[i] = (PAIR && (k == 0 || !blk_on)) ? 0.0 : v; }      // (PAIR: as in the costate sweep)
                    MPC_UNROLL for (int i = 0; i < NS; i++) { double a = bk[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += Ak[i][l] * t[l]; xk[i] = a; }
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) Xj.z[i] = xk[i];
            } else {
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
                        MPC_UNROLL for (int i = 0; i < NS; i++) { const double v = e == 0 ? wave_up1(xk[i], xk[i]) : __shfl_up(xk[i], d, 64); t[i] = k >= d ? v : 0.0; }
                        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = xk[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pl.Apow[e][i][l] * t[l]; xk[i] = a; }
                    }
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) Xj.z[i] = xk[i];
            }
            Bnd Bd; bounds(Pl, j, Bd);
            const double ws_delta = un(qd[5 * NS + 2 * NU]);
            const double ws_smin = dmin(dmax(kWsKappa * ws_delta, kWsSMinLo), kWsSMinHi), ws_mu = kWsMuFactor * ws_smin * ws_smin;
            const double smin = Sj.warm ? ws_smin : kSMin;
            MPC_UNROLL for (int i = 0; i < NC; i++) {
                const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                Xj.sl[i] = Bd.fl[i] ? dmax(v - Bd.lo[i], smin) : 1.0; Xj.sh[i] = Bd.fh[i] ? dmax(Bd.hi[i] - v, smin) : 1.0;
                const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                const double llo = Sj.warm ? dmax(ll0[i], ws_mu * isl) : kMu0 * isl, lhi = Sj.warm ? dmax(lh0[i], ws_mu * ish) : kMu0 * ish;
                Xj.ll[i] = Bd.fl[i] ? llo : 0.0; Xj.lh[i] = Bd.fh[i] ? lhi : 0.0;
            }
            X[j].put(Xj);
            phase_a(Pl, j, Sj, Xj, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    MPC_TSTAMP(1);
    // ---- tile view ------------------------------------------------------------------------------------------------------
    // lane 16 r + 4 b + c holds element (r, c) of a 4x4 tile of instance b; mm(M, S, C) = M'S + C on all four instances' tiles at
    // once.  A stage with more than four states is a grid of SB x SB such tiles (SB = 2 up to stage state 8: Wood-Berry's Delta-u
    // form, general output rows); the input block stays one tile (nu <= 2: the 2 x 2 inverse of Lambda is closed-form).
    // One wave alone on a SIMD issues one instruction every four cycles whatever its kind, so these loops are written for the
    // lowest instruction count per block:
    //   * a lane that has no use for a quantity reads the zero row RZ instead (never written after the kernel's prologue): no masks;
    //   * a lane that has nothing to store stores to the guard cells in front of T: no exec masking;
    //   * the loops run in groups of PD blocks with the LDS reads of the next group in flight (the last group's look-ahead stays
    //     inside the LDS allocation - guard cells in front, the instance data behind - and is dropped), then the N mod PD blocks left;
    //   * per field one address register stepped once per group, the blocks of a group at immediate offsets.
    auto mm = [](double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); };
    constexpr int RZ = Cfg::RZ;
    constexpr int SB = (NS + 3) / 4;      // tiles per side of a state matrix
    // Everything a pass needs per lane is rebuilt at its start from a lane index the compiler cannot see through: as loop
    // invariants of the iteration loop these registers would stay live across the element-wise phases (and spill there).
    struct TileCtx {
        int tr, tc; bool live;
        double *Tt, *trash;
        double Ar[SB][SB], Atr[SB][SB], Br[SB], Btr[SB];      // A, A', B, B' as tiles: Ar[i][j](r, c) = A[4i + r][4j + c], Btr[j](r, c) = B[4j + c][r]
        double *p_hu, *p_hz[SB], *p_kf, *p_k[SB], *p_kt[SB], *p_li;
        const double *p_a, *p_at, *p_b, *p_bt;      // LTV: this lane's element of A_k, A_k', B_k, B_k' as a function of the block (rows of T)
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
        c.Tt = T + tb * LD; c.trash = T - Cfg::GUARD;
        auto trow = [&](int row) -> double * { return c.Tt + (c.live ? row : RZ) * (NI * LD); };
        MPC_UNROLL for (int i = 0; i < SB; i++) {
            const int ri = 4 * i + tr, ci = 4 * i + tc;
            MPC_UNROLL for (int j = 0; j < SB; j++) {
                const int cj = 4 * j + tc, rj = 4 * j + tr;
                c.Ar[i][j] = (ri < NS && cj < NS) ? Pl.A[ri < NS ? ri : 0][cj < NS ? cj : 0] : 0.0;
                c.Atr[i][j] = (ri < NS && cj < NS) ? Pl.A[cj < NS ? cj : 0][ri < NS ? ri : 0] : 0.0;
                (void)rj;
            }
            c.Br[i] = (ri < NS && tc < NU) ? Pl.B[ri < NS ? ri : 0][tc < NU ? tc : 0] : 0.0;
            c.Btr[i] = (tr < NU && ci < NS) ? Pl.B[ci < NS ? ci : 0][tr < NU ? tr : 0] : 0.0;
            // vectors live in column-replicated tiles: hz rows of this lane; K (rows < NU, state columns) and K' as tiles
            c.p_hz[i] = trow(ri < NS ? RG + NU + ri : RZ);
            c.p_k[i] = trow((tr < NU && ci < NS) ? RA + tr * NS + ci : RZ);
            c.p_kt[i] = trow((ri < NS && tc < NU) ? RA + tc * NS + ri : RZ);
        }
        c.p_hu = trow(tr < NU ? RG + tr : RZ); c.p_kf = trow(tr < NU ? RK + tr : RZ);
        if (LTV) {
            c.p_a = trow((tr < NS && tc < NS) ? Cfg::RL_A + tr * NS + tc : RZ); c.p_at = trow((tr < NS && tc < NS) ? Cfg::RL_A + tc * NS + tr : RZ);
            c.p_b = trow((tr < NS && tc < NU) ? Cfg::RL_B + tr * NU + tc : RZ); c.p_bt = trow((tr < NU && tc < NS) ? Cfg::RL_B + tc * NU + tr : RZ);
        }
        // -Lambda^-1 (symmetric, rows / columns < NU) as a tile
        const int li_i = tr > tc ? tr : tc, li_j = tr > tc ? tc : tr;
        c.p_li = trow((tr < NU && tc < NU) ? RA + NKF + li_i * (li_i + 1) / 2 + li_j : RZ);
        return c;
    };
    double pd_min = 1.0;        // smallest pivot / determinant of this lane's instance so far: <= 0 means a Lambda lost definiteness
    constexpr int PD = SB == 1 ? 4 : 2;       // blocks per group

    // backward: Riccati factorisation (sigma -> K, -Lambda^-1) with the right-hand-side recursion of the predictor behind it
    auto tile_factor = [&]() {
        const PT &Pl = launder(P);
        const TileCtx c = tile_ctx(Pl);
        const int tr = c.tr, tc = c.tc; const bool live = c.live;
        double *const trash = c.trash;
        auto trow = [&](int row) -> double * { return c.Tt + (live ? row : RZ) * (NI * LD); };
        const bool in_uu = tr < NU && tc < NU;
        const double Rr = in_uu ? Pl.R[tr][tc] : 0.0;
        double Qr[SB][SB], Mtr[SB], BE0[SB], BE1[SB];
        MPC_UNROLL for (int i = 0; i < SB; i++) {
            const int ri = 4 * i + tr, ci = 4 * i + tc;
            MPC_UNROLL for (int j = 0; j < SB; j++) { const int cj = 4 * j + tc; Qr[i][j] = (ri < NS && cj < NS) ? Pl.Q[ri < NS ? ri : 0][cj < NS ? cj : 0] : 0.0; }
            Mtr[i] = (HASM && tr < NU && ci < NS) ? Pl.M[ci < NS ? ci : 0][tr < NU ? tr : 0] : 0.0;
            // rows 0 / 1 of Lambda = R~ + B'PB broadcast over all tile rows, straight from PB: (B E_i)' PB + E_i' R~
            BE0[i] = ri < NS ? Pl.B[ri < NS ? ri : 0][0] : 0.0; BE1[i] = (NU > 1 && ri < NS) ? Pl.B[ri < NS ? ri : 0][NU > 1 ? 1 : 0] : 0.0;
        }
        const double RE0 = tc < NU ? Pl.R[0][tc] : 0.0, RE1 = (NU > 1 && tc < NU) ? Pl.R[NU > 1 ? 1 : 0][tc] : 0.0;
        const double Ir = tr == tc ? 1.0 : 0.0;
        // barrier weights: sigma_z on the diagonal of P's diagonal tiles, sigma_u[r] on the diagonal of R~, sigma_u[i] at (., i) of broadcast row i
        const double *q_sz[SB];
        MPC_UNROLL for (int i = 0; i < SB; i++) { const int ri = 4 * i + tr; q_sz[i] = trow((tr == tc && ri < NS && NU + ri < NC) ? RA + NU + ri : RZ) + (N - 1); }
        const double *q_su = trow((tr == tc && tr < NU) ? RA + tr : RZ) + (N - 1);
        const double *q_s0 = trow(tc == 0 ? RA : RZ) + (N - 1), *q_s1 = trow((NU > 1 && tc == 1) ? RA + 1 : RZ) + (N - 1);
        const double *q_hu = c.p_hu + (N - 1);
        const double *q_hz[SB];
        MPC_UNROLL for (int i = 0; i < SB; i++) q_hz[i] = c.p_hz[i] + (N - 1);
        // this lane's element of -adj(Lambda) as a combination of a = L00, d = L11, off = L01 (2 x 2), element (r mod 2, c):
        //   (0,0): -d   (1,1): -a   (0,1), (1,0): +off;  -Lambda^-1 = that / det.  NU = 1: -1 in column 0.
        const int ri2 = tr & 1;
        const double c_a = (NU > 1 && tc == 1 && ri2 == 1) ? -1.0 : 0.0, c_d = (NU > 1 && tc == 0 && ri2 == 0) ? -1.0 : 0.0;
        const double c_o = (NU > 1 && tc < 2 && ri2 != tc) ? 1.0 : 0.0, c_1 = (NU == 1 && tc == 0) ? -1.0 : 0.0;
        const double w_top = tr < NU ? 1.0 : 0.0;
        const double k_a = c_a * w_top, k_d = c_d * w_top, k_o = c_o * w_top, k_1 = c_1 * w_top;      // the same, rows < NU only (the K product)
        // stores per block: K from the lanes that hold it (rows < NU of each state tile); -Lambda^-1 from rows 2..3 of tile 0, which
        // compute it as well; kff (valid in rows < NU of every column) from column 3 of tile 0 when the state leaves it free - else its own store
        const bool st_l = tr >= 2 && tc < NU && tc <= tr - 2 && tr - 2 < NU, st_f3 = NS < 4 && tr < NU && tc == 3;
        double *q_st[SB]; int stp[SB];
        MPC_UNROLL for (int j = 0; j < SB; j++) {
            const int cj = 4 * j + tc;
            const bool st_k = tr < NU && cj < NS;
            const bool any = live && (st_k || (j == 0 && (st_l || st_f3)));
            q_st[j] = any ? trow(st_k ? RA + tr * NS + cj : (st_l ? RA + NKF + (tr - 2) * (tr - 1) / 2 + tc : RK + (tr < NU ? tr : 0))) + (N - 1) : trash + PD;
            stp[j] = any ? PD : 0;
        }
        const bool sel_k0 = tr < NU && tc < NS;      // tile 0 lanes that store K (the others of tile 0 store -Lambda^-1 or kff)
        const bool st_f = live && NS >= 4 && tr < NU && tc == 0;
        double *q_sf = st_f ? c.p_kf + (N - 1) : trash + PD;
        const int stpf = st_f ? PD : 0;
        double Pm[SB][SB], PC[SB];
        MPC_UNROLL for (int i = 0; i < SB; i++) {
            const int ri = 4 * i + tr;
            MPC_UNROLL for (int j = 0; j < SB; j++) { const int cj = 4 * j + tc; Pm[i][j] = (ri < NS && cj < NS) ? Pl.Pf[ri < NS ? ri : 0][cj < NS ? cj : 0] : 0.0; }
            PC[i] = 0.0;
        }
        constexpr int NF = 2 * SB + 4 + (LTV ? 3 : 0);      // sigma_z[SB] sigma_u s0 s1 hu hz[SB] (a b bt)
        double f[PD][NF];
        const double *q_a = LTV ? c.p_a + (N - 1) : nullptr, *q_b = LTV ? c.p_b + (N - 1) : nullptr, *q_bt = LTV ? c.p_bt + (N - 1) : nullptr;
        auto fetch = [&](int d, int off) {
            MPC_UNROLL for (int i = 0; i < SB; i++) { f[d][i] = q_sz[i][off]; f[d][SB + 4 + i] = q_hz[i][off]; }
            f[d][SB] = q_su[off]; f[d][SB + 1] = q_s0[off]; f[d][SB + 2] = NU > 1 ? q_s1[off] : 0.0; f[d][SB + 3] = q_hu[off];
            if (LTV) { f[d][NF - 3] = q_a[off]; f[d][NF - 2] = q_b[off]; f[d][NF - 1] = q_bt[off]; }
        };
        MPC_UNROLL for (int d = 0; d < PD; d++) fetch(d, -d);
        auto block = [&](int d, bool more) {
            double sz[SB], HZ[SB];
            MPC_UNROLL for (int i = 0; i < SB; i++) { sz[i] = f[d][i]; HZ[i] = f[d][SB + 4 + i]; }
            const double su = f[d][SB], s0 = f[d][SB + 1], s1 = f[d][SB + 2], HU = f[d][SB + 3];
            // this block's matrices as tiles: the constants, or (LTV, one tile) the block's own from the table
            double Arl[SB][SB], Brl[SB], Btrl[SB], BE0l[SB], BE1l[SB];
            MPC_UNROLL for (int i = 0; i < SB; i++) { Brl[i] = c.Br[i]; Btrl[i] = c.Btr[i]; BE0l[i] = BE0[i]; BE1l[i] = BE1[i]; MPC_UNROLL for (int jj = 0; jj < SB; jj++) Arl[i][jj] = c.Ar[i][jj]; }
            if (LTV) {
                Arl[0][0] = f[d][NF - 3]; Brl[0] = f[d][NF - 2]; Btrl[0] = f[d][NF - 1];
                BE0l[0] = dpp_move<0x00, 0xF>(Brl[0], Brl[0]); BE1l[0] = NU > 1 ? dpp_move<0x55, 0xF>(Brl[0], Brl[0]) : 0.0;      // columns 0 / 1 of B over the quad
            }
            if (more) fetch(d, -PD - d);
            MPC_UNROLL for (int i = 0; i < SB; i++) Pm[i][i] += sz[i];
            // P A, P B, B'P (P symmetric: P' = P)
            double PA[SB][SB], PB[SB], BtP[SB];
            MPC_UNROLL for (int i = 0; i < SB; i++) {
                double pb = 0.0, bp = 0.0;
                MPC_UNROLL for (int l = 0; l < SB; l++) { pb = mm(Pm[l][i], Brl[l], pb); bp = mm(Brl[l], Pm[l][i], bp); }
                PB[i] = pb; BtP[i] = bp;
                MPC_UNROLL for (int j = 0; j < SB; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < SB; l++) a = mm(Pm[l][i], Arl[l][j], a); PA[i][j] = a; }
            }
            double X0 = RE0 + s0, X1 = RE1 + s1;
            MPC_UNROLL for (int l = 0; l < SB; l++) { X0 = mm(BE0l[l], PB[l], X0); if (NU > 1) X1 = mm(BE1l[l], PB[l], X1); }
            double Psi[SB];
            MPC_UNROLL for (int j = 0; j < SB; j++) { double a = Mtr[j]; MPC_UNROLL for (int l = 0; l < SB; l++) a = mm(Brl[l], PA[l][j], a); Psi[j] = a; }      // M' + B'PA
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
            // K = -Lambda^-1 Psi = (-adj Psi) / det: the products run while the reciprocal is refined
            double Kk[SB];
            MPC_UNROLL for (int j = 0; j < SB; j++) Kk[j] = mm(adjk, Psi[j], 0.0);
            const double rdet = frcp(det);
            MPC_UNROLL for (int j = 0; j < SB; j++) Kk[j] *= rdet;
            const double mL = adjm * rdet, mLi = adjk * rdet;
            // right-hand side of the predictor for this block (off the chain of the matrix recursion)
            double PV[SB], psv = HU;
            MPC_UNROLL for (int i = 0; i < SB; i++) { PV[i] = HZ[i] + PC[i]; psv = mm(Brl[i], PV[i], psv); }      // hu + B'(hz + p+)
            const double KFF = mm(mLi, psv, 0.0);                           // -Lambda^-1 psi, in every column
            q_st[0][-d] = sel_k0 ? Kk[0] : (st_l ? mL : KFF);
            MPC_UNROLL for (int j = 1; j < SB; j++) q_st[j][-d] = Kk[j];
            if (NS >= 4) q_sf[-d] = KFF;
            // closed-loop (Joseph) form: Q + Acl' P Acl + K' R~ K (+ M K + K' M'); wasted (and harmless) for block 0
            double Acl[SB][SB], Tm[SB][SB], RK_[SB], MK[SB][SB];
            MPC_UNROLL for (int j = 0; j < SB; j++) RK_[j] = mm(Rs, Kk[j], 0.0);                                // R~ K
            MPC_UNROLL for (int i = 0; i < SB; i++) {
                MPC_UNROLL for (int j = 0; j < SB; j++) {
                    Acl[i][j] = mm(Btrl[i], Kk[j], Arl[i][j]);              // A + B K
                    Tm[i][j] = mm(BtP[i], Kk[j], PA[i][j]);                 // P Acl = PA + PB K
                    if (HASM) MK[i][j] = mm(Mtr[i], Kk[j], 0.0);            // M K
                }
            }
            MPC_UNROLL for (int i = 0; i < SB; i++) {
                MPC_UNROLL for (int j = 0; j < SB; j++) {
                    double acc = mm(Kk[i], RK_[j], Qr[i][j]), acc2 = 0.0;
                    MPC_UNROLL for (int l = 0; l < SB; l++) acc2 = mm(Acl[l][i], Tm[l][j], acc2);
                    double pn = acc + acc2;
                    if (HASM) pn = mm(MK[j][i], Ir, pn + MK[i][j]);          // + M K + (M K)'
                    Pm[i][j] = pn;      // symmetric up to rounding; the recursion does not amplify the difference
                }
            }
            MPC_UNROLL for (int i = 0; i < SB; i++) { double a2 = mm(Kk[i], HU, 0.0); MPC_UNROLL for (int l = 0; l < SB; l++) a2 = mm(Acl[l][i], PV[l], a2); PC[i] = a2; }      // Acl'(hz + p+) + K' hu
        };
        const int G = N / PD, rem = N - G * PD;
        for (int g = 0; g < G; g++) {       // full groups: the reads of the next group (or of the remainder) are in flight
            MPC_UNROLL for (int d = 0; d < PD; d++) block(d, true);
            MPC_UNROLL for (int i = 0; i < SB; i++) { q_sz[i] -= PD; q_hz[i] -= PD; q_st[i] -= stp[i]; }
            q_su -= PD; q_s0 -= PD; q_s1 -= PD; q_hu -= PD; q_sf -= stpf;
            if (LTV) { q_a -= PD; q_b -= PD; q_bt -= PD; }
        }
        MPC_UNROLL for (int d = 0; d < PD - 1; d++) { if (d < rem) block(d, false); }
    };
    // backward: right-hand-side recursion of the corrector: h (RG rows), K, -Lambda^-1 -> kff
    auto tile_rhs = [&]() {
        const PT &Pl = launder(P);
        const TileCtx c = tile_ctx(Pl);
        const int tr = c.tr, tc = c.tc; const bool live = c.live;
        double *const trash = c.trash;
        const bool st_any = live && tr < NU && tc == 0;
        const double *q_hu = c.p_hu + (N - 1), *q_li = c.p_li + (N - 1);
        const double *q_hz[SB], *q_k[SB];
        MPC_UNROLL for (int i = 0; i < SB; i++) { q_hz[i] = c.p_hz[i] + (N - 1); q_k[i] = c.p_k[i] + (N - 1); }
        double *q_st = st_any ? c.p_kf + (N - 1) : trash + PD;
        const int stp = st_any ? PD : 0;
        double PC[SB];
        MPC_UNROLL for (int i = 0; i < SB; i++) PC[i] = 0.0;
        constexpr int NF = 2 * SB + 2 + (LTV ? 3 : 0);      // hu li hz[SB] k[SB] (a b bt)
        double f[PD][NF];
        const double *q_a = LTV ? c.p_a + (N - 1) : nullptr, *q_b = LTV ? c.p_b + (N - 1) : nullptr, *q_bt = LTV ? c.p_bt + (N - 1) : nullptr;
        auto fetch = [&](int d, int off) {
            f[d][0] = q_hu[off]; f[d][1] = q_li[off];
            MPC_UNROLL for (int i = 0; i < SB; i++) { f[d][2 + i] = q_hz[i][off]; f[d][2 + SB + i] = q_k[i][off]; }
            if (LTV) { f[d][NF - 3] = q_a[off]; f[d][NF - 2] = q_b[off]; f[d][NF - 1] = q_bt[off]; }
        };
        MPC_UNROLL for (int d = 0; d < PD; d++) fetch(d, -d);
        auto block = [&](int d, bool more) {
            const double HU = f[d][0], mLi = f[d][1];
            double HZ[SB], Kk[SB];
            MPC_UNROLL for (int i = 0; i < SB; i++) { HZ[i] = f[d][2 + i]; Kk[i] = f[d][2 + SB + i]; }
            double Arl[SB][SB], Brl[SB], Btrl[SB];
            MPC_UNROLL for (int i = 0; i < SB; i++) { Brl[i] = c.Br[i]; Btrl[i] = c.Btr[i]; MPC_UNROLL for (int jj = 0; jj < SB; jj++) Arl[i][jj] = c.Ar[i][jj]; }
            if (LTV) { Arl[0][0] = f[d][NF - 3]; Brl[0] = f[d][NF - 2]; Btrl[0] = f[d][NF - 1]; }
            if (more) fetch(d, -PD - d);
            double psv = HU;
            MPC_UNROLL for (int i = 0; i < SB; i++) psv = mm(Brl[i], HZ[i] + PC[i], psv);
            q_st[-d] = mm(mLi, psv, 0.0);
            double Acl[SB][SB], PCn[SB];
            MPC_UNROLL for (int i = 0; i < SB; i++) { MPC_UNROLL for (int j = 0; j < SB; j++) Acl[i][j] = mm(Btrl[i], Kk[j], Arl[i][j]); }
            MPC_UNROLL for (int i = 0; i < SB; i++) {
                double cst = mm(Kk[i], HU, 0.0), dyn = 0.0;
                MPC_UNROLL for (int l = 0; l < SB; l++) { cst = mm(Acl[l][i], HZ[l], cst); dyn = mm(Acl[l][i], PC[l], dyn); }
                PCn[i] = cst + dyn;
            }
            MPC_UNROLL for (int i = 0; i < SB; i++) PC[i] = PCn[i];
        };
        const int G = N / PD, rem = N - G * PD;
        for (int g = 0; g < G; g++) {
            MPC_UNROLL for (int d = 0; d < PD; d++) block(d, true);
            MPC_UNROLL for (int i = 0; i < SB; i++) { q_hz[i] -= PD; q_k[i] -= PD; }
            q_hu -= PD; q_li -= PD; q_st -= stp;
            if (LTV) { q_a -= PD; q_b -= PD; q_bt -= PD; }
        }
        MPC_UNROLL for (int d = 0; d < PD - 1; d++) { if (d < rem) block(d, false); }
    };
    // forward: Newton direction: K, kff -> du | dz
    auto tile_forward = [&]() {
        const PT &Pl = launder(P);
        const TileCtx c = tile_ctx(Pl);
        const int tr = c.tr, tc = c.tc; const bool live = c.live;
        double *const trash = c.trash;
        auto trow = [&](int row) -> double * { return c.Tt + (live ? row : RZ) * (NI * LD); };
        // stores: du from column 0 of the rows < NU and dz (state tile 0) from column 1 in one store; further state tiles from their column 1
        const bool st_u = tr < NU && tc == 0;
        double *q_st[SB]; int stp[SB];
        MPC_UNROLL for (int i = 0; i < SB; i++) {
            const int ri = 4 * i + tr;
            const bool st_z = ri < NS && tc == 1;
            const bool any = live && (st_z || (i == 0 && st_u));
            q_st[i] = any ? trow((i == 0 && st_u) ? RG + tr : RG + NU + (ri < NS ? ri : 0)) : trash;
            stp[i] = any ? PD : 0;
        }
        const double *q_kf = c.p_kf;
        const double *q_k[SB], *q_kt[SB];
        MPC_UNROLL for (int i = 0; i < SB; i++) { q_k[i] = c.p_k[i]; q_kt[i] = c.p_kt[i]; }
        double DZ[SB];
        MPC_UNROLL for (int i = 0; i < SB; i++) DZ[i] = 0.0;
        constexpr int NF = 2 * SB + 1 + (LTV ? 2 : 0);      // kff k[SB] kt[SB] (at bt)
        double f[PD][NF];
        const double *q_at = LTV ? c.p_at : nullptr, *q_bt = LTV ? c.p_bt : nullptr;
        auto fetch = [&](int d, int off) {
            f[d][0] = q_kf[off];
            MPC_UNROLL for (int i = 0; i < SB; i++) { f[d][1 + i] = q_k[i][off]; f[d][1 + SB + i] = q_kt[i][off]; }
            if (LTV) { f[d][NF - 2] = q_at[off]; f[d][NF - 1] = q_bt[off]; }
        };
        MPC_UNROLL for (int d = 0; d < PD; d++) fetch(d, d);
        auto block = [&](int d, bool more) {
            const double KFF = f[d][0];
            double Kk[SB], KkT[SB];
            MPC_UNROLL for (int i = 0; i < SB; i++) { Kk[i] = f[d][1 + i]; KkT[i] = f[d][1 + SB + i]; }
            double Atrl[SB][SB], Btrl[SB];
            MPC_UNROLL for (int i = 0; i < SB; i++) { Btrl[i] = c.Btr[i]; MPC_UNROLL for (int jj = 0; jj < SB; jj++) Atrl[i][jj] = c.Atr[i][jj]; }
            if (LTV) { Atrl[0][0] = f[d][NF - 2]; Btrl[0] = f[d][NF - 1]; }
            if (more) fetch(d, PD + d);
            double DU = KFF;
            MPC_UNROLL for (int j = 0; j < SB; j++) DU = mm(KkT[j], DZ[j], DU);                      // K dz + kff
            double DZn[SB];
            MPC_UNROLL for (int i = 0; i < SB; i++) {
                double a = mm(Btrl[i], KFF, 0.0);                                                // B kff
                MPC_UNROLL for (int j = 0; j < SB; j++) a = mm(mm(Kk[j], Btrl[i], Atrl[j][i]), DZ[j], a);        // (A + B K)'[j][i]' dz_j = Acl[i][j] dz_j
                DZn[i] = a;
            }
            q_st[0][d] = st_u ? DU : DZn[0];
            MPC_UNROLL for (int i = 1; i < SB; i++) q_st[i][d] = DZn[i];
            MPC_UNROLL for (int i = 0; i < SB; i++) DZ[i] = DZn[i];
        };
        const int G = N / PD, rem = N - G * PD;
        for (int g = 0; g < G; g++) {
            MPC_UNROLL for (int d = 0; d < PD; d++) block(d, true);
            MPC_UNROLL for (int i = 0; i < SB; i++) { q_k[i] += PD; q_kt[i] += PD; q_st[i] += stp[i]; }
            q_kf += PD;
            if (LTV) { q_at += PD; q_bt += PD; }
        }
        MPC_UNROLL for (int d = 0; d < PD - 1; d++) { if (d < rem) block(d, false); }
    };

    AReg2 dvp[NJ][NC];       // predictor direction of each instance's bounded variables, kept for the corrector's second-order terms
    for (int it = 0;; it++) {
        bool any_on = false;
        MPC_UNROLL for (int j = 0; j < NJ; j++) any_on = any_on || (PAIR ? __any(S[j].on) != 0 : S[j].on);
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
            MPC_UNROLL for (int j = 0; j < NJ; j++) {
                if (S[j].on && (bad & (0x000F000F000F000FULL << (4 * ji(j))))) { S[j].on = false; S[j].status = kInfeasible; S[j].iters = it; }
            }
        }
        // ================= element-wise: predictor step length, centring, corrector rhs -> LDS ===========================
        MPC_UNROLL for (int j = 0; j < NJ; j++) {
            WvInst &Sj = S[j];
            if (Sj.on) {
                Iter Xj; X[j].get(Xj);
                const PT &Pl = P;
                Bnd Bd; bounds(Pl, j, Bd);
                double maff_p = 1.0, s1_p = 0.0, s2_p = 0.0, pl[NC], ph[NC], dvj[NC];
                MPC_UNROLL for (int i = 0; i < NC; i++) { dvj[i] = tk(RG + i, j); a_put(dvp[j][i], dvj[i]); }      // du | dz of the predictor
                MPC_UNROLL for (int i = 0; i < NC; i++) {
                    const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                    const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                    const double rh = Bd.fh[i] ? v + Xj.sh[i] - Bd.hi[i] : 0.0, rl = Bd.fl[i] ? v - Xj.sl[i] - Bd.lo[i] : 0.0;
                    const double dsh = Bd.fh[i] ? -rh - dvj[i] : 0.0, dsl = Bd.fl[i] ? rl + dvj[i] : 0.0;
                    const double qh = dsh * ish, ql = dsl * isl;
                    const double dlh = Bd.fh[i] ? -Xj.lh[i] - Xj.lh[i] * qh : 0.0, dll = Bd.fl[i] ? -Xj.ll[i] - Xj.ll[i] * ql : 0.0;
                    maff_p = dmax(maff_p, dmax(-ql, -qh));
                    if (Bd.fl[i]) maff_p = dmax(maff_p, 1.0 + ql);
                    if (Bd.fh[i]) maff_p = dmax(maff_p, 1.0 + qh);
                    s1_p += Xj.sl[i] * dll + Xj.ll[i] * dsl + Xj.sh[i] * dlh + Xj.lh[i] * dsh;
                    s2_p += dsl * dll + dsh * dlh;
                    pl[i] = dsl * dll; ph[i] = dsh * dlh;
                }
                const double m_aff = rmax(blk_on ? maff_p : 1.0), s1 = rsum(blk_on ? s1_p : 0.0), s2 = rsum(blk_on ? s2_p : 0.0);
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
                if (blk_on) {
                    MPC_UNROLL for (int i = 0; i < NU; i++) tk(RG + i, j) = gu[i] + hc[i];
                    MPC_UNROLL for (int i = 0; i < NS; i++) tk(RG + NU + i, j) = gz[i] + hc[NU + i];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
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
        MPC_UNROLL for (int j = 0; j < NJ; j++) {
            WvInst &Sj = S[j];
            if (Sj.on) {
                Iter Xj; X[j].get(Xj);
                const PT &Pl = P;
                Bnd Bd; bounds(Pl, j, Bd);
                double dvzj[NV], dvj[NC];
                MPC_UNROLL for (int i = 0; i < NC; i++) dvj[i] = a_get(dvp[j][i]);
                MPC_UNROLL for (int i = 0; i < NV; i++) dvzj[i] = tk(RG + i, j);
                double mcc_p = kTau;
                double dsl[NC], dsh[NC], dll[NC], dlh[NC];
                MPC_UNROLL for (int i = 0; i < NC; i++) {
                    const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                    const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                    const double rh = Bd.fh[i] ? v + Xj.sh[i] - Bd.hi[i] : 0.0, rl = Bd.fl[i] ? v - Xj.sl[i] - Bd.lo[i] : 0.0;
                    // second-order products of the predictor direction (recomputed, not stored)
                    const double ash = Bd.fh[i] ? -rh - dvj[i] : 0.0, asl = Bd.fl[i] ? rl + dvj[i] : 0.0;
                    const double alh = Bd.fh[i] ? -Xj.lh[i] - Xj.lh[i] * (ash * ish) : 0.0, all_ = Bd.fl[i] ? -Xj.ll[i] - Xj.ll[i] * (asl * isl) : 0.0;
                    const double rch = Bd.fh[i] ? Xj.sh[i] * Xj.lh[i] - dmax(Sj.sm, Xj.lh[i] * kSFloor) + ash * alh : 0.0;
                    const double rcl = Bd.fl[i] ? Xj.sl[i] * Xj.ll[i] - dmax(Sj.sm, Xj.ll[i] * kSFloor) + asl * all_ : 0.0;
                    dsh[i] = Bd.fh[i] ? -rh - dvzj[i] : 0.0; dsl[i] = Bd.fl[i] ? rl + dvzj[i] : 0.0;
                    dlh[i] = Bd.fh[i] ? (-rch - Xj.lh[i] * dsh[i]) * ish : 0.0; dll[i] = Bd.fl[i] ? (-rcl - Xj.ll[i] * dsl[i]) * isl : 0.0;
                    mcc_p = dmax(mcc_p, dmax(-dsl[i] * isl, -dsh[i] * ish));
                    if (Bd.fl[i]) mcc_p = dmax(mcc_p, -dll[i] * frcp_approx(Xj.ll[i]));
                    if (Bd.fh[i]) mcc_p = dmax(mcc_p, -dlh[i] * frcp_approx(Xj.lh[i]));
                }
                const double m_cc = rmax(blk_on ? mcc_p : kTau);
                const double alpha = m_cc <= kTau ? 1.0 : kTau * frcp(m_cc);
                MPC_UNROLL for (int i = 0; i < NC; i++) { Xj.sl[i] += alpha * dsl[i]; Xj.sh[i] += alpha * dsh[i]; Xj.ll[i] += alpha * dll[i]; Xj.lh[i] += alpha * dlh[i]; }
                MPC_UNROLL for (int i = 0; i < NU; i++) Xj.u[i] += alpha * dvzj[i];
                MPC_UNROLL for (int i = 0; i < NS; i++) Xj.z[i] += alpha * dvzj[NU + i];
                X[j].put(Xj);
                phase_a(Pl, j, Sj, Xj, it + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        MPC_TSTAMP(6);
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Estimator and target problem with 16 lanes per instance (lane = 16 b + r, b = instance of the wave, r = row): the four
// instances of a wave at once, sums over the rows on the DPP network (row16_sum / row16_max), matrices exchanged through LDS.
// The lane = instance versions (kalman_lane, target_lane in mpc_device.hpp) keep 60 of 64 lanes idle for ~6000 instructions.
// Same operations on the same numbers; sums over rows are taken in tree order.
// ------------------------------------------------------------------------------------------------------------------------

// per-row constants in LDS, filled once per launch (row r < NE: Aa[r][:] Qkf[r][:] Kfix[r][:] dmin dmax; row r < NCT: W[r][:] tlo thi)
template <int NX, int NU, int NY, int ND>
struct Row16Tab {
    static constexpr int NE = NX + ND, NR = NU, NCT = NX + NU + NY;
    static constexpr int E_A = 0, E_Q = NE, E_K = 2 * NE, E_DMIN = 2 * NE + NY, E_DMAX = E_DMIN + 1, ESZ = E_DMAX + 1;
    static constexpr int T_W = 0, T_LO = NR, T_HI = NR + 1, TSZ = NR + 2;
    static constexpr int DOUBLES = NE * ESZ + NCT * TSZ;
    static constexpr bool fits = NE <= 16 && NCT <= 16;
    // exchange space per instance (in the transposing buffer, which is idle outside the OCP solve)
    static constexpr int XCH = NE * NY + NE * NE;
};

template <int NX, int NU, int NY, int ND, class PT>
__device__ __forceinline__ void row16_fill_tables(const PT &P, double *tab, int lane)
{
    using RT = Row16Tab<NX, NU, NY, ND>;
    constexpr int NE = RT::NE;
    if (lane < NE) {
        double *e = tab + lane * RT::ESZ;
        for (int j = 0; j < NE; j++) { e[RT::E_A + j] = P.Aa[lane][j]; e[RT::E_Q + j] = P.Qkf[lane][j]; }
        for (int j = 0; j < NY; j++) e[RT::E_K + j] = P.Kfix[lane][j];
        e[RT::E_DMIN] = lane >= NX ? P.dmin[lane - NX] : 0.0; e[RT::E_DMAX] = lane >= NX ? P.dmax[lane - NX] : 0.0;
    }
    if (lane < RT::NCT) {
        double *t = tab + NE * RT::ESZ + lane * RT::TSZ;
        for (int c = 0; c < NU; c++) t[RT::T_W + c] = P.W[lane][c];
        t[RT::T_LO] = P.tlo[lane]; t[RT::T_HI] = P.thi[lane];
    }
}

// Estimator step (MPC_code.py:524-534, 577-668; kalman(): Estimator.py:297-309): xi = [xhat; dhat] at kp[xi_off + r], plant state at
// kp[x_off..], covariance rows at kp[p_off + r * NE ..].  On return lane r < NE holds xi_old / xi_new of its row (also written back).
// slot_on: this lane's instance slot exists (lanes of unused slots compute along on slot 0's data and write nothing).
template <int NX, int NY, int ND, int NXP, int NU, class PT>
__device__ __forceinline__ void kalman_row16(const PT &P, int r, bool slot_on, double *kp, int x_off, int xi_off, int p_off, const double *pyp_k,
                                             const double *tab, double *xch, double &xi_old, double &xi_new)
{
    using RT = Row16Tab<NX, NU, NY, ND>;
    constexpr int NE = RT::NE;
    const bool row = r < NE && slot_on;      // lanes of an unused instance slot (fewer than four instances per wave) alias slot 0: read only
    const int rr = r < NE ? r : 0;
    const double *e = tab + rr * RT::ESZ;
    double xi[NE], innov[NY];
    MPC_UNROLL for (int i = 0; i < NE; i++) xi[i] = kp[xi_off + i];
    MPC_UNROLL for (int i = 0; i < NY; i++) {       // yhat = Fy_model(xhat, dhat) :524, y = Fy_p(x) + pyp :531-534
        double yh = P.fyc[i], yy = pyp_k[i];
        MPC_UNROLL for (int j = 0; j < NE; j++) yh += P.Ca[i][j] * xi[j];
        MPC_UNROLL for (int j = 0; j < NXP; j++) yy += P.Cp[i][j] * kp[x_off + j];
        innov[i] = yy - yh;
    }
    xi_old = kp[xi_off + rr];
    double Kr[NY];
    if (P.estimator == MPC_EST_KALMAN) {
        double Pr[NE], PCt[NY];
        MPC_UNROLL for (int j = 0; j < NE; j++) Pr[j] = kp[p_off + rr * NE + j];
        MPC_UNROLL for (int j = 0; j < NY; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += Pr[l] * P.Ca[j][l]; PCt[j] = a; }
        if (row) { MPC_UNROLL for (int j = 0; j < NY; j++) xch[r * NY + j] = PCt[j]; }
        __syncthreads();
        double all[NE][NY], S[NY][NY];
        MPC_UNROLL for (int l = 0; l < NE; l++) { MPC_UNROLL for (int j = 0; j < NY; j++) all[l][j] = xch[l * NY + j]; }
        MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = P.Rkf[i][j]; MPC_UNROLL for (int l = 0; l < NE; l++) a += P.Ca[i][l] * all[l][j]; S[i][j] = a; } }
        MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (S[i][j] + S[j][i]); S[i][j] = a; S[j][i] = a; } }
        sym_inverse<NY>(S);                                   // K = P C' S^-1   (Estimator.py:297)
        MPC_UNROLL for (int j = 0; j < NY; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) a += PCt[l] * S[l][j]; Kr[j] = a; }
        // P_corr = (I - K C) P :300 (C P = (P C')' up to the rounding of P's symmetry); U = P_corr Aa'; P+ = Aa U + Q :309
        double Pc[NE], U[NE];
        MPC_UNROLL for (int j = 0; j < NE; j++) { double a = Pr[j]; MPC_UNROLL for (int m = 0; m < NY; m++) a -= Kr[m] * all[j][m]; Pc[j] = a; }
        MPC_UNROLL for (int j = 0; j < NE; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += Pc[l] * P.Aa[j][l]; U[j] = a; }
        double *xu = xch + NE * NY;
        if (row) { MPC_UNROLL for (int j = 0; j < NE; j++) xu[r * NE + j] = U[j]; }
        __syncthreads();
        double Pn[NE];
        MPC_UNROLL for (int j = 0; j < NE; j++) Pn[j] = e[RT::E_Q + j];
        MPC_UNROLL for (int l = 0; l < NE; l++) { const double al = e[RT::E_A + l]; MPC_UNROLL for (int j = 0; j < NE; j++) Pn[j] += al * xu[l * NE + j]; }
        if (row) { MPC_UNROLL for (int j = 0; j < NE; j++) kp[p_off + r * NE + j] = Pn[j]; }
    } else {
        MPC_UNROLL for (int j = 0; j < NY; j++) Kr[j] = e[RT::E_K + j];
    }
    double v = xi_old;
    MPC_UNROLL for (int l = 0; l < NY; l++) v += Kr[l] * innov[l];                                     // :303-306
    if (P.has_dsat && r >= NX) v = dmin(dmax(v, e[RT::E_DMIN]), e[RT::E_DMAX]);                     // MPC_code.py:655-668
    xi_new = v;
    __syncthreads();      // everybody has read the old xi
    if (row) kp[xi_off + r] = v;
}

// Target problem (Target_Calc.py:20-161 + MPC_code.py:693-718) reduced to the null space of [A-I, B], one
// constraint row per lane.  dh / us_prev: this instance's data (the same in its 16 lanes); tw: its warm-start record in LDS
// (y[NR] l_lo[NC] l_hi[NC] gr[NR] w0[NC]) with validity flag twv.  Returns the status (the same in the 16 lanes); v_out = row r of
// [xs; us; ys] of the final point.
template <int NX, int NU, int NY, int ND, class PT>
__device__ __forceinline__ int target_row16(const PT &P, int r, const double *tab, const double *usp, const double *ysp, const double *dh,
                                            const double *us_prev, double *tw, int *twv, bool inst_on, double &v_out, int &iters)
{
    using RT = Row16Tab<NX, NU, NY, ND>;
    constexpr int NV = NX + NU, NC = RT::NCT, NR = NU, NE = RT::NE;
    const bool row = r < NC;
    const int rr = row ? r : 0;
    const double *t = tab + NE * RT::ESZ + rr * RT::TSZ;
    double Wr[NR];
    MPC_UNROLL for (int c = 0; c < NR; c++) Wr[c] = row ? t[RT::T_W + c] : 0.0;
    const double tlo = t[RT::T_LO], thi = t[RT::T_HI];
    const bool fl = row && fin(tlo), fh = row && fin(thi);
    const double lo = fl ? tlo : 0.0, hi = fh ? thi : 0.0;
    double cx[NX], e[NY], vp[NV], yp[NY], gr[NR], y[NR];
    MPC_UNROLL for (int i = 0; i < NX; i++) { double a = P.fxc[i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += P.Bd[i][j] * dh[j]; cx[i] = a; }
    MPC_UNROLL for (int i = 0; i < NY; i++) { double a = P.fyc[i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += P.Cd[i][j] * dh[j]; e[i] = a; }
    MPC_UNROLL for (int q = 0; q < NV; q++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NX; j++) a -= P.Ep[q][j] * cx[j]; vp[q] = a; }
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
    double w0 = 0.0;      // this lane's row of [vp; yp]
    MPC_UNROLL for (int q = 0; q < NV; q++) w0 = r == q ? vp[q] : w0;
    MPC_UNROLL for (int i = 0; i < NY; i++) w0 = r == NV + i ? yp[i] : w0;
    const double ncon = row16_sum((fl ? 1.0 : 0.0) + (fh ? 1.0 : 0.0));
    const double inv_ncon = 1.0 / dmax(ncon, 1.0);
    {
        double Hi[NR][NR];
        MPC_UNROLL for (int i = 0; i < NR; i++) { MPC_UNROLL for (int j = 0; j < NR; j++) Hi[i][j] = P.Hr[i][j]; }
        sym_inverse<NR>(Hi);
        MPC_UNROLL for (int i = 0; i < NR; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NR; j++) a += Hi[i][j] * gr[j]; y[i] = -a; }
    }
    double s_lo, s_hi, l_lo, l_hi;
    {
        double v = w0; MPC_UNROLL for (int c = 0; c < NR; c++) v += Wr[c] * y[c];
        s_lo = fl ? dmax(v - lo, kSMin) : 1.0; s_hi = fh ? dmax(hi - v, kSMin) : 1.0;
        l_lo = fl ? kMu0 * frcp(s_lo) : 0.0; l_hi = fh ? kMu0 * frcp(s_hi) : 0.0;
    }
    if (*twv != 0) {
        double dl = row ? fabs(w0 - tw[2 * NR + 2 * NC + rr]) : 0.0;
        MPC_UNROLL for (int c = 0; c < NR; c++) dl = dmax(dl, fabs(gr[c] - tw[NR + 2 * NC + c]));
        const double delta = row16_max(dl);
        if (delta <= kWsDelta) {
            const double smin = dmin(dmax(kWsKappa * delta, kWsSMinLo), kWsSMinHi), wmu = kWsMuFactor * smin * smin;
            MPC_UNROLL for (int c = 0; c < NR; c++) y[c] = tw[c];
            double v = w0; MPC_UNROLL for (int c = 0; c < NR; c++) v += Wr[c] * y[c];
            s_lo = fl ? dmax(v - lo, smin) : 1.0; s_hi = fh ? dmax(hi - v, smin) : 1.0;
            l_lo = fl ? dmax(tw[NR + rr], wmu * frcp(s_lo)) : 0.0; l_hi = fh ? dmax(tw[NR + NC + rr], wmu * frcp(s_hi)) : 0.0;
        }
    }
    double gscale = 1.0; int stall = 0, status = kMaxIter;
    MPC_UNROLL for (int c = 0; c < NR; c++) gscale = dmax(gscale, fabs(gr[c]));
    bool on = inst_on;
    iters = 0;
    for (int it = 0;; it++) {
        if (!__any(on ? 1 : 0)) break;      // the four instances of the wave have their verdicts
        // reciprocals of the slacks once per iteration (v_rcp_f64 + Newton, mpc::frcp) instead of IEEE divisions
        double v = w0; MPC_UNROLL for (int c = 0; c < NR; c++) v += Wr[c] * y[c];
        const double r_lo = fl ? v - s_lo - lo : 0.0, r_hi = fh ? v + s_hi - hi : 0.0;
        const double is_lo = frcp(s_lo), is_hi = frcp(s_hi);
        const double sig = l_lo * is_lo + l_hi * is_hi;
        const double mu = row16_sum(s_lo * l_lo + s_hi * l_hi) * inv_ncon;
        const double res_p = row16_max(dmax(fabs(r_lo), fabs(r_hi)));
        const double cres = row16_max(dmax(comp_measure(s_lo, l_lo), comp_measure(s_hi, l_hi)));
        const double lmax = row16_max(dmax(l_lo, l_hi));
        double grad[NR], res_s = 0.0;
        MPC_UNROLL for (int c = 0; c < NR; c++) {
            double a = gr[c]; MPC_UNROLL for (int j = 0; j < NR; j++) a += P.Hr[c][j] * y[j];
            a += row16_sum((l_hi - l_lo) * Wr[c]);
            grad[c] = a; res_s = dmax(res_s, fabs(a));
        }
        if (on) {
            iters = it;
            const bool ok_cp = (cres <= 1.0) && (res_p <= kTolFeas);
            stall = ok_cp ? stall + 1 : 0;
            if (ok_cp && (res_s <= kTolStat * gscale || (stall > kStallMax && res_s <= kTolStatAcc * gscale))) { status = kSolved; on = false; }
            else if (lmax > kInfeasZ * gscale || !(fabs(mu) < 1.0e300)) { status = kInfeasible; on = false; }
            else if (it == P.max_iter) { status = kMaxIter; on = false; }
        }
        double Ht[NR][NR];
        MPC_UNROLL for (int i = 0; i < NR; i++) { MPC_UNROLL for (int j = 0; j <= i; j++) { const double a = P.Hr[i][j] + row16_sum(sig * Wr[i] * Wr[j]); Ht[i][j] = a; Ht[j][i] = a; } }
        const bool pd = sym_inverse<NR>(Ht);
        if (on && !pd) { status = kInfeasible; on = false; }
        double dy[NR], ds_lo = 0.0, ds_hi = 0.0, dl_lo = 0.0, dl_hi = 0.0, sm = 0.0, alpha = 1.0;
        MPC_UNROLL for (int pass = 0; pass < 2; pass++) {
            double rc_lo, rc_hi, rhs[NR];
            if (pass == 0) { rc_lo = fl ? s_lo * l_lo : 0.0; rc_hi = fh ? s_hi * l_hi : 0.0; }
            else {
                rc_lo = fl ? s_lo * l_lo - dmax(sm, l_lo * kSFloor) + ds_lo * dl_lo : 0.0;
                rc_hi = fh ? s_hi * l_hi - dmax(sm, l_hi * kSFloor) + ds_hi * dl_hi : 0.0;
            }
            const double h = (-rc_hi + l_hi * r_hi) * is_hi + (rc_lo + l_lo * r_lo) * is_lo;
            MPC_UNROLL for (int c = 0; c < NR; c++) rhs[c] = grad[c] + row16_sum(h * Wr[c]);
            MPC_UNROLL for (int i = 0; i < NR; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NR; j++) a += Ht[i][j] * rhs[j]; dy[i] = -a; }
            // step to the boundary = 1 / m with m = max_i (-d_i / x_i); predictor capped at 1 (m >= 1), corrector at 1 / tau
            double dv = 0.0; MPC_UNROLL for (int c = 0; c < NR; c++) dv += Wr[c] * dy[c];
            ds_hi = fh ? -r_hi - dv : 0.0; ds_lo = fl ? r_lo + dv : 0.0;
            dl_hi = fh ? (-rc_hi - l_hi * ds_hi) * is_hi : 0.0;
            dl_lo = fl ? (-rc_lo - l_lo * ds_lo) * is_lo : 0.0;
            double m = dmax(pass == 0 ? 1.0 : kTau, dmax(-ds_lo * is_lo, -ds_hi * is_hi));
            if (fl) m = dmax(m, -dl_lo * frcp_approx(l_lo));
            if (fh) m = dmax(m, -dl_hi * frcp_approx(l_hi));
            m = row16_max(m);
            if (pass == 0) {
                const double amax = frcp(m);
                // rows without a bound carry s = 1, l = 0, ds = dl = 0: they add nothing
                const double s1 = row16_sum((s_lo + amax * ds_lo) * (l_lo + amax * dl_lo) + (s_hi + amax * ds_hi) * (l_hi + amax * dl_hi));
                const double mu_aff = s1 * inv_ncon, rat = mu > 0.0 ? mu_aff * frcp(mu) : 0.0;
                sm = dmax(rat * rat * rat * mu, kMuFloor);
            } else alpha = m <= kTau ? 1.0 : kTau * frcp(m);
        }
        if (on) {
            MPC_UNROLL for (int c = 0; c < NR; c++) y[c] += alpha * dy[c];
            s_lo += alpha * ds_lo; s_hi += alpha * ds_hi; l_lo += alpha * dl_lo; l_hi += alpha * dl_hi;
        }
    }
    {
        double v = w0; MPC_UNROLL for (int c = 0; c < NR; c++) v += Wr[c] * y[c];
        v_out = v;
    }
    if (inst_on) {
        if (row) { tw[NR + r] = l_lo; tw[NR + NC + r] = l_hi; tw[2 * NR + 2 * NC + r] = w0; }
        if (r == 0) { MPC_UNROLL for (int c = 0; c < NR; c++) { tw[c] = y[c]; tw[NR + 2 * NC + c] = gr[c]; } *twv = status == kSolved ? 1 : 0; }
    }
    return status;
}

}  // namespace mpc
