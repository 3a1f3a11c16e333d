// Horizon-parallel variant of the OCP solver for small and medium batches.
//
// rpdip_lane (mpc_device.hpp) maps one instance to one lane and walks the horizon sequentially four times per
// interior-point iteration, streaming every block through HBM.  At batch 4096 that occupies 64 of the 1024 SIMDs.
// Here a workgroup of NW waves owns NI = NW * IPW instances:
//   * element-wise work (slacks, multipliers, residuals, step lengths, the convergence test) runs with wave = instance
//     (IPW instances per wave, one after the other) and lane = block k of the horizon (N <= 64): sums and maxima over the
//     horizon are DPP reductions over the wave, the costate recursion of the test is a scan over the lanes;
//   * the recursions that are sequential in k run on dedicated waves while the others wait at the barrier:
//       - the matrix recursion (Riccati factorisation) on the matrix cores when the stage fits a 4x4 tile (NS <= 4, NU <= 2):
//         v_mfma_f64_4x4x4f64 multiplies four pairs of 4x4 tiles per wave, so one wave factorises four instances and NI / 4
//         waves work side by side; otherwise on wave 0 with lane = instance, as rpdip_lane does it;
//       - the vector recursions (right-hand sides, Newton direction) on the next wave with lane = instance; for the
//         predictor it follows the factorisation block by block through progress words in LDS;
//     the mappings exchange their data through a transposing buffer in LDS ([row][instance][k], padded so that the
//     lane = k and lane = instance views are conflict-free).
// Between phases the iterate of an instance (slacks, multipliers, inputs, states: ROWS_ST rows of 64 doubles) rests in
// HBM/L2, one coalesced row per quantity; a wave loads it, works, stores what changed.  That keeps the register budget
// of a wave independent of how many instances it serves (IPW) and leaves the recursion waves their registers
// (keeping the iterates in registers instead was measured: hipcc spills them next to the recursion code).
// The same rows are the warm start of the next MPC step.
// The arithmetic per block is that of rpdip_lane; horizon-wide sums are taken in another order.
#pragma once
#include "mpc_device.hpp"

#ifdef MPC_STAMPS
#define MPC_TSTAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
    if (threadIdx.x == 0 && threadIdx.y == 0) mpc_stamp_buf[(blockIdx.x & 4095) * 8 + (slot)] += t_ - stamp_prev_; stamp_prev_ = t_; } while (0)
#else
#define MPC_TSTAMP(slot) do { } while (0)
#endif
// -DMPC_STAMPS_FINE (with -DMPC_STAMPS): the iteration phases share slot 3, slot 4 takes wave 0's matrix recursion and
// slots 5, 6, 2 three segments of its loop body (sums kept in registers, written once per pass)
#ifdef MPC_STAMPS_FINE
#define MPC_TSTAMP_IT(slot) MPC_TSTAMP(3)
#define MPC_TSTAMP_FINE(slot) MPC_TSTAMP(slot)
#else
#define MPC_TSTAMP_IT(slot) MPC_TSTAMP(slot)
#define MPC_TSTAMP_FINE(slot) do { } while (0)
#endif

namespace mpc {

template <int NS, int NU, int NC, int NW, int IPW>
struct TpCfg {
    static constexpr int NI = NW * IPW;                              // instances per workgroup: NW waves, IPW instances each
    static constexpr int NV = NS + NU, NKF = NU * NS, NLI = NU * (NU + 1) / 2;
    // rows of the transposing buffer, per (instance, block):
    //   RA: sigma (element-wise -> Riccati), overwritten by K | Lambda^-1 (kept for the corrector)
    //   RG: gu + hu | gz + hz (-> predictor rhs), then du | dz (direction ->), then the same for the corrector
    //   RK: kff
    static constexpr int RA = 0, RA_SZ = (NC > NKF + NLI ? NC : NKF + NLI);
    static constexpr int RG = RA_SZ, RK = RG + NV, ROWS = RK + NU;
    static constexpr int LD = 65;                                    // 64 blocks + 1: lane = instance reads hit distinct banks
    static constexpr int T_DOUBLES = ROWS * NI * LD;
    static constexpr int QN = 5 * NS + 2 * NU + 1;                   // z0 zr c zlo zhi | ur us | ws_delta
    // state rows of an instance in HBM, [row][64 blocks]: s_lo s_hi l_lo l_hi dv (NC each) | u | z
    static constexpr int ST_SL = 0, ST_SH = NC, ST_LL = 2 * NC, ST_LH = 3 * NC, ST_DV = 4 * NC, ST_U = 5 * NC, ST_Z = 5 * NC + NU, ROWS_ST = 5 * NC + NV;
    // per-instance data the closed loop keeps in LDS across the steps of one launch (HBM copy at launch start / end):
    // the warm start of the target problem (its size depends on ny, which this struct does not know: room for ny <= 8) + flag
    static constexpr int KEEP_MAX = 2 * NU + 3 * (NS + NU + 8);
    static constexpr size_t lds_bytes() { return sizeof(double) * (T_DOUBLES + NI * QN + NI * 4 + NI * KEEP_MAX) + sizeof(int) * (4 * NI + 4); }
};

// Horizon-wide sums and maxima = reductions over the 64 lanes of a wave, on the DPP network (no LDS round trips):
// butterflies inside each row of 16 lanes (quad swaps, half-row mirror, row mirror), then row_bcast:15 / row_bcast:31
// fold the four rows into lane 63, which is read back into scalar registers (the result is wave-uniform).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double old, double v)
{
    union { double d; int i[2]; } a, o, r; a.d = v; o.d = old;
    r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], a.i[0], CTRL, ROW_MASK, 0xF, false);
    r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], a.i[1], CTRL, ROW_MASK, 0xF, false);
    return r.d;
}
// the neighbour lane's value over the whole wave: wave_shr:1 (lane i gets lane i - 1, lane 0 keeps `old`) and wave_shl:1 (lane i
// gets lane i + 1, lane 63 keeps `old`) - ALU latency, where __shfl_up / __shfl_down go through the LDS crossbar (tools/calib/dpp_wave_shift.hip)
__device__ __forceinline__ double wave_up1(double old, double v) { return dpp_move<0x138, 0xF>(old, v); }
__device__ __forceinline__ double wave_dn1(double old, double v) { return dpp_move<0x130, 0xF>(old, v); }
__device__ __forceinline__ double lane63(double v)
{
    union { double d; int i[2]; } x; x.d = v;
    x.i[0] = __builtin_amdgcn_readlane(x.i[0], 63); x.i[1] = __builtin_amdgcn_readlane(x.i[1], 63);
    return x.d;
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_move<0xB1, 0xF>(0.0, v);        // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xF>(0.0, v);        // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xF>(0.0, v);       // row_half_mirror
    v += dpp_move<0x140, 0xF>(0.0, v);       // row_mirror
    v += dpp_move<0x142, 0xA>(0.0, v);       // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xC>(0.0, v);       // row_bcast:31 into rows 2 and 3
    return lane63(v);
}
__device__ __forceinline__ double wave_max(double v)
{
    v = dmax(v, dpp_move<0xB1, 0xF>(v, v));
    v = dmax(v, dpp_move<0x4E, 0xF>(v, v));
    v = dmax(v, dpp_move<0x141, 0xF>(v, v));
    v = dmax(v, dpp_move<0x140, 0xF>(v, v));
    v = dmax(v, dpp_move<0x142, 0xA>(v, v));
    v = dmax(v, dpp_move<0x143, 0xC>(v, v));
    return lane63(v);
}
// the same over each half of the wave (lanes 0-31 | 32-63), every lane getting its half's result: the first five steps of the tree
// above, so a half whose partner holds zeros (resp. the neutral element) gives the bits of the whole-wave result.  Safe under an exec
// mask that switches whole halves off (the steps stay inside a half, v_readlane ignores exec).
__device__ __forceinline__ double lane_of(double v, int l)
{
    union { double d; int i[2]; } x; x.d = v;
    x.i[0] = __builtin_amdgcn_readlane(x.i[0], l); x.i[1] = __builtin_amdgcn_readlane(x.i[1], l);
    return x.d;
}
__device__ __forceinline__ double half_sum(double v)
{
    v += dpp_move<0xB1, 0xF>(0.0, v);
    v += dpp_move<0x4E, 0xF>(0.0, v);
    v += dpp_move<0x141, 0xF>(0.0, v);
    v += dpp_move<0x140, 0xF>(0.0, v);
    v += dpp_move<0x142, 0xA>(0.0, v);
    const double lo = lane_of(v, 31), hi = lane_of(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ double half_max(double v)
{
    v = dmax(v, dpp_move<0xB1, 0xF>(v, v));
    v = dmax(v, dpp_move<0x4E, 0xF>(v, v));
    v = dmax(v, dpp_move<0x141, 0xF>(v, v));
    v = dmax(v, dpp_move<0x140, 0xF>(v, v));
    v = dmax(v, dpp_move<0x142, 0xA>(v, v));
    const double lo = lane_of(v, 31), hi = lane_of(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ double uni(double v)       // a wave-uniform value into scalar registers
{
    union { double d; int i[2]; } x; x.d = v;
    x.i[0] = __builtin_amdgcn_readfirstlane(x.i[0]); x.i[1] = __builtin_amdgcn_readfirstlane(x.i[1]);
    return x.d;
}

// Pointers into the dynamic LDS of the workgroup
template <int NS, int NU, int NC, int NW, int IPW>
struct TpShared {
    using Cfg = TpCfg<NS, NU, NC, NW, IPW>;
    static constexpr int NI = Cfg::NI;
    double *T, *q, *red, *keep; int *flag, *iflag, *iters, *keepflag, *misc;
    __device__ explicit TpShared(double *base)
    {
        T = base; q = T + Cfg::T_DOUBLES; red = q + NI * Cfg::QN; keep = red + NI * 4;
        flag = (int *)(keep + NI * Cfg::KEEP_MAX); iflag = flag + NI; iters = iflag + NI; keepflag = iters + NI; misc = keepflag + NI;
    }
    // rows in groups of eight: inside a group the row offset fits the 16-bit immediate of ds_read / ds_write, so that unrolled code
    // with constant rows needs one address register per group and (instance, block), not one addition per access
    __device__ __forceinline__ double &t(int row, int inst, int k) const
    {
        double *g = T + (((row >> 3) * 8 * NI + inst) * Cfg::LD + k);
        return g[(row & 7) * NI * Cfg::LD];
    }
};

enum : int { kTpOk0 = 1, kTpWarm = 2, kTpValid = 4 };

// Called by all NW*64 threads (lane = threadIdx.x, wave = threadIdx.y).  Instance data come from sh.q / sh.iflag
// (written by wave 0, lanes < NI, before the call; a barrier is taken here).  On return wave 0, lane i < NI holds
// status / iters of instance i; the final iterate is in the state rows (u0 = row ST_U.. at k = 0, z1 = ST_Z.. at k = 0).
// wsg: state rows of this workgroup's instances [NI][ROWS_ST][64]; they hold the previous solve's iterate on entry.
template <int NS, int NU, bool HASM, int NC, bool MASKED, int NW, int IPW>
__device__ void tp_solve(const DevProblem &P, const TpShared<NS, NU, NC, NW, IPW> &sh, double *__restrict__ wsg,
                         int max_iter, int &status_o, int &iters_o)
{
    using Cfg = TpCfg<NS, NU, NC, NW, IPW>;
    constexpr int NV = Cfg::NV, NKF = Cfg::NKF, NLI = Cfg::NLI, RA = Cfg::RA, RG = Cfg::RG, RK = Cfg::RK, NI = Cfg::NI;
    // the wave index is uniform over a wave (the x extent of the workgroup is the wave size): told to the compiler, everything
    // indexed by it (row pointers, LDS slots of this wave's instances) is scalar arithmetic instead of per-lane address registers
    const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y), N = P.N;
    const bool worker = w == 0;
    const bool wl = worker && lane < NI;          // worker lane with an instance
    MPC_STAMP_INIT
    __syncthreads();                              // sh.q / sh.iflag are ready

    // ---- element-wise role: instances w*IPW + j, block k = lane -----------------------------------------------------
    const int k = lane;
    const bool blk_on = k < N, last = k == N - 1;
    struct Inst {          // what stays in registers between phases: wave-uniform data only
        double qz0[NS], qzr[NS], qur[NU];
        double mu, mu_sum, sm, inv_ncon, gscale;
        int stall;
        bool on, warm;
    };
    struct Iter { double sl[NC], sh[NC], ll[NC], lh[NC], u[NU], z[NS], lo[NC], hi[NC]; bool fl[NC], fh[NC]; };      // one block: iterate, bounds
    // bounds of this lane's block (the last block has the terminal ones), rebuilt per phase from the instance data in LDS
    auto bounds = [&](int wi, Iter &X) {
        const double *qd = sh.q + wi * Cfg::QN;
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            const double zlm = i >= NU ? qd[3 * NS + (i >= NU ? i - NU : 0)] : 0.0, zhm = i >= NU ? qd[4 * NS + (i >= NU ? i - NU : 0)] : 0.0;
            const double lm = i < NU ? P.ulo[i < NU ? i : 0] : zlm, hm = i < NU ? P.uhi[i < NU ? i : 0] : zhm;
            const double le = i < NU ? P.ulo[i < NU ? i : 0] : P.zlo_e[i >= NU ? i - NU : 0], he = i < NU ? P.uhi[i < NU ? i : 0] : P.zhi_e[i >= NU ? i - NU : 0];
            const double lo = last ? le : lm, hi = last ? he : hm;
            X.fl[i] = MASKED ? fin(lo) : true; X.fh[i] = MASKED ? fin(hi) : true;
            X.lo[i] = X.fl[i] ? lo : 0.0; X.hi[i] = X.fh[i] ? hi : 0.0;
        }
    };
    Inst I[IPW];
    // A row access is (wave-uniform row pointer)[lane]: global_load / store with a scalar base and one shared 32-bit vector offset.
    // Written as one 64-bit per-lane address the compiler hoists all ~150 of them into the kernel prologue and spills them.
    unsigned ku = (unsigned)k;      // made opaque at every phase boundary (fresh()), so that the addresses are built where they are used
    unsigned lq = (unsigned)lane;   // the same for wave 0's lane = instance view
    auto fresh = [&]() { asm volatile("" : "+v"(ku), "+v"(lq)); };
    fresh();                        // also for the start-up below: this function is inlined into the caller's loop over the MPC steps
    auto rowp = [&](int wi, int r) -> double * { return wsg + ((size_t)wi * Cfg::ROWS_ST + r) * 64; };
    auto load_iter = [&](int wi, Iter &X) {
        MPC_UNROLL for (int i = 0; i < NC; i++) { X.sl[i] = rowp(wi, Cfg::ST_SL + i)[ku]; X.sh[i] = rowp(wi, Cfg::ST_SH + i)[ku]; X.ll[i] = rowp(wi, Cfg::ST_LL + i)[ku]; X.lh[i] = rowp(wi, Cfg::ST_LH + i)[ku]; }
        MPC_UNROLL for (int i = 0; i < NU; i++) X.u[i] = rowp(wi, Cfg::ST_U + i)[ku];
        MPC_UNROLL for (int i = 0; i < NS; i++) X.z[i] = rowp(wi, Cfg::ST_Z + i)[ku];
        bounds(wi, X);
    };
    auto store_iter = [&](int wi, const Iter &X) {
        MPC_UNROLL for (int i = 0; i < NC; i++) { rowp(wi, Cfg::ST_SL + i)[ku] = X.sl[i]; rowp(wi, Cfg::ST_SH + i)[ku] = X.sh[i]; rowp(wi, Cfg::ST_LL + i)[ku] = X.ll[i]; rowp(wi, Cfg::ST_LH + i)[ku] = X.lh[i]; }
        MPC_UNROLL for (int i = 0; i < NU; i++) rowp(wi, Cfg::ST_U + i)[ku] = X.u[i];
        MPC_UNROLL for (int i = 0; i < NS; i++) rowp(wi, Cfg::ST_Z + i)[ku] = X.z[i];
    };
    // stage cost of this lane's block in registers: Q (or the terminal weight for the last block), R, M
    double Qk[NS][NS], Rk[NU][NU], Mk[NS][NU];
    MPC_UNROLL for (int i = 0; i < NS; i++) {
        MPC_UNROLL for (int j = 0; j <= i; j++) { const double t = vreg(last ? P.Pf[i][j] : P.Q[i][j]); Qk[i][j] = t; Qk[j][i] = t; }
        MPC_UNROLL for (int j = 0; j < NU; j++) Mk[i][j] = HASM ? vreg(P.M[i][j]) : 0.0;
    }
    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j <= i; j++) { const double t = vreg(P.R[i][j]); Rk[i][j] = t; Rk[j][i] = t; } }
    // cost gradient of the current point for this block: gu (NU), gz (NS), with the bound multipliers
    auto gradient = [&](const Inst &S, const Iter &X, double (&gu)[NU], double (&gz)[NS]) {
        double dz1[NS], du[NU];
        MPC_UNROLL for (int i = 0; i < NS; i++) dz1[i] = X.z[i] - S.qzr[i];
        MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = X.u[i] - S.qur[i];
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            double a = (NU + i < NC) ? X.lh[NU + i < NC ? NU + i : 0] - X.ll[NU + i < NC ? NU + i : 0] : 0.0;
            MPC_UNROLL for (int j = 0; j < NS; j++) a += Qk[i][j] * dz1[j];
            gz[i] = a;
        }
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            double a = X.lh[i] - X.ll[i];
            MPC_UNROLL for (int j = 0; j < NU; j++) a += Rk[i][j] * du[j];
            gu[i] = a;
        }
        if (HASM) {     // cross terms of the Delta-u form: M (u_{k+1} - ur) into gz (k < N-1), M'(z_k - zr) into gu
            double un[NU], zp[NS];
            MPC_UNROLL for (int i = 0; i < NU; i++) un[i] = __shfl_down(du[i], 1, 64);
            MPC_UNROLL for (int i = 0; i < NS; i++) { const double t = __shfl_up(dz1[i], 1, 64); zp[i] = k > 0 ? t : S.qz0[i] - S.qzr[i]; }
            if (!last) { MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) gz[i] += Mk[i][j] * un[j]; } }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) gu[i] += Mk[j][i] * zp[j]; }
        }
    };
    // Residuals, barrier weights, gradients of the iterate X -> LDS; then the convergence test of this iterate: the
    // stationarity residual needs the costates pi_k = gz_k + A' pi_{k+1}, a linear recursion with a constant matrix,
    // taken here as a parallel scan over the lanes (log2(64) steps with A^(2^j)) instead of a sequential sweep.
    auto phase_a = [&](Inst &S, const Iter &X, int wi, int it) {
        double mu_p = 0.0, resp_p = 0.0, cres_p = 0.0, lmax_p = 0.0, hb[NC];
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            const double v = i < NU ? X.u[i < NU ? i : 0] : X.z[i >= NU ? i - NU : 0];
            const double rh = X.fh[i] ? v + X.sh[i] - X.hi[i] : 0.0, rl = X.fl[i] ? v - X.sl[i] - X.lo[i] : 0.0;
            const double isl = frcp(X.sl[i]), ish = frcp(X.sh[i]);
            mu_p += X.sl[i] * X.ll[i] + X.sh[i] * X.lh[i];
            sh.t(RA + i, wi, ku) = X.ll[i] * isl + X.lh[i] * ish;
            hb[i] = X.lh[i] * (rh * ish - 1.0) + X.ll[i] * (rl * isl + 1.0);
            resp_p = dmax(resp_p, dmax(fabs(rl), fabs(rh)));
            cres_p = dmax(cres_p, dmax(comp_measure(X.sl[i], X.ll[i]), comp_measure(X.sh[i], X.lh[i])));
            lmax_p = dmax(lmax_p, dmax(X.ll[i], X.lh[i]));
        }
        double gu[NU], gz[NS], pi[NS];
        gradient(S, X, gu, gz);
        // right-hand side of the predictor: gradient + barrier term of the bounds on this variable
        MPC_UNROLL for (int i = 0; i < NU; i++) sh.t(RG + i, wi, ku) = gu[i] + hb[i];
        MPC_UNROLL for (int i = 0; i < NS; i++) { sh.t(RG + NU + i, wi, ku) = gz[i] + (NU + i < NC ? hb[NU + i < NC ? NU + i : 0] : 0.0); pi[i] = blk_on ? gz[i] : 0.0; }
        MPC_UNROLL for (int e = 0; e < 6; e++) {
            const int d = 1 << e;
            if (d < N) {
                double t[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) { const double v = __shfl_down(pi[i], d, 64); t[i] = (k + d < N) ? v : 0.0; }
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = pi[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += P.Apow[e][j][i] * t[j]; pi[i] = a; }
            }
        }
        double rs_p = 0.0;
        MPC_UNROLL for (int i = 0; i < NU; i++) { double a = gu[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += P.B[j][i] * pi[j]; rs_p = dmax(rs_p, fabs(a)); }
        S.mu_sum = wave_sum(blk_on ? mu_p : 0.0);
        const double res_p = wave_max(blk_on ? resp_p : 0.0), cres = wave_max(blk_on ? cres_p : 0.0), lmax = wave_max(blk_on ? lmax_p : 0.0);
        const double res_s = wave_max(blk_on ? rs_p : 0.0);
        S.mu = S.mu_sum * S.inv_ncon;
        if (it == 0) S.gscale = dmax(1.0, P.term_cons ? dmin(res_s, P.term_gcap) : res_s);      // mpc_device.hpp:rpdip_lane
        const bool ok_cp = (cres <= 1.0) && (res_p <= kTolFeas);
        S.stall = ok_cp ? S.stall + 1 : 0;
        int verdict = 0;
        if (ok_cp && (res_s <= kTolStat * S.gscale + P.term_floor || (S.stall > kStallMax && res_s <= kTolStatAcc * S.gscale + P.term_floor))) verdict = 1 + kSolved;
        else if (lmax > kInfeasZ * S.gscale || !(fabs(S.mu) < 1.0e300)) verdict = 1 + kInfeasible;
        else if (it == max_iter) verdict = 1 + kMaxIter;
        if (verdict != 0) S.on = false;
        if (lane == 0) { sh.flag[wi] = verdict; if (verdict != 0) sh.iters[wi] = it; }
    };

    // ---- instance constants; initial inputs (cold: us pushed inside the box; warm: previous inputs shifted one stage)
    double ll0[IPW][NC], lh0[IPW][NC], u0v[IPW][NU];
    MPC_UNROLL for (int j = 0; j < IPW; j++) {
        Inst &S = I[j];
        const int wi = w * IPW + j;
        const double *qd = sh.q + wi * Cfg::QN;
        const int myflag = sh.iflag[wi];
        S.on = (myflag & kTpValid) && (myflag & kTpOk0);
        S.warm = (myflag & kTpWarm) != 0;
        S.mu = 0.0; S.mu_sum = 0.0; S.sm = 0.0; S.gscale = 1.0; S.stall = 0;
        if (!S.on && lane == 0) { sh.flag[wi] = 1 + ((myflag & kTpValid) && !(myflag & kTpOk0) ? kInfeasible : kMaxIter); sh.iters[wi] = 0; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { S.qz0[i] = uni(qd[i]); S.qzr[i] = uni(qd[NS + i]); }
        MPC_UNROLL for (int i = 0; i < NU; i++) S.qur[i] = uni(qd[5 * NS + i]);
        double ncon = 0.0;
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            const double zlm = i >= NU ? uni(qd[3 * NS + (i >= NU ? i - NU : 0)]) : 0.0, zhm = i >= NU ? uni(qd[4 * NS + (i >= NU ? i - NU : 0)]) : 0.0;
            const double lm = i < NU ? P.ulo[i < NU ? i : 0] : zlm, hm = i < NU ? P.uhi[i < NU ? i : 0] : zhm;
            const double le = i < NU ? P.ulo[i < NU ? i : 0] : P.zlo_e[i >= NU ? i - NU : 0], he = i < NU ? P.uhi[i < NU ? i : 0] : P.zhi_e[i >= NU ? i - NU : 0];
            const bool flm = MASKED ? fin(lm) : true, fhm = MASKED ? fin(hm) : true, fle = MASKED ? fin(le) : true, fhe = MASKED ? fin(he) : true;
            ncon += (double)(N - 1) * ((flm ? 1 : 0) + (fhm ? 1 : 0)) + (fle ? 1 : 0) + (fhe ? 1 : 0);
        }
        S.inv_ncon = 1.0 / dmax(ncon, 1.0);
        const int ksrc = k + 1 < N ? k + 1 : (k < N ? N - 1 : k);     // shift by one stage, the last block repeats
        const int sft = ksrc - k;
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            const double ulo = P.ulo[i], uhi = P.uhi[i];
            const bool f_lo = fin(ulo), f_hi = fin(uhi);
            double v;
            if (S.warm) {
                v = rowp(wi, Cfg::ST_U + i)[ku + sft];
                if (f_lo) v = dmax(v, ulo);
                if (f_hi) v = dmin(v, uhi);
            } else {
                const double us = uni(qd[5 * NS + NU + i]);
                double push;
                if (f_lo && f_hi) push = 0.1 * (uhi - ulo);
                else push = 0.1 * dmax(1.0, fabs(f_lo ? ulo : (f_hi ? uhi : 0.0)));
                v = us;
                if (f_lo) v = dmax(v, ulo + push);
                if (f_hi) v = dmin(v, uhi - push);
            }
            u0v[j][i] = v;
        }
        MPC_UNROLL for (int i = 0; i < NC; i++) {
            ll0[j][i] = S.warm ? rowp(wi, Cfg::ST_LL + i)[ku + sft] : 0.0;
            lh0[j][i] = S.warm ? rowp(wi, Cfg::ST_LH + i)[ku + sft] : 0.0;
        }
    }
    // slacks and multipliers of the initial point, then the first element-wise phase
    MPC_UNROLL for (int j = 0; j < IPW; j++) {
        Inst &S = I[j];
        const int wi = w * IPW + j;
        if (S.on) {
            Iter Xj;
            MPC_UNROLL for (int i = 0; i < NU; i++) Xj.u[i] = u0v[j][i];
            // states of the initial point by forward simulation z_{k+1} = A z_k + B u_k + c: again a linear recursion with a
            // constant matrix, taken as a scan over the lanes with A^(2^e) (lane k ends up with z_{k+1})
            {
                const double *qd = sh.q + wi * Cfg::QN;
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
            bounds(wi, Xj);
            const double ws_delta = uni(sh.q[wi * Cfg::QN + 5 * NS + 2 * NU]);
            const double ws_smin = dmin(dmax(kWsKappa * ws_delta, kWsSMinLo), kWsSMinHi), ws_mu = kWsMuFactor * ws_smin * ws_smin;
            const double smin = S.warm ? ws_smin : kSMin;
            MPC_UNROLL for (int i = 0; i < NC; i++) {
                const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                Xj.sl[i] = Xj.fl[i] ? dmax(v - Xj.lo[i], smin) : 1.0; Xj.sh[i] = Xj.fh[i] ? dmax(Xj.hi[i] - v, smin) : 1.0;
                const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                const double llo = S.warm ? dmax(ll0[j][i], ws_mu * isl) : kMu0 * isl, lhi = S.warm ? dmax(lh0[j][i], ws_mu * ish) : kMu0 * ish;
                Xj.ll[i] = Xj.fl[i] ? llo : 0.0; Xj.lh[i] = Xj.fh[i] ? lhi : 0.0;
            }
            store_iter(wi, Xj);
            phase_a(S, Xj, wi, 0);
        }
    }

    // ---- the two recursion waves (lane = instance) ----------------------------------------------------------------------
    // Wave 0 runs the matrix recursion (Riccati: sigma -> K, Lambda^-1), wave 1 everything that is a vector recursion: the
    // right-hand side (backward; for the predictor it follows wave 0 block by block through a progress word in LDS) and the
    // Newton direction (forward).
    // Matrix recursion on the matrix cores when the stage fits a 4x4 tile (v_mfma_f64_4x4x4f64: four independent 4x4x4 products
    // per wave): then one wave serves four instances, NI / 4 waves factorise at once; otherwise wave 0 with lane = instance.
    constexpr bool MFMA = NS <= 4 && NU <= 2 && NI % 4 == 0 && NI / 4 < NW;
    constexpr int NMW = MFMA ? NI / 4 : 1;      // matrix waves: 0 .. NMW-1; the vector wave comes next
    const int mw = w;                                  // index of this wave among the matrix waves
    const bool matw = w < NMW, vecw = w == NMW;        // the vector wave comes right after the matrix wave(s)
    double Av[NS][NS], Bv[NS][NU];      // wave 1: model matrices in registers
    if (vecw) {
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            MPC_UNROLL for (int j = 0; j < NS; j++) Av[i][j] = vreg(P.A[i][j]);
            MPC_UNROLL for (int j = 0; j < NU; j++) Bv[i][j] = vreg(P.B[i][j]);
        }
    }
    int *const progress = sh.misc;      // per matrix wave: the block its recursion has finished last (N: none yet)
    if (worker && lane < NMW) progress[lane] = N;
    // backward recursion of the right-hand side: h (RG rows), K, Lambda^-1 -> kff; `follow`: K and Lambda^-1 of a block exist
    // only once wave 0 has announced it
    auto rhs_pass = [&](bool follow) {
        int done = N;       // every matrix wave is known to have finished this block (and the ones after it)
        auto wait_for = [&](int blk) {
#ifdef MPC_NOFOLLOW
            blk = 0;      // diagnostic build: the vector wave starts only when the whole factorisation is in LDS
#endif
            if (follow) {
                while (done > blk) {      // look again only when the known state does not cover the block
                    int d = 0;
                    MPC_UNROLL for (int m = 0; m < NMW; m++) { const int v = __hip_atomic_load(progress + m, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); d = v > d ? v : d; }
                    done = d;
                    if (done > blk) __builtin_amdgcn_s_sleep(1);
                }
            }
        };
        double pc[NS], hn[NV], Kn[NKF], ln[NLI];
        MPC_UNROLL for (int i = 0; i < NS; i++) pc[i] = 0.0;
        wait_for(N - 1);
        MPC_UNROLL for (int i = 0; i < NV; i++) hn[i] = sh.t(RG + i, lq, N - 1);
        MPC_UNROLL for (int i = 0; i < NKF; i++) Kn[i] = sh.t(RA + i, lq, N - 1);
        MPC_UNROLL for (int i = 0; i < NLI; i++) ln[i] = sh.t(RA + NKF + i, lq, N - 1);
        _Pragma("unroll 2") for (int kk = N - 1; kk >= 0; kk--) {
            double pv[NS], hu[NU], Li[NU][NU], Kf[NKF], psi[NU], kff[NU];
            MPC_UNROLL for (int i = 0; i < NU; i++) hu[i] = hn[i];
            MPC_UNROLL for (int i = 0; i < NS; i++) pv[i] = hn[NU + i] + pc[i];
            MPC_UNROLL for (int i = 0; i < NKF; i++) Kf[i] = Kn[i];
            {
                int c = 0;
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j <= i; j++) { Li[i][j] = ln[c]; Li[j][i] = ln[c]; c++; } }
            }
            const int kx = kk > 0 ? kk - 1 : 0;      // the next block's data now, they arrive while this block computes
            wait_for(kx);
            MPC_UNROLL for (int i = 0; i < NV; i++) hn[i] = sh.t(RG + i, lq, kx);
            MPC_UNROLL for (int i = 0; i < NKF; i++) Kn[i] = sh.t(RA + i, lq, kx);
            MPC_UNROLL for (int i = 0; i < NLI; i++) ln[i] = sh.t(RA + NKF + i, lq, kx);
            MPC_UNROLL for (int i = 0; i < NU; i++) { double a = hu[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += Bv[j][i] * pv[j]; psi[i] = a; }
            MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) a += Li[i][j] * psi[j]; kff[i] = -a; }
            MPC_UNROLL for (int i = 0; i < NU; i++) sh.t(RK + i, lq, kk) = kff[i];
            double pn[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                double a = 0.0;
                MPC_UNROLL for (int j = 0; j < NS; j++) a += Av[j][i] * pv[j];
                MPC_UNROLL for (int j = 0; j < NU; j++) a += Kf[j * NS + i] * psi[j];
                pn[i] = a;
            }
            MPC_UNROLL for (int i = 0; i < NS; i++) pc[i] = pn[i];
        }
    };
    // forward recursion of the Newton direction: K, kff -> du | dz
    auto direction = [&]() {
        double dz[NS], Kn[NKF], kn[NU];
        MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = 0.0;
        MPC_UNROLL for (int i = 0; i < NKF; i++) Kn[i] = sh.t(RA + i, lq, 0);
        MPC_UNROLL for (int i = 0; i < NU; i++) kn[i] = sh.t(RK + i, lq, 0);
        _Pragma("unroll 2") for (int kk = 0; kk < N; kk++) {      // two blocks per trip: the prefetched gains rotate without register copies
            double Kf[NKF], kff[NU], ddu[NU], dzn[NS];
            MPC_UNROLL for (int i = 0; i < NKF; i++) Kf[i] = Kn[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) kff[i] = kn[i];
            const int kx = kk + 1 < N ? kk + 1 : kk;      // the next block's gains now, they arrive while this block computes
            MPC_UNROLL for (int i = 0; i < NKF; i++) Kn[i] = sh.t(RA + i, lq, kx);
            MPC_UNROLL for (int i = 0; i < NU; i++) kn[i] = sh.t(RK + i, lq, kx);
            MPC_UNROLL for (int i = 0; i < NU; i++) { double a = kff[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += Kf[i * NS + j] * dz[j]; ddu[i] = a; }
            MPC_UNROLL for (int i = 0; i < NS; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) a += Av[i][j] * dz[j]; MPC_UNROLL for (int j = 0; j < NU; j++) a += Bv[i][j] * ddu[j]; dzn[i] = a; }
            MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = dzn[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) sh.t(RG + i, lq, kk) = ddu[i];
            MPC_UNROLL for (int i = 0; i < NS; i++) sh.t(RG + NU + i, lq, kk) = dz[i];
        }
    };

    // ---- the matrix recursion on the matrix cores (MFMA variant) --------------------------------------------------------------
    // v_mfma_f64_4x4x4f64 multiplies four independent pairs of 4x4 tiles per wave: lane 16 r + 4 b + c holds element (r, c) of
    // tile b; with M, S in this layout the instruction gives M'S + C (its first operand is read transposed).  Tile b of wave w
    // belongs to instance 4 w + b; matrices are zero-padded to 4x4.  Barrier weights, K and Lambda^-1 go through the same LDS
    // rows as in the lane = instance variant, one masked access per tile.
    const int tr = lane >> 4, tb = (lane >> 2) & 3, tc = lane & 3, tinst = (MFMA && matw) ? 4 * mw + tb : 0;
    const int tfl = sh.iflag[tinst];
    const bool t_valid = MFMA && matw && (tfl & kTpValid) && (tfl & kTpOk0);
    const bool in_ss = tr < NS && tc < NS, in_su = tr < NS && tc < NU, in_us = tr < NU && tc < NS, in_uu = tr < NU && tc < NU;
    auto mm = [](double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); };      // a' b + c on tiles
    bool pd_all = true;         // every Lambda of this lane's instance was positive definite so far
    // backward: Riccati factorisation, sigma -> K, Lambda^-1 in LDS; each finished block is announced to the vector wave
    auto tile_factor = [&](bool t_on) {
        const double Ar = in_ss ? P.A[tr][tc] : 0.0, Br = in_su ? P.B[tr][tc] : 0.0, Btr = in_us ? P.B[tc][tr] : 0.0;
        const double Rr = in_uu ? P.R[tr][tc] : 0.0, Qr = in_ss ? P.Q[tr][tc] : 0.0, Mtr = (HASM && in_us) ? P.M[tc][tr] : 0.0;
        // rows 0 / 1 of Lambda = R~ + B'PB broadcast over all tile rows, straight from PB: (B E_i)' PB + E_i' R~
        const double BE0 = tr < NS ? P.B[tr][0] : 0.0, BE1 = (NU > 1 && tr < NS) ? P.B[tr][NU > 1 ? 1 : 0] : 0.0;
        const double RE0 = tc < NU ? P.R[0][tc] : 0.0, RE1 = (NU > 1 && tc < NU) ? P.R[NU > 1 ? 1 : 0][tc] : 0.0;
        double Pm = in_ss ? P.Pf[tr][tc] : 0.0;
        // barrier weights: sigma_z[r] on the diagonal of P, sigma_u[r] on the diagonal of R~, sigma_u[c] in column c of the broadcast
        // rows.  Every lane reads some row of its instance (row RA where it has no use for one) and multiplies by its 0 / 1 weight.
        const bool dz_on = tr == tc && tr < NS && NU + tr < NC, du_on = tr == tc && tr < NU, db_on = tc < NU;
        unsigned tq = (unsigned)tinst, tl = (unsigned)lane;
        asm volatile("" : "+v"(tq));
        const int rz = dz_on ? RA + NU + tr : RA, ru = du_on ? RA + tr : RA, rb = db_on ? RA + tc : RA;
        double szn = sh.t(rz, tq, N - 1), sun = sh.t(ru, tq, N - 1), sbn = sh.t(rb, tq, N - 1);
        // K goes to LDS from the lanes that hold it (rows < NU of the tile), Lambda^-1 from rows 2..3, which compute it as well:
        // one store per block.  (r', c) = (r - 2, c) of the lower triangle -> row RA + NKF + r'(r' + 1)/2 + c
        const bool st_k = in_us, st_l = tr >= 2 && tc < NU && tc <= tr - 2 && tr - 2 < NU;
        const int st_row = st_k ? RA + tr * NS + tc : RA + NKF + (st_l ? (tr - 2) * (tr - 1) / 2 + tc : 0);
        const bool st_on = t_on && (st_k || st_l);
        bool pd_ok = true;
#ifdef MPC_STAMPS_FINE
        unsigned long long seg_a = 0, seg_b = 0, seg_c = 0, tprev = __builtin_amdgcn_s_memtime();      // cycle sums kept in registers
#define MPC_SEG(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - tprev; tprev = t_; } while (0)
#else
#define MPC_SEG(acc) do { } while (0)
#endif
        for (int kk = N - 1; kk >= 0; kk--) {
            // the identity tile rebuilt from the lane index on every trip (as a loop invariant it ends up in scratch, and a reload is a
            // memory round trip on the critical path)
            asm volatile("" : "+v"(tl));
            const unsigned lr = tl >> 4, lc = tl & 3;
            const double Ir = lr == lc ? 1.0 : 0.0;
            // 0 / 1 weights of this lane, rebuilt like the identity
            const double mz = (lr == lc && lr < (unsigned)(NS < NC - NU ? NS : NC - NU)) ? 1.0 : 0.0, mu_ = (lr == lc && lr < (unsigned)NU) ? 1.0 : 0.0;
            const double mb0 = lc == 0 ? 1.0 : 0.0, mb1 = (NU > 1 && lc == 1) ? 1.0 : 0.0;
            const double sz = szn, su = sun, sb = sbn;
            const int kn = kk > 0 ? kk - 1 : 0;      // next block's weights now
            szn = sh.t(rz, tq, kn); sun = sh.t(ru, tq, kn); sbn = sh.t(rb, tq, kn);
            Pm = __builtin_fma(sz, mz, Pm);
            // P A, P B, B'P (P symmetric).  The products are ordered for a short dependency chain: a matrix-core product whose operand
            // is the previous one's result waits ~60 cycles for it, and the chain of a block is what the recursion costs.
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
                MPC_SEG(seg_a);      // weights arrived, first two product levels done (a is read by the vector pipe)
                const double det = a * d - off * off, rdet = frcp(det);
                pd_ok = pd_ok && (a > 0.0) && (det > 0.0);
                Lall = tc < 2 ? ((tr & 1) == tc ? (tc == 0 ? d : a) * rdet : -off * rdet) : 0.0;
            }
            const double Li = tr < NU ? Lall : 0.0;
            const double Kk = mm(-Li, Psi, 0.0);      // K = -Lambda^-1 Psi
            if (st_on) sh.t(st_row, tq, kk) = st_k ? Kk : Lall;
            MPC_SEG(seg_b);          // inverse, K, store (the stamp waits for the store)
            if (lane == 0) __hip_atomic_store(progress + mw, kk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // block kk is in LDS
            if (kk > 0) {       // closed-loop (Joseph) form: Q + Acl' P Acl + K' R~ K (+ M K + K' M')
                const double Acl = mm(Btr, Kk, Ar), RK_ = mm(Rs, Kk, 0.0), T = mm(BtP, Kk, PA);      // A + B K,  R~ K,  P Acl = PA + PB K
                double Pn = mm(Acl, T, mm(Kk, RK_, Qr));
                if (HASM) { const double MK = mm(Mtr, Kk, 0.0); Pn = mm(MK, Ir, Pn + MK); }
                Pm = Pn;      // symmetric up to rounding; the recursion does not amplify the difference
            }
#ifdef MPC_STAMPS_FINE
            Pm = vreg(Pm);           // the update has to be complete when the stamp is taken
#endif
            MPC_SEG(seg_c);          // flag, Joseph update
        }
#ifdef MPC_STAMPS_FINE
        if (threadIdx.x == 0 && threadIdx.y == 0) { mpc_stamp_buf[(blockIdx.x & 4095) * 8 + 5] += seg_a; mpc_stamp_buf[(blockIdx.x & 4095) * 8 + 6] += seg_b; mpc_stamp_buf[(blockIdx.x & 4095) * 8 + 2] += seg_c; }
#endif
        pd_all = pd_all && pd_ok;
        if (t_on && !pd_all && tr == 0 && tc == 0) sh.flag[tinst] = -1;      // a Lambda lost definiteness: the instance stops as infeasible
    };

    const int rfl = sh.iflag[lane < NI ? lane : 0];
    const bool wk_valid = ((!MFMA && worker) || vecw) && lane < NI && (rfl & kTpValid) && (rfl & kTpOk0);

    MPC_TSTAMP(1); fresh();
    for (int it = 0;; it++) {
        bool mine = false;
        MPC_UNROLL for (int j = 0; j < IPW; j++) mine = mine || I[j].on;
        if (!__syncthreads_or(mine ? 1 : 0)) break;      // every instance of the workgroup has its verdict
        MPC_TSTAMP(2); fresh();
        const bool wk_on = wk_valid && sh.flag[lane < NI ? lane : 0] == 0;
        // ================= matrix wave(s): Riccati factorisation; the vector wave behind: predictor rhs, then the direction ===
        if (MFMA && matw) {
            const bool t_on = t_valid && sh.flag[tinst] == 0;
            if (__any(t_on ? 1 : 0)) tile_factor(t_on);
            else if (lane == 0) __hip_atomic_store(progress + mw, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // nobody waits for this wave
            MPC_TSTAMP_FINE(4);      // fine stamps: slot 4 = the factorisation alone, slot 3 = the rest of the iteration
        } else if (!MFMA && worker) {
            StageConst<NS, NU> C;
            load_stage_const<NS, NU, HASM>(P, C);
            if (wk_on) {
                double Pm[NS][NS];
                bool pd_ok = true;
                MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Pm[i][j] = C.Pf[i][j]; }
                double sg[NC];
                MPC_UNROLL for (int i = 0; i < NC; i++) sg[i] = sh.t(RA + i, lq, N - 1);
                _Pragma("unroll 2") for (int kk = N - 1; kk >= 0; kk--) {
                    double sig[NV];
                    MPC_UNROLL for (int i = 0; i < NV; i++) sig[i] = i < NC ? sg[i < NC ? i : 0] : 0.0;
                    const int kn = kk > 0 ? kk - 1 : 0;      // next block's data now, they arrive while this block computes
                    MPC_UNROLL for (int i = 0; i < NC; i++) sg[i] = sh.t(RA + i, lq, kn);
                    MPC_UNROLL for (int i = 0; i < NS; i++) Pm[i][i] += sig[NU + i];
                    double PB[NS][NU], PA[NS][NS], Lam[NU][NU], Psi[NU][NS];
                    MPC_UNROLL for (int i = 0; i < NS; i++) {
                        MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pm[i][l] * C.B[l][j]; PB[i][j] = a; }
                        MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pm[i][l] * C.A[l][j]; PA[i][j] = a; }
                    }
                    MPC_UNROLL for (int i = 0; i < NU; i++) {
                        MPC_UNROLL for (int j = 0; j <= i; j++) { double a = C.R[i][j] + (i == j ? sig[i] : 0.0); MPC_UNROLL for (int l = 0; l < NS; l++) a += C.B[l][i] * PB[l][j]; Lam[i][j] = a; Lam[j][i] = a; }
                        MPC_UNROLL for (int j = 0; j < NS; j++) { double a = HASM ? C.M[j][i] : 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += C.B[l][i] * PA[l][j]; Psi[i][j] = a; }
                    }
                    pd_ok = sym_inverse<NU>(Lam) && pd_ok;
                    double Kk[NU][NS];
                    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NU; l++) a += Lam[i][l] * Psi[l][j]; Kk[i][j] = -a; } }
                    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) sh.t(RA + i * NS + j, lq, kk) = Kk[i][j]; }
                    {
                        int c = 0;
                        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j <= i; j++) { sh.t(RA + NKF + c, lq, kk) = Lam[i][j]; c++; } }
                    }
                    __hip_atomic_store(progress, kk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // block kk is in LDS
                    if (kk > 0) {       // closed-loop (Joseph) form with T = P Acl = PA + PB K
                        double Acl[NS][NS], T[NS][NS], RK_[NU][NS];
                        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = C.A[i][j], t = PA[i][j]; MPC_UNROLL for (int l = 0; l < NU; l++) { a += C.B[i][l] * Kk[l][j]; t += PB[i][l] * Kk[l][j]; } Acl[i][j] = a; T[i][j] = t; } }
                        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = sig[i] * Kk[i][j]; MPC_UNROLL for (int l = 0; l < NU; l++) a += C.R[i][l] * Kk[l][j]; RK_[i][j] = a; } }
                        MPC_UNROLL for (int i = 0; i < NS; i++) {
                            MPC_UNROLL for (int j = 0; j <= i; j++) {
                                double a = C.Q[i][j];
                                MPC_UNROLL for (int l = 0; l < NS; l++) a += Acl[l][i] * T[l][j];
                                MPC_UNROLL for (int l = 0; l < NU; l++) a += Kk[l][i] * RK_[l][j];
                                if (HASM) { MPC_UNROLL for (int l = 0; l < NU; l++) a += C.M[i][l] * Kk[l][j] + Kk[l][i] * C.M[j][l]; }
                                Pm[i][j] = a; Pm[j][i] = a;
                            }
                        }
                    }
                }
                pd_all = pd_all && pd_ok;
            }
            MPC_TSTAMP_FINE(4);      // fine stamps: slot 4 = the matrix recursion alone, slot 3 = the rest of the iteration
            if (wl && wk_on && !pd_all) sh.flag[lane] = -1;      // a Lambda lost definiteness: the instance stops as infeasible
        } else if (vecw) {
            if (wk_on) { rhs_pass(true); direction(); }
        }
        __syncthreads();
        MPC_TSTAMP_IT(3); fresh();
        if (worker && lane < NMW) progress[lane] = N;      // for the next iteration's factorisation
        MPC_UNROLL for (int j = 0; j < IPW; j++) {
            const int wi = w * IPW + j;
            if (I[j].on && sh.flag[wi] < 0) {       // the factorisation failed
                I[j].on = false;
                if (lane == 0) { sh.flag[wi] = 1 + kInfeasible; sh.iters[wi] = it; }
            }
        }
        // ================= element-wise: predictor step length, centring, corrector rhs -> LDS ===========================
        {
            MPC_UNROLL for (int j = 0; j < IPW; j++) {
                Inst &S = I[j];
                const int wi = w * IPW + j;
                if (S.on) {
                    Iter Xj;
                    load_iter(wi, Xj);
                    double maff_p = 1.0, s1_p = 0.0, s2_p = 0.0, dv[NC], pl[NC], ph[NC];
                    MPC_UNROLL for (int i = 0; i < NC; i++) { dv[i] = sh.t(RG + i, wi, ku); rowp(wi, Cfg::ST_DV + i)[ku] = dv[i]; }      // du | dz of the predictor
                    MPC_UNROLL for (int i = 0; i < NC; i++) {
                        const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                        const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                        const double rh = Xj.fh[i] ? v + Xj.sh[i] - Xj.hi[i] : 0.0, rl = Xj.fl[i] ? v - Xj.sl[i] - Xj.lo[i] : 0.0;
                        const double dsh = Xj.fh[i] ? -rh - dv[i] : 0.0, dsl = Xj.fl[i] ? rl + dv[i] : 0.0;
                        const double qh = dsh * ish, ql = dsl * isl;
                        const double dlh = Xj.fh[i] ? -Xj.lh[i] - Xj.lh[i] * qh : 0.0, dll = Xj.fl[i] ? -Xj.ll[i] - Xj.ll[i] * ql : 0.0;
                        maff_p = dmax(maff_p, dmax(-ql, -qh));
                        if (Xj.fl[i]) maff_p = dmax(maff_p, 1.0 + ql);
                        if (Xj.fh[i]) maff_p = dmax(maff_p, 1.0 + qh);
                        s1_p += Xj.sl[i] * dll + Xj.ll[i] * dsl + Xj.sh[i] * dlh + Xj.lh[i] * dsh;
                        s2_p += dsl * dll + dsh * dlh;
                        pl[i] = dsl * dll; ph[i] = dsh * dlh;
                    }
                    const double m_aff = wave_max(blk_on ? maff_p : 1.0), s1 = wave_sum(blk_on ? s1_p : 0.0), s2 = wave_sum(blk_on ? s2_p : 0.0);
                    const double a_aff = frcp(m_aff);
                    const double mu_aff = (S.mu_sum + a_aff * s1 + a_aff * a_aff * s2) * S.inv_ncon;
                    const double rat = S.mu > 0.0 ? mu_aff * frcp(S.mu) : 0.0;
                    S.sm = dmax(rat * rat * rat * S.mu, kMuFloor);
                    double gu[NU], gz[NS], hc[NV];
                    gradient(S, Xj, gu, gz);
                    MPC_UNROLL for (int i = NC; i < NV; i++) hc[i] = 0.0;
                    MPC_UNROLL for (int i = 0; i < NC; i++) {
                        const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                        const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                        const double rh = Xj.fh[i] ? v + Xj.sh[i] - Xj.hi[i] : 0.0, rl = Xj.fl[i] ? v - Xj.sl[i] - Xj.lo[i] : 0.0;
                        const double rch = Xj.fh[i] ? Xj.sh[i] * Xj.lh[i] - dmax(S.sm, Xj.lh[i] * kSFloor) + ph[i] : 0.0;
                        const double rcl = Xj.fl[i] ? Xj.sl[i] * Xj.ll[i] - dmax(S.sm, Xj.ll[i] * kSFloor) + pl[i] : 0.0;
                        hc[i] = (-rch + Xj.lh[i] * rh) * ish + (rcl + Xj.ll[i] * rl) * isl;
                    }
                    MPC_UNROLL for (int i = 0; i < NU; i++) sh.t(RG + i, wi, ku) = gu[i] + hc[i];
                    MPC_UNROLL for (int i = 0; i < NS; i++) sh.t(RG + NU + i, wi, ku) = gz[i] + hc[NU + i];
                }
            }
        }
        __syncthreads();
        MPC_TSTAMP_IT(4); fresh();
        // ================= wave 1, lane = instance: corrector rhs recursion and direction ===============================
        if (vecw && wk_on) { rhs_pass(false); direction(); }
        __syncthreads();
        MPC_TSTAMP_IT(5); fresh();
        // ================= element-wise: corrector step length, step; then the next iterate's residuals / gradients =====
        {
            MPC_UNROLL for (int j = 0; j < IPW; j++) {
                Inst &S = I[j];
                const int wi = w * IPW + j;
                if (S.on) {
                    Iter Xj;
                    double dvaj[NC], dvzj[NV];
                    load_iter(wi, Xj);
                    MPC_UNROLL for (int i = 0; i < NC; i++) dvaj[i] = rowp(wi, Cfg::ST_DV + i)[ku];
                    MPC_UNROLL for (int i = 0; i < NV; i++) dvzj[i] = sh.t(RG + i, wi, ku);
                    double mcc_p = kTau;
                    double dsl[NC], dsh[NC], dll[NC], dlh[NC];
                    MPC_UNROLL for (int i = 0; i < NC; i++) {
                        const double v = i < NU ? Xj.u[i < NU ? i : 0] : Xj.z[i >= NU ? i - NU : 0];
                        const double isl = frcp(Xj.sl[i]), ish = frcp(Xj.sh[i]);
                        const double rh = Xj.fh[i] ? v + Xj.sh[i] - Xj.hi[i] : 0.0, rl = Xj.fl[i] ? v - Xj.sl[i] - Xj.lo[i] : 0.0;
                        // second-order products of the predictor direction (recomputed, not stored)
                        const double ash = Xj.fh[i] ? -rh - dvaj[i] : 0.0, asl = Xj.fl[i] ? rl + dvaj[i] : 0.0;
                        const double alh = Xj.fh[i] ? -Xj.lh[i] - Xj.lh[i] * (ash * ish) : 0.0, all_ = Xj.fl[i] ? -Xj.ll[i] - Xj.ll[i] * (asl * isl) : 0.0;
                        const double rch = Xj.fh[i] ? Xj.sh[i] * Xj.lh[i] - dmax(S.sm, Xj.lh[i] * kSFloor) + ash * alh : 0.0;
                        const double rcl = Xj.fl[i] ? Xj.sl[i] * Xj.ll[i] - dmax(S.sm, Xj.ll[i] * kSFloor) + asl * all_ : 0.0;
                        dsh[i] = Xj.fh[i] ? -rh - dvzj[i] : 0.0; dsl[i] = Xj.fl[i] ? rl + dvzj[i] : 0.0;
                        dlh[i] = Xj.fh[i] ? (-rch - Xj.lh[i] * dsh[i]) * ish : 0.0; dll[i] = Xj.fl[i] ? (-rcl - Xj.ll[i] * dsl[i]) * isl : 0.0;
                        mcc_p = dmax(mcc_p, dmax(-dsl[i] * isl, -dsh[i] * ish));
                        if (Xj.fl[i]) mcc_p = dmax(mcc_p, -dll[i] * frcp_approx(Xj.ll[i]));
                        if (Xj.fh[i]) mcc_p = dmax(mcc_p, -dlh[i] * frcp_approx(Xj.lh[i]));
                    }
                    const double m_cc = wave_max(blk_on ? mcc_p : kTau);
                    const double alpha = m_cc <= kTau ? 1.0 : kTau * frcp(m_cc);
                    MPC_UNROLL for (int i = 0; i < NC; i++) { Xj.sl[i] += alpha * dsl[i]; Xj.sh[i] += alpha * dsh[i]; Xj.ll[i] += alpha * dll[i]; Xj.lh[i] += alpha * dlh[i]; }
                    MPC_UNROLL for (int i = 0; i < NU; i++) Xj.u[i] += alpha * dvzj[i];
                    MPC_UNROLL for (int i = 0; i < NS; i++) Xj.z[i] += alpha * dvzj[NU + i];
                    store_iter(wi, Xj);
                    phase_a(S, Xj, wi, it + 1);
                }
            }
        }
        MPC_TSTAMP_IT(6); fresh();
    }
    // wave 0, lane i: the verdict of instance i
    __syncthreads();
    status_o = wl ? sh.flag[lane] - 1 : kMaxIter;
    iters_o = wl ? sh.iters[lane] : 0;
}

}  // namespace mpc
