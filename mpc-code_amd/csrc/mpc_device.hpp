// Device-side algorithm of libmpc_amd.so: one MPC instance per lane ("instance-per-lane" mapping).
//
// Everything here is templated on the problem dimensions so that every small-matrix loop is fully
// unrolled into fp64 FMAs on registers; the only memory traffic is the per-instance workspace of the
// OCP (wave-tiled: every access of a wave is one contiguous 1-KiB global_load/store_dwordx4) and the
// problem constants (wave-uniform: scalar cache, the hot ones pinned in VGPRs).
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
constexpr double kTolMu = 1e-14;      // ... or s*l <= kTolMu
constexpr double kMuFloor = 1e-15;    // centring target never below this
constexpr double kSFloor = 1e-11;     // ... nor below l*kSFloor
constexpr double kBoundRelax = 1e-8;  // relaxation of the stage-0 output rows
constexpr double kInfeasZ = 1e10;     // dual blow-up => infeasible
constexpr double kWsDelta = 0.3;     // closed-loop warm start: used when (xhat - prediction, dhat, xs, us) moved less than this
constexpr double kWsKappa = 1e-2;     // ... minimum slack = clip(kWsKappa * movement, kWsSMinLo, kWsSMinHi)
constexpr double kWsSMinLo = 1e-9, kWsSMinHi = 1e-6;
constexpr double kWsMuFactor = 1e4;   // ... minimum complementarity product = kWsMuFactor * (minimum slack)^2

constexpr int kMaxN = 8, kMaxM = 4, kMaxY = 8, kMaxD = 8, kMaxV = kMaxN + kMaxM, kMaxC = kMaxN + kMaxM + kMaxY,
              kMaxE = kMaxN + kMaxD;

enum : int { kSolved = 0, kMaxIter = 1, kInfeasible = 2 };

// Problem constants as the kernels read them (one copy in HBM, wave-uniform loads).
// PT in the device templates below: DevProblem, or DevProblem in the constant address space (ConstProblem) - then every access
// is a scalar load, whatever the kernel has stored to global memory in between
struct DevProblem {
    int nx, nu, ny, nd, nxp, N, du_form, duss_form, y_bounded, estimator, max_iter, has_dsat;
    int term_cons; double term_tol, term_gcap, term_floor;
    int no_warm;      // violently unstable open loop: a shifted warm start re-simulated from the new state is worse than a cold start (mpc_amd.hip:mpc_lin_create)      // terminal equality x_N = xs (Control_Calc.py:197-198): Pf carries the weight that enforces it, the residual above term_tol means unreachable
    int in_is_du, zr_us;      // stage input is v = u - u_prev (bounds on it exist); reference of the u_prev states is us (cost on u - us)
    // stage form (z = x, or [x; u_prev] when du_form)
    double A[kMaxN][kMaxN], B[kMaxN][kMaxM], Q[kMaxN][kMaxN], M[kMaxN][kMaxM], R[kMaxM][kMaxM], Pf[kMaxN][kMaxN];
    double ulo[kMaxM], uhi[kMaxM], zlo_m[kMaxN], zhi_m[kMaxN], zlo_e[kMaxN], zhi_e[kMaxN];
    // model / plant in the reference's terms
    double Am[kMaxN][kMaxN], Bm[kMaxN][kMaxM], Cm[kMaxY][kMaxN], Bd[kMaxN][kMaxD], Cd[kMaxY][kMaxD], fxc[kMaxN], fyc[kMaxY];
    double Ap[kMaxN][kMaxN], Bp[kMaxN][kMaxM], Cp[kMaxY][kMaxN];
    double ymin[kMaxY], ymax[kMaxY], dmin[kMaxD], dmax[kMaxD];
    int ymap_idx[kMaxY]; double ymap_scale[kMaxY];      // bounded output row i = ymap_scale * stage state ymap_idx (< 0: none)
    int ng, yg_row[kMaxY];                               // output rows carried as stage states of their own (general rows of C)
    // target problem in null-space coordinates
    double Ep[kMaxV][kMaxN], Zn[kMaxV][kMaxM], CZx[kMaxY][kMaxM], Hr[kMaxM][kMaxM], W[kMaxC][kMaxM], tlo[kMaxC], thi[kMaxC];
    double Qss[kMaxY][kMaxY], Rss[kMaxM][kMaxM];
    // estimator
    double Aa[kMaxE][kMaxE], Ca[kMaxY][kMaxE], Qkf[kMaxE][kMaxE], Rkf[kMaxY][kMaxY], Kfix[kMaxE][kMaxY];
    // A^(2^j), j = 0..5, of the stage form: the adjoint recursion as a parallel scan over the horizon (mpc_tp.hpp)
    double Apow[6][kMaxN][kMaxN];
    // soft output constraints (`slacks`, Control_Calc.py:39-40,186-192,228-239): the output rows are then NOT boxes of the stage problem (y_bounded = 0 above) but rows
    // widened by one shared slack vector [sl_ub; sl_lb] with weight Ws in every stage's cost - mpc_soft.hpp
    int soft;
    double Ws[2 * kMaxY][2 * kMaxY];
};

typedef __attribute__((address_space(4))) DevProblem ConstProblem;

#define MPC_UNROLL _Pragma("unroll")
// Diagnostic build only (-DMPC_STAMPS, tools/stamps.py): shader cycles per sweep accumulated into mpc_stamp_buf[wave][8].
// Never compiled into the product library; the stamp values reach no output of the solver.
#ifdef MPC_STAMPS
__device__ unsigned long long mpc_stamp_buf[4096 * 8];
#define MPC_STAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
    if (threadIdx.x == 0) mpc_stamp_buf[blockIdx.x * 8 + (slot)] += t_ - stamp_prev_; stamp_prev_ = t_; } while (0)
#define MPC_STAMP_INIT unsigned long long stamp_prev_ = __builtin_amdgcn_s_memtime();
#define MPC_STAMP_RESET stamp_prev_ = __builtin_amdgcn_s_memtime();
#else
#define MPC_STAMP(slot) do { } while (0)
#define MPC_STAMP_INIT
#define MPC_STAMP_RESET
#endif

__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }   // v_max_f64, one instruction
__device__ __forceinline__ double dmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ bool fin(double a) { return fabs(a) < 1.0e300; }
__device__ __forceinline__ double comp_measure(double s, double l)
{
    return dmin(dmin(s, l) * (1.0 / kTolC), s * l * (1.0 / kTolMu));
}

// accurate reciprocal without the IEEE division sequence: v_rcp_f64 + two Newton steps (~1 ulp for normal x)
__device__ __forceinline__ double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(e, r, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(e, r, r);
    return r;
}
// raw v_rcp_f64 (about 2^-27 relative): only for step-length ratio tests, which keep a 0.5 % margin anyway
__device__ __forceinline__ double frcp_approx(double x) { return __builtin_amdgcn_rcp(x); }

// keep a wave-uniform constant in a VGPR (stops the compiler from parking it in - and spilling - SGPRs)
__device__ __forceinline__ double vreg(double x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// ---- OCP workspace --------------------------------------------------------------
// One wave owns a contiguous slab; inside it block k (u_k, z_{k+1} and everything attached to them) is a
// run of SLOTS v2d "slots", each slot holding 64 lanes x 2 doubles = 1 KiB, so that every access of a
// wave is one fully coalesced global_load/store_dwordx4 at a compile-time offset from a scalar base.
template <int NS, int NU, int NC>
struct BlkLayout {
    static constexpr int NV = NS + NU;
    static constexpr int S = 0, L = NC, P = 2 * NC, IS = 3 * NC;   // pair slots {lo, hi}, one per bounded variable
    static constexpr int U = 4 * NC, Z = U + (NU + 1) / 2, DU = Z + (NS + 1) / 2, DZ = DU + (NU + 1) / 2,
                         KFF = DZ + (NS + 1) / 2, K = KFF + (NU + 1) / 2, LI = K + (NU * NS + 1) / 2,
                         SLOTS = LI + (NU * (NU + 1) / 2 + 1) / 2;
};

// A block pointer is a wave-uniform base plus the lane index; the pointer is explicitly in the global address
// space so that accesses are global_load/store_dwordx4 with immediate offsets.
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v2d gv2d;
struct BlkPtr {
    gv2d *base; int lane;
    __device__ __forceinline__ gv2d &operator[](int i) const { return base[i + lane]; }
};
struct Ws {
    gv2d *slab;   // wave slab (uniform), pointing at block 0; blocks -1 and N exist as guard blocks
    int N, slots, lane;
    __device__ __forceinline__ BlkPtr blk(int k) const { return BlkPtr{slab + (ptrdiff_t)k * slots * 64, lane}; }
};

template <int CNT>
__device__ __forceinline__ void ld_field(const BlkPtr blk, int slot0, double (&out)[CNT])
{
    MPC_UNROLL for (int j = 0; j < (CNT + 1) / 2; j++) {
        const v2d v = blk[(slot0 + j) * 64];
        out[2 * j] = v.x;
        if (2 * j + 1 < CNT) out[2 * j + 1 < CNT ? 2 * j + 1 : 0] = v.y;
    }
}
template <int CNT>
__device__ __forceinline__ void st_field(const BlkPtr blk, int slot0, const double (&in)[CNT])
{
    MPC_UNROLL for (int j = 0; j < (CNT + 1) / 2; j++) {
        v2d v;
        v.x = in[2 * j];
        v.y = (2 * j + 1 < CNT) ? in[2 * j + 1 < CNT ? 2 * j + 1 : 0] : 0.0;
        blk[(slot0 + j) * 64] = v;
    }
}

#include "mpc_sym.hpp"

// Per-instance data of one OCP (registers)
template <int NS, int NU>
struct OcpInst {
    double z0[NS], zr[NS], ur[NU], c[NS], us[NU], zlo_m[NS], zhi_m[NS];
    double zrN[NS];      // reference of the terminal cost: zr, or - terminal equality, lane solver - zr moved against the measured miss (term_aim)
    bool ok0;
};

// xhat, xs [NX]; us, u_prev [NU]; dhat [ND]  ->  stage-form instance
template <int NX, int NU, int NY, int ND, bool DU, int NG, class PT>
__device__ __forceinline__ void build_inst(const PT &P, const double (&xhat)[NX], const double (&xs)[NX],
                                           const double (&us)[NU], const double *dhat, const double (&u_prev)[NU],
                                           OcpInst<NX + (DU ? NU : 0) + NG, NU> &q)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG, NB = NX + (DU ? NU : 0);
    MPC_UNROLL for (int i = 0; i < NX; i++) {
        double c = P.fxc[i];
        MPC_UNROLL for (int j = 0; j < ND; j++) c += P.Bd[i][j] * dhat[j];
        q.c[i] = c; q.z0[i] = xhat[i]; q.zr[i] = xs[i];
    }
    MPC_UNROLL for (int i = 0; i < NU; i++) { q.us[i] = us[i]; q.ur[i] = DU ? 0.0 : us[i]; }
    if (DU) {
        MPC_UNROLL for (int i = 0; i < NU; i++) { q.z0[NX + i] = u_prev[i]; q.zr[NX + i] = 0.0; q.c[NX + i] = 0.0; }
        if (P.in_is_du) {      // input v = u - u_prev: cold start v = 0; the cost on u - us makes us the reference of the u_prev states
            MPC_UNROLL for (int i = 0; i < NU; i++) { q.us[i] = 0.0; if (P.zr_us) q.zr[NX + i] = us[i]; }
        }
    }
    MPC_UNROLL for (int g = 0; g < NG; g++) {      // output-row states w = C_i x: initial value, reference, affine term
        const int r = P.yg_row[g];
        double w0 = 0.0, wr = 0.0, wc = 0.0;
        if (r >= 0) { MPC_UNROLL for (int j = 0; j < NX; j++) { const double cij = P.Cm[r < 0 ? 0 : r][j]; w0 += cij * xhat[j]; wr += cij * xs[j]; wc += cij * q.c[j]; } }
        else {      // a user inequality row (yg_row = -1): w+ = Gx x + Gu u + (g0 + Gd dhat), its coefficients in row NB + g of A / B, its constant in that row of fxc / Bd; no cost on w
            wc = P.fxc[NB + g];
            MPC_UNROLL for (int j = 0; j < ND; j++) wc += P.Bd[NB + g][j] * dhat[j];
        }
        q.z0[NB + g] = w0; q.zr[NB + g] = wr; q.c[NB + g] = wc;
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
            MPC_UNROLL for (int j = 0; j < NS; j++)
                if (j == idx) { q.zlo_m[j] = dmax(q.zlo_m[j], lo); q.zhi_m[j] = dmin(q.zhi_m[j], hi); }
        }
    }
    MPC_UNROLL for (int i = 0; i < NS; i++) q.zrN[i] = q.zr[i];
}

// --------------------------------------------------------------------------------------------------------
// RPDIP: Mehrotra predictor-corrector, Riccati KKT solves, one instance per lane.
// Four sweeps over the horizon per iteration:
//   B1 (backward)  apply the previous step, residuals + convergence data, factorisation, predictor rhs
//   F1 (forward)   predictor direction, step length, second-order products
//   B2 (backward)  corrector rhs
//   F2 (forward)   corrector direction, step length
// Every sweep loads block k-1 / k+1 while it computes block k (software prefetch: one wave per SIMD has
// nobody else to hide its memory latency behind).
// Returns the status; u0/z1 receive the first input / next state of the final iterate.
// --------------------------------------------------------------------------------------------------------
template <int NS, int NU>
struct StageConst {      // the hot constants, held in VGPRs
    double A[NS][NS], B[NS][NU], Q[NS][NS], M[NS][NU], R[NU][NU], Pf[NS][NS];
};

template <int NS, int NU, bool HASM>
__device__ __forceinline__ void load_stage_const(const DevProblem &P, StageConst<NS, NU> &C)
{
    MPC_UNROLL for (int i = 0; i < NS; i++) {
        MPC_UNROLL for (int j = 0; j < NS; j++) { C.A[i][j] = vreg(P.A[i][j]); C.Q[i][j] = vreg(P.Q[i][j]); C.Pf[i][j] = vreg(P.Pf[i][j]); }
        MPC_UNROLL for (int j = 0; j < NU; j++) { C.B[i][j] = vreg(P.B[i][j]); C.M[i][j] = HASM ? vreg(P.M[i][j]) : 0.0; }
    }
    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) C.R[i][j] = vreg(P.R[i][j]); }
}

// picks the per-block copy (LTV) or the constant matrix, as a reference of the right type
template <bool LTV, class TL, class TC>
__device__ __forceinline__ const auto &ab_pick(const TL &l, const TC &c)
{
    if constexpr (LTV) return l; else return c;
}

// NC = number of bounded variables per block (NU: inputs only, NS+NU: inputs and states); MASKED = some of
// those bounds may be absent (+-inf).  The host picks the cheapest variant the problem allows.
// LTV: the stage matrices differ from block to block (the QP of one SQP iteration of the non-linear path, x+ = A_k x + B_k u + c_k):
// ltv points at this lane's column of [block][A (NS*NS) | B (NS*NU) | c (NS) | ...][64 lanes], ltv_stride entries per block (ltv_boff >= 0: entries ltv_boff.. hold the block's own state bounds lo[NS] | hi[NS], finite where the constant ones are); shift = 0 then warm-starts from the same
// stage of the workspace (an SQP iteration of the same step) instead of the next one.
template <int NS, int NU, bool HASM, int NC, bool MASKED, bool LTV = false>
__device__ int rpdip_lane(const DevProblem &P, const StageConst<NS, NU> &C, const OcpInst<NS, NU> &q, const Ws &ws,
                          int max_iter, bool warm, double ws_delta, double (&u0)[NU], double (&z1)[NS], double (&res)[3], int &iters,
                          const double *ltv = nullptr, int shift = 1, int ltv_stride = NS * (NS + NU + 1), int ltv_boff = -1)
{
    const int NLTV = ltv_stride;      // entries per block of the ltv slab (A | B | c first)
    // per-block matrices: loaded into (Al, Bl) for LTV; the constant ones are used in place otherwise
    auto load_ab = [&](int k, double (&Al)[NS][NS], double (&Bl)[NS][NU]) {
        if (LTV) {
            const double *blk = ltv + (size_t)k * NLTV * 64;
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                MPC_UNROLL for (int j = 0; j < NS; j++) Al[i][j] = blk[(i * NS + j) * 64];
                MPC_UNROLL for (int j = 0; j < NU; j++) Bl[i][j] = blk[(NS * NS + i * NU + j) * 64];
            }
        }
    };
#define MPC_LOAD_AB(k)                                                                     \
    double Al_[LTV ? NS : 1][LTV ? NS : 1], Bl_[LTV ? NS : 1][LTV ? NU : 1];               \
    if constexpr (LTV) load_ab(k, Al_, Bl_);                                               \
    const auto &AA = ab_pick<LTV>(Al_, C.A);                                               \
    const auto &BB = ab_pick<LTV>(Bl_, C.B);
    using L = BlkLayout<NS, NU, NC>;
    constexpr int NV = NS + NU;
    static_assert(NC == NU || NC == NV, "bounded variables: inputs, or inputs and states");
    const int N = ws.N;
    res[0] = res[1] = res[2] = 0.0;
    iters = 0;
    if (!q.ok0) return kInfeasible;
    MPC_STAMP_INIT

    // bounds of block k = (u_k, z_{k+1}).  cur_lo/cur_hi hold the bounds of the block being processed: the
    // sweeps switch them between "mid" (k < N-1) and "end" (k = N-1) once per sweep instead of selecting per block.
    double lo_m[NC], hi_m[NC], lo_e[NC], hi_e[NC], cur_lo[NC], cur_hi[NC];
    bool fl_m[NC], fh_m[NC], fl_e[NC], fh_e[NC], cur_fl[NC], cur_fh[NC];
    double ncon = 0.0;
    MPC_UNROLL for (int i = 0; i < NC; i++) {
        const double lm = i < NU ? P.ulo[i < NU ? i : 0] : q.zlo_m[i >= NU ? i - NU : 0], hm = i < NU ? P.uhi[i < NU ? i : 0] : q.zhi_m[i >= NU ? i - NU : 0];
        const double le = i < NU ? P.ulo[i < NU ? i : 0] : P.zlo_e[i >= NU ? i - NU : 0], he = i < NU ? P.uhi[i < NU ? i : 0] : P.zhi_e[i >= NU ? i - NU : 0];
        fl_m[i] = MASKED ? fin(lm) : true; fh_m[i] = MASKED ? fin(hm) : true; fl_e[i] = MASKED ? fin(le) : true; fh_e[i] = MASKED ? fin(he) : true;
        lo_m[i] = fl_m[i] ? lm : 0.0; hi_m[i] = fh_m[i] ? hm : 0.0; lo_e[i] = fl_e[i] ? le : 0.0; hi_e[i] = fh_e[i] ? he : 0.0;
        ncon += (double)(N - 1) * ((fl_m[i] ? 1 : 0) + (fh_m[i] ? 1 : 0)) + (fl_e[i] ? 1 : 0) + (fh_e[i] ? 1 : 0);
    }
    const double inv_ncon = 1.0 / dmax(ncon, 1.0);
    auto use_mid = [&]() { MPC_UNROLL for (int i = 0; i < NC; i++) { cur_lo[i] = lo_m[i]; cur_hi[i] = hi_m[i]; cur_fl[i] = fl_m[i]; cur_fh[i] = fh_m[i]; } };
    auto use_end = [&]() { MPC_UNROLL for (int i = 0; i < NC; i++) { cur_lo[i] = lo_e[i]; cur_hi[i] = hi_e[i]; cur_fl[i] = fl_e[i]; cur_fh[i] = fh_e[i]; } };
    // per-block state bounds (time-varying output offsets py_k move the boxes that stand for output rows)
    auto use_stage = [&](int k) {
        if (LTV && ltv_boff >= 0 && NC > NU) {
            const double *sb = ltv + ((size_t)k * NLTV + ltv_boff) * 64;
            MPC_UNROLL for (int i = NU; i < NC; i++) {
                if (cur_fl[i]) cur_lo[i] = sb[(i - NU) * 64];
                if (cur_fh[i]) cur_hi[i] = sb[(NS + i - NU) * 64];
            }
        }
    };
#define MPC_BOUNDS(k, i, lo, hi, fl, fh)                                   \
    const double lo = cur_lo[i], hi = cur_hi[i];                           \
    const bool fl = MASKED ? cur_fl[i] : true, fh = MASKED ? cur_fh[i] : true;

    // ---- initial point ---------------------------------------------------------------------------------
    // cold: u = us pushed inside its box, slacks >= kSMin, multipliers kMu0/s.
    // warm (per lane; closed loop only): the final iterate of the previous step, which is
    // still in this instance's workspace, shifted by one stage: u clipped to its box, slacks >= a floor,
    // multipliers max(previous, floor/s); the floors scale with how far the problem data moved since the last step.
    // z is simulated from the new initial state in both cases.
    {
        const double ws_smin = dmin(dmax(kWsKappa * ws_delta, kWsSMinLo), kWsSMinHi), ws_mu = kWsMuFactor * ws_smin * ws_smin;
        double ucold[NU], z[NS], zero_u[NU], zero_z[NS];
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            const double lo = P.ulo[i], hi = P.uhi[i];
            const bool fl = fin(lo), fh = fin(hi);
            double push, v = q.us[i];
            if (fl && fh) push = 0.1 * (hi - lo);
            else push = 0.1 * dmax(1.0, fabs(fl ? lo : (fh ? hi : 0.0)));
            if (fl) v = dmax(v, lo + push);
            if (fh) v = dmin(v, hi - push);
            ucold[i] = v; zero_u[i] = 0.0;
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { z[i] = q.z0[i]; zero_z[i] = 0.0; }
        use_mid();
        for (int k = 0; k < N; k++) {
            if (k == N - 1) use_end();
            use_stage(k);
            const BlkPtr b = ws.blk(k);
            const BlkPtr src = ws.blk((shift && k + 1 < N) ? k + 1 : k);      // previous step's block k+1 (read before block k is written)
            MPC_LOAD_AB(k)
            double uk[NU], uprev[NU];
            v2d lprev[NC];
            ld_field<NU>(src, L::U, uprev);
            MPC_UNROLL for (int i = 0; i < NC; i++) lprev[i] = src[(L::L + i) * 64];
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                double v = uprev[i];
                if (fin(P.ulo[i])) v = dmax(v, P.ulo[i]);
                if (fin(P.uhi[i])) v = dmin(v, P.uhi[i]);
                uk[i] = warm ? v : ucold[i];
            }
            double zn[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                double a = LTV ? ltv[((size_t)k * NLTV + NS * NS + NS * NU + i) * 64] : q.c[i];
                MPC_UNROLL for (int j = 0; j < NS; j++) a += AA[i][j] * z[j];
                MPC_UNROLL for (int j = 0; j < NU; j++) a += BB[i][j] * uk[j];
                zn[i] = a;
            }
            MPC_UNROLL for (int i = 0; i < NS; i++) z[i] = zn[i];
            MPC_UNROLL for (int i = 0; i < NC; i++) {
                MPC_BOUNDS(k, i, lo, hi, fl, fh)
                const double v = i < NU ? uk[i < NU ? i : 0] : z[i >= NU ? i - NU : 0];
                const double smin = warm ? ws_smin : kSMin;
                v2d sv, lv, iv, pz;
                sv.x = fl ? dmax(v - lo, smin) : 1.0; sv.y = fh ? dmax(hi - v, smin) : 1.0;
                iv.x = frcp(sv.x); iv.y = frcp(sv.y);
                const double llo = warm ? dmax(lprev[i].x, ws_mu * iv.x) : kMu0 * iv.x;
                const double lhi = warm ? dmax(lprev[i].y, ws_mu * iv.y) : kMu0 * iv.y;
                lv.x = fl ? llo : 0.0; lv.y = fh ? lhi : 0.0;
                pz.x = 0.0; pz.y = 0.0;
                b[(L::S + i) * 64] = sv; b[(L::L + i) * 64] = lv; b[(L::P + i) * 64] = pz;
            }
            st_field<NU>(b, L::U, uk); st_field<NS>(b, L::Z, z);
            st_field<NU>(b, L::DU, zero_u); st_field<NS>(b, L::DZ, zero_z);
        }
    }

    MPC_STAMP(0);      // init sweep
    double alpha = 0.0, sm = 0.0, gscale = 1.0;
    int stall = 0, status = kMaxIter;
    for (int it = 0;; it++) {
        // ======================= sweep B1 (backward) =================================================
        double mu_sum = 0.0, res_p = 0.0, res_s = 0.0, cres = 0.0, lmax = 0.0;
        double pi[NS], Pm[NS][NS], pcar[NS], unext_dev[NU], Anext_[LTV ? NS : 1][LTV ? NS : 1];
        if constexpr (LTV) { MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Anext_[i][j] = 0.0; } }
        const auto &Anext = ab_pick<LTV>(Anext_, C.A);      // LTI: A' pi with pi = 0 at the last block is the same thing
        bool pd_ok = true;
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            pi[i] = 0.0; pcar[i] = 0.0;
            MPC_UNROLL for (int j = 0; j < NS; j++) Pm[i][j] = C.Pf[i][j];
        }
        MPC_UNROLL for (int i = 0; i < NU; i++) unext_dev[i] = 0.0;
        double ublk[NU], zblk[NS];
        const bool upd = alpha != 0.0;
        struct B1Blk { v2d s[NC], l[NC], p[NC], is[NC]; double u[NU], z[NS], du[NU], dz[NS]; };
        auto load_b1 = [&](int k, B1Blk &d) {
            const BlkPtr b = ws.blk(k);
            MPC_UNROLL for (int i = 0; i < NC; i++) { d.s[i] = b[(L::S + i) * 64]; d.l[i] = b[(L::L + i) * 64]; }
            ld_field<NU>(b, L::U, d.u); ld_field<NS>(b, L::Z, d.z);
            if (upd) {
                MPC_UNROLL for (int i = 0; i < NC; i++) { d.p[i] = b[(L::P + i) * 64]; }
                ld_field<NU>(b, L::DU, d.du); ld_field<NS>(b, L::DZ, d.dz);
            }
        };
        B1Blk cur;
        load_b1(N - 1, cur);
        use_end();
        for (int k = N - 1; k >= 0; k--) {
            if (k == N - 2) use_mid();
            use_stage(k);
            const BlkPtr b = ws.blk(k);
            double sig[NV], dlm[NV], haff[NV];
            MPC_UNROLL for (int i = NC; i < NV; i++) { sig[i] = 0.0; dlm[i] = 0.0; haff[i] = 0.0; }
            // ---- phase A: apply the previous step to block k, residuals and barrier weights -----------
            MPC_UNROLL for (int i = 0; i < NU; i++) ublk[i] = cur.u[i];
            MPC_UNROLL for (int i = 0; i < NS; i++) zblk[i] = cur.z[i];
            MPC_UNROLL for (int i = 0; i < NC; i++) {
                MPC_BOUNDS(k, i, lo, hi, fl, fh)
                double v = i < NU ? ublk[i < NU ? i : 0] : zblk[i >= NU ? i - NU : 0];
                double sl = cur.s[i].x, sh = cur.s[i].y, ll = cur.l[i].x, lh = cur.l[i].y;
                if (upd) {   // apply the step of the previous iteration
                    const double dv = i < NU ? cur.du[i < NU ? i : 0] : cur.dz[i >= NU ? i - NU : 0];
                    const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                    const double rch = fh ? sh * lh - dmax(sm, lh * kSFloor) + cur.p[i].y : 0.0;
                    const double rcl = fl ? sl * ll - dmax(sm, ll * kSFloor) + cur.p[i].x : 0.0;
                    const double dsh = fh ? -rh - dv : 0.0, dsl = fl ? rl + dv : 0.0;
                    const double dlh = fh ? (-rch - lh * dsh) * frcp(sh) : 0.0, dll = fl ? (-rcl - ll * dsl) * frcp(sl) : 0.0;      // 1/s recomputed: one field less to stream
                    sl += alpha * dsl; sh += alpha * dsh; ll += alpha * dll; lh += alpha * dlh; v += alpha * dv;
                    if (i < NU) ublk[i < NU ? i : 0] = v; else zblk[i >= NU ? i - NU : 0] = v;
                }
                const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                const double isl = frcp(sl), ish = frcp(sh);
                if (upd) {
                    v2d t; t.x = sl; t.y = sh; b[(L::S + i) * 64] = t;
                    t.x = ll; t.y = lh; b[(L::L + i) * 64] = t;
                }
                mu_sum += sl * ll + sh * lh;
                sig[i] = ll * isl + lh * ish;
                dlm[i] = lh - ll;
                haff[i] = lh * (rh * ish - 1.0) + ll * (rl * isl + 1.0);   // h for rc = s*l
                res_p = dmax(res_p, dmax(fabs(rl), fabs(rh)));
                cres = dmax(cres, dmax(comp_measure(sl, ll), comp_measure(sh, lh)));
                lmax = dmax(lmax, dmax(ll, lh));
            }
            if (upd) {
                MPC_UNROLL for (int i = NC; i < NV; i++) zblk[i >= NU ? i - NU : 0] += alpha * cur.dz[i >= NU ? i - NU : 0];   // unbounded states
                st_field<NU>(b, L::U, ublk); st_field<NS>(b, L::Z, zblk);
            }
            // ---- prefetch block k-1 into the (now dead) buffer; it lands while phase B computes ---------
            load_b1(k - 1, cur);     // k-1 = -1 is a guard block
            // ---- phase B: Riccati step.  P_{k+1} completed with the barrier weights of z_{k+1} -----------
            MPC_UNROLL for (int i = 0; i < NS; i++) Pm[i][i] += sig[NU + i];
            MPC_LOAD_AB(k)
            double PB[NS][NU], PA[NS][NS], Lam[NU][NU], Psi[NU][NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pm[i][l] * BB[l][j]; PB[i][j] = a; }
                MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pm[i][l] * AA[l][j]; PA[i][j] = a; }
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                MPC_UNROLL for (int j = 0; j <= i; j++) { double a = C.R[i][j] + (i == j ? sig[i] : 0.0); MPC_UNROLL for (int l = 0; l < NS; l++) a += BB[l][i] * PB[l][j]; Lam[i][j] = a; Lam[j][i] = a; }
                MPC_UNROLL for (int j = 0; j < NS; j++) { double a = HASM ? C.M[j][i] : 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += BB[l][i] * PA[l][j]; Psi[i][j] = a; }
            }
            pd_ok = sym_inverse<NU>(Lam) && pd_ok;     // Lam now holds Li
            double Kk[NU][NS], Acl[NS][NS];
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NU; l++) a += Lam[i][l] * Psi[l][j]; Kk[i][j] = -a; } }
            {
                double kflat[NU * NS], liflat[NU * (NU + 1) / 2];
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) kflat[i * NS + j] = Kk[i][j]; }
                int c = 0;
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j <= i; j++) { liflat[c] = Lam[i][j]; c++; } }
                st_field<NU * NS>(b, L::K, kflat); st_field<NU * (NU + 1) / 2>(b, L::LI, liflat);
            }
            if (k > 0) {
                MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = AA[i][j]; MPC_UNROLL for (int l = 0; l < NU; l++) a += BB[i][l] * Kk[l][j]; Acl[i][j] = a; } }
                // closed-loop (Joseph) form: P_k = Q + Acl' P Acl + K' Rt K + M K + K' M'  (no cancellation)
                double T[NS][NS], RK[NU][NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += Pm[i][l] * Acl[l][j]; T[i][j] = a; } }
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = sig[i] * Kk[i][j]; MPC_UNROLL for (int l = 0; l < NU; l++) a += C.R[i][l] * Kk[l][j]; RK[i][j] = a; } }
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    MPC_UNROLL for (int j = 0; j <= i; j++) {      // symmetric in exact arithmetic: lower triangle, mirrored
                        double a = C.Q[i][j];
                        MPC_UNROLL for (int l = 0; l < NS; l++) a += Acl[l][i] * T[l][j];
                        MPC_UNROLL for (int l = 0; l < NU; l++) a += Kk[l][i] * RK[l][j];
                        if (HASM) { MPC_UNROLL for (int l = 0; l < NU; l++) a += C.M[i][l] * Kk[l][j] + Kk[l][i] * C.M[j][l]; }
                        Pm[i][j] = a; Pm[j][i] = a;
                    }
                }
            }
            // ---- gradients of the current point (cost + bound multipliers), adjoint, predictor rhs --------
            double dz1[NS], du[NU], gz1[NS], gu[NU];
            MPC_UNROLL for (int i = 0; i < NS; i++) dz1[i] = zblk[i] - q.zr[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = ublk[i] - q.ur[i];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                double a = dlm[NU + i];
                if (k == N - 1) { MPC_UNROLL for (int j = 0; j < NS; j++) a += C.Pf[i][j] * (zblk[j] - q.zrN[j]); }
                else {
                    MPC_UNROLL for (int j = 0; j < NS; j++) a += C.Q[i][j] * dz1[j];
                    if (HASM) { MPC_UNROLL for (int j = 0; j < NU; j++) a += C.M[i][j] * unext_dev[j]; }
                }
                gz1[i] = a;
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                double a = dlm[i];
                MPC_UNROLL for (int j = 0; j < NU; j++) a += C.R[i][j] * du[j];
                gu[i] = a;
            }
            if (HASM) {   // M'(z_k - zr): z_k lives in block k-1 (just prefetched, not yet updated) or is z0
                double zk[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    double zv = k > 0 ? cur.z[i] : q.z0[i];
                    if (k > 0 && upd) zv += alpha * cur.dz[i];
                    zk[i] = zv - q.zr[i];
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) gu[i] += C.M[j][i] * zk[j]; }
            }
            // adjoint pi_{k+1} = gz_{k+1} + A' pi_{k+2};  stationarity residual r_u,k = gu_k + B' pi_{k+1}
            {
                double pn[NS];
                // the costate of z_{k+1} passes through the dynamics of the NEXT block (z_{k+2} = A_{k+1} z_{k+1} + ..)
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = gz1[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += Anext[j][i] * pi[j]; pn[i] = a; }
                MPC_UNROLL for (int i = 0; i < NS; i++) pi[i] = pn[i];
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = gu[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += BB[j][i] * pi[j]; res_s = dmax(res_s, fabs(a)); }
                if constexpr (LTV) { MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Anext_[i][j] = AA[i][j]; } }
            }
            double pv[NS], qu[NU];
            MPC_UNROLL for (int i = 0; i < NS; i++) pv[i] = gz1[i] + haff[NU + i] + pcar[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) qu[i] = gu[i] + haff[i];
            {
                double psi[NU], kff[NU];
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = qu[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += BB[j][i] * pv[j]; psi[i] = a; }
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) a += Lam[i][j] * psi[j]; kff[i] = -a; }
                st_field<NU>(b, L::KFF, kff);
                if (k > 0) {     // p_k carry = Acl' pv + K' qu = A' pv + K' psi
                    double pn[NS];
                    MPC_UNROLL for (int i = 0; i < NS; i++) {
                        double a = 0.0;
                        MPC_UNROLL for (int j = 0; j < NS; j++) a += AA[j][i] * pv[j];
                        MPC_UNROLL for (int j = 0; j < NU; j++) a += Kk[j][i] * psi[j];
                        pn[i] = a;
                    }
                    MPC_UNROLL for (int i = 0; i < NS; i++) pcar[i] = pn[i];
                }
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) unext_dev[i] = du[i];
        }
        MPC_STAMP(1);  // B1
        // ---- convergence / failure tests at the current iterate ------------------------------------
        const double mu = mu_sum * inv_ncon;
        // the scale of the stationarity test is the first residual - without the terminal weight's share when that weight stands for
        // the terminal equality (a shifted warm start misses xs by a little, times 1e14)
        if (it == 0) gscale = dmax(1.0, P.term_cons ? dmin(res_s, P.term_gcap) : res_s);
        res[0] = res_s; res[1] = res_p; res[2] = mu;
        iters = it;
        MPC_UNROLL for (int i = 0; i < NU; i++) u0[i] = ublk[i];
        MPC_UNROLL for (int i = 0; i < NS; i++) z1[i] = zblk[i];
        const bool ok_cp = (cres <= 1.0) && (res_p <= kTolFeas);
        stall = ok_cp ? stall + 1 : 0;
        // term_floor (zero without a terminal equality): the terminal weight times one unit in the last place of x_N, below which
        // the residual of a representable iterate cannot go
        if (ok_cp && (res_s <= kTolStat * gscale + P.term_floor || (stall > kStallMax && res_s <= kTolStatAcc * gscale + P.term_floor))) { status = kSolved; break; }
        if (lmax > kInfeasZ * gscale || !(fabs(mu) < 1.0e300) || !pd_ok) { status = kInfeasible; break; }
        if (it == max_iter) { status = kMaxIter; break; }

        // ======================= sweep F1 (forward): predictor ======================================
        double m_aff = 1.0, s1 = 0.0, s2 = 0.0;      // m = max(1, max_i -d_i/x_i); step to the boundary = 1/m
        {
            struct F1Blk { v2d s[NC], l[NC], is[NC]; double u[NU], z[NS], kff[NU], K[NU * NS]; };
            auto load_f1 = [&](int k, F1Blk &d) {
                const BlkPtr b = ws.blk(k);
                MPC_UNROLL for (int i = 0; i < NC; i++) { d.s[i] = b[(L::S + i) * 64]; d.l[i] = b[(L::L + i) * 64]; }
                ld_field<NU>(b, L::U, d.u); ld_field<NS>(b, L::Z, d.z); ld_field<NU>(b, L::KFF, d.kff); ld_field<NU * NS>(b, L::K, d.K);
            };
            F1Blk c1;
            load_f1(0, c1);
            use_mid();
            double dz[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = 0.0;
            for (int k = 0; k < N; k++) {
                if (k == N - 1) use_end();
                use_stage(k);
                const BlkPtr b = ws.blk(k);
                const BlkPtr nb = ws.blk(k + 1);      // block N is a guard block; every field is reloaded in place
                double ddu[NU], dzn[NS];                 // right after its last use (rolling prefetch, no second buffer)
                MPC_LOAD_AB(k)
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = c1.kff[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += c1.K[i * NS + j] * dz[j]; ddu[i] = a; }
                ld_field<NU>(nb, L::KFF, c1.kff); ld_field<NU * NS>(nb, L::K, c1.K);
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) a += AA[i][j] * dz[j]; MPC_UNROLL for (int j = 0; j < NU; j++) a += BB[i][j] * ddu[j]; dzn[i] = a; }
                MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = dzn[i];
                MPC_UNROLL for (int i = 0; i < NC; i++) {
                    MPC_BOUNDS(k, i, lo, hi, fl, fh)
                    const double v = i < NU ? c1.u[i < NU ? i : 0] : c1.z[i >= NU ? i - NU : 0];
                    const double dv = i < NU ? ddu[i < NU ? i : 0] : dz[i >= NU ? i - NU : 0];
                    const double sl = c1.s[i].x, sh = c1.s[i].y, ll = c1.l[i].x, lh = c1.l[i].y, isl = frcp(sl), ish = frcp(sh);
                    c1.s[i] = nb[(L::S + i) * 64]; c1.l[i] = nb[(L::L + i) * 64];
                    const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                    const double dsh = fh ? -rh - dv : 0.0, dsl = fl ? rl + dv : 0.0;
                    // rc = s*l:  dl = -l - l*ds/s,  -dl/l = 1 + ds/s
                    const double qh = dsh * ish, ql = dsl * isl;
                    const double dlh = fh ? -lh - lh * qh : 0.0, dll = fl ? -ll - ll * ql : 0.0;
                    m_aff = dmax(m_aff, dmax(-ql, -qh));
                    if (fl) m_aff = dmax(m_aff, 1.0 + ql);
                    if (fh) m_aff = dmax(m_aff, 1.0 + qh);
                    s1 += sl * dll + ll * dsl + sh * dlh + lh * dsh;
                    s2 += dsl * dll + dsh * dlh;
                    v2d t; t.x = dsl * dll; t.y = dsh * dlh;
                    b[(L::P + i) * 64] = t;
                }
                ld_field<NU>(nb, L::U, c1.u); ld_field<NS>(nb, L::Z, c1.z);
            }
        }
        MPC_STAMP(2);  // F1
        {
            const double a_aff = frcp(m_aff);
            const double mu_aff = (mu_sum + a_aff * s1 + a_aff * a_aff * s2) * inv_ncon;
            const double rat = mu > 0.0 ? mu_aff * frcp(mu) : 0.0;
            sm = dmax(rat * rat * rat * mu, kMuFloor);
        }
        // ======================= sweep B2 (backward): corrector rhs ================================
        {
            struct B2Blk { v2d s[NC], l[NC], p[NC], is[NC]; double u[NU], z[NS], K[NU * NS], li[NU * (NU + 1) / 2]; };
            auto load_b2 = [&](int k, B2Blk &d) {
                const BlkPtr b = ws.blk(k);
                MPC_UNROLL for (int i = 0; i < NC; i++) { d.s[i] = b[(L::S + i) * 64]; d.l[i] = b[(L::L + i) * 64]; d.p[i] = b[(L::P + i) * 64]; }
                ld_field<NU>(b, L::U, d.u); ld_field<NS>(b, L::Z, d.z); ld_field<NU * NS>(b, L::K, d.K); ld_field<NU * (NU + 1) / 2>(b, L::LI, d.li);
            };
            B2Blk c2;
            load_b2(N - 1, c2);
            use_end();
            double pc[NS], und[NU], zprev[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) pc[i] = 0.0;
            MPC_UNROLL for (int i = 0; i < NU; i++) und[i] = 0.0;
            for (int k = N - 1; k >= 0; k--) {
                if (k == N - 2) use_mid();
                use_stage(k);
                const BlkPtr b = ws.blk(k);
                const BlkPtr nb = ws.blk(k - 1);       // block -1 is a guard block
                double hcc[NV], dlm[NV];
                MPC_UNROLL for (int i = NC; i < NV; i++) { hcc[i] = 0.0; dlm[i] = 0.0; }
                MPC_UNROLL for (int i = 0; i < NC; i++) {
                    MPC_BOUNDS(k, i, lo, hi, fl, fh)
                    const double v = i < NU ? c2.u[i < NU ? i : 0] : c2.z[i >= NU ? i - NU : 0];
                    const double sl = c2.s[i].x, sh = c2.s[i].y, ll = c2.l[i].x, lh = c2.l[i].y;
                    const double plo = c2.p[i].x, phi = c2.p[i].y, isl = frcp(sl), ish = frcp(sh);
                    c2.s[i] = nb[(L::S + i) * 64]; c2.l[i] = nb[(L::L + i) * 64]; c2.p[i] = nb[(L::P + i) * 64];
                    const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                    const double rch = fh ? sh * lh - dmax(sm, lh * kSFloor) + phi : 0.0;
                    const double rcl = fl ? sl * ll - dmax(sm, ll * kSFloor) + plo : 0.0;
                    hcc[i] = (-rch + lh * rh) * ish + (rcl + ll * rl) * isl;
                    dlm[i] = lh - ll;
                }
                double dz1[NS], du[NU], pv[NS], qu[NU];
                MPC_UNROLL for (int i = 0; i < NS; i++) dz1[i] = c2.z[i] - (k == N - 1 ? q.zrN[i] : q.zr[i]);      // (the terminal cost has its own reference)
                MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = c2.u[i] - q.ur[i];
                ld_field<NU>(nb, L::U, c2.u); ld_field<NS>(nb, L::Z, c2.z);
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    double a = dlm[NU + i] + hcc[NU + i] + pc[i];
                    if (k == N - 1) { MPC_UNROLL for (int j = 0; j < NS; j++) a += C.Pf[i][j] * dz1[j]; }
                    else {
                        MPC_UNROLL for (int j = 0; j < NS; j++) a += C.Q[i][j] * dz1[j];
                        if (HASM) { MPC_UNROLL for (int j = 0; j < NU; j++) a += C.M[i][j] * und[j]; }
                    }
                    pv[i] = a;
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) {
                    double a = dlm[i] + hcc[i];
                    MPC_UNROLL for (int j = 0; j < NU; j++) a += C.R[i][j] * du[j];
                    qu[i] = a;
                }
                double Li[NU][NU], Kk[NU * NS];
                {
                    int c = 0;
                    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j <= i; j++) { Li[i][j] = c2.li[c]; Li[j][i] = c2.li[c]; c++; } }
                    MPC_UNROLL for (int i = 0; i < NU * NS; i++) Kk[i] = c2.K[i];
                }
                ld_field<NU * NS>(nb, L::K, c2.K); ld_field<NU * (NU + 1) / 2>(nb, L::LI, c2.li);
                if (HASM) {      // M'(z_k - zr): z_k is block k-1 (just requested) or z0 - used last so the load has time
                    MPC_UNROLL for (int i = 0; i < NS; i++) zprev[i] = (k > 0 ? c2.z[i] : q.z0[i]) - q.zr[i];
                    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) qu[i] += C.M[j][i] * zprev[j]; }
                }
                {
                    double psi[NU], kff[NU];
                    MPC_LOAD_AB(k)
                    MPC_UNROLL for (int i = 0; i < NU; i++) { double a = qu[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += BB[j][i] * pv[j]; psi[i] = a; }
                    MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) a += Li[i][j] * psi[j]; kff[i] = -a; }
                    st_field<NU>(b, L::KFF, kff);
                    if (k > 0) {     // p_k carry = Acl' pv + K' qu = A' pv + K' psi
                        double pn[NS];
                        MPC_UNROLL for (int i = 0; i < NS; i++) {
                            double a = 0.0;
                            MPC_UNROLL for (int j = 0; j < NS; j++) a += AA[j][i] * pv[j];
                            MPC_UNROLL for (int j = 0; j < NU; j++) a += Kk[j * NS + i] * psi[j];
                            pn[i] = a;
                        }
                        MPC_UNROLL for (int i = 0; i < NS; i++) pc[i] = pn[i];
                    }
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) und[i] = du[i];
            }
        }
        MPC_STAMP(3);  // B2
        // ======================= sweep F2 (forward): corrector direction ============================
        double m_cc = kTau;      // alpha = min(1, kTau / max_i(-d_i/x_i)): the full step when the boundary is further than 1/kTau away
        {
            struct F2Blk { v2d s[NC], l[NC], p[NC], is[NC]; double u[NU], z[NS], kff[NU], K[NU * NS]; };
            auto load_f2 = [&](int k, F2Blk &d) {
                const BlkPtr b = ws.blk(k);
                MPC_UNROLL for (int i = 0; i < NC; i++) { d.s[i] = b[(L::S + i) * 64]; d.l[i] = b[(L::L + i) * 64]; d.p[i] = b[(L::P + i) * 64]; }
                ld_field<NU>(b, L::U, d.u); ld_field<NS>(b, L::Z, d.z); ld_field<NU>(b, L::KFF, d.kff); ld_field<NU * NS>(b, L::K, d.K);
            };
            F2Blk c3;
            load_f2(0, c3);
            use_mid();
            double dz[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = 0.0;
            for (int k = 0; k < N; k++) {
                if (k == N - 1) use_end();
                use_stage(k);
                const BlkPtr b = ws.blk(k);
                const BlkPtr nb = ws.blk(k + 1);
                double ddu[NU], dzn[NS];
                MPC_LOAD_AB(k)
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = c3.kff[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += c3.K[i * NS + j] * dz[j]; ddu[i] = a; }
                ld_field<NU>(nb, L::KFF, c3.kff); ld_field<NU * NS>(nb, L::K, c3.K);
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) a += AA[i][j] * dz[j]; MPC_UNROLL for (int j = 0; j < NU; j++) a += BB[i][j] * ddu[j]; dzn[i] = a; }
                MPC_UNROLL for (int i = 0; i < NS; i++) dz[i] = dzn[i];
                st_field<NU>(b, L::DU, ddu); st_field<NS>(b, L::DZ, dz);
                MPC_UNROLL for (int i = 0; i < NC; i++) {
                    MPC_BOUNDS(k, i, lo, hi, fl, fh)
                    const double v = i < NU ? c3.u[i < NU ? i : 0] : c3.z[i >= NU ? i - NU : 0];
                    const double dv = i < NU ? ddu[i < NU ? i : 0] : dz[i >= NU ? i - NU : 0];
                    const double sl = c3.s[i].x, sh = c3.s[i].y, ll = c3.l[i].x, lh = c3.l[i].y, isl = frcp(sl), ish = frcp(sh);
                    const double plo = c3.p[i].x, phi = c3.p[i].y;
                    c3.s[i] = nb[(L::S + i) * 64]; c3.l[i] = nb[(L::L + i) * 64]; c3.p[i] = nb[(L::P + i) * 64];
                    const double rh = fh ? v + sh - hi : 0.0, rl = fl ? v - sl - lo : 0.0;
                    const double rch = fh ? sh * lh - dmax(sm, lh * kSFloor) + phi : 0.0;
                    const double rcl = fl ? sl * ll - dmax(sm, ll * kSFloor) + plo : 0.0;
                    const double dsh = fh ? -rh - dv : 0.0, dsl = fl ? rl + dv : 0.0;
                    const double dlh = fh ? (-rch - lh * dsh) * ish : 0.0, dll = fl ? (-rcl - ll * dsl) * isl : 0.0;
                    m_cc = dmax(m_cc, dmax(-dsl * isl, -dsh * ish));
                    if (fl) m_cc = dmax(m_cc, -dll * frcp_approx(ll));
                    if (fh) m_cc = dmax(m_cc, -dlh * frcp_approx(lh));
                }
                ld_field<NU>(nb, L::U, c3.u); ld_field<NS>(nb, L::Z, c3.z);
            }
        }
        alpha = m_cc <= kTau ? 1.0 : kTau * frcp(m_cc);
        MPC_STAMP(4);  // F2
    }
#undef MPC_BOUNDS
#undef MPC_LOAD_AB
    return status;
}

// Terminal equality x_N = xs (TermCons, Control_Calc.py:197-198).  It is imposed through the terminal weight (mpc_amd.hip:build_problem:
// Pf = rho I on the model states, rho = 1e12 x the cost scale - the multiplier of the equality is then rho (x_N - xs), the optimum
// differs from the equality-constrained one by |multiplier| / rho, below the solver's own tolerance); a final iterate that still
// misses xs is the reference's 'Infeasible_Problem_Detected': the bounds do not let the horizon reach it.
template <int NS, int NU, int NC, int NX>
__device__ __forceinline__ bool term_missed(const DevProblem &P, const Ws &ws, const OcpInst<NS, NU> &q)
{
    using L = BlkLayout<NS, NU, NC>;
    double zN[NS], v = 0.0;
    ld_field<NS>(ws.blk(P.N - 1), L::Z, zN);
    MPC_UNROLL for (int i = 0; i < NX; i++) v = dmax(v, fabs(zN[i] - q.zr[i]) * frcp(dmax(1.0, fabs(q.zr[i]))));
    return !(v <= P.term_tol);
}

// Exact terminal equality on the lane solver ("aiming off", the method of multipliers written as a shift of the target).  The weight rho on
// |z_N - zr|^2 leaves a miss c = (multiplier of the equality) / rho - 1e-6 where the multipliers are of order 1e6, the short horizons - and the miss
// is an affine function of the terminal reference with slope -(1 - O(curvature / rho)): solving again with the terminal reference moved by -c
// leaves O(curvature / rho) c ~ 1e-12.  Returns true when another pass is worth it: the miss is beyond rounding but small enough to be the
// penalty's bias (a target the bounds keep out of reach misses by orders more, and stays status 2).
template <int NS, int NU, int NC, int NX>
__device__ __forceinline__ bool term_aim(const DevProblem &P, const Ws &ws, OcpInst<NS, NU> &q)
{
    using L = BlkLayout<NS, NU, NC>;
    double zN[NS], v = 0.0;
    ld_field<NS>(ws.blk(P.N - 1), L::Z, zN);
    MPC_UNROLL for (int i = 0; i < NX; i++) v = dmax(v, fabs(zN[i] - q.zr[i]) * frcp(dmax(1.0, fabs(q.zr[i]))));
    if (!(v > 1e-11 && v <= 1e-4)) return false;
    MPC_UNROLL for (int i = 0; i < NX; i++) q.zrN[i] -= zN[i] - q.zr[i];
    return true;
}

// --------------------------------------------------------------------------------------------------------
// target problem, reduced to the null space of [A-I, B]; registers only
// --------------------------------------------------------------------------------------------------------
// tw / twv: warm-start data of the closed loop for this instance, element f at tw[f * tws]: y[NR] l_lo[NC] l_hi[NC] gr[NR] w0[NC]
// of the last successful solve, and its validity flag; nullptr = cold (the per-call entry point).  DESIGN.md section 4.8.
template <int NX, int NU, int NY, int ND, class PT>
__device__ int target_lane(const PT &P, const double *usp, const double *ysp, const double *dhat,
                           const double (&us_prev)[NU], double (&xs)[NX], double (&us)[NU], double (&ys)[NY], int &iters,
                           double *tw = nullptr, size_t tws = 0, int32_t *twv = nullptr, const double *px0 = nullptr, const double *py0 = nullptr)
{
    constexpr int NV = NX + NU, NC = NV + NY, NR = NU;
    double cx[NX], e[NY], vp[NV], yp[NY], gr[NR], w0[NC], y[NR];
    double s_lo[NC], s_hi[NC], l_lo[NC], l_hi[NC], lo[NC], hi[NC];
    bool fl[NC], fh[NC];
    // px0 / py0: this step's model parameters p_x_k, p_y_k (def_px / def_py, MPC_code.py:492-501,693), per instance
    MPC_UNROLL for (int i = 0; i < NX; i++) { double a = P.fxc[i] + (px0 ? px0[i] : 0.0); MPC_UNROLL for (int j = 0; j < ND; j++) a += P.Bd[i][j] * dhat[j]; cx[i] = a; }
    MPC_UNROLL for (int i = 0; i < NY; i++) { double a = P.fyc[i] + (py0 ? py0[i] : 0.0); MPC_UNROLL for (int j = 0; j < ND; j++) a += P.Cd[i][j] * dhat[j]; e[i] = a; }
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
        sym_inverse<NR>(Hi);
        MPC_UNROLL for (int i = 0; i < NR; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NR; j++) a += Hi[i][j] * gr[j]; y[i] = -a; }
    }
    MPC_UNROLL for (int r = 0; r < NC; r++) {
        double v = w0[r]; MPC_UNROLL for (int c = 0; c < NR; c++) v += P.W[r][c] * y[c];
        s_lo[r] = fl[r] ? dmax(v - lo[r], kSMin) : 1.0; s_hi[r] = fh[r] ? dmax(hi[r] - v, kSMin) : 1.0;
        l_lo[r] = fl[r] ? kMu0 * frcp(s_lo[r]) : 0.0; l_hi[r] = fh[r] ? kMu0 * frcp(s_hi[r]) : 0.0;
    }
    if (tw != nullptr && *twv != 0) {
        double delta = 0.0;
        MPC_UNROLL for (int c = 0; c < NR; c++) delta = dmax(delta, fabs(gr[c] - tw[(NR + 2 * NC + c) * tws]));
        MPC_UNROLL for (int r = 0; r < NC; r++) delta = dmax(delta, fabs(w0[r] - tw[(2 * NR + 2 * NC + r) * tws]));
        if (delta <= kWsDelta) {
            const double smin = dmin(dmax(kWsKappa * delta, kWsSMinLo), kWsSMinHi), wmu = kWsMuFactor * smin * smin;
            MPC_UNROLL for (int c = 0; c < NR; c++) y[c] = tw[c * tws];
            MPC_UNROLL for (int r = 0; r < NC; r++) {
                double v = w0[r]; MPC_UNROLL for (int c = 0; c < NR; c++) v += P.W[r][c] * y[c];
                s_lo[r] = fl[r] ? dmax(v - lo[r], smin) : 1.0; s_hi[r] = fh[r] ? dmax(hi[r] - v, smin) : 1.0;
                l_lo[r] = fl[r] ? dmax(tw[(NR + r) * tws], wmu * frcp(s_lo[r])) : 0.0; l_hi[r] = fh[r] ? dmax(tw[(NR + NC + r) * tws], wmu * frcp(s_hi[r])) : 0.0;
            }
        }
    }
    double gscale = 1.0; int stall = 0, status = kMaxIter;
    MPC_UNROLL for (int c = 0; c < NR; c++) gscale = dmax(gscale, fabs(gr[c]));
    for (int it = 0;; it++) {
        // reciprocals of the slacks once per iteration (v_rcp_f64 + Newton, mpc::frcp) instead of IEEE divisions
        double mu = 0.0, res_p = 0.0, res_s = 0.0, cres = 0.0, lmax = 0.0, grad[NR], sig[NC], r_lo[NC], r_hi[NC], is_lo[NC], is_hi[NC];
        MPC_UNROLL for (int r = 0; r < NC; r++) {
            double v = w0[r]; MPC_UNROLL for (int c = 0; c < NR; c++) v += P.W[r][c] * y[c];
            r_lo[r] = fl[r] ? v - s_lo[r] - lo[r] : 0.0; r_hi[r] = fh[r] ? v + s_hi[r] - hi[r] : 0.0;
            mu += s_lo[r] * l_lo[r] + s_hi[r] * l_hi[r];
            is_lo[r] = frcp(s_lo[r]); is_hi[r] = frcp(s_hi[r]);
            sig[r] = l_lo[r] * is_lo[r] + l_hi[r] * is_hi[r];
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
        if (!sym_inverse<NR>(Ht)) { status = kInfeasible; break; }
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
                const double h = (-rc_hi[r] + l_hi[r] * r_hi[r]) * is_hi[r] + (rc_lo[r] + l_lo[r] * r_lo[r]) * is_lo[r];
                MPC_UNROLL for (int c = 0; c < NR; c++) rhs[c] += h * P.W[r][c];
            }
            MPC_UNROLL for (int i = 0; i < NR; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NR; j++) a += Ht[i][j] * rhs[j]; dy[i] = -a; }
            // step to the boundary = 1 / m with m = max_i (-d_i / x_i); predictor capped at 1 (m >= 1), corrector at 1 / tau
            double m = pass == 0 ? 1.0 : kTau, s1 = 0.0;
            MPC_UNROLL for (int r = 0; r < NC; r++) {
                double dv = 0.0; MPC_UNROLL for (int c = 0; c < NR; c++) dv += P.W[r][c] * dy[c];
                ds_hi[r] = fh[r] ? -r_hi[r] - dv : 0.0; ds_lo[r] = fl[r] ? r_lo[r] + dv : 0.0;
                dl_hi[r] = fh[r] ? (-rc_hi[r] - l_hi[r] * ds_hi[r]) * is_hi[r] : 0.0;
                dl_lo[r] = fl[r] ? (-rc_lo[r] - l_lo[r] * ds_lo[r]) * is_lo[r] : 0.0;
                m = dmax(m, dmax(-ds_lo[r] * is_lo[r], -ds_hi[r] * is_hi[r]));
                if (fl[r]) m = dmax(m, -dl_lo[r] * frcp_approx(l_lo[r]));
                if (fh[r]) m = dmax(m, -dl_hi[r] * frcp_approx(l_hi[r]));
            }
            if (pass == 0) {
                const double amax = frcp(m);
                MPC_UNROLL for (int r = 0; r < NC; r++) s1 += (s_lo[r] + amax * ds_lo[r]) * (l_lo[r] + amax * dl_lo[r]) + (s_hi[r] + amax * ds_hi[r]) * (l_hi[r] + amax * dl_hi[r]);
                const double mu_aff = s1 * inv_ncon, rat = mu > 0.0 ? mu_aff * frcp(mu) : 0.0;
                sm = dmax(rat * rat * rat * mu, kMuFloor);
            } else alpha = m <= kTau ? 1.0 : kTau * frcp(m);
        }
        MPC_UNROLL for (int c = 0; c < NR; c++) y[c] += alpha * dy[c];
        MPC_UNROLL for (int r = 0; r < NC; r++) { s_lo[r] += alpha * ds_lo[r]; s_hi[r] += alpha * ds_hi[r]; l_lo[r] += alpha * dl_lo[r]; l_hi[r] += alpha * dl_hi[r]; }
    }
    MPC_UNROLL for (int r = 0; r < NV; r++) {
        double a = vp[r]; MPC_UNROLL for (int c = 0; c < NR; c++) a += P.Zn[r][c] * y[c];
        if (r < NX) xs[r < NX ? r : 0] = a; else us[r >= NX ? r - NX : 0] = a;
    }
    MPC_UNROLL for (int i = 0; i < NY; i++) { double a = e[i]; MPC_UNROLL for (int j = 0; j < NX; j++) a += P.Cm[i][j] * xs[j]; ys[i] = a; }
    if (tw != nullptr) {
        *twv = status == kSolved ? 1 : 0;
        MPC_UNROLL for (int c = 0; c < NR; c++) { tw[c * tws] = y[c]; tw[(NR + 2 * NC + c) * tws] = gr[c]; }
        MPC_UNROLL for (int r = 0; r < NC; r++) { tw[(NR + r) * tws] = l_lo[r]; tw[(NR + NC + r) * tws] = l_hi[r]; tw[(2 * NR + 2 * NC + r) * tws] = w0[r]; }
    }
    return status;
}

// --------------------------------------------------------------------------------------------------------
// estimator: xi = [xhat; dhat], Pk row-major [NE][NE]; innov = y - yhat
// --------------------------------------------------------------------------------------------------------
// gain K for the prior covariance Pk, and the next prior (the part of the filter that does not see the data)
template <int NE, int NY, class PT>
__device__ void kalman_cov(const PT &P, double (&Pk)[NE][NE], double (&K)[NE][NY])
{
    double PCt[NE][NY], S[NY][NY];
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += Pk[i][l] * P.Ca[j][l]; PCt[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = P.Rkf[i][j]; MPC_UNROLL for (int l = 0; l < NE; l++) a += P.Ca[i][l] * PCt[l][j]; S[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (S[i][j] + S[j][i]); S[i][j] = a; S[j][i] = a; } }
    sym_inverse<NY>(S);                                   // K = P C' S^-1   (Estimator.py:297)
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) a += PCt[i][l] * S[l][j]; K[i][j] = a; } }
    double CP[NY][NE], Pc[NE][NE], T[NE][NE];
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += P.Ca[i][l] * Pk[l][j]; CP[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = Pk[i][j]; MPC_UNROLL for (int l = 0; l < NY; l++) a -= K[i][l] * CP[l][j]; Pc[i][j] = a; } }   // :300
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += P.Aa[i][l] * Pc[l][j]; T[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = P.Qkf[i][j]; MPC_UNROLL for (int l = 0; l < NE; l++) a += T[i][l] * P.Aa[j][l]; Pk[i][j] = a; } }  // :309
}

template <int NE, int NY, class PT>
__device__ void kalman_lane(const PT &P, double (&xi)[NE], double (&Pk)[NE][NE], const double (&innov)[NY])
{
    double K[NE][NY];
    kalman_cov<NE, NY>(P, Pk, K);
    MPC_UNROLL for (int i = 0; i < NE; i++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) a += K[i][l] * innov[l]; xi[i] += a; }                                              // :303-306
}

}  // namespace mpc
