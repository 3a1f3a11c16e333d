// Economic NMPC with a moving-horizon estimator on the GPU (BASELINE configs 4 and 5): device side.
//
// Mapping: ONE WAVEFRONT = ONE INSTANCE (or two / four side by side in segments of 32 / 16 lanes), LANE = STAGE of the horizon (N <= 64).  The iterate, the
// linearised stage and the factors of the Newton system live in registers; the solver's bound data - multipliers, moved bounds, slacks - in rows of the wave's LDS
// area (since round 4: in registers they were what spilled); stages talk to their neighbours through DPP wave shifts and v_readlane / ds_bpermute broadcasts; no HBM
// workspace.
//
// The three NLPs of a closed-loop step (reference MPC_code.py:485-827 with Ex_ENMPC.py) are solved by one primal-dual interior point
// method whose outer algorithm is the reference solver's (IPOPT at the defaults MPC_code.py:262-263 leaves it at [ext]; restated and
// documented in DESIGN.md section 10; the checker restates it with dense linear algebra): monotone barrier parameter, start pushed into the box, fraction to the
// boundary, exact Hessian of the Lagrangian with inertia correction, scaled optimality error.  What is particular here:
//   * OCP (opt_dyn with ContForm, Control_Calc.py:20-260): every lane integrates ITS shooting interval - state, cost quadrature and
//     their first and second forward sensitivities with respect to (x_k, u_k) through the Runge-Kutta stages (generated code,
//     econcodegen.py) - then the Newton system is factorised by a Riccati recursion over the lanes.  As a RECURSION (ric_backward: the estimator,
//     stage states beyond two, and the fallback) the stage update is computed by all lanes at once, lane k's result is the valid one and is
//     broadcast to the next iteration (v_readlane); since round 5 the OCP's matrix pass and its forward sweep are PARALLEL SCANS over the lanes
//     (ric_backward_scan, ric_forward: log2 of the lanes levels, the partner's element through ds_bpermute).  Either way
//     Lambda_k = R_k + B_k' P_{k+1} B_k > 0 is tested at every stage, which is IPOPT's inertia condition on the reduced Hessian: when it fails
//     the Hessian is shifted by delta I exactly as there.
//   * MHE (mhe_opt, Utilities.py:825-990): the same recursion with state [x; d], "input" w, a free initial state with the arrival
//     cost, and the output noise v eliminated through its (linear) defining equation.
//   * target (opt_ss, Target_Calc.py:20-161): nx + nu + ny variables; the model's fixed-point equation is eliminated with an LU of
//     (A - I) and the reduced Hessian is nu x nu; computed redundantly by every lane (wave-uniform).
#pragma once
#include <type_traits>
#ifdef EC_WAVE_EMU      // CPU test suite only (tests/wave_emu): the wave primitives on host fibers, so that this source runs - lane by lane - next to the oracle without a GPU
#include "wave_emu.hpp"
#define EC_VGPR_PIN(r) do { } while (0)
#define EC_ANY_HINT(p) ((p) != 0)      // (the fibers cannot vote inside a per-segment branch; the hint's answer for the lane itself is all its user needs)
#else
#include "mpc_tp.hpp"
#define EC_VGPR_PIN(r) asm("" : "+v"(r))
#define EC_LANE_INDEPENDENT_KERNEL
// a vote that only gates work every lane would decide for itself anyway ("does any lane of the wave need the rare path?"): the one cross-lane operation that may sit
// under a per-segment branch
#define EC_ANY_HINT(p) __any(p)
#endif
#define EC_UNI(x) mpc::uni(x)
#ifndef EC_SWEEP_SCAN_MAXNS
#define EC_SWEEP_SCAN_MAXNS 2      // stage states up to which the sweeps over the lanes run as parallel scans (ric_backward_scan, ric_forward)
#endif
#include "mpc_rk4s2.hpp"

namespace enm {
using namespace mpc;

// ---- the outer algorithm's constants (the checker carries the same) ----------------------------------------------------
constexpr double kBoundRelax = 1e-8;      // bound_relax_factor: every finite bound is relaxed by this times max(1, |bound|) before the solve, the final point projected back (honor_original_bounds)
constexpr double kPush = 1e-2, kMuInit = 0.1, kKappaEps = 10.0, kKappaMu = 0.2, kTauMin = 0.99, kKappaSigma = 1e10, kSMax = 100.0,
                 kDeltaFirst = 1e-4, kDeltaMax = 1e40;
enum : int { kStSolved = 0, kStMaxIter = 1, kStFailed = 2 };

__device__ __forceinline__ double wave_min(double v) { return -wave_max(-v); }
__device__ __forceinline__ bool finite_all(double v) { return fabs(v) < 1.0e300; }      // false for inf and NaN

__device__ __forceinline__ double relax_lo(double b) { return fin(b) ? b - kBoundRelax * dmax(1.0, fabs(b)) : b; }
__device__ __forceinline__ double relax_hi(double b) { return fin(b) ? b + kBoundRelax * dmax(1.0, fabs(b)) : b; }
__device__ __forceinline__ double push_in(double v, double lo, double hi)
{
    const bool fl = fin(lo), fh = fin(hi);
    const double gap = (fl && fh) ? kPush * (hi - lo) : INFINITY;
    const double pl = dmin(kPush * dmax(1.0, fabs(lo)), gap), ph = dmin(kPush * dmax(1.0, fabs(hi)), gap);
    if (fl) v = dmax(v, lo + pl);
    if (fh) v = dmin(v, hi - ph);
    return v;
}

template <int NP> __device__ __forceinline__ constexpr int pair_idx(int j, int k) { return j <= k ? j * NP - j * (j - 1) / 2 + (k - j) : k * NP - k * (k - 1) / 2 + (j - k); }

// ---- segments of a wave: SEG = 64 (one instance per wave), 32 or 16 lanes per instance (two or four instances side by side when the
// horizon leaves the lanes idle).  k = lane & (SEG - 1) is the stage; everything "uniform" is uniform within a segment. -------------------
template <int SEG>
struct Seg {
    static_assert(SEG == 64 || SEG == 32 || SEG == 16, "segments of 64, 32 or 16 lanes");
    __device__ static __forceinline__ int stage(int lane) { return lane & (SEG - 1); }
    // the neighbour stage's value inside the segment: stage 0 (resp. the last) keeps `old`
    __device__ static __forceinline__ double up1(double old, double v, int k) { const double s = wave_up1(old, v); return (SEG < 64 && k == 0) ? old : s; }
    __device__ static __forceinline__ double dn1(double old, double v, int k) { const double s = wave_dn1(old, v); return (SEG < 64 && k == SEG - 1) ? old : s; }
    // stage kk's value to every lane of its segment (kk wave-uniform)
    __device__ static __forceinline__ double bcast(double v, int kk, int lane)
    {
        if (SEG == 64) {
            // v_readlane leaves the value in scalar registers; it is moved to vector registers at once.  Not a matter of taste: with the
            // broadcasts of the recursions left in scalar registers (a few dozen doubles per stage, far more than the scalar file holds next
            // to the problem constants) five of six builds of the estimator kernel computed garbage from the first step on - iteration
            // counts like 314141150, not reproducible from run to run - whatever else changed (-O2, scheduler and spill options, the dense
            // recursion); with the move, or with ds_bpermute broadcasts instead, all of them are right (round 3, tools/enmpc_bcast_matrix.py).
            // (Not the partial-spill fault round 5 found and builds around - econcodegen.ENMPC_FLAGS -: round 3's builds were wrong with that option off too.)
            double r = lane_of(v, kk);
#ifndef EC_BCAST_IN_SGPRS      // (the matrix's failing leg)
            EC_VGPR_PIN(r);
#endif
            return r;
        }
        return __shfl(v, (lane & ~(SEG - 1)) + kk);
    }
    __device__ static __forceinline__ double max(double v)
    {
        if (SEG == 64) return wave_max(v);
        if (SEG == 32) return half_max(v);
        v = dmax(v, dpp_move<0xB1, 0xF>(v, v)); v = dmax(v, dpp_move<0x4E, 0xF>(v, v)); v = dmax(v, dpp_move<0x141, 0xF>(v, v)); v = dmax(v, dpp_move<0x140, 0xF>(v, v));
        return v;
    }
    __device__ static __forceinline__ double min(double v) { return -max(-v); }
    __device__ static __forceinline__ double sum(double v)
    {
        if (SEG == 64) return wave_sum(v);
        if (SEG == 32) return half_sum(v);
        v += dpp_move<0xB1, 0xF>(0.0, v); v += dpp_move<0x4E, 0xF>(0.0, v); v += dpp_move<0x141, 0xF>(0.0, v); v += dpp_move<0x140, 0xF>(0.0, v);
        return v;
    }
    __device__ static __forceinline__ bool any(bool p, int lane)
    {
        if (SEG == 64) return __any(p ? 1 : 0) != 0;
        const unsigned long long m = __ballot(p ? 1 : 0), segmask = SEG == 32 ? 0xFFFFFFFFull : 0xFFFFull;
        return ((m >> (lane & ~(SEG - 1))) & segmask) != 0ull;
    }
};

template <int SEG, int N_> __device__ __forceinline__ void bcast_vec(const double (&v)[N_], int kk, int lane, double (&o)[N_]) { MPC_UNROLL for (int i = 0; i < N_; i++) o[i] = Seg<SEG>::bcast(v[i], kk, lane); }
template <int SEG, int N_> __device__ __forceinline__ void bcast_sym(const double (&v)[N_][N_], int kk, int lane, double (&o)[N_][N_])
{
    MPC_UNROLL for (int i = 0; i < N_; i++) { MPC_UNROLL for (int j = i; j < N_; j++) { const double a = Seg<SEG>::bcast(v[i][j], kk, lane); o[i][j] = a; o[j][i] = a; } }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The interior point method on   min sum_k l_k(x_k, u_k) + Vf(x_N)   s.t.  x_{k+1} = F_k(x_k, u_k),  boxes on u_k and x_{k+1}
// (k = 0..N-1, lane k holds u_k, x_{k+1}, the costate pi_{k+1} = -(multiplier of x_{k+1} - F_k = 0), their bound multipliers and - since
// round 4 - ITS OWN COPY OF THEIR BOUNDS: the safe slack moves them).
// FREE0 = false: x_0 is given (the OCP; IPOPT makes a variable with equal bounds a parameter, MPC_code.py:734).
// FREE0 = true:  x_0 is a variable with its own box and the arrival cost 1/2 (x_0 - xbar)' Pinv (x_0 - xbar) (the MHE); every lane
//                carries the same copy of it.
// The algorithm is the reference solver's as documented in DESIGN.md section 10, "Solver" (objective scaling at the caller's point,
// least-squares multipliers, monotone barrier parameter, filter line search with second-order correction, tiny steps, safe slacks,
// acceptable stop) - same constants, same order of decisions; what differs is the linear algebra: every Newton system is a Riccati
// recursion over the lanes, split into a MATRIX pass (gains, cost-to-go matrices; repeated with a larger shift while a stage lacks positive
// curvature) and a VECTOR pass (feed-forward, cost-to-go gradient) + forward pass, which the second-order corrections repeat alone.
//   lin(xk, u, L, aux)     this lane's stage at (x_k, u_k): L.F, L.A, L.B, cost value L.l, cost gradient L.lx / L.lu, cost Hessian in L.Q / L.M / L.R;
//                          aux keeps the second-order sensitivities of the dynamics
//   addpi(aux, pi, L)      L.Q / L.M / L.R += sum_i pi_i Hessian(F_i)
//   val(xk, u, F, l)       values only (trial points of the line search)
//   term(xn, f, gv, Hv)    terminal cost at this lane's x_{k+1} (used from lane N-1)
// u / xn (/ x0v) come in as the caller's guess and leave as the final iterate.  filt: this segment's FILTER_CAP pairs of doubles in LDS.
// Returns the status.
// ---------------------------------------------------------------------------------------------------------------------------------
// ST: what is known at compile time about the stage matrices - a_kind(i, j) / b_kind(i, j) = 0 (the entry of A / B is zero), 1 (it is one), 2 (general: read
// L.A[i][j] / L.B[i][j]).  With the loops unrolled the products with zeros and ones disappear and the entries that are never read are never held.
struct DenseStage {
    __device__ static constexpr int a_kind(int, int) { return 2; }
    __device__ static constexpr int b_kind(int, int) { return 2; }
};
#define EC_A(acc, x, i_, j_) do { if (ST::a_kind(i_, j_) == 1) acc += (x); else if (ST::a_kind(i_, j_) == 2) acc += L.A[i_][j_] * (x); } while (0)
#define EC_B(acc, x, i_, j_) do { if (ST::b_kind(i_, j_) == 1) acc += (x); else if (ST::b_kind(i_, j_) == 2) acc += L.B[i_][j_] * (x); } while (0)

constexpr double kEps = 2.220446049250313e-16, kSlackMove = 1.81898940354585648e-12, kScaleMaxGrad = 100.0, kScaleMin = 1e-8, kYInitMax = 1e3, kKappaD = 1e-5,
                 kGammaTheta = 1e-5, kGammaPhi = 1e-8, kSTheta = 1.1, kSPhi = 2.3, kEtaPhi = 1e-8, kThetaMaxFact = 1e4, kThetaMinFact = 1e-4, kAlphaMinFrac = 0.05,
                 kObjMaxInc = 5.0, kKappaSoc = 0.99, kTinyStepTol = 10.0 * kEps, kTinyStepYTol = 1e-2, kDualInfTol = 1.0, kConstrViolTol = 1e-4, kComplInfTol = 1e-4,
                 kAccTol = 1e-6, kAccDualInfTol = 1e10, kAccConstrViolTol = 1e-2, kAccComplInfTol = 1e-2;
constexpr int kMaxSoc = 4, kAccIter = 15, kFilterCap = 16;

// Diagnostic build only (-DMPC_STAMPS, tools/enmpc_ipm_stamps.py): shader-clock ticks per part of an interior point iteration, accumulated per wave into
// mpc_ipm_stamp_buf[wave][16] by the first 256 waves of a launch.  Never compiled into the product library; the values reach no output of the solver.
#ifdef MPC_STAMPS
__device__ unsigned long long mpc_ipm_stamp_buf[256 * 16];
#define EC_IPM_STAMP_INIT unsigned long long ipm_prev_ = __builtin_amdgcn_s_memtime();
#define EC_IPM_STAMP(slot) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
    if (threadIdx.x == 0 && blockIdx.x < 256) mpc_ipm_stamp_buf[blockIdx.x * 16 + (slot)] += t_ - ipm_prev_; ipm_prev_ = t_; } while (0)
#else
#define EC_IPM_STAMP_INIT
#define EC_IPM_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ bool le_tol(double lhs, double rhs, double bas) { return lhs - rhs <= 10.0 * kEps * fabs(bas); }      // IPOPT's Compare_le
// slack of one bound with IPOPT's CalculateSafeSlack; a corrected slack moves `bound`
template <class BoundRef>      // (double & or a row of the LDS area)
__device__ __forceinline__ double safe_slack(double w, BoundRef &&bound, double z, double mu, bool lower)
{
    const double b = bound;
    double s = lower ? w - b : b - w;
    const double s_min = kEps * dmin(1.0, mu);
    if (__builtin_expect(EC_ANY_HINT(s < s_min ? 1 : 0), 0)) {      // (no lane of the wave in almost every call: the division is not even issued)
        if (s < s_min) {
            s = dmin(dmax(mu / z, s_min), dmax(s, 0.0) + kSlackMove * dmax(1.0, fabs(b)));
            bound = lower ? w - s : w + s;
        }
    }
    return s;
}
// A sum of logarithms as the logarithm of a product: mantissas multiplied, exponents added, ONE logarithm at the end - the barrier function of a stage with 16
// bounds costs one log instead of 16 (each some 60 instructions).  Mantissas are in [0.5, 1): a thousand factors stay clear of underflow.
struct LogSum {
    double m = 1.0; int e = 0;
    __device__ __forceinline__ void mul(double s, bool use = true) { int ei; const double f = frexp(s, &ei); m *= use ? f : 1.0; e += use ? ei : 0; }
    __device__ __forceinline__ double value() const { return log(m) + (double)e * 0.69314718055994530942; }
};
// min(1, min_i tau v_i / (-dv_i)) over the candidates with dv_i < 0 (fraction to the boundary) without a division per candidate: the smallest fraction is found by
// cross-multiplication, one division at the end.  The result is the quotient the candidate-by-candidate minimum would have picked, except between fractions that
// differ by rounding.
struct MinRatio {
    double n = 1.0, d = 1.0;
    __device__ __forceinline__ void add(double v, double dv, double tau, bool use = true) { const double ni = tau * v, di = -dv; if (use && di > 0.0 && ni * d < n * di) { n = ni; d = di; } }
    __device__ __forceinline__ double value() const { return n / d; }
};
// Per-lane arrays that live outside the register file: row r of the wave's LDS area is 64 doubles, one per lane (PS = 64) - or, in the rare-path solves that run with
// everything in private memory (restoration phase: ipm_stage MODE 1 / 2), a lane's own array (PS = 1).  REAL = false: the array does not exist (a variable class
// without bounds) - reads give `dflt`, writes vanish; with the loops unrolled nothing of it is left.
template <bool REAL>
struct LRow {
    double *p; double dflt;
    __device__ __forceinline__ operator double() const { return REAL ? *p : dflt; }
    __device__ __forceinline__ void operator=(double v) const { if (REAL) *p = v; }
    __device__ __forceinline__ void operator=(const LRow &o) const { if (REAL) *p = (double)o; }      // (row = row copies the VALUE, not the handle)
    __device__ __forceinline__ void operator+=(double v) const { if (REAL) *p = *p + v; }
    __device__ __forceinline__ LRow(double *p_, double d_) : p(p_), dflt(d_) {}
    LRow(const LRow &) = default;
};
template <bool REAL, int PS = 64>
struct LRows {
    double *base; double dflt;
    __device__ __forceinline__ LRow<REAL> operator[](int i) const { return LRow<REAL>(base + i * PS, dflt); }
};
// the compiler forgets what it knows of memory: values read from LDS before this point are read again after it instead of being kept in registers
#define EC_LDS_FENCE() asm volatile("" ::: "memory")
// the filter of one segment (= one instance) in LDS: pairs (phi, theta); every lane of the segment reads the same entries
template <int FS = 1>      // FS: stride between a list's words (1: a list of its own; 64: lists of the 64 lanes of a wave side by side in LDS)
__device__ __forceinline__ bool filter_rejects(const double *filt, int nf, double phi_t, double theta_t)
{
    bool rej = false;
    for (int e = 0; e < kFilterCap; e++) { if (e < nf) { const double ph = filt[(2 * e) * FS], th = filt[(2 * e + 1) * FS]; if (le_tol(ph, phi_t, ph) && le_tol(th, theta_t, th)) rej = true; } }
    return rej;
}
// the filter grows by (e_phi, e_th); entries the new one dominates are dropped; a full list merges it into its last entry.  Returns the new length.
template <int FS = 1>
__device__ __forceinline__ int filter_add(double *filt, int nf, double e_phi, double e_th)
{
    int k2 = 0;
    for (int e = 0; e < kFilterCap; e++) {
        if (e < nf) { const double ph = filt[(2 * e) * FS], th = filt[(2 * e + 1) * FS]; if (!(ph >= e_phi && th >= e_th)) { filt[(2 * k2) * FS] = ph; filt[(2 * k2 + 1) * FS] = th; k2++; } }
    }
    if (k2 >= kFilterCap) { filt[(2 * (k2 - 1)) * FS] = dmin(filt[(2 * (k2 - 1)) * FS], e_phi); filt[(2 * (k2 - 1) + 1) * FS] = dmin(filt[(2 * (k2 - 1) + 1) * FS], e_th); }
    else { filt[(2 * k2) * FS] = e_phi; filt[(2 * k2 + 1) * FS] = e_th; k2++; }
    return k2;
}

// rows of 64 doubles ipm_stage wants in LDS per wave (`park`)
template <int NS, int NU, bool FREE0, bool UB> constexpr int ipm_park_rows() { return 6 * ((UB ? NU : 0) + NS + (FREE0 ? NS : 0)) + NU + 2 * NS + (FREE0 ? NS : 0); }

template <int NS, int NU>
struct StageLin {
    double F[NS], A[NS][NS], B[NS][NU], l, lx[NS], lu[NU], Q[NS][NS], M[NS][NU], R[NU][NU];
};

// what a lane keeps of the factorised Newton system (K, Quu^-1, the cost-to-go behind its stage; P0i: the free initial state's block, segment-uniform)
// and of one solve with it (feed-forward, cost-to-go gradient behind its stage, step of the initial state)
template <int NS, int NU>
struct RicFac { double K[NU][NS], Qi[NU][NU], Pnx[NS][NS], P0i[NS][NS]; };
template <int NS, int NU>
struct RicVec { double kff[NU], pnx[NS], dx0[NS]; };

template <int n>
__device__ __forceinline__ bool gj_inverse(const double (&a_in)[n][n], double (&inv)[n][n])
{
    double a[n][n];
    MPC_UNROLL for (int i = 0; i < n; i++) { MPC_UNROLL for (int j = 0; j < n; j++) { a[i][j] = a_in[i][j]; inv[i][j] = (i == j) ? 1.0 : 0.0; } }
    bool ok = true;
    MPC_UNROLL for (int cidx = 0; cidx < n; cidx++) {
        MPC_UNROLL for (int r = cidx + 1; r < n; r++) {      // bring the largest entry of the column to the pivot row (conditional row swaps)
            const bool sw = fabs(a[r][cidx]) > fabs(a[cidx][cidx]);
            MPC_UNROLL for (int j = 0; j < n; j++) {
                const double t1 = a[cidx][j], t2 = a[r][j]; a[cidx][j] = sw ? t2 : t1; a[r][j] = sw ? t1 : t2;
                const double s1 = inv[cidx][j], s2 = inv[r][j]; inv[cidx][j] = sw ? s2 : s1; inv[r][j] = sw ? s1 : s2;
            }
        }
        const double pv = a[cidx][cidx];
        ok = ok && (fabs(pv) > 1e-300);
        const double ip = 1.0 / pv;
        MPC_UNROLL for (int j = 0; j < n; j++) { a[cidx][j] *= ip; inv[cidx][j] *= ip; }
        MPC_UNROLL for (int r = 0; r < n; r++) {
            if (r != cidx) {
                const double f = a[r][cidx];
                MPC_UNROLL for (int j = 0; j < n; j++) { a[r][j] -= f * a[cidx][j]; inv[r][j] -= f * inv[cidx][j]; }
            }
        }
    }
    return ok;
}


#ifndef EC_SWEEP_SERIAL
// ---- The matrix pass of the backward sweep as a PARALLEL SCAN over the lanes (stage states up to EC_SWEEP_SCAN_MAXNS: the OCP; -DEC_SWEEP_SERIAL builds the recursion alone).
// A stage is the element (A~, b~, C, eta, J) of its conditional value function  V(x, x+) = max_lam [ 1/2 x'J x + eta'x + lam'(A~ x + b~ - x+) - 1/2 lam'C lam ]
// (the stage's input eliminated: A~ = A - B Ri M', b~ = -c - B Ri gu, C = B Ri B', J = Qxx - M Ri M', eta = gx - M Ri gu with Ri = (R + Su)^-1); two neighbouring
// elements combine associatively [Saerkkae, Garcia-Fernandez: Temporal parallelization of dynamic programming and linear quadratic control, 2023]:
//     D = (I + C1 J2)^-1,  A = A2 D A1,  b = A2 D (b1 - C1 eta2) + b2,  C = A2 D C1 A2' + C2,  J = A1' J2 D A1 + J1,  eta = A1'(eta2 + J2 D (b1 - C1 eta2)) + eta1
// and the combination of the stages k .. N-1 with the terminal element (0, 0, 0, pt, Pt) carries the cost-to-go (P_k, p_k) in its (J, eta).  A suffix scan over the lanes
// (log2 SEG levels, the partner's element through ds_bpermute) gives every lane its P_{k+1}, from which it computes ITS gain and feed-forward by the recursion's own
// formulas - so the factors (K, Qi, Pnx, kff, pnx) mean what they mean after the serial sweep and the vector passes and the forward sweep take them as they are.
// Needs R_k + Su_k positive definite at EVERY stage (the recursion needs Lambda_k = R_k + Su_k + B'P_{k+1}B only): returns -1 where that fails and the caller runs
// the recursion; and a free lane for the terminal element (N < SEG).  Values part from the recursion's by rounding.
template <int NS> struct ScanEl { double A[NS][NS], b[NS], C[NS][NS], e[NS], J[NS][NS]; };

template <int NS, int NU, bool FREE0, int SEG, class ST>
__device__ __forceinline__ int ric_backward_scan(const int N, const int lane, const int k, const StageLin<NS, NU> &L, const double (&Q)[NS][NS], const double (&Mx)[NS][NU], const double (&R)[NU][NU],
                                                 const double (&Su)[NU], const double (&Sxk)[NS], const double (&Pt)[NS][NS], const double (&P0add)[NS][NS],
                                                 const double (&gu)[NU], const double (&gxk)[NS], const double (&pt)[NS], const double (&p0add)[NS], const double (&c)[NS],
                                                 RicFac<NS, NU> &Fc, RicVec<NS, NU> &Vc)
{
    using SG = Seg<SEG>;
    const bool stage = k < N;
    double Ad[NS][NS], Bd[NS][NU], Ri[NU][NU];
    MPC_UNROLL for (int i = 0; i < NS; i++) {
        MPC_UNROLL for (int j = 0; j < NS; j++) Ad[i][j] = ST::a_kind(i, j) == 0 ? 0.0 : (ST::a_kind(i, j) == 1 ? 1.0 : L.A[i][j]);
        MPC_UNROLL for (int j = 0; j < NU; j++) Bd[i][j] = ST::b_kind(i, j) == 0 ? 0.0 : (ST::b_kind(i, j) == 1 ? 1.0 : L.B[i][j]);
    }
    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) Ri[i][j] = 0.5 * (R[i][j] + R[j][i]) + (i == j ? Su[i] : 0.0); }
    const bool okr = sym_inverse<NU>(Ri);
    if (__any((stage && !okr) ? 1 : 0)) return -1;      // (the whole wave: its segments stay in step)
    ScanEl<NS> e;
    {
        double G[NU][NS], rg[NU], BR[NS][NU];
        MPC_UNROLL for (int a = 0; a < NU; a++) {
            MPC_UNROLL for (int j = 0; j < NS; j++) { double s_ = 0.0; MPC_UNROLL for (int b = 0; b < NU; b++) s_ += Ri[a][b] * Mx[j][b]; G[a][j] = s_; }
            double s_ = 0.0; MPC_UNROLL for (int b = 0; b < NU; b++) s_ += Ri[a][b] * gu[b]; rg[a] = s_;
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int a = 0; a < NU; a++) { double s_ = 0.0; MPC_UNROLL for (int b = 0; b < NU; b++) s_ += Bd[i][b] * Ri[b][a]; BR[i][a] = s_; } }
        double Ptb[NS][NS], ptb[NS];
        bcast_sym<SEG, NS>(Pt, N - 1, lane, Ptb);
        bcast_vec<SEG, NS>(pt, N - 1, lane, ptb);
        const bool term = k == N;
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            MPC_UNROLL for (int j = 0; j < NS; j++) {
                double a_ = Ad[i][j], c_ = 0.0, j_ = Q[i][j] + (i == j ? Sxk[i] : 0.0);
                MPC_UNROLL for (int a = 0; a < NU; a++) { a_ -= Bd[i][a] * G[a][j]; c_ += BR[i][a] * Bd[j][a]; j_ -= Mx[i][a] * G[a][j]; }
                e.A[i][j] = stage ? a_ : (term ? 0.0 : (i == j ? 1.0 : 0.0));
                e.C[i][j] = stage ? c_ : 0.0;
                e.J[i][j] = stage ? j_ : (term ? Ptb[i][j] : 0.0);
            }
            double b_ = -c[i], e_ = gxk[i];
            MPC_UNROLL for (int a = 0; a < NU; a++) { b_ -= Bd[i][a] * rg[a]; e_ -= Mx[i][a] * rg[a]; }
            e.b[i] = stage ? b_ : 0.0;
            e.e[i] = stage ? e_ : (term ? ptb[i] : 0.0);
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a_ = 0.5 * (e.J[i][j] + e.J[j][i]); e.J[i][j] = a_; e.J[j][i] = a_; const double c_ = 0.5 * (e.C[i][j] + e.C[j][i]); e.C[i][j] = c_; e.C[j][i] = c_; } }
    }
    const int base = lane & ~(SEG - 1);
    auto from = [&](double v, int kk) { return __shfl(v, base + (kk < SEG ? kk : k)); };      // stage kk's value (this segment), own value beyond the segment
    for (int d = 1; d < SEG; d <<= 1) {
        const bool has = k + d < SEG;
        ScanEl<NS> o;
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            MPC_UNROLL for (int j = 0; j < NS; j++) {
                const double a_ = from(e.A[i][j], k + d); o.A[i][j] = has ? a_ : (i == j ? 1.0 : 0.0);
                if (j >= i) { const double c_ = from(e.C[i][j], k + d), j_ = from(e.J[i][j], k + d); o.C[i][j] = has ? c_ : 0.0; o.J[i][j] = has ? j_ : 0.0; }
            }
            const double b_ = from(e.b[i], k + d), e_ = from(e.e[i], k + d);
            o.b[i] = has ? b_ : 0.0; o.e[i] = has ? e_ : 0.0;
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { o.C[i][j] = o.C[j][i]; o.J[i][j] = o.J[j][i]; } }
        // e <- e (x) o
        double Dm[NS][NS], Di[NS][NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double s_ = (i == j) ? 1.0 : 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) s_ += e.C[i][l] * o.J[l][j]; Dm[i][j] = s_; } }
        gj_inverse<NS>(Dm, Di);
        double y[NS][NS], DC[NS][NS], t[NS], Jy[NS][NS], w_[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            MPC_UNROLL for (int j = 0; j < NS; j++) { double s1 = 0.0, s2 = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) { s1 += Di[i][l] * e.A[l][j]; s2 += Di[i][l] * e.C[l][j]; } y[i][j] = s1; DC[i][j] = s2; }
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { double s_ = e.b[i]; MPC_UNROLL for (int l = 0; l < NS; l++) s_ -= e.C[i][l] * o.e[l]; w_[i] = s_; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { double s_ = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) s_ += Di[i][l] * w_[l]; t[i] = s_; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double s_ = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) s_ += o.J[i][l] * y[l][j]; Jy[i][j] = s_; } }
        ScanEl<NS> r;
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            MPC_UNROLL for (int j = 0; j < NS; j++) {
                double a_ = 0.0, j_ = e.J[i][j];
                MPC_UNROLL for (int l = 0; l < NS; l++) { a_ += o.A[i][l] * y[l][j]; j_ += e.A[l][i] * Jy[l][j]; }
                r.A[i][j] = a_; r.J[i][j] = j_;
            }
            double b_ = o.b[i], h_ = o.e[i];
            MPC_UNROLL for (int l = 0; l < NS; l++) { b_ += o.A[i][l] * t[l]; h_ += o.J[i][l] * t[l]; }
            r.b[i] = b_; w_[i] = h_;      // w_ <- eta2 + J2 t
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { double s_ = e.e[i]; MPC_UNROLL for (int l = 0; l < NS; l++) s_ += e.A[l][i] * w_[l]; r.e[i] = s_; }
        {
            double ADC[NS][NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double s_ = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) s_ += o.A[i][l] * DC[l][j]; ADC[i][j] = s_; } }
            MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double s_ = o.C[i][j]; MPC_UNROLL for (int l = 0; l < NS; l++) s_ += ADC[i][l] * o.A[j][l]; r.C[i][j] = s_; } }
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a_ = 0.5 * (r.J[i][j] + r.J[j][i]); r.J[i][j] = a_; r.J[j][i] = a_; const double c_ = 0.5 * (r.C[i][j] + r.C[j][i]); r.C[i][j] = c_; r.C[j][i] = c_; } }
        e = r;
    }
    // the cost-to-go behind this lane's stage, then its own factors by the recursion's formulas
    double Pn[NS][NS], pn[NS];
    MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = i; j < NS; j++) { const double a_ = from(e.J[i][j], k + 1); Pn[i][j] = a_; Pn[j][i] = a_; } pn[i] = from(e.e[i], k + 1); }
    MPC_UNROLL for (int i = 0; i < NU; i++) Vc.kff[i] = 0.0;
    MPC_UNROLL for (int i = 0; i < NS; i++) Vc.pnx[i] = 0.0;
    double Kl[NU][NS], Qil[NU][NU];
    bool ok = true;
    {
        double PA[NS][NS], PB[NS][NU];
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) EC_A(a, Pn[i][l], l, j); PA[i][j] = a; }
            MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, Pn[i][l], l, j); PB[i][j] = a; }
        }
        double Qux[NU][NS];
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            MPC_UNROLL for (int j = 0; j < NU; j++) { double a = R[i][j] + (i == j ? Su[i] : 0.0); MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, PB[l][j], l, i); Qil[i][j] = a; }
            MPC_UNROLL for (int j = 0; j < NS; j++) { double a = Mx[j][i]; MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, PA[l][j], l, i); Qux[i][j] = a; }
        }
        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (Qil[i][j] + Qil[j][i]); Qil[i][j] = a; Qil[j][i] = a; } }
        ok = sym_inverse<NU>(Qil);
        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < i; j++) Qil[i][j] = Qil[j][i]; }
        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NU; l++) a -= Qil[i][l] * Qux[l][j]; Kl[i][j] = a; } }
    }
    if (stage) {
        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Fc.K[i][j] = Kl[i][j]; MPC_UNROLL for (int j = 0; j < NU; j++) Fc.Qi[i][j] = Qil[i][j]; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Fc.Pnx[i][j] = Pn[i][j]; }
        double pc[NS], qu[NU];
        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = pn[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a -= Pn[i][l] * c[l]; pc[i] = a; }
        MPC_UNROLL for (int i = 0; i < NU; i++) { double a = gu[i]; MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, pc[l], l, i); qu[i] = a; }
        MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NU; l++) a -= Qil[i][l] * qu[l]; Vc.kff[i] = a; }
        MPC_UNROLL for (int i = 0; i < NS; i++) Vc.pnx[i] = pn[i];
    }
    bool bad = SG::any(stage && !ok, lane);
    if (FREE0) {
        double P0[NS][NS], p0[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = i; j < NS; j++) { const double a_ = from(e.J[i][j], 0) + P0add[i][j]; P0[i][j] = a_; P0[j][i] = a_; } p0[i] = from(e.e[i], 0); }
        if (!sym_inverse<NS>(P0)) bad = true;
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Fc.P0i[i][j] = j < i ? P0[j][i] : P0[i][j]; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) a -= Fc.P0i[i][j] * (p0[j] + p0add[j]); Vc.dx0[i] = a; }
    } else { MPC_UNROLL for (int i = 0; i < NS; i++) Vc.dx0[i] = 0.0; }
    return bad ? 0 : 1;
}
#endif

// BACKWARD sweep of the Riccati recursion over the lanes: all lanes compute the stage update with the broadcast cost-to-go of the stage behind them, lane kk's
// result is the valid one and is broadcast on.  MATRIX = true: gains and cost-to-go matrices from the Hessian blocks (Q, M, R) of this lane's stage, the
// diagonal terms Su (inputs), Sxk (x_k, from the neighbour that holds it), the terminal block Pt of lane N-1, the initial state's own block P0add - and,
// in the same sweep, the vector part; false: the vector part alone with the stored factors (second-order corrections).  Vector part: gradient terms gu
// (inputs), gxk (x_k), pt (lane N-1: x_N), p0add (x_0) and the constraint residual c.  Returns false for a lane whose stage lacks positive curvature.
template <int NS, int NU, bool FREE0, int SEG, class ST, bool MATRIX>
__device__ __forceinline__ bool ric_backward(const int N, const int lane, const int k, const StageLin<NS, NU> &L, const double (&Q)[NS][NS], const double (&Mx)[NS][NU], const double (&R)[NU][NU],
                                             const double (&Su)[NU], const double (&Sxk)[NS], const double (&Pt)[NS][NS], const double (&P0add)[NS][NS],
                                             const double (&gu)[NU], const double (&gxk)[NS], const double (&pt)[NS], const double (&p0add)[NS], const double (&c)[NS],
                                             RicFac<NS, NU> &Fc, RicVec<NS, NU> &Vc)
{
#ifndef EC_SWEEP_SERIAL
    if (MATRIX && NS <= EC_SWEEP_SCAN_MAXNS && N < SEG) {
        const int r_ = ric_backward_scan<NS, NU, FREE0, SEG, ST>(N, lane, k, L, Q, Mx, R, Su, Sxk, Pt, P0add, gu, gxk, pt, p0add, c, Fc, Vc);
        if (r_ >= 0) return r_ != 0;      // (-1: a stage without curvature of its own: the recursion below)
    }
#endif
    double Pn[NS][NS], pn[NS];
    if (MATRIX) bcast_sym<SEG, NS>(Pt, N - 1, lane, Pn);
    bcast_vec<SEG, NS>(pt, N - 1, lane, pn);
    MPC_UNROLL for (int i = 0; i < NU; i++) Vc.kff[i] = 0.0;      // (lanes beyond the horizon never receive theirs: zero, not indeterminate)
    MPC_UNROLL for (int i = 0; i < NS; i++) Vc.pnx[i] = 0.0;
    bool bad = false;
    for (int kk = N - 1; kk >= 0; kk--) {
        double Kl[NU][NS], Qil[NU][NU], Pk[NS][NS];
        bool ok = true;
        if (MATRIX) {
            double PA[NS][NS], PB[NS][NU];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) EC_A(a, Pn[i][l], l, j); PA[i][j] = a; }
                MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, Pn[i][l], l, j); PB[i][j] = a; }
            }
            double Qux[NU][NS], Qxx[NS][NS];
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                MPC_UNROLL for (int j = 0; j < NU; j++) { double a = R[i][j] + (i == j ? Su[i] : 0.0); MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, PB[l][j], l, i); Qil[i][j] = a; }
                MPC_UNROLL for (int j = 0; j < NS; j++) { double a = Mx[j][i]; MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, PA[l][j], l, i); Qux[i][j] = a; }
            }
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                MPC_UNROLL for (int j = 0; j < NS; j++) { double a = Q[i][j] + (i == j ? Sxk[i] : 0.0); MPC_UNROLL for (int l = 0; l < NS; l++) EC_A(a, PA[l][j], l, i); Qxx[i][j] = a; }
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (Qil[i][j] + Qil[j][i]); Qil[i][j] = a; Qil[j][i] = a; } }
            ok = sym_inverse<NU>(Qil);
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < i; j++) Qil[i][j] = Qil[j][i]; }      // (one value per pair: the inverse is held once)
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NU; l++) a -= Qil[i][l] * Qux[l][j]; Kl[i][j] = a; } }
            MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { double a = Qxx[i][j]; MPC_UNROLL for (int l = 0; l < NU; l++) a += Qux[l][i] * Kl[l][j]; Pk[i][j] = a; } }
            MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (Pk[i][j] + Pk[j][i]); Pk[i][j] = a; Pk[j][i] = a; } }
            if (k == kk) {
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Fc.K[i][j] = Kl[i][j]; MPC_UNROLL for (int j = 0; j < NU; j++) Fc.Qi[i][j] = Qil[i][j]; }
                MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Fc.Pnx[i][j] = Pn[i][j]; }
                if (!ok) bad = true;
            }
        }
        // vector part: every lane with ITS stored factors (lane kk's are the ones of this stage)
        double pc[NS], qu[NU], kl[NU], pk[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = pn[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a -= (MATRIX ? Pn[i][l] : Fc.Pnx[i][l]) * c[l]; pc[i] = a; }
        MPC_UNROLL for (int i = 0; i < NU; i++) { double a = gu[i]; MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, pc[l], l, i); qu[i] = a; }
        MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NU; l++) a -= (MATRIX ? Qil[i][l] : Fc.Qi[i][l]) * qu[l]; kl[i] = a; }
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            double a = gxk[i];
            MPC_UNROLL for (int l = 0; l < NS; l++) EC_A(a, pc[l], l, i);
            MPC_UNROLL for (int l = 0; l < NU; l++) a += (MATRIX ? Kl[l][i] : Fc.K[l][i]) * qu[l];      // Qux' kl = K' qu
            pk[i] = a;
        }
        if (k == kk) { MPC_UNROLL for (int i = 0; i < NU; i++) Vc.kff[i] = kl[i]; MPC_UNROLL for (int i = 0; i < NS; i++) Vc.pnx[i] = pn[i]; }
        if (MATRIX) bcast_sym<SEG, NS>(Pk, kk, lane, Pn);
        bcast_vec<SEG, NS>(pk, kk, lane, pn);
    }
    if (FREE0) {      // the initial state: value function of stage 0 + arrival cost + its own diagonal term
        if (MATRIX) {
            double P0[NS][NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) P0[i][j] = Pn[i][j] + P0add[i][j]; }
            if (!sym_inverse<NS>(P0)) bad = true;
            MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Fc.P0i[i][j] = j < i ? P0[j][i] : P0[i][j]; }
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) a -= Fc.P0i[i][j] * (pn[j] + p0add[j]); Vc.dx0[i] = a; }
    } else { MPC_UNROLL for (int i = 0; i < NS; i++) Vc.dx0[i] = 0.0; }
    return !bad;
}

// FORWARD sweep: the Newton step (du, dxn; dx0 in Vc) and the new costates pin
template <int NS, int NU, int SEG, class ST>
__device__ __forceinline__ void ric_forward(const int N, const int lane, const int k, const StageLin<NS, NU> &L, const RicFac<NS, NU> &Fc, const RicVec<NS, NU> &Vc, const double (&c)[NS],
                                            double (&du)[NU], double (&dxn)[NS], double (&pin)[NS])
{
#if !defined(EC_SWEEP_SERIAL) && !defined(EC_FWD_SERIAL)
    if (NS <= EC_SWEEP_SCAN_MAXNS) {
        // the forward sweep as a prefix scan: stage k is the affine map dx -> Acl_k dx + r_k (Acl = A + B K, r = B kff - c); the composition of the stages 0..k applied to
        // dx_0 is this lane's dx_{k+1} (log2 SEG levels, the partner's map through ds_bpermute).  Values part from the recursion's by rounding.
        using SG = Seg<SEG>;
        const bool stage = k < N;
        double Am[NS][NS], rm[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            MPC_UNROLL for (int j = 0; j < NS; j++) {
                double a = ST::a_kind(i, j) == 0 ? 0.0 : (ST::a_kind(i, j) == 1 ? 1.0 : L.A[i][j]);
                MPC_UNROLL for (int l = 0; l < NU; l++) EC_B(a, Fc.K[l][j], i, l);
                Am[i][j] = stage ? a : (i == j ? 1.0 : 0.0);
            }
            double a = -c[i];
            MPC_UNROLL for (int l = 0; l < NU; l++) EC_B(a, Vc.kff[l], i, l);
            rm[i] = stage ? a : 0.0;
        }
        const int base = lane & ~(SEG - 1);
        for (int d = 1; d < SEG; d <<= 1) {
            const bool has = k >= d;
            const int src = base + (has ? k - d : k);
            double Ap[NS][NS], rp[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                MPC_UNROLL for (int j = 0; j < NS; j++) { const double a = __shfl(Am[i][j], src); Ap[i][j] = has ? a : (i == j ? 1.0 : 0.0); }
                const double a = __shfl(rm[i], src); rp[i] = has ? a : 0.0;
            }
            double An[NS][NS], rn[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) a += Am[i][l] * Ap[l][j]; An[i][j] = a; }
                double a = rm[i]; MPC_UNROLL for (int l = 0; l < NS; l++) a += Am[i][l] * rp[l]; rn[i] = a;
            }
            MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Am[i][j] = An[i][j]; rm[i] = rn[i]; }
        }
        double dxk[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = rm[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += Am[i][j] * Vc.dx0[j]; dxn[i] = stage ? a : 0.0; }
        MPC_UNROLL for (int i = 0; i < NS; i++) dxk[i] = SG::up1(Vc.dx0[i], dxn[i], k);
        MPC_UNROLL for (int i = 0; i < NU; i++) { double a = Vc.kff[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += Fc.K[i][j] * dxk[j]; du[i] = stage ? a : 0.0; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = Vc.pnx[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += Fc.Pnx[i][j] * dxn[j]; pin[i] = a; }
        return;
    }
#endif
    double dx[NS];
    MPC_UNROLL for (int i = 0; i < NS; i++) { dx[i] = Vc.dx0[i]; dxn[i] = 0.0; }
    MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = 0.0;
    for (int kk = 0; kk < N; kk++) {
        double dul[NU], dxl[NS];
        MPC_UNROLL for (int i = 0; i < NU; i++) { double a = Vc.kff[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += Fc.K[i][j] * dx[j]; dul[i] = a; }
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            double a = -c[i];
            MPC_UNROLL for (int j = 0; j < NS; j++) EC_A(a, dx[j], i, j);
            MPC_UNROLL for (int j = 0; j < NU; j++) EC_B(a, dul[j], i, j);
            dxl[i] = a;
        }
        if (k == kk) { MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = dul[i]; MPC_UNROLL for (int i = 0; i < NS; i++) dxn[i] = dxl[i]; }
        bcast_vec<SEG, NS>(dxl, kk, lane, dx);
    }
    MPC_UNROLL for (int i = 0; i < NS; i++) { double a = Vc.pnx[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += Fc.Pnx[i][j] * dxn[j]; pin[i] = a; }
}

//   grd(xk, u, L)          cost value / gradient only (L.l, L.lx, L.lu): the caller's point, where IPOPT takes the scaling of the objective from
// (every functor takes the barrier parameter as its last argument: the restoration problem's objective depends on it, the callers' problems ignore it)
//
// MODE 0: the solve without the restoration phase - the form the hot kernels carry.  A line search that runs out of step lengths at an infeasible point returns
//         kStNeedResto: the caller hands the instance, untouched, to the rare path (MODE 1) - some solves in 1e5 on the benchmark workloads.
// MODE 1: the complete algorithm: such a line search enters IPOPT's RESTORATION PHASE [WB 3.3; IpRestoMinC_1Nrm, IpRestoIpoptNLP, IpRestoFilterConvCheck]:
//             min  rho sum(n + p) + sqrt(mu) / 2 |D_R (w - w_R)|^2   s.t.  x_{k+1} = F_k(x_k, u_k) - n_k + p_k,  the boxes,  n, p >= 0
//         which is a problem of THIS function's own form with the defects as 2 NS more inputs per stage - solved by MODE 2 (same iteration, same recursion over
//         the lanes, inputs [u; n; p], B~ = [B, -I, I]) until the point is enough less infeasible and acceptable to this solve's filter and iterate.
//         Bound data lives in the lane's private memory (PS = 1): nothing here is sized for the hot path.
// MODE 2: the restoration problem's iteration: first iterate, multipliers and barrier parameter given (ext.mu0, rows of `park` filled by the caller), no scaling,
//         the caller's test (xt.hook) ends it, its objective is recomposed after every change of mu (xt.recost); returns kRs*.
enum : int { kStNeedResto = 3 };
struct IpmNoExtra {};
struct IpmNoAux0 {};      // aux0(xk, u, xn): states that are FUNCTIONS of the first iterate (the slack states of user rows) get their first values from the pushed point
struct IpmRestoWs { double *park_r, *filt_r; };      // MODE 1: private areas of the inner solve (ipm_park_rows<NS, NU + 2 NS, FREE0, true>() doubles; 2 kFilterCap doubles)
template <class HookF, class RecostF> struct IpmRestoIn { double mu0; int it0; HookF hook; RecostF recost; };      // MODE 2
// the rows of the bound data (multipliers Z, bounds B, slacks S; L / H lower / upper; U inputs, X states x_{k+1}, 0 the free initial state)
template <int NS, int NU, bool FREE0, bool UB>
struct ParkRows {
    static constexpr int NUB = UB ? NU : 0, N0 = FREE0 ? NS : 0, NBV = NUB + NS + N0;
    static constexpr int ZLU = 0, BLU = ZLU + NUB, ZHU = BLU + NUB, BHU = ZHU + NUB, ZLX = BHU + NUB, BLX = ZLX + NS, ZHX = BLX + NS, BHX = ZHX + NS,
                         ZL0 = BHX + NS, BL0 = ZL0 + N0, ZH0 = BL0 + N0, BH0 = ZH0 + N0, SLU = BH0 + N0, SHU = SLU + NUB, SLX = SHU + NUB, SHX = SLX + NS,
                         SL0 = SHX + NS, SH0 = SL0 + N0, IT = SH0 + N0;
    static_assert(IT == 6 * NBV, "rows of the bound data");
};
constexpr double kRestoRho = 1000.0, kRestoKappa = 0.9, kRestoThetaMaxFact = 1e8, kRestoBoundMultReset = 1e3, kRestoFeasFact = 1e2;
enum : int { kRsRestored = 0, kRsConverged = 1, kRsLimit = 2, kRsFailed = 3 };

template <int NS, int NU, bool FREE0, bool UB, int SEG, class ST, class AUX, int MODE = 0, int PS = 64, class GrdF, class LinF, class AddPiF, class ValF, class TermF, class XT = IpmNoExtra, class Aux0F = IpmNoAux0>
__device__ __forceinline__ int ipm_stage(const int N, const int lane, const bool live, const double (&x0fix)[NS], double (&x0v)[NS], double (&u)[NU], double (&xn)[NS],
                                         double (&pi)[NS], const double (&ulo_in)[NU], const double (&uhi_in)[NU], const double (&xlo_in)[NS],
                                         const double (&xhi_in)[NS], const double (*Pinv)[NS], const double *xbar, const double tol,
                                         const int max_iter, GrdF grd, LinF lin, AddPiF addpi, ValF val, TermF term, int &iters, double *const filt, double *const park, XT ext = XT(), Aux0F aux0 = Aux0F())
{
    static_assert(MODE == 0 || PS == 1, "the rare-path solves keep their bound data in private memory");
    // what the solve's events are called: a status word of the caller's (kSt*), or - restoration problem - what its caller makes of it (kRs*)
    constexpr int S_FAIL = MODE == 2 ? (int)kRsFailed : (int)kStFailed, S_CONV = MODE == 2 ? (int)kRsConverged : (int)kStSolved, S_LIMIT = MODE == 2 ? (int)kRsLimit : (int)kStMaxIter,
                  S_TINY = MODE == 2 ? (int)kRsConverged : (int)kStMaxIter;
    using SG = Seg<SEG>;
    // Large stages (the estimator: 4 + 4 here) do not keep the factors of the Newton system across the line search - with the trial point's integration they
    // would not fit the register file, and what the compiler then moves to scratch memory is paid for in every iteration.  The second-order correction, which
    // alone needs them again (one iteration in some hundred), runs the matrix pass once more.
    constexpr bool kSocRefactor = NS * NS + NU * NS >= 24;
    const int k = SG::stage(lane);
    const bool on = k < N;
    bool flu[NU], fhu[NU], flx[NS], fhx[NS];
    // Bound multipliers z, this solve's own bounds b (the safe slack moves them) and the slacks s of the iterate LIVE IN LDS (`park`: rows of 64 lanes, this
    // lane's column; ipm_park_rows), for the inputs (u; UB), the states x_{k+1} (x) and the free initial state (0; FREE0).  They are read where they are used:
    // kept in registers (round 3; then parked across the linearisation and the sweeps) they were what the compiler moved to scratch memory around every
    // larger piece of an iteration - 400 scratch accesses per iteration of the estimator at one wave per SIMD, each waited for.
    using PR = ParkRows<NS, NU, FREE0, UB>;
    constexpr int R_IT = PR::IT;
    double *const pk = park + (PS == 64 ? lane : 0);
    const LRows<UB, PS> zlu{pk + PS * PR::ZLU, 0.0}, blu{pk + PS * PR::BLU, -INFINITY}, zhu{pk + PS * PR::ZHU, 0.0}, bhu{pk + PS * PR::BHU, INFINITY}, slu{pk + PS * PR::SLU, 1.0}, shu{pk + PS * PR::SHU, 1.0};
    const LRows<true, PS> zlx{pk + PS * PR::ZLX, 0.0}, blx{pk + PS * PR::BLX, -INFINITY}, zhx{pk + PS * PR::ZHX, 0.0}, bhx{pk + PS * PR::BHX, INFINITY}, slx{pk + PS * PR::SLX, 1.0}, shx{pk + PS * PR::SHX, 1.0};
    const LRows<FREE0, PS> zl0{pk + PS * PR::ZL0, 0.0}, bl0{pk + PS * PR::BL0, -INFINITY}, zh0{pk + PS * PR::ZH0, 0.0}, bh0{pk + PS * PR::BH0, INFINITY}, sl0{pk + PS * PR::SL0, 1.0}, sh0{pk + PS * PR::SH0, 1.0};
    int nbl = 0, nbx = 0;
    MPC_UNROLL for (int i = 0; i < NU; i++) {
        flu[i] = UB && fin(ulo_in[i]); fhu[i] = UB && fin(uhi_in[i]); nbl += (flu[i] ? 1 : 0) + (fhu[i] ? 1 : 0);
        if (MODE != 2) { blu[i] = relax_lo(ulo_in[i]); bhu[i] = relax_hi(uhi_in[i]); zlu[i] = flu[i] ? 1.0 : 0.0; zhu[i] = fhu[i] ? 1.0 : 0.0; }      // (MODE 2: the caller has filled the rows of bounds and multipliers)
    }
    MPC_UNROLL for (int i = 0; i < NS; i++) {
        flx[i] = fin(xlo_in[i]); fhx[i] = fin(xhi_in[i]); nbx += (flx[i] ? 1 : 0) + (fhx[i] ? 1 : 0); pi[i] = 0.0;
        if (MODE != 2) {
            blx[i] = relax_lo(xlo_in[i]); bhx[i] = relax_hi(xhi_in[i]); bl0[i] = relax_lo(xlo_in[i]); bh0[i] = relax_hi(xhi_in[i]);
            zlx[i] = flx[i] ? 1.0 : 0.0; zhx[i] = fhx[i] ? 1.0 : 0.0;
            zl0[i] = (FREE0 && flx[i]) ? 1.0 : 0.0; zh0[i] = (FREE0 && fhx[i]) ? 1.0 : 0.0;
        }
    }
    // The iterate itself is parked there across the sweeps (it is not touched between the linearisation and the first trial point).
    auto park_iter = [&](const bool store) {
        if (PS != 64) return;      // (private memory: nothing to gain)
        int slot = R_IT;
        auto one = [&](double &v) { if (store) park[slot * 64 + lane] = v; else v = park[slot * 64 + lane]; slot++; };
        MPC_UNROLL for (int i = 0; i < NU; i++) one(u[i]);
        MPC_UNROLL for (int i = 0; i < NS; i++) { one(xn[i]); one(pi[i]); if (FREE0) one(x0v[i]); }
    };
    const double nb = (double)(N * (nbl + nbx) + (FREE0 ? nbx : 0)), meq = (double)(N * NS);
    // damping of the variables with one bound [WB 3.7]: +1 (only a lower bound), -1 (only an upper bound), 0
    auto damp = [&](bool fl_, bool fh_) { return (fl_ && !fh_) ? 1.0 : ((fh_ && !fl_) ? -1.0 : 0.0); };
    // everything below that looks wave-uniform is uniform per SEGMENT (= per instance): mu, the shifts, the line search's state, status, the iteration
    // count.  A segment that has finished (done) keeps computing with its frozen iterate while its wave neighbours go on; `live` = false marks a
    // segment without an instance (ragged batch).  Cross-lane operations (shifts, broadcasts, reductions) are never under a per-segment branch.
    double mu = kMuInit, tau = dmax(kTauMin, 1.0 - kMuInit), delta_last = 0.0, df = 1.0, theta_max = -1.0, theta_min = -1.0;
    if constexpr (MODE == 2) { mu = ext.mu0; tau = dmax(kTauMin, 1.0 - mu); }
    const double mu_min = dmin(tol, kComplInfTol) / (kKappaEps + 1.0);
    int status = S_LIMIT, nfilt = 0, acc_count = 0;
    bool done = !live, tiny_last = false, tiny_flag = false;
    iters = 0;
    EC_IPM_STAMP_INIT
    // ---- scaling of the objective at the caller's point (IpGradientScaling), then the push into the box --------------------------------------------
    if constexpr (MODE != 2) {
        double xk[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) xk[i] = SG::up1(FREE0 ? x0v[i] : x0fix[i], xn[i], k);
        StageLin<NS, NU> L;
        grd(xk, u, L, mu);
        double fv, gv[NS], Hv[NS][NS], gm = 0.0;
        term(xn, fv, gv, Hv, mu);
        MPC_UNROLL for (int i = 0; i < NU; i++) gm = dmax(gm, fabs(L.lu[i]));
        MPC_UNROLL for (int i = 0; i < NS; i++) { const double sh = SG::dn1(0.0, L.lx[i], k); gm = dmax(gm, fabs(k == N - 1 ? gv[i] : sh)); }
        gm = SG::max(on ? gm : 0.0);
        if (FREE0) {
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                double a = SG::bcast(L.lx[i], 0, lane);
                MPC_UNROLL for (int j = 0; j < NS; j++) a += Pinv[i][j] * (x0v[j] - xbar[j]);
                gm = dmax(gm, fabs(a));
            }
        }
        df = gm > kScaleMaxGrad ? dmax(kScaleMaxGrad / gm, kScaleMin) : 1.0;
        MPC_UNROLL for (int i = 0; i < NU; i++) u[i] = push_in(u[i], blu[i], bhu[i]);
        MPC_UNROLL for (int i = 0; i < NS; i++) xn[i] = push_in(xn[i], blx[i], bhx[i]);
        if (FREE0) { MPC_UNROLL for (int i = 0; i < NS; i++) x0v[i] = push_in(x0v[i], bl0[i], bh0[i]); }
        if constexpr (!std::is_same<Aux0F, IpmNoAux0>::value) {
            // IPOPT's own slacks (of inequality rows) start from the rows' values at the PUSHED first iterate and are then pushed into their bound themselves
            // (IpDefaultIterateInitializer [ext]): the states that stand for them here
            double xkp[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) xkp[i] = SG::up1(FREE0 ? x0v[i] : x0fix[i], xn[i], k);
            aux0(xkp, u, xn);
            MPC_UNROLL for (int i = 0; i < NS; i++) xn[i] = push_in(xn[i], blx[i], bhx[i]);
        }
    }
    EC_IPM_STAMP(0);      // scaling, push
    bool first = MODE != 2;      // (wave-uniform: every segment's first iteration starts with the least-squares multipliers)
    int it = 0;
    if constexpr (MODE == 2) it = ext.it0;
    for (;;) {
        if (!done) iters = it;
        // ---- linearise this lane's stage at (x_k, u_k); x_k is the neighbour's x_{k+1} ------------------------------------------------
        double xk[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) xk[i] = SG::up1(FREE0 ? x0v[i] : x0fix[i], xn[i], k);
        StageLin<NS, NU> L;
        AUX aux;
        lin(xk, u, L, aux, mu);
        double fv, gv[NS], Hv[NS][NS];
        term(xn, fv, gv, Hv, mu);
        EC_IPM_STAMP(1);      // linearisation
        EC_LDS_FENCE();
        // the scaled problem: df f.  (An entry the generated code knows to be zero stays a literal zero - 0 * df would be a run-time value to the compiler, and the
        // sweeps below would multiply and keep in registers what the estimator's structure - constant diagonal cost Hessian, no cross terms - lets them drop.)
        auto scaled = [&](double x_) { return (__builtin_constant_p(x_) && x_ == 0.0) ? 0.0 : x_ * df; };
        if constexpr (MODE != 2) {      // (the restoration problem is not scaled)
            L.l *= df; fv *= df;
            MPC_UNROLL for (int i = 0; i < NU; i++) { L.lu[i] = scaled(L.lu[i]); MPC_UNROLL for (int j = 0; j < NU; j++) L.R[i][j] = scaled(L.R[i][j]); }
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                L.lx[i] = scaled(L.lx[i]); gv[i] = scaled(gv[i]);
                MPC_UNROLL for (int j = 0; j < NS; j++) { L.Q[i][j] = scaled(L.Q[i][j]); Hv[i][j] = scaled(Hv[i][j]); }
                MPC_UNROLL for (int j = 0; j < NU; j++) L.M[i][j] = scaled(L.M[i][j]);
            }
        }
        // gradient of the objective with respect to this lane's x_{k+1}: the next stage's cost gradient, the terminal cost's at the end; with respect to the
        // free initial state: stage 0's cost gradient + the arrival cost's (ga0)
        double gfx[NS], ga0[NS], lx0[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) { const double sh = SG::dn1(0.0, L.lx[i], k); gfx[i] = k == N - 1 ? gv[i] : sh; }
        double farr = 0.0;
        if (FREE0) {
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                double a = 0.0;
                MPC_UNROLL for (int j = 0; j < NS; j++) a += Pinv[i][j] * (x0v[j] - xbar[j]);
                farr += 0.5 * (x0v[i] - xbar[i]) * a;
                ga0[i] = df * a; lx0[i] = SG::bcast(L.lx[i], 0, lane);
            }
            farr *= df;
        } else { MPC_UNROLL for (int i = 0; i < NS; i++) { ga0[i] = 0.0; lx0[i] = 0.0; } }
        double c[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) c[i] = xn[i] - L.F[i];
        RicFac<NS, NU> Fc;
        RicVec<NS, NU> Vc;
        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Fc.K[i][j] = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) Fc.Qi[i][j] = 0.0; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { Fc.Pnx[i][j] = 0.0; Fc.P0i[i][j] = 0.0; } }
        EC_IPM_STAMP(2);      // scaling of the stage, gradients
        if (__builtin_expect(first, 0)) {
            // ---- least-squares equality multipliers [WB (36)]: the Newton system with the identity for the Hessian, no constraint residual ------------
            double Iq[NS][NS], Im[NS][NU], Ir[NU][NU], zu_[NU], zx_[NS], gu[NU], gxk[NS], pt[NS], p0a[NS], c0[NS], P0a[NS][NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { Iq[i][j] = i == j ? 1.0 : 0.0; P0a[i][j] = i == j ? 1.0 : 0.0; } MPC_UNROLL for (int j = 0; j < NU; j++) Im[i][j] = 0.0; zx_[i] = 0.0; c0[i] = 0.0; }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) Ir[i][j] = i == j ? 1.0 : 0.0; zu_[i] = 0.0; gu[i] = L.lu[i] - zlu[i] + zhu[i]; }
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                const double bz = -zlx[i] + zhx[i];
                gxk[i] = L.lx[i] + SG::up1(0.0, bz, k); pt[i] = gv[i] + bz;
                p0a[i] = FREE0 ? (ga0[i] - zl0[i] + zh0[i]) : 0.0;
            }
            ric_backward<NS, NU, FREE0, SEG, ST, true>(N, lane, k, L, Iq, Im, Ir, zu_, zx_, Iq, P0a, gu, gxk, pt, p0a, c0, Fc, Vc);
            double du_[NU], dxn_[NS], pin_[NS];
            ric_forward<NS, NU, SEG, ST>(N, lane, k, L, Fc, Vc, c0, du_, dxn_, pin_);
            double ym = 0.0;
            MPC_UNROLL for (int i = 0; i < NS; i++) ym = dmax(ym, finite_all(pin_[i]) ? fabs(pin_[i]) : INFINITY);
            ym = SG::max(on ? ym : 0.0);
            MPC_UNROLL for (int i = 0; i < NS; i++) pi[i] = ym <= kYInitMax ? pin_[i] : 0.0;
            first = false;
        }
        EC_IPM_STAMP(3);      // least-squares multipliers (first iteration)
        addpi(aux, pi, L);      // Hessian of the Lagrangian of the scaled problem: df (cost) + pi' F
        // ---- slacks (safe: a slack that rounding took below eps min(1, mu) is lifted, its bound moves) ------------------------------------------
        MPC_UNROLL for (int i = 0; i < NU; i++) { slu[i] = flu[i] ? safe_slack(u[i], blu[i], zlu[i], mu, true) : 1.0; shu[i] = fhu[i] ? safe_slack(u[i], bhu[i], zhu[i], mu, false) : 1.0; }
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            slx[i] = flx[i] ? safe_slack(xn[i], blx[i], zlx[i], mu, true) : 1.0; shx[i] = fhx[i] ? safe_slack(xn[i], bhx[i], zhx[i], mu, false) : 1.0;
            sl0[i] = (FREE0 && flx[i]) ? safe_slack(x0v[i], bl0[i], zl0[i], mu, true) : 1.0; sh0[i] = (FREE0 && fhx[i]) ? safe_slack(x0v[i], bh0[i], zh0[i], mu, false) : 1.0;
        }
        // ---- optimality error (IPOPT's E_mu with its scaling), objective, infeasibility -----------------------------------------------------------
        double gxA[NS], gnext[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = L.lx[i]; MPC_UNROLL for (int j = 0; j < NS; j++) EC_A(a, pi[j], j, i); gxA[i] = a; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { const double sh = SG::dn1(0.0, gxA[i], k); gnext[i] = k == N - 1 ? gv[i] : sh; }
        double e_st = 0.0, e_c = 0.0, s_pi = 0.0, s_z = 0.0, cmax = -INFINITY, cmin = INFINITY, th_l = 0.0;
        bool finite = finite_all(L.l);
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            double r = L.lu[i] - zlu[i] + zhu[i];
            MPC_UNROLL for (int j = 0; j < NS; j++) EC_B(r, pi[j], j, i);
            e_st = dmax(e_st, fabs(r)); s_z += zlu[i] + zhu[i];
            finite = finite && finite_all(r) && finite_all(u[i]);
            if (flu[i]) { cmax = dmax(cmax, slu[i] * zlu[i]); cmin = dmin(cmin, slu[i] * zlu[i]); }
            if (fhu[i]) { cmax = dmax(cmax, shu[i] * zhu[i]); cmin = dmin(cmin, shu[i] * zhu[i]); }
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            const double r = -pi[i] + gnext[i] - zlx[i] + zhx[i];
            e_st = dmax(e_st, fabs(r)); e_c = dmax(e_c, fabs(c[i])); th_l += fabs(c[i]); s_pi += fabs(pi[i]); s_z += zlx[i] + zhx[i];
            finite = finite && finite_all(r) && finite_all(c[i]) && finite_all(xn[i]);
            if (flx[i]) { cmax = dmax(cmax, slx[i] * zlx[i]); cmin = dmin(cmin, slx[i] * zlx[i]); }
            if (fhx[i]) { cmax = dmax(cmax, shx[i] * zhx[i]); cmin = dmin(cmin, shx[i] * zhx[i]); }
        }
        e_st = SG::max(on ? e_st : 0.0); e_c = SG::max(on ? e_c : 0.0); s_pi = SG::sum(on ? s_pi : 0.0); s_z = SG::sum(on ? s_z : 0.0);
        cmax = SG::max(on ? cmax : -INFINITY); cmin = SG::min(on ? cmin : INFINITY);
        const double theta = SG::sum(on ? th_l : 0.0);
        double fobj = SG::sum(on ? L.l + (k == N - 1 ? fv : 0.0) : 0.0) + farr;
        if (FREE0) {
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                const double r0 = SG::bcast(gxA[i], 0, lane) + ga0[i] - zl0[i] + zh0[i];      // stage 0's part of the gradient with respect to the free initial state + arrival cost
                e_st = dmax(e_st, fabs(r0)); s_z += zl0[i] + zh0[i];
                finite = finite && finite_all(r0) && finite_all(x0v[i]);
                if (flx[i]) { cmax = dmax(cmax, sl0[i] * zl0[i]); cmin = dmin(cmin, sl0[i] * zl0[i]); }
                if (fhx[i]) { cmax = dmax(cmax, sh0[i] * zh0[i]); cmin = dmin(cmin, sh0[i] * zh0[i]); }
            }
        }
        const bool nonfinite = SG::any(on && !finite, lane);
        const double s_d = dmax(kSMax, (s_pi + s_z) / dmax(meq + nb, 1.0)) / kSMax, s_c = dmax(kSMax, s_z / dmax(nb, 1.0)) / kSMax;
        auto compl_ = [&](double m_) { return nb > 0.0 ? dmax(cmax - m_, m_ - cmin) : 0.0; };
        auto err = [&](double m_) { return dmax(dmax(e_st / s_d, e_c), compl_(m_) / s_c); };
#ifdef EC_EMU_TRACE      /* diagnostic build of the CPU test suite's emulator only: the iterations of the first lane's segment */
        if (lane == 0 && getenv("EC_EMU_TRACE")) fprintf(stderr, "%c it=%3d mu=%.3e E0=%.6e e_st=%.3e e_c=%.3e compl=%.3e theta=%.6e f=%.9e done=%d\n", MODE == 2 ? 'R' : (MODE == 1 ? '1' : ' '), it, mu, err(0.0), e_st, e_c, compl_(0.0), theta, fobj / df, (int)done);
#endif
        bool hooked = false;      // restoration problem: the test of the solve that called (it holds cross-lane sums: taken by every lane, used per segment)
        if constexpr (MODE == 2) hooked = ext.hook(u, xn, x0v);
        if (!done) {
            const double e0_ = err(0.0), c0_ = compl_(0.0);
            if (nonfinite) { status = S_FAIL; done = true; }
            else if (MODE == 2 && hooked) { status = kRsRestored; done = true; }
            else if (e0_ <= tol && e_st <= kDualInfTol && e_c <= kConstrViolTol && c0_ <= kComplInfTol) { status = S_CONV; done = true; }
            else {
                if (e0_ <= kAccTol && e_st <= kAccDualInfTol && e_c <= kAccConstrViolTol && c0_ <= kAccComplInfTol) { if (++acc_count >= kAccIter) { status = S_CONV; done = true; } }
                else acc_count = 0;
                if (!done && it >= max_iter) done = true;
            }
        }
        EC_IPM_STAMP(4);      // slacks, optimality error, stopping tests
        if (__all(done ? 1 : 0)) break;
        // ---- barrier parameter (per segment): while (E_mu <= kappa_eps mu or two tiny steps in a row) mu decreases; the filter is emptied with it ---------
        {
            bool mu_changed = false, stop_tiny = false, going = !done;
            for (;;) {
                const bool dec = going && (err(mu) <= kKappaEps * mu || tiny_flag);
                if (!__any(dec ? 1 : 0)) break;
                if (dec) {
                    const double new_mu = dmax(dmin(kKappaMu * mu, mu * sqrt(mu)), mu_min);
                    if (new_mu == mu) { stop_tiny = tiny_flag; going = false; }
                    else { mu = new_mu; mu_changed = true; tiny_flag = false; }
                }
            }
            if (!done && stop_tiny) { status = S_TINY; done = true; }      // 'Search_Direction_Becomes_Too_Small': the reference accepts the point
            tiny_flag = false;
            if (mu_changed) { nfilt = 0; tau = dmax(kTauMin, 1.0 - mu); }
        }
        if constexpr (MODE == 2) {      // the restoration problem's objective changes with mu: cost terms, Hessian of the Lagrangian, the gradients taken from them, the objective's value
            ext.recost(xk, u, L, mu);
            addpi(aux, pi, L);
            term(xn, fv, gv, Hv, mu);
            MPC_UNROLL for (int i = 0; i < NS; i++) { const double sh = SG::dn1(0.0, L.lx[i], k); gfx[i] = k == N - 1 ? gv[i] : sh; if (FREE0) lx0[i] = SG::bcast(L.lx[i], 0, lane); }
            fobj = SG::sum(on ? L.l + (k == N - 1 ? fv : 0.0) : 0.0) + farr;
        }
        // ---- barrier terms; those of x_k come from the neighbour that holds x_k --------------------------------------------------------
        double Su[NU], bu[NU], Sx[NS], bx[NS], Sxk[NS], bxk[NS], S0[NS], b0[NS];
        double phl = 0.0;      // this lane's part of the barrier function
        LogSum lgl, lg0;
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            const double il = flu[i] ? 1.0 / slu[i] : 0.0, ih = fhu[i] ? 1.0 / shu[i] : 0.0, dm = damp(flu[i], fhu[i]);
            Su[i] = zlu[i] * il + zhu[i] * ih; bu[i] = -mu * il + mu * ih + kKappaD * mu * dm;
            lgl.mul(slu[i], flu[i]); lgl.mul(shu[i], fhu[i]);
            if (dm != 0.0) phl += kKappaD * mu * (dm > 0.0 ? slu[i] : shu[i]);
        }
        double ph0 = 0.0;
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            const double il = flx[i] ? 1.0 / slx[i] : 0.0, ih = fhx[i] ? 1.0 / shx[i] : 0.0, dm = damp(flx[i], fhx[i]);
            Sx[i] = zlx[i] * il + zhx[i] * ih; bx[i] = -mu * il + mu * ih + kKappaD * mu * dm;
            lgl.mul(slx[i], flx[i]); lgl.mul(shx[i], fhx[i]);
            if (dm != 0.0) phl += kKappaD * mu * (dm > 0.0 ? slx[i] : shx[i]);
            Sxk[i] = SG::up1(0.0, Sx[i], k); bxk[i] = SG::up1(0.0, bx[i], k);
            const double jl = (FREE0 && flx[i]) ? 1.0 / sl0[i] : 0.0, jh = (FREE0 && fhx[i]) ? 1.0 / sh0[i] : 0.0;
            S0[i] = zl0[i] * jl + zh0[i] * jh; b0[i] = FREE0 ? (-mu * jl + mu * jh + kKappaD * mu * dm) : 0.0;
            if (FREE0) {
                lg0.mul(sl0[i], flx[i]); lg0.mul(sh0[i], fhx[i]);
                if (dm != 0.0) ph0 += kKappaD * mu * (dm > 0.0 ? sl0[i] : sh0[i]);
            }
        }
        phl -= mu * lgl.value();
        if (FREE0) ph0 -= mu * lg0.value();
        const double phi = fobj + SG::sum(on ? phl : 0.0) + ph0;
        // ---- Newton direction: backward sweep (repeated with a larger shift while a stage lacks positive curvature), forward sweep -----------------
        double gu[NU], gxk[NS], pt[NS], p0a[NS];
        MPC_UNROLL for (int i = 0; i < NU; i++) gu[i] = L.lu[i] + bu[i];
        MPC_UNROLL for (int i = 0; i < NS; i++) { gxk[i] = L.lx[i] + bxk[i]; pt[i] = gv[i] + bx[i]; p0a[i] = FREE0 ? ga0[i] + b0[i] : 0.0; }
        EC_IPM_STAMP(5);      // barrier parameter, barrier terms
        park_iter(true); EC_LDS_FENCE();
        double delta = 0.0;
        bool failed = false;
        for (;;) {
            double Qd[NS][NS], Rd[NU][NU], Pt[NS][NS], P0a[NS][NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) {
                Qd[i][j] = L.Q[i][j] + (i == j ? delta : 0.0); Pt[i][j] = Hv[i][j] + (i == j ? Sx[i] + delta : 0.0);
                P0a[i][j] = FREE0 ? (df * 0.5 * (Pinv[i][j] + Pinv[j][i]) + (i == j ? S0[i] : 0.0)) : 0.0; } }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) Rd[i][j] = L.R[i][j] + (i == j ? delta : 0.0); }
            const bool okm = ric_backward<NS, NU, FREE0, SEG, ST, true>(N, lane, k, L, Qd, L.M, Rd, Su, Sxk, Pt, P0a, gu, gxk, pt, p0a, c, Fc, Vc);
            const bool retry = SG::any(!okm, lane) && !done && !failed;      // this segment lacks curvature: a larger shift, all over again
            if (retry) {
                delta = delta == 0.0 ? dmax(kDeltaFirst, delta_last / 3.0) : delta * (delta_last == 0.0 ? 100.0 : 8.0);
                if (delta > kDeltaMax) failed = true;
            }
            if (__builtin_expect(!__any((retry && !failed) ? 1 : 0), 1)) break;      // (segments that were fine recompute the same numbers with their own shift)
        }
        if (!done && failed) { status = S_FAIL; done = true; }
        if (!done && delta > 0.0) delta_last = delta;
        EC_IPM_STAMP(6);      // backward sweep(s)
        double du[NU], dxn[NS], dx0[NS], pin[NS];
        ric_forward<NS, NU, SEG, ST>(N, lane, k, L, Fc, Vc, c, du, dxn, pin);
        MPC_UNROLL for (int i = 0; i < NS; i++) dx0[i] = Vc.dx0[i];
        EC_LDS_FENCE(); park_iter(false);
        EC_IPM_STAMP(7);      // forward sweep
        // fraction to the boundary of a step (du_, dxn_, dx0_)
        auto max_step = [&](const double (&du_)[NU], const double (&dxn_)[NS], const double (&dx0_)[NS]) {
            MinRatio mr;
            MPC_UNROLL for (int i = 0; i < NU; i++) { mr.add(slu[i], du_[i], tau, flu[i]); mr.add(shu[i], -du_[i], tau, fhu[i]); }
            MPC_UNROLL for (int i = 0; i < NS; i++) { mr.add(slx[i], dxn_[i], tau, flx[i]); mr.add(shx[i], -dxn_[i], tau, fhx[i]); }
            double a = SG::min(on ? mr.value() : 1.0);
            if (FREE0) {
                MinRatio m0;
                MPC_UNROLL for (int i = 0; i < NS; i++) { m0.add(sl0[i], dx0_[i], tau, flx[i]); m0.add(sh0[i], -dx0_[i], tau, fhx[i]); }
                a = dmin(a, m0.value());
            }
            return a;
        };
        const double a_max = max_step(du, dxn, dx0);
        // directional derivative of the barrier function, size of the step, size of the multiplier step
        double gbd_l = 0.0, drel = 0.0, dym = 0.0;
        MPC_UNROLL for (int i = 0; i < NU; i++) { gbd_l += gu[i] * du[i]; drel = dmax(drel, fabs(du[i]) / (1.0 + fabs(u[i]))); }
        MPC_UNROLL for (int i = 0; i < NS; i++) { gbd_l += (gfx[i] + bx[i]) * dxn[i]; drel = dmax(drel, fabs(dxn[i]) / (1.0 + fabs(xn[i]))); dym = dmax(dym, fabs(pin[i] - pi[i])); }
        double gbd = SG::sum(on ? gbd_l : 0.0);
        drel = SG::max(on ? drel : 0.0); dym = SG::max(on ? dym : 0.0);
        if (FREE0) { MPC_UNROLL for (int i = 0; i < NS; i++) { gbd += (lx0[i] + ga0[i] + b0[i]) * dx0[i]; drel = dmax(drel, fabs(dx0[i]) / (1.0 + fabs(x0v[i]))); } }
        // ---- filter line search (per segment, the wave in lockstep) -------------------------------------------------------------------------------
        // the switching condition [WB (19)] alpha (-gbd)^s_phi > theta^s_theta as alpha > sw = theta^s_theta / (-gbd)^s_phi, the quotient through two logarithms and an
        // exponential (a third of two pow calls); theta = 0 gives 0, gbd -> 0 gives inf, both zero NaN: the comparisons below come out as with the powers
        const double sw = gbd < 0.0 ? exp(kSTheta * log(theta) - kSPhi * log(-gbd)) : INFINITY;
        double a_min = kGammaTheta;
        if (gbd < 0.0) {
            a_min = dmin(kGammaTheta, kGammaPhi * theta / (-gbd));
            if (theta <= theta_min) a_min = dmin(a_min, sw);
        }
        a_min *= kAlphaMinFrac;
        if (theta_max < 0.0) { theta_max = (MODE == 2 ? kRestoThetaMaxFact : kThetaMaxFact) * dmax(1.0, theta); theta_min = kThetaMinFact * dmax(1.0, theta); }
        auto ftype = [&](double alpha_) { return (theta == 0.0 && gbd > 0.0 && gbd < 100.0 * kEps) || (gbd < 0.0 && alpha_ > sw); };
        // a trial point u + a_ du_ ...: infeasibility, barrier function (safe slacks; their moved bounds are not kept), constraint values
        double theta_t = 0.0, phi_t = 0.0, ct[NS];
        bool fin_t = false;
        auto eval_point = [&](const double (&ut)[NU], const double (&xt)[NS], const double (&x0t)[NS]) {
            double xkt[NS], Ft[NS], lt, fvt, gvt[NS], Hvt[NS][NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) xkt[i] = SG::up1(FREE0 ? x0t[i] : x0fix[i], xt[i], k);
            val(xkt, ut, Ft, lt, mu);
            term(xt, fvt, gvt, Hvt, mu);
            EC_LDS_FENCE();      // (multipliers and bounds are read again rather than kept alive across the integration)
            double tht = 0.0, pht = 0.0;
            LogSum lgt, lgt0;
            bool okl = finite_all(lt);
            MPC_UNROLL for (int i = 0; i < NS; i++) { ct[i] = xt[i] - Ft[i]; tht += fabs(ct[i]); okl = okl && finite_all(ct[i]); }
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                double bl_ = blu[i], bh_ = bhu[i];
                const double dm = damp(flu[i], fhu[i]);
                const double s1 = flu[i] ? safe_slack(ut[i], bl_, zlu[i], mu, true) : 1.0, s2 = fhu[i] ? safe_slack(ut[i], bh_, zhu[i], mu, false) : 1.0;
                lgt.mul(s1, flu[i]); lgt.mul(s2, fhu[i]);
                if (dm != 0.0) pht += kKappaD * mu * (dm > 0.0 ? s1 : s2);
            }
            double pht0 = 0.0, farrt = 0.0;
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                double bl_ = blx[i], bh_ = bhx[i];
                const double dm = damp(flx[i], fhx[i]);
                const double s1 = flx[i] ? safe_slack(xt[i], bl_, zlx[i], mu, true) : 1.0, s2 = fhx[i] ? safe_slack(xt[i], bh_, zhx[i], mu, false) : 1.0;
                lgt.mul(s1, flx[i]); lgt.mul(s2, fhx[i]);
                if (dm != 0.0) pht += kKappaD * mu * (dm > 0.0 ? s1 : s2);
                if (FREE0) {
                    double cl_ = bl0[i], ch_ = bh0[i];
                    const double t1 = flx[i] ? safe_slack(x0t[i], cl_, zl0[i], mu, true) : 1.0, t2 = fhx[i] ? safe_slack(x0t[i], ch_, zh0[i], mu, false) : 1.0;
                    lgt0.mul(t1, flx[i]); lgt0.mul(t2, fhx[i]);
                    if (dm != 0.0) pht0 += kKappaD * mu * (dm > 0.0 ? t1 : t2);
                }
            }
            if (FREE0) {
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) a += Pinv[i][j] * (x0t[j] - xbar[j]); farrt += 0.5 * (x0t[i] - xbar[i]) * a; }
            }
            pht -= mu * lgt.value();
            if (FREE0) pht0 -= mu * lgt0.value();
            const bool ok_t = !SG::any(on && !okl, lane);
            theta_t = SG::sum(on ? tht : 0.0);
            phi_t = df * (SG::sum(on ? lt + (k == N - 1 ? fvt : 0.0) : 0.0) + farrt) + SG::sum(on ? pht : 0.0) + pht0;
            fin_t = ok_t && finite_all(phi_t);
            if (!fin_t) { theta_t = INFINITY; phi_t = INFINITY; }
        };
        auto trial = [&](const double a_, const double (&du_)[NU], const double (&dxn_)[NS], const double (&dx0_)[NS]) {
            double ut[NU], xt[NS], x0t[NS];
            MPC_UNROLL for (int i = 0; i < NU; i++) ut[i] = u[i] + a_ * du_[i];
            MPC_UNROLL for (int i = 0; i < NS; i++) { xt[i] = xn[i] + a_ * dxn_[i]; x0t[i] = FREE0 ? x0v[i] + a_ * dx0_[i] : 0.0; }
            eval_point(ut, xt, x0t);
        };
        // acceptable to the current iterate and to the filter?  (alpha_: the step length of the Newton step, also for a corrected step)
        auto acceptable = [&](double alpha_) {
            if (!fin_t || theta_t > theta_max) return false;
            bool ok_;
            if (alpha_ > 0.0 && ftype(alpha_) && theta <= theta_min) ok_ = le_tol(phi_t - phi, kEtaPhi * alpha_ * gbd, phi);
            else {
                bool too_steep = false;
                if (phi_t > phi) { const double bas = fabs(phi) > 10.0 ? log10(fabs(phi)) : 1.0; too_steep = log10(phi_t - phi) > kObjMaxInc + bas; }
                ok_ = !too_steep && (le_tol(theta_t, (1.0 - kGammaTheta) * theta, theta) || le_tol(phi_t - phi, -kGammaPhi * theta, phi));
            }
            return ok_ && !filter_rejects(filt, nfilt, phi_t, theta_t);
        };
        EC_IPM_STAMP(8);      // step sizes, directional derivative, the line search's thresholds
        bool tiny = drel < kTinyStepTol && theta <= 1e-4;
        bool searching = !done, accepted = false;
        int n_steps = 0;
        double alpha = a_max, a_pr = a_max, phi_acc = 0.0;
        for (;;) {
            if (!__any(searching ? 1 : 0)) break;
            trial(alpha, du, dxn, dx0);
            EC_IPM_STAMP(9);      // trial points
            const bool acc = searching && acceptable(alpha);
            bool want_soc = false;
            if (searching) {
                if (tiny && n_steps == 0 && fin_t) { accepted = true; searching = false; phi_acc = phi_t; a_pr = alpha; }      // a tiny step is taken unchecked
                else {
                    if (tiny && n_steps == 0) tiny = false;
                    if (acc) { accepted = true; searching = false; phi_acc = phi_t; a_pr = alpha; }
                    else if (fin_t && n_steps == 0 && theta <= theta_t) want_soc = true;      // the first trial step did not reduce the infeasibility
                    else { alpha *= 0.5; n_steps++; if (!(alpha > a_min)) searching = false; }
                }
            }
            if (__builtin_expect(__any(want_soc ? 1 : 0), 0)) {      // (rare: the hint keeps its operands from crowding the common path's registers)
                // ---- second-order correction [WB 2.4]: the vector sweeps again for the corrected constraint residual, up to four times -----------------
                EC_IPM_STAMP(10);      // acceptance tests
                double csoc[NS], a_soc = alpha, theta_old = 0.0, th_s = theta_t;
                MPC_UNROLL for (int i = 0; i < NS; i++) csoc[i] = c[i];
                int cnt = 0;
                bool soc = want_soc;
                for (;;) {
                    if (!__any(soc ? 1 : 0)) break;
                    if (soc) { theta_old = th_s; MPC_UNROLL for (int i = 0; i < NS; i++) csoc[i] = a_soc * csoc[i] + ct[i]; }
                    RicVec<NS, NU> Vs;
                    double dsu[NU], dsx[NS], ds0[NS], pins[NS];
                    if (kSocRefactor) {      // the matrix pass again, with the shift the Newton step was computed with
                        RicFac<NS, NU> Fs;
                        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Fs.K[i][j] = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) Fs.Qi[i][j] = 0.0; }
                        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) { Fs.Pnx[i][j] = 0.0; Fs.P0i[i][j] = 0.0; } }
                        double Qd[NS][NS], Rd[NU][NU], Pt[NS][NS], P0a[NS][NS];
                        MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) {
                            Qd[i][j] = L.Q[i][j] + (i == j ? delta : 0.0); Pt[i][j] = Hv[i][j] + (i == j ? Sx[i] + delta : 0.0);
                            P0a[i][j] = FREE0 ? (df * 0.5 * (Pinv[i][j] + Pinv[j][i]) + (i == j ? S0[i] : 0.0)) : 0.0; } }
                        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) Rd[i][j] = L.R[i][j] + (i == j ? delta : 0.0); }
                        ric_backward<NS, NU, FREE0, SEG, ST, true>(N, lane, k, L, Qd, L.M, Rd, Su, Sxk, Pt, P0a, gu, gxk, pt, p0a, csoc, Fs, Vs);
                        ric_forward<NS, NU, SEG, ST>(N, lane, k, L, Fs, Vs, csoc, dsu, dsx, pins);
                    } else {
                        ric_backward<NS, NU, FREE0, SEG, ST, false>(N, lane, k, L, L.Q, L.M, L.R, Su, Sxk, Hv, Hv, gu, gxk, pt, p0a, csoc, Fc, Vs);
                        ric_forward<NS, NU, SEG, ST>(N, lane, k, L, Fc, Vs, csoc, dsu, dsx, pins);
                    }
                    MPC_UNROLL for (int i = 0; i < NS; i++) ds0[i] = Vs.dx0[i];
                    const double as_ = max_step(dsu, dsx, ds0);
                    if (soc) a_soc = as_;
                    double ctk[NS];      // (the constraint values of the step before stay with the segments that are not correcting)
                    MPC_UNROLL for (int i = 0; i < NS; i++) ctk[i] = ct[i];
                    const double th_keep = theta_t, ph_keep = phi_t; const bool fin_keep = fin_t;
                    trial(a_soc, dsu, dsx, ds0);
                    const bool acs = soc && acceptable(alpha);      // (the tests keep the original step length)
                    if (soc) {
                        if (acs) {      // the corrected step replaces the Newton step
                            accepted = true; searching = false; soc = false; phi_acc = phi_t; a_pr = a_soc;
                            MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = dsu[i];
                            MPC_UNROLL for (int i = 0; i < NS; i++) { dxn[i] = dsx[i]; dx0[i] = ds0[i]; pin[i] = pins[i]; }
                        } else {
                            cnt++; th_s = theta_t;
                            if (!(fin_t && cnt < kMaxSoc && th_s <= kKappaSoc * theta_old)) { soc = false; alpha *= 0.5; n_steps++; if (!(alpha > a_min)) searching = false; }      // back to shorter Newton steps
                        }
                    } else { MPC_UNROLL for (int i = 0; i < NS; i++) ct[i] = ctk[i]; theta_t = th_keep; phi_t = ph_keep; fin_t = fin_keep; }
                }
                EC_IPM_STAMP(11);      // second-order corrections
            }
        }
        EC_IPM_STAMP(10);
        EC_LDS_FENCE();
        bool restored = false;      // (MODE 1) this segment comes back from the restoration phase with a new point: no step of this iteration's direction is taken
        if constexpr (MODE == 1) {
            // ---- IPOPT's restoration phase (the comment above this function; the checker restates it with dense linear algebra) --------------------------------
            const bool need = !done && !accepted && !(theta <= 1e-2 * tol);
            if (__builtin_expect(__any(need ? 1 : 0), 0)) {
                constexpr int NUR = NU + 2 * NS;
                using PRr = ParkRows<NS, NUR, FREE0, true>;
                if (need) nfilt = filter_add(filt, nfilt, phi - kGammaPhi * theta, (1.0 - kGammaTheta) * theta);      // the point the restoration starts from is never returned to
                // reference point, its scaling D_R, the defects' first values [WB (32), (33)] at mu_R = max(mu, |c|_inf)
                double cm_ = 0.0;
                MPC_UNROLL for (int i = 0; i < NS; i++) cm_ = dmax(cm_, fabs(c[i]));
                const double mu_r = dmax(mu, SG::max(on ? cm_ : 0.0));
                double uR[NU], xR[NS], x0R[NS], xkR[NS], du2[NU], dx2[NS], dxk2[NS];
                double ur[NUR], xr[NS], x0r[NS], pir[NS], ulo_r[NUR], uhi_r[NUR], xlo_r[NS], xhi_r[NS];
                double *const pr = ext.park_r;
                const LRows<true, 1> rzlu{pr + PRr::ZLU, 0.0}, rblu{pr + PRr::BLU, 0.0}, rzhu{pr + PRr::ZHU, 0.0}, rbhu{pr + PRr::BHU, 0.0}, rzlx{pr + PRr::ZLX, 0.0}, rblx{pr + PRr::BLX, 0.0},
                                     rzhx{pr + PRr::ZHX, 0.0}, rbhx{pr + PRr::BHX, 0.0};
                const LRows<FREE0, 1> rzl0{pr + PRr::ZL0, 0.0}, rbl0{pr + PRr::BL0, 0.0}, rzh0{pr + PRr::ZH0, 0.0}, rbh0{pr + PRr::BH0, 0.0};
                MPC_UNROLL for (int i = 0; i < NU; i++) {
                    uR[i] = u[i]; ur[i] = u[i]; const double d_ = 1.0 / dmax(1.0, fabs(u[i])); du2[i] = d_ * d_;
                    ulo_r[i] = blu[i]; uhi_r[i] = bhu[i];      // (this solve's bounds as they are now: relaxed, moved)
                    rblu[i] = blu[i]; rbhu[i] = bhu[i]; rzlu[i] = flu[i] ? dmin(kRestoRho, zlu[i]) : 0.0; rzhu[i] = fhu[i] ? dmin(kRestoRho, zhu[i]) : 0.0;
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    xR[i] = xn[i]; xr[i] = xn[i]; x0R[i] = FREE0 ? x0v[i] : 0.0; x0r[i] = x0R[i]; pir[i] = 0.0;
                    const double d_ = 1.0 / dmax(1.0, fabs(xn[i])); dx2[i] = d_ * d_;
                    const double d0_ = 1.0 / dmax(1.0, fabs(x0R[i]));
                    xkR[i] = SG::up1(x0R[i], xR[i], k); dxk2[i] = SG::up1(FREE0 ? d0_ * d0_ : 0.0, dx2[i], k);      // x_k's term sits in stage k's cost; a given x_0 has none
                    xlo_r[i] = blx[i]; xhi_r[i] = bhx[i];
                    rblx[i] = blx[i]; rbhx[i] = bhx[i]; rzlx[i] = flx[i] ? dmin(kRestoRho, zlx[i]) : 0.0; rzhx[i] = fhx[i] ? dmin(kRestoRho, zhx[i]) : 0.0;
                    rbl0[i] = bl0[i]; rbh0[i] = bh0[i]; rzl0[i] = (FREE0 && flx[i]) ? dmin(kRestoRho, zl0[i]) : 0.0; rzh0[i] = (FREE0 && fhx[i]) ? dmin(kRestoRho, zh0[i]) : 0.0;
                    const double a_ = mu_r / (2.0 * kRestoRho) - 0.5 * c[i], n_ = a_ + sqrt(a_ * a_ + mu_r * c[i] / (2.0 * kRestoRho)), p_ = c[i] + n_;
                    ur[NU + i] = n_; ur[NU + NS + i] = p_;
                    ulo_r[NU + i] = 0.0; ulo_r[NU + NS + i] = 0.0; uhi_r[NU + i] = INFINITY; uhi_r[NU + NS + i] = INFINITY;
                    rblu[NU + i] = 0.0; rblu[NU + NS + i] = 0.0; rbhu[NU + i] = INFINITY; rbhu[NU + NS + i] = INFINITY;
                    rzlu[NU + i] = mu_r / n_; rzlu[NU + NS + i] = mu_r / p_; rzhu[NU + i] = 0.0; rzhu[NU + NS + i] = 0.0;
                }
                struct AuxR { AUX a; };
                // the restoration problem's cost of this lane's stage: rho (n + p) + eta / 2 (|D (u - u_R)|^2 + |D (x_k - x_kR)|^2)
                auto cost_r = [&](const double (&xk_)[NS], const double (&ur_)[NUR], StageLin<NS, NUR> &Lr, const double mu_) {
                    const double eta = sqrt(mu_);
                    double l_ = 0.0, q_ = 0.0;
                    MPC_UNROLL for (int i = 0; i < NUR; i++) { MPC_UNROLL for (int j = 0; j < NUR; j++) Lr.R[i][j] = 0.0; }
                    MPC_UNROLL for (int i = 0; i < NU; i++) { const double e_ = ur_[i] - uR[i]; q_ += du2[i] * e_ * e_; Lr.lu[i] = eta * du2[i] * e_; Lr.R[i][i] = eta * du2[i]; }
                    MPC_UNROLL for (int i = 0; i < NS; i++) {
                        const double e_ = xk_[i] - xkR[i]; q_ += dxk2[i] * e_ * e_; Lr.lx[i] = eta * dxk2[i] * e_;
                        MPC_UNROLL for (int j = 0; j < NS; j++) Lr.Q[i][j] = i == j ? eta * dxk2[i] : 0.0;
                        MPC_UNROLL for (int j = 0; j < NUR; j++) Lr.M[i][j] = 0.0;
                        l_ += ur_[NU + i] + ur_[NU + NS + i]; Lr.lu[NU + i] = kRestoRho; Lr.lu[NU + NS + i] = kRestoRho;
                    }
                    Lr.l = kRestoRho * l_ + 0.5 * eta * q_;
                };
                auto lin_r = [&](const double (&xk_)[NS], const double (&ur_)[NUR], StageLin<NS, NUR> &Lr, AuxR &ax, const double mu_) {
                    double uo[NU];
                    MPC_UNROLL for (int i = 0; i < NU; i++) uo[i] = ur_[i];
                    StageLin<NS, NU> Lo;
                    lin(xk_, uo, Lo, ax.a, mu_);
                    MPC_UNROLL for (int i = 0; i < NS; i++) {
                        Lr.F[i] = Lo.F[i] - ur_[NU + i] + ur_[NU + NS + i];
                        MPC_UNROLL for (int j = 0; j < NS; j++) { Lr.A[i][j] = Lo.A[i][j]; Lr.B[i][NU + j] = i == j ? -1.0 : 0.0; Lr.B[i][NU + NS + j] = i == j ? 1.0 : 0.0; }
                        MPC_UNROLL for (int j = 0; j < NU; j++) Lr.B[i][j] = Lo.B[i][j];
                    }
                    cost_r(xk_, ur_, Lr, mu_);
                };
                auto addpi_r = [&](const AuxR &ax, const double (&pi_)[NS], StageLin<NS, NUR> &Lr) {      // + sum pi_i Hessian(F_i): the defects enter linearly
                    StageLin<NS, NU> Lt;
                    MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Lt.Q[i][j] = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) Lt.M[i][j] = 0.0; }
                    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) Lt.R[i][j] = 0.0; }
                    addpi(ax.a, pi_, Lt);
                    MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < NS; j++) Lr.Q[i][j] += Lt.Q[i][j]; MPC_UNROLL for (int j = 0; j < NU; j++) Lr.M[i][j] += Lt.M[i][j]; }
                    MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) Lr.R[i][j] += Lt.R[i][j]; }
                };
                auto val_r = [&](const double (&xk_)[NS], const double (&ur_)[NUR], double (&F_)[NS], double &l_, const double mu_) {
                    double uo[NU], lo_;
                    MPC_UNROLL for (int i = 0; i < NU; i++) uo[i] = ur_[i];
                    val(xk_, uo, F_, lo_, mu_);
                    const double eta = sqrt(mu_);
                    double s_ = 0.0, q_ = 0.0;
                    MPC_UNROLL for (int i = 0; i < NU; i++) { const double e_ = ur_[i] - uR[i]; q_ += du2[i] * e_ * e_; }
                    MPC_UNROLL for (int i = 0; i < NS; i++) { const double e_ = xk_[i] - xkR[i]; q_ += dxk2[i] * e_ * e_; s_ += ur_[NU + i] + ur_[NU + NS + i]; F_[i] += -ur_[NU + i] + ur_[NU + NS + i]; }
                    l_ = kRestoRho * s_ + 0.5 * eta * q_;
                };
                auto term_r = [&](const double (&xe_)[NS], double &fv_, double (&gv_)[NS], double (&Hv_)[NS][NS], const double mu_) {      // x_N's term (used from lane N - 1)
                    const double eta = sqrt(mu_);
                    double q_ = 0.0;
                    MPC_UNROLL for (int i = 0; i < NS; i++) { const double e_ = xe_[i] - xR[i]; q_ += dx2[i] * e_ * e_; gv_[i] = eta * dx2[i] * e_; MPC_UNROLL for (int j = 0; j < NS; j++) Hv_[i][j] = i == j ? eta * dx2[i] : 0.0; }
                    fv_ = 0.5 * eta * q_;
                };
                auto grd_r = [&](const double (&)[NS], const double (&)[NUR], StageLin<NS, NUR> &, const double) {};      // (no scaling of the restoration problem)
                // the test of this solve [IpRestoFilterConvCheck]: enough less infeasible, acceptable to the filter and to the iterate that was left - with this solve's
                // mu, multipliers and bounds as they were on entry
                auto hook = [&](const double (&ur_)[NUR], const double (&xr_)[NS], const double (&x0r_)[NS]) {
                    double uo[NU];
                    MPC_UNROLL for (int i = 0; i < NU; i++) uo[i] = ur_[i];
                    eval_point(uo, xr_, x0r_);
                    if (!fin_t || theta_t > kRestoKappa * theta) return false;
                    return !filter_rejects(filt, nfilt, phi_t, theta_t) && (le_tol(theta_t, (1.0 - kGammaTheta) * theta, theta) || le_tol(phi_t - phi, -kGammaPhi * theta, phi));
                };
                double zeroP[NS][NS], zerob[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) { zerob[i] = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) zeroP[i][j] = 0.0; }
                int it_r = 0;
                IpmRestoIn<decltype(hook), decltype(cost_r)> rin{mu_r, it + 1, hook, cost_r};
                const int rs = ipm_stage<NS, NUR, FREE0, true, SEG, DenseStage, AuxR, 2, 1>(N, lane, need, x0fix, x0r, ur, xr, pir, ulo_r, uhi_r, xlo_r, xhi_r, zeroP, zerob, tol, max_iter,
                                                                                         grd_r, lin_r, addpi_r, val_r, term_r, it_r, ext.filt_r, ext.park_r, rin);
                // the point it ends at, seen by this solve (constraint values: is a converged restoration feasible?)
                double un_[NU];
                MPC_UNROLL for (int i = 0; i < NU; i++) un_[i] = ur[i];
                eval_point(un_, xr, x0r);
                double cx_ = 0.0;
                MPC_UNROLL for (int i = 0; i < NS; i++) cx_ = dmax(cx_, fabs(ct[i]));
                cx_ = SG::max(on ? cx_ : 0.0);
                // back from the restoration: bound multipliers as if the whole move had been one Newton step, reset to 1 beyond 1e3; the bounds it moved stay moved
                double s1u[NU], s2u[NU], s1x[NS], s2x[NS], s10[NS], s20[NS], dzl_u[NU], dzh_u[NU], dzl_x[NS], dzh_x[NS], dzl_0[NS], dzh_0[NS];
                double bl_u[NU], bh_u[NU], bl_x[NS], bh_x[NS], bl_0[NS], bh_0[NS];
                MinRatio mzr;
                double zmx = 0.0;
                auto dzr = [&](bool f_, double so_, double z_, double sn_) { return f_ ? mu / so_ - z_ - z_ / so_ * (sn_ - so_) : 0.0; };
                MPC_UNROLL for (int i = 0; i < NU; i++) {
                    bl_u[i] = rblu[i]; bh_u[i] = rbhu[i];
                    s1u[i] = flu[i] ? safe_slack(un_[i], bl_u[i], zlu[i], mu, true) : 1.0; s2u[i] = fhu[i] ? safe_slack(un_[i], bh_u[i], zhu[i], mu, false) : 1.0;
                    dzl_u[i] = dzr(flu[i], slu[i], zlu[i], s1u[i]); dzh_u[i] = dzr(fhu[i], shu[i], zhu[i], s2u[i]);
                    mzr.add(zlu[i], dzl_u[i], tau, flu[i]); mzr.add(zhu[i], dzh_u[i], tau, fhu[i]);
                }
                MinRatio mz0;
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    bl_x[i] = rblx[i]; bh_x[i] = rbhx[i]; bl_0[i] = rbl0[i]; bh_0[i] = rbh0[i];
                    s1x[i] = flx[i] ? safe_slack(xr[i], bl_x[i], zlx[i], mu, true) : 1.0; s2x[i] = fhx[i] ? safe_slack(xr[i], bh_x[i], zhx[i], mu, false) : 1.0;
                    dzl_x[i] = dzr(flx[i], slx[i], zlx[i], s1x[i]); dzh_x[i] = dzr(fhx[i], shx[i], zhx[i], s2x[i]);
                    mzr.add(zlx[i], dzl_x[i], tau, flx[i]); mzr.add(zhx[i], dzh_x[i], tau, fhx[i]);
                    s10[i] = (FREE0 && flx[i]) ? safe_slack(x0r[i], bl_0[i], zl0[i], mu, true) : 1.0; s20[i] = (FREE0 && fhx[i]) ? safe_slack(x0r[i], bh_0[i], zh0[i], mu, false) : 1.0;
                    dzl_0[i] = dzr(FREE0 && flx[i], sl0[i], zl0[i], s10[i]); dzh_0[i] = dzr(FREE0 && fhx[i], sh0[i], zh0[i], s20[i]);
                    mz0.add(zl0[i], dzl_0[i], tau, FREE0 && flx[i]); mz0.add(zh0[i], dzh_0[i], tau, FREE0 && fhx[i]);
                }
                double adr = SG::min(on ? mzr.value() : 1.0);
                if (FREE0) adr = dmin(adr, mz0.value());
                MPC_UNROLL for (int i = 0; i < NU; i++) zmx = dmax(zmx, dmax(zlu[i] + adr * dzl_u[i], zhu[i] + adr * dzh_u[i]));
                MPC_UNROLL for (int i = 0; i < NS; i++) zmx = dmax(zmx, dmax(zlx[i] + adr * dzl_x[i], zhx[i] + adr * dzh_x[i]));
                zmx = SG::max(on ? zmx : 0.0);
                if (FREE0) { MPC_UNROLL for (int i = 0; i < NS; i++) zmx = dmax(zmx, dmax(zl0[i] + adr * dzl_0[i], zh0[i] + adr * dzh_0[i])); }
                if (need) {
                    if (rs != kRsRestored) {
                        if (rs == kRsLimit) status = kStMaxIter;
                        else if (rs == kRsConverged) status = cx_ <= kRestoFeasFact * tol ? kStMaxIter : kStFailed;      // the restoration problem has a minimiser here: infeasible, or feasible and not acceptable
                        else status = kStMaxIter;      // 'Restoration_Failed': the reference accepts the point
                        done = true; iters = it_r - 1;
                    } else {
                        const bool reset = zmx > kRestoBoundMultReset;
                        auto zfin = [&](bool f_, double z_, double dz_, double s_) { const double zn_ = reset ? 1.0 : z_ + adr * dz_; return f_ ? dmin(dmax(zn_, mu / (kKappaSigma * s_)), kKappaSigma * mu / s_) : 0.0; };
                        MPC_UNROLL for (int i = 0; i < NU; i++) { u[i] = un_[i]; blu[i] = bl_u[i]; bhu[i] = bh_u[i]; zlu[i] = zfin(flu[i], zlu[i], dzl_u[i], s1u[i]); zhu[i] = zfin(fhu[i], zhu[i], dzh_u[i], s2u[i]); }
                        MPC_UNROLL for (int i = 0; i < NS; i++) {
                            xn[i] = xr[i]; pi[i] = 0.0; blx[i] = bl_x[i]; bhx[i] = bh_x[i]; zlx[i] = zfin(flx[i], zlx[i], dzl_x[i], s1x[i]); zhx[i] = zfin(fhx[i], zhx[i], dzh_x[i], s2x[i]);
                            if (FREE0) { x0v[i] = x0r[i]; bl0[i] = bl_0[i]; bh0[i] = bh_0[i]; zl0[i] = zfin(flx[i], zl0[i], dzl_0[i], s10[i]); zh0[i] = zfin(fhx[i], zh0[i], dzh_0[i], s20[i]); }
                        }
                        tiny_last = false;
                        it = it_r;      // (the restoration's return counts as an iteration: its own counter went on from it + 1)
                        restored = true;
                    }
                }
            }
        }
        if (!done && !restored) {
            if (!accepted) {      // MODE 0: IPOPT enters its restoration phase here - the caller's rare path (kStNeedResto); feasible to 1e-2 tol: the point is kept ('Restoration_Failed'); MODE 2: no restoration inside the restoration
                status = MODE == 2 ? (int)kRsFailed : (theta <= 1e-2 * tol ? (int)kStMaxIter : (int)(MODE == 0 ? kStNeedResto : kStFailed)); done = true;
            } else if (tiny) { tiny_flag = tiny_last; tiny_last = dym < kTinyStepYTol; }
            else {
                tiny_last = false;
                // the filter grows unless the step was an Armijo step on the barrier function.  Every lane of the segment makes the same edit of the
                // segment's list (same values to the same LDS words, in lockstep); entries the new one dominates are dropped
                if (!ftype(alpha) || !le_tol(phi_acc - phi, kEtaPhi * alpha * gbd, phi)) nfilt = filter_add(filt, nfilt, phi - kGammaPhi * theta, (1.0 - kGammaTheta) * theta);
            }
        }
        // ---- the accepted point: multiplier steps of the direction that was taken, bounds moved with corrected slacks, multipliers within
        // kappa_Sigma of mu / slack ----------------------------------------------------------------------------------------------------------------
        double dzlu[NU], dzhu[NU], dzlx[NS], dzhx[NS], dzl0[NS], dzh0[NS];
        // multiplier step of a bound with slack s_, multiplier z_, whose variable moves by dv_ towards the bound's side: mu / s - z - z / s dv, one division
        auto dzstep = [&](bool f_, double s_, double z_, double dv_) { return f_ ? (mu - z_ * dv_) / s_ - z_ : 0.0; };
        MinRatio mz;
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            dzlu[i] = dzstep(flu[i], slu[i], zlu[i], du[i]); dzhu[i] = dzstep(fhu[i], shu[i], zhu[i], -du[i]);
            mz.add(zlu[i], dzlu[i], tau, flu[i]); mz.add(zhu[i], dzhu[i], tau, fhu[i]);
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            dzlx[i] = dzstep(flx[i], slx[i], zlx[i], dxn[i]); dzhx[i] = dzstep(fhx[i], shx[i], zhx[i], -dxn[i]);
            mz.add(zlx[i], dzlx[i], tau, flx[i]); mz.add(zhx[i], dzhx[i], tau, fhx[i]);
        }
        double adu = SG::min(on ? mz.value() : 1.0);
        if (FREE0) {
            MinRatio m0;
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                dzl0[i] = dzstep(flx[i], sl0[i], zl0[i], dx0[i]); dzh0[i] = dzstep(fhx[i], sh0[i], zh0[i], -dx0[i]);
                m0.add(zl0[i], dzl0[i], tau, flx[i]); m0.add(zh0[i], dzh0[i], tau, fhx[i]);
            }
            adu = dmin(adu, m0.value());
        }
        // the new multiplier of a bound: the step, then within [mu / (kappa_Sigma s), kappa_Sigma mu / s] of the new slack - one division
        auto newz = [&](bool f_, double z_, double dz_, double s_) { const double q = mu / s_, zn_ = z_ + adu * dz_; return f_ ? dmin(dmax(zn_, q * (1.0 / kKappaSigma)), kKappaSigma * q) : zn_; };
        if (!done && !restored) {      // (a finished segment keeps its iterate)
            MPC_UNROLL for (int i = 0; i < NU; i++) {      // (the accepted trial point again, to the bit; its slacks with the multipliers of the old point, as the trial had them; the bounds move now)
                u[i] = u[i] + a_pr * du[i];
                const double s1 = flu[i] ? safe_slack(u[i], blu[i], zlu[i], mu, true) : 1.0, s2 = fhu[i] ? safe_slack(u[i], bhu[i], zhu[i], mu, false) : 1.0;
                zlu[i] = newz(flu[i], zlu[i], dzlu[i], s1); zhu[i] = newz(fhu[i], zhu[i], dzhu[i], s2);
            }
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                xn[i] = xn[i] + a_pr * dxn[i]; pi[i] += a_pr * (pin[i] - pi[i]);
                const double s1 = flx[i] ? safe_slack(xn[i], blx[i], zlx[i], mu, true) : 1.0, s2 = fhx[i] ? safe_slack(xn[i], bhx[i], zhx[i], mu, false) : 1.0;
                zlx[i] = newz(flx[i], zlx[i], dzlx[i], s1); zhx[i] = newz(fhx[i], zhx[i], dzhx[i], s2);
                if (FREE0) {
                    x0v[i] = x0v[i] + a_pr * dx0[i];
                    const double t1 = flx[i] ? safe_slack(x0v[i], bl0[i], zl0[i], mu, true) : 1.0, t2 = fhx[i] ? safe_slack(x0v[i], bh0[i], zh0[i], mu, false) : 1.0;
                    zl0[i] = newz(flx[i], zl0[i], dzl0[i], t1); zh0[i] = newz(fhx[i], zh0[i], dzh0[i], t2);
                }
            }
            it++;
        }
        EC_IPM_STAMP(12);      // filter, multiplier steps, the new iterate
    }
    if constexpr (MODE != 2) {      // the final point goes back into the caller's bounds (honor_original_bounds)
        MPC_UNROLL for (int i = 0; i < NU; i++) { if (UB) u[i] = dmin(dmax(u[i], ulo_in[i]), uhi_in[i]); }
        MPC_UNROLL for (int i = 0; i < NS; i++) { xn[i] = dmin(dmax(xn[i], xlo_in[i]), xhi_in[i]); if (FREE0) x0v[i] = dmin(dmax(x0v[i], xlo_in[i]), xhi_in[i]); }
    }
    return status;
}

// ---- small dense helpers (wave-uniform use) ---------------------------------------------------------------------------------------------
// inverse of a general n x n matrix by Gauss-Jordan with partial pivoting; false when a pivot vanishes
// ---------------------------------------------------------------------------------------------------------------------------------
// IPOPT's restoration phase for the target problem ([WB 3.3]): the same interior point iteration on
//     min  rho sum(n + p) + sqrt(mu) / 2 |D_R (x - x_R)|^2   s.t.  c(x) + n - p = 0,  lo <= x <= hi,  n, p >= 0
// from x_R, until orig_ok(x) - the point is enough less infeasible and acceptable to the filter and the iterate of the solve that called.
// n and p are eliminated from the Newton system: (H + Sigma_x + J' Dc^-1 J) dx = ..., Dc = 1 / Sigma_n + 1 / Sigma_p, whose positive definiteness is
// the inertia condition of the full system.  One instance per lane.  x: x_R in, the restored point out; lo / hi: the caller's (moved) bounds, moved on.
// ---------------------------------------------------------------------------------------------------------------------------------
// what the target's interior point hands to its restoration phase and gets back.  The restoration is a function of its own (not inlined: it is entered by one
// solve in a thousand, and inlined its registers were the target kernel's - 3 KB of scratch per lane): the caller copies its state here, calls, copies back.
template <class M>
struct TgtRestoIO {
    static constexpr int NV = M::NX + M::NU + M::NY, MC = M::NX + M::NY, NDD = M::ND > 0 ? M::ND : 1;
    double x[NV], lo[NV], hi[NV], zl[NV], zh[NV], c[MC], d[NDD], Bd[M::NX][NDD], Cd[M::NY][NDD], filt[2 * kFilterCap];
    double mu, theta, phi, df, t, h, tol;
    int nfilt, max_iter, it;
};
template <class M>
__device__ __forceinline__ int target_resto(TgtRestoIO<M> &io)
{
    constexpr int NX = M::NX, NU = M::NU, NY = M::NY, ND = M::ND, NV = NX + NU + NY, NP = NX + NU, NPP = NP * (NP + 1) / 2, MC = NX + NY;
    double x[NV], lo[NV], hi[NV], zl_o[NV], zh_o[NV], c_r[MC], d[ND > 0 ? ND : 1], Bd[NX][ND > 0 ? ND : 1], Cd[NY][ND > 0 ? ND : 1];
    MPC_UNROLL for (int i = 0; i < NV; i++) { x[i] = io.x[i]; lo[i] = io.lo[i]; hi[i] = io.hi[i]; zl_o[i] = io.zl[i]; zh_o[i] = io.zh[i]; }
    MPC_UNROLL for (int j = 0; j < MC; j++) c_r[j] = io.c[j];
    MPC_UNROLL for (int j = 0; j < ND; j++) { d[j] = io.d[j]; MPC_UNROLL for (int i = 0; i < NX; i++) Bd[i][j] = io.Bd[i][j]; MPC_UNROLL for (int i = 0; i < NY; i++) Cd[i][j] = io.Cd[i][j]; }
    const double mu_o = io.mu, t = io.t, h = io.h, tol = io.tol;
    const int max_iter = io.max_iter;
    int it = io.it;
    // the test of the solve that called: enough less infeasible, acceptable to its filter and to the point it left (its bounds, multipliers, mu, scaling)
    auto orig_ok = [&](const double (&w)[NV]) {
        typename M::Ctx cx;
        MPC_UNROLL for (int i = 0; i < NU; i++) { cx.u[i] = w[NX + i]; cx.us[i] = 0.0; }
        MPC_UNROLL for (int i = 0; i < ND; i++) cx.d[i] = d[i];
        MPC_UNROLL for (int i = 0; i < NX; i++) cx.xs[i] = 0.0;
        double Fx[NX], f_, g_[NV], H_[NV][NV], th_t = 0.0;
        rk4_plain<typename M::Mdl>(w, cx, t, true, h, M::MX, Fx);
        M::fss(w, &f_, g_, H_);
        f_ *= io.df;
        bool ok_ = finite_all(f_);
        MPC_UNROLL for (int i = 0; i < NX; i++) { double a = Fx[i] - w[i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += Bd[i][j] * d[j]; th_t += fabs(a); ok_ = ok_ && finite_all(a); }
        MPC_UNROLL for (int i = 0; i < NY; i++) { double a = w[i] - w[NX + NU + i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += Cd[i][j] * d[j]; th_t += fabs(a); ok_ = ok_ && finite_all(a); }
        if (!ok_ || th_t > kRestoKappa * io.theta) return false;
        double ph_t = f_;
        LogSum lg;
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            double bl_ = io.lo[i], bh_ = io.hi[i];
            const bool fl_ = fin(bl_), fh_ = fin(bh_);
            const double s1 = fl_ ? safe_slack(w[i], bl_, io.zl[i], mu_o, true) : 1.0, s2 = fh_ ? safe_slack(w[i], bh_, io.zh[i], mu_o, false) : 1.0;
            lg.mul(s1, fl_); lg.mul(s2, fh_);
            if (fl_ && !fh_) ph_t += kKappaD * mu_o * s1;
            if (fh_ && !fl_) ph_t += kKappaD * mu_o * s2;
        }
        ph_t -= mu_o * lg.value();
        return !filter_rejects(io.filt, io.nfilt, ph_t, th_t) && (le_tol(th_t, (1.0 - kGammaTheta) * io.theta, io.theta) || le_tol(ph_t - io.phi, -kGammaPhi * io.theta, io.phi));
    };
    bool fl[NV], fh[NV];
    double xr[NV], dr2[NV], zl[NV], zh[NV], dmp[NV], nn[MC], pp[MC], zn[MC], zp[MC], nlo[MC], plo[MC], lam[MC];
    int nbi = 2 * MC;
    double mu = mu_o;
    MPC_UNROLL for (int j = 0; j < MC; j++) mu = dmax(mu, fabs(c_r[j]));
    MPC_UNROLL for (int i = 0; i < NV; i++) {
        xr[i] = x[i]; const double dd = 1.0 / dmax(1.0, fabs(x[i])); dr2[i] = dd * dd;
        fl[i] = fin(lo[i]); fh[i] = fin(hi[i]); nbi += (fl[i] ? 1 : 0) + (fh[i] ? 1 : 0);
        zl[i] = fl[i] ? dmin(kRestoRho, zl_o[i]) : 0.0; zh[i] = fh[i] ? dmin(kRestoRho, zh_o[i]) : 0.0;
        dmp[i] = (fl[i] && !fh[i]) ? 1.0 : ((fh[i] && !fl[i]) ? -1.0 : 0.0);
    }
    MPC_UNROLL for (int j = 0; j < MC; j++) {
        const double a = mu / (2.0 * kRestoRho) - 0.5 * c_r[j];
        nn[j] = a + sqrt(a * a + mu * c_r[j] / (2.0 * kRestoRho)); pp[j] = c_r[j] + nn[j];
        zn[j] = mu / nn[j]; zp[j] = mu / pp[j]; nlo[j] = 0.0; plo[j] = 0.0; lam[j] = 0.0;
    }
    const double nb = (double)nbi, meq = (double)MC;
    const double mu_min = dmin(tol, kComplInfTol) / (kKappaEps + 1.0);
    double tau = dmax(kTauMin, 1.0 - mu), delta_last = 0.0, theta_max = -1.0, theta_min = -1.0;
    double filt[2 * kFilterCap];
    int nfilt = 0, acc_count = 0, status = kRsLimit;
    bool tiny_last = false, tiny_flag = false;
    // constraint values at a point: c(x) + n - p
    auto cons = [&](const double (&w)[NV], const double (&n_)[MC], const double (&p_)[MC], double (&cb)[MC]) {
        typename M::Ctx cx;
        MPC_UNROLL for (int i = 0; i < NU; i++) { cx.u[i] = w[NX + i]; cx.us[i] = 0.0; }
        MPC_UNROLL for (int i = 0; i < ND; i++) cx.d[i] = d[i];
        MPC_UNROLL for (int i = 0; i < NX; i++) cx.xs[i] = 0.0;
        double Fx[NX];
        rk4_plain<typename M::Mdl>(w, cx, t, true, h, M::MX, Fx);
        MPC_UNROLL for (int i = 0; i < NX; i++) { double a = Fx[i] - w[i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += Bd[i][j] * d[j]; cb[i] = a + n_[i] - p_[i]; }
        MPC_UNROLL for (int i = 0; i < NY; i++) { double a = w[i] - w[NX + NU + i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += Cd[i][j] * d[j]; cb[NX + i] = a + n_[NX + i] - p_[NX + i]; }
    };
    auto objective = [&](const double (&w)[NV], const double (&n_)[MC], const double (&p_)[MC], double mu_) {
        double s_ = 0.0, q_ = 0.0;
        MPC_UNROLL for (int j = 0; j < MC; j++) s_ += n_[j] + p_[j];
        MPC_UNROLL for (int i = 0; i < NV; i++) { const double e = w[i] - xr[i]; q_ += dr2[i] * e * e; }
        return kRestoRho * s_ + 0.5 * sqrt(mu_) * q_;
    };
    auto barrier = [&](const double (&w)[NV], const double (&n_)[MC], const double (&p_)[MC], double f_, double mu_) {      // with the safe slacks of a trial point
        double ph = f_;
        LogSum lg;
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            double bl_ = lo[i], bh_ = hi[i];
            const double s1 = fl[i] ? safe_slack(w[i], bl_, zl[i], mu_, true) : 1.0, s2 = fh[i] ? safe_slack(w[i], bh_, zh[i], mu_, false) : 1.0;
            lg.mul(s1, fl[i]); lg.mul(s2, fh[i]);
            if (dmp[i] != 0.0) ph += kKappaD * mu_ * (dmp[i] > 0.0 ? s1 : s2);
        }
        MPC_UNROLL for (int j = 0; j < MC; j++) {
            double b1 = nlo[j], b2 = plo[j];
            const double s1 = safe_slack(n_[j], b1, zn[j], mu_, true), s2 = safe_slack(p_[j], b2, zp[j], mu_, true);
            lg.mul(s1); lg.mul(s2);
            ph += kKappaD * mu_ * (s1 + s2);
        }
        return ph - mu_ * lg.value();
    };
    for (;;) {
        typename M::Ctx cx;
        MPC_UNROLL for (int i = 0; i < NU; i++) { cx.u[i] = x[NX + i]; cx.us[i] = 0.0; }
        MPC_UNROLL for (int i = 0; i < ND; i++) cx.d[i] = d[i];
        MPC_UNROLL for (int i = 0; i < NX; i++) cx.xs[i] = 0.0;
        double Fx[NX], S[NX][NP], T[NX][NPP];
        rk4_sens2<typename M::Mdl>(x, cx, t, true, h, M::MX, Fx, S, T);
        // Jacobian of the constraints with respect to x: rows of c1 = [S_x - I, S_u, 0], rows of c2 = [I, 0, -I]
        double J[MC][NV], cb[MC];
        MPC_UNROLL for (int i = 0; i < NX; i++) {
            double a = Fx[i] - x[i];
            MPC_UNROLL for (int j = 0; j < ND; j++) a += Bd[i][j] * d[j];
            cb[i] = a + nn[i] - pp[i];
            MPC_UNROLL for (int j = 0; j < NV; j++) J[i][j] = j < NP ? S[i][j < NP ? j : 0] - (i == j ? 1.0 : 0.0) : 0.0;
        }
        MPC_UNROLL for (int i = 0; i < NY; i++) {
            double a = x[i] - x[NX + NU + i];
            MPC_UNROLL for (int j = 0; j < ND; j++) a += Cd[i][j] * d[j];
            cb[NX + i] = a + nn[NX + i] - pp[NX + i];
            MPC_UNROLL for (int j = 0; j < NV; j++) J[NX + i][j] = (j == i) ? 1.0 : ((j == NP + i) ? -1.0 : 0.0);
        }
        double eta = sqrt(mu), f = objective(x, nn, pp, mu);
        double sl[NV], sh[NV], sn[MC], sp[MC];
        double e_st = 0.0, e_c = 0.0, s_l = 0.0, s_z = 0.0, cmax = -INFINITY, cmin = INFINITY, theta = 0.0;
        bool finite = finite_all(f);
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            sl[i] = fl[i] ? safe_slack(x[i], lo[i], zl[i], mu, true) : 1.0; sh[i] = fh[i] ? safe_slack(x[i], hi[i], zh[i], mu, false) : 1.0;
            double r = eta * dr2[i] * (x[i] - xr[i]) - zl[i] + zh[i];
            MPC_UNROLL for (int j = 0; j < MC; j++) r += J[j][i] * lam[j];
            e_st = dmax(e_st, fabs(r)); s_z += zl[i] + zh[i];
            finite = finite && finite_all(r) && finite_all(x[i]);
            if (fl[i]) { cmax = dmax(cmax, sl[i] * zl[i]); cmin = dmin(cmin, sl[i] * zl[i]); }
            if (fh[i]) { cmax = dmax(cmax, sh[i] * zh[i]); cmin = dmin(cmin, sh[i] * zh[i]); }
        }
        MPC_UNROLL for (int j = 0; j < MC; j++) {
            sn[j] = safe_slack(nn[j], nlo[j], zn[j], mu, true); sp[j] = safe_slack(pp[j], plo[j], zp[j], mu, true);
            const double rn = kRestoRho + lam[j] - zn[j], rp = kRestoRho - lam[j] - zp[j];
            e_st = dmax(e_st, dmax(fabs(rn), fabs(rp))); s_z += zn[j] + zp[j]; s_l += fabs(lam[j]);
            e_c = dmax(e_c, fabs(cb[j])); theta += fabs(cb[j]);
            finite = finite && finite_all(rn) && finite_all(rp) && finite_all(cb[j]) && finite_all(nn[j]) && finite_all(pp[j]);
            cmax = dmax(cmax, dmax(sn[j] * zn[j], sp[j] * zp[j])); cmin = dmin(cmin, dmin(sn[j] * zn[j], sp[j] * zp[j]));
        }
        if (!finite) { status = kRsFailed; break; }
        if (orig_ok(x)) { status = kRsRestored; break; }
        const double s_d = dmax(kSMax, (s_l + s_z) / dmax(meq + nb, 1.0)) / kSMax, s_c = dmax(kSMax, s_z / dmax(nb, 1.0)) / kSMax;
        auto compl_ = [&](double m_) { return dmax(cmax - m_, m_ - cmin); };
        auto err = [&](double m_) { return dmax(dmax(e_st / s_d, e_c), compl_(m_) / s_c); };
        {
            const double e0_ = err(0.0), c0_ = compl_(0.0);
#ifdef EC_TRACE_TGT      /* diagnostic build only (tools/scratch): the first instance of a launch prints its iterations */
            if (blockIdx.x == 0 && threadIdx.x == 0) printf("R it=%3d mu=%.3e E0=%.6e e_st=%.3e e_c=%.3e compl=%.3e theta=%.6e\n", it, mu, e0_, e_st, e_c, c0_, theta);
#endif
            if (e0_ <= tol && e_st <= kDualInfTol && e_c <= kConstrViolTol && c0_ <= kComplInfTol) { status = kRsConverged; break; }
            if (e0_ <= kAccTol && e_st <= kAccDualInfTol && e_c <= kAccConstrViolTol && c0_ <= kAccComplInfTol) { if (++acc_count >= kAccIter) { status = kRsConverged; break; } }
            else acc_count = 0;
        }
        if (it >= max_iter) { status = kRsLimit; break; }
        {
            bool mu_changed = false, stop_tiny = false;
            while (err(mu) <= kKappaEps * mu || tiny_flag) {
                const double new_mu = dmax(dmin(kKappaMu * mu, mu * sqrt(mu)), mu_min);
                if (new_mu == mu) { stop_tiny = tiny_flag; break; }
                mu = new_mu; mu_changed = true; tiny_flag = false;
            }
            if (stop_tiny) { status = kRsConverged; break; }
            tiny_flag = false;
            if (mu_changed) { nfilt = 0; tau = dmax(kTauMin, 1.0 - mu); eta = sqrt(mu); f = objective(x, nn, pp, mu); }      // (the objective changes with mu)
        }
        // gradient of the barrier function, diagonal terms, barrier function
        double Sg[NV], gx[NV], Sn[MC], Sp[MC], gn[MC], gp[MC];
        double phi = f;
        LogSum lgp;
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            const double il = fl[i] ? 1.0 / sl[i] : 0.0, ih = fh[i] ? 1.0 / sh[i] : 0.0;
            Sg[i] = zl[i] * il + zh[i] * ih; gx[i] = eta * dr2[i] * (x[i] - xr[i]) - mu * il + mu * ih + kKappaD * mu * dmp[i];
            lgp.mul(sl[i], fl[i]); lgp.mul(sh[i], fh[i]);
            if (dmp[i] != 0.0) phi += kKappaD * mu * (dmp[i] > 0.0 ? sl[i] : sh[i]);
        }
        MPC_UNROLL for (int j = 0; j < MC; j++) {
            Sn[j] = zn[j] / sn[j]; Sp[j] = zp[j] / sp[j];
            gn[j] = kRestoRho + kKappaD * mu - mu / sn[j]; gp[j] = kRestoRho + kKappaD * mu - mu / sp[j];
            lgp.mul(sn[j]); lgp.mul(sp[j]);
            phi += kKappaD * mu * (sn[j] + sp[j]);
        }
        phi -= mu * lgp.value();
        // Hessian of lam' c with respect to x (only the model's rows are non-linear) + the proximity term
        double H[NV][NV];
        MPC_UNROLL for (int i = 0; i < NV; i++) { MPC_UNROLL for (int j = 0; j < NV; j++) H[i][j] = (i == j) ? eta * dr2[i] : 0.0; }
        MPC_UNROLL for (int a = 0; a < NP; a++) { MPC_UNROLL for (int b = 0; b < NP; b++) { double s_ = 0.0; MPC_UNROLL for (int i = 0; i < NX; i++) s_ += lam[i] * T[i][pair_idx<NP>(a, b)]; H[a][b] += s_; } }
        double Ai[NV][NV], Dci[MC];
        double delta = 0.0;
        bool failed = false;
        for (;;) {
            MPC_UNROLL for (int j = 0; j < MC; j++) Dci[j] = 1.0 / (1.0 / (Sn[j] + delta) + 1.0 / (Sp[j] + delta));
            MPC_UNROLL for (int i = 0; i < NV; i++) { MPC_UNROLL for (int l = 0; l <= i; l++) {
                double a = H[i][l] + (i == l ? Sg[i] + delta : 0.0);
                MPC_UNROLL for (int j = 0; j < MC; j++) a += J[j][i] * Dci[j] * J[j][l];
                Ai[i][l] = a; Ai[l][i] = a; } }
            if (sym_inverse<NV>(Ai)) break;
            delta = delta == 0.0 ? dmax(kDeltaFirst, delta_last / 3.0) : delta * (delta_last == 0.0 ? 100.0 : 8.0);
            if (delta > kDeltaMax) { failed = true; break; }
        }
        if (failed) { status = kRsFailed; break; }
        if (delta > 0.0) delta_last = delta;
        // Newton step for a constraint residual cc: dx, dn, dp and the new multipliers
        auto direction = [&](const double (&cc)[MC], double (&dx)[NV], double (&dn)[MC], double (&dp)[MC], double (&yn)[MC]) {
            double rc[MC], rhs[NV];
            MPC_UNROLL for (int j = 0; j < MC; j++) rc[j] = cc[j] - gn[j] / (Sn[j] + delta) + gp[j] / (Sp[j] + delta);
            MPC_UNROLL for (int i = 0; i < NV; i++) { double a = -gx[i]; MPC_UNROLL for (int j = 0; j < MC; j++) a -= J[j][i] * Dci[j] * rc[j]; rhs[i] = a; }
            MPC_UNROLL for (int i = 0; i < NV; i++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NV; l++) a += Ai[i][l] * rhs[l]; dx[i] = a; }
            MPC_UNROLL for (int j = 0; j < MC; j++) {
                double a = rc[j], lin = cc[j];
                MPC_UNROLL for (int i = 0; i < NV; i++) { a += J[j][i] * dx[i]; lin += J[j][i] * dx[i]; }
                yn[j] = Dci[j] * a;
                // Of n_j and p_j one sits at its bound (Sigma large), the other carries the infeasibility (Sigma = mu / p^2 -> 0): the first takes its step from its own
                // row of the Newton system, the second from the linearised constraint J dx + dn - dp = -cc - dividing by the small Sigma would multiply the rounding
                // of (multiplier - gradient) by 1 / Sigma and leave the constraint violated by as much (1e-8 .. 1e-5 near the end of a restoration: it then never converged)
                if (Sn[j] >= Sp[j]) { dn[j] = -(gn[j] + yn[j]) / (Sn[j] + delta); dp[j] = lin + dn[j]; }
                else { dp[j] = (yn[j] - gp[j]) / (Sp[j] + delta); dn[j] = dp[j] - lin; }
            }
        };
        auto ratio = [&](double a, double vv, double dvv) { return dvv < 0.0 ? dmin(a, -tau * vv / dvv) : a; };
        auto max_step = [&](const double (&dx_)[NV], const double (&dn_)[MC], const double (&dp_)[MC]) {
            double a = 1.0;
            MPC_UNROLL for (int i = 0; i < NV; i++) { if (fl[i]) a = ratio(a, sl[i], dx_[i]); if (fh[i]) a = ratio(a, sh[i], -dx_[i]); }
            MPC_UNROLL for (int j = 0; j < MC; j++) { a = ratio(a, sn[j], dn_[j]); a = ratio(a, sp[j], dp_[j]); }
            return a;
        };
        double dx[NV], dn[MC], dp[MC], yn[MC];
        direction(cb, dx, dn, dp, yn);
        const double a_max = max_step(dx, dn, dp);
        double gbd = 0.0, drel = 0.0, dym = 0.0;
        MPC_UNROLL for (int i = 0; i < NV; i++) { gbd += gx[i] * dx[i]; drel = dmax(drel, fabs(dx[i]) / (1.0 + fabs(x[i]))); }
        MPC_UNROLL for (int j = 0; j < MC; j++) {
            gbd += gn[j] * dn[j] + gp[j] * dp[j];
            drel = dmax(drel, dmax(fabs(dn[j]) / (1.0 + fabs(nn[j])), fabs(dp[j]) / (1.0 + fabs(pp[j])))); dym = dmax(dym, fabs(yn[j] - lam[j]));
        }
        // the switching condition [WB (19)] alpha (-gbd)^s_phi > theta^s_theta as alpha > sw = theta^s_theta / (-gbd)^s_phi, the quotient through two logarithms and an
        // exponential (a third of two pow calls); theta = 0 gives 0, gbd -> 0 gives inf, both zero NaN: the comparisons below come out as with the powers
        const double sw = gbd < 0.0 ? exp(kSTheta * log(theta) - kSPhi * log(-gbd)) : INFINITY;
        double a_min = kGammaTheta;
        if (gbd < 0.0) {
            a_min = dmin(kGammaTheta, kGammaPhi * theta / (-gbd));
            if (theta <= theta_min) a_min = dmin(a_min, sw);
        }
        a_min *= kAlphaMinFrac;
        if (theta_max < 0.0) { theta_max = kRestoThetaMaxFact * dmax(1.0, theta); theta_min = kThetaMinFact * dmax(1.0, theta); }
        auto ftype = [&](double alpha_) { return (theta == 0.0 && gbd > 0.0 && gbd < 100.0 * kEps) || (gbd < 0.0 && alpha_ > sw); };
        double xt[NV], nt[MC], pt[MC], ct[MC], theta_t = 0.0, phi_t = 0.0;
        bool ok_t = false;
        auto trial = [&](double a_, const double (&dx_)[NV], const double (&dn_)[MC], const double (&dp_)[MC]) {
            MPC_UNROLL for (int i = 0; i < NV; i++) xt[i] = x[i] + a_ * dx_[i];
            MPC_UNROLL for (int j = 0; j < MC; j++) { nt[j] = nn[j] + a_ * dn_[j]; pt[j] = pp[j] + a_ * dp_[j]; }
            cons(xt, nt, pt, ct);
            const double ft = objective(xt, nt, pt, mu);
            ok_t = finite_all(ft); theta_t = 0.0;
            MPC_UNROLL for (int j = 0; j < MC; j++) { theta_t += fabs(ct[j]); ok_t = ok_t && finite_all(ct[j]); }
            phi_t = barrier(xt, nt, pt, ft, mu);
            if (!ok_t || !finite_all(phi_t)) { ok_t = false; theta_t = INFINITY; phi_t = INFINITY; }
        };
        auto acceptable = [&](double alpha_) {
            if (!ok_t || theta_t > theta_max) return false;
            bool ok_;
            if (alpha_ > 0.0 && ftype(alpha_) && theta <= theta_min) ok_ = le_tol(phi_t - phi, kEtaPhi * alpha_ * gbd, phi);
            else {
                if (phi_t > phi) { const double bas = fabs(phi) > 10.0 ? log10(fabs(phi)) : 1.0; if (log10(phi_t - phi) > kObjMaxInc + bas) return false; }
                ok_ = le_tol(theta_t, (1.0 - kGammaTheta) * theta, theta) || le_tol(phi_t - phi, -kGammaPhi * theta, phi);
            }
            return ok_ && !filter_rejects(filt, nfilt, phi_t, theta_t);
        };
        bool accepted = false, soc_taken = false;
        double alpha = a_max, a_soc = a_max;
        double dsx[NV], dsn[MC], dsp[MC], ys[MC];
        bool tiny = drel < kTinyStepTol && theta <= 1e-4;
        if (tiny) {
            trial(a_max, dx, dn, dp);
            if (ok_t) { accepted = true; tiny_flag = tiny_last; tiny_last = dym < kTinyStepYTol; }
            else tiny = false;
        }
        if (!tiny) {
            tiny_last = false;
            int n_steps = 0;
            while (alpha > a_min || n_steps == 0) {
                trial(alpha, dx, dn, dp);
                if (acceptable(alpha)) { accepted = true; break; }
                if (ok_t && n_steps == 0 && theta <= theta_t) {      // second-order correction
                    double cs[MC], theta_old = 0.0, th_s = theta_t;
                    MPC_UNROLL for (int j = 0; j < MC; j++) cs[j] = cb[j];
                    int cnt = 0;
                    a_soc = alpha;
                    while (cnt < kMaxSoc && !accepted && (cnt == 0 || th_s <= kKappaSoc * theta_old)) {
                        theta_old = th_s;
                        MPC_UNROLL for (int j = 0; j < MC; j++) cs[j] = a_soc * cs[j] + ct[j];
                        direction(cs, dsx, dsn, dsp, ys);
                        a_soc = max_step(dsx, dsn, dsp);
                        trial(a_soc, dsx, dsn, dsp);
                        if (acceptable(alpha)) { accepted = true; soc_taken = true; }
                        else { cnt++; th_s = theta_t; if (!ok_t) break; }
                    }
                    if (accepted) break;
                }
                alpha *= 0.5;
                n_steps++;
            }
            if (!accepted) { status = kRsFailed; break; }      // (no restoration inside the restoration phase: 'Restoration_Failed')
            if (!ftype(alpha) || !le_tol(phi_t - phi, kEtaPhi * alpha * gbd, phi)) nfilt = filter_add(filt, nfilt, phi - kGammaPhi * theta, (1.0 - kGammaTheta) * theta);
        }
        // the accepted point: multiplier steps of the direction that was taken
        const double a_pr = soc_taken ? a_soc : alpha;
        double adu = 1.0, dzl[NV], dzh[NV], dzn[MC], dzp[MC];
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            const double d_ = soc_taken ? dsx[i] : dx[i];
            dzl[i] = fl[i] ? mu / sl[i] - zl[i] - zl[i] / sl[i] * d_ : 0.0;
            dzh[i] = fh[i] ? mu / sh[i] - zh[i] + zh[i] / sh[i] * d_ : 0.0;
            if (fl[i]) adu = ratio(adu, zl[i], dzl[i]);
            if (fh[i]) adu = ratio(adu, zh[i], dzh[i]);
        }
        MPC_UNROLL for (int j = 0; j < MC; j++) {
            dzn[j] = mu / sn[j] - zn[j] - zn[j] / sn[j] * (soc_taken ? dsn[j] : dn[j]);
            dzp[j] = mu / sp[j] - zp[j] - zp[j] / sp[j] * (soc_taken ? dsp[j] : dp[j]);
            adu = ratio(adu, zn[j], dzn[j]); adu = ratio(adu, zp[j], dzp[j]);
        }
        auto clampz = [&](double z, double s_) { return dmin(dmax(z, mu / (kKappaSigma * s_)), kKappaSigma * mu / s_); };
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            x[i] = xt[i];
            const double s1 = fl[i] ? safe_slack(x[i], lo[i], zl[i], mu, true) : 1.0, s2 = fh[i] ? safe_slack(x[i], hi[i], zh[i], mu, false) : 1.0;
            zl[i] += adu * dzl[i]; zh[i] += adu * dzh[i];
            if (fl[i]) zl[i] = clampz(zl[i], s1);
            if (fh[i]) zh[i] = clampz(zh[i], s2);
        }
        MPC_UNROLL for (int j = 0; j < MC; j++) {
            nn[j] = nt[j]; pp[j] = pt[j];
            const double s1 = safe_slack(nn[j], nlo[j], zn[j], mu, true), s2 = safe_slack(pp[j], plo[j], zp[j], mu, true);
            zn[j] += adu * dzn[j]; zp[j] += adu * dzp[j];
            zn[j] = clampz(zn[j], s1); zp[j] = clampz(zp[j], s2);
            lam[j] += a_pr * ((soc_taken ? ys[j] : yn[j]) - lam[j]);
        }
        it++;
    }
    MPC_UNROLL for (int i = 0; i < NV; i++) { io.x[i] = x[i]; io.lo[i] = lo[i]; io.hi[i] = hi[i]; }      // (bounds the restoration moved stay moved)
    io.it = it;
    return status;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Target: min fss(xs, us, ys)  s.t.  Fx_model(xs, us, d) - xs = 0,  xs + Cd d - ys = 0,  boxes   (Target_Calc.py:20-161 with
// StateFeedback outputs).  The same algorithm as ipm_stage; the Newton system is reduced to the nu inputs:
// ys and xs follow from the two (linearised) equalities, so the inertia test is the sign of the nu x nu reduced Hessian.  One instance per
// lane (or wave-uniform: every lane computes it).  v = [xs; us; ys] comes in as the first guess (MPC_code.py:696-700).
// ---------------------------------------------------------------------------------------------------------------------------------
template <class M, int FS>      // filt: this solve's filter list (FS: stride between its words, see filter_rejects)
__device__ __forceinline__ int target_ipm(double (&v)[M::NX + M::NU + M::NY], const double *d, const double (*Bd)[M::ND > 0 ? M::ND : 1],
                                          const double (*Cd)[M::ND > 0 ? M::ND : 1], const double *lo_in, const double *hi_in, double t, double h,
                                          const double tol, const int max_iter, int &iters, double *const filt, double *const tpk)
{
    constexpr int NX = M::NX, NU = M::NU, NY = M::NY, ND = M::ND, NV = NX + NU + NY, NP = NX + NU, NPP = NP * (NP + 1) / 2;
    static_assert(NY == NX, "StateFeedback outputs");
    bool fl[NV], fh[NV];
    // bound multipliers, this solve's own bounds and the slacks live in LDS as in ipm_stage (tpk: this lane's column of 6 NV rows of 64 doubles) - kept in
    // registers they were three quarters of this kernel's 840 scratch accesses per iteration
    const LRows<true> zl{tpk, 0.0}, zh{tpk + 64 * NV, 0.0}, lo{tpk + 128 * NV, 0.0}, hi{tpk + 192 * NV, 0.0}, sl{tpk + 256 * NV, 1.0}, sh{tpk + 320 * NV, 1.0};
    double lam1[NX], lam2[NY], dmp[NV];
    int nbi = 0;
    MPC_UNROLL for (int i = 0; i < NV; i++) {
        lo[i] = relax_lo(lo_in[i]); hi[i] = relax_hi(hi_in[i]);
        fl[i] = fin(lo_in[i]); fh[i] = fin(hi_in[i]); zl[i] = fl[i] ? 1.0 : 0.0; zh[i] = fh[i] ? 1.0 : 0.0; nbi += (fl[i] ? 1 : 0) + (fh[i] ? 1 : 0);
        dmp[i] = (fl[i] && !fh[i]) ? 1.0 : ((fh[i] && !fl[i]) ? -1.0 : 0.0);
    }
    // scaling of the objective at the caller's point, then the push into the box
    double df = 1.0;
    {
        double f_, g_[NV], H_[NV][NV], gm = 0.0;
        M::fss(v, &f_, g_, H_);
        MPC_UNROLL for (int i = 0; i < NV; i++) gm = dmax(gm, fabs(g_[i]));
        df = gm > kScaleMaxGrad ? dmax(kScaleMaxGrad / gm, kScaleMin) : 1.0;
    }
    MPC_UNROLL for (int i = 0; i < NV; i++) v[i] = push_in(v[i], lo[i], hi[i]);
    MPC_UNROLL for (int i = 0; i < NX; i++) lam1[i] = 0.0;
    MPC_UNROLL for (int i = 0; i < NY; i++) lam2[i] = 0.0;
    const double nb = (double)nbi, meq = (double)(NX + NY);
    const double mu_min = dmin(tol, kComplInfTol) / (kKappaEps + 1.0);
    double mu = kMuInit, tau = dmax(kTauMin, 1.0 - kMuInit), delta_last = 0.0, theta_max = -1.0, theta_min = -1.0;
    int status = kStMaxIter, nfilt = 0, acc_count = 0;
    bool tiny_last = false, tiny_flag = false;
    iters = 0;
    // values of cost and constraints at a point (trial points of the line search)
    auto values = [&](const double (&w)[NV], double &f_, double (&c1_)[NX], double (&c2_)[NY]) {
        typename M::Ctx cx;
        MPC_UNROLL for (int i = 0; i < NU; i++) { cx.u[i] = w[NX + i]; cx.us[i] = 0.0; }
        MPC_UNROLL for (int i = 0; i < ND; i++) cx.d[i] = d[i];
        MPC_UNROLL for (int i = 0; i < NX; i++) cx.xs[i] = 0.0;
        double Fx[NX], g_[NV], H_[NV][NV];
        rk4_plain<typename M::Mdl>(w, cx, t, true, h, M::MX, Fx);
        M::fss(w, &f_, g_, H_);
        f_ *= df;
        MPC_UNROLL for (int i = 0; i < NX; i++) { double a = Fx[i] - w[i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += Bd[i][j] * d[j]; c1_[i] = a; }
        MPC_UNROLL for (int i = 0; i < NY; i++) { double a = w[i] - w[NX + NU + i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += Cd[i][j] * d[j]; c2_[i] = a; }
    };
    auto barrier = [&](const double (&w)[NV], double f_, double mu_) {      // with the safe slacks of a trial point (its moved bounds are not kept)
        double ph = f_;
        LogSum lg;
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            double bl_ = lo[i], bh_ = hi[i];
            const double s1 = fl[i] ? safe_slack(w[i], bl_, zl[i], mu_, true) : 1.0, s2 = fh[i] ? safe_slack(w[i], bh_, zh[i], mu_, false) : 1.0;
            lg.mul(s1, fl[i]); lg.mul(s2, fh[i]);
            if (dmp[i] != 0.0) ph += kKappaD * mu_ * (dmp[i] > 0.0 ? s1 : s2);
        }
        return ph - mu_ * lg.value();
    };
    int it = 0;
    for (;; it++) {
        iters = it;
        typename M::Ctx cx;
        MPC_UNROLL for (int i = 0; i < NU; i++) { cx.u[i] = v[NX + i]; cx.us[i] = 0.0; }
        MPC_UNROLL for (int i = 0; i < ND; i++) cx.d[i] = d[i];
        MPC_UNROLL for (int i = 0; i < NX; i++) cx.xs[i] = 0.0;
        double Fx[NX], S[NX][NP], T[NX][NPP];
        rk4_sens2<typename M::Mdl>(v, cx, t, true, h, M::MX, Fx, S, T);
        EC_LDS_FENCE();
        double f, g[NV], Hc[NV][NV];
        M::fss(v, &f, g, Hc);
        f *= df;
        MPC_UNROLL for (int i = 0; i < NV; i++) { g[i] *= df; MPC_UNROLL for (int j = 0; j < NV; j++) Hc[i][j] *= df; }
        double c1[NX], c2[NY], J1[NX][NX];
        MPC_UNROLL for (int i = 0; i < NX; i++) {
            double a = Fx[i] - v[i];
            MPC_UNROLL for (int j = 0; j < ND; j++) a += Bd[i][j] * d[j];
            c1[i] = a;
            MPC_UNROLL for (int j = 0; j < NX; j++) J1[i][j] = S[i][j] - (i == j ? 1.0 : 0.0);
        }
        MPC_UNROLL for (int i = 0; i < NY; i++) { double a = v[i] - v[NX + NU + i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += Cd[i][j] * d[j]; c2[i] = a; }
        // null-space basis Z = [Zx; I; Zx] of the linearised equalities
        double W[NX][NX];
        if (!gj_inverse<NX>(J1, W)) { status = kStFailed; break; }
        double Z[NV][NU];
        MPC_UNROLL for (int i = 0; i < NX; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NX; l++) a -= W[i][l] * S[l][NX + j]; Z[i][j] = a; Z[NP + i][j] = a; } }
        MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) Z[NX + i][j] = (i == j) ? 1.0 : 0.0; }
        // multipliers of the two equality blocks from the stationarity rows of ys and xs, given (Hessian x step + gradient) in Hd
        auto mults = [&](const double (&Hd)[NV], double (&l1n)[NX], double (&l2n)[NY]) {
            MPC_UNROLL for (int i = 0; i < NY; i++) l2n[i] = Hd[NP + i];
            MPC_UNROLL for (int i = 0; i < NX; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NX; j++) a -= W[j][i] * (Hd[j] + l2n[j]); l1n[i] = a; }
        };
        if (it == 0) {      // least-squares multipliers [WB (36)]: identity for the Hessian, gradient g - zl + zh, no constraint residual
            double gg[NV], Hr[NU][NU], rr[NU], dus[NU], Hd[NV], l1n[NX], l2n[NY];
            MPC_UNROLL for (int i = 0; i < NV; i++) gg[i] = g[i] - zl[i] + zh[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NV; l++) a += Z[l][i] * Z[l][j]; Hr[i][j] = a; }
                double a = 0.0;
                MPC_UNROLL for (int l = 0; l < NV; l++) a += Z[l][i] * gg[l];
                rr[i] = a;
            }
            if (sym_inverse<NU>(Hr)) {
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) a -= Hr[i][j] * rr[j]; dus[i] = a; }
                MPC_UNROLL for (int i = 0; i < NV; i++) { double a = gg[i]; MPC_UNROLL for (int j = 0; j < NU; j++) a += Z[i][j] * dus[j]; Hd[i] = a; }
                mults(Hd, l1n, l2n);
                double ym = 0.0;
                MPC_UNROLL for (int i = 0; i < NX; i++) ym = dmax(ym, finite_all(l1n[i]) ? fabs(l1n[i]) : INFINITY);
                MPC_UNROLL for (int i = 0; i < NY; i++) ym = dmax(ym, finite_all(l2n[i]) ? fabs(l2n[i]) : INFINITY);
                if (ym <= kYInitMax) { MPC_UNROLL for (int i = 0; i < NX; i++) lam1[i] = l1n[i]; MPC_UNROLL for (int i = 0; i < NY; i++) lam2[i] = l2n[i]; }
            }
        }
        // Hessian of the Lagrangian of the scaled problem: df cost + lam1' Fx
        double H[NV][NV];
        MPC_UNROLL for (int i = 0; i < NV; i++) { MPC_UNROLL for (int j = 0; j < NV; j++) H[i][j] = Hc[i][j]; }
        MPC_UNROLL for (int a = 0; a < NP; a++) { MPC_UNROLL for (int b = 0; b < NP; b++) { double s = 0.0; MPC_UNROLL for (int i = 0; i < NX; i++) s += lam1[i] * T[i][pair_idx<NP>(a, b)]; H[a][b] += s; } }
        double e_st = 0.0, e_c = 0.0, s_l = 0.0, s_z = 0.0, cmax = -INFINITY, cmin = INFINITY, theta = 0.0;
        bool finite = finite_all(f);
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            sl[i] = fl[i] ? safe_slack(v[i], lo[i], zl[i], mu, true) : 1.0; sh[i] = fh[i] ? safe_slack(v[i], hi[i], zh[i], mu, false) : 1.0;
            double r = g[i] - zl[i] + zh[i];
            if (i < NX) { MPC_UNROLL for (int j = 0; j < NX; j++) r += J1[j][i < NX ? i : 0] * lam1[j]; r += lam2[i < NY ? i : 0]; }      // C = I
            else if (i < NP) { MPC_UNROLL for (int j = 0; j < NX; j++) r += S[j][i < NP ? i : 0] * lam1[j]; }
            else r -= lam2[i >= NP ? i - NP : 0];
            e_st = dmax(e_st, fabs(r)); s_z += zl[i] + zh[i];
            finite = finite && finite_all(r) && finite_all(v[i]);
            if (fl[i]) { cmax = dmax(cmax, sl[i] * zl[i]); cmin = dmin(cmin, sl[i] * zl[i]); }
            if (fh[i]) { cmax = dmax(cmax, sh[i] * zh[i]); cmin = dmin(cmin, sh[i] * zh[i]); }
        }
        MPC_UNROLL for (int i = 0; i < NX; i++) { e_c = dmax(e_c, fabs(c1[i])); theta += fabs(c1[i]); s_l += fabs(lam1[i]); finite = finite && finite_all(c1[i]); }
        MPC_UNROLL for (int i = 0; i < NY; i++) { e_c = dmax(e_c, fabs(c2[i])); theta += fabs(c2[i]); s_l += fabs(lam2[i]); finite = finite && finite_all(c2[i]); }
        if (!finite) { status = kStFailed; break; }
        const double s_d = dmax(kSMax, (s_l + s_z) / dmax(meq + nb, 1.0)) / kSMax, s_c = dmax(kSMax, s_z / dmax(nb, 1.0)) / kSMax;
        auto compl_ = [&](double m_) { return nb > 0.0 ? dmax(cmax - m_, m_ - cmin) : 0.0; };
        auto err = [&](double m_) { return dmax(dmax(e_st / s_d, e_c), compl_(m_) / s_c); };
        {
            const double e0_ = err(0.0), c0_ = compl_(0.0);
#ifdef EC_TRACE_TGT      /* diagnostic build only (tools/scratch): the first instance of a launch prints its iterations */
            if (blockIdx.x == 0 && threadIdx.x == 0) printf("  it=%3d mu=%.3e E0=%.6e e_st=%.3e e_c=%.3e compl=%.3e theta=%.6e\n", it, mu, e0_, e_st, e_c, c0_, theta);
#endif
            if (e0_ <= tol && e_st <= kDualInfTol && e_c <= kConstrViolTol && c0_ <= kComplInfTol) { status = kStSolved; break; }
            if (e0_ <= kAccTol && e_st <= kAccDualInfTol && e_c <= kAccConstrViolTol && c0_ <= kAccComplInfTol) { if (++acc_count >= kAccIter) { status = kStSolved; break; } }
            else acc_count = 0;
        }
        if (it >= max_iter) break;
        {
            bool mu_changed = false, stop_tiny = false;
            while (err(mu) <= kKappaEps * mu || tiny_flag) {
                const double new_mu = dmax(dmin(kKappaMu * mu, mu * sqrt(mu)), mu_min);
                if (new_mu == mu) { stop_tiny = tiny_flag; break; }
                mu = new_mu; mu_changed = true; tiny_flag = false;
            }
            if (stop_tiny) { status = kStMaxIter; break; }
            tiny_flag = false;
            if (mu_changed) { nfilt = 0; tau = dmax(kTauMin, 1.0 - mu); }
        }
        double Sg[NV], gt[NV];
        double phi = f;
        LogSum lgp;
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            const double il = fl[i] ? 1.0 / sl[i] : 0.0, ih = fh[i] ? 1.0 / sh[i] : 0.0;
            Sg[i] = zl[i] * il + zh[i] * ih; gt[i] = g[i] - mu * il + mu * ih + kKappaD * mu * dmp[i];
            lgp.mul(sl[i], fl[i]); lgp.mul(sh[i], fh[i]);
            if (dmp[i] != 0.0) phi += kKappaD * mu * (dmp[i] > 0.0 ? sl[i] : sh[i]);
        }
        phi -= mu * lgp.value();
        // reduced Hessian with the shift delta while it lacks positive curvature
        double Hr[NU][NU];
        double delta = 0.0;
        bool failed = false;
        for (;;) {
            double HZ[NV][NU];
            MPC_UNROLL for (int i = 0; i < NV; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) { double a = (Sg[i] + delta) * Z[i][j]; MPC_UNROLL for (int l = 0; l < NV; l++) a += H[i][l] * Z[l][j]; HZ[i][j] = a; } }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NV; l++) a += Z[l][i] * HZ[l][j]; Hr[i][j] = a; } }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (Hr[i][j] + Hr[j][i]); Hr[i][j] = a; Hr[j][i] = a; } }
            if (sym_inverse<NU>(Hr)) break;
            delta = delta == 0.0 ? dmax(kDeltaFirst, delta_last / 3.0) : delta * (delta_last == 0.0 ? 100.0 : 8.0);
            if (delta > kDeltaMax) { failed = true; break; }
        }
        if (failed) { status = kStFailed; break; }
        if (delta > 0.0) delta_last = delta;
        // Newton step for a constraint residual (c1_, c2_): particular step of the linearised equalities + reduced step; new multipliers
        auto direction = [&](const double (&c1_)[NX], const double (&c2_)[NY], double (&dv)[NV], double (&l1n)[NX], double (&l2n)[NY]) {
            double sp[NV], hs[NV], rr[NU], dus[NU], Hd[NV];
            MPC_UNROLL for (int i = 0; i < NX; i++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NX; l++) a -= W[i][l] * c1_[l]; sp[i] = a; sp[NP + i] = a + c2_[i]; }
            MPC_UNROLL for (int i = 0; i < NU; i++) sp[NX + i] = 0.0;
            MPC_UNROLL for (int i = 0; i < NV; i++) { double a = (Sg[i] + delta) * sp[i] + gt[i]; MPC_UNROLL for (int l = 0; l < NV; l++) a += H[i][l] * sp[l]; hs[i] = a; }
            MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NV; l++) a += Z[l][i] * hs[l]; rr[i] = a; }
            MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) a -= Hr[i][j] * rr[j]; dus[i] = a; }
            MPC_UNROLL for (int i = 0; i < NV; i++) { double a = sp[i]; MPC_UNROLL for (int j = 0; j < NU; j++) a += Z[i][j] * dus[j]; dv[i] = a; }
            MPC_UNROLL for (int i = 0; i < NV; i++) { double a = (Sg[i] + delta) * dv[i] + gt[i]; MPC_UNROLL for (int l = 0; l < NV; l++) a += H[i][l] * dv[l]; Hd[i] = a; }
            mults(Hd, l1n, l2n);
        };
        auto ratio = [&](double a, double vv, double dvv) { return dvv < 0.0 ? dmin(a, -tau * vv / dvv) : a; };
        auto max_step = [&](const double (&dv_)[NV]) { MinRatio mr; MPC_UNROLL for (int i = 0; i < NV; i++) { mr.add(sl[i], dv_[i], tau, fl[i]); mr.add(sh[i], -dv_[i], tau, fh[i]); } return mr.value(); };
        double dv[NV], l1n[NX], l2n[NY];
        direction(c1, c2, dv, l1n, l2n);
        const double a_max = max_step(dv);
        double gbd = 0.0, drel = 0.0, dym = 0.0;
        MPC_UNROLL for (int i = 0; i < NV; i++) { gbd += gt[i] * dv[i]; drel = dmax(drel, fabs(dv[i]) / (1.0 + fabs(v[i]))); }
        MPC_UNROLL for (int i = 0; i < NX; i++) dym = dmax(dym, fabs(l1n[i] - lam1[i]));
        MPC_UNROLL for (int i = 0; i < NY; i++) dym = dmax(dym, fabs(l2n[i] - lam2[i]));
        // ---- filter line search ------------------------------------------------------------------------------------------------------------------
        // the switching condition [WB (19)] alpha (-gbd)^s_phi > theta^s_theta as alpha > sw = theta^s_theta / (-gbd)^s_phi, the quotient through two logarithms and an
        // exponential (a third of two pow calls); theta = 0 gives 0, gbd -> 0 gives inf, both zero NaN: the comparisons below come out as with the powers
        const double sw = gbd < 0.0 ? exp(kSTheta * log(theta) - kSPhi * log(-gbd)) : INFINITY;
        double a_min = kGammaTheta;
        if (gbd < 0.0) {
            a_min = dmin(kGammaTheta, kGammaPhi * theta / (-gbd));
            if (theta <= theta_min) a_min = dmin(a_min, sw);
        }
        a_min *= kAlphaMinFrac;
        if (theta_max < 0.0) { theta_max = kThetaMaxFact * dmax(1.0, theta); theta_min = kThetaMinFact * dmax(1.0, theta); }
        auto ftype = [&](double alpha_) { return (theta == 0.0 && gbd > 0.0 && gbd < 100.0 * kEps) || (gbd < 0.0 && alpha_ > sw); };
        double vt[NV], c1t[NX], c2t[NY], theta_t = 0.0, phi_t = 0.0;
        bool ok_t = false;
        auto trial = [&](double a_, const double (&d_)[NV]) {
            MPC_UNROLL for (int i = 0; i < NV; i++) vt[i] = v[i] + a_ * d_[i];
            double ft;
            values(vt, ft, c1t, c2t);
            EC_LDS_FENCE();
            ok_t = finite_all(ft); theta_t = 0.0;
            MPC_UNROLL for (int i = 0; i < NX; i++) { theta_t += fabs(c1t[i]); ok_t = ok_t && finite_all(c1t[i]); }
            MPC_UNROLL for (int i = 0; i < NY; i++) { theta_t += fabs(c2t[i]); ok_t = ok_t && finite_all(c2t[i]); }
            phi_t = barrier(vt, ft, mu);
            if (!ok_t || !finite_all(phi_t)) { ok_t = false; theta_t = INFINITY; phi_t = INFINITY; }
        };
        auto acceptable = [&](double alpha_) {
            if (!ok_t || theta_t > theta_max) return false;
            bool ok_;
            if (alpha_ > 0.0 && ftype(alpha_) && theta <= theta_min) ok_ = le_tol(phi_t - phi, kEtaPhi * alpha_ * gbd, phi);
            else {
                if (phi_t > phi) { const double bas = fabs(phi) > 10.0 ? log10(fabs(phi)) : 1.0; if (log10(phi_t - phi) > kObjMaxInc + bas) return false; }
                ok_ = le_tol(theta_t, (1.0 - kGammaTheta) * theta, theta) || le_tol(phi_t - phi, -kGammaPhi * theta, phi);
            }
            return ok_ && !filter_rejects<FS>(filt, nfilt, phi_t, theta_t);
        };
        bool accepted = false, soc_taken = false;
        double alpha = a_max, a_soc = a_max;
        double ds[NV], l1s[NX], l2s[NY];
        bool tiny = drel < kTinyStepTol && theta <= 1e-4;
        if (tiny) {
            trial(a_max, dv);
            if (ok_t) { accepted = true; tiny_flag = tiny_last; tiny_last = dym < kTinyStepYTol; }
            else tiny = false;
        }
        if (!tiny) {
            tiny_last = false;
            int n_steps = 0;
            while (alpha > a_min || n_steps == 0) {
                trial(alpha, dv);
                if (acceptable(alpha)) { accepted = true; break; }
                if (ok_t && n_steps == 0 && theta <= theta_t) {      // second-order correction
                    double cs1[NX], cs2[NY], theta_old = 0.0, th_s = theta_t;
                    MPC_UNROLL for (int i = 0; i < NX; i++) cs1[i] = c1[i];
                    MPC_UNROLL for (int i = 0; i < NY; i++) cs2[i] = c2[i];
                    int cnt = 0;
                    a_soc = alpha;
                    while (cnt < kMaxSoc && !accepted && (cnt == 0 || th_s <= kKappaSoc * theta_old)) {
                        theta_old = th_s;
                        MPC_UNROLL for (int i = 0; i < NX; i++) cs1[i] = a_soc * cs1[i] + c1t[i];
                        MPC_UNROLL for (int i = 0; i < NY; i++) cs2[i] = a_soc * cs2[i] + c2t[i];
                        direction(cs1, cs2, ds, l1s, l2s);
                        a_soc = max_step(ds);
                        trial(a_soc, ds);
                        if (acceptable(alpha)) { accepted = true; soc_taken = true; }
                        else { cnt++; th_s = theta_t; if (!ok_t) break; }
                    }
                    if (accepted) break;
                }
                alpha *= 0.5;
                n_steps++;
            }
            auto augment = [&]() { nfilt = filter_add<FS>(filt, nfilt, phi - kGammaPhi * theta, (1.0 - kGammaTheta) * theta); };      // the filter grows by the current point
            if (!accepted) {
                // ---- IPOPT's restoration phase ---------------------------------------------------------------------------------------------------
                if (theta <= 1e-2 * tol) { status = kStMaxIter; break; }      // 'Restoration_Failed' at a feasible point: the reference accepts the point
                augment();                                                    // the point the restoration starts from is never returned to
                TgtRestoIO<M> io;
                MPC_UNROLL for (int i = 0; i < NV; i++) { io.x[i] = v[i]; io.lo[i] = lo[i]; io.hi[i] = hi[i]; io.zl[i] = zl[i]; io.zh[i] = zh[i]; }
                MPC_UNROLL for (int i = 0; i < NX; i++) io.c[i] = c1[i];
                MPC_UNROLL for (int i = 0; i < NY; i++) io.c[NX + i] = c2[i];
                MPC_UNROLL for (int j = 0; j < ND; j++) { io.d[j] = d[j]; MPC_UNROLL for (int i = 0; i < NX; i++) io.Bd[i][j] = Bd[i][j]; MPC_UNROLL for (int i = 0; i < NY; i++) io.Cd[i][j] = Cd[i][j]; }
                for (int e = 0; e < 2 * kFilterCap; e++) io.filt[e] = e < 2 * nfilt ? filt[e * FS] : 0.0;
                io.mu = mu; io.theta = theta; io.phi = phi; io.df = df; io.t = t; io.h = h; io.tol = tol; io.nfilt = nfilt; io.max_iter = max_iter; io.it = it + 1;
                const int rs = target_resto<M>(io);
                double xr_[NV];
                MPC_UNROLL for (int i = 0; i < NV; i++) { xr_[i] = io.x[i]; lo[i] = io.lo[i]; hi[i] = io.hi[i]; }
                const int it_r = io.it;
                it = it_r - 1; iters = it;
                if (rs != kRsRestored) {
                    if (rs == kRsLimit) status = kStMaxIter;
                    else if (rs == kRsConverged) {      // the restoration problem has a minimiser here: infeasible, or feasible and not acceptable
                        double f_, a1[NX], a2[NY], cm = 0.0;
                        values(xr_, f_, a1, a2);
                        MPC_UNROLL for (int i = 0; i < NX; i++) cm = dmax(cm, fabs(a1[i]));
                        MPC_UNROLL for (int i = 0; i < NY; i++) cm = dmax(cm, fabs(a2[i]));
                        status = cm <= kRestoFeasFact * tol ? kStMaxIter : kStFailed;
                    } else status = kStMaxIter;      // 'Restoration_Failed': the reference accepts the point
                    break;
                }
                // back from the restoration: bound multipliers as if the whole move had been one Newton step, reset to 1 beyond 1e3; equality multipliers zero
                double adu_ = 1.0, zmax = 0.0, dzl_[NV], dzh_[NV], st1[NV], st2[NV];
                MPC_UNROLL for (int i = 0; i < NV; i++) {
                    st1[i] = fl[i] ? safe_slack(xr_[i], lo[i], zl[i], mu, true) : 1.0; st2[i] = fh[i] ? safe_slack(xr_[i], hi[i], zh[i], mu, false) : 1.0;
                    dzl_[i] = fl[i] ? mu / sl[i] - zl[i] - zl[i] / sl[i] * (st1[i] - sl[i]) : 0.0;
                    dzh_[i] = fh[i] ? mu / sh[i] - zh[i] - zh[i] / sh[i] * (st2[i] - sh[i]) : 0.0;
                    if (fl[i]) adu_ = ratio(adu_, zl[i], dzl_[i]);
                    if (fh[i]) adu_ = ratio(adu_, zh[i], dzh_[i]);
                }
                MPC_UNROLL for (int i = 0; i < NV; i++) { zl[i] += adu_ * dzl_[i]; zh[i] += adu_ * dzh_[i]; zmax = dmax(zmax, dmax(zl[i], zh[i])); }
                MPC_UNROLL for (int i = 0; i < NV; i++) {
                    if (zmax > kRestoBoundMultReset) { zl[i] = fl[i] ? 1.0 : 0.0; zh[i] = fh[i] ? 1.0 : 0.0; }
                    v[i] = xr_[i];
                    if (fl[i]) zl[i] = dmin(dmax(zl[i], mu / (kKappaSigma * st1[i])), kKappaSigma * mu / st1[i]);
                    if (fh[i]) zh[i] = dmin(dmax(zh[i], mu / (kKappaSigma * st2[i])), kKappaSigma * mu / st2[i]);
                }
                MPC_UNROLL for (int i = 0; i < NX; i++) lam1[i] = 0.0;
                MPC_UNROLL for (int i = 0; i < NY; i++) lam2[i] = 0.0;
                continue;      // (the loop's increment counts the restoration's return as an iteration)
            }
            if (!ftype(alpha) || !le_tol(phi_t - phi, kEtaPhi * alpha * gbd, phi)) augment();
        }
        // the accepted point vt: multiplier steps of the direction that was taken
        const double a_pr = soc_taken ? a_soc : alpha;
        double dzl[NV], dzh[NV];
        MinRatio mz;
        MPC_UNROLL for (int i = 0; i < NV; i++) {      // (as ipm_stage: one division per multiplier step, one per new multiplier, one for the step length)
            const double d_ = soc_taken ? ds[i] : dv[i];
            dzl[i] = fl[i] ? (mu - zl[i] * d_) / sl[i] - zl[i] : 0.0;
            dzh[i] = fh[i] ? (mu + zh[i] * d_) / sh[i] - zh[i] : 0.0;
            mz.add(zl[i], dzl[i], tau, fl[i]); mz.add(zh[i], dzh[i], tau, fh[i]);
        }
        const double adu = mz.value();
        auto newz = [&](bool f_, double z_, double dz_, double s_) { const double q = mu / s_, zn_ = z_ + adu * dz_; return f_ ? dmin(dmax(zn_, q * (1.0 / kKappaSigma)), kKappaSigma * q) : zn_; };
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            v[i] = vt[i];
            const double s1 = fl[i] ? safe_slack(v[i], lo[i], zl[i], mu, true) : 1.0, s2 = fh[i] ? safe_slack(v[i], hi[i], zh[i], mu, false) : 1.0;
            zl[i] = newz(fl[i], zl[i], dzl[i], s1); zh[i] = newz(fh[i], zh[i], dzh[i], s2);
        }
        MPC_UNROLL for (int i = 0; i < NX; i++) lam1[i] += a_pr * ((soc_taken ? l1s[i] : l1n[i]) - lam1[i]);
        MPC_UNROLL for (int i = 0; i < NY; i++) lam2[i] += a_pr * ((soc_taken ? l2s[i] : l2n[i]) - lam2[i]);
    }
    MPC_UNROLL for (int i = 0; i < NV; i++) v[i] = dmin(dmax(v[i], lo_in[i]), hi_in[i]);      // honor_original_bounds
    return status;
}

}  // namespace enm
