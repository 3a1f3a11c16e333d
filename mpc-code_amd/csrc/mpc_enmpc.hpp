// Economic NMPC with a moving-horizon estimator on the GPU (BASELINE configs 4 and 5): device side.
//
// Mapping: ONE WAVEFRONT = ONE INSTANCE, LANE = STAGE of the horizon (N <= 64).  Everything an interior-point iteration touches lives in
// registers; stages talk to their neighbours through DPP wave shifts and v_readlane broadcasts - no LDS, no HBM workspace.
//
// The three NLPs of a closed-loop step (reference MPC_code.py:485-827 with Ex_ENMPC.py) are solved by one primal-dual interior point
// method whose outer algorithm is the reference solver's (IPOPT at the defaults MPC_code.py:262-263 leaves it at [ext]; restated and
// documented in DESIGN.md section 10; the checker restates it with dense linear algebra): monotone barrier parameter, start pushed into the box, fraction to the
// boundary, exact Hessian of the Lagrangian with inertia correction, scaled optimality error.  What is particular here:
//   * OCP (opt_dyn with ContForm, Control_Calc.py:20-260): every lane integrates ITS shooting interval - state, cost quadrature and
//     their first and second forward sensitivities with respect to (x_k, u_k) through the Runge-Kutta stages (generated code,
//     econcodegen.py) - then the Newton system is factorised by a Riccati recursion over the lanes: the stage update is computed by
//     all lanes at once, lane k's result is the valid one and is broadcast to the next iteration (v_readlane).  The recursion needs
//     Lambda_k = R_k + B_k' P_{k+1} B_k > 0 at every stage, which is IPOPT's inertia condition on the reduced Hessian: when it fails
//     the Hessian is shifted by delta I exactly as there.
//   * MHE (mhe_opt, Utilities.py:825-990): the same recursion with state [x; d], "input" w, a free initial state with the arrival
//     cost, and the output noise v eliminated through its (linear) defining equation.
//   * target (opt_ss, Target_Calc.py:20-161): nx + nu + ny variables; the model's fixed-point equation is eliminated with an LU of
//     (A - I) and the reduced Hessian is nu x nu; computed redundantly by every lane (wave-uniform).
#pragma once
#include "mpc_tp.hpp"
#include "mpc_rk4s2.hpp"

namespace enm {
using namespace mpc;

// ---- the outer algorithm's constants (the checker carries the same) ----------------------------------------------------
constexpr double kPush = 1e-2, kMuInit = 0.1, kKappaEps = 10.0, kKappaMu = 0.2, kTauMin = 0.99, kKappaSigma = 1e10, kSMax = 100.0,
                 kDeltaFirst = 1e-4, kDeltaMax = 1e40;
enum : int { kStSolved = 0, kStMaxIter = 1, kStFailed = 2 };

__device__ __forceinline__ double wave_min(double v) { return -wave_max(-v); }
__device__ __forceinline__ bool finite_all(double v) { return fabs(v) < 1.0e300; }      // false for inf and NaN

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

// ---- one stage of a stage-structured NLP, linearised ------------------------------------------------------------------------------
// x+ = F(x, u), cost l(x, u);  A = F_x, B = F_u, (lx, lu) = grad l, [Q M; M' R] = Hessian of l + pi' F  (pi = costate of x+)
template <int NS, int NU>
struct StageLin {
    double F[NS], A[NS][NS], B[NS][NU], lx[NS], lu[NU], Q[NS][NS], M[NS][NU], R[NU][NU];
};

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
            double r = lane_of(v, kk);
#ifndef EC_BCAST_IN_SGPRS      // (the matrix's failing leg)
            asm("" : "+v"(r));
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
// (k = 0..N-1, lane k holds u_k, x_{k+1}, the costate pi_{k+1} = -(multiplier of x_{k+1} - F_k = 0) and their bound multipliers).
// FREE0 = false: x_0 is given (the OCP; IPOPT makes a variable with equal bounds a parameter, MPC_code.py:734).
// FREE0 = true:  x_0 is a variable with its own box and the arrival cost 1/2 (x_0 - xbar)' Pinv (x_0 - xbar) (the MHE); every lane
//                carries the same copy of it.
// lin(xk, u, pi, L): this lane's stage linearised; term(xn, gv, Hv): terminal cost at this lane's x_{k+1} (used from lane N-1).
// u / xn come in as the first guess (pushed into the box here) and leave as the final iterate.  Returns the status.
// ---------------------------------------------------------------------------------------------------------------------------------
// ST: what is known at compile time about the stage matrices - a_kind(i, j) / b_kind(i, j) = 0 (the entry of A / B is zero), 1 (it is one), 2 (general: read
// L.A[i][j] / L.B[i][j]).  With the loops unrolled the products with zeros and ones disappear and the entries that are never read are never held.
struct DenseStage {
    __device__ static constexpr int a_kind(int, int) { return 2; }
    __device__ static constexpr int b_kind(int, int) { return 2; }
};
#define EC_A(acc, x, i_, j_) do { if (ST::a_kind(i_, j_) == 1) acc += (x); else if (ST::a_kind(i_, j_) == 2) acc += L.A[i_][j_] * (x); } while (0)
#define EC_B(acc, x, i_, j_) do { if (ST::b_kind(i_, j_) == 1) acc += (x); else if (ST::b_kind(i_, j_) == 2) acc += L.B[i_][j_] * (x); } while (0)

template <int NS, int NU, bool FREE0, int SEG, class ST, class LinF, class TermF>
__device__ __forceinline__ int ipm_stage(const int N, const int lane, const bool live, const double (&x0fix)[NS], double (&x0v)[NS], double (&u)[NU], double (&xn)[NS],
                                         double (&pi)[NS], const double (&ulo)[NU], const double (&uhi)[NU], const double (&xlo)[NS],
                                         const double (&xhi)[NS], const double (*Pinv)[NS], const double *xbar, const double tol,
                                         const int max_iter, LinF lin, TermF term, int &iters)
{
    using SG = Seg<SEG>;
    const int k = SG::stage(lane);
    const bool on = k < N;
    bool flu[NU], fhu[NU], flx[NS], fhx[NS];
    double zlu[NU], zhu[NU], zlx[NS], zhx[NS], zl0[NS], zh0[NS];
    int nbl = 0;      // finite bounds per stage
    MPC_UNROLL for (int i = 0; i < NU; i++) { flu[i] = fin(ulo[i]); fhu[i] = fin(uhi[i]); zlu[i] = flu[i] ? 1.0 : 0.0; zhu[i] = fhu[i] ? 1.0 : 0.0; nbl += (flu[i] ? 1 : 0) + (fhu[i] ? 1 : 0); u[i] = push_in(u[i], ulo[i], uhi[i]); }
    int nbx = 0;
    MPC_UNROLL for (int i = 0; i < NS; i++) { flx[i] = fin(xlo[i]); fhx[i] = fin(xhi[i]); zlx[i] = flx[i] ? 1.0 : 0.0; zhx[i] = fhx[i] ? 1.0 : 0.0; nbx += (flx[i] ? 1 : 0) + (fhx[i] ? 1 : 0); xn[i] = push_in(xn[i], xlo[i], xhi[i]); pi[i] = 0.0; }
    MPC_UNROLL for (int i = 0; i < NS; i++) { zl0[i] = (FREE0 && flx[i]) ? 1.0 : 0.0; zh0[i] = (FREE0 && fhx[i]) ? 1.0 : 0.0; if (FREE0) x0v[i] = push_in(x0v[i], xlo[i], xhi[i]); }
    const double nb = (double)(N * (nbl + nbx) + (FREE0 ? nbx : 0)), meq = (double)(N * NS);
    // everything below that looks wave-uniform is uniform per SEGMENT (= per instance): mu, the shifts, status, the iteration count.  A
    // segment that has finished (done) keeps computing with its frozen iterate while its wave neighbours go on; `live` = false marks a
    // segment without an instance (ragged batch)
    double mu = kMuInit, delta_last = 0.0;
    int status = kStMaxIter;
    bool done = !live;
    iters = 0;
    for (int it = 0;; it++) {
        if (!done) iters = it;
        // ---- linearise this lane's stage at (x_k, u_k); x_k is the neighbour's x_{k+1} ------------------------------------------------
        double xk[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) xk[i] = SG::up1(FREE0 ? x0v[i] : x0fix[i], xn[i], k);
        StageLin<NS, NU> L;
        lin(xk, u, pi, L);
        double gv[NS], Hv[NS][NS];
        term(xn, gv, Hv);
        double c[NS], gxA[NS], gnext[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            c[i] = xn[i] - L.F[i];
            double a = L.lx[i];
            MPC_UNROLL for (int j = 0; j < NS; j++) EC_A(a, pi[j], j, i);
            gxA[i] = a;
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { const double sh = SG::dn1(0.0, gxA[i], k); gnext[i] = k == N - 1 ? gv[i] : sh; }
        double slu[NU], shu[NU], slx[NS], shx[NS], sl0[NS], sh0[NS];
        MPC_UNROLL for (int i = 0; i < NU; i++) { slu[i] = flu[i] ? u[i] - ulo[i] : 1.0; shu[i] = fhu[i] ? uhi[i] - u[i] : 1.0; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { slx[i] = flx[i] ? xn[i] - xlo[i] : 1.0; shx[i] = fhx[i] ? xhi[i] - xn[i] : 1.0; sl0[i] = (FREE0 && flx[i]) ? x0v[i] - xlo[i] : 1.0; sh0[i] = (FREE0 && fhx[i]) ? xhi[i] - x0v[i] : 1.0; }
        // ---- optimality error (IPOPT's E_mu with its scaling) ----------------------------------------------------------------------------
        double e_st = 0.0, e_c = 0.0, s_pi = 0.0, s_z = 0.0, cmax = -INFINITY, cmin = INFINITY;
        bool finite = true;
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
            e_st = dmax(e_st, fabs(r)); e_c = dmax(e_c, fabs(c[i])); s_pi += fabs(pi[i]); s_z += zlx[i] + zhx[i];
            finite = finite && finite_all(r) && finite_all(c[i]) && finite_all(xn[i]);
            if (flx[i]) { cmax = dmax(cmax, slx[i] * zlx[i]); cmin = dmin(cmin, slx[i] * zlx[i]); }
            if (fhx[i]) { cmax = dmax(cmax, shx[i] * zhx[i]); cmin = dmin(cmin, shx[i] * zhx[i]); }
        }
        e_st = SG::max(on ? e_st : 0.0); e_c = SG::max(on ? e_c : 0.0); s_pi = SG::sum(on ? s_pi : 0.0); s_z = SG::sum(on ? s_z : 0.0);
        cmax = SG::max(on ? cmax : -INFINITY); cmin = SG::min(on ? cmin : INFINITY);
        double g0[NS], ga0[NS];      // gradient with respect to the free initial state: stage 0's part + arrival cost
        if (FREE0) {
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                const double ga = SG::bcast(gxA[i], 0, lane);
                double a = ga;
                MPC_UNROLL for (int j = 0; j < NS; j++) a += Pinv[i][j] * (x0v[j] - xbar[j]);
                g0[i] = a; ga0[i] = ga;
                const double r0 = a - zl0[i] + zh0[i];
                e_st = dmax(e_st, fabs(r0)); s_z += zl0[i] + zh0[i];
                finite = finite && finite_all(r0) && finite_all(x0v[i]);      // (the residual, as for the other variables: an infinite multiplier must end the solve as failed)
                if (flx[i]) { cmax = dmax(cmax, sl0[i] * zl0[i]); cmin = dmin(cmin, sl0[i] * zl0[i]); }
                if (fhx[i]) { cmax = dmax(cmax, sh0[i] * zh0[i]); cmin = dmin(cmin, sh0[i] * zh0[i]); }
            }
        }
        const bool nonfinite = SG::any(on && !finite, lane);
        const double s_d = dmax(kSMax, (s_pi + s_z) / dmax(meq + nb, 1.0)) / kSMax, s_c = dmax(kSMax, s_z / dmax(nb, 1.0)) / kSMax;
        auto err = [&](double m_) { return dmax(dmax(e_st / s_d, e_c), nb > 0.0 ? dmax(cmax - m_, m_ - cmin) / s_c : 0.0); };
        if (!done) {
            if (nonfinite) { status = kStFailed; done = true; }
            else if (err(0.0) <= tol) { status = kStSolved; done = true; }
            else if (it >= max_iter) done = true;
        }
        if (__all(done ? 1 : 0)) break;
        for (;;) {      // (per segment) while (mu > tol / 10 && E_mu <= kappa_eps mu) mu = ...
            const bool dec = !done && mu > tol / 10.0 && err(mu) <= kKappaEps * mu;
            if (!__any(dec ? 1 : 0)) break;
            if (dec) mu = dmax(tol / 10.0, dmin(kKappaMu * mu, mu * sqrt(mu)));
        }
        const double tau = dmax(kTauMin, 1.0 - mu);
        // ---- barrier terms; those of x_k come from the neighbour that holds x_k --------------------------------------------------------
        double Su[NU], bu[NU], Sx[NS], bx[NS], Sxk[NS], bxk[NS], S0[NS], b0[NS];
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            const double il = flu[i] ? 1.0 / slu[i] : 0.0, ih = fhu[i] ? 1.0 / shu[i] : 0.0;
            Su[i] = zlu[i] * il + zhu[i] * ih; bu[i] = -mu * il + mu * ih;
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            const double il = flx[i] ? 1.0 / slx[i] : 0.0, ih = fhx[i] ? 1.0 / shx[i] : 0.0;
            Sx[i] = zlx[i] * il + zhx[i] * ih; bx[i] = -mu * il + mu * ih;
            Sxk[i] = SG::up1(0.0, Sx[i], k); bxk[i] = SG::up1(0.0, bx[i], k);
            const double jl = (FREE0 && flx[i]) ? 1.0 / sl0[i] : 0.0, jh = (FREE0 && fhx[i]) ? 1.0 / sh0[i] : 0.0;
            S0[i] = zl0[i] * jl + zh0[i] * jh; b0[i] = -mu * jl + mu * jh;
        }
        // ---- Riccati factorisation over the lanes, repeated with a larger shift while a stage lacks positive curvature ----------------
        double K[NU][NS], kff[NU], Pnx[NS][NS], pnx[NS], dx0[NS];
        // (lanes beyond the horizon never receive theirs: zero, not indeterminate - an indeterminate value lets the optimiser pick what suits it,
        // and with B = I known at compile time it picked something that broke the lanes inside the horizon, SEG = 64, round 3)
        MPC_UNROLL for (int i = 0; i < NU; i++) { kff[i] = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) K[i][j] = 0.0; }
        MPC_UNROLL for (int i = 0; i < NS; i++) { pnx[i] = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) Pnx[i][j] = 0.0; }
        double delta = 0.0;
        bool failed = false;
        for (;;) {
            double Pt[NS][NS], pt[NS], Pn[NS][NS], pn[NS];
            MPC_UNROLL for (int i = 0; i < NS; i++) { pt[i] = gv[i] + bx[i]; MPC_UNROLL for (int j = 0; j < NS; j++) Pt[i][j] = Hv[i][j] + (i == j ? Sx[i] + delta : 0.0); }
            bcast_sym<SEG, NS>(Pt, N - 1, lane, Pn); bcast_vec<SEG, NS>(pt, N - 1, lane, pn);
            bool bad = false;
            for (int kk = N - 1; kk >= 0; kk--) {
                double PA[NS][NS], PB[NS][NU], pc[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) EC_A(a, Pn[i][l], l, j); PA[i][j] = a; }
                    MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, Pn[i][l], l, j); PB[i][j] = a; }
                    double a = pn[i];
                    MPC_UNROLL for (int l = 0; l < NS; l++) a -= Pn[i][l] * c[l];
                    pc[i] = a;
                }
                double Quu[NU][NU], Qux[NU][NS], Qxx[NS][NS], qu[NU], qx[NS];
                MPC_UNROLL for (int i = 0; i < NU; i++) {
                    MPC_UNROLL for (int j = 0; j < NU; j++) { double a = L.R[i][j] + (i == j ? Su[i] + delta : 0.0); MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, PB[l][j], l, i); Quu[i][j] = a; }
                    MPC_UNROLL for (int j = 0; j < NS; j++) { double a = L.M[j][i]; MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, PA[l][j], l, i); Qux[i][j] = a; }
                    double a = L.lu[i] + bu[i];
                    MPC_UNROLL for (int l = 0; l < NS; l++) EC_B(a, pc[l], l, i);
                    qu[i] = a;
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    MPC_UNROLL for (int j = 0; j < NS; j++) { double a = L.Q[i][j] + (i == j ? Sxk[i] + delta : 0.0); MPC_UNROLL for (int l = 0; l < NS; l++) EC_A(a, PA[l][j], l, i); Qxx[i][j] = a; }
                    double a = L.lx[i] + bxk[i];
                    MPC_UNROLL for (int l = 0; l < NS; l++) EC_A(a, pc[l], l, i);
                    qx[i] = a;
                }
                MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (Quu[i][j] + Quu[j][i]); Quu[i][j] = a; Quu[j][i] = a; } }
                const bool ok = sym_inverse<NU>(Quu);
                double Kl[NU][NS], kl[NU], Pk[NS][NS], pk[NS];
                MPC_UNROLL for (int i = 0; i < NU; i++) {
                    MPC_UNROLL for (int j = 0; j < NS; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NU; l++) a -= Quu[i][l] * Qux[l][j]; Kl[i][j] = a; }
                    double a = 0.0;
                    MPC_UNROLL for (int l = 0; l < NU; l++) a -= Quu[i][l] * qu[l];
                    kl[i] = a;
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) {
                    MPC_UNROLL for (int j = 0; j < NS; j++) { double a = Qxx[i][j]; MPC_UNROLL for (int l = 0; l < NU; l++) a += Qux[l][i] * Kl[l][j]; Pk[i][j] = a; }
                    double a = qx[i];
                    MPC_UNROLL for (int l = 0; l < NU; l++) a += Qux[l][i] * kl[l];
                    pk[i] = a;
                }
                MPC_UNROLL for (int i = 0; i < NS; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (Pk[i][j] + Pk[j][i]); Pk[i][j] = a; Pk[j][i] = a; } }
                if (k == kk) {
                    MPC_UNROLL for (int i = 0; i < NU; i++) { kff[i] = kl[i]; MPC_UNROLL for (int j = 0; j < NS; j++) K[i][j] = Kl[i][j]; }
                    MPC_UNROLL for (int i = 0; i < NS; i++) { pnx[i] = pn[i]; MPC_UNROLL for (int j = 0; j < NS; j++) Pnx[i][j] = Pn[i][j]; }
                    if (!ok) bad = true;
                }
                bcast_sym<SEG, NS>(Pk, kk, lane, Pn); bcast_vec<SEG, NS>(pk, kk, lane, pn);
            }
            if (FREE0) {      // the initial state: value function of stage 0 + arrival cost + its own barrier
                double P0[NS][NS], p0[NS];
                MPC_UNROLL for (int i = 0; i < NS; i++) { p0[i] = pn[i] + (g0[i] - ga0[i]) + b0[i]; MPC_UNROLL for (int j = 0; j < NS; j++) P0[i][j] = Pn[i][j] + 0.5 * (Pinv[i][j] + Pinv[j][i]) + (i == j ? S0[i] : 0.0); }
                if (!sym_inverse<NS>(P0)) bad = true;
                MPC_UNROLL for (int i = 0; i < NS; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NS; j++) a -= P0[i][j] * p0[j]; dx0[i] = a; }
            } else { MPC_UNROLL for (int i = 0; i < NS; i++) dx0[i] = 0.0; }
            const bool retry = SG::any(bad, lane) && !done && !failed;      // this segment lacks curvature: a larger shift, all over again
            if (retry) {
                delta = delta == 0.0 ? dmax(kDeltaFirst, delta_last / 3.0) : delta * (delta_last == 0.0 ? 100.0 : 8.0);
                if (delta > kDeltaMax) failed = true;
            }
            if (!__any((retry && !failed) ? 1 : 0)) break;      // (segments that were fine recompute the same numbers with their own shift)
        }
        if (!done && failed) { status = kStFailed; done = true; }
        if (!done && delta > 0.0) delta_last = delta;
        // ---- Newton direction: forward over the lanes ---------------------------------------------------------------------------------
        double du[NU], dxn[NS], dx[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) { dx[i] = dx0[i]; dxn[i] = 0.0; }
        MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = 0.0;
        for (int kk = 0; kk < N; kk++) {
            double dul[NU], dxl[NS];
            MPC_UNROLL for (int i = 0; i < NU; i++) { double a = kff[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += K[i][j] * dx[j]; dul[i] = a; }
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                double a = -c[i];
                MPC_UNROLL for (int j = 0; j < NS; j++) EC_A(a, dx[j], i, j);
                MPC_UNROLL for (int j = 0; j < NU; j++) EC_B(a, dul[j], i, j);
                dxl[i] = a;
            }
            if (k == kk) { MPC_UNROLL for (int i = 0; i < NU; i++) du[i] = dul[i]; MPC_UNROLL for (int i = 0; i < NS; i++) dxn[i] = dxl[i]; }
            bcast_vec<SEG, NS>(dxl, kk, lane, dx);
        }
        double pin[NS];
        MPC_UNROLL for (int i = 0; i < NS; i++) { double a = pnx[i]; MPC_UNROLL for (int j = 0; j < NS; j++) a += Pnx[i][j] * dxn[j]; pin[i] = a; }
        // ---- multiplier steps, fraction to the boundary ------------------------------------------------------------------------------------
        double dzlu[NU], dzhu[NU], dzlx[NS], dzhx[NS], dzl0[NS], dzh0[NS];
        double apr = 1.0, adu = 1.0;
        auto ratio = [&](double a, double v, double dv) { return dv < 0.0 ? dmin(a, -tau * v / dv) : a; };
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            dzlu[i] = flu[i] ? mu / slu[i] - zlu[i] - zlu[i] / slu[i] * du[i] : 0.0;
            dzhu[i] = fhu[i] ? mu / shu[i] - zhu[i] + zhu[i] / shu[i] * du[i] : 0.0;
            if (flu[i]) { apr = ratio(apr, slu[i], du[i]); adu = ratio(adu, zlu[i], dzlu[i]); }
            if (fhu[i]) { apr = ratio(apr, shu[i], -du[i]); adu = ratio(adu, zhu[i], dzhu[i]); }
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            dzlx[i] = flx[i] ? mu / slx[i] - zlx[i] - zlx[i] / slx[i] * dxn[i] : 0.0;
            dzhx[i] = fhx[i] ? mu / shx[i] - zhx[i] + zhx[i] / shx[i] * dxn[i] : 0.0;
            if (flx[i]) { apr = ratio(apr, slx[i], dxn[i]); adu = ratio(adu, zlx[i], dzlx[i]); }
            if (fhx[i]) { apr = ratio(apr, shx[i], -dxn[i]); adu = ratio(adu, zhx[i], dzhx[i]); }
        }
        apr = SG::min(on ? apr : 1.0); adu = SG::min(on ? adu : 1.0);
        if (FREE0) {
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                dzl0[i] = flx[i] ? mu / sl0[i] - zl0[i] - zl0[i] / sl0[i] * dx0[i] : 0.0;
                dzh0[i] = fhx[i] ? mu / sh0[i] - zh0[i] + zh0[i] / sh0[i] * dx0[i] : 0.0;
                if (flx[i]) { apr = ratio(apr, sl0[i], dx0[i]); adu = ratio(adu, zl0[i], dzl0[i]); }
                if (fhx[i]) { apr = ratio(apr, sh0[i], -dx0[i]); adu = ratio(adu, zh0[i], dzh0[i]); }
            }
        }
        // ---- step; multipliers kept within kappa_Sigma of mu / slack --------------------------------------------------------------------
        auto clampz = [&](double z, double s) { return dmin(dmax(z, mu / (kKappaSigma * s)), kKappaSigma * mu / s); };
        if (!done) {      // (a finished segment keeps its iterate)
        MPC_UNROLL for (int i = 0; i < NU; i++) {
            u[i] += apr * du[i];
            zlu[i] += adu * dzlu[i]; zhu[i] += adu * dzhu[i];
            if (flu[i]) zlu[i] = clampz(zlu[i], u[i] - ulo[i]);
            if (fhu[i]) zhu[i] = clampz(zhu[i], uhi[i] - u[i]);
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            xn[i] += apr * dxn[i]; pi[i] += apr * (pin[i] - pi[i]);
            zlx[i] += adu * dzlx[i]; zhx[i] += adu * dzhx[i];
            if (flx[i]) zlx[i] = clampz(zlx[i], xn[i] - xlo[i]);
            if (fhx[i]) zhx[i] = clampz(zhx[i], xhi[i] - xn[i]);
            if (FREE0) {
                x0v[i] += apr * dx0[i];
                zl0[i] += adu * dzl0[i]; zh0[i] += adu * dzh0[i];
                if (flx[i]) zl0[i] = clampz(zl0[i], x0v[i] - xlo[i]);
                if (fhx[i]) zh0[i] = clampz(zh0[i], xhi[i] - x0v[i]);
            }
        }
        }
    }
    return status;
}

// ---- small dense helpers (wave-uniform use) ---------------------------------------------------------------------------------------------
// inverse of a general n x n matrix by Gauss-Jordan with partial pivoting; false when a pivot vanishes
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Target: min fss(xs, us, ys)  s.t.  Fx_model(xs, us, d) - xs = 0,  xs + Cd d - ys = 0,  boxes   (Target_Calc.py:20-161 with
// StateFeedback outputs).  Same outer algorithm; the Newton system is reduced to the nu inputs: ys and xs follow from the two
// (linearised) equalities, so the inertia test is the sign of the nu x nu reduced Hessian.  Wave-uniform: every lane computes it.
// v = [xs; us; ys] comes in as the first guess (MPC_code.py:696-700).
// ---------------------------------------------------------------------------------------------------------------------------------
template <class M>
__device__ __forceinline__ int target_ipm(double (&v)[M::NX + M::NU + M::NY], const double *d, const double (*Bd)[M::ND > 0 ? M::ND : 1],
                                          const double (*Cd)[M::ND > 0 ? M::ND : 1], const double *lo, const double *hi, double t, double h,
                                          const double tol, const int max_iter, int &iters)
{
    constexpr int NX = M::NX, NU = M::NU, NY = M::NY, ND = M::ND, NV = NX + NU + NY, NP = NX + NU, NPP = NP * (NP + 1) / 2;
    static_assert(NY == NX, "StateFeedback outputs");
    bool fl[NV], fh[NV];
    double zl[NV], zh[NV], lam1[NX], lam2[NY];
    int nbi = 0;
    MPC_UNROLL for (int i = 0; i < NV; i++) { fl[i] = fin(lo[i]); fh[i] = fin(hi[i]); zl[i] = fl[i] ? 1.0 : 0.0; zh[i] = fh[i] ? 1.0 : 0.0; nbi += (fl[i] ? 1 : 0) + (fh[i] ? 1 : 0); v[i] = push_in(v[i], lo[i], hi[i]); }
    MPC_UNROLL for (int i = 0; i < NX; i++) lam1[i] = 0.0;
    MPC_UNROLL for (int i = 0; i < NY; i++) lam2[i] = 0.0;
    const double nb = (double)nbi, meq = (double)(NX + NY);
    double mu = kMuInit, delta_last = 0.0;
    int status = kStMaxIter;
    iters = 0;
    for (int it = 0;; it++) {
        iters = it;
        typename M::Ctx cx;
        MPC_UNROLL for (int i = 0; i < NU; i++) { cx.u[i] = v[NX + i]; cx.us[i] = 0.0; }
        MPC_UNROLL for (int i = 0; i < ND; i++) cx.d[i] = d[i];
        MPC_UNROLL for (int i = 0; i < NX; i++) cx.xs[i] = 0.0;
        double Fx[NX], S[NX][NP], T[NX][NPP];
        rk4_sens2<typename M::Mdl>(v, cx, t, true, h, M::MX, Fx, S, T);
        double f, g[NV], Hc[NV][NV];
        M::fss(v, &f, g, Hc);
        double c1[NX], c2[NY], J1[NX][NX];
        MPC_UNROLL for (int i = 0; i < NX; i++) {
            double a = Fx[i] - v[i];
            MPC_UNROLL for (int j = 0; j < ND; j++) a += Bd[i][j] * d[j];
            c1[i] = a;
            MPC_UNROLL for (int j = 0; j < NX; j++) J1[i][j] = S[i][j] - (i == j ? 1.0 : 0.0);
        }
        MPC_UNROLL for (int i = 0; i < NY; i++) { double a = v[i] - v[NX + NU + i]; MPC_UNROLL for (int j = 0; j < ND; j++) a += Cd[i][j] * d[j]; c2[i] = a; }
        // Hessian of the Lagrangian: cost + lam1' Fx
        double H[NV][NV];
        MPC_UNROLL for (int i = 0; i < NV; i++) { MPC_UNROLL for (int j = 0; j < NV; j++) H[i][j] = Hc[i][j]; }
        MPC_UNROLL for (int a = 0; a < NP; a++) { MPC_UNROLL for (int b = 0; b < NP; b++) { double s = 0.0; MPC_UNROLL for (int i = 0; i < NX; i++) s += lam1[i] * T[i][pair_idx<NP>(a, b)]; H[a][b] += s; } }
        double sl[NV], sh[NV], stat[NV];
        double e_st = 0.0, e_c = 0.0, s_l = 0.0, s_z = 0.0, cmax = -INFINITY, cmin = INFINITY;
        bool finite = true;
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            sl[i] = fl[i] ? v[i] - lo[i] : 1.0; sh[i] = fh[i] ? hi[i] - v[i] : 1.0;
            double r = g[i] - zl[i] + zh[i];
            if (i < NX) { MPC_UNROLL for (int j = 0; j < NX; j++) r += J1[j][i < NX ? i : 0] * lam1[j]; r += lam2[i < NY ? i : 0]; }      // C = I
            else if (i < NP) { MPC_UNROLL for (int j = 0; j < NX; j++) r += S[j][i < NP ? i : 0] * lam1[j]; }
            else r -= lam2[i >= NP ? i - NP : 0];
            stat[i] = r;
            e_st = dmax(e_st, fabs(r)); s_z += zl[i] + zh[i];
            finite = finite && finite_all(r) && finite_all(v[i]);
            if (fl[i]) { cmax = dmax(cmax, sl[i] * zl[i]); cmin = dmin(cmin, sl[i] * zl[i]); }
            if (fh[i]) { cmax = dmax(cmax, sh[i] * zh[i]); cmin = dmin(cmin, sh[i] * zh[i]); }
        }
        MPC_UNROLL for (int i = 0; i < NX; i++) { e_c = dmax(e_c, fabs(c1[i])); s_l += fabs(lam1[i]); finite = finite && finite_all(c1[i]); }
        MPC_UNROLL for (int i = 0; i < NY; i++) { e_c = dmax(e_c, fabs(c2[i])); s_l += fabs(lam2[i]); }
        if (!finite) { status = kStFailed; break; }
        const double s_d = dmax(kSMax, (s_l + s_z) / dmax(meq + nb, 1.0)) / kSMax, s_c = dmax(kSMax, s_z / dmax(nb, 1.0)) / kSMax;
        auto err = [&](double m_) { return dmax(dmax(e_st / s_d, e_c), nb > 0.0 ? dmax(cmax - m_, m_ - cmin) / s_c : 0.0); };
        if (err(0.0) <= tol) { status = kStSolved; break; }
        if (it >= max_iter) break;
        while (mu > tol / 10.0 && err(mu) <= kKappaEps * mu) mu = dmax(tol / 10.0, dmin(kKappaMu * mu, mu * sqrt(mu)));
        const double tau = dmax(kTauMin, 1.0 - mu);
        double Sg[NV], gt[NV];
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            const double il = fl[i] ? 1.0 / sl[i] : 0.0, ih = fh[i] ? 1.0 / sh[i] : 0.0;
            Sg[i] = zl[i] * il + zh[i] * ih; gt[i] = g[i] - mu * il + mu * ih;
        }
        // null-space basis Z = [Zx; I; Zx] and particular step sp = [spx; 0; spx + c2] of the linearised equalities
        double W[NX][NX];
        if (!gj_inverse<NX>(J1, W)) { status = kStFailed; break; }
        double Z[NV][NU], sp[NV];
        MPC_UNROLL for (int i = 0; i < NX; i++) {
            MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NX; l++) a -= W[i][l] * S[l][NX + j]; Z[i][j] = a; Z[NP + i][j] = a; }
            double a = 0.0;
            MPC_UNROLL for (int l = 0; l < NX; l++) a -= W[i][l] * c1[l];
            sp[i] = a; sp[NP + i] = a + c2[i];
        }
        MPC_UNROLL for (int i = 0; i < NU; i++) { sp[NX + i] = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) Z[NX + i][j] = (i == j) ? 1.0 : 0.0; }
        double dv[NV], Hd[NV];
        double delta = 0.0;
        bool failed = false;
        for (;;) {
            double HZ[NV][NU], Hr[NU][NU], rr[NU], hs[NV];
            MPC_UNROLL for (int i = 0; i < NV; i++) {
                MPC_UNROLL for (int j = 0; j < NU; j++) { double a = (Sg[i] + delta) * Z[i][j]; MPC_UNROLL for (int l = 0; l < NV; l++) a += H[i][l] * Z[l][j]; HZ[i][j] = a; }
                double a = (Sg[i] + delta) * sp[i] + gt[i];
                MPC_UNROLL for (int l = 0; l < NV; l++) a += H[i][l] * sp[l];
                hs[i] = a;
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) {
                MPC_UNROLL for (int j = 0; j < NU; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NV; l++) a += Z[l][i] * HZ[l][j]; Hr[i][j] = a; }
                double a = 0.0;
                MPC_UNROLL for (int l = 0; l < NV; l++) a += Z[l][i] * hs[l];
                rr[i] = a;
            }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (Hr[i][j] + Hr[j][i]); Hr[i][j] = a; Hr[j][i] = a; } }
            if (sym_inverse<NU>(Hr)) {
                double dus[NU];
                MPC_UNROLL for (int i = 0; i < NU; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NU; j++) a -= Hr[i][j] * rr[j]; dus[i] = a; }
                MPC_UNROLL for (int i = 0; i < NV; i++) { double a = sp[i]; MPC_UNROLL for (int j = 0; j < NU; j++) a += Z[i][j] * dus[j]; dv[i] = a; }
                MPC_UNROLL for (int i = 0; i < NV; i++) { double a = (Sg[i] + delta) * dv[i] + gt[i]; MPC_UNROLL for (int l = 0; l < NV; l++) a += H[i][l] * dv[l]; Hd[i] = a; }
                break;
            }
            delta = delta == 0.0 ? dmax(kDeltaFirst, delta_last / 3.0) : delta * (delta_last == 0.0 ? 100.0 : 8.0);
            if (delta > kDeltaMax) { failed = true; break; }
        }
        if (failed) { status = kStFailed; break; }
        if (delta > 0.0) delta_last = delta;
        // new multipliers from the stationarity rows of ys and xs
        double l1n[NX], l2n[NY];
        MPC_UNROLL for (int i = 0; i < NY; i++) l2n[i] = Hd[NP + i];
        MPC_UNROLL for (int i = 0; i < NX; i++) { double a = 0.0; MPC_UNROLL for (int j = 0; j < NX; j++) a -= W[j][i] * (Hd[j] + l2n[j]); l1n[i] = a; }
        double apr = 1.0, adu = 1.0, dzl[NV], dzh[NV];
        auto ratio = [&](double a, double vv, double dvv) { return dvv < 0.0 ? dmin(a, -tau * vv / dvv) : a; };
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            dzl[i] = fl[i] ? mu / sl[i] - zl[i] - zl[i] / sl[i] * dv[i] : 0.0;
            dzh[i] = fh[i] ? mu / sh[i] - zh[i] + zh[i] / sh[i] * dv[i] : 0.0;
            if (fl[i]) { apr = ratio(apr, sl[i], dv[i]); adu = ratio(adu, zl[i], dzl[i]); }
            if (fh[i]) { apr = ratio(apr, sh[i], -dv[i]); adu = ratio(adu, zh[i], dzh[i]); }
        }
        auto clampz = [&](double z, double s) { return dmin(dmax(z, mu / (kKappaSigma * s)), kKappaSigma * mu / s); };
        MPC_UNROLL for (int i = 0; i < NV; i++) {
            v[i] += apr * dv[i];
            zl[i] += adu * dzl[i]; zh[i] += adu * dzh[i];
            if (fl[i]) zl[i] = clampz(zl[i], v[i] - lo[i]);
            if (fh[i]) zh[i] = clampz(zh[i], hi[i] - v[i]);
        }
        MPC_UNROLL for (int i = 0; i < NX; i++) lam1[i] += apr * (l1n[i] - lam1[i]);
        MPC_UNROLL for (int i = 0; i < NY; i++) lam2[i] += apr * (l2n[i] - lam2[i]);
    }
    return status;
}

}  // namespace enm
