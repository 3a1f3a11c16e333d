// libmpc_amd.so - hand-written HIP (gfx950) implementation of include/mpc_amd.h.
//
// Host side: turns the reference-level problem description into the stage form / null-space form the
// kernels use, owns all device memory, launches on one stream per handle, times launches with HIP events.
// Device side: mpc_device.hpp.  No PyTorch, no CUDA-compat layer, no CPU fallback: every entry point
// fails loudly when the device path is unavailable.
#include "../../include/mpc_amd.h"
#include "mpc_device.hpp"
#include "mpc_tp.hpp"
#include "mpc_wave.hpp"
#include "mpc_soft.hpp"
#ifdef MPC_NL_PLANT_HEADER
// A non-linear plant for the fused closed loop (Ex_LMPC_nlplant.py, Ex_LMPCxp_nlplant.py: linear controller, User_fxp_Cont as the
// simulated process): struct NlPlant { NXP, NU, MX; __device__ static void f(x, u, t, xdot); } generated from the traced Ex-file
// function (mpc-code_amd/nlcodegen.py:emit_plant_header); a library built with it carries exactly one dimension set.
#include MPC_NL_PLANT_HEADER
#endif

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace mpc;

// ---------------------------------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(x)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) return fail(-10, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#ifndef MPC_PART2
extern "C" const char *mpc_last_error(void) { return g_err; }
#endif

// ---------------------------------------------------------------------------------------------------
// compiled dimension sets: NX, NU, NY, ND, NXP, DU, NG  (NG = bounded output rows that are not a multiple of one state;
// each is carried as one more stage state, see build_problem)
// ---------------------------------------------------------------------------------------------------
#ifndef MPC_DIM_LIST
#define MPC_DEFAULT_DIM_LIST 1
// The default library is built from two objects compiled side by side (capi.build_library: this file with -DMPC_HAVE_PART2 and with -DMPC_PART2):
// the first carries the C-ABI and the BASELINE dimension sets, the second the kernels of the other sets behind mpc_part2_launchers().  One object
// (neither macro: tools, diagnostic builds) carries all of them.
#define MPC_DIM_LIST_A(X) \
    X(3, 2, 3, 3, 3, 0, 0) /* Ex_LMPC_CSTR */ \
    X(4, 2, 2, 2, 4, 1, 0) /* Ex_LMPC_WB (cost on Delta-u: stage state 6) */
#define MPC_DIM_LIST_B(X) \
    X(3, 2, 2, 2, 3, 1, 0) /* Ex_LMPC_nlplant (linear controller, Delta-u cost, non-linear plant on the host) */ \
    X(4, 2, 2, 2, 3, 1, 1) /* Ex_LMPCxp_nlplant (model state 4, plant state 3, one general output row: stage state 7) */ \
    X(2, 1, 1, 1, 2, 0, 0) /* double integrator (tests: LQR known answer) */ \
    X(2, 1, 1, 1, 2, 1, 0) \
    X(2, 1, 1, 1, 2, 0, 1) /* double integrator with a bound on x0 + x1 (tests: general output row) */
#if defined(MPC_PART2)
#define MPC_DIM_LIST(X) MPC_DIM_LIST_B(X)
#elif defined(MPC_HAVE_PART2)
#define MPC_DIM_LIST(X) MPC_DIM_LIST_A(X)
#else
#define MPC_DIM_LIST(X) MPC_DIM_LIST_A(X) MPC_DIM_LIST_B(X)
#endif
#endif

// ---------------------------------------------------------------------------------------------------
// kernels.  Device arrays are structure-of-arrays [dim][Bs], instance index fastest.
// ---------------------------------------------------------------------------------------------------
struct OcpArgs {
    const double *xhat, *xs, *us, *dhat, *u_prev;   // in
    double *u_out, *xnext_out, *res;                // out ([nu][Bs], [nx][Bs], [3][Bs])
    int32_t *status, *iters;
    double *ws;                                     // workspace rows
    int B; size_t Bs;
};

template <int NX, int NU, int NY, int ND, bool DU, int NG, int NC, bool MASKED>
__global__ __launch_bounds__(64) void ocp_kernel(const DevProblem *__restrict__ Pp, OcpArgs a)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const DevProblem &P = *Pp;
    double xhat[NX], xs[NX], us[NU], up[NU], dh[ND > 0 ? ND : 1];
    MPC_UNROLL for (int i = 0; i < NX; i++) { xhat[i] = a.xhat[i * a.Bs + b]; xs[i] = a.xs[i * a.Bs + b]; }
    MPC_UNROLL for (int i = 0; i < NU; i++) { us[i] = a.us[i * a.Bs + b]; up[i] = a.u_prev[i * a.Bs + b]; }
    MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = a.dhat[i * a.Bs + b];
    OcpInst<NS, NU> q;
    build_inst<NX, NU, NY, ND, DU, NG>(P, xhat, xs, us, dh, up, q);
    StageConst<NS, NU> C;
    load_stage_const<NS, NU, DU>(P, C);
    constexpr int SL = BlkLayout<NS, NU, NC>::SLOTS;
    Ws ws{(gv2d *)((v2d *)a.ws + ((size_t)blockIdx.x * (P.N + 2) + 1) * SL * 64), P.N, SL, (int)threadIdx.x};
    double u0[NU], z1[NS], res[3];
    int it;
    int st, it_sum = 0;
    for (int pass = 0;; pass++) {      // (terminal equality: up to two more passes with the terminal reference aimed off, mpc_device.hpp:term_aim)
        st = rpdip_lane<NS, NU, DU, NC, MASKED>(P, C, q, ws, P.max_iter, false, 0.0, u0, z1, res, it);
        it_sum += it;
        if (!(P.term_cons && pass < 2 && st != kInfeasible && term_aim<NS, NU, NC, NX>(P, ws, q))) break;
    }
    it = it_sum;
    if (P.term_cons && st != kInfeasible && term_missed<NS, NU, NC, NX>(P, ws, q)) st = kInfeasible;
    a.status[b] = st; a.iters[b] = it;
    MPC_UNROLL for (int i = 0; i < 3; i++) a.res[i * a.Bs + b] = res[i];
    if (st != kInfeasible) {
        MPC_UNROLL for (int i = 0; i < NU; i++) a.u_out[i * a.Bs + b] = (DU && P.in_is_du) ? z1[DU ? NX + i : 0] : u0[i];      // input v = u - u_prev: u is the u_prev part of z_1
        MPC_UNROLL for (int i = 0; i < NX; i++) a.xnext_out[i * a.Bs + b] = z1[i];
    }
}

// the soft problem of one instance: stage data of the hard problem (build_inst with y_bounded = 0: the stage boxes are the state bounds alone) + the output rows and their weight
template <int NX, int NU, int NY, int ND, bool DU, int NS>
__device__ __forceinline__ void soft_prob_fill(const DevProblem &P, const OcpInst<NS, NU> &q, const double *dh, SoftProb<NS, NU, NY> &S)
{
    S.N = P.N; S.max_iter = P.max_iter;
    for (int i = 0; i < NS; i++) {
        for (int j = 0; j < NS; j++) { S.A[i][j] = P.A[i][j]; S.Q[i][j] = P.Q[i][j]; S.Pf[i][j] = P.Pf[i][j]; }
        for (int j = 0; j < NU; j++) { S.B[i][j] = P.B[i][j]; S.M[i][j] = DU ? P.M[i][j] : 0.0; }
        S.c[i] = q.c[i]; S.z0[i] = q.z0[i]; S.zr[i] = q.zr[i]; S.zrN[i] = q.zrN[i];
        S.zlo[i] = q.zlo_m[i]; S.zhi[i] = q.zhi_m[i]; S.zlo_e[i] = P.zlo_e[i]; S.zhi_e[i] = P.zhi_e[i];
    }
    for (int i = 0; i < NU; i++) { for (int j = 0; j < NU; j++) S.R[i][j] = P.R[i][j]; S.ur[i] = q.ur[i]; S.us[i] = q.us[i]; S.ulo[i] = P.ulo[i]; S.uhi[i] = P.uhi[i]; }
    for (int i = 0; i < NY; i++) {
        double e = P.fyc[i];
        for (int j = 0; j < ND; j++) e += P.Cd[i][j] * dh[j];
        S.cy[i] = e; S.ymin[i] = P.ymin[i]; S.ymax[i] = P.ymax[i];
        for (int j = 0; j < NS; j++) S.Cy[i][j] = j < NX ? P.Cm[i][j < NX ? j : 0] : 0.0;      // outputs read the model states (not the u_prev part of the stage state)
    }
    for (int i = 0; i < 2 * NY; i++) for (int j = 0; j < 2 * NY; j++) S.Ws[i][j] = P.Ws[i][j];
}

// The same call for a problem with SOFT output constraints (`slacks = True`, Control_Calc.py:39-40,186-192,228-239): one slack vector shared by all stages - the
// arrowhead solver of mpc_soft.hpp, one instance per lane, workspace [wave][block][field][64 lanes] in HBM.  sl_out [2 NY][Bs]: the optimal slacks (MPC_code.py:800).
struct OcpSoftArgs { OcpArgs o; double *sl_out; };

template <int NX, int NU, int NY, int ND, bool DU, int NG>
__global__ __launch_bounds__(64) void ocp_kernel_soft(const DevProblem *__restrict__ Pp, OcpSoftArgs w)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG;
    const OcpArgs &a = w.o;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const DevProblem &P = *Pp;
    double xhat[NX], xs[NX], us[NU], up[NU], dh[ND > 0 ? ND : 1];
    MPC_UNROLL for (int i = 0; i < NX; i++) { xhat[i] = a.xhat[i * a.Bs + b]; xs[i] = a.xs[i * a.Bs + b]; }
    MPC_UNROLL for (int i = 0; i < NU; i++) { us[i] = a.us[i * a.Bs + b]; up[i] = a.u_prev[i * a.Bs + b]; }
    MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = a.dhat[i * a.Bs + b];
    OcpInst<NS, NU> q;
    build_inst<NX, NU, NY, ND, DU, NG>(P, xhat, xs, us, dh, up, q);      // (y_bounded = 0 in a soft problem: the stage boxes are the state bounds alone)
    SoftProb<NS, NU, NY> S;
    soft_prob_fill<NX, NU, NY, ND, DU, NS>(P, q, dh, S);
    using LY = SoftLayout<NS, NU, NY>;
    double *const ws = a.ws + (size_t)blockIdx.x * LY::FIELDS * P.N * 64 + threadIdx.x;
    double u0[NU], z1[NS], sl[2 * NY], res[3];
    int it;
    const int st = soft_solve<NS, NU, NY, 64>(S, ws, u0, z1, sl, res, it);
    a.status[b] = st; a.iters[b] = it;
    MPC_UNROLL for (int i = 0; i < 3; i++) a.res[i * a.Bs + b] = res[i];
    for (int i = 0; i < 2 * NY; i++) w.sl_out[i * a.Bs + b] = sl[i];
    if (st != kInfeasible) {
        MPC_UNROLL for (int i = 0; i < NU; i++) a.u_out[i * a.Bs + b] = (DU && P.in_is_du) ? z1[DU ? NX + i : 0] : u0[i];
        MPC_UNROLL for (int i = 0; i < NX; i++) a.xnext_out[i * a.Bs + b] = z1[i];
    }
}

// The same call with time-varying model parameters over the horizon (def_px / def_py, MPC_code.py:492-497, Control_Calc.py:43-57):
// x_{k+1} = A x_k + B u_k + Bd d + px_k, y_k = C x_k + Cd d + py_k.  The stage form keeps its matrices; the affine term and the
// boxes that stand for the output rows become per block, which is what the time-varying lane solver takes (mpc_device.hpp:
// rpdip_lane<.., LTV>: slab [block][A | B | c | lo | hi][64 lanes]).  One instantiation per dimension set (all bounds maskable).
struct OcpPxyArgs { OcpArgs o; const double *px, *py; double *lin; };      // px [N][NX][Bs], py [N][NY][Bs]; either may be null

template <int NX, int NU, int NY, int ND, bool DU, int NG>
__global__ __launch_bounds__(64) void ocp_kernel_pxy(const DevProblem *__restrict__ Pp, OcpPxyArgs w)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG, NB = NX + (DU ? NU : 0), NC = NS + NU, NLTV = NS * (NS + NU + 1), NLIN = NLTV + 2 * NS;
    const OcpArgs &a = w.o;
    const int lane = threadIdx.x, b = blockIdx.x * 64 + lane;
    if (b >= a.B) return;
    const DevProblem &P = *Pp;
    const int N = P.N;
    double xhat[NX], xs[NX], us[NU], up[NU], dh[ND > 0 ? ND : 1];
    MPC_UNROLL for (int i = 0; i < NX; i++) { xhat[i] = a.xhat[i * a.Bs + b]; xs[i] = a.xs[i * a.Bs + b]; }
    MPC_UNROLL for (int i = 0; i < NU; i++) { us[i] = a.us[i * a.Bs + b]; up[i] = a.u_prev[i * a.Bs + b]; }
    MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = a.dhat[i * a.Bs + b];
    OcpInst<NS, NU> q;
    build_inst<NX, NU, NY, ND, DU, NG>(P, xhat, xs, us, dh, up, q);
    StageConst<NS, NU> C;
    load_stage_const<NS, NU, DU>(P, C);
    double e0[NY];      // output offsets without py
    MPC_UNROLL for (int i = 0; i < NY; i++) { double e = P.fyc[i]; MPC_UNROLL for (int j = 0; j < ND; j++) e += P.Cd[i][j] * dh[j]; e0[i] = e; }
    if (P.y_bounded && w.py) {      // the stage-0 row is a test of the given x_0 (Control_Calc.py:128-151), with py_0
        q.ok0 = true;
        MPC_UNROLL for (int i = 0; i < NY; i++) {
            double y0 = e0[i] + w.py[(size_t)i * a.Bs + b];
            MPC_UNROLL for (int j = 0; j < NX; j++) y0 += P.Cm[i][j] * xhat[j];
            const double rl = kBoundRelax * dmax(1.0, fabs(P.ymin[i])), rh = kBoundRelax * dmax(1.0, fabs(P.ymax[i]));
            if (!(y0 >= P.ymin[i] - rl) || !(y0 <= P.ymax[i] + rh)) q.ok0 = false;
        }
    }
    double *const lin = w.lin + (size_t)blockIdx.x * N * NLIN * 64 + lane;
    for (int k = 0; k < N; k++) {
        double *lb = lin + (size_t)k * NLIN * 64;
        double pxk[NX];
        MPC_UNROLL for (int i = 0; i < NX; i++) pxk[i] = w.px ? w.px[((size_t)k * NX + i) * a.Bs + b] : 0.0;
        MPC_UNROLL for (int i = 0; i < NS; i++) {
            MPC_UNROLL for (int j = 0; j < NS; j++) lb[(i * NS + j) * 64] = C.A[i][j];
            MPC_UNROLL for (int j = 0; j < NU; j++) lb[(NS * NS + i * NU + j) * 64] = C.B[i][j];
            double c = q.c[i];
            if (i < NX) c += pxk[i < NX ? i : 0];
            if (i >= NB) { const int r = P.yg_row[i >= NB ? i - NB : 0]; MPC_UNROLL for (int j = 0; j < NX; j++) c += P.Cm[r][j] * pxk[j]; }      // w+ = C_i x+
            lb[(NS * NS + NS * NU + i) * 64] = c;
        }
        // bounds of z_{k+1}: the constant boxes, cut by the output rows of stage k + 1 (none at the terminal state)
        double lo[NS], hi[NS];
        const bool end = k == N - 1;
        MPC_UNROLL for (int i = 0; i < NS; i++) { lo[i] = end ? P.zlo_e[i] : P.zlo_m[i]; hi[i] = end ? P.zhi_e[i] : P.zhi_m[i]; }
        if (P.y_bounded && !end) {
            MPC_UNROLL for (int i = 0; i < NY; i++) {
                const double e = e0[i] + (w.py ? w.py[((size_t)(k + 1) * NY + i) * a.Bs + b] : 0.0);
                const double sc = P.ymap_scale[i];
                const double aa = (P.ymin[i] - e) / sc, bb = (P.ymax[i] - e) / sc;
                const double l = sc > 0 ? aa : bb, h = sc > 0 ? bb : aa;
                const int idx = P.ymap_idx[i];
                MPC_UNROLL for (int j = 0; j < NS; j++)
                    if (j == idx) { lo[j] = dmax(lo[j], l); hi[j] = dmin(hi[j], h); }
            }
        }
        MPC_UNROLL for (int i = 0; i < NS; i++) { lb[(NLTV + i) * 64] = lo[i]; lb[(NLTV + NS + i) * 64] = hi[i]; }
    }
    constexpr int SL = BlkLayout<NS, NU, NC>::SLOTS;
    Ws ws{(gv2d *)((v2d *)a.ws + ((size_t)blockIdx.x * (N + 2) + 1) * SL * 64), N, SL, lane};
    double u0[NU], z1[NS], res[3];
    int it;
    int st, it_sum = 0;
    for (int pass = 0;; pass++) {
        st = rpdip_lane<NS, NU, DU, NC, true, true>(P, C, q, ws, P.max_iter, false, 0.0, u0, z1, res, it, lin, 1, NLIN, NLTV);
        it_sum += it;
        if (!(P.term_cons && pass < 2 && st != kInfeasible && term_aim<NS, NU, NC, NX>(P, ws, q))) break;
    }
    it = it_sum;
    if (P.term_cons && st != kInfeasible && term_missed<NS, NU, NC, NX>(P, ws, q)) st = kInfeasible;
    a.status[b] = st; a.iters[b] = it;
    MPC_UNROLL for (int i = 0; i < 3; i++) a.res[i * a.Bs + b] = res[i];
    if (st != kInfeasible) {
        MPC_UNROLL for (int i = 0; i < NU; i++) a.u_out[i * a.Bs + b] = (DU && P.in_is_du) ? z1[DU ? NX + i : 0] : u0[i];
        MPC_UNROLL for (int i = 0; i < NX; i++) a.xnext_out[i * a.Bs + b] = z1[i];
    }
}

struct TargetArgs {
    const double *usp, *ysp, *dhat, *us_prev, *px0, *py0;      // px0 / py0 [dim][Bs]: mpc_set_model_offsets, or null
    double *xs, *us, *ys;
    int32_t *status, *iters;
    int B; size_t Bs;
};

template <int NX, int NU, int NY, int ND>
__global__ __launch_bounds__(64) void target_kernel(const DevProblem *__restrict__ Pp, TargetArgs a)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const DevProblem &P = *Pp;
    double usp[NU], ysp[NY], dh[ND > 0 ? ND : 1], usprev[NU], xs[NX], us[NU], ys[NY];
    MPC_UNROLL for (int i = 0; i < NU; i++) { usp[i] = a.usp[i * a.Bs + b]; usprev[i] = a.us_prev[i * a.Bs + b]; }
    MPC_UNROLL for (int i = 0; i < NY; i++) ysp[i] = a.ysp[i * a.Bs + b];
    MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = a.dhat[i * a.Bs + b];
    int it;
    double px0[NX], py0[NY];
    if (a.px0) { MPC_UNROLL for (int i = 0; i < NX; i++) px0[i] = a.px0[i * a.Bs + b]; }
    if (a.py0) { MPC_UNROLL for (int i = 0; i < NY; i++) py0[i] = a.py0[i * a.Bs + b]; }
    const int st = target_lane<NX, NU, NY, ND>(P, usp, ysp, dh, usprev, xs, us, ys, it, nullptr, 0, nullptr, a.px0 ? px0 : nullptr, a.py0 ? py0 : nullptr);
    a.status[b] = st; a.iters[b] = it;
    MPC_UNROLL for (int i = 0; i < NX; i++) a.xs[i * a.Bs + b] = xs[i];
    MPC_UNROLL for (int i = 0; i < NU; i++) a.us[i * a.Bs + b] = us[i];
    MPC_UNROLL for (int i = 0; i < NY; i++) a.ys[i * a.Bs + b] = ys[i];
}

struct KfArgs {
    const double *y, *py0; double *xi, *Pk;
    int B; size_t Bs;
};

template <int NX, int NY, int ND>
__global__ __launch_bounds__(64) void kf_kernel(const DevProblem *__restrict__ Pp, KfArgs a)
{
    constexpr int NE = NX + ND;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const DevProblem &P = *Pp;
    double xi[NE], innov[NY];
    MPC_UNROLL for (int i = 0; i < NE; i++) xi[i] = a.xi[i * a.Bs + b];
    MPC_UNROLL for (int i = 0; i < NY; i++) {       // yhat = Fy_model(xhat, dhat), MPC_code.py:524
        double yh = P.fyc[i] + (a.py0 ? a.py0[i * a.Bs + b] : 0.0);
        MPC_UNROLL for (int j = 0; j < NE; j++) yh += P.Ca[i][j] * xi[j];
        innov[i] = a.y[i * a.Bs + b] - yh;
    }
    if (P.estimator == MPC_EST_KALMAN) {
        double Pk[NE][NE];
        MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) Pk[i][j] = a.Pk[(i * NE + j) * a.Bs + b]; }
        kalman_lane<NE, NY>(P, xi, Pk, innov);
        MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) a.Pk[(i * NE + j) * a.Bs + b] = Pk[i][j]; }
    } else if (P.estimator == MPC_EST_FIXED_GAIN) {
        MPC_UNROLL for (int i = 0; i < NE; i++) { double s = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) s += P.Kfix[i][l] * innov[l]; xi[i] += s; }
    }
    MPC_UNROLL for (int i = 0; i < NE; i++) a.xi[i * a.Bs + b] = xi[i];
}

// The closed loop, MPC_code.py:485-827: `nsteps` consecutive steps per launch, state read from / written
// back to HBM once per launch.
struct LoopArgs {
    double *x, *xhat, *dhat, *Pk, *u, *xs, *us;      // state [dim][Bs]
    const double *ysp, *usp, *pxp, *pyp;             // schedules [step][dim], already offset to k0
    double *U, *XHAT, *XS, *US, *YS, *XP, *DHAT;     // logs [step][dim][Bs] offset to k0, or nullptr
    int32_t *st_dyn, *st_ss, *it_dyn, *it_ss;        // logs [step][Bs] offset to k0, or nullptr
    int32_t *ws_valid;                               // [Bs] 1 if the workspace holds a solved OCP of the previous step
    int32_t *kf_valid;                               // [Bs] 1 if Kg / Pn hold the filter gain of this step and the prior after it
    double *Kg, *Pn;                                 // [ne*ny][Bs], [ne*ne][Bs] (horizon-parallel kernel: computed one step ahead)
    double *tw; int32_t *tw_valid;                   // warm start of the target solve: [2*nu + 3*(nx+nu+ny)][Bs], [Bs]
    double *ws;
    int B, nsteps; size_t Bs;
    int N;                                           // horizon (host side: sizes the dynamic LDS of the wave-autonomous kernel)
    double t0, h;                                    // time of the launch's first step, sampling interval (a user plant integrates in time)
    int wv_ni;                                       // wave-autonomous kernel: instances per wave (0 = by batch size; option "wave_instances")
    const double *px_h, *py_h;                       // def_px / def_py over the horizon, [step][N][nx] / [step][N][ny] offset to k0 (loop_kernel_pxy), or nullptr
    double *lin;                                     // loop_kernel_pxy: slab of per-block stage data
    double *SL, *soft_ws, *sl_keep;                  // loop_kernel_soft: log of the slacks [step][2 ny][Bs] offset to k0 (or nullptr), the arrowhead solver's workspace, the last accepted slacks [2 ny][Bs]
};

// x_p(t + h) (MPC_code.py:813-816).  Linear plant: Ap x + Bp u + pxp (Utilities.py:45-49).  User plant: MX classical RK4 steps of
// dx/dt = f(x, u, t) over h with the inputs held and time carried along, then + pxp (Utilities.py:58-82, casadi.simpleRK; LinPar).
template <int NXP, int NU, class PT>
__device__ __forceinline__ void plant_next(const PT &P, const double (&x)[NXP], const double (&u)[NU], const double *pxp_k, double t, double h, double (&xn)[NXP])
{
#ifdef MPC_NL_PLANT_HEADER
    static_assert(NlPlant::NXP == NXP && NlPlant::NU == NU, "the plant header belongs to another dimension set");
    (void)P;
    const double dt = h / NlPlant::MX;
    double xx[NXP];
    MPC_UNROLL for (int i = 0; i < NXP; i++) xx[i] = x[i];
    for (int s = 0; s < NlPlant::MX; s++) {
        const double ts = t + s * dt;
        double k1[NXP], k2[NXP], k3[NXP], k4[NXP], xa[NXP];
        NlPlant::f(xx, u, ts, k1);
        MPC_UNROLL for (int i = 0; i < NXP; i++) xa[i] = xx[i] + 0.5 * dt * k1[i];
        NlPlant::f(xa, u, ts + 0.5 * dt, k2);
        MPC_UNROLL for (int i = 0; i < NXP; i++) xa[i] = xx[i] + 0.5 * dt * k2[i];
        NlPlant::f(xa, u, ts + 0.5 * dt, k3);
        MPC_UNROLL for (int i = 0; i < NXP; i++) xa[i] = xx[i] + dt * k3[i];
        NlPlant::f(xa, u, ts + dt, k4);
        MPC_UNROLL for (int i = 0; i < NXP; i++) xx[i] += dt / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    MPC_UNROLL for (int i = 0; i < NXP; i++) xn[i] = xx[i] + pxp_k[i];
#else
    (void)t; (void)h;
    MPC_UNROLL for (int i = 0; i < NXP; i++) {
        double v = pxp_k[i];
        MPC_UNROLL for (int j = 0; j < NXP; j++) v += P.Ap[i][j] * x[j];
        MPC_UNROLL for (int j = 0; j < NU; j++) v += P.Bp[i][j] * u[j];
        xn[i] = v;
    }
#endif
}

template <int NX, int NU, int NY, int ND, int NXP, bool DU, int NG, int NC, bool MASKED>
__global__ __launch_bounds__(64) void loop_kernel(const DevProblem *__restrict__ Pp, LoopArgs a)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG, NE = NX + ND;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const DevProblem &P = *Pp;
    const size_t Bs = a.Bs;
    double x[NXP], xh[NX], dh[ND > 0 ? ND : 1], u[NU], xs[NX], us[NU];
    MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = a.x[i * Bs + b];
    MPC_UNROLL for (int i = 0; i < NX; i++) { xh[i] = a.xhat[i * Bs + b]; xs[i] = a.xs[i * Bs + b]; }
    MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = a.dhat[i * Bs + b];
    MPC_UNROLL for (int i = 0; i < NU; i++) { u[i] = a.u[i * Bs + b]; us[i] = a.us[i * Bs + b]; }
    StageConst<NS, NU> C;
    load_stage_const<NS, NU, DU>(P, C);
    constexpr int SL = BlkLayout<NS, NU, NC>::SLOTS;
    Ws ws{(gv2d *)((v2d *)a.ws + ((size_t)blockIdx.x * (P.N + 2) + 1) * SL * 64), P.N, SL, (int)threadIdx.x};
    bool ws_valid = a.ws_valid[b] != 0;
    for (int k = 0; k < a.nsteps; k++) {
        // data of the previous step, for the warm-start test: prediction of xhat, dhat, target
        double xh_pred[NX], dh_prev[ND > 0 ? ND : 1], xs_prev[NX], us_prev[NU];
        MPC_UNROLL for (int i = 0; i < NX; i++) { xh_pred[i] = xh[i]; xs_prev[i] = xs[i]; }
        MPC_UNROLL for (int i = 0; i < ND; i++) dh_prev[i] = dh[i];
        MPC_UNROLL for (int i = 0; i < NU; i++) us_prev[i] = us[i];
        if (a.XP) { MPC_UNROLL for (int i = 0; i < NXP; i++) a.XP[((size_t)k * NXP + i) * Bs + b] = x[i]; }
        if (a.XHAT) { MPC_UNROLL for (int i = 0; i < NX; i++) a.XHAT[((size_t)k * NX + i) * Bs + b] = xh[i]; }
        // ---- measure and estimate (MPC_code.py:524-534, 577-668) ---------------------------------
        if (P.estimator != MPC_EST_NONE) {
            double xi[NE], innov[NY];
            MPC_UNROLL for (int i = 0; i < NX; i++) xi[i] = xh[i];
            MPC_UNROLL for (int i = 0; i < ND; i++) xi[NX + i] = dh[i];
            MPC_UNROLL for (int i = 0; i < NY; i++) {
                double yh = P.fyc[i], yy = a.pyp[k * NY + i];
                MPC_UNROLL for (int j = 0; j < NE; j++) yh += P.Ca[i][j] * xi[j];
                MPC_UNROLL for (int j = 0; j < NXP; j++) yy += P.Cp[i][j] * x[j];
                innov[i] = yy - yh;
            }
            if (P.estimator == MPC_EST_KALMAN) {
                double Pk[NE][NE];
                MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) Pk[i][j] = a.Pk[(i * NE + j) * Bs + b]; }
                kalman_lane<NE, NY>(P, xi, Pk, innov);
                MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) a.Pk[(i * NE + j) * Bs + b] = Pk[i][j]; }
            } else {
                MPC_UNROLL for (int i = 0; i < NE; i++) { double s = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) s += P.Kfix[i][l] * innov[l]; xi[i] += s; }
            }
            MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = xi[i];
            MPC_UNROLL for (int i = 0; i < ND; i++) { double d = xi[NX + i]; if (P.has_dsat) d = dmin(dmax(d, P.dmin[i]), P.dmax[i]); dh[i] = d; }
        }
        if (a.DHAT) { MPC_UNROLL for (int i = 0; i < ND; i++) a.DHAT[((size_t)k * ND + i) * Bs + b] = dh[i]; }
        // ---- target (MPC_code.py:693-718): keep the previous one when infeasible ------------------
        double usp[NU], ysp[NY], xs_n[NX], us_n[NU], ys_n[NY];
        MPC_UNROLL for (int i = 0; i < NU; i++) usp[i] = a.usp[k * NU + i];
        MPC_UNROLL for (int i = 0; i < NY; i++) ysp[i] = a.ysp[k * NY + i];
        int it_ss;
        const int st_ss = target_lane<NX, NU, NY, ND>(P, usp, ysp, dh, us, xs_n, us_n, ys_n, it_ss, a.tw + b, Bs, a.tw_valid + b);
        if (st_ss != kInfeasible) {
            MPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = xs_n[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) us[i] = us_n[i];
        }
        if (a.XS) { MPC_UNROLL for (int i = 0; i < NX; i++) a.XS[((size_t)k * NX + i) * Bs + b] = xs[i]; }
        if (a.US) { MPC_UNROLL for (int i = 0; i < NU; i++) a.US[((size_t)k * NU + i) * Bs + b] = us[i]; }
        if (a.YS) {   // ys = Fy_model(xs, us, dhat), MPC_code.py:730
            MPC_UNROLL for (int i = 0; i < NY; i++) {
                double v = P.fyc[i];
                MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Cm[i][j] * xs[j];
                MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Cd[i][j] * dh[j];
                a.YS[((size_t)k * NY + i) * Bs + b] = v;
            }
        }
        // ---- OCP (MPC_code.py:733-805) -------------------------------------------------------------
        OcpInst<NS, NU> q;
        build_inst<NX, NU, NY, ND, DU, NG>(P, xh, xs, us, dh, u, q);
        double u0[NU], z1[NS], res[3];
        int it_dyn;
        double delta = 0.0;
        MPC_UNROLL for (int i = 0; i < NX; i++) delta = dmax(delta, dmax(fabs(xh[i] - xh_pred[i]), fabs(xs[i] - xs_prev[i])));
        MPC_UNROLL for (int i = 0; i < ND; i++) delta = dmax(delta, fabs(dh[i] - dh_prev[i]));
        MPC_UNROLL for (int i = 0; i < NU; i++) delta = dmax(delta, fabs(us[i] - us_prev[i]));
        const bool warm = ws_valid && delta <= kWsDelta && !P.no_warm;
        int st_dyn, it_sum = 0;
        for (int pass = 0;; pass++) {      // (terminal equality: further passes - cold - with the terminal reference aimed off, mpc_device.hpp:term_aim)
            st_dyn = rpdip_lane<NS, NU, DU, NC, MASKED>(P, C, q, ws, P.max_iter, warm && pass == 0, delta, u0, z1, res, it_dyn);
            it_sum += it_dyn;
            if (!(P.term_cons && pass < 2 && st_dyn != kInfeasible && term_aim<NS, NU, NC, NX>(P, ws, q))) break;
        }
        it_dyn = it_sum;
        if (P.term_cons && st_dyn != kInfeasible && term_missed<NS, NU, NC, NX>(P, ws, q)) st_dyn = kInfeasible;
        ws_valid = st_dyn == kSolved;
        if (st_dyn != kInfeasible) {
            MPC_UNROLL for (int i = 0; i < NU; i++) u[i] = (DU && P.in_is_du) ? z1[DU ? NX + i : 0] : u0[i];          // :798
            MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = z1[i];         // :799
        } else {                                                           // :804-805 hold u, propagate the model
            double xn[NX];
            MPC_UNROLL for (int i = 0; i < NX; i++) {
                double v = P.fxc[i];
                MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Am[i][j] * xh[j];
                MPC_UNROLL for (int j = 0; j < NU; j++) v += P.Bm[i][j] * u[j];
                MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Bd[i][j] * dh[j];
                xn[i] = v;
            }
            MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = xn[i];
        }
        if (a.U) { MPC_UNROLL for (int i = 0; i < NU; i++) a.U[((size_t)k * NU + i) * Bs + b] = u[i]; }
        if (a.st_dyn) { a.st_dyn[(size_t)k * Bs + b] = st_dyn; a.st_ss[(size_t)k * Bs + b] = st_ss; a.it_dyn[(size_t)k * Bs + b] = it_dyn; a.it_ss[(size_t)k * Bs + b] = it_ss; }
        // ---- plant (MPC_code.py:813-816) -----------------------------------------------------------
        {
            double xn[NXP];
            plant_next<NXP, NU>(P, x, u, a.pxp + k * NXP, a.t0 + k * a.h, a.h, xn);
            MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = xn[i];
        }
    }
    MPC_UNROLL for (int i = 0; i < NXP; i++) a.x[i * Bs + b] = x[i];
    MPC_UNROLL for (int i = 0; i < NX; i++) { a.xhat[i * Bs + b] = xh[i]; a.xs[i * Bs + b] = xs[i]; }
    MPC_UNROLL for (int i = 0; i < ND; i++) a.dhat[i * Bs + b] = dh[i];
    MPC_UNROLL for (int i = 0; i < NU; i++) { a.u[i * Bs + b] = u[i]; a.us[i * Bs + b] = us[i]; }
    a.ws_valid[b] = ws_valid ? 1 : 0;
}

// The closed loop of a problem with SOFT output constraints (slacks = True; mpc_soft.hpp): loop_kernel's step - instance per lane, the same estimator, target and
// plant code - with the arrowhead solver as its OCP, cold every step like ocp_kernel_soft: a fused run is the loop of the three C-ABI calls per step.  The optimal
// slacks of every step go to the log SL (MPC_code.py:800).  One instantiation per dimension set.
template <int NX, int NU, int NY, int ND, int NXP, bool DU, int NG>
__global__ __launch_bounds__(64) void loop_kernel_soft(const DevProblem *__restrict__ Pp, LoopArgs a)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG, NE = NX + ND;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const DevProblem &P = *Pp;
    const size_t Bs = a.Bs;
    double x[NXP], xh[NX], dh[ND > 0 ? ND : 1], u[NU], xs[NX], us[NU];
    MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = a.x[i * Bs + b];
    MPC_UNROLL for (int i = 0; i < NX; i++) { xh[i] = a.xhat[i * Bs + b]; xs[i] = a.xs[i * Bs + b]; }
    MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = a.dhat[i * Bs + b];
    MPC_UNROLL for (int i = 0; i < NU; i++) { u[i] = a.u[i * Bs + b]; us[i] = a.us[i * Bs + b]; }
    using LY = SoftLayout<NS, NU, NY>;
    double *const ws = a.soft_ws + (size_t)blockIdx.x * LY::FIELDS * P.N * 64 + threadIdx.x;
    double slk[2 * NY];      // the slacks of the last accepted OCP (MPC_code.py:800, 808-809: Sl.append(sl_k))
    for (int i = 0; i < 2 * NY; i++) slk[i] = a.sl_keep[i * Bs + b];
    for (int k = 0; k < a.nsteps; k++) {
        if (a.XP) { MPC_UNROLL for (int i = 0; i < NXP; i++) a.XP[((size_t)k * NXP + i) * Bs + b] = x[i]; }
        if (a.XHAT) { MPC_UNROLL for (int i = 0; i < NX; i++) a.XHAT[((size_t)k * NX + i) * Bs + b] = xh[i]; }
        // ---- measure and estimate (MPC_code.py:524-534, 577-668) ---------------------------------
        if (P.estimator != MPC_EST_NONE) {
            double xi[NE], innov[NY];
            MPC_UNROLL for (int i = 0; i < NX; i++) xi[i] = xh[i];
            MPC_UNROLL for (int i = 0; i < ND; i++) xi[NX + i] = dh[i];
            MPC_UNROLL for (int i = 0; i < NY; i++) {
                double yh = P.fyc[i], yy = a.pyp[k * NY + i];
                MPC_UNROLL for (int j = 0; j < NE; j++) yh += P.Ca[i][j] * xi[j];
                MPC_UNROLL for (int j = 0; j < NXP; j++) yy += P.Cp[i][j] * x[j];
                innov[i] = yy - yh;
            }
            if (P.estimator == MPC_EST_KALMAN) {
                double Pk[NE][NE];
                MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) Pk[i][j] = a.Pk[(i * NE + j) * Bs + b]; }
                kalman_lane<NE, NY>(P, xi, Pk, innov);
                MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) a.Pk[(i * NE + j) * Bs + b] = Pk[i][j]; }
            } else {
                MPC_UNROLL for (int i = 0; i < NE; i++) { double s_ = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) s_ += P.Kfix[i][l] * innov[l]; xi[i] += s_; }
            }
            MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = xi[i];
            MPC_UNROLL for (int i = 0; i < ND; i++) { double d = xi[NX + i]; if (P.has_dsat) d = dmin(dmax(d, P.dmin[i]), P.dmax[i]); dh[i] = d; }
        }
        if (a.DHAT) { MPC_UNROLL for (int i = 0; i < ND; i++) a.DHAT[((size_t)k * ND + i) * Bs + b] = dh[i]; }
        // ---- target (MPC_code.py:693-718): keep the previous one when infeasible ------------------
        double usp[NU], ysp[NY], xs_n[NX], us_n[NU], ys_n[NY];
        MPC_UNROLL for (int i = 0; i < NU; i++) usp[i] = a.usp[k * NU + i];
        MPC_UNROLL for (int i = 0; i < NY; i++) ysp[i] = a.ysp[k * NY + i];
        int it_ss;
        const int st_ss = target_lane<NX, NU, NY, ND>(P, usp, ysp, dh, us, xs_n, us_n, ys_n, it_ss, a.tw + b, Bs, a.tw_valid + b);
        if (st_ss != kInfeasible) {
            MPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = xs_n[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) us[i] = us_n[i];
        }
        if (a.XS) { MPC_UNROLL for (int i = 0; i < NX; i++) a.XS[((size_t)k * NX + i) * Bs + b] = xs[i]; }
        if (a.US) { MPC_UNROLL for (int i = 0; i < NU; i++) a.US[((size_t)k * NU + i) * Bs + b] = us[i]; }
        if (a.YS) {   // ys = Fy_model(xs, us, dhat), MPC_code.py:730
            MPC_UNROLL for (int i = 0; i < NY; i++) {
                double v = P.fyc[i];
                MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Cm[i][j] * xs[j];
                MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Cd[i][j] * dh[j];
                a.YS[((size_t)k * NY + i) * Bs + b] = v;
            }
        }
        // ---- OCP with the shared slack vector (MPC_code.py:733-805; Control_Calc.py:39-40,186-188,228-239) ---------------------------------------------
        OcpInst<NS, NU> q;
        build_inst<NX, NU, NY, ND, DU, NG>(P, xh, xs, us, dh, u, q);
        SoftProb<NS, NU, NY> S;
        soft_prob_fill<NX, NU, NY, ND, DU, NS>(P, q, dh, S);
        double u0[NU], z1[NS], sl[2 * NY], res[3];
        int it_dyn;
        const int st_dyn = soft_solve<NS, NU, NY, 64>(S, ws, u0, z1, sl, res, it_dyn);
        if (st_dyn != kInfeasible) {
            MPC_UNROLL for (int i = 0; i < NU; i++) u[i] = (DU && P.in_is_du) ? z1[DU ? NX + i : 0] : u0[i];          // :798
            MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = z1[i];         // :799
            for (int i = 0; i < 2 * NY; i++) slk[i] = sl[i];               // :800
        } else {                                                           // :804-805 hold u, propagate the model
            double xn[NX];
            MPC_UNROLL for (int i = 0; i < NX; i++) {
                double v = P.fxc[i];
                MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Am[i][j] * xh[j];
                MPC_UNROLL for (int j = 0; j < NU; j++) v += P.Bm[i][j] * u[j];
                MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Bd[i][j] * dh[j];
                xn[i] = v;
            }
            MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = xn[i];
        }
        if (a.U) { MPC_UNROLL for (int i = 0; i < NU; i++) a.U[((size_t)k * NU + i) * Bs + b] = u[i]; }
        if (a.SL) { for (int i = 0; i < 2 * NY; i++) a.SL[((size_t)k * 2 * NY + i) * Bs + b] = slk[i]; }
        if (a.st_dyn) { a.st_dyn[(size_t)k * Bs + b] = st_dyn; a.st_ss[(size_t)k * Bs + b] = st_ss; a.it_dyn[(size_t)k * Bs + b] = it_dyn; a.it_ss[(size_t)k * Bs + b] = it_ss; }
        // ---- plant (MPC_code.py:813-816) -----------------------------------------------------------
        {
            double xn[NXP];
            plant_next<NXP, NU>(P, x, u, a.pxp + k * NXP, a.t0 + k * a.h, a.h, xn);
            MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = xn[i];
        }
    }
    MPC_UNROLL for (int i = 0; i < NXP; i++) a.x[i * Bs + b] = x[i];
    MPC_UNROLL for (int i = 0; i < NX; i++) { a.xhat[i * Bs + b] = xh[i]; a.xs[i * Bs + b] = xs[i]; }
    MPC_UNROLL for (int i = 0; i < ND; i++) a.dhat[i * Bs + b] = dh[i];
    MPC_UNROLL for (int i = 0; i < NU; i++) { a.u[i * Bs + b] = u[i]; a.us[i * Bs + b] = us[i]; }
    for (int i = 0; i < 2 * NY; i++) a.sl_keep[i * Bs + b] = slk[i];
    a.ws_valid[b] = 0;
}

// The closed loop with model parameters that vary over the horizon (def_px / def_py, MPC_code.py:492-510): at step k the OCP sees
// px_i = def_px(t_k + i), py_i = def_py(t_k + i), i = 0..N-1 (the reference's indexing: time plus stage index), estimator, target, the
// stage-0 output test and the hold rule see the first of them (p_x_k, p_y_k: MPC_code.py:500-502,524,693,770-772,804).  Instance per lane
// like loop_kernel; the OCP is ocp_kernel_pxy's - per-block affine terms and boxes in a slab, rpdip_lane<.., LTV>, cold every step - so a
// fused run equals the call-by-call one (three C-ABI calls per step with mpc_set_model_offsets) bit for bit.  One instantiation per
// dimension set (all bounds maskable).
template <int NX, int NU, int NY, int ND, int NXP, bool DU, int NG>
__global__ __launch_bounds__(64) void loop_kernel_pxy(const DevProblem *__restrict__ Pp, LoopArgs a)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG, NB = NX + (DU ? NU : 0), NE = NX + ND, NC = NS + NU, NLTV = NS * (NS + NU + 1), NLIN = NLTV + 2 * NS;
    const int lane = threadIdx.x, b = blockIdx.x * 64 + lane;
    if (b >= a.B) return;
    const DevProblem &P = *Pp;
    const size_t Bs = a.Bs;
    const int N = P.N;
    double x[NXP], xh[NX], dh[ND > 0 ? ND : 1], u[NU], xs[NX], us[NU];
    MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = a.x[i * Bs + b];
    MPC_UNROLL for (int i = 0; i < NX; i++) { xh[i] = a.xhat[i * Bs + b]; xs[i] = a.xs[i * Bs + b]; }
    MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = a.dhat[i * Bs + b];
    MPC_UNROLL for (int i = 0; i < NU; i++) { u[i] = a.u[i * Bs + b]; us[i] = a.us[i * Bs + b]; }
    StageConst<NS, NU> C;
    load_stage_const<NS, NU, DU>(P, C);
    constexpr int SL = BlkLayout<NS, NU, NC>::SLOTS;
    Ws ws{(gv2d *)((v2d *)a.ws + ((size_t)blockIdx.x * (N + 2) + 1) * SL * 64), N, SL, lane};
    double *const lin = a.lin + (size_t)blockIdx.x * N * NLIN * 64 + lane;
    for (int k = 0; k < a.nsteps; k++) {
        const double *pxh = a.px_h ? a.px_h + (size_t)k * N * NX : nullptr, *pyh = a.py_h ? a.py_h + (size_t)k * N * NY : nullptr;
        double px0[NX], py0[NY];
        MPC_UNROLL for (int i = 0; i < NX; i++) px0[i] = pxh ? pxh[i] : 0.0;
        MPC_UNROLL for (int i = 0; i < NY; i++) py0[i] = pyh ? pyh[i] : 0.0;
        if (a.XP) { MPC_UNROLL for (int i = 0; i < NXP; i++) a.XP[((size_t)k * NXP + i) * Bs + b] = x[i]; }
        if (a.XHAT) { MPC_UNROLL for (int i = 0; i < NX; i++) a.XHAT[((size_t)k * NX + i) * Bs + b] = xh[i]; }
        // ---- measure and estimate (MPC_code.py:524-534, 577-668): yhat = Fy_model(xhat, dhat) + p_y_k ----
        if (P.estimator != MPC_EST_NONE) {
            double xi[NE], innov[NY];
            MPC_UNROLL for (int i = 0; i < NX; i++) xi[i] = xh[i];
            MPC_UNROLL for (int i = 0; i < ND; i++) xi[NX + i] = dh[i];
            MPC_UNROLL for (int i = 0; i < NY; i++) {
                double yh = P.fyc[i] + py0[i], yy = a.pyp[k * NY + i];
                MPC_UNROLL for (int j = 0; j < NE; j++) yh += P.Ca[i][j] * xi[j];
                MPC_UNROLL for (int j = 0; j < NXP; j++) yy += P.Cp[i][j] * x[j];
                innov[i] = (yy + py0[i]) - yh;      // the plant's output carries p_ymp = p_y_k too (MPC_code.py:505-507,534)
            }
            if (P.estimator == MPC_EST_KALMAN) {
                double Pk[NE][NE];
                MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) Pk[i][j] = a.Pk[(i * NE + j) * Bs + b]; }
                kalman_lane<NE, NY>(P, xi, Pk, innov);
                MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) a.Pk[(i * NE + j) * Bs + b] = Pk[i][j]; }
            } else {
                MPC_UNROLL for (int i = 0; i < NE; i++) { double s = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) s += P.Kfix[i][l] * innov[l]; xi[i] += s; }
            }
            MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = xi[i];
            MPC_UNROLL for (int i = 0; i < ND; i++) { double d = xi[NX + i]; if (P.has_dsat) d = dmin(dmax(d, P.dmin[i]), P.dmax[i]); dh[i] = d; }
        }
        if (a.DHAT) { MPC_UNROLL for (int i = 0; i < ND; i++) a.DHAT[((size_t)k * ND + i) * Bs + b] = dh[i]; }
        // ---- target with p_x_k, p_y_k in its equalities (MPC_code.py:693-718), cold like the per-call entry point ----
        double usp[NU], ysp[NY], xs_n[NX], us_n[NU], ys_n[NY];
        MPC_UNROLL for (int i = 0; i < NU; i++) usp[i] = a.usp[k * NU + i];
        MPC_UNROLL for (int i = 0; i < NY; i++) ysp[i] = a.ysp[k * NY + i];
        int it_ss;
        const int st_ss = target_lane<NX, NU, NY, ND>(P, usp, ysp, dh, us, xs_n, us_n, ys_n, it_ss, nullptr, 0, nullptr, pxh ? px0 : nullptr, pyh ? py0 : nullptr);
        if (st_ss != kInfeasible) {
            MPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = xs_n[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) us[i] = us_n[i];
        }
        if (a.XS) { MPC_UNROLL for (int i = 0; i < NX; i++) a.XS[((size_t)k * NX + i) * Bs + b] = xs[i]; }
        if (a.US) { MPC_UNROLL for (int i = 0; i < NU; i++) a.US[((size_t)k * NU + i) * Bs + b] = us[i]; }
        if (a.YS) {   // ys = Fy_model(xs, us, dhat, p_y_k), MPC_code.py:730
            MPC_UNROLL for (int i = 0; i < NY; i++) {
                double v = P.fyc[i] + py0[i];
                MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Cm[i][j] * xs[j];
                MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Cd[i][j] * dh[j];
                a.YS[((size_t)k * NY + i) * Bs + b] = v;
            }
        }
        // ---- OCP (MPC_code.py:733-805): per-block affine terms and boxes -> slab (as ocp_kernel_pxy) ----
        OcpInst<NS, NU> q;
        build_inst<NX, NU, NY, ND, DU, NG>(P, xh, xs, us, dh, u, q);
        double e0[NY];      // output offsets without py
        MPC_UNROLL for (int i = 0; i < NY; i++) { double e = P.fyc[i]; MPC_UNROLL for (int j = 0; j < ND; j++) e += P.Cd[i][j] * dh[j]; e0[i] = e; }
        if (P.y_bounded && pyh) {      // the stage-0 row is a test of the given x_0 (Control_Calc.py:128-151), with py_0
            q.ok0 = true;
            MPC_UNROLL for (int i = 0; i < NY; i++) {
                double y0 = e0[i] + py0[i];
                MPC_UNROLL for (int j = 0; j < NX; j++) y0 += P.Cm[i][j] * xh[j];
                const double rl = kBoundRelax * dmax(1.0, fabs(P.ymin[i])), rh = kBoundRelax * dmax(1.0, fabs(P.ymax[i]));
                if (!(y0 >= P.ymin[i] - rl) || !(y0 <= P.ymax[i] + rh)) q.ok0 = false;
            }
        }
        for (int kk = 0; kk < N; kk++) {
            double *lb = lin + (size_t)kk * NLIN * 64;
            double pxk[NX];
            MPC_UNROLL for (int i = 0; i < NX; i++) pxk[i] = pxh ? pxh[(size_t)kk * NX + i] : 0.0;
            MPC_UNROLL for (int i = 0; i < NS; i++) {
                MPC_UNROLL for (int j = 0; j < NS; j++) lb[(i * NS + j) * 64] = C.A[i][j];
                MPC_UNROLL for (int j = 0; j < NU; j++) lb[(NS * NS + i * NU + j) * 64] = C.B[i][j];
                double c = q.c[i];
                if (i < NX) c += pxk[i < NX ? i : 0];
                if (i >= NB) { const int r = P.yg_row[i >= NB ? i - NB : 0]; MPC_UNROLL for (int j = 0; j < NX; j++) c += P.Cm[r][j] * pxk[j]; }
                lb[(NS * NS + NS * NU + i) * 64] = c;
            }
            double lo[NS], hi[NS];
            const bool end = kk == N - 1;
            MPC_UNROLL for (int i = 0; i < NS; i++) { lo[i] = end ? P.zlo_e[i] : P.zlo_m[i]; hi[i] = end ? P.zhi_e[i] : P.zhi_m[i]; }
            if (P.y_bounded && !end) {
                MPC_UNROLL for (int i = 0; i < NY; i++) {
                    const double e = e0[i] + (pyh ? pyh[(size_t)(kk + 1) * NY + i] : 0.0);
                    const double sc = P.ymap_scale[i];
                    const double aa = (P.ymin[i] - e) / sc, bb = (P.ymax[i] - e) / sc;
                    const double l = sc > 0 ? aa : bb, hh = sc > 0 ? bb : aa;
                    const int idx = P.ymap_idx[i];
                    MPC_UNROLL for (int j = 0; j < NS; j++)
                        if (j == idx) { lo[j] = dmax(lo[j], l); hi[j] = dmin(hi[j], hh); }
                }
            }
            MPC_UNROLL for (int i = 0; i < NS; i++) { lb[(NLTV + i) * 64] = lo[i]; lb[(NLTV + NS + i) * 64] = hi[i]; }
        }
        double u0[NU], z1[NS], res[3];
        int it_dyn;
        int st_dyn, it_sum = 0;
        for (int pass = 0;; pass++) {
            st_dyn = rpdip_lane<NS, NU, DU, NC, true, true>(P, C, q, ws, P.max_iter, false, 0.0, u0, z1, res, it_dyn, lin, 1, NLIN, NLTV);
            it_sum += it_dyn;
            if (!(P.term_cons && pass < 2 && st_dyn != kInfeasible && term_aim<NS, NU, NC, NX>(P, ws, q))) break;
        }
        it_dyn = it_sum;
        if (P.term_cons && st_dyn != kInfeasible && term_missed<NS, NU, NC, NX>(P, ws, q)) st_dyn = kInfeasible;
        if (st_dyn != kInfeasible) {
            MPC_UNROLL for (int i = 0; i < NU; i++) u[i] = (DU && P.in_is_du) ? z1[DU ? NX + i : 0] : u0[i];          // :798
            MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = z1[i];         // :799
        } else {                                                           // :804-805 hold u, propagate the model (with p_x_k)
            double xn[NX];
            MPC_UNROLL for (int i = 0; i < NX; i++) {
                double v = P.fxc[i] + px0[i];
                MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Am[i][j] * xh[j];
                MPC_UNROLL for (int j = 0; j < NU; j++) v += P.Bm[i][j] * u[j];
                MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Bd[i][j] * dh[j];
                xn[i] = v;
            }
            MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = xn[i];
        }
        if (a.U) { MPC_UNROLL for (int i = 0; i < NU; i++) a.U[((size_t)k * NU + i) * Bs + b] = u[i]; }
        if (a.st_dyn) { a.st_dyn[(size_t)k * Bs + b] = st_dyn; a.st_ss[(size_t)k * Bs + b] = st_ss; a.it_dyn[(size_t)k * Bs + b] = it_dyn; a.it_ss[(size_t)k * Bs + b] = it_ss; }
        {
            double xn[NXP], pxpk[NXP];      // the plant's state equation carries p_xmp = p_x_k too (MPC_code.py:500-504,813-816)
            MPC_UNROLL for (int i = 0; i < NXP; i++) pxpk[i] = a.pxp[k * NXP + i] + ((pxh && i < NX) ? px0[i < NX ? i : 0] : 0.0);
            plant_next<NXP, NU>(P, x, u, pxpk, a.t0 + k * a.h, a.h, xn);
            MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = xn[i];
        }
    }
    MPC_UNROLL for (int i = 0; i < NXP; i++) a.x[i * Bs + b] = x[i];
    MPC_UNROLL for (int i = 0; i < NX; i++) { a.xhat[i * Bs + b] = xh[i]; a.xs[i * Bs + b] = xs[i]; }
    MPC_UNROLL for (int i = 0; i < ND; i++) a.dhat[i * Bs + b] = dh[i];
    MPC_UNROLL for (int i = 0; i < NU; i++) { a.u[i * Bs + b] = u[i]; a.us[i * Bs + b] = us[i]; }
    a.ws_valid[b] = 0;      // (this kernel's workspace layout is not a warm start for the others)
}

// The closed loop with the horizon-parallel OCP solver (mpc_tp.hpp): a workgroup of NI waves owns NI instances.
// Wave 0, lane i < NI does for instance i what one lane of loop_kernel does (estimator, target, hold rules, plant);
// all waves solve the OCPs together.  Between the two halves of a step the loop state lives in HBM.
template <int NX, int NU, int NY, int ND, int NXP, bool DU, int NG, int NC, bool MASKED, int NW, int IPW>
__global__ __launch_bounds__(64 * NW) void loop_kernel_tp(const DevProblem *__restrict__ Pp, LoopArgs a)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG, NE = NX + ND;
    using Cfg = TpCfg<NS, NU, NC, NW, IPW>;
    constexpr int NI = Cfg::NI;
    extern __shared__ double tp_smem[];
    const TpShared<NS, NU, NC, NW, IPW> sh(tp_smem);
    const DevProblem &P = *Pp;
    const size_t Bs = a.Bs;
    const int lane = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);       // uniform over a wave: scalar branches, no exec masking
    const bool wl = wave == 0 && lane < NI;
    const int b = blockIdx.x * NI + lane;
    const bool valid = wl && b < a.B;
    double *wsg = a.ws + (size_t)blockIdx.x * NI * Cfg::ROWS_ST * 64;      // state rows of this workgroup's instances
    // the warm start of the target problem lives in LDS for the steps of this launch: a dependent round trip to HBM costs
    // about 12 k cycles while the other workgroups stream their iterates, and the target solve would make three per step
    constexpr int NTW = 2 * NU + 3 * (NX + NU + NY);
    static_assert(NTW <= Cfg::KEEP_MAX, "target warm start does not fit its LDS slot");
    double *twk = sh.keep + lane * Cfg::KEEP_MAX;
    if (valid) {
        sh.keepflag[lane] = a.tw_valid[b];
        MPC_UNROLL for (int f = 0; f < NTW; f++) twk[f] = (a.tw + (size_t)f * Bs)[(unsigned)b];
    }
    MPC_STAMP_INIT
    for (int k = 0; k < a.nsteps; k++) {
        // the instance index is made opaque once per step: the address arithmetic of the ~40 per-instance arrays below is loop
        // invariant, and hoisted out of this loop it occupies registers (then scratch) for the whole kernel
        unsigned bq = (unsigned)b;      // 32-bit index next to a uniform column pointer: global_load with scalar base + vector offset
        asm volatile("" : "+v"(bq));
        double x[NXP], xh[NX], dh[ND > 0 ? ND : 1], u[NU], xs[NX], us[NU];
        double xh_pred[NX], dh_prev[ND > 0 ? ND : 1], xs_prev[NX], us_prev[NU], delta_est = 0.0;
        if (valid) {
            MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = (a.x + (size_t)(i) * Bs)[bq];
            MPC_UNROLL for (int i = 0; i < NX; i++) { xh[i] = (a.xhat + (size_t)(i) * Bs)[bq]; xs[i] = (a.xs + (size_t)(i) * Bs)[bq]; }
            MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = (a.dhat + (size_t)(i) * Bs)[bq];
            MPC_UNROLL for (int i = 0; i < NU; i++) { u[i] = (a.u + (size_t)(i) * Bs)[bq]; us[i] = (a.us + (size_t)(i) * Bs)[bq]; }
            MPC_UNROLL for (int i = 0; i < NX; i++) { xh_pred[i] = xh[i]; xs_prev[i] = xs[i]; }
            MPC_UNROLL for (int i = 0; i < ND; i++) dh_prev[i] = dh[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) us_prev[i] = us[i];
            if (a.XP) { MPC_UNROLL for (int i = 0; i < NXP; i++) (a.XP + (size_t)((size_t)k * NXP + i) * Bs)[bq] = x[i]; }
            if (a.XHAT) { MPC_UNROLL for (int i = 0; i < NX; i++) (a.XHAT + (size_t)((size_t)k * NX + i) * Bs)[bq] = xh[i]; }
            // ---- measure and estimate (MPC_code.py:524-534, 577-668) ---------------------------------
            if (P.estimator != MPC_EST_NONE) {
                double xi[NE], innov[NY];
                MPC_UNROLL for (int i = 0; i < NX; i++) xi[i] = xh[i];
                MPC_UNROLL for (int i = 0; i < ND; i++) xi[NX + i] = dh[i];
                MPC_UNROLL for (int i = 0; i < NY; i++) {
                    double yh = P.fyc[i], yy = a.pyp[k * NY + i];
                    MPC_UNROLL for (int j = 0; j < NE; j++) yh += P.Ca[i][j] * xi[j];
                    MPC_UNROLL for (int j = 0; j < NXP; j++) yy += P.Cp[i][j] * x[j];
                    innov[i] = yy - yh;
                }
                if (P.estimator == MPC_EST_KALMAN) {
                    double K[NE][NY];
                    if (a.kf_valid[bq] != 0) {      // the gain was computed one step ahead (look-ahead below, which also moves Pn into Pk)
                        MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) K[i][j] = (a.Kg + (size_t)(i * NY + j) * Bs)[bq]; }
                    } else {
                        double Pk[NE][NE];
                        MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) Pk[i][j] = (a.Pk + (size_t)(i * NE + j) * Bs)[bq]; }
                        kalman_cov<NE, NY>(P, Pk, K);
                        MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) (a.Pk + (size_t)(i * NE + j) * Bs)[bq] = Pk[i][j]; }
                    }
                    MPC_UNROLL for (int i = 0; i < NE; i++) { double s = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) s += K[i][l] * innov[l]; xi[i] += s; }      // Estimator.py:303-306
                } else {
                    MPC_UNROLL for (int i = 0; i < NE; i++) { double s = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) s += P.Kfix[i][l] * innov[l]; xi[i] += s; }
                }
                MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = xi[i];
                MPC_UNROLL for (int i = 0; i < ND; i++) { double d = xi[NX + i]; if (P.has_dsat) d = dmin(dmax(d, P.dmin[i]), P.dmax[i]); dh[i] = d; }
            }
            if (a.DHAT) { MPC_UNROLL for (int i = 0; i < ND; i++) (a.DHAT + (size_t)((size_t)k * ND + i) * Bs)[bq] = dh[i]; }
            // warm-start test, the part that does not need the target (so that the predictions need not survive the target solve)
            MPC_UNROLL for (int i = 0; i < NX; i++) delta_est = dmax(delta_est, fabs(xh[i] - xh_pred[i]));
            MPC_UNROLL for (int i = 0; i < ND; i++) delta_est = dmax(delta_est, fabs(dh[i] - dh_prev[i]));
            MPC_TSTAMP(7);
        }
        // While wave 0 solves the target problems, wave 1 (lane = instance) advances the covariance side of the Kalman
        // filter by one step: the gain of the next step and the prior after it depend on the model only, not on the data
        // (Estimator.py:297-309).
        __syncthreads();
        if (wave == 1 && P.estimator == MPC_EST_KALMAN && lane < NI && blockIdx.x * NI + lane < a.B) {
            unsigned bi = blockIdx.x * NI + lane;
            asm volatile("" : "+v"(bi));      // opaque per step, like bq above
            double Pk[NE][NE], K[NE][NY];
            if (a.kf_valid[bi] != 0) {       // the prior of the next step is Pn (wave 0 used Kg for this step): it becomes Pk
                MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) Pk[i][j] = (a.Pn + (size_t)(i * NE + j) * Bs)[bi]; }
                MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) (a.Pk + (size_t)(i * NE + j) * Bs)[bi] = Pk[i][j]; }
            } else {                         // wave 0 ran the whole filter for this step and left the next prior in Pk
                MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) Pk[i][j] = (a.Pk + (size_t)(i * NE + j) * Bs)[bi]; }
            }
            kalman_cov<NE, NY>(P, Pk, K);
            MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) (a.Kg + (size_t)(i * NY + j) * Bs)[bi] = K[i][j]; }
            MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) (a.Pn + (size_t)(i * NE + j) * Bs)[bi] = Pk[i][j]; }
            a.kf_valid[bi] = 1;
        }
        if (valid) {
            // ---- target (MPC_code.py:693-718): keep the previous one when infeasible ------------------
            double usp[NU], ysp[NY], xs_n[NX], us_n[NU], ys_n[NY];
            MPC_UNROLL for (int i = 0; i < NU; i++) usp[i] = a.usp[k * NU + i];
            MPC_UNROLL for (int i = 0; i < NY; i++) ysp[i] = a.ysp[k * NY + i];
            int it_ss;
            const int st_ss = target_lane<NX, NU, NY, ND>(P, usp, ysp, dh, us, xs_n, us_n, ys_n, it_ss, twk, 1, sh.keepflag + lane);
            if (st_ss != kInfeasible) {
                MPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = xs_n[i];
                MPC_UNROLL for (int i = 0; i < NU; i++) us[i] = us_n[i];
            }
            if (a.XS) { MPC_UNROLL for (int i = 0; i < NX; i++) (a.XS + (size_t)((size_t)k * NX + i) * Bs)[bq] = xs[i]; }
            if (a.US) { MPC_UNROLL for (int i = 0; i < NU; i++) (a.US + (size_t)((size_t)k * NU + i) * Bs)[bq] = us[i]; }
            if (a.YS) {   // ys = Fy_model(xs, us, dhat), MPC_code.py:730
                MPC_UNROLL for (int i = 0; i < NY; i++) {
                    double v = P.fyc[i];
                    MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Cm[i][j] * xs[j];
                    MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Cd[i][j] * dh[j];
                    (a.YS + (size_t)((size_t)k * NY + i) * Bs)[bq] = v;
                }
            }
            if (a.st_dyn) { (a.st_ss + (size_t)k * Bs)[bq] = st_ss; (a.it_ss + (size_t)k * Bs)[bq] = it_ss; }
            // ---- OCP data (MPC_code.py:733-761) and the warm-start test -> LDS; loop state -> HBM ------------
            OcpInst<NS, NU> q;
            build_inst<NX, NU, NY, ND, DU, NG>(P, xh, xs, us, dh, u, q);
            double delta = delta_est;
            MPC_UNROLL for (int i = 0; i < NX; i++) delta = dmax(delta, fabs(xs[i] - xs_prev[i]));
            MPC_UNROLL for (int i = 0; i < NU; i++) delta = dmax(delta, fabs(us[i] - us_prev[i]));
            const bool warm = a.ws_valid[bq] != 0 && delta <= kWsDelta && !P.no_warm;
            double *qd = sh.q + lane * Cfg::QN;
            MPC_UNROLL for (int i = 0; i < NS; i++) { qd[i] = q.z0[i]; qd[NS + i] = q.zr[i]; qd[2 * NS + i] = q.c[i]; qd[3 * NS + i] = q.zlo_m[i]; qd[4 * NS + i] = q.zhi_m[i]; }
            MPC_UNROLL for (int i = 0; i < NU; i++) { qd[5 * NS + i] = q.ur[i]; qd[5 * NS + NU + i] = q.us[i]; }
            qd[5 * NS + 2 * NU] = delta;
            sh.iflag[lane] = kTpValid | (q.ok0 ? kTpOk0 : 0) | (warm ? kTpWarm : 0);
            MPC_UNROLL for (int i = 0; i < NX; i++) { (a.xhat + (size_t)(i) * Bs)[bq] = xh[i]; (a.xs + (size_t)(i) * Bs)[bq] = xs[i]; }
            MPC_UNROLL for (int i = 0; i < ND; i++) (a.dhat + (size_t)(i) * Bs)[bq] = dh[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) (a.us + (size_t)(i) * Bs)[bq] = us[i];
        } else if (wl) sh.iflag[lane] = 0;
        int st_dyn, it_dyn;
        MPC_TSTAMP(0);
        tp_solve<NS, NU, DU, NC, MASKED, NW, IPW>(P, sh, wsg, P.max_iter, st_dyn, it_dyn);
        if (valid && P.term_cons && st_dyn != kInfeasible) {      // terminal equality missed: unreachable (mpc_device.hpp:term_missed)
            const double *fin_rows = wsg + (size_t)lane * Cfg::ROWS_ST * 64, *qd = sh.q + lane * Cfg::QN;
            double v = 0.0;
            MPC_UNROLL for (int i = 0; i < NX; i++) v = dmax(v, fabs(fin_rows[(Cfg::ST_Z + i) * 64 + P.N - 1] - qd[NS + i]) * frcp(dmax(1.0, fabs(qd[NS + i]))));
            if (!(v <= P.term_tol)) st_dyn = kInfeasible;
        }
        if (valid) {
            // ---- accept or hold (MPC_code.py:798-805), plant (MPC_code.py:813-816) ---------------------
            double x[NXP], xh[NX], u[NU];
            MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = (a.x + (size_t)(i) * Bs)[bq];
            MPC_UNROLL for (int i = 0; i < NU; i++) u[i] = (a.u + (size_t)(i) * Bs)[bq];
            if (st_dyn != kInfeasible) {
                const double *fin_rows = wsg + (size_t)lane * Cfg::ROWS_ST * 64;     // final iterate, block 0
                MPC_UNROLL for (int i = 0; i < NU; i++) u[i] = (DU && P.in_is_du) ? fin_rows[(Cfg::ST_Z + (DU ? NX + i : 0)) * 64] : fin_rows[(Cfg::ST_U + i) * 64];          // :798
                MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = fin_rows[(Cfg::ST_Z + i) * 64];         // :799
            } else {                                                           // :804-805 hold u, propagate the model
                double xo[NX], dh[ND > 0 ? ND : 1];
                MPC_UNROLL for (int i = 0; i < NX; i++) xo[i] = (a.xhat + (size_t)(i) * Bs)[bq];
                MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = (a.dhat + (size_t)(i) * Bs)[bq];
                MPC_UNROLL for (int i = 0; i < NX; i++) {
                    double v = P.fxc[i];
                    MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Am[i][j] * xo[j];
                    MPC_UNROLL for (int j = 0; j < NU; j++) v += P.Bm[i][j] * u[j];
                    MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Bd[i][j] * dh[j];
                    xh[i] = v;
                }
            }
            if (a.U) { MPC_UNROLL for (int i = 0; i < NU; i++) (a.U + (size_t)((size_t)k * NU + i) * Bs)[bq] = u[i]; }
            if (a.st_dyn) { (a.st_dyn + (size_t)k * Bs)[bq] = st_dyn; (a.it_dyn + (size_t)k * Bs)[bq] = it_dyn; }
            double xn[NXP];
            plant_next<NXP, NU>(P, x, u, a.pxp + k * NXP, a.t0 + k * a.h, a.h, xn);
            MPC_UNROLL for (int i = 0; i < NXP; i++) (a.x + (size_t)(i) * Bs)[bq] = xn[i];
            MPC_UNROLL for (int i = 0; i < NX; i++) (a.xhat + (size_t)(i) * Bs)[bq] = xh[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) (a.u + (size_t)(i) * Bs)[bq] = u[i];
            a.ws_valid[bq] = st_dyn == kSolved ? 1 : 0;
        }
        __syncthreads();
        MPC_STAMP_RESET
    }
    if (valid) {
        a.tw_valid[b] = sh.keepflag[lane];
        MPC_UNROLL for (int f = 0; f < NTW; f++) (a.tw + (size_t)f * Bs)[(unsigned)b] = twk[f];
    }
}

// Terminal equality on the wave solver (mpc_device.hpp:term_aim): after a solve the lane of the last block measures each instance's miss, aims
// the instance's terminal reference off by it and notes the iterations spent; returns - wave-uniform - whether any instance wants another pass,
// with the flags of that pass set (warm, from the iterate as it is: an instance that needs none converges at once).
template <int NS, int NU, int NC, int NX, int NI, class Cfg>
__device__ __forceinline__ bool wv_term_aim(int N, double *q, int *iflag, int *aimv, int *itacc, const WvIterA<NS, NU, NC> (&X)[NI], const WvInst (&S)[NI])
{
    const int lane = threadIdx.x;
    if (lane == N - 1) {
        MPC_UNROLL for (int j = 0; j < NI; j++) {
            double c[NX], v = 0.0;
            MPC_UNROLL for (int i = 0; i < NX; i++) { const double zr = q[j * Cfg::QN + NS + i]; c[i] = a_get(X[j].z[i]) - zr; v = dmax(v, fabs(c[i]) * frcp(dmax(1.0, fabs(zr)))); }
            const int f = iflag[j];
            const bool again = (f & kWvValid) && (f & kWvOk0) && S[j].status != kInfeasible && v > 1e-11 && v <= 1e-4;
            if (again) { MPC_UNROLL for (int i = 0; i < NX; i++) q[j * Cfg::QN + Cfg::QZN + i] -= c[i]; q[j * Cfg::QN + 5 * NS + 2 * NU] = v; }
            aimv[j] = again ? 1 : 0;
        }
    }
    __syncthreads();
    int any = 0;
    MPC_UNROLL for (int j = 0; j < NI; j++) any |= aimv[j];
    any = __builtin_amdgcn_readfirstlane(any);
    if (any && lane < NI) {
        int st = S[0].status, it = S[0].iters;
        MPC_UNROLL for (int j = 1; j < NI; j++) { if (lane == j) { st = S[j].status; it = S[j].iters; } }
        const int f = iflag[lane];
        if ((f & kWvValid) && (f & kWvOk0) && st != kInfeasible) { iflag[lane] = kWvValid | kWvOk0 | kWvWarm | kWvNoShift; itacc[lane] += it; }
    }
    __syncthreads();
    return any != 0;
}

// The closed loop on autonomous waves (mpc_wave.hpp): one wave = one workgroup = four instances, for all steps of the launch.
// Lane i < 4 does for instance i what one lane of loop_kernel does (estimator, target, hold rules, plant) on state kept in LDS;
// all 64 lanes solve the four OCPs.  HBM sees the state at the first and the last step of a launch and the logs in between.
template <int NX, int NU, int NY, int ND, int NXP, bool DU, int NG, int NC, bool MASKED, int NI>
struct WvKernelCfg {
    static constexpr int NS = NX + (DU ? NU : 0) + NG, NE = NX + ND, NDD = ND > 0 ? ND : 1;
    using Cfg = WvCfg<NS, NU, NC, NI>;
    static constexpr int NTW = 2 * NU + 3 * (NX + NU + NY);
    // per-instance state kept in LDS across the steps of a launch
    static constexpr int K_X = 0, K_XH = NXP, K_DH = K_XH + NX, K_U = K_DH + NDD, K_XS = K_U + NU, K_US = K_XS + NX, K_P = K_US + NU,
                         K_TW = K_P + NE * NE, KEEP = K_TW + NTW;
    // The transposing buffer doubles as the exchange area of the 16-lanes-per-instance estimator (four instances x XCH doubles from its
    // start): a very short horizon makes the buffer smaller than that (N = 2, CSTR: 192 against 216 doubles - the estimator overwrote the
    // instance data behind it), so the region is the larger of the two.
    static constexpr int t_region(int N)
    {
        const int t = Cfg::t_doubles(N), x = Cfg::GUARD + (Row16Tab<NX, NU, NY, ND>::fits ? 4 * Row16Tab<NX, NU, NY, ND>::XCH : 0);
        return t > x ? t : x;
    }
    static constexpr size_t lds_bytes(int N) { return sizeof(double) * ((size_t)t_region(N) + NI * Cfg::QN + NI * Cfg::OUT + NI * KEEP + (Row16Tab<NX, NU, NY, ND>::fits ? Row16Tab<NX, NU, NY, ND>::DOUBLES : 0)) + sizeof(int) * 24; }
    static constexpr int ni() { return NI; }
};

template <int NX, int NU, int NY, int ND, int NXP, bool DU, int NG, int NC, bool MASKED, int NI>
__global__ __launch_bounds__(64, 1) void loop_kernel_wv(const DevProblem *__restrict__ Pp, LoopArgs a)
{
    using KC = WvKernelCfg<NX, NU, NY, ND, NXP, DU, NG, NC, MASKED, NI>;
    using Cfg = typename KC::Cfg;
    constexpr int NS = KC::NS, NE = KC::NE, NDD = KC::NDD, NTW = KC::NTW, KEEP = KC::KEEP;
    extern __shared__ double wv_smem[];
    const DevProblem &P0 = *Pp;
    const int LD = Cfg::ld(P0.N);
    double *const T = wv_smem + Cfg::GUARD, *const q = wv_smem + KC::t_region(P0.N), *const outv = q + NI * Cfg::QN, *const keep = outv + NI * Cfg::OUT;
    int *const iflag = (int *)(keep + NI * KEEP), *const twv = iflag + 4, *const wsv = twv + 4, *const aimv = wsv + 4, *const itacc = aimv + 4;      // (terminal equality: wv_term_aim)
    using RT = Row16Tab<NX, NU, NY, ND>;
    double *const tab = (double *)(wsv + 16);      // per-row constants of the 16-lanes-per-instance phases
    // the problem constants are read through the constant address space: immutable by definition, so every access is a scalar
    // load whatever the kernel has stored to global memory in between (as plain global data they turn into vector loads + waits)
    const ConstProblem &P = *(const ConstProblem *)Pp;
    const size_t Bs = a.Bs;
    const int lane = threadIdx.x;
    const int il = lane < NI ? lane : 0;      // lane i < NI is the lane of instance i
    const int b = blockIdx.x * NI + il;
    const bool valid = lane < NI && b < a.B;
    double *const kp = keep + il * KEEP;
    if (valid) {
        MPC_UNROLL for (int i = 0; i < NXP; i++) kp[KC::K_X + i] = (a.x + (size_t)i * Bs)[b];
        MPC_UNROLL for (int i = 0; i < NX; i++) { kp[KC::K_XH + i] = (a.xhat + (size_t)i * Bs)[b]; kp[KC::K_XS + i] = (a.xs + (size_t)i * Bs)[b]; }
        MPC_UNROLL for (int i = 0; i < ND; i++) kp[KC::K_DH + i] = (a.dhat + (size_t)i * Bs)[b];
        MPC_UNROLL for (int i = 0; i < NU; i++) { kp[KC::K_U + i] = (a.u + (size_t)i * Bs)[b]; kp[KC::K_US + i] = (a.us + (size_t)i * Bs)[b]; }
        if (P.estimator == MPC_EST_KALMAN) { for (int i = 0; i < NE * NE; i++) kp[KC::K_P + i] = (a.Pk + (size_t)i * Bs)[b]; }
        for (int f = 0; f < NTW; f++) kp[KC::K_TW + f] = (a.tw + (size_t)f * Bs)[b];
        twv[lane] = a.tw_valid[b]; wsv[lane] = a.ws_valid[b];
    } else if (lane < NI) { twv[lane] = 0; wsv[lane] = 0; }
    for (int i = lane; i < NI * LD; i += 64) T[Cfg::RZ * NI * LD + i] = 0.0;      // the zero row of the tile view
    if (lane < Cfg::GUARD) wv_smem[lane] = 0.0;
    if (RT::fits) row16_fill_tables<NX, NU, NY, ND>(P, tab, lane);
    __syncthreads();
    // resident iterates: the warm start of a previous launch (inputs and bound multipliers), lane = block
    WvIterA<NS, NU, NC> X[NI];
    WvInst S[NI];
    MPC_UNROLL for (int j = 0; j < NI; j++) {
        const double *rows = a.ws + ((size_t)(blockIdx.x * NI + j) * Cfg::ROWS_WS) * 64;
        const bool w = __builtin_amdgcn_readfirstlane(wsv[j]) != 0;
        WvIter<NS, NU, NC> X0;
        MPC_UNROLL for (int i = 0; i < NU; i++) X0.u[i] = w ? rows[i * 64 + lane] : 0.0;
        MPC_UNROLL for (int i = 0; i < NC; i++) { X0.ll[i] = w ? rows[(NU + i) * 64 + lane] : 0.0; X0.lh[i] = w ? rows[(NU + NC + i) * 64 + lane] : 0.0; X0.sl[i] = 1.0; X0.sh[i] = 1.0; }
        MPC_UNROLL for (int i = 0; i < NS; i++) X0.z[i] = 0.0;
        X[j].put(X0);
    }
    MPC_STAMP_INIT
    int redo = 0;      // terminal equality: passes of this step's OCP with the terminal reference aimed off (wv_term_aim) - the step's loop body once more, solve onwards
    for (int k = 0; k < a.nsteps; k++) {
        unsigned bq = (unsigned)b;      // opaque per step: the address arithmetic of the log arrays stays next to the stores
        asm volatile("" : "+v"(bq));
        if (redo == 0) {
        if constexpr (RT::fits) {
            // ---- estimator and target with 16 lanes per instance (mpc_wave.hpp) --------------------------------------
            const int b16 = lane >> 4, r16 = lane & 15, b16c = b16 < NI ? b16 : 0;
            const unsigned bi = (unsigned)(blockIdx.x * NI + b16c);
            const bool inst16 = b16 < NI && (int)bi < a.B;
            double *const kq = keep + b16c * KEEP;
            if (inst16) {
                if (a.XP && r16 < NXP) (a.XP + (size_t)((size_t)k * NXP + r16) * Bs)[bi] = kq[KC::K_X + r16];
                if (a.XHAT && r16 < NX) (a.XHAT + (size_t)((size_t)k * NX + r16) * Bs)[bi] = kq[KC::K_XH + r16];
            }
            double xi_old = 0.0, xi_new = 0.0;
            if (P.estimator != MPC_EST_NONE)
                kalman_row16<NX, NY, ND, NXP, NU>(P, r16, b16 < NI, kq, KC::K_X, KC::K_XH, KC::K_P, a.pyp + k * NY, tab, T + b16c * RT::XCH, xi_old, xi_new);
            if (inst16 && a.DHAT && r16 >= NX && r16 < NE) (a.DHAT + (size_t)((size_t)k * ND + (r16 - NX)) * Bs)[bi] = xi_new;
            double delta = row16_max(r16 < NE ? fabs(xi_new - xi_old) : 0.0);      // warm-start test: estimate against its prediction
            __syncthreads();
            // (the estimator's exchange area is the start of T: with a very short horizon it reaches the zero row of the tile view)
            if (Cfg::RZ * NI * LD < 4 * RT::XCH) { for (int i = lane; i < NI * LD; i += 64) T[Cfg::RZ * NI * LD + i] = 0.0; }
            double dh[NDD], usv[NU];
            MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = kq[KC::K_DH + i];
            MPC_UNROLL for (int i = 0; i < NU; i++) usv[i] = kq[KC::K_US + i];
            int it_ss; double vrow;
            const int st_ss = target_row16<NX, NU, NY, ND>(P, r16, tab, a.usp + k * NU, a.ysp + k * NY, dh, usv, kq + KC::K_TW, twv + b16c, inst16, vrow, it_ss);
            // accept, or keep the previous target when infeasible (MPC_code.py:714-718); rows 0..NX-1 = xs, NX..NX+NU-1 = us
            const bool trow = r16 < NX + NU && b16 < NI;      // lanes of unused instance slots (NI < 4) alias slot 0: they must not write
            const double prev = trow ? kq[KC::K_XS + r16] : 0.0;      // K_XS.. and K_US.. are adjacent
            const double tnew = (st_ss != kInfeasible && trow) ? vrow : prev;
            delta = dmax(delta, row16_max(fabs(tnew - prev)));
            if (trow) kq[KC::K_XS + r16] = tnew;
            if (inst16) {
                if (a.XS && r16 < NX) (a.XS + (size_t)((size_t)k * NX + r16) * Bs)[bi] = tnew;
                if (a.US && r16 >= NX && trow) (a.US + (size_t)((size_t)k * NU + (r16 - NX)) * Bs)[bi] = tnew;
                if (a.st_dyn && r16 == 0) { (a.st_ss + (size_t)k * Bs)[bi] = st_ss; (a.it_ss + (size_t)k * Bs)[bi] = it_ss; }
            }
            if (r16 == 0 && b16 < NI) outv[b16c * Cfg::OUT] = delta;
            __syncthreads();
        }
        if (valid) {
            double xh[NX], dh[NDD], u[NU], xs[NX], us[NU], delta = 0.0;
            MPC_UNROLL for (int i = 0; i < NX; i++) { xh[i] = kp[KC::K_XH + i]; xs[i] = kp[KC::K_XS + i]; }
            MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = kp[KC::K_DH + i];
            MPC_UNROLL for (int i = 0; i < NU; i++) { u[i] = kp[KC::K_U + i]; us[i] = kp[KC::K_US + i]; }
            if constexpr (RT::fits) delta = outv[lane * Cfg::OUT];
            else {
                double x[NXP];
                double xh_pred[NX], dh_prev[NDD], xs_prev[NX], us_prev[NU];
                MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = kp[KC::K_X + i];
                MPC_UNROLL for (int i = 0; i < NX; i++) { xh[i] = kp[KC::K_XH + i]; xs[i] = kp[KC::K_XS + i]; xh_pred[i] = xh[i]; xs_prev[i] = xs[i]; }
                MPC_UNROLL for (int i = 0; i < ND; i++) { dh[i] = kp[KC::K_DH + i]; dh_prev[i] = dh[i]; }
                MPC_UNROLL for (int i = 0; i < NU; i++) { u[i] = kp[KC::K_U + i]; us[i] = kp[KC::K_US + i]; us_prev[i] = us[i]; }
                if (a.XP) { MPC_UNROLL for (int i = 0; i < NXP; i++) (a.XP + (size_t)((size_t)k * NXP + i) * Bs)[bq] = x[i]; }
                if (a.XHAT) { MPC_UNROLL for (int i = 0; i < NX; i++) (a.XHAT + (size_t)((size_t)k * NX + i) * Bs)[bq] = xh[i]; }
                // ---- measure and estimate (MPC_code.py:524-534, 577-668) ---------------------------------
                if (P.estimator != MPC_EST_NONE) {
                    double xi[NE], innov[NY];
                    MPC_UNROLL for (int i = 0; i < NX; i++) xi[i] = xh[i];
                    MPC_UNROLL for (int i = 0; i < ND; i++) xi[NX + i] = dh[i];
                    MPC_UNROLL for (int i = 0; i < NY; i++) {
                        double yh = P.fyc[i], yy = a.pyp[k * NY + i];
                        MPC_UNROLL for (int j = 0; j < NE; j++) yh += P.Ca[i][j] * xi[j];
                        MPC_UNROLL for (int j = 0; j < NXP; j++) yy += P.Cp[i][j] * x[j];
                        innov[i] = yy - yh;
                    }
                    if (P.estimator == MPC_EST_KALMAN) {
                        double Pk[NE][NE];
                        MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) Pk[i][j] = kp[KC::K_P + i * NE + j]; }
                        kalman_lane<NE, NY>(P, xi, Pk, innov);
                        MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) kp[KC::K_P + i * NE + j] = Pk[i][j]; }
                    } else {
                        MPC_UNROLL for (int i = 0; i < NE; i++) { double s = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) s += P.Kfix[i][l] * innov[l]; xi[i] += s; }
                    }
                    MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = xi[i];
                    MPC_UNROLL for (int i = 0; i < ND; i++) { double d = xi[NX + i]; if (P.has_dsat) d = dmin(dmax(d, P.dmin[i]), P.dmax[i]); dh[i] = d; }
                }
                if (a.DHAT) { MPC_UNROLL for (int i = 0; i < ND; i++) (a.DHAT + (size_t)((size_t)k * ND + i) * Bs)[bq] = dh[i]; }
                MPC_UNROLL for (int i = 0; i < NX; i++) delta = dmax(delta, fabs(xh[i] - xh_pred[i]));
                MPC_UNROLL for (int i = 0; i < ND; i++) delta = dmax(delta, fabs(dh[i] - dh_prev[i]));
                // ---- target (MPC_code.py:693-718): keep the previous one when infeasible ------------------
                double usp[NU], ysp[NY], xs_n[NX], us_n[NU], ys_n[NY];
                MPC_UNROLL for (int i = 0; i < NU; i++) usp[i] = a.usp[k * NU + i];
                MPC_UNROLL for (int i = 0; i < NY; i++) ysp[i] = a.ysp[k * NY + i];
                int it_ss;
                const int st_ss = target_lane<NX, NU, NY, ND>(P, usp, ysp, dh, us, xs_n, us_n, ys_n, it_ss, kp + KC::K_TW, 1, twv + lane);
                if (st_ss != kInfeasible) {
                    MPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = xs_n[i];
                    MPC_UNROLL for (int i = 0; i < NU; i++) us[i] = us_n[i];
                }
                if (a.XS) { MPC_UNROLL for (int i = 0; i < NX; i++) (a.XS + (size_t)((size_t)k * NX + i) * Bs)[bq] = xs[i]; }
                if (a.US) { MPC_UNROLL for (int i = 0; i < NU; i++) (a.US + (size_t)((size_t)k * NU + i) * Bs)[bq] = us[i]; }
                if (a.st_dyn) { (a.st_ss + (size_t)k * Bs)[bq] = st_ss; (a.it_ss + (size_t)k * Bs)[bq] = it_ss; }
                MPC_UNROLL for (int i = 0; i < NX; i++) delta = dmax(delta, fabs(xs[i] - xs_prev[i]));
                MPC_UNROLL for (int i = 0; i < NU; i++) delta = dmax(delta, fabs(us[i] - us_prev[i]));
            }
            if (a.YS) {   // ys = Fy_model(xs, us, dhat), MPC_code.py:730
                MPC_UNROLL for (int i = 0; i < NY; i++) {
                    double v = P.fyc[i];
                    MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Cm[i][j] * xs[j];
                    MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Cd[i][j] * dh[j];
                    (a.YS + (size_t)((size_t)k * NY + i) * Bs)[bq] = v;
                }
            }
            // ---- OCP data (MPC_code.py:733-761) and the warm-start test -> LDS ------------------------------
            OcpInst<NS, NU> qi;
            build_inst<NX, NU, NY, ND, DU, NG>(P, xh, xs, us, dh, u, qi);
            const bool warm = wsv[lane] != 0 && delta <= kWsDelta && !P.no_warm;
            double *qd = q + lane * Cfg::QN;
            MPC_UNROLL for (int i = 0; i < NS; i++) { qd[i] = qi.z0[i]; qd[NS + i] = qi.zr[i]; qd[2 * NS + i] = qi.c[i]; qd[3 * NS + i] = qi.zlo_m[i]; qd[4 * NS + i] = qi.zhi_m[i]; qd[Cfg::QZN + i] = qi.zrN[i]; }
            MPC_UNROLL for (int i = 0; i < NU; i++) { qd[5 * NS + i] = qi.ur[i]; qd[5 * NS + NU + i] = qi.us[i]; }
            qd[5 * NS + 2 * NU] = delta;
            iflag[lane] = kWvValid | (qi.ok0 ? kWvOk0 : 0) | (warm ? kWvWarm : 0);
            itacc[lane] = 0;
            MPC_UNROLL for (int i = 0; i < NX; i++) { kp[KC::K_XH + i] = xh[i]; kp[KC::K_XS + i] = xs[i]; }
            MPC_UNROLL for (int i = 0; i < ND; i++) kp[KC::K_DH + i] = dh[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) kp[KC::K_US + i] = us[i];
        } else if (lane < NI) iflag[lane] = 0;
        __syncthreads();
        }
        MPC_TSTAMP(0);
        wv_solve<NS, NU, DU, NC, MASKED, NI>(P, T, q, iflag, X, S, P.max_iter);
        MPC_STAMP_RESET
        // first input and next state of the final iterates: block 0 = lane 0
        if (lane == 0) {
            MPC_UNROLL for (int j = 0; j < NI; j++) {
                MPC_UNROLL for (int i = 0; i < NU; i++) outv[j * Cfg::OUT + i] = a_get(X[j].u[i]);
                MPC_UNROLL for (int i = 0; i < NS; i++) outv[j * Cfg::OUT + NU + i] = a_get(X[j].z[i]);
            }
        }
        if (P.term_cons && lane == P.N - 1) {      // terminal equality (mpc_device.hpp:term_missed, term_aim): the last block is lane N - 1
            MPC_UNROLL for (int j = 0; j < NI; j++) {
                double v = 0.0, c[NX];
                MPC_UNROLL for (int i = 0; i < NX; i++) { const double zr = q[j * Cfg::QN + NS + i]; c[i] = a_get(X[j].z[i]) - zr; v = dmax(v, fabs(c[i]) * frcp(dmax(1.0, fabs(zr)))); }
                outv[j * Cfg::OUT + NU + NS] = v;
                // another pass with the terminal reference aimed off by the miss?  (at most two; not for a target out of reach or a failed solve)
                const int f = iflag[j];
                const bool again = redo < 2 && (f & kWvValid) && (f & kWvOk0) && S[j].status != kInfeasible && v > 1e-11 && v <= 1e-4;
                if (again) { MPC_UNROLL for (int i = 0; i < NX; i++) q[j * Cfg::QN + Cfg::QZN + i] -= c[i]; q[j * Cfg::QN + 5 * NS + 2 * NU] = v; }
                aimv[j] = again ? 1 : 0;
            }
        }
        __syncthreads();
        if (P.term_cons && redo < 2) {
            int any = 0;
            MPC_UNROLL for (int j = 0; j < NI; j++) any |= aimv[j];
            if (__builtin_amdgcn_readfirstlane(any)) {      // the step's loop body once more, from the solve onwards: warm, from the iterates as they are
                if (lane < NI) {
                    int st = S[0].status, it = S[0].iters;
                    MPC_UNROLL for (int j = 1; j < NI; j++) { if (lane == j) { st = S[j].status; it = S[j].iters; } }
                    const int f = iflag[lane];
                    if ((f & kWvValid) && (f & kWvOk0) && st != kInfeasible) { iflag[lane] = kWvValid | kWvOk0 | kWvWarm | kWvNoShift; itacc[lane] += it; }
                }
                __syncthreads();
                redo++; k--;
                continue;
            }
        }
        redo = 0;
        if (valid) {
            // ---- accept or hold (MPC_code.py:798-805), plant (MPC_code.py:813-816) ---------------------
            int st_dyn = S[0].status, it_dyn = S[0].iters;
            MPC_UNROLL for (int j = 1; j < NI; j++) { if (lane == j) { st_dyn = S[j].status; it_dyn = S[j].iters; } }
            if (P.term_cons) it_dyn += itacc[lane];
            if (P.term_cons && st_dyn != kInfeasible && !(outv[lane * Cfg::OUT + NU + NS] <= P.term_tol)) st_dyn = kInfeasible;
            double x[NXP], xh[NX], u[NU];
            MPC_UNROLL for (int i = 0; i < NXP; i++) x[i] = kp[KC::K_X + i];
            MPC_UNROLL for (int i = 0; i < NU; i++) u[i] = kp[KC::K_U + i];
            if (st_dyn != kInfeasible) {
                MPC_UNROLL for (int i = 0; i < NU; i++) u[i] = (DU && P.in_is_du) ? outv[lane * Cfg::OUT + NU + (DU ? NX + i : 0)] : outv[lane * Cfg::OUT + i];               // :798
                MPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = outv[lane * Cfg::OUT + NU + i];         // :799
            } else {                                                           // :804-805 hold u, propagate the model
                double xo[NX], dh[NDD];
                MPC_UNROLL for (int i = 0; i < NX; i++) xo[i] = kp[KC::K_XH + i];
                MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = kp[KC::K_DH + i];
                MPC_UNROLL for (int i = 0; i < NX; i++) {
                    double v = P.fxc[i];
                    MPC_UNROLL for (int j = 0; j < NX; j++) v += P.Am[i][j] * xo[j];
                    MPC_UNROLL for (int j = 0; j < NU; j++) v += P.Bm[i][j] * u[j];
                    MPC_UNROLL for (int j = 0; j < ND; j++) v += P.Bd[i][j] * dh[j];
                    xh[i] = v;
                }
            }
            if (a.U) { MPC_UNROLL for (int i = 0; i < NU; i++) (a.U + (size_t)((size_t)k * NU + i) * Bs)[bq] = u[i]; }
            if (a.st_dyn) { (a.st_dyn + (size_t)k * Bs)[bq] = st_dyn; (a.it_dyn + (size_t)k * Bs)[bq] = it_dyn; }
            {
                double xn[NXP];
                plant_next<NXP, NU>(P, x, u, a.pxp + k * NXP, a.t0 + k * a.h, a.h, xn);
                MPC_UNROLL for (int i = 0; i < NXP; i++) kp[KC::K_X + i] = xn[i];
            }
            MPC_UNROLL for (int i = 0; i < NX; i++) kp[KC::K_XH + i] = xh[i];
            MPC_UNROLL for (int i = 0; i < NU; i++) kp[KC::K_U + i] = u[i];
            wsv[lane] = st_dyn == kSolved ? 1 : 0;
        }
        __syncthreads();
        MPC_TSTAMP(7);
    }
    if (valid) {
        MPC_UNROLL for (int i = 0; i < NXP; i++) (a.x + (size_t)i * Bs)[b] = kp[KC::K_X + i];
        MPC_UNROLL for (int i = 0; i < NX; i++) { (a.xhat + (size_t)i * Bs)[b] = kp[KC::K_XH + i]; (a.xs + (size_t)i * Bs)[b] = kp[KC::K_XS + i]; }
        MPC_UNROLL for (int i = 0; i < ND; i++) (a.dhat + (size_t)i * Bs)[b] = kp[KC::K_DH + i];
        MPC_UNROLL for (int i = 0; i < NU; i++) { (a.u + (size_t)i * Bs)[b] = kp[KC::K_U + i]; (a.us + (size_t)i * Bs)[b] = kp[KC::K_US + i]; }
        if (P.estimator == MPC_EST_KALMAN) { for (int i = 0; i < NE * NE; i++) (a.Pk + (size_t)i * Bs)[b] = kp[KC::K_P + i]; }
        for (int f = 0; f < NTW; f++) (a.tw + (size_t)f * Bs)[b] = kp[KC::K_TW + f];
        a.tw_valid[b] = twv[lane]; a.ws_valid[b] = wsv[lane];
    }
    MPC_UNROLL for (int j = 0; j < NI; j++) {
        double *rows = a.ws + ((size_t)(blockIdx.x * NI + j) * Cfg::ROWS_WS) * 64;
        MPC_UNROLL for (int i = 0; i < NU; i++) rows[i * 64 + lane] = a_get(X[j].u[i]);
        MPC_UNROLL for (int i = 0; i < NC; i++) { rows[(NU + i) * 64 + lane] = a_get(X[j].ll[i]); rows[(NU + NC + i) * 64 + lane] = a_get(X[j].lh[i]); }
    }
}

// One solver(...) call per instance (MPC_code.py:776-781) on the wave-autonomous solver: the per-call entry point mpc_ocp_solve.
// With the warm start on (option "ocp_warm_start") the bound multipliers of the previous call stay in the handle's workspace and
// are shifted by one stage like the closed loop does; the inputs come from the caller's guess w when there is
// one (the reference hands IPOPT the shifted previous optimum, MPC_code.py:740-764), else from the previous call as well.
struct OcpWvArgs {
    OcpArgs o;
    const double *u_guess;      // [B][nu][64] inputs of the caller's guess (lane = stage), or nullptr
    double *traj;               // [B][nu + ns][64] final iterate: u | z rows (lane = stage), or nullptr
    double *prev;               // [2 nx + nd + nu][Bs] data of the previous call: x1 prediction, dhat, xs, us
    int32_t *valid;             // [Bs] the workspace holds the multipliers of a solved previous call
    int warm_on;
};

template <int NX, int NU, int NY, int ND, bool DU, int NG, int NC, bool MASKED, int NI>
__global__ __launch_bounds__(64, 1) void ocp_kernel_wv(const DevProblem *__restrict__ Pp, OcpWvArgs w)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG, NDD = ND > 0 ? ND : 1;
    using Cfg = WvCfg<NS, NU, NC, NI>;
    const OcpArgs &a = w.o;
    extern __shared__ double wv_smem[];
    const ConstProblem &P = *(const ConstProblem *)Pp;
    const int N = Pp->N, LD = Cfg::ld(N);
    double *const T = wv_smem + Cfg::GUARD, *const q = wv_smem + Cfg::t_doubles(N), *const outv = q + NI * Cfg::QN;
    int *const iflag = (int *)(outv + NI * Cfg::OUT), *const aimv = iflag + 4, *const itacc = aimv + 4;
    const size_t Bs = a.Bs;
    const int lane = threadIdx.x;
    const int il = lane < NI ? lane : 0;
    const int b = blockIdx.x * NI + il;
    const bool valid = lane < NI && b < a.B;
    for (int i = lane; i < NI * LD; i += 64) T[Cfg::RZ * NI * LD + i] = 0.0;
    if (lane < Cfg::GUARD) wv_smem[lane] = 0.0;
    if (valid) {
        double xh[NX], xs[NX], us[NU], up[NU], dh[NDD];
        MPC_UNROLL for (int i = 0; i < NX; i++) { xh[i] = a.xhat[i * Bs + b]; xs[i] = a.xs[i * Bs + b]; }
        MPC_UNROLL for (int i = 0; i < NU; i++) { us[i] = a.us[i * Bs + b]; up[i] = a.u_prev[i * Bs + b]; }
        MPC_UNROLL for (int i = 0; i < ND; i++) dh[i] = a.dhat[i * Bs + b];
        OcpInst<NS, NU> qi;
        build_inst<NX, NU, NY, ND, DU, NG>(P, xh, xs, us, dh, up, qi);
        // warm-start test against the previous call's data: estimate vs its prediction, disturbance, target
        double delta = 0.0;
        MPC_UNROLL for (int i = 0; i < NX; i++) delta = dmax(delta, dmax(fabs(xh[i] - w.prev[i * Bs + b]), fabs(xs[i] - w.prev[(NX + ND + i) * Bs + b])));
        MPC_UNROLL for (int i = 0; i < ND; i++) delta = dmax(delta, fabs(dh[i] - w.prev[(NX + i) * Bs + b]));
        MPC_UNROLL for (int i = 0; i < NU; i++) delta = dmax(delta, fabs(us[i] - w.prev[(2 * NX + ND + i) * Bs + b]));
        const bool warm = w.warm_on && w.valid[b] != 0 && delta <= kWsDelta && !P.no_warm;
        double *qd = q + lane * Cfg::QN;
        MPC_UNROLL for (int i = 0; i < NS; i++) { qd[i] = qi.z0[i]; qd[NS + i] = qi.zr[i]; qd[2 * NS + i] = qi.c[i]; qd[3 * NS + i] = qi.zlo_m[i]; qd[4 * NS + i] = qi.zhi_m[i]; qd[Cfg::QZN + i] = qi.zrN[i]; }
        MPC_UNROLL for (int i = 0; i < NU; i++) { qd[5 * NS + i] = qi.ur[i]; qd[5 * NS + NU + i] = qi.us[i]; }
        qd[5 * NS + 2 * NU] = delta;
        iflag[lane] = kWvValid | (qi.ok0 ? kWvOk0 : 0) | (warm ? kWvWarm : 0) | ((warm && w.u_guess) ? kWvKeepU : 0);
        itacc[lane] = 0;
        MPC_UNROLL for (int i = 0; i < ND; i++) w.prev[(NX + i) * Bs + b] = dh[i];
        MPC_UNROLL for (int i = 0; i < NX; i++) w.prev[(NX + ND + i) * Bs + b] = xs[i];
        MPC_UNROLL for (int i = 0; i < NU; i++) w.prev[(2 * NX + ND + i) * Bs + b] = us[i];
    } else if (lane < NI) iflag[lane] = 0;
    __syncthreads();
    WvIterA<NS, NU, NC> X[NI];
    WvInst S[NI];
    MPC_UNROLL for (int j = 0; j < NI; j++) {
        const size_t bj = (size_t)blockIdx.x * NI + j;
        const double *rows = a.ws + (bj * Cfg::ROWS_WS) * 64;
        const int fl = __builtin_amdgcn_readfirstlane(iflag[j]);
        const bool wm = (fl & kWvWarm) != 0, ku = (fl & kWvKeepU) != 0;
        WvIter<NS, NU, NC> X0;
        MPC_UNROLL for (int i = 0; i < NU; i++) X0.u[i] = wm ? (ku ? w.u_guess[(bj * NU + i) * 64 + lane] : rows[i * 64 + lane]) : 0.0;
        MPC_UNROLL for (int i = 0; i < NC; i++) { X0.ll[i] = wm ? rows[(NU + i) * 64 + lane] : 0.0; X0.lh[i] = wm ? rows[(NU + NC + i) * 64 + lane] : 0.0; X0.sl[i] = 1.0; X0.sh[i] = 1.0; }
        MPC_UNROLL for (int i = 0; i < NS; i++) X0.z[i] = 0.0;
        X[j].put(X0);
    }
    for (int pass = 0;; pass++) {      // (terminal equality: wv_term_aim)
        wv_solve<NS, NU, DU, NC, MASKED, NI>(P, T, q, iflag, X, S, P.max_iter);
        if (!P.term_cons || pass == 2 || !wv_term_aim<NS, NU, NC, NX, NI, Cfg>(N, q, iflag, aimv, itacc, X, S)) break;
    }
    MPC_UNROLL for (int j = 0; j < NI; j++) {
        const size_t bj = (size_t)blockIdx.x * NI + j;
        double *rows = a.ws + (bj * Cfg::ROWS_WS) * 64;
        WvIter<NS, NU, NC> Xf; X[j].get(Xf);
        MPC_UNROLL for (int i = 0; i < NU; i++) rows[i * 64 + lane] = Xf.u[i];
        MPC_UNROLL for (int i = 0; i < NC; i++) { rows[(NU + i) * 64 + lane] = Xf.ll[i]; rows[(NU + NC + i) * 64 + lane] = Xf.lh[i]; }
        if (w.traj) {
            MPC_UNROLL for (int i = 0; i < NU; i++) w.traj[(bj * (NU + NS) + i) * 64 + lane] = Xf.u[i];
            MPC_UNROLL for (int i = 0; i < NS; i++) w.traj[(bj * (NU + NS) + NU + i) * 64 + lane] = Xf.z[i];
        }
        if (lane == 0) {
            MPC_UNROLL for (int i = 0; i < NU; i++) outv[j * Cfg::OUT + i] = Xf.u[i];
            MPC_UNROLL for (int i = 0; i < NS; i++) outv[j * Cfg::OUT + NU + i] = Xf.z[i];
        }
        if (P.term_cons && lane == P.N - 1) {      // terminal equality (mpc_device.hpp:term_missed)
            double v = 0.0;
            MPC_UNROLL for (int i = 0; i < NX; i++) { const double zr = q[j * Cfg::QN + NS + i]; v = dmax(v, fabs(Xf.z[i] - zr) * frcp(dmax(1.0, fabs(zr)))); }
            outv[j * Cfg::OUT + NU + NS] = v;
        }
    }
    __syncthreads();
    if (valid) {
        int st = S[0].status, it = S[0].iters; double r0 = S[0].res_s, r1 = S[0].res_p, r2 = S[0].mu;
        MPC_UNROLL for (int j = 1; j < NI; j++) { if (lane == j) { st = S[j].status; it = S[j].iters; r0 = S[j].res_s; r1 = S[j].res_p; r2 = S[j].mu; } }
        if (P.term_cons) it += itacc[lane];
        if (P.term_cons && st != kInfeasible && !(outv[lane * Cfg::OUT + NU + NS] <= P.term_tol)) st = kInfeasible;
        a.status[b] = st; a.iters[b] = it;
        a.res[0 * Bs + b] = r0; a.res[1 * Bs + b] = r1; a.res[2 * Bs + b] = r2;
        w.valid[b] = st == kSolved ? 1 : 0;
        if (st != kInfeasible) {
            MPC_UNROLL for (int i = 0; i < NU; i++) a.u_out[i * Bs + b] = (DU && P.in_is_du) ? outv[lane * Cfg::OUT + NU + (DU ? NX + i : 0)] : outv[lane * Cfg::OUT + i];
            MPC_UNROLL for (int i = 0; i < NX; i++) { const double v = outv[lane * Cfg::OUT + NU + i]; a.xnext_out[i * Bs + b] = v; w.prev[i * Bs + b] = v; }
        }
    }
}

// dense [B][nu] copy of u for the all-gather of u* 
#ifndef MPC_PART2
__global__ void pack_u_kernel(const double *__restrict__ u, double *__restrict__ dst, int B, size_t Bs, int nu)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int i = 0; i < nu; i++) dst[(size_t)b * nu + i] = u[i * Bs + b];
}
#endif

// ---------------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------------
struct Launchers {
    void (*ocp)(const DevProblem *, OcpArgs, hipStream_t);
    void (*ocp_pxy)(const DevProblem *, OcpPxyArgs, hipStream_t);      // time-varying px / py: one variant (all bounds maskable)
    void (*ocp_soft)(const DevProblem *, OcpSoftArgs, hipStream_t);    // soft output constraints (mpc_soft.hpp)
    int soft_fields;                                                   // doubles per block and lane of its workspace
    void (*loop_soft)(const DevProblem *, LoopArgs, hipStream_t);      // the closed loop with soft output constraints (instance per lane)
    void (*loop_pxy)(const DevProblem *, LoopArgs, hipStream_t);       // the closed loop with def_px / def_py schedules (instance per lane)
    int pxy_ws_rows, pxy_nc, pxy_lin;                                  // its workspace rows / bounded variables / slab entries per block
    void (*target)(const DevProblem *, TargetArgs, hipStream_t);
    void (*kf)(const DevProblem *, KfArgs, hipStream_t);
    void (*loop)(const DevProblem *, LoopArgs, hipStream_t);
    int (*loop_tp)(const DevProblem *, LoopArgs, hipStream_t);     // horizon-parallel variant (N <= 64), nullptr if it does not fit
    int (*loop_wv)(const DevProblem *, LoopArgs, hipStream_t);     // wave-autonomous variant (N <= 64, stage state <= 8, nu <= 2), nullptr otherwise
    int (*ocp_wv)(const DevProblem *, OcpWvArgs, int, hipStream_t);    // the per-call solve on the same solver
    int wv_ns;
    size_t wv_ws_per_inst, wv_lds;
    int ws_rows, nc, tp_ni;
    int tp_max_batch;           // auto choice of the loop kernel: largest batch the horizon-parallel kernel is preferred for
    size_t tp_ws_per_inst, tp_ws_per_group, tp_lds;
};

static constexpr int kTpMaxBatch = 16384;     // auto choice of the loop kernel without the matrix-core factorisation, see loop_mode()

// bound modes: which variant of the OCP kernels a problem may use (cheapest first)
enum { kBoundsAllFinite = 1, kBoundsInputsOnly = 2, kBoundsGeneric = 0 };

template <int NX, int NU, int NY, int ND, int NXP, bool DU, int NG, int NC, bool MASKED, int NI>
static int launch_loop_wv(const DevProblem *p, LoopArgs a, hipStream_t s, int dev)
{
    using KC = WvKernelCfg<NX, NU, NY, ND, NXP, DU, NG, NC, MASKED, NI>;
    auto kern = loop_kernel_wv<NX, NU, NY, ND, NXP, DU, NG, NC, MASKED, NI>;
    static bool attr_set[64] = {};      // per device: more than 64 KB of dynamic LDS has to be asked for
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KC::lds_bytes(64)) != hipSuccess) return -1;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((a.B + NI - 1) / NI), dim3(64), KC::lds_bytes(a.N), s, p, a);
    return 0;
}

template <int NX, int NU, int NY, int ND, int NXP, bool DU, int NG, int NC, bool MASKED>
static Launchers make_launchers_mode()
{
    Launchers l;
    l.ocp = [](const DevProblem *p, OcpArgs a, hipStream_t s) { hipLaunchKernelGGL((ocp_kernel<NX, NU, NY, ND, DU, NG, NC, MASKED>), dim3((a.B + 63) / 64), dim3(64), 0, s, p, a); };
    {
        constexpr int NSZ = NX + (DU ? NU : 0) + NG;
        l.ocp_pxy = [](const DevProblem *p, OcpPxyArgs a, hipStream_t s) { hipLaunchKernelGGL((ocp_kernel_pxy<NX, NU, NY, ND, DU, NG>), dim3((a.o.B + 63) / 64), dim3(64), 0, s, p, a); };
        l.loop_pxy = [](const DevProblem *p, LoopArgs a, hipStream_t s) { hipLaunchKernelGGL((loop_kernel_pxy<NX, NU, NY, ND, NXP, DU, NG>), dim3((a.B + 63) / 64), dim3(64), 0, s, p, a); };
        l.pxy_ws_rows = 2 * BlkLayout<NSZ, NU, NSZ + NU>::SLOTS; l.pxy_nc = NSZ + NU; l.pxy_lin = NSZ * (NSZ + NU + 1) + 2 * NSZ;
        l.ocp_soft = [](const DevProblem *p, OcpSoftArgs a, hipStream_t s) { hipLaunchKernelGGL((ocp_kernel_soft<NX, NU, NY, ND, DU, NG>), dim3((a.o.B + 63) / 64), dim3(64), 0, s, p, a); };
        l.soft_fields = SoftLayout<NSZ, NU, NY>::FIELDS;
        l.loop_soft = [](const DevProblem *p, LoopArgs a, hipStream_t s) { hipLaunchKernelGGL((loop_kernel_soft<NX, NU, NY, ND, NXP, DU, NG>), dim3((a.B + 63) / 64), dim3(64), 0, s, p, a); };
    }
    l.target = [](const DevProblem *p, TargetArgs a, hipStream_t s) { hipLaunchKernelGGL((target_kernel<NX, NU, NY, ND>), dim3((a.B + 63) / 64), dim3(64), 0, s, p, a); };
    l.kf = [](const DevProblem *p, KfArgs a, hipStream_t s) { hipLaunchKernelGGL((kf_kernel<NX, NY, ND>), dim3((a.B + 63) / 64), dim3(64), 0, s, p, a); };
    l.loop = [](const DevProblem *p, LoopArgs a, hipStream_t s) { hipLaunchKernelGGL((loop_kernel<NX, NU, NY, ND, NXP, DU, NG, NC, MASKED>), dim3((a.B + 63) / 64), dim3(64), 0, s, p, a); };
    l.ws_rows = 2 * BlkLayout<NX + (DU ? NU : 0) + NG, NU, NC>::SLOTS;   // doubles per instance per block
    {
        // eight waves (256 VGPRs each); two instances per wave when their transposing buffer still fits the 160 KB of LDS
        constexpr int NW = 8, NSZ = NX + (DU ? NU : 0) + NG;
        constexpr int IPW = TpCfg<NSZ, NU, NC, NW, 2>::lds_bytes() <= 160 * 1024 ? 2 : 1;
        using Cfg = TpCfg<NSZ, NU, NC, NW, IPW>;
        constexpr size_t lds = Cfg::lds_bytes();
        l.tp_ni = Cfg::NI; l.tp_lds = lds; l.tp_ws_per_inst = sizeof(double) * 64 * Cfg::ROWS_ST; l.tp_ws_per_group = 0;
        l.loop_tp = nullptr;
        // measured on LMPC-CSTR / Wood-Berry: with the factorisation on the matrix cores (stage fits a 4x4 tile)
        // the horizon-parallel kernel wins at every batch size; with the lane = instance factorisation it wins up to about 16384
        l.tp_max_batch = (NSZ <= 4 && NU <= 2) ? INT32_MAX : kTpMaxBatch;
        if (lds <= 160 * 1024) {
            l.loop_tp = [](const DevProblem *p, LoopArgs a, hipStream_t s) -> int {
                auto kern = loop_kernel_tp<NX, NU, NY, ND, NXP, DU, NG, NC, MASKED, NW, IPW>;
                static bool attr_set[64] = {};      // per device: more than 64 KB of dynamic LDS has to be asked for
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
                if (!attr_set[dev]) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
                    attr_set[dev] = true;
                }
                hipLaunchKernelGGL(kern, dim3((a.B + Cfg::NI - 1) / Cfg::NI), dim3(64, NW), lds, s, p, a);
                return 0;
            };
        }
    }
    l.loop_wv = nullptr; l.ocp_wv = nullptr; l.wv_ns = 0; l.wv_ws_per_inst = 0; l.wv_lds = 0;
    if constexpr (NX + (DU ? NU : 0) + NG <= 8 && NU <= 2) {
        // instances per wave: four (one per tile of a matrix-core product) when their resident iterates fit the 256 accumulation
        // registers they are parked in (4 NC slacks / multipliers + NV primal + NC predictor direction, two registers each), else two
        constexpr int NSZ_ = NX + (DU ? NU : 0) + NG;
        constexpr int NI = (5 * NC + NSZ_ + NU) * 2 * 4 <= 248 ? 4 : 2;
        using KC = WvKernelCfg<NX, NU, NY, ND, NXP, DU, NG, NC, MASKED, NI>;
        constexpr size_t lds = KC::lds_bytes(64);      // the longest horizon; a launch asks for what its own horizon needs
        l.wv_lds = lds; l.wv_ws_per_inst = sizeof(double) * 64 * KC::Cfg::ROWS_WS;
        if (lds <= 160 * 1024) {
            // Instances per wave by the size of the batch: a wave is alone on its SIMD (512 registers), so a batch of fewer than NI waves
            // per SIMD leaves SIMDs empty while the others work through NI instances one after the other in the element-wise phases.
            // Up to one instance per SIMD: NI = 1; up to two: NI = 2 (the matrix-core passes cost the same for 1, 2 or 4 tiles).  The
            // arithmetic of an instance does not depend on its neighbours in the wave: same results bit for bit (tests/test_gpu_parity.py).
            l.loop_wv = [](const DevProblem *p, LoopArgs a, hipStream_t s) -> int {
                int dev = 0, cus = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return -1;
                const int simds = 4 * cus;
                int ni = a.wv_ni > 0 ? a.wv_ni : (a.B <= simds ? 1 : (a.B <= 2 * simds ? 2 : NI));
                if (ni > NI) ni = NI;
                if (ni == 3) ni = 2;
                // (the one- and two-instance variants cost build time: in the default library - seven dimension sets - only the two BASELINE problems carry
                // them; a library built for one dimension set always does)
#ifdef MPC_DEFAULT_DIM_LIST
                constexpr bool small = (NX == 3 && NU == 2 && NY == 3 && ND == 3 && !DU && NG == 0) || (NX == 4 && NU == 2 && NY == 2 && ND == 2 && NXP == 4 && DU && NG == 0);
#else
                constexpr bool small = true;
#endif
                if constexpr (small) {
                    if (ni == 1) return launch_loop_wv<NX, NU, NY, ND, NXP, DU, NG, NC, MASKED, 1>(p, a, s, dev);
                    if (ni == 2) return launch_loop_wv<NX, NU, NY, ND, NXP, DU, NG, NC, MASKED, 2>(p, a, s, dev);
                }
                return launch_loop_wv<NX, NU, NY, ND, NXP, DU, NG, NC, MASKED, NI>(p, a, s, dev);
            };
            l.wv_ns = NSZ_;
            l.ocp_wv = [](const DevProblem *p, OcpWvArgs a, int N, hipStream_t s) -> int {
                auto kern = ocp_kernel_wv<NX, NU, NY, ND, DU, NG, NC, MASKED, NI>;
                static bool attr_set[64] = {};
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
                if (!attr_set[dev]) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
                    attr_set[dev] = true;
                }
                using Cfg = typename KC::Cfg;
                const size_t bytes = sizeof(double) * ((size_t)Cfg::t_doubles(N) + NI * Cfg::QN + NI * Cfg::OUT) + sizeof(int) * 16;
                hipLaunchKernelGGL(kern, dim3((a.o.B + NI - 1) / NI), dim3(64), bytes, s, p, a);
                return 0;
            };
        }
    }
    l.nc = NC;
    return l;
}

template <int NX, int NU, int NY, int ND, int NXP, bool DU, int NG>
static Launchers make_launchers(int mode)
{
    constexpr int NS = NX + (DU ? NU : 0) + NG;
    if (NG == 0 && mode == kBoundsAllFinite) return make_launchers_mode<NX, NU, NY, ND, NXP, DU, NG, NS + NU, false>();
    if (NG == 0 && mode == kBoundsInputsOnly) return make_launchers_mode<NX, NU, NY, ND, NXP, DU, NG, NU, false>();
    return make_launchers_mode<NX, NU, NY, ND, NXP, DU, NG, NS + NU, true>();      // output-row states are free at the terminal stage: masks
}

#ifdef MPC_PART2
// second object of the default library: nothing but the kernels of its dimension sets
extern "C" int mpc_part2_launchers(int nx, int nu, int ny, int nd, int nxp, int du, int ng, int mode, void *out)
{
    bool found = false;
#define MPC_TRY_DIM(NX, NU, NY, ND, NXP, DU, NG)                                                              \
    if (!found && nx == NX && nu == NU && ny == NY && nd == ND && nxp == NXP && (du != 0) == (DU != 0) && ng == NG) { \
        *(Launchers *)out = make_launchers<NX, NU, NY, ND, NXP, (DU != 0), NG>(mode);                          \
        found = true;                                                                                         \
    }
    MPC_DIM_LIST(MPC_TRY_DIM)
#undef MPC_TRY_DIM
    return found ? 1 : 0;
}
#else
#ifdef MPC_HAVE_PART2
extern "C" int mpc_part2_launchers(int nx, int nu, int ny, int nd, int nxp, int du, int ng, int mode, void *out);
#endif

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    int ensure(size_t n)
    {
        if (n <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        HIP_TRY(hipMalloc(&p, n));
        bytes = n;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct mpc_handle {
    DevProblem hp;              // host copy
    DevProblem *dp = nullptr;   // device copy
    Launchers L;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false; int n_launches = 0;
    int steps_per_launch = 50;  // closed-loop steps per kernel launch (a launch starts with cold scalar / instruction caches)
    int loop_kernel_opt = 0;    // option "loop_kernel": 0 = choose by batch size, 1 = instance per lane, 2 = horizon-parallel
    int wv_ni_opt = 0;          // option "wave_instances": instances per wave of the wave-autonomous loop kernel (0 = by batch size, else 1, 2 or 4)
    int ws_mode = -1;           // which loop kernel's layout the workspace holds (-1 = none: next OCPs start cold)
    double h_sample = 1.0;      // sampling interval (mpc_lin_desc.h_sample): the time a user plant integrates over
    // per-call scratch (solve API)
    DevBuf scratch, ws;
    // loop state
    int B = 0; size_t Bs = 0; int max_steps = 0, log_level = 0, sched_steps = 0, last_k0 = 0, last_n = 0;
    bool state_set = false;     // mpc_loop_set_state has supplied the whole state since the last mpc_loop_alloc
    // per-call OCP on the wave-autonomous solver: data of the previous call (warm start), caller's guess, final trajectory
    int ocp_warm = 0, ocp_kernel_opt = 0, pc_B = 0;
    DevBuf pc_prev, pc_valid, pc_guess, pc_traj;
    // time-varying model parameters: horizon values of one mpc_ocp_solve call, its slab and workspace; this step's p_x_k / p_y_k for
    // mpc_target_solve / mpc_kf_update (mpc_set_model_offsets)
    DevBuf soft_ws, soft_sl, soft_keep; int soft_B = 0;      // soft output constraints: the arrowhead solver's workspace, the last call's optimal slacks
    DevBuf pxy_in, pxy_lin, pxy_ws, off_px, off_py; int off_B = 0; bool off_has_px = false, off_has_py = false;
    DevBuf msch; int msch_steps = 0; bool msch_px = false, msch_py = false;      // def_px / def_py over the horizon for every step of the fused loop
    DevBuf st_x, st_xhat, st_dhat, st_P, st_u, st_xs, st_us, st_flag, st_Kg, st_Pn, st_tw, sch, logs, logi;
    std::map<std::string, std::pair<size_t, int>> log_off;   // name -> (offset in doubles / ints, dim)
    // multi-GPU (one process per GPU): RCCL communicator over the ranks of the job, staging buffers of the collectives
    ncclComm_t comm = nullptr; int rank = 0, world = 1;
    DevBuf coll_send, coll_recv;
};

static size_t pad64(size_t b) { return (b + 63) / 64 * 64; }

// host [B][d] -> SoA staging [d][Bs]
static void to_soa(const double *src, int B, int d, size_t Bs, double *dst)
{
    for (int i = 0; i < d; i++) {
        double *row = dst + (size_t)i * Bs;
        for (int b = 0; b < B; b++) row[b] = src[(size_t)b * d + i];
        for (size_t b = B; b < Bs; b++) row[b] = 0.0;
    }
}
static void from_soa(const double *src, int B, int d, size_t Bs, double *dst)
{
    for (int i = 0; i < d; i++) {
        const double *row = src + (size_t)i * Bs;
        for (int b = 0; b < B; b++) dst[(size_t)b * d + i] = row[b];
    }
}

// rows carried as stage states of their own: bounded output rows that are not a multiple of one state (rows[] (optional) receives their indices), then the user
// inequality rows (rows[] = -1)
static int general_output_rows(const mpc_lin_desc *d, int *rows)
{
    int ng = 0;
    if (d->y_bounded && !d->slacks) {      // (soft output rows are rows of their own solver, not stage states)
        for (int i = 0; i < d->ny; i++) {
            int cnt = 0;
            for (int j = 0; j < d->nx; j++) if (d->C[i * d->nx + j] != 0.0) cnt++;
            if (cnt != 1 && (std::isfinite(d->ymin[i]) || std::isfinite(d->ymax[i]))) { if (rows && ng < kMaxY) rows[ng] = i; ng++; }
        }
    }
    for (int i = 0; i < d->n_user_rows; i++) { if (rows && ng < kMaxY) rows[ng] = -1; ng++; }
    return ng;
}

// Householder QR of [A-I, B]' and the reduced target problem
static int build_target(const mpc_lin_desc *d, DevProblem &P)
{
    const int n = d->nx, m = d->nu, q = d->ny, nv = n + m;
    double Qf[kMaxV][kMaxV], Rm[kMaxV][kMaxN];
    for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) Qf[i][j] = (i == j) ? 1.0 : 0.0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) Rm[j][i] = d->A[i * n + j] - (i == j ? 1.0 : 0.0);
        for (int j = 0; j < m; j++) Rm[n + j][i] = d->B[i * m + j];
    }
    for (int k = 0; k < n; k++) {
        double v[kMaxV], nrm = 0.0, vn = 0.0;
        for (int i = k; i < nv; i++) nrm += Rm[i][k] * Rm[i][k];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) return -1;
        const double alpha = Rm[k][k] > 0 ? -nrm : nrm;
        for (int i = 0; i < nv; i++) v[i] = i < k ? 0.0 : Rm[i][k];
        v[k] -= alpha;
        for (int i = k; i < nv; i++) vn += v[i] * v[i];
        if (vn > 0.0) {
            for (int j = 0; j < n; j++) { double s = 0.0; for (int i = k; i < nv; i++) s += v[i] * Rm[i][j]; s *= 2.0 / vn; for (int i = k; i < nv; i++) Rm[i][j] -= s * v[i]; }
            for (int j = 0; j < nv; j++) { double s = 0.0; for (int i = k; i < nv; i++) s += Qf[j][i] * v[i]; s *= 2.0 / vn; for (int i = k; i < nv; i++) Qf[j][i] -= s * v[i]; }
        }
    }
    double rmax = 0.0;
    for (int k = 0; k < n; k++) rmax = std::fmax(rmax, std::fabs(Rm[k][k]));
    for (int k = 0; k < n; k++) if (std::fabs(Rm[k][k]) < 1e-12 * rmax) return -1;
    double Rti[kMaxN][kMaxN];
    for (int c = 0; c < n; c++)
        for (int i = 0; i < n; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= Rm[j][i] * Rti[j][c];
            Rti[i][c] = s / Rm[i][i];
        }
    for (int r = 0; r < nv; r++) for (int c = 0; c < n; c++) { double s = 0.0; for (int j = 0; j < n; j++) s += Qf[r][j] * Rti[j][c]; P.Ep[r][c] = s; }
    for (int r = 0; r < nv; r++) for (int c = 0; c < m; c++) P.Zn[r][c] = Qf[r][n + c];
    for (int i = 0; i < q; i++) for (int c = 0; c < m; c++) { double s = 0.0; for (int j = 0; j < n; j++) s += d->C[i * n + j] * P.Zn[j][c]; P.CZx[i][c] = s; }
    for (int a = 0; a < m; a++) for (int b = 0; b < m; b++) {
        double s = 0.0;
        for (int i = 0; i < q; i++) for (int j = 0; j < q; j++) s += P.CZx[i][a] * d->Qss[i * q + j] * P.CZx[j][b];
        for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) s += P.Zn[n + i][a] * d->Rss[i * m + j] * P.Zn[n + j][b];
        P.Hr[a][b] = s;
    }
    for (int a = 0; a < m; a++) for (int b = 0; b < a; b++) { const double s = 0.5 * (P.Hr[a][b] + P.Hr[b][a]); P.Hr[a][b] = P.Hr[b][a] = s; }
    for (int r = 0; r < nv; r++) for (int c = 0; c < m; c++) P.W[r][c] = P.Zn[r][c];
    for (int r = 0; r < q; r++) for (int c = 0; c < m; c++) P.W[nv + r][c] = P.CZx[r][c];
    for (int i = 0; i < n; i++) { P.tlo[i] = d->xmin_ss[i]; P.thi[i] = d->xmax_ss[i]; }
    for (int i = 0; i < m; i++) { P.tlo[n + i] = d->umin_ss[i]; P.thi[n + i] = d->umax_ss[i]; }
    for (int i = 0; i < q; i++) { P.tlo[nv + i] = d->ymin_ss[i]; P.thi[nv + i] = d->ymax_ss[i]; }
    for (int i = 0; i < q; i++) for (int j = 0; j < q; j++) P.Qss[i][j] = d->Qss[i * q + j];
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) P.Rss[i][j] = d->Rss[i * m + j];
    // the reduced Hessian must be positive definite (unique target)
    if (m == 1) { if (!(P.Hr[0][0] > 0)) return -2; }
    else {
        double c[kMaxM][kMaxM];
        for (int i = 0; i < m; i++) for (int j = 0; j <= i; j++) {
            double a = P.Hr[i][j];
            for (int k = 0; k < j; k++) a -= c[i][k] * c[j][k];
            if (i == j) { if (!(a > 0)) return -2; c[i][i] = std::sqrt(a); } else c[i][j] = a / c[j][j];
        }
    }
    return 0;
}

// the g2 rows exist (DuFree False, Control_Calc.py:64-67)
static bool du_bounded(const mpc_lin_desc *d) { return d->Dumin != nullptr || d->Dumax != nullptr; }
// the stage state carries u_prev: cost on u_k - u_{k-1}, or bounds on it
static bool stage_has_uprev(const mpc_lin_desc *d) { return d->du_form != 0 || du_bounded(d); }

static int build_problem(const mpc_lin_desc *d, DevProblem &P)
{
    std::memset(&P, 0, sizeof(P));
    const int n0 = d->nx, m = d->nu, q = d->ny, nd = d->nd, nxp = d->nxp;
    P.nx = n0; P.nu = m; P.ny = q; P.nd = nd; P.nxp = nxp; P.N = d->N;
    P.du_form = d->du_form; P.duss_form = d->duss_form; P.y_bounded = d->slacks ? 0 : d->y_bounded; P.estimator = d->estimator;
    if (d->slacks) {
        if (!d->Ws) return fail(-1, "slacks = 1 needs the slack weight Ws [2 ny][2 ny]");
        if (!d->y_bounded) return fail(-2, "slacks = 1 without output bounds");
        if (d->term_cons) return fail(-8, "soft constraints together with a terminal equality are not carried");
        P.soft = 1;
        for (int i = 0; i < 2 * q; i++) for (int j = 0; j < 2 * q; j++) P.Ws[i][j] = d->Ws[i * 2 * q + j];
    }
    P.max_iter = d->max_iter > 0 ? d->max_iter : 100;
    for (int i = 0; i < n0; i++) {
        for (int j = 0; j < n0; j++) { P.A[i][j] = P.Am[i][j] = d->A[i * n0 + j]; P.Q[i][j] = d->Q[i * n0 + j]; P.Pf[i][j] = d->P[i * n0 + j]; }
        for (int j = 0; j < m; j++) P.B[i][j] = P.Bm[i][j] = d->B[i * m + j];
        for (int j = 0; j < nd; j++) P.Bd[i][j] = d->Bd[i * nd + j];
        P.fxc[i] = d->fx_const ? d->fx_const[i] : 0.0;
        P.zlo_m[i] = P.zlo_e[i] = d->xmin[i]; P.zhi_m[i] = P.zhi_e[i] = d->xmax[i];
    }
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < m; j++) P.R[i][j] = d->R[i * m + j];
        P.ulo[i] = d->umin[i]; P.uhi[i] = d->umax[i];
    }
    if (d->term_cons) {
        // Terminal equality x_N = xs (Control_Calc.py:197-198) through the terminal weight: rho (x_N - xs) is the multiplier of the
        // equality, the optimum is the constrained one up to |multiplier| / rho (mpc_device.hpp:term_missed decides "unreachable")
        double scale = 1.0;
        for (int i = 0; i < n0 * n0; i++) scale = std::fmax(scale, std::fabs(d->Q[i]));
        for (int i = 0; i < m * m; i++) scale = std::fmax(scale, std::fabs(d->R[i]));
        for (int i = 0; i < n0; i++) for (int j = 0; j < n0; j++) P.Pf[i][j] = i == j ? 1e12 * scale : 0.0;
        P.term_cons = 1; P.term_tol = 1e-6; P.term_gcap = 1e3 * scale;
        P.term_floor = 8.0 * 2.2e-16 * P.Pf[0][0];      // a few units in the last place of x_N (|x| of order one to ten) times the weight
    }
    if (du_bounded(d)) {
        // Bounds on u_k - u_{k-1} (g2 rows, Control_Calc.py:163-169,241-243): stage form with input v_k = u_k - u_{k-1} and state
        // z = [x; u_prev]:  z+ = [[A, B], [0, I]] z + [B; I] v.  The g2 rows are the box on the input, the u bounds a box on the
        // u_prev part of z_1..z_N (z_{k+1}'s is u_k), the cost on u - us couples z_u and v (M), the cost on u_k - u_{k-1} is v'Sv.
        P.in_is_du = 1; P.zr_us = d->du_form ? 0 : 1;
        for (int i = 0; i < m; i++) {
            P.ulo[i] = d->Dumin ? d->Dumin[i] : -INFINITY; P.uhi[i] = d->Dumax ? d->Dumax[i] : INFINITY;
            P.B[n0 + i][i] = 1.0; P.A[n0 + i][n0 + i] = 1.0;
            for (int r = 0; r < n0; r++) P.A[r][n0 + i] = d->B[r * m + i];
            if (!d->du_form) { for (int j = 0; j < m; j++) { P.Q[n0 + i][n0 + j] = d->R[i * m + j]; P.M[n0 + i][j] = d->R[i * m + j]; } }
            P.zlo_m[n0 + i] = P.zlo_e[n0 + i] = d->umin[i]; P.zhi_m[n0 + i] = P.zhi_e[n0 + i] = d->umax[i];
        }
    } else if (d->du_form) {   // z = [x; u_prev], input u  (Control_Calc.py:163-166,180-181)
        for (int i = 0; i < m; i++) {
            P.B[n0 + i][i] = 1.0;
            for (int j = 0; j < m; j++) { P.Q[n0 + i][n0 + j] = d->R[i * m + j]; P.M[n0 + i][j] = -d->R[i * m + j]; }
            P.zlo_m[n0 + i] = P.zlo_e[n0 + i] = -INFINITY; P.zhi_m[n0 + i] = P.zhi_e[n0 + i] = INFINITY;
        }
    }
    for (int i = 0; i < q; i++) {
        for (int j = 0; j < n0; j++) P.Cm[i][j] = d->C[i * n0 + j];
        for (int j = 0; j < nd; j++) P.Cd[i][j] = d->Cd[i * nd + j];
        for (int j = 0; j < nxp; j++) P.Cp[i][j] = d->Cp[i * nxp + j];
        P.fyc[i] = d->fy_const ? d->fy_const[i] : 0.0;
        P.ymin[i] = d->ymin[i]; P.ymax[i] = d->ymax[i];
        P.ymap_idx[i] = 0; P.ymap_scale[i] = 1.0;
    }
    for (int i = 0; i < nxp; i++) {
        for (int j = 0; j < nxp; j++) P.Ap[i][j] = d->Ap[i * nxp + j];
        for (int j = 0; j < m; j++) P.Bp[i][j] = d->Bp[i * m + j];
    }
    // Bounded output rows (Control_Calc.py:130,150-151,229-230).  A row with a single non-zero entry is a box on that state.
    // Any other row i gets a stage state of its own, w = C_i x, carried by w+ = C_i (A x + B u + c): the row becomes a box on w
    // at k = 1..N-1 (the terminal state has no output row); no cost on w.
    P.ng = general_output_rows(d, P.yg_row);
    if (d->n_user_rows) {
        // User inequality rows (Control_Calc.py:94-100,132-147): Gx x_k + Gu u_k + Gd dhat + g0 <= 0 at k = 0..N-1, each a stage state w+ = Gx x + Gu u + const with the box
        // (-inf, 0] at k = 1..N (terminal state included: the row of stage N-1).  With the input-move form (stage state [x; u_prev], input v = u - u_prev) u = u_prev + v.
        if (!d->Gx || !d->Gu || !d->g0 || (nd > 0 && !d->Gd)) return fail(-1, "n_user_rows = %d needs Gx, Gu, Gd, g0", d->n_user_rows);
        if (d->slacks || d->term_cons) return fail(-8, "user inequality rows together with slacks or a terminal equality are not carried");
        const int nb = n0 + (stage_has_uprev(d) ? m : 0);
        for (int g = P.ng - d->n_user_rows, i = 0; g < P.ng; g++, i++) {
            const int r = nb + g;
            for (int j = 0; j < n0; j++) P.A[r][j] = d->Gx[i * n0 + j];
            for (int j = 0; j < m; j++) { P.B[r][j] = d->Gu[i * m + j]; if (du_bounded(d)) P.A[r][n0 + j] = d->Gu[i * m + j]; }
            for (int j = 0; j < nd; j++) P.Bd[r][j] = d->Gd[i * nd + j];
            P.fxc[r] = d->g0[i];
            P.zlo_m[r] = P.zlo_e[r] = -INFINITY; P.zhi_m[r] = P.zhi_e[r] = 0.0;
        }
    }
    if (d->y_bounded && !d->slacks) {
        const int nb = n0 + (stage_has_uprev(d) ? m : 0);
        for (int i = 0; i < q; i++) {
            P.ymap_idx[i] = -1;       // unbounded rows and rows that are identically zero: nothing to map
            for (int j = 0; j < n0; j++) if (d->C[i * n0 + j] != 0.0) { P.ymap_idx[i] = j; P.ymap_scale[i] = d->C[i * n0 + j]; }
        }
        for (int g = 0; g < P.ng - d->n_user_rows; g++) {
            const int i = P.yg_row[g], r = nb + g;
            P.ymap_idx[i] = r; P.ymap_scale[i] = 1.0;
            for (int j = 0; j < n0; j++) { double acc = 0.0; for (int l = 0; l < n0; l++) acc += d->C[i * n0 + l] * d->A[l * n0 + j]; P.A[r][j] = acc; }
            for (int j = 0; j < m; j++) { double acc = 0.0; for (int l = 0; l < n0; l++) acc += d->C[i * n0 + l] * d->B[l * m + j]; P.B[r][j] = acc; }
            P.zlo_m[r] = P.zlo_e[r] = -INFINITY; P.zhi_m[r] = P.zhi_e[r] = INFINITY;
        }
    }
    for (int i = 0; i < kMaxN; i++) for (int j = 0; j < kMaxN; j++) P.Apow[0][i][j] = P.A[i][j];
    for (int e = 1; e < 6; e++)
        for (int i = 0; i < kMaxN; i++) for (int j = 0; j < kMaxN; j++) {
            double acc = 0.0;
            for (int l = 0; l < kMaxN; l++) acc += P.Apow[e - 1][i][l] * P.Apow[e - 1][l][j];
            P.Apow[e][i][j] = acc;
        }
    P.has_dsat = (d->dmin && d->dmax) ? 1 : 0;
    for (int i = 0; i < nd; i++) { P.dmin[i] = d->dmin ? d->dmin[i] : -INFINITY; P.dmax[i] = d->dmax ? d->dmax[i] : INFINITY; }
    const int ne = n0 + nd;
    for (int i = 0; i < ne; i++) P.Aa[i][i] = 1.0;
    for (int i = 0; i < n0; i++) { for (int j = 0; j < n0; j++) P.Aa[i][j] = d->A[i * n0 + j]; for (int j = 0; j < nd; j++) P.Aa[i][n0 + j] = d->Bd[i * nd + j]; }
    for (int i = 0; i < q; i++) { for (int j = 0; j < n0; j++) P.Ca[i][j] = d->C[i * n0 + j]; for (int j = 0; j < nd; j++) P.Ca[i][n0 + j] = d->Cd[i * nd + j]; }
    if (d->estimator == MPC_EST_KALMAN) {
        if (!d->Q_kf || !d->R_kf) return fail(-3, "MPC_EST_KALMAN needs Q_kf and R_kf");
        for (int i = 0; i < ne; i++) for (int j = 0; j < ne; j++) P.Qkf[i][j] = d->Q_kf[i * ne + j];
        for (int i = 0; i < q; i++) for (int j = 0; j < q; j++) P.Rkf[i][j] = d->R_kf[i * q + j];
    } else if (d->estimator == MPC_EST_FIXED_GAIN) {
        if (!d->K) return fail(-3, "MPC_EST_FIXED_GAIN needs K");
        for (int i = 0; i < ne; i++) for (int j = 0; j < q; j++) P.Kfix[i][j] = d->K[i * q + j];
    }
    const int rc = build_target(d, P);
    if (rc == -1) return fail(-4, "[A-I, B] is rank deficient: no steady state for arbitrary disturbances");
    if (rc == -2) return fail(-4, "reduced Hessian of the target problem is not positive definite");
    return 0;
}

// cheapest kernel variant the bounds allow: every bounded variable two-sided finite => no masks at all
static int bound_mode(const mpc_lin_desc *d)
{
    bool u_all = true, x_all = true, x_none = true;
    for (int i = 0; i < d->nu; i++) u_all = u_all && std::isfinite(d->umin[i]) && std::isfinite(d->umax[i]);
    for (int i = 0; i < d->nx; i++) {
        const bool lo = std::isfinite(d->xmin[i]), hi = std::isfinite(d->xmax[i]);
        x_all = x_all && lo && hi; x_none = x_none && !lo && !hi;
    }
    bool y_any = false;
    if (d->y_bounded && !d->slacks) for (int i = 0; i < d->ny; i++) y_any = y_any || std::isfinite(d->ymin[i]) || std::isfinite(d->ymax[i]);
    if (du_bounded(d) || d->n_user_rows) return kBoundsGeneric;      // (a user row's stage state is bounded from above only)
    if (u_all && x_all && !d->du_form) return kBoundsAllFinite;        // Delta-u form carries unbounded u_prev states
    if (u_all && x_none && !y_any) return kBoundsInputsOnly;
    return kBoundsGeneric;
}

extern "C" int mpc_lin_create(const mpc_lin_desc *d, mpc_handle **out)
{
    if (!d || !out) return fail(-1, "null argument");
    *out = nullptr;
    const int ns = d->nx + (stage_has_uprev(d) ? d->nu : 0) + general_output_rows(d, nullptr);
    if (d->nx < 1 || d->nu < 1 || ns > kMaxN || d->nu > kMaxM || d->ny > kMaxY || d->nd > kMaxD || d->nxp > kMaxN || d->N < 2 || d->N > 512)
        return fail(-2, "dimensions out of range (stage state <= %d, nu <= %d, ny <= %d, nd <= %d, 2 <= N <= 512)", kMaxN, kMaxM, kMaxY, kMaxD);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(-11, "no HIP device: the MI355X path is required (there is no CPU fallback)");
    if (d->device < 0 || d->device >= ndev) return fail(-11, "device %d out of range (have %d)", d->device, ndev);
    mpc_handle *h = new mpc_handle();
    bool found = false;
    const int ng = general_output_rows(d, nullptr);
#define MPC_TRY_DIM(NX, NU, NY, ND, NXP, DU, NG)                                                              \
    if (!found && d->nx == NX && d->nu == NU && d->ny == NY && d->nd == ND && d->nxp == NXP && stage_has_uprev(d) == (DU != 0) && ng == NG) { \
        h->L = make_launchers<NX, NU, NY, ND, NXP, (DU != 0), NG>(bound_mode(d));                             \
        found = true;                                                                                         \
    }
    MPC_DIM_LIST(MPC_TRY_DIM)
#undef MPC_TRY_DIM
#ifdef MPC_HAVE_PART2
    if (!found) found = mpc_part2_launchers(d->nx, d->nu, d->ny, d->nd, d->nxp, stage_has_uprev(d) ? 1 : 0, ng, bound_mode(d), &h->L) != 0;
#endif
    if (!found) {
        delete h;
        return fail(-5, "no kernel compiled for nx=%d nu=%d ny=%d nd=%d nxp=%d du_form=%d general_output_rows=%d (build info: %s)", d->nx, d->nu, d->ny, d->nd, d->nxp, (int)stage_has_uprev(d), ng, mpc_build_info());
    }
#ifdef MPC_NL_PLANT_HEADER
    if (!d->nl_plant) { delete h; return fail(-5, "this library simulates a user plant (User_fxp_Cont): the descriptor has nl_plant = 0"); }
#else
    if (d->nl_plant) { delete h; return fail(-5, "no kernel compiled for a user plant (nl_plant = 1): build the library with the plant's generated header (capi.Solver does)"); }
#endif
    int rc = build_problem(d, h->hp);
    if (rc != 0) { delete h; return rc; }
    h->h_sample = d->h_sample > 0.0 ? d->h_sample : 1.0;
    {      // |A^32| > 1e4: the open-loop simulation every start rests on amplifies the shift's small mismatch by that much over the horizon
        double nrm = 0.0;
        for (int i = 0; i < kMaxN; i++) for (int j = 0; j < kMaxN; j++) nrm = std::fmax(nrm, std::fabs(h->hp.Apow[5][i][j]));
        h->hp.no_warm = nrm > 1e4 ? 1 : 0;
    }
    h->device = d->device;
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e == hipSuccess) e = hipMalloc((void **)&h->dp, sizeof(DevProblem));
    if (e == hipSuccess) e = hipMemcpy(h->dp, &h->hp, sizeof(DevProblem), hipMemcpyHostToDevice);
    if (e != hipSuccess) { const int c = fail(-10, "device set-up failed: %s", hipGetErrorString(e)); mpc_destroy(h); return c; }
    *out = h;
    return 0;
}

extern "C" int mpc_comm_destroy(mpc_handle *h);

extern "C" void mpc_destroy(mpc_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)mpc_comm_destroy(h);
    h->coll_send.release(); h->coll_recv.release();
    h->pc_prev.release(); h->pc_valid.release(); h->pc_guess.release(); h->pc_traj.release();
    h->soft_ws.release(); h->soft_sl.release(); h->soft_keep.release();
    h->pxy_in.release(); h->pxy_lin.release(); h->pxy_ws.release(); h->off_px.release(); h->off_py.release(); h->msch.release();
    for (DevBuf *b : {&h->scratch, &h->ws, &h->st_x, &h->st_xhat, &h->st_dhat, &h->st_P, &h->st_u, &h->st_xs, &h->st_us, &h->st_flag, &h->st_Kg, &h->st_Pn, &h->st_tw, &h->sch, &h->logs, &h->logi}) b->release();
    if (h->dp) (void)hipFree(h->dp);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" const char *mpc_build_info(void)
{
    static std::string s;
    if (s.empty()) {
        s = "gfx950;loop_kernels=wave-autonomous(N<=64&ns<=8&nu<=2),horizon-parallel(N<=64;mfma-riccati:ns<=4&nu<=2,else-batch<=16384),instance-per-lane;dims(nx/nu/ny/nd/nxp/du/ng)=";
#define MPC_INFO_DIM(NX, NU, NY, ND, NXP, DU, NG) s += #NX "/" #NU "/" #NY "/" #ND "/" #NXP "/" #DU "/" #NG ",";
        MPC_DIM_LIST(MPC_INFO_DIM)
#ifdef MPC_HAVE_PART2
        MPC_DIM_LIST_B(MPC_INFO_DIM)
#endif
#undef MPC_INFO_DIM
        s.pop_back();
#ifdef MPC_NL_PLANT_HEADER
        s += ";nlplant";
#endif
    }
    return s.c_str();
}

extern "C" int mpc_set_option(mpc_handle *h, const char *name, double value)
{
    if (!h || !name) return fail(-1, "null argument");
    if (!std::strcmp(name, "steps_per_launch")) { h->steps_per_launch = value >= 1 ? (int)value : 1; return 0; }
    if (!std::strcmp(name, "loop_kernel")) {
        const int v = (int)value;
        if (v < 0 || v > 3) return fail(-1, "loop_kernel must be 0 (auto), 1 (instance per lane), 2 (horizon-parallel) or 3 (wave-autonomous)");
        if (v == 3 && (!h->L.loop_wv || h->hp.N > 64)) return fail(-8, "the wave-autonomous kernel needs N <= 64, stage state <= 8 and nu <= 2");
        if (v == 2 && (!h->L.loop_tp || h->hp.N > 64)) return fail(-8, "the horizon-parallel kernel needs N <= 64 and a problem that fits the LDS");
        h->loop_kernel_opt = v;
        return 0;
    }
    if (!std::strcmp(name, "wave_instances")) {
        const int v = (int)value;
        if (v != 0 && v != 1 && v != 2 && v != 4) return fail(-1, "wave_instances must be 0 (by batch size), 1, 2 or 4");
        h->wv_ni_opt = v;
        return 0;
    }
    if (!std::strcmp(name, "ocp_warm_start")) { h->ocp_warm = value != 0.0; h->pc_B = 0; return 0; }
    if (!std::strcmp(name, "ocp_kernel")) {
        const int v = (int)value;
        if (v != 0 && v != 1 && v != 3) return fail(-1, "ocp_kernel must be 0 (auto), 1 (instance per lane) or 3 (wave-autonomous)");
        if (v == 3 && (!h->L.ocp_wv || h->hp.N > 64)) return fail(-8, "the wave-autonomous solver needs N <= 64, stage state <= 8 and nu <= 2");
        h->ocp_kernel_opt = v;
        return 0;
    }
    return fail(-1, "unknown option '%s'", name);
}

static int loop_mode(const mpc_handle *h);
static bool ocp_uses_wave(const mpc_handle *h);

extern "C" int mpc_get_option(mpc_handle *h, const char *name, double *value)
{
    if (!h || !name || !value) return fail(-1, "null argument");
    if (!std::strcmp(name, "steps_per_launch")) { *value = h->steps_per_launch; return 0; }
    if (!std::strcmp(name, "loop_kernel")) { *value = loop_mode(h); return 0; }
    if (!std::strcmp(name, "wave_instances")) { *value = h->wv_ni_opt; return 0; }
    if (!std::strcmp(name, "ocp_warm_start")) { *value = h->ocp_warm; return 0; }
    if (!std::strcmp(name, "ocp_kernel")) { *value = ocp_uses_wave(h) ? 3 : 1; return 0; }
    return fail(-1, "unknown option '%s'", name);
}

extern "C" void *mpc_stream(mpc_handle *h) { return h ? (void *)h->stream : nullptr; }

extern "C" float mpc_last_kernel_ms(mpc_handle *h, int32_t *n_launches)
{
    if (!h || !h->timed) return -1.0f;
    float ms = -1.0f;
    (void)hipSetDevice(h->device);
    if (hipEventSynchronize(h->ev1) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.0f;
    if (n_launches) *n_launches = h->n_launches;
    return ms;
}

// ---------------------------------------------------------------------------------------------------
// per-call solvers with host buffers
// ---------------------------------------------------------------------------------------------------
static int ensure_ws(mpc_handle *h, size_t Bs)
{
    const size_t lane_bytes = (size_t)h->L.ws_rows * (h->hp.N + 2) * Bs * sizeof(double);   // +2 guard blocks per wave
    const size_t groups = (Bs + h->L.tp_ni - 1) / h->L.tp_ni;
    const size_t tp_bytes = groups * h->L.tp_ni * h->L.tp_ws_per_inst + groups * h->L.tp_ws_per_group;
    const size_t wv_bytes = ((Bs + 3) / 4 * 4) * h->L.wv_ws_per_inst;
    return h->ws.ensure(std::max(std::max(lane_bytes, tp_bytes), wv_bytes));
}

// Which solver mpc_ocp_solve runs on: the wave-autonomous one where it exists, except (auto) for models whose open-loop response
// over the horizon is violently unstable (|A^32| > 1e4, e.g. Ex_LMPC_nlplant with |eig A|^50 ~ 1e12): its initial trajectory and
// costates come from lane scans with A^(2^e), which lose about three more digits there than the lane kernel's sequential sweeps
// (measured: 1e-7 relative on u* instead of 3e-9).
static bool ocp_uses_wave(const mpc_handle *h)
{
    if (!h->L.ocp_wv || h->hp.N > 64 || h->ocp_kernel_opt == 1) return false;
    if (h->ocp_kernel_opt == 3) return true;
    double nrm = 0.0;
    for (int i = 0; i < kMaxN; i++) for (int j = 0; j < kMaxN; j++) nrm = std::fmax(nrm, std::fabs(h->hp.Apow[5][i][j]));
    return nrm <= 1e4;
}

extern "C" int mpc_ocp_solve(mpc_handle *h, int32_t B, const double *xhat, const double *xs, const double *us,
                             const double *dhat, const double *u_prev, const double *px, const double *py,
                             double *w_inout, double *u_out, double *xnext_out, int32_t *status, int32_t *iters,
                             double *kkt_res)
{
    if (!h || B < 1 || !xhat || !xs || !us || !u_prev || !u_out || !xnext_out || !status) return fail(-1, "null argument");
    const DevProblem &P = h->hp;
    if (P.nd > 0 && !dhat) return fail(-1, "dhat is required when nd > 0");
    HIP_TRY(hipSetDevice(h->device));
    const size_t Bs = pad64(B);
    const int nx = P.nx, nu = P.nu, nd = P.nd;
    double *const w_out = w_inout;
    // layout of the scratch buffer (doubles): in: xhat xs us dhat u_prev | out: u x1 res | ints: status iters
    const size_t n_in = (size_t)(2 * nx + 2 * nu + nd) * Bs, n_out = (size_t)(nu + nx + 3) * Bs;
    const size_t bytes = (n_in + n_out) * sizeof(double) + 2 * Bs * sizeof(int32_t);
    if (h->scratch.ensure(bytes)) return -10;
    if (ensure_ws(h, Bs)) return -10;
    h->ws_mode = -1;            // this solve overwrites the workspace a resident loop may have been warm-starting from
    std::vector<double> stage(n_in + n_out);
    double *sp = stage.data();
    to_soa(xhat, B, nx, Bs, sp); to_soa(xs, B, nx, Bs, sp + (size_t)nx * Bs); to_soa(us, B, nu, Bs, sp + (size_t)2 * nx * Bs);
    if (nd) to_soa(dhat, B, nd, Bs, sp + (size_t)(2 * nx + nu) * Bs);
    to_soa(u_prev, B, nu, Bs, sp + (size_t)(2 * nx + nu + nd) * Bs);
    double *d = (double *)h->scratch.p;
    HIP_TRY(hipMemcpyAsync(d, sp, n_in * sizeof(double), hipMemcpyHostToDevice, h->stream));
    OcpArgs a;
    a.xhat = d; a.xs = d + (size_t)nx * Bs; a.us = d + (size_t)2 * nx * Bs; a.dhat = d + (size_t)(2 * nx + nu) * Bs;
    a.u_prev = d + (size_t)(2 * nx + nu + nd) * Bs;
    a.u_out = d + n_in; a.xnext_out = a.u_out + (size_t)nu * Bs; a.res = a.xnext_out + (size_t)nx * Bs;
    a.status = (int32_t *)(d + n_in + n_out); a.iters = a.status + Bs;
    a.ws = (double *)h->ws.p; a.B = B; a.Bs = Bs;
    HIP_TRY(hipMemsetAsync(a.u_out, 0, n_out * sizeof(double), h->stream));
    const bool pxy = px || py;
    if (P.soft) {
        // soft output constraints: the arrowhead solver, one instance per lane (mpc_soft.hpp)
        if (pxy) return fail(-8, "soft constraints with horizon parameters (px / py) are not carried");
        const int ny = P.ny;
        if (h->soft_ws.ensure((size_t)(Bs / 64) * h->L.soft_fields * P.N * 64 * sizeof(double)) || h->soft_sl.ensure((size_t)2 * ny * Bs * sizeof(double))) return -10;
        OcpSoftArgs sa;
        sa.o = a; sa.o.ws = (double *)h->soft_ws.p; sa.sl_out = (double *)h->soft_sl.p;
        HIP_TRY(hipEventRecord(h->ev0, h->stream));
        h->L.ocp_soft(h->dp, sa, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(h->ev1, h->stream));
        h->timed = true; h->n_launches = 1; h->soft_B = B;
        HIP_TRY(hipMemcpyAsync(sp + n_in, a.u_out, n_out * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        std::vector<int32_t> sti(2 * Bs);
        HIP_TRY(hipMemcpyAsync(sti.data(), a.status, 2 * Bs * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        const double *uo = sp + n_in, *xo = uo + (size_t)nu * Bs, *ro = xo + (size_t)nx * Bs;
        for (int b = 0; b < B; b++) {
            status[b] = sti[b];
            if (iters) iters[b] = sti[Bs + b];
            if (sti[b] != MPC_STATUS_INFEASIBLE) {
                for (int i = 0; i < nu; i++) u_out[(size_t)b * nu + i] = uo[(size_t)i * Bs + b];
                for (int i = 0; i < nx; i++) xnext_out[(size_t)b * nx + i] = xo[(size_t)i * Bs + b];
            }
            if (kkt_res) for (int i = 0; i < 3; i++) kkt_res[(size_t)b * 3 + i] = ro[(size_t)i * Bs + b];
        }
        return 0;
    }
    const bool wave = !pxy && ocp_uses_wave(h);
    const int ns_w = h->L.wv_ns, N = P.N;
    std::vector<double> guess;
    if (pxy) {
        // horizon values [B][N][dim] -> [N][dim][Bs]; the lane solver with per-block affine terms and output boxes
        const int ny = P.ny;
        const size_t npx = px ? (size_t)N * nx * Bs : 0, npy = py ? (size_t)N * ny * Bs : 0;
        std::vector<double> hv(npx + npy, 0.0);
        for (int b = 0; b < B; b++)
            for (int k = 0; k < N; k++) {
                if (px) for (int i = 0; i < nx; i++) hv[((size_t)k * nx + i) * Bs + b] = px[((size_t)b * N + k) * nx + i];
                if (py) for (int i = 0; i < ny; i++) hv[npx + ((size_t)k * ny + i) * Bs + b] = py[((size_t)b * N + k) * ny + i];
            }
        if (h->pxy_in.ensure(hv.size() * sizeof(double)) || h->pxy_lin.ensure((size_t)N * h->L.pxy_lin * Bs * sizeof(double)) ||
            h->pxy_ws.ensure((size_t)h->L.pxy_ws_rows * (N + 2) * Bs * sizeof(double)))
            return -10;
        HIP_TRY(hipMemcpyAsync(h->pxy_in.p, hv.data(), hv.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));      // hv leaves scope before the kernel would otherwise read the copy
        OcpPxyArgs wa;
        wa.o = a; wa.o.ws = (double *)h->pxy_ws.p;
        wa.px = px ? (const double *)h->pxy_in.p : nullptr; wa.py = py ? (const double *)h->pxy_in.p + npx : nullptr;
        wa.lin = (double *)h->pxy_lin.p;
        HIP_TRY(hipEventRecord(h->ev0, h->stream));
        h->L.ocp_pxy(h->dp, wa, h->stream);
    } else if (wave) {
        // resident data of the previous call (warm start): reset when the batch changes
        const size_t nprev = (size_t)(2 * nx + nd + nu) * Bs;
        if (h->pc_prev.ensure(nprev * sizeof(double)) || h->pc_valid.ensure(Bs * sizeof(int32_t))) return -10;
        if (h->pc_B != B) {
            HIP_TRY(hipMemsetAsync(h->pc_prev.p, 0, nprev * sizeof(double), h->stream));
            HIP_TRY(hipMemsetAsync(h->pc_valid.p, 0, Bs * sizeof(int32_t), h->stream));
            h->pc_B = B;
        }
        OcpWvArgs wa;
        wa.o = a; wa.prev = (double *)h->pc_prev.p; wa.valid = (int32_t *)h->pc_valid.p; wa.warm_on = h->ocp_warm;
        wa.u_guess = nullptr; wa.traj = nullptr;
        const int nxu = nx + nu, nw = nx * (N + 1) + nu * N;
        if (w_out && h->ocp_warm && std::isfinite(w_out[0])) {
            // the caller's guess (MPC_code.py:740-764 hands the shifted previous optimum): its inputs, as the solver's own input
            // variable (u, or v = u_k - u_{k-1} when bounds on it exist), lane = stage
            guess.assign((size_t)Bs * nu * 64, 0.0);
            for (int b = 0; b < B; b++) {
                const double *w = w_out + (size_t)b * nw;
                for (int k = 0; k < N; k++)
                    for (int i = 0; i < nu; i++) {
                        const double uk = w[k * nxu + nx + i], um = k > 0 ? w[(k - 1) * nxu + nx + i] : u_prev[(size_t)b * nu + i];
                        guess[((size_t)b * nu + i) * 64 + k] = P.in_is_du ? uk - um : uk;
                    }
            }
            if (h->pc_guess.ensure(guess.size() * sizeof(double))) return -10;
            HIP_TRY(hipMemcpyAsync(h->pc_guess.p, guess.data(), guess.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
            wa.u_guess = (const double *)h->pc_guess.p;
        }
        if (w_out) {
            if (h->pc_traj.ensure((size_t)Bs * (nu + ns_w) * 64 * sizeof(double))) return -10;
            wa.traj = (double *)h->pc_traj.p;
        }
        HIP_TRY(hipEventRecord(h->ev0, h->stream));
        if (h->L.ocp_wv(h->dp, wa, N, h->stream)) return fail(-9, "cannot configure the wave-autonomous solver");
    } else {
        HIP_TRY(hipEventRecord(h->ev0, h->stream));
        h->L.ocp(h->dp, a, h->stream);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->timed = true; h->n_launches = 1;
    std::vector<int32_t> ist(2 * Bs);
    HIP_TRY(hipMemcpyAsync(sp + n_in, a.u_out, n_out * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(ist.data(), a.status, 2 * Bs * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const double *o = sp + n_in;
    for (int b = 0; b < B; b++) {
        status[b] = ist[b];
        if (iters) iters[b] = ist[Bs + b];
        if (ist[b] != kInfeasible) {
            for (int i = 0; i < nu; i++) u_out[(size_t)b * nu + i] = o[(size_t)i * Bs + b];
            for (int i = 0; i < nx; i++) xnext_out[(size_t)b * nx + i] = o[(size_t)(nu + i) * Bs + b];
        }
        if (kkt_res) for (int i = 0; i < 3; i++) kkt_res[(size_t)b * 3 + i] = o[(size_t)(nu + nx + i) * Bs + b];
    }
    if (w_out && wave) {   // primal trajectory in opt_dyn's order [x0,u0,...,xN] (Control_Calc.py:31-37) from the final iterate's rows
        const int nxu = nx + nu, nrow = nu + ns_w;
        std::vector<double> tr((size_t)Bs * nrow * 64);
        HIP_TRY(hipMemcpy(tr.data(), h->pc_traj.p, tr.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int b = 0; b < B; b++) {
            if (ist[b] == kInfeasible) continue;
            double *w = w_out + (size_t)b * (nx * (N + 1) + nu * N);
            const double *r = tr.data() + (size_t)b * nrow * 64;
            for (int i = 0; i < nx; i++) w[i] = xhat[(size_t)b * nx + i];
            for (int k = 0; k < N; k++) {
                for (int i = 0; i < nu; i++) w[k * nxu + nx + i] = P.in_is_du ? r[(nu + nx + i) * 64 + k] : r[i * 64 + k];
                for (int i = 0; i < nx; i++) w[(k + 1) * nxu + i] = r[(nu + i) * 64 + k];
            }
        }
    } else if (w_out) {   // the same from the lane kernel's workspace
        const int ns = nx + ((P.du_form || P.in_is_du) ? nu : 0), nv = ns + nu, nxu = nx + nu;
        const int ws_rows = pxy ? h->L.pxy_ws_rows : h->L.ws_rows, nc = pxy ? h->L.pxy_nc : h->L.nc;
        const int slots = ws_rows / 2, slotU = 4 * nc, slotZ = slotU + (nu + 1) / 2; (void)nv;
        std::vector<double> wsh((size_t)ws_rows * (N + 2) * Bs);
        HIP_TRY(hipMemcpy(wsh.data(), pxy ? h->pxy_ws.p : h->ws.p, wsh.size() * sizeof(double), hipMemcpyDeviceToHost));
        auto at = [&](int b, int k, int slot, int comp) { return wsh[((((size_t)(b / 64) * (N + 2) + k + 1) * slots + slot) * 64 + (b % 64)) * 2 + comp]; };
        for (int b = 0; b < B; b++) {
            if (ist[b] == kInfeasible) continue;
            double *w = w_out + (size_t)b * (nx * (N + 1) + nu * N);
            for (int i = 0; i < nx; i++) w[i] = xhat[(size_t)b * nx + i];
            for (int k = 0; k < N; k++) {
                for (int i = 0; i < nu; i++) w[k * nxu + nx + i] = P.in_is_du ? at(b, k, slotZ + (nx + i) / 2, (nx + i) % 2) : at(b, k, slotU + i / 2, i % 2);
                for (int i = 0; i < nx; i++) w[(k + 1) * nxu + i] = at(b, k, slotZ + i / 2, i % 2);
            }
        }
    }
    return 0;
}

extern "C" int mpc_get_slacks(mpc_handle *h, int32_t B, double *sl_out)
{
    if (!h || !sl_out) return fail(-1, "null argument");
    if (!h->hp.soft) return fail(-8, "this problem has no soft constraints");
    if (B != h->soft_B || B < 1) return fail(-1, "mpc_get_slacks: the last mpc_ocp_solve call had %d instances, not %d", h->soft_B, B);
    HIP_TRY(hipSetDevice(h->device));
    const size_t Bs = pad64(B);
    const int ns = 2 * h->hp.ny;
    std::vector<double> st((size_t)ns * Bs);
    HIP_TRY(hipMemcpy(st.data(), h->soft_sl.p, st.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++) for (int i = 0; i < ns; i++) sl_out[(size_t)b * ns + i] = st[(size_t)i * Bs + b];
    return 0;
}

extern "C" int mpc_target_solve(mpc_handle *h, int32_t B, const double *usp, const double *ysp, const double *xsp,
                                const double *dhat, const double *us_prev, double *xs, double *us, double *ys,
                                int32_t *status, int32_t *iters)
{
    (void)xsp;   // the reference passes xsp but Fss_obj never uses dx for matrix-defined costs (Utilities.py:299-313)
    if (!h || B < 1 || !usp || !ysp || !us_prev || !xs || !us || !ys || !status) return fail(-1, "null argument");
    const DevProblem &P = h->hp;
    if (P.nd > 0 && !dhat) return fail(-1, "dhat is required when nd > 0");
    HIP_TRY(hipSetDevice(h->device));
    const size_t Bs = pad64(B);
    const int nx = P.nx, nu = P.nu, ny = P.ny, nd = P.nd;
    const size_t n_in = (size_t)(2 * nu + ny + nd) * Bs, n_out = (size_t)(nx + nu + ny) * Bs;
    if (h->scratch.ensure((n_in + n_out) * sizeof(double) + 2 * Bs * sizeof(int32_t))) return -10;
    std::vector<double> stage(n_in + n_out);
    double *sp = stage.data();
    to_soa(usp, B, nu, Bs, sp); to_soa(ysp, B, ny, Bs, sp + (size_t)nu * Bs);
    if (nd) to_soa(dhat, B, nd, Bs, sp + (size_t)(nu + ny) * Bs);
    to_soa(us_prev, B, nu, Bs, sp + (size_t)(nu + ny + nd) * Bs);
    double *d = (double *)h->scratch.p;
    HIP_TRY(hipMemcpyAsync(d, sp, n_in * sizeof(double), hipMemcpyHostToDevice, h->stream));
    TargetArgs a;
    a.usp = d; a.ysp = d + (size_t)nu * Bs; a.dhat = d + (size_t)(nu + ny) * Bs; a.us_prev = d + (size_t)(nu + ny + nd) * Bs;
    a.xs = d + n_in; a.us = a.xs + (size_t)nx * Bs; a.ys = a.us + (size_t)nu * Bs;
    a.status = (int32_t *)(d + n_in + n_out); a.iters = a.status + Bs; a.B = B; a.Bs = Bs;
    if ((h->off_has_px || h->off_has_py) && h->off_B != B) return fail(-1, "mpc_set_model_offsets was called for a batch of %d, this call has %d", h->off_B, B);
    a.px0 = h->off_has_px ? (const double *)h->off_px.p : nullptr; a.py0 = h->off_has_py ? (const double *)h->off_py.p : nullptr;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    h->L.target(h->dp, a, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->timed = true; h->n_launches = 1;
    std::vector<int32_t> ist(2 * Bs);
    HIP_TRY(hipMemcpyAsync(sp + n_in, a.xs, n_out * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(ist.data(), a.status, 2 * Bs * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    from_soa(sp + n_in, B, nx, Bs, xs); from_soa(sp + n_in + (size_t)nx * Bs, B, nu, Bs, us);
    from_soa(sp + n_in + (size_t)(nx + nu) * Bs, B, ny, Bs, ys);
    for (int b = 0; b < B; b++) { status[b] = ist[b]; if (iters) iters[b] = ist[Bs + b]; }
    return 0;
}

extern "C" int mpc_set_model_offsets(mpc_handle *h, int32_t B, const double *px0, const double *py0)
{
    if (!h || B < 1) return fail(-1, "bad argument");
    const DevProblem &P = h->hp;
    HIP_TRY(hipSetDevice(h->device));
    const size_t Bs = pad64(B);
    h->off_has_px = h->off_has_py = false; h->off_B = B;
    std::vector<double> st;
    if (px0) {
        st.assign((size_t)P.nx * Bs, 0.0); to_soa(px0, B, P.nx, Bs, st.data());
        if (h->off_px.ensure(st.size() * sizeof(double))) return -10;
        HIP_TRY(hipMemcpy(h->off_px.p, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice));
        h->off_has_px = true;
    }
    if (py0) {
        st.assign((size_t)P.ny * Bs, 0.0); to_soa(py0, B, P.ny, Bs, st.data());
        if (h->off_py.ensure(st.size() * sizeof(double))) return -10;
        HIP_TRY(hipMemcpy(h->off_py.p, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice));
        h->off_has_py = true;
    }
    return 0;
}

extern "C" int mpc_kf_update(mpc_handle *h, int32_t B, const double *y, double *xi, double *Pk)
{
    if (!h || B < 1 || !y || !xi) return fail(-1, "null argument");
    const DevProblem &P = h->hp;
    if (P.estimator == MPC_EST_NONE) return fail(-7, "the problem has no estimator");
    if (P.estimator == MPC_EST_KALMAN && !Pk) return fail(-1, "P is required for MPC_EST_KALMAN");
    HIP_TRY(hipSetDevice(h->device));
    const size_t Bs = pad64(B);
    const int ne = P.nx + P.nd, ny = P.ny;
    const bool kal = P.estimator == MPC_EST_KALMAN;
    const size_t n_all = (size_t)(ny + ne + (kal ? ne * ne : 0)) * Bs;
    if (h->scratch.ensure(n_all * sizeof(double))) return -10;
    std::vector<double> stage(n_all);
    double *sp = stage.data();
    to_soa(y, B, ny, Bs, sp); to_soa(xi, B, ne, Bs, sp + (size_t)ny * Bs);
    if (kal) to_soa(Pk, B, ne * ne, Bs, sp + (size_t)(ny + ne) * Bs);
    double *d = (double *)h->scratch.p;
    HIP_TRY(hipMemcpyAsync(d, sp, n_all * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->off_has_py && h->off_B != B) return fail(-1, "mpc_set_model_offsets was called for a batch of %d, this call has %d", h->off_B, B);
    KfArgs a{d, h->off_has_py ? (const double *)h->off_py.p : nullptr, d + (size_t)ny * Bs, d + (size_t)(ny + ne) * Bs, B, Bs};
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    h->L.kf(h->dp, a, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->timed = true; h->n_launches = 1;
    HIP_TRY(hipMemcpyAsync(sp, d, n_all * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    from_soa(sp + (size_t)ny * Bs, B, ne, Bs, xi);
    if (kal) from_soa(sp + (size_t)(ny + ne) * Bs, B, ne * ne, Bs, Pk);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// resident closed loop
// ---------------------------------------------------------------------------------------------------
static const char *kLogD[] = {"U", "X_HAT", "XS", "US", "YS", "Xp", "D_HAT"};
static const char *kLogI[] = {"STATUS_DYN", "STATUS_SS", "ITERS_DYN", "ITERS_SS"};

extern "C" int mpc_loop_alloc(mpc_handle *h, int32_t B, int32_t max_steps, int32_t log_level)
{
    if (!h || B < 1 || max_steps < 1) return fail(-1, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const DevProblem &P = h->hp;
    const size_t Bs = pad64(B);
    const int ne = P.nx + P.nd;
    if (h->st_x.ensure((size_t)P.nxp * Bs * 8) || h->st_xhat.ensure((size_t)P.nx * Bs * 8) || h->st_dhat.ensure((size_t)(P.nd ? P.nd : 1) * Bs * 8) ||
        h->st_P.ensure((size_t)ne * ne * Bs * 8) || h->st_u.ensure((size_t)P.nu * Bs * 8) || h->st_xs.ensure((size_t)P.nx * Bs * 8) ||
        h->st_us.ensure((size_t)P.nu * Bs * 8) || h->st_flag.ensure(3 * Bs * 4) ||
        h->st_Kg.ensure((size_t)ne * P.ny * Bs * 8) || h->st_Pn.ensure((size_t)ne * ne * Bs * 8) ||
        h->st_tw.ensure((size_t)(2 * P.nu + 3 * (P.nx + P.nu + P.ny)) * Bs * 8))
        return -10;
    // [0,Bs): OCP warm start valid, [Bs,2Bs): filter look-ahead valid, [2Bs,3Bs): target warm start valid.  All resident state is
    // cleared on the handle's (non-blocking) stream: nothing a kernel reads is ever uninitialised, whatever the caller sets later
    HIP_TRY(hipMemsetAsync(h->st_flag.p, 0, 3 * Bs * 4, h->stream));
    for (DevBuf *b : {&h->st_x, &h->st_xhat, &h->st_dhat, &h->st_P, &h->st_u, &h->st_xs, &h->st_us, &h->st_Kg, &h->st_Pn, &h->st_tw})
        HIP_TRY(hipMemsetAsync(b->p, 0, b->bytes, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->state_set = false; h->msch_steps = 0; h->msch_px = h->msch_py = false;
    if (ensure_ws(h, Bs)) return -10;
    const int sdim = P.ny + P.nu + P.nxp + P.ny;   // ysp usp pxp pyp
    if (h->sch.ensure((size_t)max_steps * sdim * 8)) return -10;
    h->log_off.clear();
    size_t off = 0;
    const int dims[] = {P.nu, P.nx, P.nx, P.nu, P.ny, P.nxp, P.nd};
    for (int i = 0; i < 7; i++) {
        const bool on = log_level >= MPC_LOG_ALL || (log_level >= MPC_LOG_U && i == 0);
        if (on && dims[i] > 0) { h->log_off[kLogD[i]] = {off, dims[i]}; off += (size_t)max_steps * dims[i] * Bs; }
    }
    if (P.soft) {      // soft output constraints: the optimal slack vector of every step (log "SL", with the inputs), and the arrowhead solver's workspace
        if (log_level >= MPC_LOG_U) { h->log_off["SL"] = {off, 2 * P.ny}; off += (size_t)max_steps * 2 * P.ny * Bs; }
        if (h->soft_ws.ensure((size_t)(Bs / 64) * h->L.soft_fields * P.N * 64 * sizeof(double)) || h->soft_keep.ensure((size_t)2 * P.ny * Bs * sizeof(double))) return -10;
        HIP_TRY(hipMemsetAsync(h->soft_keep.p, 0, h->soft_keep.bytes, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (h->logs.ensure(off ? off * 8 : 8)) return -10;
    if (log_level >= MPC_LOG_U) {
        for (int i = 0; i < 4; i++) h->log_off[kLogI[i]] = {(size_t)i * max_steps * Bs, 0};
        if (h->logi.ensure((size_t)4 * max_steps * Bs * 4)) return -10;
    }
    h->B = B; h->Bs = Bs; h->max_steps = max_steps; h->log_level = log_level; h->sched_steps = 0;
    return 0;
}

static int up_state(mpc_handle *h, DevBuf &buf, const double *src, int d)
{
    if (!src || d == 0) return 0;
    std::vector<double> st((size_t)d * h->Bs);
    to_soa(src, h->B, d, h->Bs, st.data());
    HIP_TRY(hipMemcpyAsync(buf.p, st.data(), st.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}
static int down_state(mpc_handle *h, DevBuf &buf, double *dst, int d)
{
    if (!dst || d == 0) return 0;
    std::vector<double> st((size_t)d * h->Bs);
    HIP_TRY(hipMemcpyAsync(st.data(), buf.p, st.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    from_soa(st.data(), h->B, d, h->Bs, dst);
    return 0;
}

extern "C" int mpc_loop_set_state(mpc_handle *h, const double *x_p, const double *xhat, const double *dhat,
                                  const double *Pk, const double *u, const double *xs, const double *us)
{
    if (!h || h->B == 0) return fail(-1, "mpc_loop_alloc first");
    HIP_TRY(hipSetDevice(h->device));
    const DevProblem &P = h->hp;
    const int ne = P.nx + P.nd;
    if (!h->state_set) {
        // first call after mpc_loop_alloc: the whole state, nothing is guessed (the reference falls back to zeros when P0 is
        // absent, MPC_code.py:455-458: pass zeros explicitly).  Later calls may update a subset (NULL = keep).
        if (!x_p || !xhat || !u || !xs || !us) return fail(-1, "the first mpc_loop_set_state after mpc_loop_alloc needs x_p, xhat, u, xs and us");
        if (P.nd > 0 && !dhat) return fail(-1, "the first mpc_loop_set_state after mpc_loop_alloc needs dhat (nd = %d)", P.nd);
        if (P.estimator == MPC_EST_KALMAN && !Pk) return fail(-1, "the first mpc_loop_set_state after mpc_loop_alloc needs the covariance P (MPC_EST_KALMAN)");
    }
    int rc = 0;
    rc |= up_state(h, h->st_x, x_p, P.nxp); rc |= up_state(h, h->st_xhat, xhat, P.nx); rc |= up_state(h, h->st_dhat, dhat, P.nd);
    rc |= up_state(h, h->st_P, Pk, ne * ne); rc |= up_state(h, h->st_u, u, P.nu); rc |= up_state(h, h->st_xs, xs, P.nx);
    rc |= up_state(h, h->st_us, us, P.nu);
    // a new state invalidates the warm start: the next OCP of every instance starts cold (and no slack vector has been accepted yet)
    HIP_TRY(hipMemsetAsync(h->st_flag.p, 0, 3 * h->Bs * 4, h->stream));
    if (P.soft && h->soft_keep.p) HIP_TRY(hipMemsetAsync(h->soft_keep.p, 0, h->soft_keep.bytes, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (rc) return -10;
    h->state_set = true;
    return 0;
}

extern "C" int mpc_loop_get_state(mpc_handle *h, double *x_p, double *xhat, double *dhat, double *Pk, double *u,
                                  double *xs, double *us)
{
    if (!h || h->B == 0) return fail(-1, "mpc_loop_alloc first");
    HIP_TRY(hipSetDevice(h->device));
    const DevProblem &P = h->hp;
    const int ne = P.nx + P.nd;
    int rc = 0;
    rc |= down_state(h, h->st_x, x_p, P.nxp); rc |= down_state(h, h->st_xhat, xhat, P.nx); rc |= down_state(h, h->st_dhat, dhat, P.nd);
    rc |= down_state(h, h->st_P, Pk, ne * ne); rc |= down_state(h, h->st_u, u, P.nu); rc |= down_state(h, h->st_xs, xs, P.nx);
    rc |= down_state(h, h->st_us, us, P.nu);
    return rc ? -10 : 0;
}

extern "C" int mpc_loop_set_schedule(mpc_handle *h, int32_t nsteps, const double *ysp, const double *usp,
                                     const double *xsp, const double *pxp, const double *pyp)
{
    (void)xsp;
    if (!h || h->B == 0) return fail(-1, "mpc_loop_alloc first");
    if (nsteps < 1 || nsteps > h->max_steps) return fail(-1, "nsteps %d exceeds the allocated %d", nsteps, h->max_steps);
    if (!ysp || !usp) return fail(-1, "ysp and usp are required");
    HIP_TRY(hipSetDevice(h->device));
    const DevProblem &P = h->hp;
    const size_t ms = h->max_steps;
    std::vector<double> st(ms * (P.ny + P.nu + P.nxp + P.ny), 0.0);
    double *a = st.data(), *b = a + ms * P.ny, *c = b + ms * P.nu, *e = c + ms * P.nxp;
    std::memcpy(a, ysp, sizeof(double) * nsteps * P.ny);
    std::memcpy(b, usp, sizeof(double) * nsteps * P.nu);
    if (pxp) std::memcpy(c, pxp, sizeof(double) * nsteps * P.nxp);
    if (pyp) std::memcpy(e, pyp, sizeof(double) * nsteps * P.ny);
    HIP_TRY(hipMemcpy(h->sch.p, st.data(), st.size() * 8, hipMemcpyHostToDevice));
    h->sched_steps = nsteps;
    return 0;
}

// def_px / def_py for the fused loop (MPC_code.py:492-497): px [nsteps][N][nx] with px[k][i] = def_px(t_k + i), py [nsteps][N][ny] likewise; either
// may be NULL; both NULL switches the schedules off again.  With a schedule set mpc_loop_run launches loop_kernel_pxy.
extern "C" int mpc_loop_set_model_schedule(mpc_handle *h, int32_t nsteps, const double *px, const double *py)
{
    if (!h || h->B == 0) return fail(-1, "mpc_loop_alloc first");
    if (!px && !py) { h->msch_steps = 0; h->msch_px = h->msch_py = false; return 0; }
    if (nsteps < 1 || nsteps > h->max_steps) return fail(-1, "nsteps %d exceeds the allocated %d", nsteps, h->max_steps);
    HIP_TRY(hipSetDevice(h->device));
    const DevProblem &P = h->hp;
    const size_t npx = (size_t)nsteps * P.N * P.nx, npy = (size_t)nsteps * P.N * P.ny;
    if (h->msch.ensure((npx + npy) * sizeof(double))) return -10;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (px) HIP_TRY(hipMemcpy(h->msch.p, px, npx * sizeof(double), hipMemcpyHostToDevice));
    if (py) HIP_TRY(hipMemcpy((double *)h->msch.p + npx, py, npy * sizeof(double), hipMemcpyHostToDevice));
    h->msch_steps = nsteps; h->msch_px = px != nullptr; h->msch_py = py != nullptr;
    return 0;
}

// Which closed-loop kernel: 1 = one instance per lane (loop_kernel; fills the chip from about 65536 instances),
// 2 = horizon-parallel (loop_kernel_tp; one wave per instance, for batches that leave the chip idle otherwise).
static int loop_mode(const mpc_handle *h)
{
    if (h->loop_kernel_opt != 0) return h->loop_kernel_opt;
    if (h->hp.term_cons && !(h->L.loop_wv && h->hp.N <= 64)) return 1;      // terminal equality: exact on the lane and the wave-autonomous solver (term_aim), by weight alone on the horizon-parallel one
    {      // violently unstable open loop: the kernels whose recursions are scans with A^(2^e) lose digits there (see ocp_uses_wave)
        double nrm = 0.0;
        for (int i = 0; i < kMaxN; i++) for (int j = 0; j < kMaxN; j++) nrm = std::fmax(nrm, std::fabs(h->hp.Apow[5][i][j]));
        if (nrm > 1e4) return 1;
    }
    if (h->L.loop_wv && h->hp.N <= 64) return 3;
    if (!h->L.loop_tp || h->hp.N > 64) return 1;
    return h->B <= h->L.tp_max_batch ? 2 : 1;
}

extern "C" int mpc_loop_run(mpc_handle *h, int32_t k0, int32_t nsteps)
{
    if (!h || h->B == 0) return fail(-1, "mpc_loop_alloc first");
    if (k0 < 0 || nsteps < 1 || k0 + nsteps > h->sched_steps) return fail(-1, "steps [%d,%d) outside the schedule of %d steps", k0, k0 + nsteps, h->sched_steps);
    if (!h->state_set) return fail(-1, "mpc_loop_set_state first (the resident state is all zeros after mpc_loop_alloc)");
    HIP_TRY(hipSetDevice(h->device));
    const DevProblem &P = h->hp;
    const size_t Bs = h->Bs, ms = h->max_steps;
    const double *sch = (const double *)h->sch.p;
    const bool pxy = h->msch_steps > 0;      // def_px / def_py schedules: the instance-per-lane loop with per-block stage data
    if (pxy && k0 + nsteps > h->msch_steps) return fail(-1, "steps [%d,%d) outside the model-parameter schedule of %d steps", k0, k0 + nsteps, h->msch_steps);
    if (pxy) {
        const size_t N = P.N;
        if (h->pxy_lin.ensure(N * h->L.pxy_lin * Bs * sizeof(double)) || h->pxy_ws.ensure((size_t)h->L.pxy_ws_rows * (N + 2) * Bs * sizeof(double))) return -10;
    }
    if (pxy && P.soft) return fail(-8, "soft constraints with horizon parameters (def_px / def_py) are not carried");
    const int mode = pxy ? 5 : (P.soft ? 6 : loop_mode(h));
    if (mode != h->ws_mode) {      // the workspace holds another layout (or a per-call solve used it): next OCPs start cold
        HIP_TRY(hipMemsetAsync(h->st_flag.p, 0, 3 * Bs * 4, h->stream));
        h->ws_mode = mode;
    }
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    int launches = 0;
    for (int k = k0; k < k0 + nsteps; k += h->steps_per_launch) {
        const int n = std::min(h->steps_per_launch, k0 + nsteps - k);
        LoopArgs a;
        a.x = (double *)h->st_x.p; a.xhat = (double *)h->st_xhat.p; a.dhat = (double *)h->st_dhat.p; a.Pk = (double *)h->st_P.p;
        a.u = (double *)h->st_u.p; a.xs = (double *)h->st_xs.p; a.us = (double *)h->st_us.p;
        a.ysp = sch + (size_t)k * P.ny; a.usp = sch + ms * P.ny + (size_t)k * P.nu;
        a.pxp = sch + ms * (P.ny + P.nu) + (size_t)k * P.nxp; a.pyp = sch + ms * (P.ny + P.nu + P.nxp) + (size_t)k * P.ny;
        auto dl = [&](const char *nm) -> double * {
            auto it = h->log_off.find(nm);
            if (it == h->log_off.end()) return nullptr;
            return (double *)h->logs.p + it->second.first + (size_t)k * it->second.second * Bs;
        };
        a.U = dl("U"); a.XHAT = dl("X_HAT"); a.XS = dl("XS"); a.US = dl("US"); a.YS = dl("YS"); a.XP = dl("Xp"); a.DHAT = dl("D_HAT");
        if (h->log_level >= MPC_LOG_U) {
            int32_t *li = (int32_t *)h->logi.p;
            a.st_dyn = li + (size_t)k * Bs; a.st_ss = li + ms * Bs + (size_t)k * Bs; a.it_dyn = li + 2 * ms * Bs + (size_t)k * Bs; a.it_ss = li + 3 * ms * Bs + (size_t)k * Bs;
        } else a.st_dyn = a.st_ss = a.it_dyn = a.it_ss = nullptr;
        a.ws_valid = (int32_t *)h->st_flag.p; a.kf_valid = a.ws_valid + Bs; a.Kg = (double *)h->st_Kg.p; a.Pn = (double *)h->st_Pn.p;
        a.tw = (double *)h->st_tw.p; a.tw_valid = a.ws_valid + 2 * Bs;
        a.ws = (double *)h->ws.p; a.B = h->B; a.nsteps = n; a.Bs = Bs; a.N = P.N;
        a.h = h->h_sample; a.t0 = k * h->h_sample; a.wv_ni = h->wv_ni_opt;
        a.px_h = a.py_h = nullptr; a.lin = nullptr; a.SL = dl("SL"); a.soft_ws = (double *)h->soft_ws.p; a.sl_keep = (double *)h->soft_keep.p;
        if (P.soft) h->L.loop_soft(h->dp, a, h->stream);
        else if (pxy) {
            const size_t npx = (size_t)h->msch_steps * P.N * P.nx;
            a.px_h = h->msch_px ? (const double *)h->msch.p + (size_t)k * P.N * P.nx : nullptr;
            a.py_h = h->msch_py ? (const double *)h->msch.p + npx + (size_t)k * P.N * P.ny : nullptr;
            a.lin = (double *)h->pxy_lin.p; a.ws = (double *)h->pxy_ws.p;
            h->L.loop_pxy(h->dp, a, h->stream);
        } else
        if (mode == 3) { if (h->L.loop_wv(h->dp, a, h->stream)) return fail(-9, "cannot configure the wave-autonomous kernel (LDS %zu bytes)", h->L.wv_lds); }
        else if (mode == 2) { if (h->L.loop_tp(h->dp, a, h->stream)) return fail(-9, "cannot configure the horizon-parallel kernel (LDS %zu bytes)", h->L.tp_lds); }
        else h->L.loop(h->dp, a, h->stream);
        launches++;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->timed = true; h->n_launches = launches; h->last_k0 = k0; h->last_n = nsteps;
    return 0;
}

extern "C" int mpc_loop_sync(mpc_handle *h)
{
    if (!h) return fail(-1, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int mpc_loop_get_log(mpc_handle *h, const char *name, void *out)
{
    if (!h || !name || !out) return fail(-1, "null argument");
    auto it = h->log_off.find(name);
    if (it == h->log_off.end()) return fail(-8, "log '%s' was not enabled in mpc_loop_alloc", name);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t Bs = h->Bs; const int B = h->B, ns = h->sched_steps;
    if (it->second.second > 0) {
        const int d = it->second.second;
        std::vector<double> st((size_t)ns * d * Bs);
        HIP_TRY(hipMemcpy(st.data(), (double *)h->logs.p + it->second.first, st.size() * 8, hipMemcpyDeviceToHost));
        double *o = (double *)out;
        for (int k = 0; k < ns; k++) from_soa(st.data() + (size_t)k * d * Bs, B, d, Bs, o + (size_t)k * B * d);
    } else {
        std::vector<int32_t> st((size_t)ns * Bs);
        HIP_TRY(hipMemcpy(st.data(), (int32_t *)h->logi.p + it->second.first, st.size() * 4, hipMemcpyDeviceToHost));
        int32_t *o = (int32_t *)out;
        for (int k = 0; k < ns; k++) for (int b = 0; b < B; b++) o[(size_t)k * B + b] = st[(size_t)k * Bs + b];
    }
    return 0;
}

extern "C" void *mpc_dev_ptr(mpc_handle *h, const char *name, int64_t *bpad)
{
    if (!h || !name) return nullptr;
    if (bpad) *bpad = (int64_t)h->Bs;
    const std::string n(name);
    if (n == "x_p") return h->st_x.p;
    if (n == "xhat") return h->st_xhat.p;
    if (n == "dhat") return h->st_dhat.p;
    if (n == "P") return h->st_P.p;
    if (n == "u") return h->st_u.p;
    if (n == "xs") return h->st_xs.p;
    if (n == "us") return h->st_us.p;
    if (n == "coll_recv") return h->coll_recv.p;
    auto it = h->log_off.find(n);
    if (it == h->log_off.end()) return nullptr;
    if (it->second.second > 0) return (double *)h->logs.p + it->second.first;
    return (int32_t *)h->logi.p + it->second.first;
}

extern "C" int mpc_pack_u(mpc_handle *h, void *dst_dev)
{
    if (!h || h->B == 0 || !dst_dev) return fail(-1, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(pack_u_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, (const double *)h->st_u.p, (double *)dst_dev, h->B, h->Bs, h->hp.nu);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// multi-GPU: one process per GPU, RCCL over xGMI.  Instances are independent, so the only
// exchanges are the all-gather of the controls (per step: mpc_allgather_u; per run: mpc_allgather_log) and the
// job-level barrier / reductions of the benchmark harness.  librccl is opened on first use: a single-GPU user of
// the library never loads it.
// ---------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
int rccl_load()
{
    if (g_rccl.lib) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) return fail(-12, "librccl.so not found: %s", dlerror());
#define MPC_RCCL_SYM(field, sym) *(void **)(&g_rccl.field) = dlsym(lib, sym); if (!g_rccl.field) { dlclose(lib); return fail(-12, "librccl lacks %s", sym); }
    MPC_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") MPC_RCCL_SYM(CommInitRank, "ncclCommInitRank") MPC_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    MPC_RCCL_SYM(AllGather, "ncclAllGather") MPC_RCCL_SYM(AllReduce, "ncclAllReduce") MPC_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef MPC_RCCL_SYM
    g_rccl.lib = lib;
    return 0;
}
}  // namespace
#define RCCL_TRY(x)                                                                                      \
    do {                                                                                                 \
        ncclResult_t r_ = (x);                                                                           \
        if (r_ != ncclSuccess) return fail(-12, "%s failed: %s (%s:%d)", #x, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// RCCL writes its version banner to stdout when it initialises (NCCL_DEBUG=VERSION, as on the benchmark boxes): while one of its set-up
// calls runs, file descriptor 1 points at stderr, so that a caller's stdout carries only what the caller prints (bench.py: one JSON line).
namespace {
struct StdoutToStderr {
    int saved = -1;
    StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) (void)dup2(2, 1); }
    ~StdoutToStderr() { fflush(stdout); if (saved >= 0) { (void)dup2(saved, 1); close(saved); } }
};
}

extern "C" int mpc_comm_unique_id(char *out128)
{
    if (!out128) return fail(-1, "null argument");
    if (rccl_load()) return -12;
    ncclUniqueId id;
    StdoutToStderr quiet;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == MPC_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(out128, &id, sizeof(id));
    return 0;
}

extern "C" int mpc_comm_init(mpc_handle *h, int32_t rank, int32_t world, const char *id128)
{
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return fail(-1, "bad argument");
    if (h->comm) return fail(-1, "the handle already has a communicator");
    if (rccl_load()) return -12;
    HIP_TRY(hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    {
        StdoutToStderr quiet;
        RCCL_TRY(g_rccl.CommInitRank(&h->comm, world, id, rank));
    }
    h->rank = rank; h->world = world;
    return 0;
}

extern "C" int mpc_comm_destroy(mpc_handle *h)
{
    if (!h || !h->comm) return 0;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->rank = 0; h->world = 1;
    return 0;
}

extern "C" int mpc_comm_rank(mpc_handle *h, int32_t *rank, int32_t *world)
{
    if (!h) return fail(-1, "null handle");
    if (rank) *rank = h->rank;
    if (world) *world = h->world;
    return 0;
}

// all-gather of `bytes` bytes per rank between host buffers, staged through device memory (rank r's block lands at recv + r * bytes)
extern "C" int mpc_comm_allgather(mpc_handle *h, const void *send, size_t bytes, void *recv)
{
    if (!h || !send || !recv || bytes == 0) return fail(-1, "bad argument");
    if (!h->comm) { std::memcpy(recv, send, bytes); return 0; }
    HIP_TRY(hipSetDevice(h->device));
    if (h->coll_send.ensure(bytes) || h->coll_recv.ensure(bytes * h->world)) return -10;
    HIP_TRY(hipMemcpyAsync(h->coll_send.p, send, bytes, hipMemcpyHostToDevice, h->stream));
    RCCL_TRY(g_rccl.AllGather(h->coll_send.p, h->coll_recv.p, bytes, ncclChar, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(recv, h->coll_recv.p, bytes * h->world, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// max over the ranks of n host doubles, in place
extern "C" int mpc_comm_allreduce_max(mpc_handle *h, double *inout, int32_t n)
{
    if (!h || !inout || n < 1) return fail(-1, "bad argument");
    if (!h->comm) return 0;
    HIP_TRY(hipSetDevice(h->device));
    if (h->coll_send.ensure(sizeof(double) * n)) return -10;
    HIP_TRY(hipMemcpyAsync(h->coll_send.p, inout, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    RCCL_TRY(g_rccl.AllReduce(h->coll_send.p, h->coll_send.p, n, ncclDouble, ncclMax, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(inout, h->coll_send.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// everything queued on this handle's stream on every rank has completed when this returns
extern "C" int mpc_comm_barrier(mpc_handle *h)
{
    if (!h) return fail(-1, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (h->comm) { double one = 1.0; const int rc = mpc_comm_allreduce_max(h, &one, 1); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// u* of the last closed-loop step of every rank: ncclAllGather(u_local[B][nu]).  u_all (host, optional)
// receives [world][B][nu]; the gathered block also stays on the device (mpc_dev_ptr "coll_recv").
extern "C" int mpc_allgather_u(mpc_handle *h, double *u_all)
{
    if (!h || h->B == 0) return fail(-1, "mpc_loop_alloc first");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->B * h->hp.nu, bytes = n * sizeof(double);
    if (h->coll_send.ensure(bytes) || h->coll_recv.ensure(bytes * h->world)) return -10;
    hipLaunchKernelGGL(pack_u_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, (const double *)h->st_u.p, (double *)h->coll_send.p, h->B, h->Bs, h->hp.nu);
    HIP_TRY(hipGetLastError());
    if (h->comm) RCCL_TRY(g_rccl.AllGather(h->coll_send.p, h->coll_recv.p, n, ncclDouble, h->comm, h->stream));
    else HIP_TRY(hipMemcpyAsync(h->coll_recv.p, h->coll_send.p, bytes, hipMemcpyDeviceToDevice, h->stream));
    if (u_all) { HIP_TRY(hipMemcpyAsync(u_all, h->coll_recv.p, bytes * h->world, hipMemcpyDeviceToHost, h->stream)); HIP_TRY(hipStreamSynchronize(h->stream)); }
    return 0;
}

// steps [k0, k0 + nsteps) of a float64 log of every rank, gathered device to device straight from the log ([step][dim][Bpad] per
// rank, asynchronous on the handle's stream); out (host, optional) receives [world][nsteps][B][dim]
extern "C" int mpc_allgather_log(mpc_handle *h, const char *name, int32_t k0, int32_t nsteps, double *out)
{
    if (!h || h->B == 0 || !name) return fail(-1, "bad argument");
    auto it = h->log_off.find(name);
    if (it == h->log_off.end() || it->second.second == 0) return fail(-8, "float64 log '%s' was not enabled in mpc_loop_alloc", name);
    if (k0 < 0 || nsteps < 1 || k0 + nsteps > h->max_steps) return fail(-1, "steps out of range");
    HIP_TRY(hipSetDevice(h->device));
    const int d = it->second.second;
    const size_t row = (size_t)d * h->Bs, n = (size_t)nsteps * row;
    if (h->coll_recv.ensure(n * sizeof(double) * h->world)) return -10;
    const double *src = (const double *)h->logs.p + it->second.first + (size_t)k0 * row;
    if (h->comm) RCCL_TRY(g_rccl.AllGather(src, h->coll_recv.p, n, ncclDouble, h->comm, h->stream));
    else HIP_TRY(hipMemcpyAsync(h->coll_recv.p, src, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (out) {
        std::vector<double> st(n * h->world);
        HIP_TRY(hipMemcpyAsync(st.data(), h->coll_recv.p, st.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int r = 0; r < h->world; r++)
            for (int k = 0; k < nsteps; k++)
                from_soa(st.data() + ((size_t)r * nsteps + k) * row, h->B, d, h->Bs, out + (((size_t)r * nsteps + k) * h->B) * d);
    }
    return 0;
}

#ifdef MPC_STAMPS   /* diagnostic build only, see tools/stamps.py */
extern "C" int mpc_debug_stamps(unsigned long long *out, int n, int reset)
{
    if (out) { if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mpc::mpc_stamp_buf), sizeof(unsigned long long) * n) != hipSuccess) return -1; }
    if (reset) { static unsigned long long z[4096 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(mpc::mpc_stamp_buf), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

extern "C" int mpc_pack_log(mpc_handle *h, const char *name, int32_t k0, int32_t nsteps, void *dst_dev)
{
    if (!h || h->B == 0 || !name || !dst_dev) return fail(-1, "bad argument");
    auto it = h->log_off.find(name);
    if (it == h->log_off.end() || it->second.second == 0) return fail(-8, "float64 log '%s' was not enabled in mpc_loop_alloc", name);
    if (k0 < 0 || nsteps < 1 || k0 + nsteps > h->max_steps) return fail(-1, "steps out of range");
    HIP_TRY(hipSetDevice(h->device));
    const size_t row = (size_t)it->second.second * h->Bs;
    HIP_TRY(hipMemcpyAsync(dst_dev, (double *)h->logs.p + it->second.first + (size_t)k0 * row, (size_t)nsteps * row * 8, hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

extern "C" int mpc_closed_loop(mpc_handle *h, int32_t B, int32_t nsteps, double *x_p, double *xhat, double *dhat,
                               double *Pk, double *u, double *xs, double *us, const double *ysp, const double *usp,
                               const double *xsp, const double *pxp, const double *pyp, double *U_log)
{
    if (!h) return fail(-1, "null handle");
    int rc;
    if (h->B != B || h->max_steps < nsteps || (U_log && h->log_level < MPC_LOG_U))
        if ((rc = mpc_loop_alloc(h, B, nsteps, U_log ? MPC_LOG_U : MPC_LOG_NONE))) return rc;
    if ((rc = mpc_loop_set_state(h, x_p, xhat, dhat, Pk, u, xs, us))) return rc;
    if ((rc = mpc_loop_set_schedule(h, nsteps, ysp, usp, xsp, pxp, pyp))) return rc;
    if ((rc = mpc_loop_run(h, 0, nsteps))) return rc;
    if ((rc = mpc_loop_sync(h))) return rc;
    if ((rc = mpc_loop_get_state(h, x_p, xhat, dhat, Pk, u, xs, us))) return rc;
    if (U_log && (rc = mpc_loop_get_log(h, "U", U_log))) return rc;
    return 0;
}
#endif      // MPC_PART2
