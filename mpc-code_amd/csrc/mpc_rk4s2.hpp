// Runge-Kutta integration with forward sensitivities of first and second order: the part of the economic MPC path that is plain C++
// (no wave intrinsics), so that tests/test_enmpc.py can compile it with g++ against a generated model header and check the product's
// derivatives on the CPU (-D__device__= -D__forceinline__=inline).  Used by mpc_enmpc.hpp on the device.
#pragma once
#ifndef MPC_UNROLL
#define MPC_UNROLL _Pragma("unroll")
#endif
// The step sizes are the same in every lane: on the device they are moved to scalar registers (EC_UNI = mpc::uni, set by mpc_enmpc.hpp).  Left in vector
// registers they are loop invariants of the Runge-Kutta loop that the register allocator spills and reloads at each of their sixty uses per step (round 4:
// 64 scratch loads per step in the target kernel).
#ifndef EC_UNI
#define EC_UNI(x) (x)
#endif

namespace enm {

// ---- Runge-Kutta with forward sensitivities of first and second order ---------------------------------------------------------------
// R: a generated right-hand side (EcModel::Ocp / Mdl / Mhe): NR rows, the first NCX of them states with unit initial sensitivity,
// NP sensitivity columns (initial states, then inputs), NPP = NP (NP + 1) / 2 packed pairs.
template <class R, class Ctx>
__device__ __forceinline__ void rk4_sens2(const double *x0, const Ctx &c, double t0, bool advance_t, double h, int M, double *xn,
                                          double (*S)[R::NP], double (*T)[R::NPP])
{
    constexpr int NR = R::NR, NP = R::NP, NPP = R::NPP, NCX = R::NCX;
    const double dt = EC_UNI(h / M), hdt = EC_UNI(0.5 * dt), dt6 = EC_UNI(dt * (1.0 / 6.0)), dt3 = EC_UNI(dt * (1.0 / 3.0));
    double x[NR];
    MPC_UNROLL for (int i = 0; i < NR; i++) {
        x[i] = i < NCX ? x0[i < NCX ? i : 0] : 0.0;
        MPC_UNROLL for (int j = 0; j < NP; j++) S[i][j] = (i == j && i < NCX) ? 1.0 : 0.0;
        MPC_UNROLL for (int j = 0; j < NPP; j++) T[i][j] = 0.0;
    }
    for (int s = 0; s < M; s++) {
        const double ts = advance_t ? t0 + s * dt : t0;
        double xa[NR], Sa[NR][NP], Ta[NR][NPP], k[NR], dK[NR][NP], d2K[NR][NPP];
        MPC_UNROLL for (int i = 0; i < NR; i++) {
            xa[i] = x[i]; k[i] = 0.0;
            MPC_UNROLL for (int j = 0; j < NP; j++) { Sa[i][j] = S[i][j]; dK[i][j] = 0.0; }
            MPC_UNROLL for (int j = 0; j < NPP; j++) { Ta[i][j] = T[i][j]; d2K[i][j] = 0.0; }
        }
        MPC_UNROLL for (int st = 0; st < 4; st++) {
            const double adt = st == 0 ? 0.0 : (st == 3 ? dt : hdt), wdt = (st == 0 || st == 3) ? dt6 : dt3;      // a dt (a = 0, 1/2, 1/2, 1), dt w (w = 1/6, 1/3, 1/3, 1/6)
            double Xi[NR], dXi[NR][NP], d2Xi[NR][NPP];
            MPC_UNROLL for (int i = 0; i < NR; i++) {
                Xi[i] = x[i] + adt * k[i];
                MPC_UNROLL for (int j = 0; j < NP; j++) dXi[i][j] = S[i][j] + adt * dK[i][j];
                MPC_UNROLL for (int j = 0; j < NPP; j++) d2Xi[i][j] = T[i][j] + adt * d2K[i][j];
            }
            R::eval2(Xi, c, advance_t ? ts + adt : ts, dXi, d2Xi, k, dK, d2K);
            MPC_UNROLL for (int i = 0; i < NR; i++) {
                xa[i] += wdt * k[i];
                MPC_UNROLL for (int j = 0; j < NP; j++) Sa[i][j] += wdt * dK[i][j];
                MPC_UNROLL for (int j = 0; j < NPP; j++) Ta[i][j] += wdt * d2K[i][j];
            }
        }
        MPC_UNROLL for (int i = 0; i < NR; i++) {
            x[i] = xa[i];
            MPC_UNROLL for (int j = 0; j < NP; j++) S[i][j] = Sa[i][j];
            MPC_UNROLL for (int j = 0; j < NPP; j++) T[i][j] = Ta[i][j];
        }
    }
    MPC_UNROLL for (int i = 0; i < NR; i++) xn[i] = x[i];
}

// first-order sensitivities only (the gradient IPOPT scales the objective with): the generated eval2 with nothing asked of its second-order outputs -
// inlined, the compiler drops their arithmetic
template <class R, class Ctx>
__device__ __forceinline__ void rk4_sens1(const double *x0, const Ctx &c, double t0, bool advance_t, double h, int M, double *xn, double (*S)[R::NP])
{
    constexpr int NR = R::NR, NP = R::NP, NPP = R::NPP, NCX = R::NCX;
    const double dt = EC_UNI(h / M), hdt = EC_UNI(0.5 * dt), dt6 = EC_UNI(dt * (1.0 / 6.0)), dt3 = EC_UNI(dt * (1.0 / 3.0));
    double x[NR];
    MPC_UNROLL for (int i = 0; i < NR; i++) {
        x[i] = i < NCX ? x0[i < NCX ? i : 0] : 0.0;
        MPC_UNROLL for (int j = 0; j < NP; j++) S[i][j] = (i == j && i < NCX) ? 1.0 : 0.0;
    }
    for (int s = 0; s < M; s++) {
        const double ts = advance_t ? t0 + s * dt : t0;
        double xa[NR], Sa[NR][NP], k[NR], dK[NR][NP];
        MPC_UNROLL for (int i = 0; i < NR; i++) { xa[i] = x[i]; k[i] = 0.0; MPC_UNROLL for (int j = 0; j < NP; j++) { Sa[i][j] = S[i][j]; dK[i][j] = 0.0; } }
        MPC_UNROLL for (int st = 0; st < 4; st++) {
            const double adt = st == 0 ? 0.0 : (st == 3 ? dt : hdt), wdt = (st == 0 || st == 3) ? dt6 : dt3;
            double Xi[NR], dXi[NR][NP], d2Xi[NR][NPP], d2K[NR][NPP];
            MPC_UNROLL for (int i = 0; i < NR; i++) {
                Xi[i] = x[i] + adt * k[i];
                MPC_UNROLL for (int j = 0; j < NP; j++) dXi[i][j] = S[i][j] + adt * dK[i][j];
                MPC_UNROLL for (int j = 0; j < NPP; j++) d2Xi[i][j] = 0.0;
            }
            R::eval2(Xi, c, advance_t ? ts + adt : ts, dXi, d2Xi, k, dK, d2K);
            MPC_UNROLL for (int i = 0; i < NR; i++) {
                xa[i] += wdt * k[i];
                MPC_UNROLL for (int j = 0; j < NP; j++) Sa[i][j] += wdt * dK[i][j];
            }
        }
        MPC_UNROLL for (int i = 0; i < NR; i++) { x[i] = xa[i]; MPC_UNROLL for (int j = 0; j < NP; j++) S[i][j] = Sa[i][j]; }
    }
    MPC_UNROLL for (int i = 0; i < NR; i++) xn[i] = x[i];
}

// values only (plant, hold rule, first guess of the estimator)
template <class R, class Ctx>
__device__ __forceinline__ void rk4_plain(const double *x0, const Ctx &c, double t0, bool advance_t, double h, int M, double *xn)
{
    constexpr int NR = R::NR, NCX = R::NCX;
    const double dt = EC_UNI(h / M), hdt = EC_UNI(0.5 * dt), dt6 = EC_UNI(dt / 6.0);
    double x[NR];
    MPC_UNROLL for (int i = 0; i < NR; i++) x[i] = i < NCX ? x0[i < NCX ? i : 0] : 0.0;
    for (int s = 0; s < M; s++) {
        const double ts = advance_t ? t0 + s * dt : t0;
        double k1[NR], k2[NR], k3[NR], k4[NR], xa[NR];
        R::eval0(x, c, ts, k1);
        MPC_UNROLL for (int i = 0; i < NR; i++) xa[i] = x[i] + hdt * k1[i];
        R::eval0(xa, c, advance_t ? ts + hdt : ts, k2);
        MPC_UNROLL for (int i = 0; i < NR; i++) xa[i] = x[i] + hdt * k2[i];
        R::eval0(xa, c, advance_t ? ts + hdt : ts, k3);
        MPC_UNROLL for (int i = 0; i < NR; i++) xa[i] = x[i] + dt * k3[i];
        R::eval0(xa, c, advance_t ? ts + dt : ts, k4);
        MPC_UNROLL for (int i = 0; i < NR; i++) x[i] += dt6 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    MPC_UNROLL for (int i = 0; i < NR; i++) xn[i] = x[i];
}

}  // namespace enm
