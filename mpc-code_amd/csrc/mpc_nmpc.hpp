// Non-linear MPC on the GPU (BASELINE config 3): device side.
//
// One instance per lane.  Per closed-loop step (reference MPC_code.py:485-827 with a non-linear model):
//   measure          y = Fy_p(x_p)                                            (:531-534; generated NlModel::hp)
//   estimate         extended Kalman filter on [x; d], d+ = d                 (Estimator.py:313-386; ekf_lane)
//   target           the NLP of opt_ss (Target_Calc.py:20-161) by SQP: each iteration linearises the discrete model at the iterate
//                    and solves the linear target QP - target_lane, fed per-instance matrices (TargetLocal + target_prepare)
//   OCP              the NLP of opt_dyn (Control_Calc.py:20-260) by SQP on the Riccati-PDIP solver: each iteration linearises the
//                    model along the current trajectory (A_k, B_k, c_k by RK4 with forward sensitivities) and solves the
//                    time-varying QP with rpdip_lane<..., LTV>; one iteration per step = real-time iteration, iterated to a fixed
//                    point = the NLP's KKT point (the cost is quadratic: the Gauss-Newton Hessian is the cost's own)
//   plant            x_p+ = Fx_p(x_p, u)                                      (:813-816; RK4 of NlModel::fp)
// The discrete model is the reference's: MX classical Runge-Kutta steps of the Ex-file's continuous function per sampling
// interval with time carried along (Utilities.py:157-183, casadi.simpleRK); NlModel is generated from the traced Ex-file
// functions (mpc-code_amd/nlcodegen.py).
#pragma once
#include "mpc_device.hpp"

namespace mpc {

// x(t + h): MX RK4 steps of dx/dt = f(x, u, d, t)
// Each integrator comes twice: *_inl (always inlined: the wave-autonomous kernel parks its iterates in the accumulation registers,
// and a call would spill all of them around it) and a real function of the same name for the instance-per-lane kernel (code size).
template <class M>
__device__ __forceinline__ void rk4_model_inl(const double *x0, const double *u, const double *d, double t, double h, double *xn)
{
    constexpr int NX = M::NX;
    if (M::DISCRETE) { M::f(x0, u, d, t, xn); return; }      // the user's map is the step (Utilities.py:186-198)
    const double dt = h / M::MX;
    double x[NX];
    MPC_UNROLL for (int i = 0; i < NX; i++) x[i] = x0[i];
    for (int s = 0; s < M::MX; s++) {
        const double ts = t + s * dt;
        double k1[NX], k2[NX], k3[NX], k4[NX], xa[NX];
        M::f(x, u, d, ts, k1);
        MPC_UNROLL for (int i = 0; i < NX; i++) xa[i] = x[i] + 0.5 * dt * k1[i];
        M::f(xa, u, d, ts + 0.5 * dt, k2);
        MPC_UNROLL for (int i = 0; i < NX; i++) xa[i] = x[i] + 0.5 * dt * k2[i];
        M::f(xa, u, d, ts + 0.5 * dt, k3);
        MPC_UNROLL for (int i = 0; i < NX; i++) xa[i] = x[i] + dt * k3[i];
        M::f(xa, u, d, ts + dt, k4);
        MPC_UNROLL for (int i = 0; i < NX; i++) x[i] += dt / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    MPC_UNROLL for (int i = 0; i < NX; i++) xn[i] = x[i];
}

template <class M>
__device__ __noinline__ void rk4_model(const double *x0, const double *u, const double *d, double t, double h, double *xn) { rk4_model_inl<M>(x0, u, d, t, h, xn); }

template <class M>
__device__ __forceinline__ void rk4_plant_inl(const double *x0, const double *u, double t, double h, double *xn)
{
    constexpr int NX = M::NXP;
    if (M::PLANT_DISCRETE) { M::fp(x0, u, t, xn); return; }
    const double dt = h / M::MX;
    double x[NX];
    MPC_UNROLL for (int i = 0; i < NX; i++) x[i] = x0[i];
    for (int s = 0; s < M::MX; s++) {
        const double ts = t + s * dt;
        double k1[NX], k2[NX], k3[NX], k4[NX], xa[NX];
        M::fp(x, u, ts, k1);
        MPC_UNROLL for (int i = 0; i < NX; i++) xa[i] = x[i] + 0.5 * dt * k1[i];
        M::fp(xa, u, ts + 0.5 * dt, k2);
        MPC_UNROLL for (int i = 0; i < NX; i++) xa[i] = x[i] + 0.5 * dt * k2[i];
        M::fp(xa, u, ts + 0.5 * dt, k3);
        MPC_UNROLL for (int i = 0; i < NX; i++) xa[i] = x[i] + dt * k3[i];
        M::fp(xa, u, ts + dt, k4);
        MPC_UNROLL for (int i = 0; i < NX; i++) x[i] += dt / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    MPC_UNROLL for (int i = 0; i < NX; i++) xn[i] = x[i];
}

template <class M>
__device__ __noinline__ void rk4_plant(const double *x0, const double *u, double t, double h, double *xn) { rk4_plant_inl<M>(x0, u, t, h, xn); }

// The same with forward sensitivities: S = d x(t+h) / d [x0 | u | d], propagated through every Runge-Kutta stage
// (dK_i = f_x(X_i) dX_i + [0 | f_u | f_d](X_i)).  Out: xn, A = S[:, :NX], B = S[:, NX:NX+NU], G = S[:, NX+NU:].
// WITHG = false: the sensitivities with respect to the disturbance are not propagated (G is left untouched): target and OCP
// linearise in (x, u) only.
template <class M, bool WITHG = true>
__device__ __forceinline__ void rk4_model_sens_inl(const double *x0, const double *u, const double *d, double t, double h,
                                                   double *xn, double (*A)[M::NX], double (*B)[M::NU], double (*G)[M::ND > 0 ? M::ND : 1])
{
    constexpr int NX = M::NX, NU = M::NU, ND = M::ND, NDD = ND > 0 ? ND : 1, NP = NX + NU + (WITHG ? ND : 0);
    if (M::DISCRETE) { M::f_jac(x0, u, d, t, xn, A, B, G); return; }
    const double dt = h / M::MX;
    double x[NX], S[NX][NP];
    MPC_UNROLL for (int i = 0; i < NX; i++) { x[i] = x0[i]; MPC_UNROLL for (int j = 0; j < NP; j++) S[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int s = 0; s < M::MX; s++) {
        const double ts = t + s * dt;
        double kx[NX], xacc[NX], Sacc[NX][NP], Xi[NX], dX[NX][NP], kprev[NX], dKprev[NX][NP];
        MPC_UNROLL for (int i = 0; i < NX; i++) { xacc[i] = x[i]; kprev[i] = 0.0; MPC_UNROLL for (int j = 0; j < NP; j++) { Sacc[i][j] = S[i][j]; dKprev[i][j] = 0.0; } }
        MPC_UNROLL for (int st = 0; st < 4; st++) {
            const double a = st == 0 ? 0.0 : (st == 3 ? 1.0 : 0.5), w = (st == 0 || st == 3) ? 1.0 / 6.0 : 1.0 / 3.0;
            MPC_UNROLL for (int i = 0; i < NX; i++) { Xi[i] = x[i] + a * dt * kprev[i]; MPC_UNROLL for (int j = 0; j < NP; j++) dX[i][j] = S[i][j] + a * dt * dKprev[i][j]; }
            double fx[NX][NX], fu[NX][NU], fd[NX][NDD];
            M::f_jac(Xi, u, d, ts + a * dt, kx, fx, fu, fd);
            double dK[NX][NP];
            MPC_UNROLL for (int i = 0; i < NX; i++) {
                MPC_UNROLL for (int j = 0; j < NP; j++) {
                    double v = j < NX ? 0.0 : (j < NX + NU ? fu[i][j < NX + NU && j >= NX ? j - NX : 0] : fd[i][j >= NX + NU ? j - NX - NU : 0]);
                    MPC_UNROLL for (int l = 0; l < NX; l++) v += fx[i][l] * dX[l][j];
                    dK[i][j] = v;
                }
            }
            MPC_UNROLL for (int i = 0; i < NX; i++) {
                xacc[i] += dt * w * kx[i]; kprev[i] = kx[i];
                MPC_UNROLL for (int j = 0; j < NP; j++) { Sacc[i][j] += dt * w * dK[i][j]; dKprev[i][j] = dK[i][j]; }
            }
        }
        MPC_UNROLL for (int i = 0; i < NX; i++) { x[i] = xacc[i]; MPC_UNROLL for (int j = 0; j < NP; j++) S[i][j] = Sacc[i][j]; }
    }
    MPC_UNROLL for (int i = 0; i < NX; i++) {
        xn[i] = x[i];
        MPC_UNROLL for (int j = 0; j < NX; j++) A[i][j] = S[i][j];
        MPC_UNROLL for (int j = 0; j < NU; j++) B[i][j] = S[i][NX + j];
        if (WITHG) { MPC_UNROLL for (int j = 0; j < ND; j++) G[i][j] = S[i][NX + NU + (WITHG ? j : 0)]; }
    }
}

template <class M, bool WITHG = true>
__device__ __noinline__ void rk4_model_sens(const double *x0, const double *u, const double *d, double t, double h,
                                            double *xn, double (*A)[M::NX], double (*B)[M::NU], double (*G)[M::ND > 0 ? M::ND : 1])
{
    rk4_model_sens_inl<M, WITHG>(x0, u, d, t, h, xn, A, B, G);
}

// Extended Kalman filter, Estimator.py:313-386: gain and correction with the output Jacobian at the prior, then the prior of the
// next step through the Jacobian of [Fx_model(x, u, d); d] at the corrected estimate.
template <class M, bool INL = false>
__device__ __forceinline__ void ekf_lane(const DevProblem &P, double (&xi)[M::NX + M::ND], double (&Pk)[M::NX + M::ND][M::NX + M::ND], const double (&y)[M::NY],
                         const double *u, double t, double h)
{
    constexpr int NX = M::NX, ND = M::ND, NE = NX + ND, NY = M::NY, NU = M::NU, NDD = ND > 0 ? ND : 1;
    double yhat[NY], hx[NY][NX], hd[NY][NDD], C[NY][NE];
    M::h(xi, u, xi + NX, t, yhat);
    M::h_jac(xi, u, xi + NX, t, hx, hd);
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < NX; j++) C[i][j] = hx[i][j]; MPC_UNROLL for (int j = 0; j < ND; j++) C[i][NX + j] = hd[i][j]; }
    double PCt[NE][NY], S[NY][NY], K[NE][NY];
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += Pk[i][l] * C[j][l]; PCt[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = P.Rkf[i][j]; MPC_UNROLL for (int l = 0; l < NE; l++) a += C[i][l] * PCt[l][j]; S[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < i; j++) { const double a = 0.5 * (S[i][j] + S[j][i]); S[i][j] = a; S[j][i] = a; } }
    sym_inverse<NY>(S);
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) a += PCt[i][l] * S[l][j]; K[i][j] = a; } }
    double CP[NY][NE], Pc[NE][NE];
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += C[i][l] * Pk[l][j]; CP[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = Pk[i][j]; MPC_UNROLL for (int l = 0; l < NY; l++) a -= K[i][l] * CP[l][j]; Pc[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NE; i++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NY; l++) a += K[i][l] * (y[l] - yhat[l]); xi[i] += a; }
    double xn[NX], A[NX][NX], B[NX][NU], G[NX][NDD], Aa[NE][NE], T[NE][NE];
    if (INL) rk4_model_sens_inl<M>(xi, u, xi + NX, t, h, xn, A, B, G); else rk4_model_sens<M>(xi, u, xi + NX, t, h, xn, A, B, G);
    MPC_UNROLL for (int i = 0; i < NE; i++) {
        MPC_UNROLL for (int j = 0; j < NE; j++) {
            double v = (i == j) ? 1.0 : 0.0;
            if (i < NX) v = j < NX ? A[i < NX ? i : 0][j < NX ? j : 0] : G[i < NX ? i : 0][j >= NX ? j - NX : 0];
            Aa[i][j] = v;
        }
    }
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = 0.0; MPC_UNROLL for (int l = 0; l < NE; l++) a += Aa[i][l] * Pc[l][j]; T[i][j] = a; } }
    MPC_UNROLL for (int i = 0; i < NE; i++) { MPC_UNROLL for (int j = 0; j < NE; j++) { double a = P.Qkf[i][j]; MPC_UNROLL for (int l = 0; l < NE; l++) a += T[i][l] * Aa[j][l]; Pk[i][j] = a; } }
}

// Per-instance data of the linear target QP, under the names target_lane reads from the problem struct
template <int NX, int NU, int NY>
struct TargetLocal {
    static constexpr int NV = NX + NU, NC = NV + NY;
    int duss_form, max_iter;
    double fxc[NX], fyc[NY], Bd[NX][1], Cd[NY][1];
    double Ep[NV][NX], Zn[NV][NU], Cm[NY][NX], CZx[NY][NU], Hr[NU][NU], W[NC][NU], tlo[NC], thi[NC], Qss[NY][NY], Rss[NU][NU];
};

// Householder QR of [A - I, B]' -> particular-solution map Ep and null-space basis Zn; reduced Hessian and constraint rows.
// The device twin of build_target (mpc_amd.hip), for matrices that differ from instance to instance.
template <int NX, int NU, int NY>
__device__ bool target_prepare(const double (&A)[NX][NX], const double (&B)[NX][NU], TargetLocal<NX, NU, NY> &T)
{
    constexpr int NV = NX + NU;
    double Qf[NV][NV], Rm[NV][NX];
    MPC_UNROLL for (int i = 0; i < NV; i++) { MPC_UNROLL for (int j = 0; j < NV; j++) Qf[i][j] = (i == j) ? 1.0 : 0.0; }
    MPC_UNROLL for (int i = 0; i < NX; i++) {
        MPC_UNROLL for (int j = 0; j < NX; j++) Rm[j][i] = A[i][j] - (i == j ? 1.0 : 0.0);
        MPC_UNROLL for (int j = 0; j < NU; j++) Rm[NX + j][i] = B[i][j];
    }
    MPC_UNROLL for (int k = 0; k < NX; k++) {
        double v[NV], nrm = 0.0, vn = 0.0;
        MPC_UNROLL for (int i = 0; i < NV; i++) if (i >= k) nrm += Rm[i][k] * Rm[i][k];
        nrm = sqrt(nrm);
        if (!(nrm > 0.0)) return false;
        const double alpha = Rm[k][k] > 0 ? -nrm : nrm;
        MPC_UNROLL for (int i = 0; i < NV; i++) v[i] = i < k ? 0.0 : Rm[i][k];
        v[k] -= alpha;
        MPC_UNROLL for (int i = 0; i < NV; i++) if (i >= k) vn += v[i] * v[i];
        if (vn > 0.0) {
            const double s2 = 2.0 / vn;
            MPC_UNROLL for (int j = 0; j < NX; j++) { double s = 0.0; MPC_UNROLL for (int i = 0; i < NV; i++) if (i >= k) s += v[i] * Rm[i][j]; s *= s2; MPC_UNROLL for (int i = 0; i < NV; i++) if (i >= k) Rm[i][j] -= s * v[i]; }
            MPC_UNROLL for (int j = 0; j < NV; j++) { double s = 0.0; MPC_UNROLL for (int i = 0; i < NV; i++) if (i >= k) s += Qf[j][i] * v[i]; s *= s2; MPC_UNROLL for (int i = 0; i < NV; i++) if (i >= k) Qf[j][i] -= s * v[i]; }
        }
    }
    double rmax = 0.0;
    MPC_UNROLL for (int k = 0; k < NX; k++) rmax = dmax(rmax, fabs(Rm[k][k]));
    MPC_UNROLL for (int k = 0; k < NX; k++) if (fabs(Rm[k][k]) < 1e-12 * rmax) return false;
    double Rti[NX][NX];
    MPC_UNROLL for (int c = 0; c < NX; c++) {
        MPC_UNROLL for (int i = 0; i < NX; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            MPC_UNROLL for (int j = 0; j < NX; j++) if (j < i) s -= Rm[j][i] * Rti[j][c];
            Rti[i][c] = s / Rm[i][i];
        }
    }
    MPC_UNROLL for (int r = 0; r < NV; r++) {
        MPC_UNROLL for (int c = 0; c < NX; c++) { double s = 0.0; MPC_UNROLL for (int j = 0; j < NX; j++) s += Qf[r][j] * Rti[j][c]; T.Ep[r][c] = s; }
        MPC_UNROLL for (int c = 0; c < NU; c++) T.Zn[r][c] = Qf[r][NX + c];
    }
    MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int c = 0; c < NU; c++) { double s = 0.0; MPC_UNROLL for (int j = 0; j < NX; j++) s += T.Cm[i][j] * T.Zn[j][c]; T.CZx[i][c] = s; } }
    MPC_UNROLL for (int a = 0; a < NU; a++) {
        MPC_UNROLL for (int b = 0; b < NU; b++) {
            double s = 0.0;
            MPC_UNROLL for (int i = 0; i < NY; i++) { MPC_UNROLL for (int j = 0; j < NY; j++) s += T.CZx[i][a] * T.Qss[i][j] * T.CZx[j][b]; }
            MPC_UNROLL for (int i = 0; i < NU; i++) { MPC_UNROLL for (int j = 0; j < NU; j++) s += T.Zn[NX + i][a] * T.Rss[i][j] * T.Zn[NX + j][b]; }
            T.Hr[a][b] = s;
        }
    }
    MPC_UNROLL for (int a = 0; a < NU; a++) { MPC_UNROLL for (int b = 0; b < a; b++) { const double s = 0.5 * (T.Hr[a][b] + T.Hr[b][a]); T.Hr[a][b] = s; T.Hr[b][a] = s; } }
    MPC_UNROLL for (int r = 0; r < NV; r++) { MPC_UNROLL for (int c = 0; c < NU; c++) T.W[r][c] = T.Zn[r][c]; }
    MPC_UNROLL for (int r = 0; r < NY; r++) { MPC_UNROLL for (int c = 0; c < NU; c++) T.W[NV + r][c] = T.CZx[r][c]; }
    return true;
}

}  // namespace mpc
