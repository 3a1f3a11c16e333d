// TEST INFRASTRUCTURE: the soft-constraint solver of mpc-code_amd/csrc/mpc_soft.hpp (plain C++: one instance per lane, no wave intrinsics) compiled for the host,
// so that tests/test_soft.py runs the product's source next to the dense oracle without a GPU.   g++ -O2 -shared -fPIC -DSOFT_DIMS=NS,NU,NY
#include "mpc_soft.hpp"
#include <cstring>
#include <vector>
#ifndef SOFT_DIMS
#define SOFT_DIMS 3, 2, 3
#endif
template <int NS, int NU, int NY>
static int run(const double *flat, int N, int max_iter, double *u0, double *z1, double *sl, double *res, int *iters, double *w_out)
{
    using namespace mpc;
    SoftProb<NS, NU, NY> P;
    const double *f = flat;
    auto take = [&](double *dst, int n) { std::memcpy(dst, f, sizeof(double) * n); f += n; };
    P.N = N; P.max_iter = max_iter;
    take(&P.A[0][0], NS * NS); take(&P.B[0][0], NS * NU); take(&P.Q[0][0], NS * NS); take(&P.M[0][0], NS * NU); take(&P.R[0][0], NU * NU); take(&P.Pf[0][0], NS * NS);
    take(P.c, NS); take(P.z0, NS); take(P.zr, NS); take(P.zrN, NS); take(P.ur, NU); take(P.us, NU);
    take(P.ulo, NU); take(P.uhi, NU); take(P.zlo, NS); take(P.zhi, NS); take(P.zlo_e, NS); take(P.zhi_e, NS);
    take(&P.Cy[0][0], NY * NS); take(P.cy, NY); take(P.ymin, NY); take(P.ymax, NY); take(&P.Ws[0][0], 4 * NY * NY);
    using LY = SoftLayout<NS, NU, NY>;
    std::vector<double> ws((size_t)LY::FIELDS * N, 0.0);
    double u0_[NU], z1_[NS], sl_[2 * NY], res_[3];
    int it = 0;
    const int st = soft_solve<NS, NU, NY, 1>(P, ws.data(), u0_, z1_, sl_, res_, it);
    std::memcpy(u0, u0_, sizeof(u0_)); std::memcpy(z1, z1_, sizeof(z1_)); std::memcpy(sl, sl_, sizeof(sl_)); std::memcpy(res, res_, sizeof(res_));
    *iters = it;
    if (w_out) for (int k = 0; k < N; k++) { for (int i = 0; i < NU; i++) w_out[k * (NU + NS) + i] = ws[(size_t)k * LY::FIELDS + LY::U + i]; for (int i = 0; i < NS; i++) w_out[k * (NU + NS) + NU + i] = ws[(size_t)k * LY::FIELDS + LY::Z + i]; }
    return st;
}
extern "C" int soft_solve_host(const double *flat, int N, int max_iter, double *u0, double *z1, double *sl, double *res, int *iters, double *w_out)
{
    return run<SOFT_DIMS>(flat, N, max_iter, u0, z1, sl, res, iters, w_out);
}
