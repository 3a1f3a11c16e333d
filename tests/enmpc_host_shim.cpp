// Test infrastructure: the product's GENERATED model code (mpc-code_amd/econcodegen.py) and its Runge-Kutta sensitivity driver
// (mpc-code_amd/csrc/mpc_rk4s2.hpp) compiled for the HOST, so that tests/test_enmpc.py can compare the product's first and second
// derivatives with the oracle's complex-step ones without a GPU.  Built by the test with
//   g++ -O1 -std=c++17 -shared -fPIC -D__device__= -D__host__= -D__forceinline__=inline -DMPC_EC_MODEL_HEADER="..." tests/enmpc_host_shim.cpp
#define MPC_UNROLL
#include MPC_EC_MODEL_HEADER
#include "../mpc-code_amd/csrc/mpc_rk4s2.hpp"

using M = EcModel;

static M::Ctx ctx(const double *u, const double *d, const double *xs, const double *us)
{
    M::Ctx c;
    for (int i = 0; i < M::NU; i++) { c.u[i] = u[i]; c.us[i] = us ? us[i] : 0.0; }
    for (int i = 0; i < M::ND; i++) c.d[i] = d ? d[i] : 0.0;
    for (int i = 0; i < M::NX; i++) c.xs[i] = xs ? xs[i] : 0.0;
    return c;
}

extern "C" {
// shooting interval of the OCP: rows x (NX) then the cost quadrature; S [NR][NP], T [NR][NPP] row-major
void shim_ocp(const double *x, const double *u, const double *d, const double *xs, const double *us, double h, int quad, double *xn, double *S, double *T)
{
    using R = M::Ocp;
    const M::Ctx c = ctx(u, d, xs, us);
    enm::rk4_sens2<R>(x, c, 0.0, false, h, quad, xn, (double (*)[R::NP])S, (double (*)[R::NPP])T);
}
void shim_mdl(const double *x, const double *u, const double *d, double h, double *xn, double *S, double *T)
{
    using R = M::Mdl;
    const M::Ctx c = ctx(u, d, nullptr, nullptr);
    enm::rk4_sens2<R>(x, c, 0.0, true, h, M::MX, xn, (double (*)[R::NP])S, (double (*)[R::NPP])T);
}
void shim_mhe(const double *x, const double *u, double h, double *xn, double *S, double *T)
{
    using R = M::Mhe;
    const M::Ctx c = ctx(u, nullptr, nullptr, nullptr);
    enm::rk4_sens2<R>(x, c, 0.0, true, h, M::MX, xn, (double (*)[R::NP])S, (double (*)[R::NPP])T);
}
void shim_plant(const double *x, const double *u, double h, double *xn)
{
    const M::Ctx c = ctx(u, nullptr, nullptr, nullptr);
    enm::rk4_plain<M::Plant>(x, c, 0.0, true, h, M::MX, xn);
}
void shim_fss(const double *w, double *f, double *g, double *H) { M::fss(w, f, g, (double (*)[M::NX + M::NU + M::NY])H); }
void shim_vfin(const double *x, const double *xs, double *f, double *g, double *H) { M::vfin(x, xs, f, g, (double (*)[M::NX])H); }
void shim_cmhe(const double *wv, double *f, double *g, double *H) { M::cmhe(wv, 0.0, f, g, (double (*)[M::NW + M::NY])H); }
void shim_dims(int *out) { out[0] = M::NX; out[1] = M::NU; out[2] = M::NY; out[3] = M::ND; out[4] = M::NW; out[5] = M::MX; }
}
