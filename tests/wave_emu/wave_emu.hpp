// TEST INFRASTRUCTURE: namespace mpc's wave primitives (mpc-code_amd/csrc/mpc_tp.hpp, mpc_device.hpp) on the host fibers of tests/wave_emu/include/hip/hip_runtime.h.
// Included by mpc_enmpc.hpp in place of mpc_tp.hpp when the kernel source is compiled for the CPU test suite (-DEC_WAVE_EMU); see that header's note.
// The reductions run the same DPP steps in the same order as the device code, so sums come out with the device's association of the additions.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace mpc {
#define MPC_UNROLL
#define MPC_STAMP(slot) do { } while (0)
#define MPC_STAMP_INIT
#define MPC_STAMP_RESET
inline double dmax(double a, double b) { return __builtin_fmax(a, b); }
inline double dmin(double a, double b) { return __builtin_fmin(a, b); }
inline bool fin(double a) { return fabs(a) < 1.0e300; }
inline double frcp(double x) { return 1.0 / x; }      // (the device: v_rcp_f64 + two Newton steps, about an ulp)
#include "mpc_sym.hpp"

template <int CTRL, int ROW_MASK>
inline double dpp_move(double old, double v)
{
    const double *s = emu::gather(v);
    const int i = threadIdx.x, src = emu::dpp_src(CTRL, i);
    if (!((ROW_MASK >> (i >> 4)) & 1) || src < 0 || src >= emu::g_wave.n) return old;
    return s[src];
}
inline double wave_up1(double old, double v) { return dpp_move<0x138, 0xF>(old, v); }
inline double wave_dn1(double old, double v) { return dpp_move<0x130, 0xF>(old, v); }
inline double lane_of(double v, int l) { const double *s = emu::gather(v); return s[l]; }
inline double lane63(double v) { return lane_of(v, 63); }
inline double wave_sum(double v)
{
    v += dpp_move<0xB1, 0xF>(0.0, v); v += dpp_move<0x4E, 0xF>(0.0, v); v += dpp_move<0x141, 0xF>(0.0, v); v += dpp_move<0x140, 0xF>(0.0, v);
    v += dpp_move<0x142, 0xA>(0.0, v); v += dpp_move<0x143, 0xC>(0.0, v);
    return lane63(v);
}
inline double wave_max(double v)
{
    v = dmax(v, dpp_move<0xB1, 0xF>(v, v)); v = dmax(v, dpp_move<0x4E, 0xF>(v, v)); v = dmax(v, dpp_move<0x141, 0xF>(v, v)); v = dmax(v, dpp_move<0x140, 0xF>(v, v));
    v = dmax(v, dpp_move<0x142, 0xA>(v, v)); v = dmax(v, dpp_move<0x143, 0xC>(v, v));
    return lane63(v);
}
inline double half_sum(double v)
{
    v += dpp_move<0xB1, 0xF>(0.0, v); v += dpp_move<0x4E, 0xF>(0.0, v); v += dpp_move<0x141, 0xF>(0.0, v); v += dpp_move<0x140, 0xF>(0.0, v);
    v += dpp_move<0x142, 0xA>(0.0, v);
    const double lo = lane_of(v, 31), hi = lane_of(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}
inline double half_max(double v)
{
    v = dmax(v, dpp_move<0xB1, 0xF>(v, v)); v = dmax(v, dpp_move<0x4E, 0xF>(v, v)); v = dmax(v, dpp_move<0x141, 0xF>(v, v)); v = dmax(v, dpp_move<0x140, 0xF>(v, v));
    v = dmax(v, dpp_move<0x142, 0xA>(v, v));
    const double lo = lane_of(v, 31), hi = lane_of(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}
inline double uni(double v) { return v; }      // (the device: readfirstlane of a wave-uniform value)
}  // namespace mpc
