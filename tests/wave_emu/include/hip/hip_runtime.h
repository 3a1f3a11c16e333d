// TEST INFRASTRUCTURE (never shipped, never loaded by the product): a wavefront of the MI355X on host fibers, and just enough of the HIP runtime's names,
// so that mpc-code_amd/csrc/mpc_enmpc.hip - kernels AND host side, unchanged - compiles with g++ into a library that exports the C-ABI of include/mpc_enmpc.h.
// The CPU test suite (tests/test_wave_emu.py) loads that library through the product's own ctypes binding and compares the KERNEL SOURCE, lane by lane, with the
// oracle: the lanes of a workgroup are 64 fibers of one host thread that run in lockstep from one cross-lane operation to the next (DPP moves, v_readlane,
// ds_bpermute, votes: mpc-code_amd/csrc/mpc_tp.hpp names them; their lane patterns are restated here from the ISA's definition of the DPP controls).
// What this is NOT: a CPU fallback.  The product's loader (mpc-code_amd/enmpc.py, econcodegen.py) never builds or opens it; `g++ -DEC_WAVE_EMU` is a test recipe.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <sys/mman.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <time.h>

#define __device__
#define __host__
#define __global__ static
#define __forceinline__ inline
#define __shared__ static __attribute__((section("emu_lds")))      // (one section: EMU_POISON fills it before every launch - LDS holds what the last kernel left)
#define __launch_bounds__(...)

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

extern "C" char __start_emu_lds[] __attribute__((weak)), __stop_emu_lds[] __attribute__((weak));      // (absent from a translation unit without __shared__ arrays)
inline int emu_poison() { static const int p = getenv("EMU_POISON") ? atoi(getenv("EMU_POISON")) : 0; return p; }      // 1: 0xFF bytes (NaN doubles, -1 integers); 2: finite random doubles
inline void emu_fill(void *q, size_t n)
{
    if (emu_poison() == 1) { std::memset(q, 0xFF, n); return; }
    static unsigned long long st = 88172645463325252ull;
    for (size_t i = 0; i + 8 <= n; i += 8) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; const double v = ((double)(st >> 11) / 9007199254740992.0 - 0.5) * 20.0; std::memcpy((char *)q + i, &v, 8); }
}
namespace emu {
struct Idx { int x = 0, y = 0, z = 0; };
struct Fiber { void *sp = nullptr; void *stack = nullptr; bool alive = false; unsigned long xcnt = 0; };
struct Wave {
    Fiber f[64]; void *main_sp = nullptr; int n = 0, cur = -1;
    double slot[2][64]; unsigned long long islot[2][64];
    const std::function<void()> *body = nullptr;
    bool independent = false;      // lane = instance kernels: no lane talks to another, control flow diverges freely (votes answer for the lane itself)
};
inline Wave g_wave;
inline Idx g_tid, g_bid, g_bdim, g_gdim;
constexpr size_t kStack = 4u << 20;

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n");

inline int next_alive(int from)
{
    for (int d = 1; d <= g_wave.n; d++) { const int j = (from + d) % g_wave.n; if (g_wave.f[j].alive) return j; }
    return -1;
}
inline void go(int to)      // (from the fiber that is running)
{
    const int me = g_wave.cur;
    g_wave.cur = to; g_tid.x = to;
    emu_switch(&g_wave.f[me].sp, g_wave.f[to].sp);
}
// every live lane has reached this point when a lane returns from it (round-robin: the lanes of a wave execute the same sequence of cross-lane operations)
inline void sync()
{
    const int to = next_alive(g_wave.cur);
    if (to >= 0 && to != g_wave.cur) go(to);
}
inline void fiber_entry()
{
    (*g_wave.body)();
    const int me = g_wave.cur;
    g_wave.f[me].alive = false;
    const int to = next_alive(me);
    void *dummy;
    if (to >= 0) { g_wave.cur = to; g_tid.x = to; emu_switch(&dummy, g_wave.f[to].sp); }
    else { g_wave.cur = -1; emu_switch(&dummy, g_wave.main_sp); }
    abort();
}
inline void run_block(int nlanes, const std::function<void()> &body)
{
    Wave &w = g_wave;
    w.n = nlanes; w.body = &body; w.independent = false;
    for (int i = 0; i < nlanes; i++) {
        Fiber &f = w.f[i];
        if (!f.stack) { f.stack = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0); if (f.stack == MAP_FAILED) abort(); }
        void **top = (void **)((char *)f.stack + kStack);      // 16-byte aligned
        top[-1] = nullptr;                       // where a caller's return address would sit
        top[-2] = (void *)&fiber_entry;          // popped by emu_switch's ret
        for (int r = 3; r <= 8; r++) top[-r] = nullptr;      // rbp rbx r12 r13 r14 r15
        f.sp = (void *)(top - 8); f.alive = true; f.xcnt = 0;
    }
    w.cur = 0; g_tid.x = 0;
    emu_switch(&w.main_sp, w.f[0].sp);
}
// EMU_WATCHDOG=<seconds>: a launch that runs longer prints the native stack of the lane that is running and aborts (a lane pattern that never meets its partners)
inline void watchdog_fire(int) { void *bt[64]; const int n = backtrace(bt, 64); backtrace_symbols_fd(bt, n, 2); _exit(99); }
inline void launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
    static const int wd = getenv("EMU_WATCHDOG") ? atoi(getenv("EMU_WATCHDOG")) : 0;
    if (wd > 0) { signal(SIGALRM, watchdog_fire); alarm(wd); }
    if (block.x > 64 || block.y != 1 || grid.y != 1) abort();
    if (emu_poison() && __start_emu_lds) emu_fill(__start_emu_lds, (size_t)(__stop_emu_lds - __start_emu_lds));
    g_bdim.x = (int)block.x; g_gdim.x = (int)grid.x;
    for (unsigned b = 0; b < grid.x; b++) { g_bid.x = (int)b; run_block((int)block.x, body); }
    if (wd > 0) alarm(0);
}
// ---- cross-lane operations: every lane publishes its value, then reads the lanes it wants (two buffers: a lane may publish its next value before a slower
// lane has read this one) ----
inline const double *gather(double v)
{
    Wave &w = g_wave; Fiber &f = w.f[w.cur];
    const int par = (int)(f.xcnt++ & 1);
    w.slot[par][w.cur] = v;
    sync();
    return w.slot[par];
}
// the votes of the lanes that take part in this vote (a lane that has left the kernel earlier does not; one that leaves after voting does), and who they are
inline unsigned long long ballot(bool p, unsigned long long *voters = nullptr)
{
    Wave &w = g_wave; Fiber &f = w.f[w.cur];
    const unsigned long c = f.xcnt++;
    const int par = (int)(c & 1);
    w.islot[par][w.cur] = p ? 1ull : 0ull;
    sync();
    unsigned long long m = 0, v = 0;
    for (int i = 0; i < w.n; i++) if (w.f[i].xcnt > c) { v |= 1ull << i; if (w.islot[par][i]) m |= 1ull << i; }
    if (voters) *voters = v;
    return m;
}
// source lane of a DPP control for lane i of a 64-lane wave; -1: no source (bound_ctrl off: the destination keeps `old`)
inline int dpp_src(int ctrl, int i)
{
    if (ctrl <= 0xFF) return (i & ~3) | ((ctrl >> (2 * (i & 3))) & 3);      // quad_perm
    if (ctrl == 0x138) return i >= 1 ? i - 1 : -1;                           // wave_shr:1
    if (ctrl == 0x130) return i <= 62 ? i + 1 : -1;                          // wave_shl:1
    if (ctrl == 0x140) return (i & ~15) | (15 - (i & 15));                   // row_mirror
    if (ctrl == 0x141) return (i & ~7) | (7 - (i & 7));                      // row_half_mirror
    if (ctrl == 0x142) return i >= 16 ? (i & ~15) - 1 : -1;                  // row_bcast:15 (lane 15 of the row before)
    if (ctrl == 0x143) return i >= 32 ? 31 : -1;                             // row_bcast:31
    abort();
}
struct IndependentScope { IndependentScope() { g_wave.independent = true; } };
}  // namespace emu

#define threadIdx (emu::g_tid)
#define blockIdx (emu::g_bid)
#define blockDim (emu::g_bdim)
#define gridDim (emu::g_gdim)
#define EC_LANE_INDEPENDENT_KERNEL emu::IndependentScope ec_independent_scope_;

inline int __any(int p) { if (emu::g_wave.independent) return p != 0; return emu::ballot(p != 0) != 0ull; }
inline int __all(int p)
{
    if (emu::g_wave.independent) return p != 0;
    unsigned long long voters = 0;
    const unsigned long long m = emu::ballot(p != 0, &voters);
    return m == voters;
}
inline unsigned long long __ballot(int p) { return emu::ballot(p != 0); }
inline double __shfl(double v, int src) { const double *s = emu::gather(v); return s[src & 63]; }

// ---- the HIP runtime's names the host side of mpc_enmpc.hip uses: device memory is host memory, streams and events are tokens, a launch runs now ----
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
struct hipDeviceProp_t { int multiProcessorCount = 256; };
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
// EMU_POISON=1 | 2: device allocations and (before every launch) the __shared__ arrays are filled with 0xFF bytes - NaN doubles, -1 integers - or with finite random doubles
// instead of zeros: a read before the first write, which zeroed host memory hides and a GPU does not, shows
inline hipError_t hipMalloc(void **p, size_t n) { *p = std::calloc(n ? n : 1, 1); if (*p && emu_poison()) emu_fill(*p, n); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < h; r++) std::memcpy((char *)d + r * dp, (const char *)s + r * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (void *)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// (an event is the host clock's reading when it was recorded)
inline double emu_now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = std::calloc(1, sizeof(double)); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { *(double *)e = emu_now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*(double *)b - *(double *)a) + 1e-6f; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) emu::launch((grid), (block), [&]() { kern(__VA_ARGS__); })
