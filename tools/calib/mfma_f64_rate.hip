// Issue rate and dependent latency of v_mfma_f64_4x4x4f64: CH independent accumulator chains per wave, 1..8 waves of one workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH, bool DEP_A>
__global__ void k(double *out, unsigned long long *cyc, int iters)
{
    double acc[CH], a = 1.0 + 1e-9 * threadIdx.x, b = 1e-3 * (threadIdx.x & 3);
#pragma unroll
    for (int i = 0; i < CH; i++) acc[i] = 1e-3 * threadIdx.x + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) acc[i] = DEP_A ? __builtin_amdgcn_mfma_f64_4x4x4f64(acc[i], b, 0.0, 0, 0, 0) : __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += acc[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int CH, bool DEP_A>
void run(int waves = 1)
{
    double *out; unsigned long long *cyc, h[16];
    if (hipMalloc(&out, 1024 * sizeof(double)) != hipSuccess || hipMalloc(&cyc, 128) != hipSuccess) return;
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<CH, DEP_A>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
    (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    double m = 0.0;
    for (int i = 0; i < waves; i++) m += (double)h[i] / waves;
    printf("waves=%d chains=%d result feeds %s: %.1f cycles per MFMA per wave\n", waves, CH, DEP_A ? "the A operand" : "the accumulator", m / (iters * 8.0 * CH));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<1, false>(); run<2, false>(); run<4, false>(); run<8, false>();
    run<1, true>(); run<2, true>(); run<4, true>();
    for (int w : {2, 4, 8}) { run<1, true>(w); run<4, false>(w); }      // several waves of one workgroup: is the pipe shared?
    return 0;
}
