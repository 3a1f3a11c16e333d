// Issue rate of v_fma_f64 on one SIMD as a function of where its operands live.  CH independent accumulator chains per lane,
// W waves per workgroup (W = 1: one wave alone on a SIMD; W = 8: two waves per SIMD).  MODE 0: acc = fma(acc, s, s) with two
// scalar-register operands; 1: fma(acc, v, s); 2: fma(acc, v, v) with all three operands in vector registers (what a small
// dense matrix product with per-lane matrices looks like).  Prints shader-clock cycles per FMA per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o fma_rate tools/calib/fma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH, int MODE>
__global__ void fma_chain(double *out, unsigned long long *cyc, int iters, double a, double b)
{
    double acc[CH], x[CH], y[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        acc[i] = threadIdx.x * 1e-3 + i; x[i] = a + 1e-9 * (threadIdx.x + i); y[i] = b + 1e-9 * (threadIdx.x + 2 * i);
        asm volatile("" : "+v"(x[i]), "+v"(y[i]));
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (MODE == 0) acc[i] = __builtin_fma(acc[i], a, b);
                else if (MODE == 1) acc[i] = __builtin_fma(acc[i], x[i], b);
                else acc[i] = __builtin_fma(acc[i], x[i], y[(i + r) % CH]);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CH, int MODE>
void run(int waves)
{
    const int iters = 2000;
    double *out; unsigned long long *cyc;
    if (hipMalloc(&out, 64 * 8 * 256 * sizeof(double)) != hipSuccess || hipMalloc(&cyc, 8 * 256 * sizeof(unsigned long long)) != hipSuccess) return;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((fma_chain<CH, MODE>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 0.999, 1e-3);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(8 * 256);
    (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0.0;
    for (int i = 0; i < 256 * waves; i++) mean += (double)h[i];
    mean /= 256 * waves;
    printf("mode=%d chains=%2d waves/WG=%d: %.2f cycles per FMA per wave\n", MODE, CH, waves, mean / (iters * 8.0 * CH));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    for (int w : {1, 8}) {
        run<1, 0>(w); run<4, 0>(w); run<16, 0>(w);
        run<1, 1>(w); run<4, 1>(w); run<16, 1>(w);
        run<1, 2>(w); run<4, 2>(w); run<16, 2>(w);
    }
    return 0;
}
