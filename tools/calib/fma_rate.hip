// Issue rate and dependent latency of v_fma_f64 on one SIMD: CH independent accumulator chains per lane, W waves per
// workgroup (W = 1: one wave alone on a SIMD; W = 8: two waves per SIMD).  Prints shader-clock cycles per FMA per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o fma_rate tools/calib/fma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH>
__global__ void fma_chain(double *out, unsigned long long *cyc, int iters, double a, double b) {
    double acc[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) acc[i] = threadIdx.x * 1e-3 + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) acc[i] = __builtin_fma(acc[i], a, b);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CH>
void run(int waves, int active_lanes) {
    const int iters = 2000;
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 8 * 256 * sizeof(double));
    hipMalloc(&cyc, 8 * 256 * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(fma_chain<CH>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 0.999, 1e-3);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(8 * 256);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0.0;
    for (int i = 0; i < 256 * waves; i++) mean += (double)h[i];
    mean /= 256 * waves;
    printf("chains=%d waves/WG=%d: %.2f cycles per FMA per wave (%.2f per SIMD slot)\n", CH, waves, mean / (iters * 8.0 * CH),
           mean / (iters * 8.0 * CH) / (waves > 4 ? waves / 4.0 : 1.0));
    hipFree(out); hipFree(cyc);
    (void)active_lanes;
}

int main() {
    for (int w : {1, 4, 8, 16}) {
        run<1>(w, 64); run<2>(w, 64); run<4>(w, 64); run<8>(w, 64); run<16>(w, 64);
    }
    return 0;
}
