// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths this library uses (the guide calibrates 16 B per lane only):
// streams NBYTES through a copy kernel with 8-byte and with 16-byte accesses per lane; compare the counters with 2 x NBYTES moved.
//   hipcc --offload-arch=gfx950 -O3 -o build_diag/fetch_calib tools/calib/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -d out -- build_diag/fetch_calib        (and a second pass with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ void copy_b8(const double *__restrict__ in, double *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy_b16(const v2d *__restrict__ in, v2d *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main()
{
    const size_t bytes = (size_t)1 << 30;      // 1 GiB in, 1 GiB out: far beyond the 256 MiB Infinity Cache
    double *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipLaunchKernelGGL(copy_b8, dim3(4096), dim3(256), 0, 0, a, b, bytes / 8);
    hipLaunchKernelGGL(copy_b16, dim3(4096), dim3(256), 0, 0, (const v2d *)a, (v2d *)b, bytes / 16);
    hipDeviceSynchronize();
    printf("copied %zu bytes twice (8 B and 16 B per lane)\n", bytes);
    return 0;
}
