// What wave_shr:1 / wave_shl:1 (DPP controls 0x138 / 0x130) move on gfx950: prints, per lane, the lane whose value arrived.
// hipcc --offload-arch=gfx950 -O2 -o dpp_wave_shift dpp_wave_shift.hip && ./dpp_wave_shift
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *p)
{
    const int v = threadIdx.x;
    p[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);
    p[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false);
}
int main()
{
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shr:1 (0x138):"); for (int i = 0; i < 64; i++) printf(" %d", h[i]); printf("\n");
    printf("wave_shl:1 (0x130):"); for (int i = 0; i < 64; i++) printf(" %d", h[64 + i]); printf("\n");
    return 0;
}
