// Operand layout of v_mfma_f64_4x4x4f64 (four independent 4x4x4 products per wave, one element of each operand per lane):
// one-hot probes.  For lanes la (A) and lb (B) inside block 0 the product has exactly one non-zero D element iff the k indices
// agree; the lane where it shows up gives (i, j).  Prints, for block 0, k(la), i(la), k(lb), j(lb) and the D lane map.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(double *out)      // out[la][lb][64]
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 16; la++)
        for (int lb = 0; lb < 16; lb++) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 16 + lb) * 64 + lane] = d;
        }
}

int main()
{
    double *d; static double h[16 * 16 * 64];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("D lane for (A lane la, B lane lb), block 0; '.' = no output (k indices differ)\n     ");
    for (int lb = 0; lb < 16; lb++) printf("%3d", lb);
    printf("\n");
    for (int la = 0; la < 16; la++) {
        printf("la=%2d", la);
        for (int lb = 0; lb < 16; lb++) {
            int hit = -1, cnt = 0;
            for (int l = 0; l < 64; l++) if (h[(la * 16 + lb) * 64 + l] != 0.0) { hit = l; cnt++; }
            if (cnt == 0) printf("  ."); else if (cnt == 1) printf("%3d", hit); else printf("  *");
        }
        printf("\n");
    }
    return 0;
}
