// Small dense helpers that are plain C++ (no wave intrinsics): included by mpc_device.hpp inside namespace mpc, after frcp and MPC_UNROLL are defined -
// and by the test-side wave emulator (tests/wave_emu), which brings its own frcp.  No include guard on purpose: it is a fragment of the enclosing namespace.
// symmetric inverse by LDL' with reciprocal pivots (no sqrt, no IEEE division); false if a pivot is not > 0
template <int n>
__device__ __forceinline__ bool sym_inverse(double (&a)[n][n])
{
    double l[n][n], d[n], dinv[n];
    bool ok = true;
    MPC_UNROLL for (int j = 0; j < n; j++) {
        double dj = a[j][j];
        MPC_UNROLL for (int k = 0; k < j; k++) dj -= l[j][k] * l[j][k] * d[k];
        ok = ok && (dj > 0.0);
        d[j] = dj;
        dinv[j] = frcp(dj);
        MPC_UNROLL for (int i = j + 1; i < n; i++) {
            double v = a[i][j];
            MPC_UNROLL for (int k = 0; k < j; k++) v -= l[i][k] * l[j][k] * d[k];
            l[i][j] = v * dinv[j];
        }
    }
    MPC_UNROLL for (int col = 0; col < n; col++) {      // inverse = L^-T D^-1 L^-1, column by column
        double y[n];
        MPC_UNROLL for (int i = 0; i < n; i++) {
            double v = (i == col) ? 1.0 : 0.0;
            MPC_UNROLL for (int k = 0; k < i; k++) v -= l[i][k] * y[k];
            y[i] = v;
        }
        MPC_UNROLL for (int i = n - 1; i >= 0; i--) {
            double v = y[i] * dinv[i];
            MPC_UNROLL for (int k = i + 1; k < n; k++) v -= l[k][i] * a[k][col];
            a[i][col] = v;
        }
    }
    return ok;
}
