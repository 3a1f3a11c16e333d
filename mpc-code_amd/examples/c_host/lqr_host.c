/*
 * A C host of libmpc_amd.so (include/mpc_amd.h): what a maintainer of a compiled front end would write.
 *
 * Problem: double integrator x+ = [[1, 1], [0, 1]] x + [0.5; 1] u, y = x_0 + d, cost x'x + u'u, |u| <= 1, N = 20 (a dimension set of the
 * default library).  The terminal weight is the solution of the discrete Riccati equation (the reference calls SciPy for it,
 * Utilities.py:409; here a fixed-point iteration), so for states whose optimal input stays inside the bounds the first move of the OCP is the
 * LQR law u0 = -K x for ANY horizon - the known answer of SURVEY.md section 8c(2).  The program solves a batch of such OCPs through
 * mpc_ocp_solve and checks them against -K x, then one OCP whose input saturates.
 *
 *   gcc -std=c99 -Wall -Wextra -pedantic -I include mpc-code_amd/examples/c_host/lqr_host.c -o lqr_host -L mpc-code_amd/csrc -lmpc_amd -lm
 *   LD_LIBRARY_PATH=mpc-code_amd/csrc ./lqr_host          exit code 0: all right; 2: the library refuses (no GPU: there is no CPU path); 1: wrong numbers
 */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "mpc_amd.h"

#define NXS 2
#define BATCH 5

int main(void)
{
    const double A[4] = {1, 1, 0, 1}, Bm[2] = {0.5, 1}, C[2] = {1, 0}, Bd[2] = {0, 0}, Cd[1] = {1}, z2[2] = {0, 0}, z1[1] = {0};
    const double Q[4] = {1, 0, 0, 1}, R[1] = {1}, Qss[1] = {1}, Rss[1] = {0};
    const double inf = INFINITY, umin[1] = {-1}, umax[1] = {1}, xlo[2] = {-inf, -inf}, xhi[2] = {inf, inf}, ylo[1] = {-inf}, yhi[1] = {inf};
    double P[4] = {1, 0, 0, 1}, K[2];
    for (int it = 0; it < 500; it++) {      /* P = Q + A'PA - A'PB (R + B'PB)^-1 B'PA */
        double PA[4], PB[2], s = R[0], AtPB[2], AtPA[4];
        for (int i = 0; i < 2; i++) { PB[i] = P[2 * i] * Bm[0] + P[2 * i + 1] * Bm[1]; for (int j = 0; j < 2; j++) PA[2 * i + j] = P[2 * i] * A[j] + P[2 * i + 1] * A[2 + j]; }
        s += Bm[0] * PB[0] + Bm[1] * PB[1];
        for (int i = 0; i < 2; i++) { AtPB[i] = A[i] * PB[0] + A[2 + i] * PB[1]; for (int j = 0; j < 2; j++) AtPA[2 * i + j] = A[i] * PA[j] + A[2 + i] * PA[2 + j]; }
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) P[2 * i + j] = Q[2 * i + j] + AtPA[2 * i + j] - AtPB[i] * AtPB[j] / s;
        K[0] = AtPB[0] / s; K[1] = AtPB[1] / s;
    }
    mpc_lin_desc d;
    memset(&d, 0, sizeof d);      /* zero first: fields a later version of the header adds (soft constraints ...) then mean "off" */
    mpc_handle *h = NULL;
    d.nx = 2; d.nu = 1; d.ny = 1; d.nd = 1; d.nxp = 2; d.N = 20; d.du_form = 0; d.duss_form = 0; d.y_bounded = 0; d.estimator = MPC_EST_NONE; d.max_iter = 100; d.device = 0;
    d.A = A; d.B = Bm; d.C = C; d.Bd = Bd; d.Cd = Cd; d.fx_const = z2; d.fy_const = z1; d.Ap = A; d.Bp = Bm; d.Cp = C;
    d.Q = Q; d.R = R; d.P = P; d.Qss = Qss; d.Rss = Rss;
    d.umin = umin; d.umax = umax; d.xmin = xlo; d.xmax = xhi; d.ymin = ylo; d.ymax = yhi;
    d.umin_ss = umin; d.umax_ss = umax; d.xmin_ss = xlo; d.xmax_ss = xhi; d.ymin_ss = ylo; d.ymax_ss = yhi;
    d.dmin = NULL; d.dmax = NULL; d.Q_kf = NULL; d.R_kf = NULL; d.K = NULL; d.Dumin = NULL; d.Dumax = NULL;
    d.term_cons = 0; d.nl_plant = 0; d.h_sample = 1.0;
    if (mpc_lin_create(&d, &h)) { fprintf(stderr, "mpc_lin_create: %s\n", mpc_last_error()); return 2; }
    /* four states inside the unconstrained region and one whose first move saturates */
    const double xhat[BATCH * NXS] = {0.2, 0.1, -0.3, 0.05, 0.1, -0.2, 0.0, 0.3, 3.0, 1.0};
    double xs[BATCH * NXS] = {0}, us[BATCH] = {0}, dh[BATCH] = {0}, up[BATCH] = {0}, u[BATCH], xn[BATCH * NXS];
    int32_t st[BATCH], it[BATCH];
    if (mpc_ocp_solve(h, BATCH, xhat, xs, us, dh, up, NULL, NULL, NULL, u, xn, st, it, NULL)) { fprintf(stderr, "mpc_ocp_solve: %s\n", mpc_last_error()); mpc_destroy(h); return 2; }
    int bad = 0;
    for (int b = 0; b < BATCH; b++) {
        const double lqr = -(K[0] * xhat[2 * b] + K[1] * xhat[2 * b + 1]);
        const double expect = b < 4 ? lqr : -1.0;
        printf("instance %d: status %d, %2d iterations, u0 = % .9f (LQR law % .9f), x1 = [% .6f % .6f]\n", b, (int)st[b], (int)it[b], u[b], lqr, xn[2 * b], xn[2 * b + 1]);
        if (st[b] != MPC_STATUS_SOLVED || fabs(u[b] - expect) > 1e-7) bad = 1;
        if (fabs(xn[2 * b] - (xhat[2 * b] + xhat[2 * b + 1] + 0.5 * u[b])) > 1e-9 || fabs(xn[2 * b + 1] - (xhat[2 * b + 1] + u[b])) > 1e-9) bad = 1;      /* x1 = A x + B u0 */
    }
    mpc_destroy(h);
    printf(bad ? "WRONG\n" : "ok: first moves equal the LQR law where it is feasible, the bound where it is not\n");
    return bad;
}
