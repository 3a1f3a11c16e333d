/*
 * A C host of the economic path's per-call seam (include/mpc_enmpc.h): the reference's loop body MPC_code.py:485-827 for Ex_ENMPC.py, written the way a
 * maintainer of a compiled front end would - per step the measurement, enmpc_mhe_update (defEstimator(..., 'mhe'), :583-641), enmpc_target_solve
 * (solver_ss(...), :704-709), enmpc_ocp_solve (solver(...), :776-781) and the plant (here the device's: enmpc_plant_step).
 *
 * The library is per model (mpc-code_amd/econcodegen.py traces the Ex-file's functions and compiles them in): this program links against the one of the
 * shipped example, whose path the build is given.  Problem data below are the Ex-file's (Ex_ENMPC.py:20,100-135,188-193,255).
 *
 *   gcc -std=c99 -Wall -Wextra -pedantic -I include mpc-code_amd/examples/c_host/enmpc_host.c mpc-code_amd/csrc/jit/libmpc_enmpc_<hash>.so -o enmpc_host -lm
 *   ./enmpc_host [steps]        prints "step k: u ... xs ... us ... status d/s/m iters d/s/m" per step
 *   exit code 0: ran; 2: the library refuses (no GPU: there is no CPU path); 1: a call failed
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mpc_enmpc.h"

#define B 2
#define NX 2
#define NU 1
#define ND 2
#define NE 4

static int check(int rc, const char *what)
{
    if (rc) fprintf(stderr, "%s failed (%d): %s\n", what, rc, enmpc_last_error());
    return rc;
}

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 3;
    const double inf = INFINITY;
    const double umin[NU] = {0.0}, umax[NU] = {2.0}, xmin[NX] = {0.0, 0.0}, xmax[NX] = {1.0, 1.0}, ylo[2] = {-inf, -inf}, yhi[2] = {inf, inf};
    const double elo[NE] = {0.0, 0.0, -inf, -inf}, ehi[NE] = {1.0, 1.0, inf, inf};
    const double Bd[NX * ND] = {0, 0, 0, 0}, Cd[2 * ND] = {1, 0, 0, 1}, G[NE * NE] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, P0[NE * NE] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const double x0_m[NX] = {1.2, 0.5}, u0[NU] = {0.0};
    enmpc_desc d = {0};
    d.nx = NX; d.nu = NU; d.ny = 2; d.nd = ND; d.nxp = 2; d.nw = NE;
    d.N = 25; d.N_mhe = 10; d.max_iter = 200; d.quad_steps = 20; d.device = 0; d.mhe_update = 0; d.h = 2.0; d.tol = 1e-8; d.tol_mhe = 1e-10;
    d.umin = umin; d.umax = umax; d.xmin = xmin; d.xmax = xmax;
    d.umin_ss = umin; d.umax_ss = umax; d.xmin_ss = xmin; d.xmax_ss = xmax; d.ymin_ss = ylo; d.ymax_ss = yhi;
    d.xmin_mhe = elo; d.xmax_mhe = ehi; d.dmin = NULL; d.dmax = NULL;
    d.Bd = Bd; d.Cd = Cd; d.G_mhe = G; d.P0 = P0; d.x0_m = x0_m; d.u0 = u0;
    enmpc_handle *h = NULL;
    if (enmpc_create(&d, &h)) { fprintf(stderr, "enmpc_create: %s\n", enmpc_last_error()); return 2; }
    printf("library: %s\n", enmpc_build_info());
    /* two instances: the shipped start and another plant state; model state, input, estimator prior as the Ex-file's */
    double x_p[B * NX] = {0.9, 0.1, 0.6, 0.3}, xhat[B * NX] = {1.2, 0.5, 1.2, 0.5}, dhat[B * ND] = {0, 0, 0, 0}, u[B * NU] = {0, 0}, x_bar[B * NE] = {1.2, 0.5, 0, 0, 1.2, 0.5, 0, 0};
    double xs[B * NX], us[B * NU], xes[B * NE];
    int32_t st_d[B], st_s[B], st_m[B], it_d[B], it_s[B], it_m[B];
    if (check(enmpc_alloc(h, B, 1), "enmpc_alloc") || check(enmpc_set_state(h, x_p, xhat, dhat, u, x_bar), "enmpc_set_state")) return 1;
    for (int k = 0; k < steps; k++) {
        /* y_act = Fy_p(x_p): with StateFeedback the measurement is the plant state (Utilities.py:84-86) */
        if (check(enmpc_mhe_update(h, x_p, u, xhat, dhat, xes, st_m, it_m), "enmpc_mhe_update")) return 1;
        if (check(enmpc_target_solve(h, dhat, xs, us, st_s, it_s), "enmpc_target_solve")) return 1;
        if (check(enmpc_ocp_solve(h, xhat, dhat, xs, us, u, xhat, st_d, it_d), "enmpc_ocp_solve")) return 1;      /* xhat <- the optimiser's next state (:799) */
        if (check(enmpc_plant_step(h, u, x_p), "enmpc_plant_step")) return 1;
        for (int b = 0; b < B; b++)
            printf("step %d instance %d: u %.12f xs %.12f %.12f us %.12f status %d/%d/%d iters %d/%d/%d\n", k, b, u[b], xs[NX * b], xs[NX * b + 1], us[b],
                   (int)st_d[b], (int)st_s[b], (int)st_m[b], (int)it_d[b], (int)it_s[b], (int)it_m[b]);
    }
    enmpc_destroy(h);
    return 0;
}
