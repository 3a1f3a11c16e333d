/* The three public headers are plain C99 and self-contained: every entry point they declare is referenced here (compiled, never run). */
#include "mpc_amd.h"
#include "mpc_nmpc.h"
#include "mpc_enmpc.h"

typedef void (*fn)(void);
fn mpc_entry_points[] = {
    (fn)mpc_lin_create, (fn)mpc_destroy, (fn)mpc_last_error, (fn)mpc_ocp_solve, (fn)mpc_target_solve, (fn)mpc_kf_update, (fn)mpc_set_model_offsets,
    (fn)mpc_loop_alloc, (fn)mpc_loop_set_state, (fn)mpc_loop_set_schedule, (fn)mpc_loop_set_model_schedule, (fn)mpc_loop_run, (fn)mpc_loop_sync, (fn)mpc_loop_get_log,
    (fn)nmpc_create, (fn)nmpc_destroy, (fn)nmpc_run, (fn)nmpc_set_groups,
    (fn)enmpc_create, (fn)enmpc_destroy, (fn)enmpc_run, (fn)enmpc_set_groups, (fn)enmpc_set_state,
};
