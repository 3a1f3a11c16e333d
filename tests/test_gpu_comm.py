"""The collective inside the library (mpc_comm_*, mpc_allgather_u, mpc_allgather_log) on the one GPU of the
box: a communicator of one rank goes through the same RCCL calls as N > 1 (what `bench.py --force-dist` runs).  The N > 1 host
side (partition, rendezvous, stitching) is in test_shard_gloo.py."""
import os
import tempfile

import numpy as np
import pytest

from conftest import gpu_available

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gpu_available(), reason="needs the MI355X")]


def test_single_rank_communicator_gathers_what_the_kernel_logged_and_keeps_stdout_clean(pkg):
    from mpc_code_amd import capi
    p = pkg.load_problem(pkg.example_path("cstr_lmpc.py"))
    s = capi.Solver(p, device=0)
    B, K = 192, 6
    x0 = np.random.default_rng(5).uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(B, 3))
    try:
        # RCCL prints a version banner when NCCL_DEBUG asks for it (the benchmark boxes do): it must not reach stdout, where bench.py's
        # one JSON line goes - file descriptor 1 is captured while the communicator is set up
        os.environ.setdefault("NCCL_DEBUG", "VERSION")
        with tempfile.TemporaryFile() as cap:
            saved = os.dup(1)
            os.dup2(cap.fileno(), 1)
            try:
                uid = s.comm_unique_id()
                s.comm_init(0, 1, uid)
            finally:
                os.dup2(saved, 1); os.close(saved)
            cap.seek(0)
            assert cap.read() == b""
        assert s.comm_rank() == (0, 1)
        s.loop_alloc(B, K, capi.LOG_ALL); s.loop_set_state(x0, x0); s.loop_set_schedule(p.schedules(K))
        s.loop_run(0, K); s.loop_sync()
        U = s.loop_get_log("U")[:K]
        g = s.allgather_log("U", 0, K)
        assert g.shape == (1, K, B, p.nu) and np.array_equal(g[0], U)
        gx = s.allgather_log("X_HAT", 2, 3)
        assert np.array_equal(gx[0], s.loop_get_log("X_HAT")[2:5])
        assert np.array_equal(s.allgather_u()[0], U[-1])
        s.comm_barrier()
        assert np.array_equal(s.comm_allreduce_max([1.5, -2.0]), [1.5, -2.0])
        st = s.loop_get_log("STATUS_DYN")[:K]
        assert np.array_equal(s.comm_allgather(st)[0], st)
        with pytest.raises(RuntimeError):
            s.comm_init(0, 1, uid)          # a handle has one communicator
    finally:
        s.close()
