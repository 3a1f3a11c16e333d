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


def test_the_economic_librarys_own_communicator_gathers_its_device_log(pkg):
    """BASELINE configs[3] / [4] shard over eight GPUs: the per-model library carries the collective itself (enmpc_comm_*, enmpc_allgather_log: one
    ncclAllGather from the device log), no second library and no download in between; here with the one rank of the box, through shard.RcclComm - the
    object bench.py drives for N > 1."""
    from mpc_code_amd import enmpc, shard
    p = pkg.load_problem(pkg.example_path("reactor_enmpc.py"))
    s = enmpc.EnmpcSolver(p, device=0)
    B, K = 130, 5
    x0 = np.random.default_rng(5).uniform([0.5, 0.0], [1.0, 0.5], size=(B, 2))
    try:
        os.environ.setdefault("NCCL_DEBUG", "VERSION")
        with tempfile.TemporaryFile() as cap:
            saved = os.dup(1)
            os.dup2(cap.fileno(), 1)
            try:
                comm = shard.RcclComm(s, 0, 1, rendezvous_file=os.path.join(tempfile.gettempdir(), f"mpc_amd_test_{os.getpid()}.id"))
            finally:
                os.dup2(saved, 1); os.close(saved)
            cap.seek(0)
            assert cap.read() == b""
        assert s.comm_rank() == (0, 1)
        s.alloc(B, K); s.set_state(x0); s.run(0, K)
        s.allgather_log("U", 0, K, to_host=False)      # asynchronous, device to device, behind the run on the handle's stream
        comm.barrier()
        U = s.get_log("U")
        g = s.allgather_log("U", 0, K)
        assert g.shape == (1, K, B, p.nu) and np.array_equal(g[0], U)
        assert np.array_equal(s.allgather_log("X_ES", 1, 3)[0], s.get_log("X_ES")[1:4])
        assert comm.max(2.5) == 2.5
        st = s.get_log("STATUS_DYN")
        assert np.array_equal(comm.allgather(st)[0], st)
        assert np.array_equal(shard.allgather_rows(U[-1], B, comm), U[-1])
    finally:
        s.close()
