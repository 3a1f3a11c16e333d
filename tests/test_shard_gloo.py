"""Multi-rank path on CPU: world_size 2, gloo.  The batch shards with no data-path collective and the
controls are collected with one all-gather; here the per-rank compute is the oracle's C
restatement standing in for the GPU (there is none in this container) - the sharding and gathering code is the
product's own (mpc-code_amd/shard.py)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT, bench_x0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
    import torch.distributed as dist
    import mpc_code_amd as m
    from mpc_code_amd.shard import shard, shard_bounds, allgather_rows
    import oracle_c
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
    rank = dist.get_rank()
    p = m.load_problem(m.example_path("cstr_lmpc.py"))
    total = 37                                     # ragged: 19 + 18
    rng = np.random.default_rng(20250614)
    x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(total, 3))
    mine = shard(x0, 2, rank)
    assert mine.shape[0] == (19 if rank == 0 else 18)
    L = oracle_c.OracleC(p).closed_loop(6, mine, mine, nthreads=2)
    U_local = np.moveaxis(L["U"], 1, 0)            # [B_local][step][nu]
    U_all = allgather_rows(np.ascontiguousarray(U_local), total)
    st_all = allgather_rows(np.ascontiguousarray(L["STATUS_DYN"].T), total)
    np.savez({out!r} + str(rank) + ".npz", U=U_all, st=st_all)
    dist.barrier(); dist.destroy_process_group()
""")


def test_two_ranks_shard_and_allgather(tmp_path, cstr, oracle_c):
    port = _free_port()
    out = str(tmp_path / "r")
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT, port=port, out=out))
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=300) == 0
    x0 = bench_x0(37)
    ref = oracle_c.OracleC(cstr).closed_loop(6, x0, x0)
    for r in range(2):
        g = np.load(out + f"{r}.npz")
        assert g["U"].shape == (37, 6, 2)
        assert np.array_equal(g["U"], np.moveaxis(ref["U"], 1, 0))          # same code, same inputs: bit-identical
        assert np.array_equal(g["st"], ref["STATUS_DYN"].T)


def test_shard_bounds_cover_the_batch():
    from mpc_code_amd.shard import shard_bounds
    for total in (1, 7, 64, 4096, 65537):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(total, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
