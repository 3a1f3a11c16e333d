"""Multi-rank path on CPU: world_size 2, gloo.  The batch shards with no data-path collective and the
controls are collected with one all-gather.  The product's host side of that path
(mpc-code_amd/shard.py: who owns which rows, padding and stitching of ragged shards, the file rendezvous that
carries rank 0's 128-byte RCCL id) runs here unchanged; what is substituted is the transport (a gloo process group
behind shard.py's communicator interface instead of RCCL inside libmpc_amd.so) and the per-rank compute (the oracle's
C restatement instead of the GPU - there is none in this container)."""
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
    import torch
    import torch.distributed as dist
    import mpc_code_amd as m
    from mpc_code_amd.shard import shard, shard_bounds, allgather_rows, exchange_unique_id
    import oracle_c
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
    rank = dist.get_rank()

    class GlooComm:      # shard.py's communicator interface over gloo
        rank, world = dist.get_rank(), dist.get_world_size()
        def allgather(self, a):
            send = torch.from_numpy(np.ascontiguousarray(a))
            recv = [torch.empty_like(send) for _ in range(self.world)]
            dist.all_gather(recv, send)
            return np.stack([r.numpy() for r in recv])
    comm = GlooComm()
    # the rendezvous that carries rank 0's RCCL id: 128 bytes through a file, atomically
    uid = exchange_unique_id((lambda: bytes(range(128))) if rank == 0 else None, rank, 2, {out!r} + ".id", timeout=60)
    assert uid == bytes(range(128))
    p = m.load_problem(m.example_path("cstr_lmpc.py"))
    total = 37                                     # ragged: 19 + 18
    rng = np.random.default_rng(20250614)
    x0 = rng.uniform([-0.5, -8.0, -5.0], [0.5, 8.0, 5.0], size=(total, 3))
    mine = shard(x0, 2, rank)
    assert mine.shape[0] == (19 if rank == 0 else 18)
    L = oracle_c.OracleC(p).closed_loop(6, mine, mine, nthreads=2)
    U_local = np.moveaxis(L["U"], 1, 0)            # [B_local][step][nu]
    U_all = allgather_rows(np.ascontiguousarray(U_local), total, comm)
    st_all = allgather_rows(np.ascontiguousarray(L["STATUS_DYN"].T), total, comm)
    try:                                           # a shard of the wrong size is an error, not a silent truncation
        allgather_rows(U_local[:-1], total, comm)
        raise SystemExit("allgather_rows accepted a short shard")
    except ValueError:
        pass
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


def test_single_rank_gather_checks_the_total():
    import pytest
    from mpc_code_amd.shard import allgather_rows
    a = np.arange(12.0).reshape(6, 2)
    assert np.array_equal(allgather_rows(a, 6), a)
    with pytest.raises(ValueError):
        allgather_rows(a, 7)


def test_shard_bounds_cover_the_batch():
    from mpc_code_amd.shard import shard_bounds
    for total in (1, 7, 64, 4096, 65537):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(total, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launcher_starts_one_rank_per_gpu_and_refuses_a_mismatch():
    """``bench.py --gpus N`` without a launcher starts N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE, an explicit rendezvous file), and
    under a launcher it refuses a WORLD_SIZE that disagrees with --gpus: a line whose n_gpus differs from --gpus is never printed.
    ``--dry-run`` takes the ranks through the same file rendezvous that carries the RCCL id, without a GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MPC_AMD_RDZV_FILE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"dry_run": True, "n_gpus": 4, "gpus_arg": 4, "local_ranks_seen": [0, 1, 2, 3]}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="8", RANK="0"), capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=8" in r.stderr and r.stdout.strip() == ""
    # without a GPU a real run fails loudly in every rank and the launcher passes the failure on (no CPU fallback, no line)
    from conftest import gpu_available
    if not gpu_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and '"metric"' not in r.stdout
