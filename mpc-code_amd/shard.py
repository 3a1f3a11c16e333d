"""Batch partition across ranks and the all-gather of u* (SURVEY.md section 8e).

Instances are independent (nothing in reference ``MPC_code.py:485-827`` couples two runs), so the
batch shards with no data-path collective; the only exchange is collecting the controls.  One
process per GPU.  The collective lives in the library (RCCL over xGMI: ``mpc_allgather_u``,
``mpc_allgather_log``, ``mpc_comm_*`` of ``include/mpc_amd.h``); this module holds the host side:
who owns which instances, how ragged shards are padded and stitched, and the rendezvous that hands
rank 0's RCCL id to the other ranks.  No PyTorch.

A *communicator* here is anything with ``rank``, ``world`` and ``allgather(array) -> [world, ...]`` for
equal-sized host arrays: :class:`RcclComm` (the product's, through the C-ABI) or, in the CPU tests, an
adapter over a gloo process group.
"""
from __future__ import annotations

import os
import time
from typing import Optional

import numpy as np


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous block [lo, hi) of rank ``rank``; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(array: np.ndarray, world: int, rank: int) -> np.ndarray:
    lo, hi = shard_bounds(array.shape[0], world, rank)
    return array[lo:hi]


def allgather_rows(local: np.ndarray, total: int, comm=None) -> np.ndarray:
    """All-gather host arrays whose leading axis is the (sharded) batch; returns [total, ...] on every rank.

    ``total`` is the size of the whole batch; ``local`` must be exactly this rank's block of it
    (``shard_bounds``) - anything else is an error, not a silent truncation."""
    local = np.ascontiguousarray(local)
    if comm is None or comm.world == 1:
        if local.shape[0] != total:
            raise ValueError(f"one rank owns the whole batch: got {local.shape[0]} rows, total says {total}")
        return local
    world, rank = comm.world, comm.rank
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    lo, hi = sizes[rank]
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} of {world} owns rows [{lo}, {hi}) of {total} but holds {local.shape[0]} rows")
    pad = max(h - l for l, h in sizes)
    buf = np.zeros((pad,) + local.shape[1:], dtype=local.dtype)
    buf[: local.shape[0]] = local
    recv = comm.allgather(buf)
    return np.concatenate([recv[r][: h - l] for r, (l, h) in enumerate(sizes)], axis=0)


class RcclComm:
    """The ranks of one job, one GPU each: RCCL inside ``libmpc_amd.so`` on the solver's device and stream."""

    def __init__(self, solver, rank: int, world: int, rendezvous_file: Optional[str] = None, timeout: float = 120.0):
        self.solver, self.rank, self.world = solver, int(rank), int(world)
        uid = exchange_unique_id(solver.comm_unique_id if rank == 0 else None, self.rank, self.world, rendezvous_file, timeout)
        solver.comm_init(self.rank, self.world, uid)
        self._file = rendezvous_path(rendezvous_file)
        self.barrier()
        if self.rank == 0:
            try:
                os.unlink(self._file)
            except OSError:
                pass

    def allgather(self, array: np.ndarray) -> np.ndarray:
        return self.solver.comm_allgather(array)

    def barrier(self):
        self.solver.comm_barrier()

    def max(self, value: float) -> float:
        return float(self.solver.comm_allreduce_max([value])[0])


def rendezvous_path(path: Optional[str] = None) -> str:
    """Where rank 0 leaves the 128-byte RCCL id: all ranks of a job are children of one launcher on one node, so the
    launcher's pid (plus the rendezvous port it was given) names the job."""
    if path:
        return path
    if os.environ.get("MPC_AMD_RDZV_FILE"):
        return os.environ["MPC_AMD_RDZV_FILE"]
    tag = f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mpc_amd_rccl_{tag}.id")


def exchange_unique_id(make_id, rank: int, world: int, path: Optional[str] = None, timeout: float = 120.0) -> bytes:
    """Rank 0 creates the id (``make_id()``) and publishes it atomically; the others wait for the file."""
    f = rendezvous_path(path)
    if rank == 0:
        uid = make_id()
        tmp = f + f".{os.getpid()}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(uid)
        os.replace(tmp, f)
        return uid
    t0 = time.time()
    while True:
        try:
            with open(f, "rb") as fh:
                uid = fh.read()
            if len(uid) == 128:
                return uid
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no RCCL id at {f} after {timeout:.0f} s (is rank 0 alive?)")
        time.sleep(0.01)
