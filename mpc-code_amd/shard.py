"""Batch partition across ranks and the all-gather of u* (SURVEY.md section 8e).

Instances are independent (nothing in reference ``MPC_code.py:485-827`` couples two runs), so the
batch shards with no data-path collective; the only exchange is collecting the controls.  One
process per GPU; ``torch.distributed`` is the transport (backend ``nccl`` = RCCL over xGMI on the
GPU box, ``gloo`` in CPU tests).  PyTorch is plumbing here - it never computes.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous block [lo, hi) of rank ``rank``; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(array: np.ndarray, world: int, rank: int) -> np.ndarray:
    lo, hi = shard_bounds(array.shape[0], world, rank)
    return array[lo:hi]


def allgather_rows(local: np.ndarray, total: int, group=None) -> np.ndarray:
    """All-gather host arrays whose leading axis is the (sharded) batch; returns [total, ...] on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(local)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = np.zeros((pad,) + local.shape[1:], dtype=local.dtype)
    buf[: local.shape[0]] = local
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    send = torch.from_numpy(buf).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    return np.concatenate([r.cpu().numpy()[: hi - lo] for r, (lo, hi) in zip(recv, sizes)], axis=0)


def allgather_device(send, group=None):
    """All-gather a device tensor of equal size on every rank (RCCL); returns the [world, ...] tensor."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return send.unsqueeze(0)
    world = dist.get_world_size(group)
    recv = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv
