"""Tensor-parallel process group: one process per GPU, RCCL over xGMI
(`torch.distributed` backend "nccl" IS RCCL on ROCm); gloo on CPU for tests.

Mirrors the small part of /root/reference/python/sglang/srt/distributed/
parallel_state.py (:237 GroupCoordinator, :648-758 all_reduce) and
communication_op.py:18-107 that the hot path calls.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

_TP_GROUP = None
_TP_SIZE = 1
_TP_RANK = 0


def init_distributed_environment(backend: Optional[str] = None, tp_size: Optional[int] = None) -> None:
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract)."""
    global _TP_GROUP, _TP_SIZE, _TP_RANK
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        _TP_GROUP, _TP_SIZE, _TP_RANK = None, 1, 0
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    _TP_SIZE = tp_size or world
    assert world % _TP_SIZE == 0
    # consecutive ranks form a TP group (one node: all of them)
    for g0 in range(0, world, _TP_SIZE):
        ranks = list(range(g0, g0 + _TP_SIZE))
        grp = dist.new_group(ranks) if _TP_SIZE != world else dist.group.WORLD
        if rank in ranks:
            _TP_GROUP = grp
            _TP_RANK = rank - g0


def destroy() -> None:
    global _TP_GROUP, _TP_SIZE, _TP_RANK
    if dist.is_initialized():
        dist.destroy_process_group()
    _TP_GROUP, _TP_SIZE, _TP_RANK = None, 1, 0


def get_tensor_model_parallel_world_size() -> int:
    return _TP_SIZE


def get_tensor_model_parallel_rank() -> int:
    return _TP_RANK


def get_tp_group():
    return _TP_GROUP


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """SUM over the TP ranks (communication_op.py:18).  RCCL enqueues on its own
    stream behind an event on the current one, i.e. on a side HIP stream."""
    if _TP_SIZE == 1:
        return x
    dist.all_reduce(x, group=_TP_GROUP)
    return x


def tensor_model_parallel_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    if _TP_SIZE == 1:
        return x
    if dim < 0:
        dim += x.dim()
    parts = [torch.empty_like(x) for _ in range(_TP_SIZE)]
    dist.all_gather(parts, x.contiguous(), group=_TP_GROUP)
    return torch.cat(parts, dim=dim)


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()
