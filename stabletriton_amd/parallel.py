"""Prompt-parallel replicas: one process per GPU, independent trajectories.

The reference has no distributed code at all.  The path shards
by independent units: rank r owns prompts r, r+W, ...; every rank holds a full
UNet replica; the only collective is a one-off broadcast of the weights from
rank 0 in a few large flat buckets (RCCL over xGMI on GPUs - backend "nccl" -
or gloo on CPUs for the tests).  No per-step traffic.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_prompts(n_prompts: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership of independent prompts."""
    return list(range(rank, n_prompts, world))


@torch.no_grad()
def broadcast_parameters(tensors: Iterable[torch.Tensor], src: int = 0, bucket_bytes: int = 512 << 20) -> int:
    """Broadcast tensors in place from `src` using flat buckets (few, large collectives:
    xGMI links are point-to-point, so message count matters more than on a switch).
    Returns the number of broadcasts issued."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    tensors = list(tensors)
    n_bcast = 0
    by_kind: Dict[tuple, List[torch.Tensor]] = {}
    for t in tensors:
        by_kind.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), group in by_kind.items():
        esize = torch.empty((), dtype=dtype).element_size()
        cap = max(1, bucket_bytes // esize)
        i = 0
        while i < len(group):
            j, total = i, 0
            while j < len(group) and (total == 0 or total + group[j].numel() <= cap):
                total += group[j].numel()
                j += 1
            flat = torch.empty(total, dtype=dtype, device=device)
            if dist.get_rank() == src:
                off = 0
                for t in group[i:j]:
                    flat[off:off + t.numel()].copy_(t.reshape(-1))
                    off += t.numel()
            dist.broadcast(flat, src=src)
            n_bcast += 1
            if dist.get_rank() != src:
                off = 0
                for t in group[i:j]:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
            del flat
            i = j
    return n_bcast


def broadcast_module(module: torch.nn.Module, src: int = 0, bucket_bytes: int = 512 << 20) -> int:
    return broadcast_parameters([p.data for p in module.parameters()] + [b.data for b in module.buffers()],
                                src, bucket_bytes)


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
