"""Multi-GPU work partitioning for the two hot paths (one process per GPU, torch.distributed).

PatchMatch shards by reference image exactly like the reference's thread-per-GPU problem pool
(mvs/patch_match.cc:177,190-204,394): no data-path collective, only a barrier and a max-reduce of
the elapsed time (bench.py). The helpers live here so that the N > 1 path is testable on CPU with
the gloo backend (tests/test_distributed.py).
"""
from __future__ import annotations

import numpy as np
from typing import List, Sequence, Tuple


def shard_problems(num_problems: int, rank: int, world_size: int) -> List[int]:
    """Round-robin assignment of problem indices to ranks (the reference hands problems to
    whichever GPU worker is free; with equal-cost problems that is round-robin)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, num_problems, world_size))


def rank_window(rank: int, refs_per_rank: int, half_window: int) -> Tuple[int, int]:
    """Weak-scaling layout used by bench.py: rank r owns `refs_per_rank` consecutive reference
    cameras of the ring starting at r * refs_per_rank and needs `half_window` neighbours on both
    sides as sources. Returns (first_view_index, num_views)."""
    first_ref = rank * refs_per_rank
    return first_ref - half_window, refs_per_rank + 2 * half_window


def max_over_ranks(value: float, device=None) -> float:
    """Elapsed time of the slowest rank (bench contract: barrier, time, MAX over ranks)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local_count: int, device=None) -> List[int]:
    """Units (reference images) processed by every rank: value = sum(units) / max(time)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(local_count)]
    t = torch.tensor([local_count], dtype=torch.int64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def exchange_maps(local: dict, device=None) -> dict:
    """All-gather of per-image results (image index -> tensor) between the photometric and the
    geometric pass: every rank ends up with every image's depth/normal map,
    replacing the reference's write-to-disk / read-back (patch_match.cc:507-508,530-531).
    Works with gloo (CPU tensors) and nccl/RCCL (device tensors)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local)
    gathered: List[dict] = [None] * dist.get_world_size()  # type: ignore
    dist.all_gather_object(gathered, {k: v.cpu() for k, v in local.items()})
    merged: dict = {}
    for part in gathered:
        merged.update(part)
    if device is not None:
        merged = {k: v.to(device) for k, v in merged.items()}
    return merged


_LAST_EXCHANGE: dict = {}


def last_exchange_info() -> dict:
    """How the last exchange_maps_device moved its data: {"transport": "local" | "rccl" | "gloo-staged",
    "bytes": payload all-gathered per rank, "images": maps this rank ends up with}."""
    return dict(_LAST_EXCHANGE)


def exchange_maps_device(local: dict, device) -> dict:
    """Device-side all-gather of the photometric maps (image index -> device tensor [4, H, W])
    between the two passes (replaces the reference's write-to-disk / read-back,
    patch_match.cc:507-508,530-531). One rank: the tensors are returned as they are (nothing leaves
    HBM). RCCL: every rank packs its maps into one flat device buffer, padded to the longest, and
    `all_gather_into_tensor` moves them GPU to GPU; only the (index, shape) lists travel as objects.
    gloo (CPU-side process groups, several ranks on one GPU in the tests): the same packing with
    the flat buffer staged through host memory."""
    import torch
    import torch.distributed as dist
    _LAST_EXCHANGE.clear()
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        _LAST_EXCHANGE.update(transport="local", bytes=0, images=len(local))
        return dict(local)
    world = dist.get_world_size()
    keys = sorted(local)
    meta_local = [(int(k), tuple(int(x) for x in local[k].shape)) for k in keys]
    meta: List[list] = [None] * world  # type: ignore
    dist.all_gather_object(meta, meta_local)
    sizes = [sum(int(np.prod(shape)) for _, shape in m) for m in meta]
    longest = max(max(sizes), 1)
    flat = torch.zeros(longest, dtype=torch.float32, device=device)
    if keys:
        flat[:sizes[dist.get_rank()]].copy_(torch.cat([local[k].reshape(-1).to(device=device, dtype=torch.float32) for k in keys]))
    if dist.get_backend() == "nccl":
        gathered = torch.empty(world * longest, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(gathered, flat)
        transport = "rccl"
    else:
        parts = [torch.empty(longest, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, flat.cpu())
        gathered = torch.cat(parts).to(device)
        transport = "gloo-staged"
    merged: dict = {}
    for r, m in enumerate(meta):
        off = r * longest
        for k, shape in m:
            n = int(np.prod(shape))
            merged[k] = local[k] if r == dist.get_rank() else gathered[off:off + n].view(*shape)
            off += n
    _LAST_EXCHANGE.update(transport=transport, bytes=4 * longest, images=len(merged))
    return merged
