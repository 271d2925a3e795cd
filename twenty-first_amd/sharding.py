"""Batch sharding of independent transforms / trees across ranks.

The hot path has no exchange step: polynomials and Merkle trees are independent units, so N ranks
(one process per GPU, launched with torch.distributed.run) each take a contiguous slice of the batch
index -- device g of G gets units [g*B/G, (g+1)*B/G) -- and no data-path collective exists.  The only
collective is the final gather of small results (40-byte roots / completion), which goes over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Tuple


def shard_range(total_units: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous split [lo, hi) of `total_units` over `world_size` ranks; the first (total % world)
    ranks take one extra unit, so any batch size works (not only multiples of the world size)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(total_units, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def all_shards(total_units: int, world_size: int) -> List[Tuple[int, int]]:
    return [shard_range(total_units, world_size, r) for r in range(world_size)]


def gather_roots(local_roots, total_units: int):
    """All-gather per-rank digests (a (k, 5) int64 tensor each) into the (total_units, 5) tensor of the
    whole job, in batch order.  Payload is 40 bytes per unit: bandwidth-irrelevant."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    rank = dist.get_rank()
    shards = all_shards(total_units, world)
    maxlen = max(hi - lo for lo, hi in shards)
    buf = torch.zeros((maxlen, 5), dtype=torch.int64, device=local_roots.device)
    lo, hi = shards[rank]
    buf[: hi - lo] = local_roots.reshape(-1, 5)
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    parts = [gathered[r][: shards[r][1] - shards[r][0]] for r in range(world)]
    return torch.cat(parts, dim=0)
