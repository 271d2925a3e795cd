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


def roots_digest(all_roots, hash_varlen):
    """One digest for the whole job: Tip5 hash_varlen (tip5/mod.rs:640-660) over the gathered roots in batch order, 5 words per
    unit.  The roots of a job do not depend on how it was split, so this digest must be the same for every world size (bench.py
    prints it as `roots_digest`; 1 / 2 / 4 / 8 GPUs must agree).  `hash_varlen(flat_int64_tensor) -> (5,) int64 tensor` is the
    HIP path on the GPU node (tf.device.tip5_hash_varlen_rows on one row) and the oracle in the CPU test."""
    return hash_varlen(all_roots.reshape(-1).contiguous()).reshape(5)


def identical_on_all_ranks(t) -> bool:
    """True when every rank holds the same tensor `t` (all_gather + compare; a few words: bandwidth-irrelevant)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    t = t.contiguous()
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return all(bool(torch.equal(parts[0], q)) for q in parts[1:])


def all_ranks_true(flag: bool, device=None) -> bool:
    """Logical AND of a per-rank verdict over the ranks (MIN all-reduce)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def sharded_tree(local_leaves, build_subtree, finish_top):
    """ONE Merkle tree across the ranks, by the reference's own subtree split (MerkleTree::par_new hands the num_threads
    subtrees below the top log2(num_threads) layers to one worker each, util_types/merkle_tree.rs:165-212, :247-275): rank g of G
    (a power of two) owns leaves [g * n / G, (g + 1) * n / G), builds that subtree on its GPU, the G subtree roots (40 bytes
    each) are all-gathered, and every rank finishes the top log2(G) levels itself -- the only traffic is G x 40 bytes.

    local_leaves      (n / G, 5) int64 tensor on this rank's device
    build_subtree(l)  -> this rank's subtree as a (2 n / G, 5) node tensor in heap order (nodes[1] = subtree root), e.g.
                         tf.device.merkle_build; the oracle in the CPU test
    finish_top(r)     -> node array (2 G, 5) of the tree whose leaves are the G gathered subtree roots (same builder, G leaves)
    Returns (root (5,), local_subtree_nodes, top_nodes).  global_node() below maps a heap index of the whole tree to its
    place: the top G - 1 internal nodes are top_nodes[1 .. G), everything deeper sits in one rank's subtree."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    if world & (world - 1):
        raise ValueError("a single tree shards over a power-of-two number of ranks (the reference's subtree split)")
    sub = build_subtree(local_leaves)
    sub_root = sub.reshape(-1, 5)[1:2].contiguous()
    if world == 1:
        return sub_root[0], sub, sub.reshape(-1, 5)[:2]
    gathered = gather_roots(sub_root, world)          # (G, 5), rank order = leaf order
    top = finish_top(gathered)
    return top.reshape(-1, 5)[1], sub, top


def global_node(index: int, num_leafs: int, world_size: int):
    """Where node `index` (heap order, 1 = root) of a tree sharded by sharded_tree lives: ("top", i) for the top
    log2(G) levels (index < G; i = index in the top tree), else ("rank", g, local) with `local` the heap index inside rank g's
    subtree.  Mirrors the index arithmetic of subtrees_mut (util_types/merkle_tree.rs:247-275)."""
    if index < 1 or index >= 2 * num_leafs:
        raise IndexError(index)
    if index < world_size:
        return ("top", index)
    depth = index.bit_length() - 1
    g_depth = world_size.bit_length() - 1
    rank = (index >> (depth - g_depth)) - world_size
    local = (1 << (depth - g_depth)) | (index & ((1 << (depth - g_depth)) - 1))
    return ("rank", rank, local)
