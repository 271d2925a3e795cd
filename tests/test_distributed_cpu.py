"""world_size-2 `gloo` test of the N>1 path on CPU: contiguous batch sharding + root gather.

The per-rank compute here is the CPU oracle (this is a test of the sharding/gather logic, which is all
the multi-GPU path adds -- there is no data-path collective); on the GPU node each rank
runs the same code with the HIP path and backend "nccl" (RCCL over xGMI).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_trees, n_leaves, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from oracle import tfo
    from twenty_first_amd.sharding import gather_roots, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total_trees, world, rank)
    roots = []
    for tree in range(lo, hi):
        leaves = tfo.fill_random(5 * n_leaves, 0x7F210005 + (tree << 32))
        roots.append(tfo.merkle_build(leaves)[5:10].view(np.int64))
    local = torch.from_numpy(np.stack(roots)) if roots else torch.zeros((0, 5), dtype=torch.int64)
    all_roots = gather_roots(local, total_trees)
    # NTT shards: every rank transforms its own slice; the "gather" is just concatenation order
    n = 256
    xs = [tfo.ntt(tfo.fill_random(n, 0x7F210002 + (u << 32))) for u in range(lo, hi)]
    sums = torch.tensor([int(x.sum() % (1 << 62)) for x in xs] + [0] * (total_trees - (hi - lo)), dtype=torch.int64)
    dist.barrier()
    if rank == 0:
        np.save(out_path, all_roots.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("total_trees", [4, 5])
def test_two_rank_sharding_and_gather(tmp_path, total_trees):
    import torch.multiprocessing as mp

    from oracle import tfo

    world, n_leaves = 2, 16
    port = _free_port()
    out = str(tmp_path / "roots.npy")
    mp.spawn(_worker, args=(world, port, total_trees, n_leaves, out), nprocs=world, join=True)
    got = np.load(out).view(np.uint64)
    assert got.shape == (total_trees, 5)
    for tree in range(total_trees):
        leaves = tfo.fill_random(5 * n_leaves, 0x7F210005 + (tree << 32))
        want = tfo.merkle_build(leaves)[5:10]
        assert np.array_equal(got[tree], want)


def _tree_worker(rank, world, port, n_leaves, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from oracle import tfo
    from twenty_first_amd.sharding import sharded_tree

    dist.init_process_group("gloo", rank=rank, world_size=world)
    leaves = tfo.fill_random(5 * n_leaves, 0x7F210003).view(np.int64).reshape(-1, 5)
    per = n_leaves // world
    local = torch.from_numpy(leaves[rank * per:(rank + 1) * per].copy())

    def build(l):  # the oracle stands in for tf.device.merkle_build (this tests the split / gather / finish logic)
        return torch.from_numpy(tfo.merkle_build(l.numpy().view(np.uint64).reshape(-1)).view(np.int64).reshape(-1, 5))

    root, sub, top = sharded_tree(local, build, build)
    np.save(f"{out_path}_{rank}.npy", np.concatenate([root.numpy().reshape(1, 5), top.numpy().reshape(-1, 5), sub.numpy().reshape(-1, 5)]))
    dist.barrier()
    dist.destroy_process_group()


def test_single_tree_sharded_over_two_ranks_by_subtrees(tmp_path):
    """SURVEY 8(e) 'optionally': one tree across G ranks by the reference's subtree split (merkle_tree.rs:247-275): every node
    of the sharded tree equals the node of the unsharded one, found through sharding.global_node."""
    import torch.multiprocessing as mp

    from oracle import tfo
    from twenty_first_amd.sharding import global_node

    world, n_leaves = 2, 64
    port = _free_port()
    out = str(tmp_path / "tree")
    mp.spawn(_tree_worker, args=(world, port, n_leaves, out), nprocs=world, join=True)
    want = tfo.merkle_build(tfo.fill_random(5 * n_leaves, 0x7F210003)).reshape(-1, 5)
    parts = [np.load(f"{out}_{r}.npy").view(np.uint64) for r in range(world)]
    for r in range(world):
        assert np.array_equal(parts[r][0], want[1])                      # every rank ends with the root
    tops = [p[1:1 + 2 * world] for p in parts]
    subs = [p[1 + 2 * world:] for p in parts]
    for idx in range(1, 2 * n_leaves):
        where = global_node(idx, n_leaves, world)
        got = tops[0][where[1]] if where[0] == "top" else subs[where[1]][where[2]]
        assert np.array_equal(got, want[idx]), (idx, where)


def _job_worker(rank, world, port, total_trees, n_leaves, out_path):
    """The config-5 job of bench.py with the oracle as per-rank compute: shard the trees, build, gather the roots, digest them, and
    run the cross-rank checks bench.py runs (identical_on_all_ranks, all_ranks_true)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from oracle import tfo
    from twenty_first_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(total_trees, world, rank)
    seed = 0x7F210005 ^ (1 << 40)
    roots = []
    for tree in range(lo, hi):  # tree u of the job = leaves [u 5 nl, (u + 1) 5 nl) of ONE counter-based sequence (bench.py's layout)
        leaves = tfo.fill_random(5 * n_leaves, seed, first_index=tree * 5 * n_leaves)
        roots.append(tfo.merkle_build(leaves)[5:10].view(np.int64))
    local = torch.from_numpy(np.stack(roots)) if roots else torch.zeros((0, 5), dtype=torch.int64)
    all_roots = sharding.gather_roots(local, total_trees)

    def hash_varlen(flat):  # the oracle stands in for tf.device.tip5_hash_varlen_rows
        return torch.from_numpy(tfo.hash_varlen(flat.numpy().view(np.uint64)).view(np.int64).copy())

    digest = sharding.roots_digest(all_roots, hash_varlen)
    same = sharding.identical_on_all_ranks(all_roots) and sharding.identical_on_all_ranks(digest)
    # a deliberately rank-dependent tensor must be reported as different, a mismatch on ONE rank must reach every rank
    differs = not sharding.identical_on_all_ranks(torch.tensor([rank], dtype=torch.int64)) if world > 1 else True
    verdict = sharding.all_ranks_true(rank != world - 1 or world == 1)
    np.save(f"{out_path}_{rank}.npy", np.concatenate([digest.numpy().reshape(5), np.array([int(same), int(differs), int(verdict)], dtype=np.int64)]))
    dist.barrier()
    dist.destroy_process_group()


def test_roots_digest_is_the_same_for_one_and_two_ranks(tmp_path):
    """bench.py --config 5 prints `roots_digest` (Tip5 hash_varlen of the gathered roots) and requires it identical at 1 / 2 / 4 / 8
    GPUs: here the same job at world sizes 1 and 2 over gloo with the oracle as compute, against the digest computed serially."""
    import torch.multiprocessing as mp

    from oracle import tfo

    total_trees, n_leaves = 7, 32
    seed = 0x7F210005 ^ (1 << 40)
    serial = np.concatenate([tfo.merkle_build(tfo.fill_random(5 * n_leaves, seed, first_index=t * 5 * n_leaves))[5:10] for t in range(total_trees)])
    want = tfo.hash_varlen(serial)
    for world in (1, 2):
        port = _free_port()
        out = str(tmp_path / f"job{world}")
        mp.spawn(_job_worker, args=(world, port, total_trees, n_leaves, out), nprocs=world, join=True)
        for r in range(world):
            got = np.load(f"{out}_{r}.npy")
            assert np.array_equal(got[:5].view(np.uint64), want), (world, r)
            assert got[5] == 1 and got[6] == 1, (world, r)          # gathered roots identical everywhere; a differing tensor is detected
            assert got[7] == (1 if world == 1 else 0), (world, r)   # one rank's mismatch reaches every rank
