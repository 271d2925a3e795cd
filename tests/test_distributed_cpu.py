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
