"""bench.py's legs at world sizes 1 and 2 WITHOUT a GPU: the real bench code (sharding, the per-rank parity checks, the AND-reduced
verdicts, the rank-0 CPU legs behind barriers, the roots digest and its cross-rank checks, the single tree sharded by subtrees) runs
over gloo with a stand-in for `tf.device` that computes with the oracle on CPU tensors.  What this covers is control flow that only
exists for N > 1 and had never executed before a multi-GPU box sees it -- not the HIP path, which the `-m gpu` tests and the GPU
runs of bench.py cover.  The stand-in lives here, in tests/: bench.py itself has no CPU mode."""
import argparse
import importlib.util
import json
import os
import socket
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import twenty_first_amd as real_tf
    from oracle import tfo

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class FakeEvent:
        def __init__(self, enable_timing=False):
            self.t = 0.0

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return max((other.t - self.t) * 1e3, 1e-6)

    class Cuda:
        Event = FakeEvent

        @staticmethod
        def synchronize():
            pass

    class TorchShim:  # real torch, except torch.cuda.Event / synchronize
        cuda = Cuda

        def __getattr__(self, name):
            return getattr(torch, name)

    def words(t):
        return t.numpy().view(np.uint64)  # CPU tensors (and their contiguous slices) share memory with numpy

    class Device:  # the handful of tf.device entry points the legs use, computed by the oracle
        @staticmethod
        def fill_random(out, seed, first_index=0, stream=None):
            words(out)[:] = tfo.fill_random(out.numel(), seed, first_index=first_index)

        @staticmethod
        def ntt_(x, n, batch=1, width=1, inverse=False, stream=None):
            a = words(x)
            a[:] = tfo.ntt(a, width=width, inverse=inverse, batch=batch)

        @staticmethod
        def merkle_build(leaves, n_leaves, nodes_out, batch=1, stream=None):
            lv, nd = words(leaves), words(nodes_out)
            for b in range(batch):
                nd[b * 10 * n_leaves:(b + 1) * 10 * n_leaves] = tfo.merkle_build(lv[b * 5 * n_leaves:(b + 1) * 5 * n_leaves])

        @staticmethod
        def tip5_hash_varlen_rows(rows, row_len, out, stream=None):
            words(out)[:] = tfo.hash_varlen_rows(words(rows), row_len).reshape(-1)

    class Lib:  # the real library's host-side entry points; the one that needs a device is stubbed
        def __getattr__(self, name):
            return getattr(real_tf.lib(), name)

        @staticmethod
        def tf_debug_sclk_mhz():
            return 0.0

    class Tf:
        device = Device
        Digest = real_tf.Digest
        BFieldElement = real_tf.BFieldElement

        @staticmethod
        def lib():
            return Lib()

    dev = torch.device("cpu")
    use_dist = True

    def barrier():
        dist.barrier()

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    args = argparse.Namespace(gpus=world, steps=2, warmup=1, config=2, log_n=10, batch=6, no_settle=True, no_cpu_baseline=False, force_dist=True,
                              no_extra=False, c5_ntts=5, c5_trees=3, c5_log_n=8, merkle_log_leaves=8)
    tfs = Tf()
    ctx = dict(tf=tfs, torch=TorchShim(), dist=dist, np=np, dev=dev, world=world, rank=rank, use_dist=use_dist, barrier=barrier,
               max_over_ranks=max_over_ranks, args=args, ident=bench.library_identity(tfs), cpu_cache={})
    head = bench.ntt_headline(ctx)
    head["merkle"] = bench.merkle_leg(ctx)
    head["config5"] = bench.config5_leg(ctx, steps=2, warmup=1, headline=True)
    if rank == 0:
        json.dump(head, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bench_legs_at_world_sizes_1_and_2_over_gloo(tmp_path):
    import torch.multiprocessing as mp

    from oracle import tfo

    recs = {}
    for world in (1, 2):
        out = str(tmp_path / f"bench{world}.json")
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        recs[world] = json.load(open(out))
    for world, r in recs.items():
        assert r["n_gpus"] == world and r["parity"].startswith("bit-exact") and r["cpu_baseline"]["value"] > 0
        if world == 2:
            assert "each of the 2 ranks" in r["parity"]
        m = r["merkle"]
        assert m["parity"].startswith("root and") and m["cpu_baseline"]["value"] > 0 and m["n_gpus"] == world
        c5 = r["config5"]
        assert c5["parity"].startswith("bit-exact") and c5["roots_identical_on_all_ranks"] is True and c5["cpu_baseline"]["merkle_value"] > 0
        assert c5["config"]["ntts_this_rank"] == (5 if world == 1 else 3) and c5["config"]["trees_this_rank"] == (3 if world == 1 else 2)
    # the job's digest does not depend on how it was split, and it is the digest of the roots computed serially
    assert recs[1]["config5"]["roots_digest"] == recs[2]["config5"]["roots_digest"]
    seed = 0x7F210005 ^ (1 << 40)
    nl = 1 << 8
    roots = np.concatenate([tfo.merkle_build(tfo.fill_random(5 * nl, seed, first_index=t * 5 * nl))[5:10] for t in range(3)])
    want = tfo.hash_varlen(roots)
    assert recs[2]["config5"]["roots_digest"] == b"".join(int(tfo.bfe_value(int(v))).to_bytes(8, "little") for v in want).hex()
    # one 2^8-leaf tree over two ranks by the reference's subtree split has the root of the same tree built on one rank
    sharded = recs[2]["merkle"]["single_tree_sharded"]
    assert "error" not in sharded and sharded["root"] == recs[1]["merkle"]["root"]
