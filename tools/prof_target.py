#!/usr/bin/env python3
"""Profiling target: the three BASELINE workloads and nothing else (no parity launches, no CPU legs), so that every
dispatch of a kernel under rocprofv3 is a full-size one.  tools/prof_r02.sh runs it under --kernel-trace --stats and the
separate --pmc passes.

    python tools/prof_target.py [--ntt 12] [--merkle 3] [--coset 2] [--tile-mib M] [--pipe K]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ntt", type=int, default=12)
    ap.add_argument("--merkle", type=int, default=3)
    ap.add_argument("--coset", type=int, default=2)
    ap.add_argument("--tile-mib", type=int, default=0)
    ap.add_argument("--pipe", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=0, help="repetitions of bench.py's commit_pipeline shapes: 128 columns x 2^18 -> LDE to 2^21 -> hash rows -> tree")
    ap.add_argument("--two-pass", type=int, default=-1, help="tf_set_ntt_two_pass mode for the coset leg (3: one-workgroup 2048 x 8 first pass)")
    a = ap.parse_args()
    import torch

    import twenty_first_amd as tf

    if a.tile_mib:
        tf.lib().tf_set_ntt_tile_bytes(a.tile_mib << 20)
    if a.pipe:
        tf.lib().tf_set_ntt_pipe(a.pipe)
    if a.two_pass >= 0:
        tf.lib().tf_set_ntt_two_pass(a.two_pass)
    dev = torch.device("cuda", 0)
    ident_path = os.environ.get("TF_PROF_IDENTITY")  # tools/prof_r02.sh: which library these counters belong to
    if ident_path:
        import json

        json.dump({"tf_version": int(tf.lib().tf_version()), "source_hash": tf.lib().tf_source_hash().decode()}, open(ident_path, "w"))
    if a.ntt:
        n, batch = 1 << 20, 256
        x = torch.empty(n * batch, dtype=torch.int64, device=dev)
        tf.device.fill_random(x, 0x7F210002)
        for _ in range(a.ntt):
            tf.device.ntt_(x, n, batch=batch)
        torch.cuda.synchronize()
        del x
    if a.merkle:
        nl = 1 << 24
        leaves = torch.empty(5 * nl, dtype=torch.int64, device=dev)
        tf.device.fill_random(leaves, 0x7F210003)
        nodes = torch.empty(10 * nl, dtype=torch.int64, device=dev)
        for _ in range(a.merkle):
            tf.device.merkle_build(leaves, nl, nodes)
        torch.cuda.synchronize()
        del leaves, nodes
    if a.pipeline:
        cols, n, m = 128, 1 << 18, 1 << 21
        vals = torch.empty(cols * n, dtype=torch.int64, device=dev)
        tf.device.fill_random(vals, 0x7F210007)
        ext = torch.empty(cols * m, dtype=torch.int64, device=dev)
        nodes = torch.empty(10 * m, dtype=torch.int64, device=dev)
        for _ in range(a.pipeline):
            tf.device.lde(vals, n, tf.BFieldElement.new(1), ext, m, tf.BFieldElement.new(7), batch=cols)
            tf.device.merkle_from_columns(ext, m, cols, nodes)
        torch.cuda.synchronize()
        del vals, ext, nodes
    if a.coset:
        n, b = 1 << 22, 64
        c = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
        tf.device.fill_random(c, 0x7F210004)
        o = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
        for _ in range(a.coset):
            tf.device.coset_evaluate(c, n, tf.BFieldElement.new(7), o, n, batch=b, width=3)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
