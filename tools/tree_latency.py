#!/usr/bin/env python3
"""Walks over a PREPARED zerofier tree (tf_zerofier_tree_*), microseconds per call from HIP events over back-to-back calls:
evaluation and interpolation of n = 2^log points, BFE and XFE.  usage: tree_latency.py [logs...]   (TF_TREE_LEAF_LOG to A/B the leaf)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf
logs = [int(a) for a in sys.argv[1:]] or [8, 10, 12, 14, 16]
for width in (1, 3):
    for log in logs:
        n = 1 << log
        dom = torch.empty(n * width, dtype=torch.int64, device="cuda"); f = torch.empty(n * width, dtype=torch.int64, device="cuda")
        tf.device.fill_random(dom, 1); tf.device.fill_random(f, 2)
        vals = torch.empty_like(f); back = torch.empty_like(f)
        with tf.device.ZerofierTree(dom, width=width) as tree:
            tree.batch_evaluate(f, n, vals); tree.interpolate(vals, back)
            torch.cuda.synchronize()
            assert torch.equal(back, f)
            res = []
            for fn in (lambda: tree.batch_evaluate(f, n, vals), lambda: tree.interpolate(vals, back)):
                for _ in range(5): fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50): fn()
                e1.record(); torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"leaf_log {os.environ.get('TF_TREE_LEAF_LOG', 'default')} width {width} 2^{log:2d} points, prepared tree: evaluate {res[0]:8.1f} us   interpolate {res[1]:8.1f} us", flush=True)
