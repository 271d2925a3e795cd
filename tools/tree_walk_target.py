#!/usr/bin/env python3
"""Profiling target: walks over a PREPARED zerofier tree (tf_zerofier_tree_*), BFE by default: 5 evaluations and 5 interpolations
of n = 2^log points (first arguments: width, log)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf
width = int(sys.argv[1]) if len(sys.argv) > 1 else 1
log = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 1 << log
dom = torch.empty(n * width, dtype=torch.int64, device="cuda"); f = torch.empty(n * width, dtype=torch.int64, device="cuda")
tf.device.fill_random(dom, 1); tf.device.fill_random(f, 2)
vals = torch.empty_like(f); back = torch.empty_like(f)
with tf.device.ZerofierTree(dom, width=width) as tree:
    tree.batch_evaluate(f, n, vals); tree.interpolate(vals, back)
    torch.cuda.synchronize()
    for _ in range(5):
        tree.batch_evaluate(f, n, vals)
    torch.cuda.synchronize()
    for _ in range(5):
        tree.interpolate(vals, back)
    torch.cuda.synchronize()
