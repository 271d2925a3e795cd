#!/usr/bin/env python3
"""Profiling target: zerofier-tree batch evaluation, n = m = 2^16 BFE, 5 calls."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf
L = tf.lib()
width = int(sys.argv[1]) if len(sys.argv) > 1 else 1
log = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = m = 1 << log
c = torch.empty(n * width, dtype=torch.int64, device="cuda"); p = torch.empty(m * width, dtype=torch.int64, device="cuda"); o = torch.empty(m * width, dtype=torch.int64, device="cuda")
tf.device.fill_random(c, 1); tf.device.fill_random(p, 2)
L.tf_set_batch_eval_route(2)
for _ in range(5):
    tf.device.batch_evaluate(c, n, p, o, width=width)
torch.cuda.synchronize()
