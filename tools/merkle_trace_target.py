#!/usr/bin/env python3
"""Profiling target: 2^log-leaf Merkle builds (default 2^24, 6 builds) and one flat hash_pairs / permute call of 2^22 items."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf
log = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log
leaves = torch.empty(n * 5, dtype=torch.int64, device="cuda")
nodes = torch.empty(2 * n * 5, dtype=torch.int64, device="cuda")
tf.device.fill_random(leaves, 3)
for _ in range(6):
    tf.device.merkle_build(leaves, n, nodes)
torch.cuda.synchronize()
m = 1 << 22
inp = torch.empty(m * 10, dtype=torch.int64, device="cuda"); out = torch.empty(m * 5, dtype=torch.int64, device="cuda")
tf.device.fill_random(inp, 4)
for _ in range(3):
    tf.device.tip5_hash_pairs(inp, out)
st = torch.empty(m * 16, dtype=torch.int64, device="cuda")
tf.device.fill_random(st, 5)
for _ in range(3):
    tf.device.tip5_permute_(st)
torch.cuda.synchronize()
