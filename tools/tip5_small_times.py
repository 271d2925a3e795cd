#!/usr/bin/env python3
"""HIP-event times of hash_pairs / hash_varlen at small counts (the levels of a tree near its top, small batches)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

def best(fn, reps=12):
    t = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    return min(t)

row = []
for log in range(4, 18):
    m = 1 << log
    inp = torch.empty(m * 10, dtype=torch.int64, device="cuda"); out = torch.empty(m * 5, dtype=torch.int64, device="cuda")
    tf.device.fill_random(inp, 4)
    hp = best(lambda: tf.device.tip5_hash_pairs(inp, out))
    rows = torch.empty(m * 33, dtype=torch.int64, device="cuda"); tf.device.fill_random(rows, 6)
    hv = best(lambda: tf.device.tip5_hash_varlen_rows(rows, 33, out))
    row.append(f"2^{log:2d}: hash_pairs {hp*1e3:7.1f} us  hash_varlen(33) {hv*1e3:7.1f} us")
print("\n".join(row))
for log in (8, 12, 16, 20):
    n = 1 << log
    leaves = torch.empty(n * 5, dtype=torch.int64, device="cuda"); nodes = torch.empty(2 * n * 5, dtype=torch.int64, device="cuda")
    tf.device.fill_random(leaves, 3)
    print(f"2^{log}-leaf Merkle build: {best(lambda: tf.device.merkle_build(leaves, n, nodes))*1e3:8.1f} us")
