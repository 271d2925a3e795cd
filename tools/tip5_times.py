#!/usr/bin/env python3
"""HIP-event times of the Tip5 entry points at Merkle-level sizes (no profiler): permute / hash_pairs of 2^20..2^23 items, a 2^24-leaf build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

def best(fn, reps=8):
    t = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    return min(t), sorted(t)[len(t) // 2]

for log in (16, 18, 20, 22, 23):
    m = 1 << log
    inp = torch.empty(m * 10, dtype=torch.int64, device="cuda"); out = torch.empty(m * 5, dtype=torch.int64, device="cuda")
    tf.device.fill_random(inp, 4)
    st = torch.empty(m * 16, dtype=torch.int64, device="cuda")
    tf.device.fill_random(st, 5)
    hp = best(lambda: tf.device.tip5_hash_pairs(inp, out))
    pm = best(lambda: tf.device.tip5_permute_(st))
    print(f"2^{log:2d} items: hash_pairs {hp[0]*1e3:9.1f} us (median {hp[1]*1e3:9.1f}) = {m/hp[0]/1e6:6.3f} G/s   permute {pm[0]*1e3:9.1f} us (median {pm[1]*1e3:9.1f}) = {m/pm[0]/1e6:6.3f} G/s")
    del inp, out, st
n = 1 << 24
leaves = torch.empty(n * 5, dtype=torch.int64, device="cuda"); nodes = torch.empty(2 * n * 5, dtype=torch.int64, device="cuda")
tf.device.fill_random(leaves, 3)
mb = best(lambda: tf.device.merkle_build(leaves, n, nodes))
print(f"2^24-leaf Merkle build: {mb[0]*1e3:9.1f} us (median {mb[1]*1e3:9.1f}) = {n/mb[0]/1e6:6.3f} G leaves/s")
