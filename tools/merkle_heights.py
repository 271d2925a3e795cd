#!/usr/bin/env python3
"""HIP-event times of MerkleTree builds (full node array) and root-only calls by height; the library is the one TF_HIP_LIBRARY names."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

def best(fn, reps=20):
    t = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    t.sort()
    return t[0], t[len(t) // 2]

print("library", os.environ.get("TF_HIP_LIBRARY", "libtf_hip.so (default)"))
for log in (4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24):
    n = 1 << log
    leaves = torch.empty(n * 5, dtype=torch.int64, device="cuda"); nodes = torch.empty(2 * n * 5, dtype=torch.int64, device="cuda")
    root = torch.empty(5, dtype=torch.int64, device="cuda")
    tf.device.fill_random(leaves, 3)
    b = best(lambda: tf.device.merkle_build(leaves, n, nodes))
    r = best(lambda: tf.device.merkle_root(leaves, n, root))
    print(f"height {log:2d}: build {b[0]*1e3:8.1f} us (median {b[1]*1e3:8.1f})   root {r[0]*1e3:8.1f} us (median {r[1]*1e3:8.1f})")
