#!/usr/bin/env python3
"""Device-resident Merkle build time by tree size (shows the latency-bound tail of the level sweep)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0")
for log_n in [4, 8, 10, 12, 14, 16, 18, 20, 22, 24]:
    n = 1 << log_n
    g = torch.Generator(device=dev); g.manual_seed(log_n)
    leaves = torch.randint(0, 2**62, (5 * n,), dtype=torch.int64, device=dev, generator=g)
    nodes = torch.empty(10 * n, dtype=torch.int64, device=dev)
    for _ in range(3):
        tf.device.merkle_build(leaves, n, nodes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        tf.device.merkle_build(leaves, n, nodes)
    e1.record(); torch.cuda.synchronize()
    print(f"2^{log_n:2d} leaves: {e0.elapsed_time(e1) / reps * 1e3:9.1f} us", flush=True)
