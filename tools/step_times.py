#!/usr/bin/env python3
"""Windowed step times of the bench workload over a few seconds (to see whether a slow process recovers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
import bench
dev = torch.device("cuda:0")
n, batch = 1 << 20, 256
x = bench.synth_words(n * batch, dev, 1)
out = []
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        tf.device.ntt_(x, n, batch=batch)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 50)
print("windows of 50 steps (ms/step): " + " ".join(f"{v:.2f}" for v in out), flush=True)
