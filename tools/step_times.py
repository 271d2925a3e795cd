#!/usr/bin/env python3
"""Per-step durations of the bench workload (256 x 2^20 forward NTT), to see whether a slow run is uniformly slow or has outliers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
import bench
dev = torch.device("cuda:0")
n, batch = 1 << 20, 256
x = bench.synth_words(n * batch, dev, 1)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for i in range(40):
    tf.device.ntt_(x, n, batch=batch)
    evs[i + 1].record()
torch.cuda.synchronize()
print(" ".join(f"{evs[i].elapsed_time(evs[i + 1]):.2f}" for i in range(40)), flush=True)
