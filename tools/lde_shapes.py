#!/usr/bin/env python3
"""Device-resident coset evaluation at low-degree-extension shapes (n_coeffs << order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
off = tf.BFieldElement.new(7)
for width in (1, 3):
    for log_c, log_m in [(14, 16), (16, 19), (17, 20), (18, 21), (19, 21), (20, 21), (19, 22), (20, 22), (21, 22), (20, 23), (22, 24)]:
        total_out = (1 << 28) // (2 if width == 3 else 1)
        batch = max(1, total_out // ((1 << log_m) * width))
        nc, m = 1 << log_c, 1 << log_m
        c = torch.randint(0, 2**62, (batch * nc * width,), dtype=torch.int64, device=dev, generator=g)
        out = torch.empty(batch * m * width, dtype=torch.int64, device=dev)
        for _ in range(3):
            tf.device.coset_evaluate(c, nc, off, out, m, batch=batch, width=width)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            tf.device.coset_evaluate(c, nc, off, out, m, batch=batch, width=width)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"width {width} 2^{log_c} -> 2^{log_m} x {batch:5d}: {ms:7.3f} ms  {batch * m / ms / 1e6:7.1f} G points/s", flush=True)
