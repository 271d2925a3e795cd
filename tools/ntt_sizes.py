#!/usr/bin/env python3
"""Device-resident forward NTT throughput by transform length, 2^28 words per call (BFE) / 3 * 2^26 (XFE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
buf = torch.randint(0, 2**62, (1 << 28,), dtype=torch.int64, device=dev, generator=g)
widths = [int(w) for w in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 3]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5, 28)
for width in widths:
    for log_n in [l for l in list(range(5, 27)) + [28] if lo <= l <= hi]:
        n = 1 << log_n
        total = (1 << 28) if width == 1 else 3 * (1 << 26)
        batch = total // (n * width)
        if batch == 0:
            continue
        x = buf[: batch * n * width]
        for _ in range(2):
            tf.device.ntt_(x, n, batch=batch, width=width)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            tf.device.ntt_(x, n, batch=batch, width=width)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        passes = 1 if log_n <= 10 else (2 if log_n <= 20 else 3)
        print(f"width {width} 2^{log_n:2d} x {batch:8d}: {ms:7.3f} ms  {batch * n / ms / 1e6:7.1f} GFelts/s  {ms / passes * 1e3 / (batch * n * width / 2**28):7.1f} us per pass per 2^28 words", flush=True)
