#!/usr/bin/env python3
"""Device-resident latency of small batches (the reference calls ntt on ONE slice at a time): us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
buf = torch.randint(0, 2**62, (1 << 26,), dtype=torch.int64, device=dev, generator=g)
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for width in (1, 3):
    for log_n in (10, 12, 14, 16, 18, 20, 22, 24):
        n = 1 << log_n
        row = []
        for batch in (1, 2, 4, 8, 16):
            if n * batch * width > buf.numel():
                continue
            x = buf[: n * batch * width]
            row.append(f"b{batch}: {timed(lambda: tf.device.ntt_(x, n, batch=batch, width=width)):7.1f}")
        print(f"width {width} 2^{log_n:2d}  " + "  ".join(row), flush=True)
