#!/usr/bin/env python3
"""Device-resident coset evaluation with n_coeffs == order (the plain fast_coset_evaluate shape), 2^28 words per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
off = tf.BFieldElement.new(7)
for width, logs in ((1, (16, 18, 20, 22)), (3, (16, 18, 20, 22))):
    for log_n in logs:
        n = 1 << log_n
        total = (1 << 28) if width == 1 else 3 * (1 << 26)
        batch = total // (n * width)
        c = torch.randint(0, 2**62, (batch * n * width,), dtype=torch.int64, device=dev, generator=g)
        out = torch.empty_like(c)
        for _ in range(4):
            tf.device.coset_evaluate(c, n, off, out, n, batch=batch, width=width)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            tf.device.coset_evaluate(c, n, off, out, n, batch=batch, width=width)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"width {width} order 2^{log_n} x {batch:5d}: {ms:7.3f} ms  {batch * n / ms / 1e6:7.1f} G points/s", flush=True)
