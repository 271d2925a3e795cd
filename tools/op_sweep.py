#!/usr/bin/env python3
"""Device-resident time of every transform-shaped entry point at 256 x 2^20 BFE / 64 x 2^20 XFE (2^28 / 3*2^26 words)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
off = tf.BFieldElement.new(7)
def timed(fn, reps=10):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for width in (1, 3):
    n = 1 << 20
    batch = 256 if width == 1 else 64
    a = torch.randint(0, 2**62, (batch * n * width,), dtype=torch.int64, device=dev, generator=g)
    b = torch.randint(0, 2**62, (batch * n * width,), dtype=torch.int64, device=dev, generator=g)
    o = torch.empty_like(a)
    o2 = torch.empty(batch * (2 * (n // 2) - 1) * width, dtype=torch.int64, device=dev)
    rows = [
        ("ntt", lambda: tf.device.ntt_(a, n, batch=batch, width=width)),
        ("intt", lambda: tf.device.ntt_(a, n, batch=batch, width=width, inverse=True)),
        ("coset_evaluate", lambda: tf.device.coset_evaluate(a, n, off, o, n, batch=batch, width=width)),
        ("coset_interpolate", lambda: tf.device.coset_interpolate(a, n, off, o, batch=batch, width=width)),
        ("hadamard", lambda: tf.device.hadamard(a, b, o, width=width)),
        ("poly_mul 2^19 x 2^19", lambda: tf.device.poly_mul(a[: batch * (n // 2) * width], n // 2, b[: batch * (n // 2) * width], n // 2, o2, batch=batch, width=width)),
        ("lde 2^19 -> 2^20", lambda: tf.device.lde(a[: batch * (n // 2) * width], n // 2, off, o, n, off, batch=batch, width=width)),
    ]
    for name, fn in rows:
        print(f"width {width} {name:22s} {timed(fn):8.3f} ms", flush=True)
