#!/usr/bin/env python3
"""fast_multiply with a product length that is / is not a multiple of 16 words: cost of the misaligned truncated store."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
def timed(fn, reps=10):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for log_n, batch in ((20, 256), (18, 1024), (16, 4096), (22, 64), (14, 16384), (12, 65536)):
    n = 1 << log_n
    for width in (1, 3):
        bt = batch if width == 1 else batch // 4
        na = n // 2
        a = torch.randint(0, 2**62, (bt * na * width,), dtype=torch.int64, device=dev, generator=g)
        for nb in (na, na - 15, na - 16):
            b = torch.randint(0, 2**62, (bt * nb * width,), dtype=torch.int64, device=dev, generator=g)
            o = torch.empty(bt * (na + nb - 1) * width, dtype=torch.int64, device=dev)
            t = timed(lambda: tf.device.poly_mul(a, na, b, nb, o, batch=bt, width=width))
            print(f"2^{log_n} width {width} batch {bt}: na 2^{log_n-1} nb na-{na-nb:2d} product {na+nb-1} ({(na+nb-1)%16} mod 16): {t:8.3f} ms", flush=True)
