#!/usr/bin/env python3
"""Latency-shaped transform (tf_set_ntt_latency_kernel) against the pass / block kernels: same words, and microseconds per call
(HIP events over back-to-back calls) for n = 2^6 .. 2^12 at growing batch -- the crossover the planner's threshold comes from.
   usage: lat_sweep.py [width]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
from twenty_first_amd import _lib
lib = _lib.lib()
width = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (6, 12)
dev = torch.device("cuda:0")


def timed(fn, reps=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ok = True
for log_n in range(lo, hi + 1):
    n = 1 << log_n
    for log_total in sorted(set((log_n, 13, 16, 18, 20, 21, 22, 23, 24))):
        if log_total < log_n:
            continue
        batch = max(1, (1 << log_total) // (n * (1 if width == 1 else 4)))
        src = torch.empty(n * batch * width, dtype=torch.int64, device=dev)
        tf.device.fill_random(src, 100 + log_n)
        res = {}
        for mode in (0, 1):
            lib.tf_set_ntt_latency_kernel(mode)
            x = src.clone()
            tf.device.ntt_(x, n, batch=batch, width=width)
            fwd = x.clone()
            tf.device.ntt_(x, n, batch=batch, width=width, inverse=True)
            torch.cuda.synchronize()
            rt = torch.equal(x, src)
            us_f = timed(lambda: tf.device.ntt_(x, n, batch=batch, width=width))
            us_i = timed(lambda: tf.device.ntt_(x, n, batch=batch, width=width, inverse=True))
            res[mode] = (fwd, rt, us_f, us_i)
        lib.tf_set_ntt_latency_kernel(-1)
        same = torch.equal(res[0][0], res[1][0]) and res[0][1] and res[1][1]
        ok &= same
        print(f"width {width} n 2^{log_n:2d} batch {batch:6d} ({batch * n * width:8d} words): pass/block {res[0][2]:7.1f} / {res[0][3]:7.1f} us   latency kernel {res[1][2]:7.1f} / {res[1][3]:7.1f} us  "
              f"x{res[0][2] / res[1][2]:.2f}  same words + round trip: {same}", flush=True)
print("ALL SAME" if ok else "MISMATCH")
