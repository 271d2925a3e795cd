#!/usr/bin/env python3
"""Single-slice ntt through the HOST-pointer entry point (H2D + transform + D2H, synchronous) against the CPU port on one
thread: where the PCIe round trip starts to pay (the cut-off INTEGRATION.md recommends for the Rust fast path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import twenty_first_amd as tf
from oracle import tfo
def best(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e6
for width in (1, 3):
    for log_n in range(8, 25, 2):
        n = 1 << log_n
        x = tfo.fill_random(n * width, 5 + log_n)
        y = x.copy()
        tf.ntt(y, width=width)  # tables
        gpu = best(lambda: tf.ntt(y, width=width), 7 if log_n < 22 else 3)
        cpu = best(lambda: tfo.ntt(x, width=width), 5 if log_n < 20 else 2)
        print(f"width {width} 2^{log_n:2d}: GPU incl. PCIe {gpu:10.1f} us   CPU port, 1 thread {cpu:12.1f} us   ratio {cpu / gpu:7.1f}", flush=True)
