#!/usr/bin/env python3
"""Profiling target: the latency-bound reference bench shapes, 5 calls each with a marker sync in between:
coset evaluate / interpolate 2^10 and 2^17 (bfe, xfe), Merkle builds of height 16 and 20, ntt 2^7 / 2^18."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda", 0)
seven = tf.BFieldElement.new(7)
def dt(words, seed):
    t = torch.empty(words, dtype=torch.int64, device=dev); tf.device.fill_random(t, seed); return t
for log_n in (10, 17):
    n = 1 << log_n
    for width in (1, 3):
        c, o = dt(n * width, 9), torch.empty(n * width, dtype=torch.int64, device=dev)
        for _ in range(5):
            tf.device.coset_evaluate(c, n, seven, o, n, width=width)
        torch.cuda.synchronize()
        for _ in range(5):
            tf.device.coset_interpolate(c, n, seven, o, width=width)
        torch.cuda.synchronize()
for h in (16, 20):
    n = 1 << h
    lv, nodes = dt(5 * n, 7), torch.empty(10 * n, dtype=torch.int64, device=dev)
    for _ in range(5):
        tf.device.merkle_build(lv, n, nodes)
    torch.cuda.synchronize()
for log_n in (7, 18):
    x = dt(1 << log_n, 3)
    for _ in range(5):
        tf.device.ntt_(x, 1 << log_n)
    torch.cuda.synchronize()
