#!/usr/bin/env python3
"""Host latency of a device-resident coset evaluation behind queued work, in a process whose table caches are FULL (the state a long-lived
process or the test suite leaves): the coset power cache (16 tables) and the inter-pass table budget are exhausted first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import twenty_first_amd as tf

lib = tf.lib()
dev = torch.device("cuda:0")
off = tf.BFieldElement.new(7)
big = torch.zeros(48 << 20, dtype=torch.int64, device=dev)
nc, order, polys = 1 << 15, 1 << 16, 24
c = torch.zeros(3 * nc * polys, dtype=torch.int64, device=dev)
o = torch.empty(3 * order * polys, dtype=torch.int64, device=dev)
s = torch.cuda.Stream()

def probe(tag):
    lib.tf_prepare_coset_eval(nc, off, order, polys, 3)
    lat = []
    with torch.cuda.stream(s):
        for _ in range(6):
            tf.device.ntt_(big, 1 << 20, batch=48, stream=s)
            t = time.perf_counter()
            tf.device.coset_evaluate(c, nc, off, o, order, batch=polys, width=3, stream=s)
            lat.append((time.perf_counter() - t) * 1e6)
    s.synchronize()
    print(f"{tag}: coset_evaluate host latency behind a queued 48 x 2^20 transform: " + " ".join(f"{x:.0f}" for x in lat) + " us", flush=True)

probe("clean process")
if "--fill-pow" in sys.argv or "--all" in sys.argv:
    for i in range(20):  # 20 different (offset, n) pairs: the power cache holds 16
        m = 1000 + i
        cc = torch.zeros(m, dtype=torch.int64, device=dev); oo = torch.empty(2048, dtype=torch.int64, device=dev)
        tf.device.coset_evaluate(cc, m, tf.BFieldElement.new(9 + i), oo, 2048)
    torch.cuda.synchronize()
    probe("power cache full")
if "--fill-post" in sys.argv or "--all" in sys.argv:
    for log_n in (28, 27, 26):
        x = torch.zeros(1 << log_n, dtype=torch.int64, device=dev)
        tf.device.ntt_(x, 1 << log_n); tf.device.ntt_(x, 1 << log_n, inverse=True)
        del x
    torch.cuda.synchronize()
    probe("inter-pass table budget exhausted")
if "--fill-scratch" in sys.argv or "--all" in sys.argv:
    for i in range(20):
        n = 1 << 16
        x = torch.zeros(n * (3 + 2 * i), dtype=torch.int64, device=dev)
        tf.device.ntt_(x, n, batch=3 + 2 * i)
        del x
    torch.cuda.synchronize()
    probe("twenty scratch sizes later")
