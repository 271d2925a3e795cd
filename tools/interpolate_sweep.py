#!/usr/bin/env python3
"""tf_poly_zerofier_*_dev / tf_poly_interpolate_*_dev by point count, device-resident: the zerofier, one interpolant, eight
interpolants over the same domain (batch_fast_interpolate), each checked by evaluating back on the domain; beside them the oracle's
O(n^2) lagrange_interpolate on the host at the sizes it finishes in seconds (the reference's own route there,
math/polynomial.rs:1514-1519)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import twenty_first_amd as tf
from oracle import tfo

dev = torch.device("cuda", 0)
width = int(sys.argv[1]) if len(sys.argv) > 1 else 1
max_log = int(sys.argv[2]) if len(sys.argv) > 2 else (20 if width == 1 else 18)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for log_n in range(8, max_log + 1, 2):
    n = 1 << log_n
    rows = 8
    dom = torch.empty(n * width, dtype=torch.int64, device=dev)
    vals = torch.empty(rows * n * width, dtype=torch.int64, device=dev)
    tf.device.fill_random(dom, 100 + log_n)
    tf.device.fill_random(vals, 200 + log_n)
    z = torch.empty((n + 1) * width, dtype=torch.int64, device=dev)
    one = torch.empty(n * width, dtype=torch.int64, device=dev)
    many = torch.empty(rows * n * width, dtype=torch.int64, device=dev)
    reps = 5 if log_n <= 16 else 2
    t_z = timed(lambda: tf.device.zerofier(dom, z, width=width), reps)
    t_1 = timed(lambda: tf.device.interpolate(dom, vals[: n * width], one, rows=1, width=width), reps)
    t_8 = timed(lambda: tf.device.interpolate(dom, vals, many, rows=rows, width=width), reps)
    with tf.device.ZerofierTree(dom, width=width) as tree:   # the tree kept across calls: evaluation and interpolation per use
        ev = torch.empty(n * width, dtype=torch.int64, device=dev)
        t_he = timed(lambda: tree.batch_evaluate(one, n, ev), reps)
        t_h1 = timed(lambda: tree.interpolate(vals[: n * width], one, rows=1), reps)
        t_h8 = timed(lambda: tree.interpolate(vals, many, rows=rows), reps)
    back = torch.empty(n * width, dtype=torch.int64, device=dev)
    tf.device.batch_evaluate(many[(rows - 1) * n * width:], n, dom, back, width=width)
    ok = torch.equal(back, vals[(rows - 1) * n * width:]) and torch.equal(many[: n * width], one)
    cpu = ""
    if log_n <= 12:
        d_h, v_h = dom.cpu().numpy().view(np.uint64), vals[: n * width].cpu().numpy().view(np.uint64)
        t0 = time.perf_counter()
        want = tfo.lagrange_interpolate(d_h, v_h, width)
        cpu = f"   host lagrange (1 thread) {1e3 * (time.perf_counter() - t0):9.1f} ms, {'same words' if np.array_equal(want, one.cpu().numpy().view(np.uint64)) else 'MISMATCH'}"
    print(f"width {width} n 2^{log_n}: zerofier {t_z:8.3f} ms   interpolate {t_1:8.3f} ms   8 rows {t_8:8.3f} ms   "
          f"| prepared tree: evaluate {t_he:7.3f}  interpolate {t_h1:7.3f}  8 rows {t_h8:7.3f} ms   "
          f"{'round trip ok' if ok else 'ROUND TRIP MISMATCH'}{cpu}", flush=True)
