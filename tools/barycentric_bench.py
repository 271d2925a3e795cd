#!/usr/bin/env python3
"""barycentric_evaluate for a table of codewords at one out-of-domain point (tf_barycentric_evaluate_*_dev) beside the route through
the interpolant (tf_coset_extrapolate_*_dev: inverse transform + Horner), device-resident; both give the same words (checked)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import twenty_first_amd as tf
from oracle import tfo

dev = torch.device("cuda", 0)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


x = tfo.fill_random(3, 77)
for width, log_n, batch in [(1, 16, 256), (1, 20, 64), (1, 20, 256), (1, 22, 64), (3, 20, 64), (3, 22, 16)]:
    n = 1 << log_n
    cw = torch.empty(batch * n * width, dtype=torch.int64, device=dev)
    tf.device.fill_random(cw, 5)
    out = torch.empty(3 * batch, dtype=torch.int64, device=dev)
    t_b = timed(lambda: tf.device.barycentric_evaluate(cw, n, x, out, batch=batch, width=width))
    gb = batch * n * width * 8 / 1e9
    line = f"width {width} {batch:4d} codewords x 2^{log_n}: barycentric {t_b:8.3f} ms = {gb / t_b * 1e3:7.1f} GB/s of codeword bytes"
    if width == 3:  # the interpolant route needs points of the codewords' field: available for XFE codewords at an XFE point
        pts = torch.from_numpy(x.view(np.int64)).to(dev)
        ref = torch.empty(3 * batch, dtype=torch.int64, device=dev)
        t_e = timed(lambda: tf.device.coset_extrapolate(tfo.bfe_new(1), cw, n, pts, ref, batch=batch, width=3), reps=2)
        line += f"   inverse transform + Horner {t_e:8.3f} ms   {'same words' if torch.equal(ref, out) else 'MISMATCH'}"
    print(line, flush=True)
