#!/usr/bin/env python3
"""coset_extrapolate of a table of codewords at many points: the zerofier-tree route (all codewords' chunks walk the tree together)
against Horner and against what the router picks, device-resident."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf
from oracle import tfo


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


off = tfo.bfe_new(7)
for width, log_n, batch, m in ((1, 16, 64, 1 << 10), (1, 16, 64, 1 << 12), (1, 16, 64, 1 << 14), (1, 18, 32, 1 << 14), (3, 16, 32, 1 << 12)):
    n = 1 << log_n
    cw = torch.empty(batch * n * width, dtype=torch.int64, device="cuda")
    tf.device.fill_random(cw, 1)
    pts = torch.empty(m * width, dtype=torch.int64, device="cuda")
    tf.device.fill_random(pts, 2)
    out = torch.empty(batch * m * width, dtype=torch.int64, device="cuda")
    res = {}
    for name, route in (("tree", 2), ("horner", 1), ("auto", 0)):
        tf.lib().tf_set_batch_eval_route(route)
        res[name] = timed(lambda: tf.device.coset_extrapolate(off, cw, n, pts, out, batch=batch, width=width))
    tf.lib().tf_set_batch_eval_route(0)
    print(f"coset_extrapolate width {width}: {batch} codewords of 2^{log_n} at {m} points: tree {res['tree']:.3f} ms  horner {res['horner']:.3f} ms  "
          f"automatic {res['auto']:.3f} ms", flush=True)
