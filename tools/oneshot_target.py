#!/usr/bin/env python3
"""Profiling target: ONE-SHOT calls (tree built inside the call, the reference's own call shape): 5 x tf_poly_interpolate and
5 x tf_poly_batch_evaluate (tree route) of n = 2^log points, BFE by default (arguments: width, log)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf
width = int(sys.argv[1]) if len(sys.argv) > 1 else 1
log = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = 1 << log
dom = torch.empty(n * width, dtype=torch.int64, device="cuda"); f = torch.empty(n * width, dtype=torch.int64, device="cuda")
tf.device.fill_random(dom, 1); tf.device.fill_random(f, 2)
vals = torch.empty_like(f); back = torch.empty_like(f)
tf.lib().tf_set_batch_eval_route(2)
tf.device.batch_evaluate(f, n, dom, vals, width=width); tf.device.interpolate(dom, vals, back, width=width)
torch.cuda.synchronize()
assert torch.equal(back, f)
for fn in (lambda: tf.device.batch_evaluate(f, n, dom, vals, width=width), lambda: tf.device.interpolate(dom, vals, back, width=width)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"width {width} 2^{log}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per call")

# isolated calls: host clock from the call to the result being there (a synchronisation before and after every call)
import time
for name, fn in (("batch_evaluate", lambda: tf.device.batch_evaluate(f, n, dom, vals, width=width)), ("interpolate", lambda: tf.device.interpolate(dom, vals, back, width=width))):
    ts = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print(f"width {width} 2^{log}: isolated {name} median {ts[len(ts) // 2]:.1f} us, best {ts[0]:.1f} us (host clock)")
