#!/usr/bin/env python3
"""tf_poly_batch_evaluate_bfe_dev: Horner route vs zerofier-tree route (tf_set_batch_eval_route) by polynomial length n and
point count m, device-resident; both routes return the same words (checked).  Sets the crossover used by tree_route()."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

L = tf.lib()
dev = torch.device("cuda", 0)
width = int(sys.argv[1]) if len(sys.argv) > 1 else 1


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


PAIRS = [(12, 12), (14, 12), (14, 14), (16, 13), (16, 14), (16, 16), (18, 14), (18, 16), (18, 18), (20, 16), (20, 20), (22, 16)]
if len(sys.argv) > 2 and sys.argv[2] == "fine":  # the grid the router's cost model is fitted on
    PAIRS = [(ln, lm) for ln in range(10, 21, 2) for lm in range(9, 19) if lm <= ln + 2]
for log_n, log_m in PAIRS:
    n, m = 1 << log_n, 1 << log_m
    c = torch.empty(n * width, dtype=torch.int64, device=dev)
    p = torch.empty(m * width, dtype=torch.int64, device=dev)
    tf.device.fill_random(c, 1)
    tf.device.fill_random(p, 2)
    ot = torch.empty(m * width, dtype=torch.int64, device=dev)
    oh = torch.empty(m * width, dtype=torch.int64, device=dev)
    L.tf_set_batch_eval_route(2)
    t_tree = timed(lambda: tf.device.batch_evaluate(c, n, p, ot, width=width), 3)
    horner_cost = n * m
    if horner_cost <= (1 << 38):
        L.tf_set_batch_eval_route(1)
        t_h = timed(lambda: tf.device.batch_evaluate(c, n, p, oh, width=width), 2 if horner_cost > (1 << 34) else 5)
        same = torch.equal(ot, oh)
    else:
        t_h, same = float("nan"), None
    L.tf_set_batch_eval_route(0)
    print(f"width {width} n 2^{log_n} m 2^{log_m}: tree {t_tree:9.3f} ms   horner {t_h:10.3f} ms   {'same words' if same else ('MISMATCH' if same is False else 'horner skipped')}", flush=True)
