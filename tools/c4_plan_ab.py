#!/usr/bin/env python3
"""BASELINE configs[3] (64 XFieldElement polynomials x 2^22 coefficients, fast_coset_evaluate) on the three plans of a 2^22-point
transform, same process, same box, same words:
   mode 2   1024 x 4096, the 4096-point last pass as four 1024-point classes per tile (PRE4; round 4)
   mode 1   2048 x 2048, both passes as pairs of 1024-point workgroups (PRE2; round 3)
   mode 0   three passes (rounds 1-2)
plus a plain forward 2^22-point NTT (64 x 3 BFE-equivalent) on the same plans.   usage: python tools/c4_plan_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import twenty_first_amd as tf

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
n, b = 1 << 22, 64
c = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
tf.device.fill_random(c, 0x7F210004)
o = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
off = tf.BFieldElement.new(7)
lib = tf.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ref = None
for rnd in range(2):  # two rounds: the order of the plans does not decide the result
    for mode, name in ((2, "1024 x 4096, radix-4 last pass (PRE4)"), (1, "2048 x 2048, pairs (PRE2)          "), (0, "three passes                        ")):
        lib.tf_set_ntt_two_pass(mode)
        ms = timed(lambda: tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3))
        if ref is None:
            ref = o.clone()
        same = bool(torch.equal(o, ref))
        gbs = 48.0 * n * b / (ms * 1e-3) / 1e9
        ms_n = timed(lambda: tf.device.ntt_(o, n, batch=b, width=3))
        print(f"round {rnd} mode {mode} {name}: coset_evaluate {ms:7.3f} ms  {n * b / ms / 1e6:6.2f} G points/s  {gbs / 8000:.4f} of the 48 B/point roofline   "
              f"same words: {same}   | plain forward NTT {ms_n:7.3f} ms", flush=True)
lib.tf_set_ntt_two_pass(-1)
ms = timed(lambda: tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3))
print(f"automatic plan: coset_evaluate {ms:7.3f} ms   same words: {bool(torch.equal(o, ref))}")
