#!/usr/bin/env python3
"""BASELINE configs[3] (64 XFE polynomials x 2^22, fast_coset_evaluate) under the three-pass radix splits of the planner
(TF_NTT_EXPERIMENT=1 TF_NTT_SPLIT3="a0,a1"), plus plain XFE / BFE transforms of the same length."""
import os, sys
os.environ["TF_NTT_EXPERIMENT"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

dev = torch.device("cuda", 0)
n, b = 1 << 22, 64
c = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
tf.device.fill_random(c, 0x7F210004)
o = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
off = tf.BFieldElement.new(7)
ref = None


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for split in ["", "7,5", "5,7", "6,6", "8,4", "4,8", "6,5", "5,6"]:
    if split:
        os.environ["TF_NTT_SPLIT3"] = split
    else:
        os.environ.pop("TF_NTT_SPLIT3", None)
    try:
        ms = timed(lambda: tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3))
        res = o[: 3 * n * 4].clone()
        ok = "" if ref is None else ("bit-exact" if torch.equal(res, ref) else "MISMATCH")
        if ref is None:
            ref = res
        x = c[: 3 * n * 16]
        ms_x = timed(lambda: tf.device.ntt_(x, n, batch=16, width=3))
        y = c[: n * 64]
        ms_b = timed(lambda: tf.device.ntt_(y, n, batch=64))
        print(f"split {split or 'default':8s}: coset eval 64 x 2^22 XFE {ms:7.3f} ms {ok:10s} | ntt 16 x 2^22 XFE {ms_x:6.3f} ms | ntt 64 x 2^22 BFE {ms_b:6.3f} ms", flush=True)
    except Exception as e:
        print(f"split {split}: {e}")
