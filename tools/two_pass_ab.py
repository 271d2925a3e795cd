#!/usr/bin/env python3
"""A/B of the two-pass plan for 2^21 / 2^22 points (tf_set_ntt_two_pass) against the three-pass plan: same words, time per call.
   usage: two_pass_ab.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
from twenty_first_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator(device=dev); g.manual_seed(7)
P = (1 << 64) - (1 << 32) + 1
OFFSET = 7 * ((1 << 32) - 1) % P  # raw Montgomery word of 7


def rnd(words):
    t = torch.empty(words, dtype=torch.int64, device=dev)
    tf.device.fill_random(t, 0x7F210003, 0)
    return t


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def ab(name, make_fn, out_of):
    res = {}
    for mode in (0, 1):
        lib.tf_set_ntt_two_pass(mode)
        fn = make_fn()
        ms = timed(fn)
        res[mode] = (ms, out_of().clone())
    lib.tf_set_ntt_two_pass(-1)
    same = torch.equal(res[0][1], res[1][1])
    print(f"{name:58s} three-pass {res[0][0]:8.3f} ms   two-pass {res[1][0]:8.3f} ms   x{res[0][0] / res[1][0]:.3f}   same words: {same}", flush=True)
    return same


ok = True
for width, total in ((1, 1 << 28), (3, 3 << 26)):
    for log_n in (21, 22):
        n = 1 << log_n
        batch = total // (n * width)
        src = rnd(total)
        # forward, then inverse of the forward result (round trip must give the input back)
        x = src.clone()
        def mk_f():
            def f():
                x.copy_(src); tf.device.ntt_(x, n, batch=batch, width=width)
            return f
        ok &= ab(f"width {width} ntt 2^{log_n} x {batch} (incl. copy)", mk_f, lambda: x)
        y = x.clone()
        def mk_i():
            def f():
                y.copy_(x); tf.device.ntt_(y, n, batch=batch, width=width, inverse=True)
            return f
        ok &= ab(f"width {width} intt 2^{log_n} x {batch} (incl. copy)", mk_i, lambda: y)
        rt = torch.equal(y, src)
        print(f"    round trip equals the input: {rt}")
        ok &= rt
        out = torch.empty_like(src)
        for nc in (n, n - 5, n // 2 + 3):
            coeffs = src[: nc * batch * width]
            def mk_c():
                def f():
                    tf.device.coset_evaluate(coeffs, nc, OFFSET, out, n, batch=batch, width=width)
                return f
            ok &= ab(f"width {width} coset_evaluate {nc} -> 2^{log_n} x {batch}", mk_c, lambda: out)
        back = torch.empty_like(src)
        def mk_ci():
            def f():
                tf.device.coset_interpolate(out, n, OFFSET, back, batch=batch, width=width)
            return f
        ok &= ab(f"width {width} coset_interpolate 2^{log_n} x {batch}", mk_ci, lambda: back)
        del src, x, y, out, back
        torch.cuda.empty_cache()
print("ALL SAME" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
