#!/usr/bin/env python3
"""Round 6: the first pass of the two-pass plans as ONE workgroup per 2048-row x 8-column tile (ntt_col2048_kernel, mode 3) against the
pairs of 1024-point workgroups (PRE2, mode 1) and the three-pass plan (mode 0): BASELINE configs[3] (64 XFieldElement polynomials
x 2^22, fast_coset_evaluate), and 2^28 words of plain / coset work at 2^21 and 2^22 points.  Same process, same box, same words.
   usage: python tools/c8_ab.py [reps] [--parity]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import twenty_first_amd as tf

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
dev = torch.device("cuda:0")
lib = tf.lib()
off = tf.BFieldElement.new(7)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
MODES = ((3, "one workgroup per 2048 x 8 tile (c8)"), (1, "pairs of 1024-point workgroups (PRE2)"), (0, "three passes                        "))


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


if "--parity" in sys.argv:
    from oracle import tfo
    for log_n, width, batch in ((21, 1, 3), (21, 3, 2), (22, 1, 2), (22, 3, 2)):
        n = 1 << log_n
        x = tfo.fill_random(batch * n * width, 4000 + log_n + width)
        lib.tf_set_ntt_two_pass(3)
        y = x.copy()
        tf.ntt(y, width=width, batch=batch)
        ok_f = np.array_equal(y, tfo.ntt(x, width=width, batch=batch, threads=16))
        tf.intt(y, width=width, batch=batch)
        ok_i = np.array_equal(y, x)
        one = x[: n * width]
        oks = []
        for nc in (n, n // 2 + 3, n - 5, 1000):
            ev = tf.fast_coset_evaluate(one[: nc * width], off, n, width=width)
            oks.append(bool(np.array_equal(ev, tfo.coset_evaluate(one[: nc * width], off, n, width=width))))
        ci = tf.fast_coset_interpolate(one, off, width=width)
        ok_ci = np.array_equal(ci, tfo.coset_interpolate(one, off, width=width))
        print(f"parity mode 3  2^{log_n} width {width} batch {batch}: forward {ok_f}  inverse round trip {ok_i}  coset evaluation (full, half+3, n-5, 1000 coefficients) {oks}  "
              f"coset interpolation {ok_ci}", flush=True)
    lib.tf_set_ntt_two_pass(-1)

n, b = 1 << 22, 64
c = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
tf.device.fill_random(c, 0x7F210004)
o = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
ref = None
for rnd in range(2):
    for mode, name in MODES:
        lib.tf_set_ntt_two_pass(mode)
        ms = timed(lambda: tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3))
        if ref is None:
            ref = o.clone()
        same = bool(torch.equal(o, ref))
        print(f"configs[3] round {rnd} mode {mode} {name}: coset_evaluate {ms:7.3f} ms  {48.0 * n * b / (ms * 1e-3) / 1e9 / 8000:.4f} of the 48 B/point roofline  same words: {same}", flush=True)
del c, o, ref
torch.cuda.empty_cache()
for log_n in (21, 22):
    for width in (1, 3):
        n = 1 << log_n
        b = (1 << 28) // n if width == 1 else (1 << 26) // n
        x = torch.empty(width * n * b, dtype=torch.int64, device=dev)
        y = torch.empty(width * n * b, dtype=torch.int64, device=dev)
        tf.device.fill_random(x, 77 + log_n)
        line = []
        for mode, name in MODES:
            lib.tf_set_ntt_two_pass(mode)
            ms_n = timed(lambda: tf.device.ntt_(y, n, batch=b, width=width))
            ms_i = timed(lambda: tf.device.ntt_(y, n, batch=b, width=width, inverse=True))
            ms_c = timed(lambda: tf.device.coset_evaluate(x, n, off, y, n, batch=b, width=width))
            line.append(f"mode {mode}: ntt {ms_n:6.3f}  intt {ms_i:6.3f}  coset_evaluate {ms_c:6.3f} ms")
        print(f"2^{log_n} width {width} batch {b}:  " + "  |  ".join(line), flush=True)
lib.tf_set_ntt_two_pass(-1)
