#!/usr/bin/env python3
"""Batches of n <= 32-point transforms (argument: n, default 32): parity of ragged batches against the oracle, then the time of 2^28 words (BFE) / 3 * 2^26 (XFE).
Laboratory library with TF_NTT_NO_ROWS32=1: ntt_tiny_kernel / the generic row pass; with TF_NTT_ROWS32_WG=1: round 2 workgroup-tile kernel (n = 32)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import twenty_first_amd as tf
from oracle import tfo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LOG = N.bit_length() - 1
for width in (1, 3):
    ok = True
    for batch in (1, 2, 3, 20, 21, 22, 41, 42, 43, 63, 64, 65, 127, 129, 1000, 5000, (1 << 16) + 5, (1 << 18) + 3):
        for inv in (False, True):
            x = tfo.fill_random(N * batch * width, 900 + batch)
            d = torch.from_numpy(x.view(np.int64)).cuda()
            tf.device.ntt_(d, N, batch=batch, inverse=inv, width=width); torch.cuda.synchronize()
            want = tfo.ntt(x, batch=batch, inverse=inv, width=width)
            ok &= bool(np.array_equal(d.cpu().numpy().view(np.uint64), want))
    print(f"  width {width}: parity of ragged batches, both directions:", ok)
    batch = ((1 << 28) if width == 1 else (1 << 26)) // N
    d = torch.empty(N * batch * width, dtype=torch.int64, device="cuda"); tf.device.fill_random(d, 5)
    for inv in (False, True):
        for _ in range(3): tf.device.ntt_(d, N, batch=batch, inverse=inv, width=width)
        t = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); tf.device.ntt_(d, N, batch=batch, inverse=inv, width=width); b.record(); torch.cuda.synchronize(); t.append(a.elapsed_time(b))
        print(f"  width {width}: {batch} x 2^{LOG} {'inverse' if inv else 'forward'}: {min(t):.3f} ms (median {sorted(t)[5]:.3f})")
