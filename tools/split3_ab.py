#!/usr/bin/env python3
"""Three-pass plans (n, 32, 1024) vs (32, n, 1024): which column pass should carry the exchange-free radix 32?
TF_NTT_EXPERIMENT=1 TF_NTT_SPLIT3 per length; plain transforms and coset evaluations, BFE and XFE, 2^28 / 3 * 2^26 words."""
import os, sys
os.environ["TF_NTT_EXPERIMENT"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

dev = torch.device("cuda", 0)
buf = torch.empty(1 << 28, dtype=torch.int64, device=dev)
out = torch.empty(1 << 28, dtype=torch.int64, device=dev)
off = tf.BFieldElement.new(7)


def timed(fn, reps=8):
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


for width in (1, 3):
    for log_n in range(21, 26):
        n = 1 << log_n
        total = (1 << 28) if width == 1 else 3 * (1 << 26)
        batch = total // (n * width)
        if batch == 0:
            continue
        row = []
        for split in (f"{log_n - 15},5", f"5,{log_n - 15}"):
            a0, a1 = (int(v) for v in split.split(","))
            if not (5 <= a0 <= 10 and 5 <= a1 <= 10):
                row.append((split, float("nan"), float("nan")))
                continue
            os.environ["TF_NTT_SPLIT3"] = split
            tf.device.fill_random(buf[:total], 3)
            x = buf[:total]
            t_ntt = timed(lambda: tf.device.ntt_(x, n, batch=batch, width=width))
            t_ce = timed(lambda: tf.device.coset_evaluate(buf[:total], n, off, out[:total], n, batch=batch, width=width))
            row.append((split, t_ntt, t_ce))
        print(f"width {width} 2^{log_n} x {batch:4d}: " + "   ".join(f"({s:5s},10): ntt {a:6.3f} ms  coset eval {b:6.3f} ms" for s, a, b in row), flush=True)
