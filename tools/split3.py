#!/usr/bin/env python3
"""Sweep the radix split (a0, a1, a2) of three-pass transforms (TF_NTT_SPLIT3 experiment switch): median of 5 timings each."""
import os, sys, statistics
os.environ["TF_NTT_EXPERIMENT"] = "1"  # enables the TF_NTT_SPLIT2 / TF_NTT_SPLIT3 planner switches
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randint(0, 2**62, (1 << 28,), dtype=torch.int64, device=dev, generator=g)
WIDTH = int(os.environ.get('WIDTH', '1'))
def timed(n, batch):
    batch = max(1, batch // (4 if WIDTH == 3 else 1))
    for _ in range(2):
        tf.device.ntt_(x[: n * batch * WIDTH], n, batch=batch, width=WIDTH)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            tf.device.ntt_(x[: n * batch * WIDTH], n, batch=batch, width=WIDTH)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    return statistics.median(ts)
for log_n in range(11, 21) if (len(sys.argv) > 1 and sys.argv[1] == "two") else []:
    n = 1 << log_n
    batch = (1 << 28) // n
    os.environ.pop("TF_NTT_SPLIT2", None)
    res = [("default", timed(n, batch))]
    for a0 in range(5, 11):
        if 5 <= log_n - a0 <= 10:
            os.environ["TF_NTT_SPLIT2"] = str(a0)
            res.append((f"{a0},{log_n - a0}", timed(n, batch)))
    os.environ.pop("TF_NTT_SPLIT2", None)
    print(f"2^{log_n}: default {res[0][1]:.3f} | " + "  ".join(f"{k}:{v:.3f}" for k, v in sorted(res[1:], key=lambda r: r[1])), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "two":
    sys.exit(0)
for log_n in [int(a) for a in sys.argv[1:]] or range(21, 29):
    n = 1 << log_n
    batch = (1 << 28) // n
    os.environ.pop("TF_NTT_SPLIT3", None)
    res = [("default", timed(n, batch))]
    for a0 in range(5, 11):
        for a1 in range(5, 11):
            a2 = log_n - a0 - a1
            if 5 <= a2 <= 10:
                os.environ["TF_NTT_SPLIT3"] = f"{a0},{a1}"
                res.append((f"{a0},{a1},{a2}", timed(n, batch)))
    os.environ.pop("TF_NTT_SPLIT3", None)
    res2 = sorted(res[1:], key=lambda r: r[1])
    print(f"2^{log_n}: default {res[0][1]:.3f} | best " + "  ".join(f"{k}:{v:.3f}" for k, v in res2[:6]) + " | worst " + "  ".join(f"{k}:{v:.3f}" for k, v in res2[-2:]), flush=True)
