#!/usr/bin/env python3
"""Per-wave phase timeline of ntt_pass_kernel (TF_NTT_ABLATE=3 build path): where do wave cycles go?"""
import os, sys
os.environ["TF_NTT_ABLATE"] = "3"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import twenty_first_amd as tf
import ctypes as C
n, batch = 1 << 20, 64
x = torch.randint(0, 2**62, (n * batch,), dtype=torch.int64, device="cuda")
tf.lib().tf_debug_stamps(None, 0)
for _ in range(2):
    tf.device.ntt_(x, n, batch=batch)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8 * 6, dtype=np.uint64)
tf.lib().tf_debug_stamps(C.c_void_p(buf.ctypes.data), buf.size)
st = buf.reshape(-1, 6).astype(np.int64)
st = st[st[:, 0] > 0]
names = ["issue loads", "wait loads", "step1+inner tw", "LDS exchange", "step2+tw+stores"]
d = np.diff(st, axis=1)
print("waves sampled", len(st), "(last launch = pass 2 of the tile)")
for i, nm in enumerate(names):
    print(f"{nm:18s} mean {d[:, i].mean():10.0f} cyc   median {np.median(d[:, i]):10.0f}   p90 {np.percentile(d[:, i], 90):10.0f}")
tot = st[:, 5] - st[:, 0]
print(f"{'total':18s} mean {tot.mean():10.0f} cyc")
t0 = st[:, 0].min()
print("kernel span (cycles, from first to last stamp):", st[:, 5].max() - t0)
