#!/usr/bin/env python3
"""TF_NTT_PIPE on multi-tile calls: 1024 x 2^20 BFE (four 2 GiB tiles, the shape of BASELINE configs[4] per GPU at N = 4) and configs[3]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0"); lib = tf.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n, b = 1 << 20, 1024
x = torch.empty(n * b, dtype=torch.int64, device=dev); tf.device.fill_random(x, 5)
res = []
for k in (1, 2, 1, 2):
    lib.tf_set_ntt_pipe(k)
    res.append(f"pipe {k}: {timed(lambda: tf.device.ntt_(x, n, batch=b)):.3f} ms")
print("1024 x 2^20 BFE ntt: " + " | ".join(res), flush=True)
del x; torch.cuda.empty_cache()
n, b = 1 << 22, 64
c = torch.empty(3 * n * b, dtype=torch.int64, device=dev); tf.device.fill_random(c, 6)
o = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
off = tf.BFieldElement.new(7)
res = []
for k in (1, 2, 1, 2):
    lib.tf_set_ntt_pipe(k)
    res.append(f"pipe {k}: {timed(lambda: tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3), 10):.3f} ms")
print("configs[3]: " + " | ".join(res), flush=True)
res = []
for k in (1, 2, 1, 2):
    lib.tf_set_ntt_pipe(k)
    res.append(f"pipe {k}: {timed(lambda: tf.device.ntt_(o, n, batch=b, width=3), 10):.3f} ms")
print("64 x 2^22 XFE ntt: " + " | ".join(res), flush=True)
