#!/usr/bin/env python3
"""Profiling target: batches of short transforms only (tools/prof_r02.sh with TF_PROF_CMD): 2^28 words of 32-point and 8-point
BFieldElement transforms, 3 * 2^26 words of 32-point XFieldElement transforms, 2^28 words of 64-point BFieldElement transforms."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf
ident = os.environ.get("TF_PROF_IDENTITY")
if ident:
    json.dump({"tf_version": int(tf.lib().tf_version()), "source_hash": tf.lib().tf_source_hash().decode()}, open(ident, "w"))
d = torch.empty(1 << 28, dtype=torch.int64, device="cuda"); tf.device.fill_random(d, 11)
for n, width in ((32, 1), (8, 1), (64, 1)):
    for _ in range(6):
        tf.device.ntt_(d, n, batch=(1 << 28) // n, width=width)
x = d[:3 << 26]
for _ in range(6):
    tf.device.ntt_(x, 32, batch=(1 << 26) // 32, width=3)
torch.cuda.synchronize()
