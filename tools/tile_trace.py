#!/usr/bin/env python3
"""256 x 2^20 BFE forward NTTs with the scratch between the passes cut into batch tiles of T MiB (tf_set_ntt_tile_bytes), for a
kernel trace: does the LAST pass get faster when the tile it reads was just written and still sits in the 256 MiB Infinity Cache?
   rocprofv3 --kernel-trace ... -- python tools/tile_trace.py <tile MiB> [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import twenty_first_amd as tf

mib, reps = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 20
tf.lib().tf_set_ntt_tile_bytes(mib << 20)
n, batch = 1 << 20, 256
x = torch.empty(n * batch, dtype=torch.int64, device="cuda")
tf.device.fill_random(x, 0x7F210002)
for _ in range(10):
    tf.device.ntt_(x, n, batch=batch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    tf.device.ntt_(x, n, batch=batch)
e1.record()
torch.cuda.synchronize()
print(f"tile {mib} MiB: {e0.elapsed_time(e1) / reps:.4f} ms per 256 x 2^20", flush=True)
