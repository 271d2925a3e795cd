#!/usr/bin/env python3
"""A/B sweep of the batch-tile size and the number of tile streams (TF_NTT_TILE_BYTES x TF_NTT_PIPE) on the headline
workload (256 x 2^20 BFE forward NTT, in place).  Every configuration's output is compared word for word with the
default plan's (2 GiB tile, one stream).  Usage: python tools/pipe_sweep.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import twenty_first_amd as tf

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, batch = 1 << 20, 256
dev = torch.device("cuda", 0)
x = torch.empty(n * batch, dtype=torch.int64, device=dev)
L = tf.lib()


def run(tile_mib, pipe):
    L.tf_set_ntt_tile_bytes(tile_mib << 20)
    L.tf_set_ntt_pipe(pipe)
    tf.device.fill_random(x, 0x7F210002)
    tf.device.ntt_(x, n, batch=batch)
    torch.cuda.synchronize()
    res = x.clone()
    for _ in range(15):
        tf.device.ntt_(x, n, batch=batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(steps):
            tf.device.ntt_(x, n, batch=batch)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps)
    return res, best


for _ in range(2):  # clocks up
    ref, base = run(2048, 1)
print(f"tile 2048 MiB pipe 1 : {base:.4f} ms/step  (reference plan)", flush=True)
for tile in (2048, 1024, 512, 256, 128, 64, 32):
    for pipe in (1, 2, 3, 4):
        if tile == 2048 and pipe == 1:
            continue
        if tile == 2048 and pipe > 1:
            continue  # a single tile: nothing to pipeline
        res, ms = run(tile, pipe)
        ok = torch.equal(res, ref)
        print(f"tile {tile:5d} MiB pipe {pipe} : {ms:.4f} ms/step  {'bit-exact' if ok else 'MISMATCH'}", flush=True)
_, again = run(2048, 1)
print(f"tile 2048 MiB pipe 1 : {again:.4f} ms/step  (reference plan, end of sweep)")
