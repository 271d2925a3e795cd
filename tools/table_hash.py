#!/usr/bin/env python3
"""Row hashing of a device-resident table -> Merkle tree: column-major (one codeword per column) vs row-major input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
def timed(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for log_rows, n_cols, width in [(20, 64, 1), (22, 32, 1), (20, 24, 3), (16, 200, 1)]:
    n_rows = 1 << log_rows
    t = torch.randint(0, 2**62, (n_cols * n_rows * width,), dtype=torch.int64, device=dev, generator=g)
    nodes = torch.empty(10 * n_rows, dtype=torch.int64, device=dev)
    ms_c = timed(lambda: tf.device.merkle_from_columns(t, n_rows, n_cols, nodes, width=width))
    ms_r = timed(lambda: tf.device.merkle_from_rows(t, n_cols * width, n_rows, nodes))
    perms = n_rows * ((n_cols * width) // 10 + 1) + n_rows
    print(f"2^{log_rows} rows x {n_cols} {'XFE' if width == 3 else 'BFE'} columns ({t.numel() * 8 / 2**20:.0f} MiB): column-major {ms_c:.3f} ms, row-major {ms_r:.3f} ms, "
          f"{perms / ms_c / 1e6:.2f} G permutations/s", flush=True)
