#!/usr/bin/env python3
"""hash_varlen over the rows of a table, column-major (one codeword per column: tf_hash_table_rows_*_dev, the producer of a prover's
leaves) and row-major (tf_tip5_hash_varlen_rows_dev), and the columns -> Merkle tree call; HIP events, best of 8 (median)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

def best(fn, reps=8):
    t = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    t.sort()
    return t[0], t[len(t) // 2]

print("library", os.environ.get("TF_HIP_LIBRARY", "libtf_hip.so (default)"))
for log_rows, cols, width in ((21, 128, 1), (20, 64, 1), (20, 32, 3), (18, 256, 1), (21, 33, 1)):
    n = 1 << log_rows
    table = torch.empty(n * cols * width, dtype=torch.int64, device="cuda"); tf.device.fill_random(table, 9)
    out = torch.empty(n * 5, dtype=torch.int64, device="cuda"); nodes = torch.empty(2 * n * 5, dtype=torch.int64, device="cuda")
    perms = n * (cols * width // 10 + 1)
    c = best(lambda: tf.device.hash_table_rows(table, n, cols, out, width=width))
    r = best(lambda: tf.device.tip5_hash_varlen_rows(table, cols * width, out))
    m = best(lambda: tf.device.merkle_from_columns(table, n, cols, nodes, width=width))
    print(f"2^{log_rows} rows x {cols} columns (width {width}): column-major {c[0]*1e3:8.1f} us (median {c[1]*1e3:8.1f}) = {perms/c[0]/1e6:6.3f} G perm/s   "
          f"row-major {r[0]*1e3:8.1f} us = {perms/r[0]/1e6:6.3f} G perm/s   columns -> tree {m[0]*1e3:8.1f} us")
