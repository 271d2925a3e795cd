#!/usr/bin/env python3
"""Host-pointer path (tf_ntt_bfe on host memory): is there anything left to overlap?  Compares, for a batch of 2^20-point BFE
slices resident in HOST memory, (a) the library's synchronous host entry point, (b) one H2D copy + device transform + one D2H
copy from PINNED memory, (c) a chunked three-stream pipeline (H2D of chunk k+1, transform of chunk k, D2H of chunk k-1 in
flight together) from pinned memory, (d) the same pipeline from pageable memory.  All variants produce the same words."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import twenty_first_amd as tf

n = 1 << 20
dev = torch.device("cuda", 0)


def pipeline(host, batch, chunk, streams=3):
    """host: CPU int64 tensor (pinned or pageable) of batch * n words, transformed in place through the device"""
    nch = batch // chunk
    bufs = [torch.empty(chunk * n, dtype=torch.int64, device=dev) for _ in range(streams)]
    st = [torch.cuda.Stream() for _ in range(streams)]
    for k in range(nch):
        s, b = st[k % streams], bufs[k % streams]
        h = host[k * chunk * n:(k + 1) * chunk * n]
        with torch.cuda.stream(s):
            b.copy_(h, non_blocking=True)
            tf.device.ntt_(b, n, batch=chunk)
            h.copy_(b, non_blocking=True)
    torch.cuda.synchronize()


for batch in (32, 128, 512):
    ref = np.random.default_rng(5).integers(0, 2 ** 63, size=batch * n, dtype=np.int64)
    want = None
    rows = []
    # (a) library host entry point, pageable numpy buffer
    x = ref.copy().view(np.uint64)
    tf.ntt(x, batch=batch)
    want = x.copy()
    x = ref.copy().view(np.uint64)
    t0 = time.perf_counter(); tf.ntt(x, batch=batch); ta = time.perf_counter() - t0
    rows.append(("tf_ntt_bfe (library host entry point, pageable)", ta, np.array_equal(x, want)))
    # (b) pinned, monolithic
    hp = torch.from_numpy(ref.copy()).pin_memory()
    d = torch.empty(batch * n, dtype=torch.int64, device=dev)
    for rep in range(2):
        hp.copy_(torch.from_numpy(ref))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d.copy_(hp, non_blocking=True); tf.device.ntt_(d, n, batch=batch); hp.copy_(d, non_blocking=True); torch.cuda.synchronize()
        tb = time.perf_counter() - t0
    rows.append(("pinned, one H2D + transform + one D2H", tb, np.array_equal(hp.numpy().view(np.uint64), want)))
    # (c) pinned, chunked pipeline
    for chunk in (4, 8, 16):
        if batch % chunk:
            continue
        for rep in range(2):
            hp.copy_(torch.from_numpy(ref))
            torch.cuda.synchronize(); t0 = time.perf_counter(); pipeline(hp, batch, chunk); tc = time.perf_counter() - t0
        rows.append((f"pinned, 3-stream pipeline, chunks of {chunk}", tc, np.array_equal(hp.numpy().view(np.uint64), want)))
    # (d) pageable, chunked pipeline
    hq = torch.from_numpy(ref.copy())
    for rep in range(2):
        hq.copy_(torch.from_numpy(ref))
        torch.cuda.synchronize(); t0 = time.perf_counter(); pipeline(hq, batch, 8); td = time.perf_counter() - t0
    rows.append(("pageable, 3-stream pipeline, chunks of 8", td, np.array_equal(hq.numpy().view(np.uint64), want)))
    mib = batch * n * 8 / 2 ** 20
    print(f"batch {batch} x 2^20 BFE ({mib:.0f} MiB each way)")
    for name, t, ok in rows:
        print(f"   {name:55s} {t * 1e3:9.2f} ms  {batch * n / t / 1e9:7.3f} GFelts/s  {2 * mib / 1024 / t:6.1f} GiB/s both ways  {'same words' if ok else 'MISMATCH'}")
    del hp, hq, d
