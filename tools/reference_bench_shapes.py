#!/usr/bin/env python3
"""Every shape of the reference's own criterion benches for the hot path, one table:
    benches/ntt.rs:15-30,48-82          bfe/xfe ntt/intt at 2^7, 2^18, 2^23 (ONE slice per call)
    benches/tip5.rs:13-48               hash_10, hash_pair, hash_varlen 10 / 16 384 elements, 65 536 x hash_10 (par_iter)
    benches/merkle_tree.rs:11-40        heights 16 / 20: par_new / sequential_new (full tree), par / sequential frugal root
    benches/polynomial_coset.rs:15-47   2^10 / 2^17, offset 7: fast_coset_evaluate / fast_coset_interpolate, bfe / xfe
Per shape: device-resident microseconds per call (back-to-back calls on one stream, HIP events), the literal drop-in call on HOST
pointers (H2D + compute + D2H, synchronous, pageable numpy memory), and the oracle (the CPU restatement, -O3 -march=native) on one
core and -- where the reference itself is parallel -- on all cores.  Rows where the GPU loses are part of the table.
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import twenty_first_amd as tf
from oracle import tfo

tfo.build()
tfo.use_native_build()
dev = torch.device("cuda", 0)
cores = os.cpu_count() or 1
seven = tf.BFieldElement.new(7)


def dev_us(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


def host_us(fn, reps):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e6


def cpu_us(fn, reps=3):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e6


def dt(words, seed):
    t = torch.empty(words, dtype=torch.int64, device=dev)
    tf.device.fill_random(t, seed)
    return t


rows = []


def row(name, d, h, c1, call=None):
    rows.append((name, d, h, c1, call))
    fmt = lambda v: "        -" if v is None else f"{v:9.1f}"
    print(f"{name:58s} {fmt(d)} {fmt(h)} {fmt(c1)} {fmt(call)}", flush=True)


print(f"# {torch.cuda.get_device_name(0)}, library {tf.lib().tf_source_hash().decode()}, {cores} host CPUs; all times in microseconds per call")
print(f"{'shape (reference bench)':58s} {'device':>9s} {'host ptr':>9s} {'cpu 1':>9s} {'cpu all':>9s}")
# ---- benches/ntt.rs
for log_n in (7, 18, 23):
    n = 1 << log_n
    for width, nm in ((1, "bfe"), (3, "xfe")):
        for inv, op in ((False, "ntt"), (True, "intt")):
            x = dt(n * width, 100 + log_n)
            hx = tfo.fill_random(n * width, 200 + log_n)
            reps_d = 200 if log_n < 20 else 20
            d = dev_us(lambda: tf.device.ntt_(x, n, width=width, inverse=inv), reps_d)
            h = host_us(lambda: tf.ntt(hx, width=width, _inverse=inv), 20 if log_n < 20 else 5)
            c = cpu_us(lambda: tfo.ntt(hx, width=width, inverse=inv), 3)
            row(f"{nm}_{op}/len/{log_n}  (ntt.rs)", d, h, c)
            del x
# ---- benches/tip5.rs
one10 = dt(10, 1)
o5 = torch.empty(5, dtype=torch.int64, device=dev)
h10 = tfo.fill_random(10, 2)
row("hash_10  (tip5.rs; one call = one permutation)", dev_us(lambda: tf.device.tip5_hash_pairs(one10, o5), 200), host_us(lambda: tf.Tip5.hash_10(h10), 50),
    cpu_us(lambda: tfo.hash_10(h10), 20))
row("hash_pair", dev_us(lambda: tf.device.tip5_hash_pairs(one10, o5), 200), host_us(lambda: tf.Tip5.hash_pair(h10[:5], h10[5:]), 50),
    cpu_us(lambda: tfo.hash_pair(h10[:5], h10[5:]), 20))
for ln in (10, 16384):
    r = dt(ln, 3)
    hr = tfo.fill_random(ln, 4)
    row(f"hash_varlen/len/{ln}  (one sequential sponge)", dev_us(lambda: tf.device.tip5_hash_varlen_rows(r, ln, o5), 50 if ln > 100 else 200),
        host_us(lambda: tf.Tip5.hash_varlen(hr), 20), cpu_us(lambda: tfo.hash_varlen(hr), 5))
cnt = 65536
inp, out = dt(cnt * 10, 5), torch.empty(cnt * 5, dtype=torch.int64, device=dev)
hin = tfo.fill_random(cnt * 10, 6)


def par_hash():
    k = min(cores, 64)
    step = (cnt + k - 1) // k
    with ThreadPoolExecutor(k) as ex:
        list(ex.map(lambda i: tfo.hash_pairs(hin[i * step * 10:(i + 1) * step * 10]), range(k)))


row("hash_parallel/len/65536  (par_iter of hash_10)", dev_us(lambda: tf.device.tip5_hash_pairs(inp, out), 100), host_us(lambda: tf.Tip5.hash_pairs(hin), 10),
    cpu_us(lambda: tfo.hash_pairs(hin), 2), cpu_us(par_hash, 3))
# ---- benches/merkle_tree.rs
for h in (16, 20):
    n = 1 << h
    lv, nodes, root = dt(5 * n, 7), torch.empty(10 * n, dtype=torch.int64, device=dev), torch.empty(5, dtype=torch.int64, device=dev)
    hl = tfo.fill_random(5 * n, 8)
    d_full, d_root = dev_us(lambda: tf.device.merkle_build(lv, n, nodes), 50), dev_us(lambda: tf.device.merkle_root(lv, n, root), 50)
    h_full, h_root = host_us(lambda: tf.MerkleTree.par_new(hl), 5), host_us(lambda: tf.MerkleTree.par_frugal_root(hl), 5)
    c_seq, c_par = cpu_us(lambda: tfo.merkle_build(hl), 2), cpu_us(lambda: tfo.merkle_build(hl, threads=min(cores, 64)), 3)
    c_fr = cpu_us(lambda: tfo.merkle_frugal_root(hl), 2)
    row(f"merkle_tree_parallel / _sequential /height/{h}  (full tree)", d_full, h_full, c_seq, c_par)
    row(f"merkle_root_frugal_parallel / _sequential /height/{h}", d_root, h_root, c_fr)
    del lv, nodes
# ---- benches/polynomial_coset.rs
for log_n in (10, 17):
    n = 1 << log_n
    for width, nm in ((1, "bfe"), (3, "xfe")):
        c, o = dt(n * width, 9), torch.empty(n * width, dtype=torch.int64, device=dev)
        hc = tfo.fill_random(n * width, 10)
        row(f"coset-evaluate {nm}-pol/{n}  (polynomial_coset.rs, offset 7)", dev_us(lambda: tf.device.coset_evaluate(c, n, seven, o, n, width=width), 200),
            host_us(lambda: tf.fast_coset_evaluate(hc, seven, n, width=width), 20), cpu_us(lambda: tfo.coset_evaluate(hc, seven, n, width=width), 3))
        row(f"coset-interpolate {nm}-pol/{n}", dev_us(lambda: tf.device.coset_interpolate(c, n, seven, o, width=width), 200),
            host_us(lambda: tf.fast_coset_interpolate(hc, seven, width=width), 20), cpu_us(lambda: tfo.coset_interpolate(hc, seven, width=width), 3))
print("# device: buffers resident in HBM, back-to-back calls on one stream.  host ptr: the drop-in call of INTEGRATION.md on pageable host memory, PCIe both ways.")
print("# cpu 1 / cpu all: oracle/tf_oracle.c (-O3 -march=native) on one core / on min(64, all) cores where the reference bench itself is parallel (rayon).")
