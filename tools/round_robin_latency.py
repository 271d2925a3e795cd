#!/usr/bin/env python3
"""One host thread driving several devices (device 0 listed K times on a one-GPU box) through the *_dev entry points: host time per call
against the device time of the queued work, cold (first call of a shape), after tf_prepare_*, and in steady state; and the cost of a
tf_*_multi call's worker threads on a shape where they could show (2^12-point transforms)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import twenty_first_amd as tf

lib = tf.lib()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
devs = [g % torch.cuda.device_count() for g in range(K)]
n, batch, nl, trees = 1 << 20, 48, 1 << 18, 8
streams = [torch.cuda.Stream(device=d) for d in devs]
x = [torch.zeros(n * batch, dtype=torch.int64, device=f"cuda:{d}") for d in devs]
lv = [torch.zeros(5 * nl * trees, dtype=torch.int64, device=f"cuda:{d}") for d in devs]
nd = [torch.empty(10 * nl * trees, dtype=torch.int64, device=f"cuda:{d}") for d in devs]
torch.cuda.synchronize()

def us(f):
    t = time.perf_counter(); f(); return (time.perf_counter() - t) * 1e6

print(f"devices {devs}; shapes: {batch} x 2^20 BFE ntt, {trees} trees x 2^18 leaves")
tf.set_device(devs[0])
print(f"COLD first call (builds tables, opens kernel attributes, creates pool / scratch): ntt {us(lambda: tf.device.ntt_(x[0], n, batch=batch, stream=streams[0])):9.0f} us   "
      f"merkle {us(lambda: tf.device.merkle_build(lv[0], nl, nd[0], batch=trees, stream=streams[0])):9.0f} us")
torch.cuda.synchronize()
for d in sorted(set(devs)):
    tf.set_device(d)
    print(f"tf_prepare_ntt {us(lambda: lib.tf_prepare_ntt(n, batch, 1, 0)):9.0f} us   tf_prepare_merkle {us(lambda: lib.tf_prepare_merkle(nl, trees)):9.0f} us  (device {d})")
for rnd in range(4):
    calls, t0 = [], time.perf_counter()
    for g, d in enumerate(devs):
        tf.set_device(d)
        calls.append(us(lambda: tf.device.ntt_(x[g], n, batch=batch, stream=streams[g])))
        calls.append(us(lambda: tf.device.merkle_build(lv[g], nl, nd[g], batch=trees, stream=streams[g])))
    t_host = (time.perf_counter() - t0) * 1e6
    for s in streams: s.synchronize()
    t_all = (time.perf_counter() - t0) * 1e6
    print(f"round {rnd}: host {t_host:8.0f} us for {len(calls)} calls (ntt {np.mean(calls[0::2]):6.0f} / merkle {np.mean(calls[1::2]):6.0f} us per call, max {max(calls):6.0f}), "
          f"all streams idle after {t_all:8.0f} us")
tf.set_device(0)
# worker threads of a tf_*_multi call
h = np.zeros(4 << 12, dtype=np.uint64)
tf.ntt(h, batch=4); tf.ntt(h, batch=4, devices=[0] * 4)
single = np.median([us(lambda: tf.ntt(h, batch=4)) for _ in range(50)])
for k in (1, 2, 4, 8):
    multi = np.median([us(lambda: tf.ntt(h, batch=4, devices=[0] * k)) for _ in range(50)])
    print(f"4 x 2^12 host-pointer ntt: single-device call {single:7.1f} us, tf_ntt_bfe_multi over {k} worker(s) {multi:7.1f} us (median of 50)")
