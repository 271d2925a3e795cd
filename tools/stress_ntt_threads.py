#!/usr/bin/env python3
"""Plain transforms from concurrent host threads, each on its own stream: forward then inverse must give the input back and the
forward words must match the single-threaded call.  usage: stress_ntt_threads.py [seconds] [threads] [logs...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
logs = [int(a) for a in sys.argv[3:]] or [15, 16, 17, 13]
if os.environ.get('STRESS_MIN_PASSES'):
    tf.lib().tf_set_ntt_min_passes(int(os.environ['STRESS_MIN_PASSES']))
t_end = time.time() + seconds
bad, calls, rnd = [], 0, 0
while time.time() < t_end:
    rnd += 1
    for log in logs:
        n = 1 << log
        xs, want = [], []
        for k in range(nthreads):
            x = torch.empty(2 * n, dtype=torch.int64, device="cuda")
            tf.device.fill_random(x, 31 * rnd + 7 * log + k)
            y = x.clone()
            tf.device.ntt_(y, n, batch=2)
            xs.append(x); want.append(y)
        torch.cuda.synchronize()

        def worker(k):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(6):
                    y = xs[k].clone()
                    tf.device.ntt_(y, n, batch=2, stream=st)
                    z = y.clone()
                    tf.device.ntt_(z, n, batch=2, inverse=True, stream=st)
                    st.synchronize()
                    if not torch.equal(y, want[k]) or not torch.equal(z, xs[k]):
                        bad.append((rnd, log, k))
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
        for t in ths: t.start()
        for t in ths: t.join()
        calls += 12 * nthreads
print(f"{calls} concurrent transforms over {rnd} rounds, {nthreads} threads, logs {logs}: {'all words match' if not bad else 'MISMATCHES ' + str(bad[:10])}")
sys.exit(1 if bad else 0)
