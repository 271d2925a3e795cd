#!/usr/bin/env python3
"""Prepared zerofier trees under concurrent host threads, each on its own stream: every evaluation / interpolation must give the
words the single-threaded call gave.  usage: stress_threads.py [seconds] [threads]   (a fresh process: the first use of several
kernels falls inside the threaded section, which is what tests/test_gpu_next_rows.py's four-thread test met once)"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if os.environ.get('STRESS_MIN_PASSES'):
    tf.lib().tf_set_ntt_min_passes(int(os.environ['STRESS_MIN_PASSES']))
t_end = time.time() + seconds
calls, bad = 0, []
round_no = 0
while time.time() < t_end:
    round_no += 1
    for log in (13, 9, 12, 10, 11, 14, 8):
        n = 1 << log
        dom = torch.empty(n, dtype=torch.int64, device="cuda")
        tf.device.fill_random(dom, 1000 * round_no + log)
        polys, want = [], []
        for k in range(nthreads):
            p = torch.empty(n, dtype=torch.int64, device="cuda")
            tf.device.fill_random(p, 7000 * round_no + 10 * log + k)
            polys.append(p)
        with tf.device.ZerofierTree(dom) as tree:
            torch.cuda.synchronize()
            if round_no > 1:  # (round 1: first use of the walk's kernels inside the threads)
                for p in polys:
                    v = torch.empty(n, dtype=torch.int64, device="cuda")
                    tree.batch_evaluate(p, n, v)
                    want.append(v)
                torch.cuda.synchronize()

            def worker(k):
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    for _ in range(4):
                        v = torch.empty(n, dtype=torch.int64, device="cuda")
                        b = torch.empty(n, dtype=torch.int64, device="cuda")
                        tree.batch_evaluate(polys[k], n, v, stream=st)
                        tree.interpolate(v, b, stream=st)
                        st.synchronize()
                        if not torch.equal(b, polys[k]) or (want and not torch.equal(v, want[k])):
                            bad.append((round_no, log, k))

            ths = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
            for t in ths: t.start()
            for t in ths: t.join()
            calls += 8 * nthreads
print(f"{calls} concurrent tree calls over {round_no} rounds, {nthreads} threads: {'all words match' if not bad else 'MISMATCHES ' + str(bad[:10])}")
sys.exit(1 if bad else 0)
