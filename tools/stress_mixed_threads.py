#!/usr/bin/env python3
"""A mix of entry points from concurrent host threads, each thread on its own stream and with its own mix order: coset evaluation /
interpolation with per-thread offsets (the coset power-table cache), polynomial products (stream-ordered temporaries), one-shot
batch evaluation and interpolation (trees built and freed per call), Merkle builds, LDE.  Every result is compared with the words
the same call gave single-threaded.  usage: stress_mixed_threads.py [seconds] [threads]"""
import os, sys, threading, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
D = tf.device


def rnd(n, seed, stream=None):
    t = torch.empty(n, dtype=torch.int64, device="cuda")
    D.fill_random(t, seed, stream=stream)
    return t


def jobs_for(k, rnd_no):
    """list of (name, fn(stream) -> tensor)"""
    s0 = 100000 * rnd_no + 1000 * k
    out = []
    for log in (10, 13, 16):
        n = 1 << log
        c = rnd(n, s0 + log)
        off = int(rnd(1, s0 + 50 + log).cpu()[0].item()) & 0x7fffffffffffffff
        if off in (0,):
            off = 7

        def ce(st, c=c, n=n, off=off):
            o = torch.empty(2 * n, dtype=torch.int64, device="cuda")
            D.coset_evaluate(c, n, off, o, 2 * n, stream=st)
            return o
        out.append((f"coset_evaluate 2^{log}", ce))

        def ci(st, c=c, n=n, off=off):
            o = torch.empty(n, dtype=torch.int64, device="cuda")
            D.coset_interpolate(c, n, off, o, stream=st)
            return o
        out.append((f"coset_interpolate 2^{log}", ci))
        b = rnd(n, s0 + 70 + log)

        def pm(st, c=c, b=b, n=n):
            o = torch.empty(2 * n - 1, dtype=torch.int64, device="cuda")
            D.poly_mul(c, n, b, n, o, stream=st)
            return o
        out.append((f"poly_mul 2^{log}", pm))
    n = 1 << 12
    dom, f = rnd(n, s0 + 90), rnd(4 * n, s0 + 91)

    def be(st):
        o = torch.empty(n, dtype=torch.int64, device="cuda")
        D.batch_evaluate(f, 4 * n, dom, o, stream=st)
        return o
    out.append(("batch_evaluate 2^14 x 2^12", be))
    vals = rnd(n, s0 + 92)

    def ip(st):
        o = torch.empty(n, dtype=torch.int64, device="cuda")
        D.interpolate(dom, vals, o, stream=st)
        return o
    out.append(("interpolate 2^12", ip))
    leaves = rnd(5 << 12, s0 + 93)

    def mk(st):
        o = torch.empty(2 * 5 << 12, dtype=torch.int64, device="cuda")
        D.merkle_build(leaves, 1 << 12, o, stream=st)
        return o
    out.append(("merkle 2^12", mk))
    return out


tf.lib().tf_set_batch_eval_route(2)
t_end = time.time() + seconds
bad, calls, rnd_no = [], 0, 0
while time.time() < t_end:
    rnd_no += 1
    per = [jobs_for(k, rnd_no) for k in range(nthreads)]
    want = [[fn(None) for _, fn in jobs] for jobs in per]
    torch.cuda.synchronize()

    def worker(k):
        st = torch.cuda.Stream()
        order = list(range(len(per[k])))
        random.Random(k * 7919 + rnd_no).shuffle(order)
        with torch.cuda.stream(st):
            for rep in range(3):
                got = [(i, per[k][i][1](st)) for i in order]
                st.synchronize()
                for i, g in got:
                    if not torch.equal(g, want[k][i]):
                        bad.append((rnd_no, k, per[k][i][0]))
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
    for t in ths: t.start()
    for t in ths: t.join()
    calls += 3 * sum(len(j) for j in per)
print(f"{calls} concurrent calls over {rnd_no} rounds, {nthreads} threads: {'all words match' if not bad else 'MISMATCHES ' + str(bad[:12])}")
sys.exit(1 if bad else 0)
