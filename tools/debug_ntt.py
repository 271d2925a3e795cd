import sys, numpy as np
sys.path.insert(0, '.')
import twenty_first_amd as tf
from oracle import tfo
logs = list(range(0, 15)) + [16, 18, 20, 21, 22]
for inv in [False, True]:
    for log_n in logs:
        n = 1 << log_n
        batch = 5 if log_n <= 12 else (3 if log_n <= 18 else 1)
        x = tfo.fill_random(n * batch, 1000 + log_n)
        want = tfo.ntt(x, inverse=inv, batch=batch, threads=8)
        want1 = tfo.ntt(x, inverse=inv, batch=batch, threads=1)
        got = x.copy(); tf.ntt(got, batch=batch, _inverse=inv)
        bad = np.nonzero(got != want)[0]
        bad1 = np.nonzero(got != want1)[0]
        o = np.nonzero(want != want1)[0]
        if bad.size or bad1.size or o.size:
            print(log_n, inv, 'gpu-vs-oracle8', bad.size, 'gpu-vs-oracle1', bad1.size, 'oracle8-vs-oracle1', o.size, bad[:8], o[:8])
print('done')
for rep in range(3):
    log_n, inv = 22, True
    n = 1 << log_n
    x = tfo.fill_random(n, 1000 + log_n)
    want = tfo.ntt(x, inverse=inv)
    got = x.copy(); tf.ntt(got, _inverse=inv)
    bad = np.nonzero(got != want)[0]
    print('rep', rep, 'bad', bad.size)
    if bad.size:
        k1 = bad % 256
        print('  k1 values:', np.unique(k1), ' counts', np.bincount(k1)[np.unique(k1)])
        hi = bad >> 8
        print('  (k2+N2*k3) range', hi.min(), hi.max(), 'unique', np.unique(hi).size)
# fresh different size using same a1=8: 2^23 inverse (a1=8,a2=8,a3=7)
