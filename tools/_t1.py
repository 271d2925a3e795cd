import sys, numpy as np
sys.path.insert(0, '.')
import twenty_first_amd as tf
from oracle import tfo
for n in [4, 16, 32, 64, 1024, 2048, 1<<14, 1<<20]:
    x = tfo.fill_random(n, 5)
    y = x.copy(); tf.ntt(y)
    z = tfo.ntt(x)
    print(n, np.array_equal(y, z), int((y != z).sum()), np.nonzero(y != z)[0][:8], flush=True)
