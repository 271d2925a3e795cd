#!/usr/bin/env python3
"""One-off parity check of the largest single transforms (not part of the pytest suite: the oracle needs minutes)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import twenty_first_amd as tf
from oracle import tfo
WIDTH = int(os.environ.get("WIDTH", "1"))  # 3: XFieldElement slices
for log_n in [int(a) for a in sys.argv[1:]] or [26, 28]:
    n = 1 << log_n
    x = tfo.fill_random(n * WIDTH, 0xABC + log_n)
    t0 = time.time(); want = tfo.ntt(x, width=WIDTH, threads=3 if WIDTH == 3 else 1) if WIDTH == 3 else tfo.ntt(x); t1 = time.time()
    d = torch.from_numpy(x.view(np.int64)).cuda()
    tf.device.ntt_(d, n, width=WIDTH); torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.uint64)
    ok_fwd = np.array_equal(got, want)
    del got, want
    t2 = time.time(); tf.device.ntt_(d, n, width=WIDTH, inverse=True); torch.cuda.synchronize(); t3 = time.time()
    back = d.cpu().numpy().view(np.uint64)
    ok_inv = np.array_equal(back, x)
    print(f"width {WIDTH} 2^{log_n}: forward match={ok_fwd}  inverse round trip={ok_inv}  oracle {t1-t0:.1f}s  gpu inverse {(t3-t2)*1e3:.1f} ms", flush=True)
    del d, back, x
    torch.cuda.empty_cache()
