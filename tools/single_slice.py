import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import twenty_first_amd as tf
for log_n in (16, 18, 20, 22):
    n = 1 << log_n
    x = torch.empty(n, dtype=torch.int64, device="cuda"); tf.device.fill_random(x, 1)
    for _ in range(20): tf.device.ntt_(x, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): tf.device.ntt_(x, n)
    e1.record(); torch.cuda.synchronize()
    print(f"single 2^{log_n} slice: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us", flush=True)
