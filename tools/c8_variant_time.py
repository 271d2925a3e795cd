#!/usr/bin/env python3
"""BASELINE configs[3] on the library TF_HIP_LIBRARY names (tools/build_variant.sh), modes given on the command line (default 3 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
modes = [int(a) for a in sys.argv[1:]] or [3, 1]
dev = torch.device("cuda:0"); lib = tf.lib()
n, b = 1 << 22, 64
c = torch.empty(3 * n * b, dtype=torch.int64, device=dev); tf.device.fill_random(c, 0x7F210004)
o = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
off = tf.BFieldElement.new(7)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = []
for rnd in range(2):
    for m in modes:
        lib.tf_set_ntt_two_pass(m)
        out.append(f"mode {m}: {timed(lambda: tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3)):.3f} ms")
print(os.path.basename(os.environ.get("TF_HIP_LIBRARY", "libtf_hip.so")), lib.tf_source_hash().decode(), " | ".join(out), flush=True)
