#!/usr/bin/env python3
"""Does the step time of the bench workload depend on WHICH memory the 2 GiB buffers land in?  Several fresh buffers per process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import twenty_first_amd as tf
import bench
dev = torch.device("cuda:0")
n, batch = 1 << 20, 256
keep = []
def timed(x, reps=8):
    for _ in range(3):
        tf.device.ntt_(x, n, batch=batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        tf.device.ntt_(x, n, batch=batch)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for trial in range(6):
    x = bench.synth_words(n * batch, dev, trial)
    keep.append(x)
    print(f"pid {os.getpid()} buffer {trial} at {x.data_ptr():#x}: {timed(x):.3f} ms/step", flush=True)
