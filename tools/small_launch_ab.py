import os, sys
sys.path.insert(0, "/root/repo")
import torch
import twenty_first_amd as tf
L = tf.lib()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for log in (18, 20):
    n = 1 << log
    dom = torch.empty(n, dtype=torch.int64, device="cuda"); f = torch.empty(n, dtype=torch.int64, device="cuda")
    tf.device.fill_random(dom, 1); tf.device.fill_random(f, 2)
    vals = torch.empty_like(f); back = torch.empty_like(f)
    for mode in (-1, 1, 0):
        L.tf_set_ntt_small_launch(mode)
        t_o = timed(lambda: tf.device.interpolate(dom, vals if mode != -1 else f, back), 3)
        with tf.device.ZerofierTree(dom) as tree:
            t_e = timed(lambda: tree.batch_evaluate(f, n, vals)); t_i = timed(lambda: tree.interpolate(vals, back))
        print(f"2^{log} small_launch mode {mode:2d}: one-shot interpolate {t_o:8.1f} us  prepared evaluate {t_e:8.1f}  interpolate {t_i:8.1f}", flush=True)
    L.tf_set_ntt_small_launch(-1)
# plain transforms of 2^21 .. 2^23 words in several shapes
for words_log in (21, 22, 23):
    for log in (15, 17, 19):
        n = 1 << log; batch = 1 << (words_log - log)
        x = torch.empty(n * batch, dtype=torch.int64, device="cuda"); tf.device.fill_random(x, 3)
        r = []
        for mode in (-1, 1, 0):
            L.tf_set_ntt_small_launch(mode)
            r.append(timed(lambda: tf.device.ntt_(x, n, batch=batch), 20))
        L.tf_set_ntt_small_launch(-1)
        print(f"{batch:4d} x 2^{log} (2^{words_log} words): auto {r[0]:7.1f} us  small always {r[1]:7.1f}  never {r[2]:7.1f}", flush=True)
