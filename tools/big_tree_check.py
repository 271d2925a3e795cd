import sys, time
sys.path.insert(0, "/root/repo")
import torch
import twenty_first_amd as tf
for width, log_n in ((1, 22), (1, 24), (3, 22)):
    n = 1 << log_n
    dom = torch.empty(n * width, dtype=torch.int64, device="cuda"); tf.device.fill_random(dom, 1)
    f = torch.empty(n * width, dtype=torch.int64, device="cuda"); tf.device.fill_random(f, 2)
    vals = torch.empty_like(f); back = torch.empty_like(f)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tf.device.batch_evaluate(f, n, dom, vals, width=width)
    tf.device.interpolate(dom, vals, back, width=width)
    z = torch.empty((n + 1) * width, dtype=torch.int64, device="cuda")
    tf.device.zerofier(dom, z, width=width)
    zv = torch.empty_like(dom)
    tf.device.batch_evaluate(z, n + 1, dom, zv, width=width)
    torch.cuda.synchronize()
    print(width, log_n, "round trip", torch.equal(back, f), "zerofier vanishes", not zv.any().item(), f"{time.perf_counter() - t0:.2f} s", f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB torch", flush=True)
    del dom, f, vals, back, z, zv
