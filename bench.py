#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: a batch of 256 x 2^20-point BFieldElement forward NTTs per GPU,
device-resident in and out, in place (math/ntt.rs:67-82 semantics, bit-exact).  One "step" = one pass of
the hot path over the whole batch = one tf_ntt_bfe_dev call.  Independent transforms shard across ranks
with no data-path collective, so N ranks run N x 256 transforms: weak scaling.

Prints ONE JSON line (rank 0).  `value` = (N * 256 * 2^20 * K elements) / (max-over-ranks time of the K steps).
  roofline    : dominant kernel ntt_pass_kernel; achieved = algorithmic bytes per launch / average launch
                duration from HIP events over the timed region.  A 2^20 transform is 2 launches of that
                kernel (one per pass), and SURVEY.md 8(d) prices a transform at 16 B/element, so one
                launch carries 8 B/element x (elements it touches).
  cpu_baseline: the CPU oracle (C restatement of the reference algorithm, kind "port") timed on this box's
                host cores on the same 256 x 2^20 workload, one transform per thread (what a rayon caller
                of the single-threaded ntt() does).  Rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P = 0xFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def synth_words(numel, device, seed):
    """Synthetic canonical field elements: uniform 64-bit words with the (2^-32 fraction of) words >= p
    folded back into range."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randint(-(2 ** 63), 2 ** 63 - 1, (numel,), dtype=torch.int64, device=device, generator=g)
    bad = (x < 0) & (x >= -(2 ** 32 - 1))  # as u64: >= p
    x = torch.where(bad, x + 2 ** 32, x)
    return x


def cpu_baseline_ntt(log_n, batch, sample_host_words):
    """Oracle leg: time the CPU restatement on the same workload shape; returns (dict, outputs of the sample)."""
    import numpy as np

    from oracle import tfo

    n = 1 << log_n
    cores = os.cpu_count() or 1
    x = tfo.fill_random(n * batch, 0x7F210002)
    tfo.ntt(x[:n].copy())  # build the oracle's twiddle cache outside the timed region (the reference caches too)
    t0 = time.perf_counter()
    tfo.ntt(x[: n * 4].copy(), batch=4, threads=1)
    t1 = time.perf_counter()
    single = 4 * n / (t1 - t0) / 1e9
    # thread counts up to every host CPU (one transform per thread); the best one is the baseline -- on the round-1 boxes
    # 64 threads beat 256 (0.36 vs 0.14 GFelts/s: the slices fall out of the shared caches)
    tried = {}
    for th in sorted(set(max(1, min(c, batch)) for c in (16, 32, 64, 128, cores))):
        t0 = time.perf_counter()
        tfo.ntt(x, batch=batch, threads=th)
        t1 = time.perf_counter()
        tried[th] = batch * n / (t1 - t0) / 1e9
    threads = max(tried, key=tried.get)
    multi = tried[threads]
    sample_out = None
    if sample_host_words is not None:
        k = sample_host_words.size // n
        sample_out = tfo.ntt(sample_host_words, batch=k, threads=min(threads, k))
    info = {
        "value": round(multi, 4),
        "unit": "GFelts/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{batch} x 2^{log_n} BFE forward NTT (the full workload shape), one transform per thread; best of "
                  + ", ".join(f"{k} threads: {v:.3f}" for k, v in tried.items())
                  + f" GFelts/s on {cores} host CPUs; C restatement of math/ntt.rs:153-215 (oracle/tf_oracle.c)",
        "single_thread_value": round(single, 5),
    }
    return info, sample_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="transforms per GPU")
    ap.add_argument("--no-settle", action="store_true", help="skip the untimed stabilisation passes before the warmup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even for one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--no-extra", action="store_true", help="skip the Merkle / coset-evaluation side measurements")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import twenty_first_amd as tf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available() or tf.lib().tf_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: the product has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    log_n, batch = args.log_n, args.batch
    n = 1 << log_n
    x = synth_words(n * batch, dev, 0x7F210002 + rank)

    # in-run parity sample: first 2 transforms of this rank's batch
    sample_in = x[: 2 * n].clone()

    launches_per_step = tf.lib().tf_ntt_launch_count(n, batch, 1)

    # settle: untimed passes until the GPU is in steady state.  A cold GPU runs the first passes at 4.2 -> 2.3 ms and only
    # reaches its steady 2.2 ms after ~50 (tools/step_times.py); some boxes have been seen to sit at ~1/4 speed (8.4 ms per
    # pass, every pass of a process) -- the shader clock measured here tells such a run from a slow kernel.  Run windows
    # of 25 passes until two consecutive windows agree within 2 % (at least 4 windows, at most 40), then the W warmup steps.
    settle = 0
    sclk_before = tf.lib().tf_debug_sclk_mhz()

    def copy_gbs():
        """Platform sanity reference: a plain 2 GiB device copy (read + write), GB/s."""
        y = torch.empty_like(x)
        y.copy_(x)
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(3):
            y.copy_(x)
        c1.record()
        torch.cuda.synchronize()
        return 3 * 2 * x.numel() * 8 / (c0.elapsed_time(c1) * 1e-3) / 1e9

    copy_before = copy_gbs()
    if not args.no_settle:
        prev = None
        for w in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(25):
                tf.device.ntt_(x, n, batch=batch)
            e1.record()
            torch.cuda.synchronize()
            cur = e0.elapsed_time(e1) / 25
            settle += 25
            if w >= 3 and prev is not None and abs(cur - prev) <= 0.02 * prev:
                break
            prev = cur
    for _ in range(args.warmup):
        tf.device.ntt_(x, n, batch=batch)
    barrier()

    # parity of the first warmup step is checked on a fresh run of the sample (x has been transformed W times)
    sample_gpu = sample_in.clone()
    tf.device.ntt_(sample_gpu, n, batch=2)
    torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        tf.device.ntt_(x, n, batch=batch)
    ev1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    ev_ms = ev0.elapsed_time(ev1)
    # per-step spread, measured AFTER the timed region (an event per step costs ~2 % when it sits inside it)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    evs[0].record()
    for i in range(10):
        tf.device.ntt_(x, n, batch=batch)
        evs[i + 1].record()
    torch.cuda.synchronize()
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(10))
    sclk_after = tf.lib().tf_debug_sclk_mhz()
    copy_after = copy_gbs()
    barrier()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_elems = world * batch * n * args.steps
    value = total_elems / elapsed / 1e9
    ms_per_step = elapsed / args.steps * 1e3

    # roofline of the dominant kernel (this rank)
    launches = launches_per_step * args.steps
    avg_launch_ms = ev_ms / launches
    alg_bytes_per_launch = 16.0 * batch * n / launches_per_step  # 16 B/element per transform, spread over its launches
    achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic_ntt.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("log_n") == log_n and tj.get("batch") == batch and tj.get("launches_per_step") == launches_per_step:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm",
        "kernel": "tfk::ntt_pass_kernel<false, 0, 0, false, true> (pass 1) and <false, 0, 0, true, false> (pass 2), one launch of each per step",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "algorithmic_bytes_per_launch": alg_bytes_per_launch,
        "avg_launch_ms": round(avg_launch_ms, 5),
        "launches_per_step": launches_per_step,
    }
    if log_n == 20:
        # Informational second roofline: what actually bounds the kernel is integer VALU issue.  Static
        # instruction counts of the two pass kernels (tools/isa_count.py on the committed build: 4114 + 3454 VALU
        # instructions per thread = 32 elements per pass) against the measured issue rate of the 64-bit integer building
        # blocks, 0.55 G wave-instructions/s per SIMD x 1024 SIMDs (profiles/microbench_r01_*.txt).
        valu_instr_per_element = (4114 + 3454) / 32.0
        wave_instr = batch * n * valu_instr_per_element / 64.0
        floor_ms = wave_instr / (1024 * 0.55e9) * 1e3
        roofline["valu_bound"] = {"valu_instr_per_element": round(valu_instr_per_element, 1), "floor_ms_per_step": round(floor_ms, 3),
                                  "frac": round(floor_ms / (ev_ms / args.steps), 3),
                                  "note": "integer VALU issue is the binding resource, not HBM"}

    out = {
        "metric": "goldilocks_ntt_gfelts_per_s",
        "value": round(value, 3),
        "unit": "GFelts/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{batch} x 2^{log_n}-point BFieldElement forward NTT per GPU, in place, device-resident (BASELINE configs[1])",
            "batch_per_gpu": batch,
            "n": n,
            "parallelism": f"batch-sharded x{world}, no data-path collective",
        },
        "roofline": roofline,
        "step_ms_after": {"min": round(step_ms[0], 4), "median": round(step_ms[len(step_ms) // 2], 4), "max": round(step_ms[-1], 4),
                          "note": "10 extra steps with an event each, after the timed region"},
        "settle_steps": settle,
        "sclk_mhz": {"before_settle": round(sclk_before, 0), "after_timed_region": round(sclk_after, 0)},
        "device_copy_gbs": {"before_settle": round(copy_before, 0), "after_timed_region": round(copy_after, 0),
                            "note": "torch copy of the 2 GiB workload buffer, read + write: a platform reference (normally ~4500-5000)"},
    }

    if rank == 0:
        # ---- cpu baseline + in-run parity (oracle = checker only)
        if world == 1 and not args.no_cpu_baseline:
            sample_host = sample_in.cpu().numpy().view(np.uint64)
            info, sample_out = cpu_baseline_ntt(log_n, batch, sample_host)
            out["cpu_baseline"] = info
            got = sample_gpu.cpu().numpy().view(np.uint64)
            out["parity"] = "bit-exact vs oracle on 2 transforms" if np.array_equal(got, sample_out) else "MISMATCH"
            if out["parity"] == "MISMATCH":
                print(json.dumps(out))
                raise SystemExit("GPU output differs from the oracle")
        elif world == 1:
            out["cpu_baseline"] = None
        # ---- side measurements (not part of `value`): Merkle leaves/s and XFE coset evaluation
        if world == 1 and not args.no_extra:
            out["extra"] = side_measurements(tf, torch, dev)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def side_measurements(tf, torch, dev):
    """BASELINE configs[2] and [3] shapes, a few iterations each (reported, not the headline value)."""
    extra = {}
    try:
        nl = 1 << 24
        leaves = synth_words(5 * nl, dev, 3)
        nodes = torch.empty(10 * nl, dtype=torch.int64, device=dev)
        for _ in range(6):  # the GPU has been idle during the CPU baseline: let the clocks come back
            tf.device.merkle_build(leaves, nl, nodes)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters):
            tf.device.merkle_build(leaves, nl, nodes)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        extra["merkle_2p24"] = {"ms": round(ms, 3), "leaves_per_s": round(nl / ms * 1e3, 1),
                                "hbm_frac_at_120B_per_leaf": round(120.0 * nl / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del leaves, nodes
        n, b = 1 << 22, 16
        c = synth_words(3 * n * b, dev, 4)
        o = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
        off = tf.BFieldElement.new(7)
        for _ in range(4):
            tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        extra["xfe_coset_eval_16x2p22"] = {"ms": round(ms, 3), "gfelts_per_s": round(n * b / ms / 1e6, 3),
                                          "hbm_frac_at_48B_per_point": round(48.0 * n * b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del c, o
        # the callers on either side of the path, device-resident
        def _timed(fn, reps=8):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        nh = 1 << 19
        pa, pb = synth_words(256 * nh, dev, 6), synth_words(256 * nh, dev, 7)
        po = torch.empty(256 * (2 * nh - 1), dtype=torch.int64, device=dev)
        ms = _timed(lambda: tf.device.poly_mul(pa, nh, pb, nh, po, batch=256))
        extra["fast_multiply_256x_2p19_by_2p19"] = {"ms": round(ms, 3), "output_coefficients_per_s": round(256 * (2 * nh - 1) / ms * 1e3, 1)}
        del pa, pb, po
        lv = synth_words(128 * (1 << 18), dev, 8)
        lo = torch.empty(128 * (1 << 21), dtype=torch.int64, device=dev)
        ms = _timed(lambda: tf.device.lde(lv, 1 << 18, tf.BFieldElement.new(1), lo, 1 << 21, off, batch=128))
        extra["lde_128x_2p18_to_2p21"] = {"ms": round(ms, 3), "g_points_per_s": round(128 * (1 << 21) / ms / 1e6, 3)}
        # the 128 codewords are now a column-major table of 2^21 rows: hash every row, build the tree
        tn = torch.empty(10 * (1 << 21), dtype=torch.int64, device=dev)
        ms = _timed(lambda: tf.device.merkle_from_columns(lo, 1 << 21, 128, tn))
        extra["merkle_from_columns_2p21_rows_x128"] = {"ms": round(ms, 3), "rows_per_s": round((1 << 21) / ms * 1e3, 1)}
        del lv, lo, tn
        # PCIe-inclusive figure of the host-pointer entry point (pageable numpy buffers, 32 x 2^20 BFE = 256 MiB each way)
        import time as _t

        import numpy as _np

        hb = 32
        hx = _np.random.default_rng(5).integers(0, 2 ** 63, size=hb * (1 << 20), dtype=_np.uint64)
        tf.ntt(hx, batch=hb)
        t0 = _t.perf_counter()
        tf.ntt(hx, batch=hb)
        dt = _t.perf_counter() - t0
        extra["host_pointer_ntt_32x2p20"] = {"ms": round(dt * 1e3, 2), "gfelts_per_s": round(hb * (1 << 20) / dt / 1e9, 3),
                                             "note": "tf_ntt_bfe on pageable host memory: H2D + 2 passes + D2H, synchronous"}
    except Exception as e:  # side measurements never invalidate the headline line
        extra["error"] = repr(e)
    return extra


if __name__ == "__main__":
    main()
