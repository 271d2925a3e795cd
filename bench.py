#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 2|5]

N > 1 may be launched either by the driver (python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...) or simply as `python bench.py --gpus N`: with no
WORLD_SIZE in the environment bench.py re-launches itself under torch.distributed.run, one rank per GPU.

Default workload (--config 2) = BASELINE.json configs[1]: a batch of 256 x 2^20-point BFieldElement forward NTTs per
GPU, device-resident in and out, in place (math/ntt.rs:67-82 semantics, bit-exact).  One "step" = one pass of the hot
path over the whole batch = one tf_ntt_bfe_dev call.  Independent transforms shard across ranks with no data-path
collective, so N ranks run N x 256 transforms: weak scaling.

Prints ONE JSON line (rank 0).  `value` = (N * 256 * 2^20 * K elements) / (max-over-ranks time of the K steps).
  roofline    : dominant kernel ntt_pass_kernel; achieved = algorithmic bytes per launch / average launch duration from
                HIP events over the timed region.  A 2^20 transform is 2 launches of that kernel (one per pass), and
                SURVEY.md 8(d) prices a transform at 16 B/element, so one launch carries 8 B/element x (elements it touches).
  cpu_baseline: the CPU oracle (C restatement of the reference algorithm, kind "port") timed on this box's host cores on
                the same 256 x 2^20 workload, one transform per thread (what a rayon caller of the single-threaded ntt()
                does).  Rank 0, N = 1 only.
  merkle      : the second half of BASELINE.json's metric -- Tip5 Merkle leaves/s on configs[2] (one 2^24-leaf tree per
                GPU, full node array, root all-gathered over RCCL when N > 1), with its own integer-VALU roofline and the
                oracle's par_new restatement (util_types/merkle_tree.rs:165-212) timed on the host cores in the same run;
                the GPU root is compared with the oracle's.
  coset_eval  : configs[3], 64 XFieldElement polynomials x 2^22 coefficients, fast_coset_evaluate (48 B/point roofline).
  config5     : configs[4] run as a STRONG-scaling job: 4096 NTTs of 2^20 points + 256 trees of 2^20 leaves split over the
                N ranks by sharding.shard_range, roots through sharding.gather_roots (RCCL all_gather).  `--config 5`
                makes this job the timed headline (`value` = its NTT GFelts/s, "scaling": "strong").
Inputs are the SplitMix64-seeded synthetic elements of SURVEY.md 8(d), generated on the device (tf_debug_fill_random_dev;
the oracle's tfo.fill_random is the same counter-based sequence, so every parity sample is reproducible from its seed).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P = 0xFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# Integer-VALU issue peak: 256 CUs x 4 SIMDs, one wave64 instruction of the 64-bit integer building blocks (v_mad_u64_u32,
# carry adds, v_lshl_add_u64 ...) per 4 cycles per SIMD (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.00 quad-cycles on the pass
# and Tip5 kernels, profiles/) at the 2.4 GHz peak engine clock.
VALU_PEAK_GWIPS = 256 * 4 * 2.4 / 4.0  # 614.4 G wave-instructions/s
SEED_C2, SEED_C3, SEED_C4, SEED_C5 = 0x7F210002, 0x7F210003, 0x7F210004, 0x7F210005  # seed_c = 0x7F21_0000 + c
SEED_CP = 0x7F210006  # the commitment-pipeline leg (not a BASELINE config)


def self_spawn(args):
    """`python bench.py --gpus N` without torch.distributed.run: start the N ranks ourselves."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def cpu_baseline_ntt(log_n, batch, seed):
    """Oracle leg: time the CPU restatement on the same workload shape; returns (dict, first two transforms' outputs)."""
    from oracle import tfo

    flags = tfo.use_native_build()
    n = 1 << log_n
    cores = os.cpu_count() or 1
    x = tfo.fill_random(n * batch, seed)
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
    info = {
        "value": round(multi, 4),
        "unit": "GFelts/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{batch} x 2^{log_n} BFE forward NTT (the full workload shape), one transform per thread; best of "
                  + ", ".join(f"{k} threads: {v:.3f}" for k, v in tried.items())
                  + f" GFelts/s on {cores} host CPUs; C restatement of math/ntt.rs:153-215 (oracle/tf_oracle.c)",
        "build_flags": flags,
        "single_thread_value": round(single, 5),
    }
    return info, None


def cpu_baseline_merkle(n_leaves, seed):
    """Oracle leg of the Merkle metric: par_new restatement on the same leaves; returns (dict, root, a node sample)."""
    from oracle import tfo

    flags = tfo.use_native_build()
    cores = os.cpu_count() or 1
    leaves = tfo.fill_random(5 * n_leaves, seed)
    ns = min(n_leaves, 1 << 20)  # single-thread figure on a bounded sample (sequential_new, ~1 s)
    t0 = time.perf_counter()
    tfo.merkle_build(leaves[: 5 * ns])
    single = ns / (time.perf_counter() - t0)
    tried, nodes = {}, None
    for th in sorted(set(max(1, c) for c in (32, 64, cores))):
        t0 = time.perf_counter()
        nodes = tfo.merkle_build(leaves, threads=th)
        tried[th] = n_leaves / (time.perf_counter() - t0)
    threads = max(tried, key=tried.get)
    info = {
        "value": round(tried[threads], 1),
        "unit": "leaves/s",
        "cores": threads,
        "kind": "port",
        "sample": f"the full 2^{n_leaves.bit_length() - 1}-leaf tree (MerkleTree::par_new restatement, util_types/merkle_tree.rs:165-212; bench shape "
                  "benches/merkle_tree.rs:11-40); best of " + ", ".join(f"{k} threads: {v / 1e6:.2f} M" for k, v in tried.items())
                  + f" leaves/s on {cores} host CPUs",
        "build_flags": flags,
        "single_thread_value": round(single, 1),
    }
    return info, nodes[5:10].copy(), nodes


def cpu_baseline_coset(n, batch, seed, offset_raw):
    """Oracle leg of configs[3]: `batch` XFieldElement polynomials of n coefficients, fast_coset_evaluate on the coset of order n,
    one polynomial per thread (benches/polynomial_coset.rs:15-47 is the reference's own, single-polynomial, shape).  Returns
    (dict, the evaluations of polynomial 0)."""
    from oracle import tfo

    flags = tfo.use_native_build()
    cores = os.cpu_count() or 1
    c = tfo.fill_random(3 * n * batch, seed)
    tfo.ntt(c[:n].copy())  # twiddle cache outside the timed region
    t0 = time.perf_counter()
    want0 = tfo.coset_evaluate(c[: 3 * n], offset_raw, n, width=3)
    single = n / (time.perf_counter() - t0) / 1e9
    tried = {}
    for th in sorted(set(max(1, min(k, batch)) for k in (16, 32, 64, cores))):
        t0 = time.perf_counter()
        tfo.coset_evaluate_batch(c, offset_raw, n, batch, width=3, threads=th)
        tried[th] = batch * n / (time.perf_counter() - t0) / 1e9
    threads = max(tried, key=tried.get)
    info = {
        "value": round(tried[threads], 5),
        "unit": "G points/s",
        "cores": threads,
        "kind": "port",
        "sample": f"all {batch} polynomials (2^{n.bit_length() - 1} XFE coefficients each), one polynomial per thread; best of "
                  + ", ".join(f"{k} threads: {v:.4f}" for k, v in tried.items())
                  + f" G points/s on {cores} host CPUs; scale + ntt restatement of math/polynomial.rs:760-773,1374-1399 (oracle/tf_oracle.c)",
        "build_flags": flags,
        "single_thread_value": round(single, 5),
    }
    return info, want0


def library_identity(tf):
    """What the timed library is: its ABI version and the hash of the sources it was built from (tf_source_hash, stamped by
    csrc/Makefile).  Stored profiles carry the same pair; a figure taken from a profile of ANOTHER build is dropped."""
    L = tf.lib()
    return {"tf_version": int(L.tf_version()), "source_hash": L.tf_source_hash().decode()}


def profile_matches(profile, ident):
    lib = (profile or {}).get("library")
    return bool(lib) and lib.get("source_hash") == ident["source_hash"] and lib.get("tf_version") == ident["tf_version"]


def sharding_all_true(ctx, flag):
    from twenty_first_amd import sharding

    return sharding.all_ranks_true(flag, device=ctx["dev"]) if ctx["use_dist"] else bool(flag)


def check_ntt_units(ctx, seed, units, n, threads=1):
    """Word-for-word parity of whole transforms against the oracle: unit u of the job is elements [u n, (u + 1) n) of the seed's
    counter-based sequence; the inputs are regenerated on the device and on the host, transformed by the HIP path and by the
    oracle (math/ntt.rs:153-215 restatement), and compared.  Returns True when every word of every unit agrees."""
    from oracle import tfo

    tf, torch, np, dev = ctx["tf"], ctx["torch"], ctx["np"], ctx["dev"]
    units = list(units)
    if not units:
        return True
    t = torch.empty(len(units) * n, dtype=torch.int64, device=dev)
    for i, u in enumerate(units):
        tf.device.fill_random(t[i * n:(i + 1) * n], seed, first_index=u * n)
    tf.device.ntt_(t, n, batch=len(units))
    torch.cuda.synchronize()
    got = t.cpu().numpy().view(np.uint64)
    host = np.concatenate([tfo.fill_random(n, seed, first_index=u * n) for u in units])
    want = tfo.ntt(host, batch=len(units), threads=max(1, min(threads, len(units))))
    return bool(np.array_equal(got, want))


def check_tree_units(ctx, seed, units, nl, threads=1):
    """Every node of whole trees against the oracle's par_new restatement (util_types/merkle_tree.rs:165-212): tree u of the job
    has leaves [u 5 nl, (u + 1) 5 nl) of the seed's sequence."""
    from oracle import tfo

    tf, torch, np, dev = ctx["tf"], ctx["torch"], ctx["np"], ctx["dev"]
    for u in units:
        lv = torch.empty(5 * nl, dtype=torch.int64, device=dev)
        tf.device.fill_random(lv, seed, first_index=u * 5 * nl)
        nodes = torch.empty(10 * nl, dtype=torch.int64, device=dev)
        tf.device.merkle_build(lv, nl, nodes)
        torch.cuda.synchronize()
        want = tfo.merkle_build(tfo.fill_random(5 * nl, seed, first_index=u * 5 * nl), threads=threads)
        if not np.array_equal(nodes.cpu().numpy().view(np.uint64)[5:], want[5:]):
            return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=(2, 5), help="2: BASELINE configs[1] per GPU, weak scaling (default); 5: configs[4], strong scaling")
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="transforms per GPU (config 2)")
    ap.add_argument("--no-settle", action="store_true", help="skip the untimed stabilisation passes before the warmup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even for one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--no-extra", action="store_true", help="headline NTT leg only (no Merkle / coset-evaluation / config-5 legs)")
    ap.add_argument("--c5-ntts", type=int, default=4096, help="config 5: transforms of 2^20 points in the whole job")
    ap.add_argument("--c5-trees", type=int, default=256, help="config 5: trees of 2^20 leaves in the whole job")
    ap.add_argument("--c5-log-n", type=int, default=20, help="diagnostic: log2 of config 5's transform length and leaves per tree (BASELINE: 20; the record names the size)")
    ap.add_argument("--merkle-log-leaves", type=int, default=24, help="diagnostic: log2 of the Merkle leg's leaves per tree (BASELINE: 24; the record names the size)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))
    # stdout carries exactly ONE line, the JSON record: RCCL prints a version banner to stdout when the process group comes up,
    # so everything else that writes to file descriptor 1 is sent to stderr and the record goes out through a private copy
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import twenty_first_amd as tf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available() or tf.lib().tf_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: the product has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: only {torch.cuda.device_count()} GPU(s) visible, --gpus {args.gpus} needs one per rank")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        # bring the communicator up NOW (RCCL builds its rings lazily inside the first collective: hundreds of milliseconds with
        # the GPU idle).  Left to the barrier in front of the timed region that idle period lets the clocks drop and the first timed
        # steps run cold: measured on one GPU with --force-dist, 1.97 ms per step over the 20 timed steps against 1.85 ms for the ten
        # steps after them (profiles/r03_bench_forcedist_before.json)
        _w = torch.zeros(1, dtype=torch.float64, device=dev)
        dist.all_reduce(_w, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not use_dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if not args.no_cpu_baseline:
        # The oracle is loaded by every rank (each checks its own shard).  Rank 0 goes first: it rebuilds the portable library if it is
        # stale -- so that no two ranks ever run the compiler on the same file -- and then switches ITSELF to the -O3 -march=native build
        # for the timed CPU legs (which has to be chosen before the oracle is first loaded); the other ranks load the portable one after.
        from oracle import tfo

        if rank == 0:
            tfo.build()
            tfo.use_native_build()
            tfo.lib()
        barrier()
        if rank != 0:
            tfo.lib()

    ctx = dict(tf=tf, torch=torch, dist=dist, np=np, dev=dev, world=world, rank=rank, use_dist=use_dist, barrier=barrier,
               max_over_ranks=max_over_ranks, args=args, ident=library_identity(tf), cpu_cache={})

    if args.config == 5:
        out = config5_leg(ctx, steps=args.steps, warmup=args.warmup, headline=True)
    else:
        out = ntt_headline(ctx)

    if not args.no_extra:
        # the other legs run on every rank (they contain collectives); rank 0 reports
        try:
            out["merkle"] = merkle_leg(ctx)
        except Exception as e:  # a side leg never invalidates the headline line
            out["merkle"] = {"error": repr(e)}
        if world == 1:
            try:
                out["coset_eval"] = coset_leg(ctx)
            except Exception as e:
                out["coset_eval"] = {"error": repr(e)}
        if args.config != 5:
            try:
                out["config5"] = config5_leg(ctx, steps=3, warmup=1, headline=False)
            except Exception as e:
                out["config5"] = {"error": repr(e)}
        if world == 1:
            try:
                out["commit_pipeline"] = commit_pipeline_leg(ctx)
            except SystemExit:
                raise
            except Exception as e:
                out["commit_pipeline"] = {"error": repr(e)}
            out["extra"] = side_measurements(tf, torch, dev)
    if use_dist:
        # the first multi-GPU record must be complete on its own: rank 0 runs the C++ host mirror's self-test (every visible device
        # driven through the C ABI from its own host thread, and one tf_*_multi call over all of them) and embeds its evidence lines
        if rank == 0:
            out["c_abi_selftest"] = c_abi_selftest(tf, np)
        barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        out["summary"] = summary_block(out)  # LAST key: the driver keeps the tail of the line
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


SUMMARY_LEGS = ("headline", "merkle", "coset_eval", "config5", "commit_pipeline")


def _short_parity(p):
    if not isinstance(p, str):
        return None
    if p.startswith("MISMATCH"):
        return "MISMATCH"
    if p.startswith("not checked"):
        return "not checked"
    return "green: " + p[:70]


def summary_block(out):
    """Every leg's number in <= 1.5 KB, emitted as the LAST key of the line: value / unit, ms_per_step, roofline.frac, cpu_baseline.value +
    cores, parity.  (The driver's record keeps only the tail of a long line; round 5's lost `merkle.value` that way.)"""
    def leg(d):
        if not isinstance(d, dict):
            return None
        if "error" in d:
            return {"error": str(d["error"])[:80]}
        cb = d.get("cpu_baseline") or {}
        rf = d.get("roofline") or {}
        r = {"value": d.get("value"), "unit": d.get("unit"), "ms": d.get("ms_per_step"), "frac": rf.get("frac"), "bound": rf.get("bound"),
             "cpu": cb.get("value"), "cores": cb.get("cores"), "parity": _short_parity(d.get("parity"))}
        return {k: v for k, v in r.items() if v is not None}
    sm = {"headline": leg(out)}
    for k in SUMMARY_LEGS[1:]:
        if k in out:
            sm[k] = leg(out[k])
    c5 = out.get("config5")
    if isinstance(c5, dict) and isinstance(c5.get("merkle"), dict):
        sm["config5"]["trees_leaves_per_s"] = c5["merkle"].get("value")
    cp = out.get("commit_pipeline")
    if isinstance(cp, dict) and "error" not in cp:
        for part in ("lde", "rows_and_tree"):
            if isinstance(cp.get(part), dict):
                sm["commit_pipeline"][part + "_ms"] = cp[part].get("ms")
                sm["commit_pipeline"][part + "_frac"] = (cp[part].get("roofline") or {}).get("frac")
        sm["commit_pipeline"]["lde_traffic_x"] = (cp.get("lde", {}).get("roofline") or {}).get("traffic_over_compulsory")
    sm["n_gpus"] = out.get("n_gpus")
    sm["library"] = (out.get("roofline") or {}).get("profile_matches_library", {}).get("library", {}).get("source_hash")
    return sm


def parse_summary_from_tail(text):
    """The `summary` object out of the (possibly truncated) tail of a bench line: what a reader of the driver's record does."""
    key = '"summary": '
    at = text.rfind(key)
    if at < 0:
        return None
    body = text[at + len(key):].strip()
    if body.endswith("}"):
        body = body[:-1]  # the closing brace of the whole record
    return json.loads(body)


# ------------------------------------------------------------------------------------------------ headline: configs[1]
def ntt_headline(ctx):
    tf, torch, np, dev, args = ctx["tf"], ctx["torch"], ctx["np"], ctx["dev"], ctx["args"]
    world, rank, barrier = ctx["world"], ctx["rank"], ctx["barrier"]
    log_n, batch = args.log_n, args.batch
    n = 1 << log_n
    x = torch.empty(n * batch, dtype=torch.int64, device=dev)
    # the whole job is ONE counter-based sequence: rank r holds elements [r * batch * n, (r + 1) * batch * n)
    tf.device.fill_random(x, SEED_C2, first_index=rank * batch * n)
    launches_per_step = tf.lib().tf_ntt_launch_count(n, batch, 1)

    # settle: untimed passes until the GPU is in steady state.  A cold GPU runs the first passes at 4.2 -> 2.3 ms and only
    # reaches its steady state after ~50 (tools/step_times.py); some boxes have been seen to sit at ~1/4 speed (8.4 ms per
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
        prev, best = None, None
        for w in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(25):
                tf.device.ntt_(x, n, batch=batch)
            e1.record()
            torch.cuda.synchronize()
            cur = e0.elapsed_time(e1) / 25
            settle += 25
            best = cur if best is None else min(best, cur)
            # steady = two consecutive windows agree within 2 % AND sit within 5 % of the best window seen so far
            if w >= 3 and prev is not None and abs(cur - prev) <= 0.02 * prev and cur <= 1.05 * best:
                break
            prev = cur
    for _ in range(args.warmup):
        tf.device.ntt_(x, n, batch=batch)
    barrier()

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
    # "median of >= 10 runs".  A SECOND pass of the same K steps, an event after every step (an event per step costs
    # ~2 % when it sits inside the timed region, so the single interval above stays free of them): `value` is priced on the median of
    # these K per-step intervals (max over ranks), the single-interval mean of the first pass is reported beside it.
    k2 = max(args.steps, 10)
    barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(k2 + 1)]
    evs[0].record()
    for i in range(k2):
        tf.device.ntt_(x, n, batch=batch)
        evs[i + 1].record()
    torch.cuda.synchronize()
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(k2))
    median_ms = 0.5 * (step_ms[(k2 - 1) // 2] + step_ms[k2 // 2])
    sclk_after = tf.lib().tf_debug_sclk_mhz()
    copy_after = copy_gbs()
    barrier()
    elapsed = ctx["max_over_ranks"](elapsed)
    median_ms = ctx["max_over_ranks"](median_ms * 1e-3) * 1e3

    total_elems = world * batch * n * args.steps
    mean_value = total_elems / elapsed / 1e9
    mean_ms_per_step = elapsed / args.steps * 1e3
    value = world * batch * n / (median_ms * 1e-3) / 1e9
    ms_per_step = median_ms

    # roofline of the dominant kernel (this rank)
    launches = launches_per_step * args.steps
    avg_launch_ms = ev_ms / launches
    alg_bytes_per_launch = 16.0 * batch * n / launches_per_step  # 16 B/element per transform, spread over its launches
    achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    # stored profiles (PMC passes of tools/prof_r02.sh) are only quoted when they were taken on THIS build of the library
    ident = ctx["ident"]
    tj = load_profile_json("hbm_traffic_ntt.json")
    vc = load_profile_json("valu_counts.json")
    tj_ok = bool(tj) and profile_matches(tj, ident) and tj.get("log_n") == log_n and tj.get("batch") == batch and tj.get("launches_per_step") == launches_per_step
    vc_ok = bool(vc) and profile_matches(vc, ident) and log_n == 20 and bool(vc.get("ntt_valu_wave_instr_per_transform_2p20"))
    traffic = tj.get("hbm_bytes_per_launch") if tj_ok else None
    busy = [k.get("valu_busy_frac_at_4_cycles") for k in (tj.get("per_kernel") or {}).values()] if tj_ok else []
    busy = [b for b in busy if b is not None]
    # `bound`: SURVEY.md 8(d) prices this kernel against HBM (16 B/element) and achieved / peak / frac are that roofline; the
    # resource that actually limits it, by the stored counters of this build, is VALU issue (two resident workgroups per CU keep
    # the vector ALU busy ~0.88 of the cycles while HBM runs at about half its rate) -- `limiting_resource` says which one the
    # profile shows, and `valu_bound` below is that second roofline.
    valu_limited = bool(busy) and min(busy) >= 0.75
    roofline = {
        "bound": "hbm",  # what achieved / peak / frac below are priced against (16 B per element and transform)
        "limiting_resource": "valu" if valu_limited else "hbm",
        "priced_against": "hbm (SURVEY.md 8(d): 16 B per element and transform)",
        "bound_evidence": (f"stored PMC profile of this build: VALU busy {min(busy):.2f}-{max(busy):.2f} of the cycles at 4 cycles per instruction on the two pass kernels "
                           "(profiles/hbm_traffic_ntt.json per_kernel)") if busy else "no stored profile of this build: the HBM roofline named by SURVEY.md 8(d)",
        "kernel": "tfk::ntt_pass_kernel (column pass + transposing pass; one launch of each per batch tile)",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "traffic_source": (f"profiles/hbm_traffic_ntt.json ({tj.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')}; a stored "
                           "profile of this kernel, shape and library build, NOT measured in this run)") if traffic else None,
        "profile_matches_library": {"hbm_traffic_ntt.json": tj_ok, "valu_counts.json": vc_ok, "library": ident},
        "algorithmic_bytes_per_launch": alg_bytes_per_launch,
        "avg_launch_ms": round(avg_launch_ms, 5),
        "launches_per_step": launches_per_step,
        "tile_bytes": int(tf.lib().tf_get_ntt_tile_bytes()),
        "tile_streams": int(tf.lib().tf_get_ntt_pipe()),
    }
    if vc_ok:
        # Informational second roofline: integer VALU issue.  The instruction count is the DYNAMIC one,
        # SQ_INSTS_VALU of the two pass kernels under rocprofv3 --pmc (profiles/valu_counts.json, tools/prof_r02.sh).
        wi = vc["ntt_valu_wave_instr_per_transform_2p20"] * batch
        gw = wi / (ev_ms / args.steps * 1e-3) / 1e9
        roofline["valu_bound"] = {"wave_instr_per_step": wi, "valu_instr_per_element": round(wi * 64.0 / (batch * n), 1),
                                  "achieved": round(gw, 1), "peak": VALU_PEAK_GWIPS, "unit": "G wave-instr/s",
                                  "frac": round(gw / VALU_PEAK_GWIPS, 3),
                                  "source": "instruction count: SQ_INSTS_VALU of the two pass kernels in profiles/valu_counts.json (a stored rocprofv3 --pmc "
                                            "profile of this library build, NOT this run) x this run's step time; peak = 1024 SIMDs x 2.4 GHz / 4 cycles: every "
                                            "instruction of these kernels is of the 4-cycle class (carry adds, v_mad_u64_u32, VOP3: 0.46-0.58 G wave-instr/s/SIMD in "
                                            "profiles/r03_instr_rates.txt against 0.93-1.15 for plain v_add_u32 / logic / right shifts); the clock the chip "
                                            "holds under this kernel is in the stored profile, not measured here"}

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
            "inputs": f"SplitMix64, seed 0x{SEED_C2:X} (SURVEY.md 8(d)), generated on the device",
        },
        "roofline": roofline,
        "timing": {"value_from": f"median of {k2} per-step event intervals (second pass of the same steps), max over ranks",
                   "per_step_ms": {"min": round(step_ms[0], 4), "median": round(median_ms, 4), "max": round(step_ms[-1], 4), "n": k2},
                   "single_interval": {"ms_per_step": round(mean_ms_per_step, 4), "value": round(mean_value, 3),
                                       "note": f"the contract's region: {args.steps} steps between two barriers + synchronisations, wall clock, max over ranks"}},
        "settle_steps": settle,
        "sclk_mhz": {"before_settle": round(sclk_before, 0), "after_timed_region": round(sclk_after, 0),
                     "note": "one-wave idle probe; the clock under load is in roofline.valu_bound (from GRBM_GUI_ACTIVE)"},
        "device_copy_gbs": {"before_settle": round(copy_before, 0), "after_timed_region": round(copy_after, 0),
                            "note": "torch copy of the 2 GiB workload buffer, read + write: a platform reference (normally ~4500-5000)"},
    }
    del x
    # ---- parity (every rank, every world size): whole transforms of THIS rank's shard against the oracle, word for word, on a
    # fresh run after the timed region.  One rank: the first 32 and the last transform of the batch; several ranks: the first
    # and the last transform of every rank's shard (rank r holds transforms [r batch, (r + 1) batch) of the job's sequence).
    cores = os.cpu_count() or 1
    if args.no_cpu_baseline:
        out["parity"] = "not checked (--no-cpu-baseline)"
        out["cpu_baseline"] = None
        return out
    first, last = rank * batch, rank * batch + batch - 1
    # several ranks: the first two and the last transform of the shard -- three transforms, so that the check call is above the
    # 2^21 words under which the planner switches to its narrow-tile kernels and runs the SAME pass kernels as the timed batch
    units = sorted(set(list(range(first, first + min(batch, 32))) + [last])) if world == 1 else sorted({first, min(first + 1, last), last})
    ok = check_ntt_units(ctx, SEED_C2, units, n, threads=max(1, min(32, cores // world)))
    all_ok = sharding_all_true(ctx, ok)
    out["parity"] = (("bit-exact vs oracle, word for word, on " + (f"{len(units)} transforms (the first {len(units) - 1} and the last of the batch)" if world == 1 else
                      f"the first two and the last transform of each of the {world} ranks' shards ({len(units) * world} transforms in calls of {len(units)}, the timed plan; every rank checked its own)"))
                     if all_ok else "MISMATCH")
    if not all_ok:
        if rank == 0:
            sys.stderr.write(json.dumps(out) + "\n")
        raise SystemExit(f"rank {rank}: GPU output differs from the oracle" if not ok else f"rank {rank}: another rank reported a mismatch")
    # ---- CPU baseline: rank 0, on the GPU box's host cores, the full single-GPU workload shape, after a barrier (the other
    # ranks sit idle at the next one)
    barrier()
    if rank == 0:
        info, _ = cpu_baseline_ntt(log_n, batch, SEED_C2)
        if world > 1:
            info["sample"] += f"; timed on rank 0 while the other {world - 1} rank(s) wait at a barrier"
        out["cpu_baseline"] = info
        ctx["cpu_cache"][("ntt", log_n, batch)] = info
    barrier()
    return out


# ------------------------------------------------------------------------------------------------ configs[2]: Merkle leg
def merkle_leg(ctx):
    """One 2^24-leaf tree per GPU (full node array); with N > 1 the N roots are all-gathered inside the timed region."""
    tf, torch, np, dev, args = ctx["tf"], ctx["torch"], ctx["np"], ctx["dev"], ctx["args"]
    world, rank, barrier, use_dist = ctx["world"], ctx["rank"], ctx["barrier"], ctx["use_dist"]
    from twenty_first_amd import sharding

    log_nl = args.merkle_log_leaves
    nl = 1 << log_nl
    leaves = torch.empty(5 * nl, dtype=torch.int64, device=dev)
    tf.device.fill_random(leaves, SEED_C3, first_index=rank * 5 * nl)
    nodes = torch.empty(10 * nl, dtype=torch.int64, device=dev)

    def step():
        tf.device.merkle_build(leaves, nl, nodes)
        if use_dist:
            return sharding.gather_roots(nodes[5:10].reshape(1, 5), world)
        return nodes[5:10].reshape(1, 5)

    for _ in range(6):  # the GPU may have been idle during a CPU baseline: let the clocks come back
        step()
    iters = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        roots = step()
    e1.record()
    torch.cuda.synchronize()
    elapsed = ctx["max_over_ranks"](time.perf_counter() - t0)
    ms = elapsed / iters * 1e3
    res = {
        "metric": "tip5_merkle_leaves_per_s",
        "value": round(world * nl / (elapsed / iters), 1),
        "unit": "leaves/s",
        "n_gpus": world,
        "scaling": "weak",
        "ms_per_step": round(ms, 4),
        "device_ms_per_tree": round(e0.elapsed_time(e1) / iters, 4),
        "config": {"workload": f"one 2^{log_nl}-leaf Tip5 Merkle tree per GPU: hash_pair ladder to the full 2^{log_nl + 1}-node array" + (" (BASELINE configs[2])" if log_nl == 24 else " (NOT the BASELINE size)")
                               + ("; roots all-gathered over RCCL inside the timed region" if use_dist else ""),
                   "inputs": f"SplitMix64, seed 0x{SEED_C3:X}"},
        "hbm_frac_at_120B_per_leaf": round(120.0 * nl / (e0.elapsed_time(e1) / iters * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
    }
    vc = load_profile_json("valu_counts.json")
    res["profile_matches_library"] = {"valu_counts.json": bool(vc) and profile_matches(vc, ctx["ident"])}
    if vc and profile_matches(vc, ctx["ident"]) and vc.get("merkle_valu_wave_instr_per_tree_2p24") and log_nl == 24:
        # Vector-ALU roofline of the level sweep.  Since round 6 the MDS runs on v_mfma_i32_16x16x64_i8, a pipe that -- unlike the f64 one --
        # works BESIDE the vector ALU (profiles/r05_mfma_valu_mix.txt): the bound is the vector instructions alone, 4 issue cycles each;
        # the matrix pipe's own occupancy (16 cycles per MFMA) is reported beside it.  Counts: SQ_INSTS_VALU (which includes the MFMAs)
        # and SQ_INSTS_MFMA of a stored profile of this library build.
        wi = vc["merkle_valu_wave_instr_per_tree_2p24"]
        mf = vc.get("merkle_mfma_wave_instr_per_tree_2p24") or 0.0
        secs = e0.elapsed_time(e1) / iters * 1e-3
        cyc = 4.0 * (wi - mf)
        g = cyc / secs / 1e9
        peak = 1024 * 2.4  # G issue cycles / s: 1024 SIMDs x 2.4 GHz
        res["roofline"] = {"bound": "valu", "kernel": "tfk::tip5_hash_pairs_mx_kernel (level sweep, 4 lanes per permutation, MDS on v_mfma_i32_16x16x64_i8) + 16-lane kernels near the top",
                           "achieved": round(g, 1), "peak": round(peak, 1), "unit": "G vector-ALU issue cycles/s (4 per VALU instruction)", "frac": round(g / peak, 3),
                           "matrix_pipe_frac": round(16.0 * mf / secs / 1e9 / peak, 3),
                           "valu_wave_instr_per_tree": wi, "mfma_wave_instr_per_tree": mf,
                           "valu_instr_per_hash_pair_per_lane_quartet": round((wi - mf) * 16.0 / (nl - 1), 1), "mfma_per_16_hash_pairs": round(mf * 16.0 / (nl - 1), 2),
                           "status": "per 16 hash_pairs and round: 12 i8 MFMA (192 cycles, beside the vector ALU) + ~285 VALU instructions (180 of them the twelve x^7 = 36 "
                                     "Montgomery products per quartet column, 8 the XORs that bias the bytes, 56 + 20 the recombination of the ten planes, the rest byte look-ups); "
                                     "frac < 1 is the clock the chip holds under this mix and the ramp of the small levels (DESIGN.md 4.3)",
                           "source": "instruction counts: SQ_INSTS_VALU / SQ_INSTS_MFMA under rocprofv3 --pmc (profiles/valu_counts.json, a stored profile of this library build, "
                                     "NOT this run) x this run's time; peak = 1024 SIMDs x 2.4 GHz"}
    if use_dist and world & (world - 1) == 0:
        # the same 2^24-leaf tree as ONE tree across the ranks (reference's subtree split, sharding.sharded_tree): rank g builds
        # the subtree over leaves [g n / G, (g + 1) n / G), the G subtree roots are all-gathered, every rank finishes the top
        try:
            per = nl // world
            sl = torch.empty(5 * per, dtype=torch.int64, device=dev)
            tf.device.fill_random(sl, SEED_C3, first_index=rank * 5 * per)   # the single-GPU tree's leaves, this rank's slice
            sub_nodes = torch.empty(10 * per, dtype=torch.int64, device=dev)
            top_nodes = torch.empty(10 * world, dtype=torch.int64, device=dev)

            def build_sub(l):
                tf.device.merkle_build(l.reshape(-1), per, sub_nodes)
                return sub_nodes.view(-1, 5)

            def finish(r):
                if world == 1:
                    return r
                tf.device.merkle_build(r.reshape(-1).contiguous(), world, top_nodes)
                return top_nodes.view(-1, 5)

            for _ in range(3):
                sroot, _, _ = sharding.sharded_tree(sl.view(-1, 5), build_sub, finish)
            barrier()
            t0 = time.perf_counter()
            for _ in range(iters):
                sroot, _, _ = sharding.sharded_tree(sl.view(-1, 5), build_sub, finish)
            barrier()
            dt = ctx["max_over_ranks"](time.perf_counter() - t0) / iters
            res["single_tree_sharded"] = {"value": round(nl / dt, 1), "unit": "leaves/s", "ms_per_tree": round(dt * 1e3, 4), "scaling": "strong",
                                          "root": tf.Digest.to_hex(sroot.cpu().numpy().view(np.uint64)),
                                          "note": f"one 2^{log_nl}-leaf tree over {world} ranks: subtrees of 2^{log_nl} / {world} leaves + all_gather of {world} roots + {world.bit_length() - 1} finishing levels"}
            del sl, sub_nodes
        except Exception as e:
            res["single_tree_sharded"] = {"error": repr(e)}
    if not args.no_cpu_baseline:
        # parity, every rank: the root and 69k sampled nodes of THIS rank's 2^24-leaf tree against the oracle's par_new on the same
        # leaves (rank r's leaves are elements [r 5 nl, (r + 1) 5 nl) of the seed's sequence).  Rank 0 takes the oracle tree from its
        # timed CPU leg; the other ranks build theirs first, then wait while rank 0 times the baseline on an otherwise idle host.
        from oracle import tfo

        cores = os.cpu_count() or 1
        idx = np.unique(np.concatenate([np.arange(1, min(4096, 2 * nl)), np.random.default_rng(3).integers(1, 2 * nl, 1 << 16)]))
        got_root = roots[rank if use_dist else 0].cpu().numpy().view(np.uint64)
        got_nodes = nodes.view(-1, 5)[torch.from_numpy(idx).to(dev)].cpu().numpy().view(np.uint64)
        ok = True
        if rank != 0:
            want = tfo.merkle_build(tfo.fill_random(5 * nl, SEED_C3, first_index=rank * 5 * nl), threads=max(1, cores // world)).reshape(-1, 5)
            ok = bool(np.array_equal(got_root, want[1]) and np.array_equal(got_nodes, want[idx]))
            del want
        barrier()
        if rank == 0:
            info, want_root, want_nodes = cpu_baseline_merkle(nl, SEED_C3)
            if world > 1:
                info["sample"] += f"; timed on rank 0 while the other {world - 1} rank(s) wait at a barrier"
            res["cpu_baseline"] = info
            ok = bool(np.array_equal(got_root, want_root) and np.array_equal(got_nodes, want_nodes.reshape(-1, 5)[idx]))
            res["root"] = tf.Digest.to_hex(got_root)
        all_ok = sharding_all_true(ctx, ok)
        res["parity"] = (("root and 69k sampled nodes match the oracle's par_new" + (f" on every one of the {world} ranks' trees" if world > 1 else ""))
                         if all_ok else "MISMATCH")
        if not all_ok:
            raise SystemExit(f"rank {rank}: GPU Merkle tree differs from the oracle" if not ok else f"rank {rank}: another rank reported a Merkle mismatch")
    del leaves, nodes
    return res


# ------------------------------------------------------------------------------------------------ configs[3]: XFE coset evaluation
def coset_bound(ctx):
    """The configs[3] roofline is priced against HBM (48 B/point): `bound` says so; `limiting_resource` is "valu"
    when the stored PMC profile of this library build shows both PRE2 pass kernels VALU-busy (profiles/valu_counts.json)."""
    vc = load_profile_json("valu_counts.json")
    busy = (vc or {}).get("coset_eval_valu_busy_frac_at_4_cycles") if profile_matches(vc, ctx["ident"]) else None
    if busy and min(busy) >= 0.75:
        return {"bound": "hbm", "limiting_resource": "valu", "priced_against": "hbm (48 B per point)",
                "bound_evidence": f"stored PMC profile of this build: VALU busy {min(busy):.2f}-{max(busy):.2f} on the two pass kernels (profiles/valu_counts.json)"}
    return {"bound": "hbm", "limiting_resource": "hbm", "priced_against": "hbm (48 B per point)",
            "bound_evidence": "no stored profile of this build" if not busy else f"stored profile: VALU busy {min(busy):.2f}-{max(busy):.2f}"}


def coset_leg(ctx):
    tf, torch, np, dev, args = ctx["tf"], ctx["torch"], ctx["np"], ctx["dev"], ctx["args"]
    n, b = 1 << 22, 64
    c = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
    tf.device.fill_random(c, SEED_C4)
    o = torch.empty(3 * n * b, dtype=torch.int64, device=dev)
    off = tf.BFieldElement.new(7)
    for _ in range(4):
        tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3)
    torch.cuda.synchronize()
    iters = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gbs = 48.0 * n * b / (ms * 1e-3) / 1e9
    # the same call on the three-pass plan of rounds 1-2 (tf_set_ntt_two_pass(0)), in this run on this box: what the two-pass plan
    # (a 2048-point pass as pairs of 1024-point workgroups) buys
    three_ms = None
    try:
        tf.lib().tf_set_ntt_two_pass(0)
        for _ in range(2):
            tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3)
        e1.record()
        torch.cuda.synchronize()
        three_ms = e0.elapsed_time(e1) / 5
    finally:
        tf.lib().tf_set_ntt_two_pass(-1)
    tf.device.coset_evaluate(c, n, off, o, n, batch=b, width=3)  # the parity sample below is the two-pass plan's output
    torch.cuda.synchronize()
    res = {
        "metric": "xfe_coset_evaluate_gpoints_per_s",
        "value": round(n * b / ms / 1e6, 3),
        "unit": "G points/s",
        "ms_per_step": round(ms, 4),
        "config": {"workload": "64 XFieldElement polynomials x 2^22 coefficients, fast_coset_evaluate(offset = 7, order = 2^22) (BASELINE configs[3])",
                   "inputs": f"SplitMix64, seed 0x{SEED_C4:X}"},
        "roofline": dict(coset_bound(ctx), achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                         algorithmic_bytes_per_point=48, launches_per_step=int(tf.lib().tf_ntt_launch_count(n, b, 3)),
                         plan="two global passes (2048 x 2048 points; 32 B/point through HBM each way)"),
        "three_pass_plan_ms": round(three_ms, 4) if three_ms else None,
    }
    if not args.no_cpu_baseline:
        got0 = o[: 3 * n].cpu().numpy().view(np.uint64)
        del c, o  # the CPU leg holds the same 6 GiB on the host; free the device side first
        info, want = cpu_baseline_coset(n, b, SEED_C4, off)
        res["cpu_baseline"] = info
        ok = np.array_equal(got0, want)
        res["parity"] = "polynomial 0 bit-exact vs oracle (all 3 * 2^22 words)" if ok else "MISMATCH"
        if not ok:
            raise SystemExit("GPU coset evaluation differs from the oracle")
        return res
    del c, o
    return res


# ------------------------------------------------------------------------------------------------ configs[4]: the sharded job
def config5_leg(ctx, steps, warmup, headline):
    """4096 x 2^20 NTTs + 256 trees of 2^20 leaves, split contiguously over the ranks (sharding.shard_range); the roots of
    all trees are all-gathered with sharding.gather_roots (RCCL).  Strong scaling: the job is fixed, N varies."""
    tf, torch, np, dev, args = ctx["tf"], ctx["torch"], ctx["np"], ctx["dev"], ctx["args"]
    world, rank, barrier, use_dist = ctx["world"], ctx["rank"], ctx["barrier"], ctx["use_dist"]
    from twenty_first_amd import sharding

    log_n5 = args.c5_log_n
    n = nl = 1 << log_n5
    t_lo, t_hi = sharding.shard_range(args.c5_ntts, world, rank)
    m_lo, m_hi = sharding.shard_range(args.c5_trees, world, rank)
    nt, nm = t_hi - t_lo, m_hi - m_lo
    x = torch.empty(max(nt, 1) * n, dtype=torch.int64, device=dev)
    tf.device.fill_random(x[: nt * n], SEED_C5, first_index=t_lo * n)
    leaves = torch.empty(max(nm, 1) * 5 * nl, dtype=torch.int64, device=dev)
    tf.device.fill_random(leaves[: nm * 5 * nl], SEED_C5 ^ (1 << 40), first_index=m_lo * 5 * nl)
    nodes = torch.empty(max(nm, 1) * 10 * nl, dtype=torch.int64, device=dev)
    roots_local = torch.empty(nm * 5, dtype=torch.int64, device=dev)

    def ntt_phase():
        if nt:
            tf.device.ntt_(x[: nt * n], n, batch=nt)

    def tree_phase():
        if nm:
            tf.device.merkle_build(leaves[: nm * 5 * nl], nl, nodes[: nm * 10 * nl], batch=nm)
            roots_local.copy_(nodes[: nm * 10 * nl].view(nm, 2 * nl, 5)[:, 1, :].reshape(-1))
        if use_dist:
            return sharding.gather_roots(roots_local.view(-1, 5), args.c5_trees)
        return roots_local.view(-1, 5)

    for _ in range(warmup):
        ntt_phase()
        roots = tree_phase()
    barrier()
    t_ntt = t_tree = 0.0
    for _ in range(steps):
        barrier()
        t0 = time.perf_counter()
        ntt_phase()
        barrier()
        t1 = time.perf_counter()
        roots = tree_phase()
        barrier()
        t2 = time.perf_counter()
        t_ntt += t1 - t0
        t_tree += t2 - t1
    t_ntt, t_tree = ctx["max_over_ranks"](t_ntt), ctx["max_over_ranks"](t_tree)
    ntt_val = args.c5_ntts * n * steps / t_ntt / 1e9
    tree_val = args.c5_trees * nl * steps / t_tree
    # integrity of the collective: every rank ends with all roots, in batch order; rank 0's own slice sits where it belongs
    assert roots.shape == (args.c5_trees, 5)
    assert torch.equal(roots[m_lo:m_hi].reshape(-1), roots_local)
    res = {
        "metric": "goldilocks_ntt_gfelts_per_s",
        "value": round(ntt_val, 3),
        "unit": "GFelts/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round((t_ntt + t_tree) / steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": f"{args.c5_ntts} x 2^{log_n5}-point BFE NTTs + {args.c5_trees} x 2^{log_n5}-leaf Tip5 Merkle trees, the whole job split contiguously over "
                               f"{world} GPU(s), roots all-gathered" + (" (BASELINE configs[4])" if (log_n5, args.c5_ntts, args.c5_trees) == (20, 4096, 256) else " (NOT the BASELINE shape)"),
                   "ntts_this_rank": nt, "trees_this_rank": nm,
                   "parallelism": f"contiguous batch split x{world} (sharding.shard_range); one RCCL all_gather of 40-byte roots per step",
                   "inputs": f"SplitMix64, seed 0x{SEED_C5:X}"},
        "ntt_ms_per_step": round(t_ntt / steps * 1e3, 4),
        "merkle": {"metric": "tip5_merkle_leaves_per_s", "value": round(tree_val, 1), "unit": "leaves/s",
                   "ms_per_step": round(t_tree / steps * 1e3, 4), "includes": "tree builds + root gather"},
        "root_of_tree_0": tf.Digest.to_hex(roots[0].cpu().numpy().view(np.uint64)),
    }
    if headline:
        alg = 16.0 * nt * n
        gbs = alg * steps / t_ntt / 1e9
        res["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                           "traffic": None, "note": "this rank's NTT phase: 16 B/element over the max-over-ranks time; the kernels are the headline's "
                                                    "(its roofline block says which resource bounds them)"}
    del x, leaves, nodes
    # ---- one digest for the whole job (every world size must print the same one): Tip5 hash_varlen of the gathered roots, by the
    # HIP path; every rank checks it against the oracle's hash_varlen of the same roots and against every other rank's digest
    dig = torch.empty(5, dtype=torch.int64, device=dev)

    def gpu_hash_varlen(flat):
        tf.device.tip5_hash_varlen_rows(flat, flat.numel(), dig)
        return dig

    digest = sharding.roots_digest(roots, gpu_hash_varlen)
    torch.cuda.synchronize()
    res["roots_digest"] = tf.Digest.to_hex(digest.cpu().numpy().view(np.uint64))
    same = sharding.identical_on_all_ranks(roots) and sharding.identical_on_all_ranks(digest)
    if not same:
        raise SystemExit(f"rank {rank}: the gathered roots / their digest differ between the ranks")
    res["roots_identical_on_all_ranks"] = True
    if args.no_cpu_baseline:
        res["parity"] = "not checked (--no-cpu-baseline)"
        res["cpu_baseline"] = None
        return res
    # ---- parity, every rank: the first and the last transform and the first and the last tree (every node) of THIS rank's shard
    # against the oracle, plus the digest
    from oracle import tfo

    cores = os.cpu_count() or 1
    th = max(1, min(16, cores // world))
    ok = bool(np.array_equal(digest.cpu().numpy().view(np.uint64), tfo.hash_varlen(roots.cpu().numpy().view(np.uint64).reshape(-1))))
    # (three transforms per check call: above 2^21 words, i.e. the pass kernels of the timed batch, not the narrow-tile plan)
    ok = ok and check_ntt_units(ctx, SEED_C5, sorted({t_lo, min(t_lo + 1, t_hi - 1), t_hi - 1}) if nt else [], n, threads=3)
    tree_units = sorted({m_lo, m_hi - 1}) if nm else []
    ok = ok and check_tree_units(ctx, SEED_C5 ^ (1 << 40), tree_units, nl, threads=th)
    # ... and the gathered roots of those trees are the ones this rank just rebuilt and checked node by node
    all_ok = sharding_all_true(ctx, ok)
    res["parity"] = ((f"bit-exact vs oracle on every rank: first two + last transform and first + last tree (all 2^{log_n5 + 1} nodes) of each of the {world} shard(s), "
                      "and the roots digest (oracle hash_varlen of the gathered roots)") if all_ok else "MISMATCH")
    if not all_ok:
        raise SystemExit(f"rank {rank}: config 5 differs from the oracle" if not ok else f"rank {rank}: another rank reported a config-5 mismatch")
    barrier()
    if rank == 0:
        # CPU baseline on a REDUCED sample of the job (the whole job is 4096 transforms + 256 trees: ~3 minutes of host time):
        # 256 of the 4096 transforms, one per thread (the headline's CPU leg when it ran in this process), and 4 of the 256 trees
        nb5 = min(256, args.c5_ntts)
        info = ctx["cpu_cache"].get(("ntt", log_n5, nb5))
        reused = info is not None
        if info is None:
            info, _ = cpu_baseline_ntt(log_n5, nb5, SEED_C5)
        info = dict(info)
        tfo.use_native_build()
        lv = tfo.fill_random(5 * nl, SEED_C5 ^ (1 << 40))
        tried = {}
        for tth in sorted(set(max(1, k) for k in (32, 64, cores))):
            t0 = time.perf_counter()
            tfo.merkle_build(lv, threads=tth)
            tried[tth] = nl / (time.perf_counter() - t0)
        bt = max(tried, key=tried.get)
        info["sample"] = (f"REDUCED sample of the job: NTT = {nb5} of the {args.c5_ntts} transforms" + (" (the headline leg's timing, same shape, this process)" if reused else "")
                          + "; " + info["sample"] + f" | Merkle = 1 of the {args.c5_trees} trees (2^{log_n5} leaves, par_new restatement), best of "
                          + ", ".join(f"{k} threads: {v / 1e6:.2f} M" for k, v in tried.items()) + " leaves/s"
                          + (f"; timed on rank 0 while the other {world - 1} rank(s) wait at a barrier" if world > 1 else ""))
        info["merkle_value"] = round(tried[bt], 1)
        info["merkle_unit"] = "leaves/s"
        info["merkle_cores"] = bt
        res["cpu_baseline"] = info
    barrier()
    return res


def commit_pipeline_leg(ctx):
    """One prover-shaped composite, device-resident end to end: 128 columns of 2^18 values on the coset 1 * <w>
    -> low-degree extension to 2^21 points on 7 * <w'> (tf_lde_bfe_dev = fast_coset_interpolate + fast_coset_evaluate,
    math/polynomial.rs:1907-1918, :1374-1399) -> hash_varlen of every row of that column-major table (tip5/mod.rs:617-623)
    -> Merkle tree over the 2^21 row digests (util_types/merkle_tree.rs:165-212).  The root is compared with the oracle's, the oracle
    pipeline is timed on all host cores, and each half is priced against its roofline."""
    tf, torch, np, dev, args = ctx["tf"], ctx["torch"], ctx["np"], ctx["dev"], ctx["args"]
    cols, log_n, log_m = 128, 18, 21
    n, m = 1 << log_n, 1 << log_m
    one, seven = tf.BFieldElement.new(1), tf.BFieldElement.new(7)
    vals = torch.empty(cols * n, dtype=torch.int64, device=dev)
    tf.device.fill_random(vals, SEED_CP)
    ext = torch.empty(cols * m, dtype=torch.int64, device=dev)
    nodes = torch.empty(10 * m, dtype=torch.int64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step(timed=False):
        if timed:
            ev[0].record()
        tf.device.lde(vals, n, one, ext, m, seven, batch=cols)
        if timed:
            ev[1].record()
        tf.device.merkle_from_columns(ext, m, cols, nodes)
        if timed:
            ev[2].record()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    iters, lde_ms, tree_ms = 8, 0.0, 0.0
    t0 = time.perf_counter()
    for _ in range(iters):
        step(True)
        torch.cuda.synchronize()
        lde_ms += ev[0].elapsed_time(ev[1])
        tree_ms += ev[1].elapsed_time(ev[2])
    wall_ms = (time.perf_counter() - t0) / iters * 1e3
    lde_ms, tree_ms = lde_ms / iters, tree_ms / iters
    row_len = cols
    perms = m * (row_len // 10 + 1) + (m - 1)  # hash_varlen of a 128-word row absorbs 13 chunks (12 full + the padded one); the tree adds m - 1
    lde_bytes = 8.0 * cols * (n + m)           # compulsory: 2^18 words read and 2^21 words written per column (verdict r5 item 6)
    res = {
        "metric": "commit_pipeline_rows_per_s", "value": round(m / ((lde_ms + tree_ms) * 1e-3), 1), "unit": "rows/s",
        "ms_per_step": round(lde_ms + tree_ms, 4), "wall_ms_per_step": round(wall_ms, 4),
        "config": {"workload": f"{cols} BFieldElement columns x 2^{log_n} values -> LDE to 2^{log_m} points (coset 7) -> hash_varlen of the 2^{log_m} rows -> Merkle tree; device-resident",
                   "inputs": f"SplitMix64, seed 0x{SEED_CP:X}"},
        "lde": {"ms": round(lde_ms, 4), "g_points_per_s": round(cols * m / lde_ms / 1e6, 3),
                "roofline": {"bound": "hbm", "achieved": round(lde_bytes / (lde_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(lde_bytes / (lde_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": lde_bytes,
                             "note": "compulsory bytes of the whole low-degree extension: every column's 2^18 values read once, its 2^21 evaluations written once "
                                     "(8 B each).  The kernels move more: two passes of the inverse transform over 2^18 points and two of the 8-coset forward "
                                     "transform, whose first pass reads the coefficients once per coset -- `traffic` below is what the memory side saw"}},
        "rows_and_tree": {"ms": round(tree_ms, 4), "permutations": perms, "g_permutations_per_s": round(perms / tree_ms / 1e6, 3),
                          "hbm_frac": round((8.0 * cols * m + 120.0 * m) / (tree_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    pj = load_profile_json("pipeline_counters.json")  # tools/prof_r02.sh on tools/prof_target.py --pipeline (make_profile_records.py)
    if pj and profile_matches(pj, ctx["ident"]):
        if pj.get("lde_hbm_bytes_per_call"):
            res["lde"]["roofline"]["traffic"] = pj["lde_hbm_bytes_per_call"]
            res["lde"]["roofline"]["traffic_over_compulsory"] = round(pj["lde_hbm_bytes_per_call"] / lde_bytes, 2)
            res["lde"]["roofline"]["traffic_source"] = "profiles/pipeline_counters.json (FETCH_SIZE x 2 + WRITE_SIZE of the LDE's kernels, a stored profile of this library build)"
            res["lde"]["per_kernel"] = pj.get("lde_kernels")
        if pj.get("rows_valu_wave_instr_per_call"):
            wi, mf = pj["rows_valu_wave_instr_per_call"], pj.get("rows_mfma_wave_instr_per_call") or 0.0
            g = 4.0 * (wi - mf) / (tree_ms * 1e-3) / 1e9
            res["rows_and_tree"]["roofline"] = {"bound": "valu", "achieved": round(g, 1), "peak": round(1024 * 2.4, 1), "unit": "G vector-ALU issue cycles/s (4 per VALU instruction)",
                                                "frac": round(g / (1024 * 2.4), 3), "matrix_pipe_frac": round(16.0 * mf / (tree_ms * 1e-3) / 1e9 / (1024 * 2.4), 3),
                                                "valu_instr_per_permutation_per_lane_quartet": round((wi - mf) * 16.0 / perms, 1),
                                                "per_kernel": pj.get("rows_kernels"),
                                                "source": "SQ_INSTS_VALU / SQ_INSTS_MFMA of tip5_hash_table_rows_mx_kernel and the tree's kernels under rocprofv3 --pmc "
                                                          "(profiles/pipeline_counters.json, a stored profile of this library build) x this run's time"}
    if args.no_cpu_baseline:
        res["parity"] = "not checked (--no-cpu-baseline)"
        res["cpu_baseline"] = None
        return res
    from oracle import tfo

    flags = tfo.use_native_build()
    cores = os.cpu_count() or 1
    th = max(1, min(64, cores))
    got_root = nodes[5:10].cpu().numpy().view(np.uint64)
    got_ext_col0 = ext[:m].cpu().numpy().view(np.uint64)
    del vals, ext, nodes
    hv = tfo.fill_random(cols * n, SEED_CP)
    tfo.ntt(hv[:n].copy())
    tfo.ntt(np.zeros(m, dtype=np.uint64))  # twiddle caches outside the timed region
    t0 = time.perf_counter()
    co = tfo.intt(hv, batch=cols, threads=th)            # offset 1: fast_coset_interpolate is the inverse transform
    hext = tfo.coset_evaluate_batch(co, seven, m, cols, threads=th)
    t_lde = time.perf_counter() - t0
    rows = np.ascontiguousarray(hext.reshape(cols, m).T).reshape(-1)  # the same table row-major (not timed: a CPU prover would keep it that way)
    t0 = time.perf_counter()
    digs = tfo.hash_varlen_rows(rows, row_len, threads=cores)
    t_rows = time.perf_counter() - t0
    t0 = time.perf_counter()
    want_nodes = tfo.merkle_build(digs, threads=th)
    t_tree = time.perf_counter() - t0
    ok = bool(np.array_equal(got_root, want_nodes[5:10]) and np.array_equal(got_ext_col0, hext[:m]))
    res["root"] = tf.Digest.to_hex(got_root)
    res["parity"] = "root of the tree and all 2^21 points of codeword 0 match the oracle pipeline (inverse transform -> coset evaluation -> hash_varlen rows -> par_new)" if ok else "MISMATCH"
    res["cpu_baseline"] = {"value": round(m / (t_lde + t_rows + t_tree), 1), "unit": "rows/s", "cores": cores, "kind": "port",
                           "seconds": {"lde": round(t_lde, 3), "hash_rows": round(t_rows, 3), "tree": round(t_tree, 3)},
                           "sample": f"the whole pipeline once: {cols} columns, one per thread on {th} threads for the transforms (intt + coset evaluation restatements), "
                                     f"the 2^{log_m} rows dealt to {cores} threads for hash_varlen, par_new on {th} threads; the column-major -> row-major copy in between is not timed",
                           "build_flags": flags}
    if not ok:
        raise SystemExit("commit pipeline: GPU result differs from the oracle")
    return res


def c_abi_selftest(tf, np):
    """twenty-first_amd/host/selftest (C++ mirror over the C ABI) + one tf_ntt_bfe_multi / tf_merkle_root_multi call over every
    visible device from this process, compared with the single-device words."""
    rec = {}
    try:
        exe = os.path.join(ROOT, "twenty-first_amd", "host", "selftest")
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        keep = [l.strip() for l in r.stdout.splitlines() if ("host thread" in l or "tf_*_multi" in l or "C ABI from" in l or "selftest:" in l or "subtree" in l)]
        rec["selftest"] = {"rc": r.returncode, "lines": keep, "stderr_tail": r.stderr.strip().splitlines()[-3:]}
    except Exception as e:
        rec["selftest"] = {"error": repr(e)}
    try:
        n, batch, nl = 1 << 16, 24, 1 << 12
        x = np.random.default_rng(11).integers(0, P, size=n * batch, dtype=np.uint64)
        lv = np.random.default_rng(12).integers(0, P, size=5 * nl * batch, dtype=np.uint64)
        a, b = x.copy(), x.copy()
        tf.ntt(a, batch=batch)
        tf.ntt(b, batch=batch, devices="all")
        ra, rb = tf.MerkleTree.roots_batch(lv, nl), tf.MerkleTree.roots_batch(lv, nl, devices="all")
        rec["multi_call"] = {"devices": int(tf.lib().tf_device_count()), "units": batch,
                             "ntt_same_words_as_single_device": bool(np.array_equal(a, b)),
                             "merkle_roots_same_as_single_device": bool(np.array_equal(ra, rb)),
                             "what": "tf_ntt_bfe_multi (24 x 2^16) and tf_merkle_root_multi (24 trees of 2^12 leaves) over every visible device, host-resident batch"}
    except Exception as e:
        rec["multi_call"] = {"error": repr(e)}
    try:
        # ONE tree over every visible device (at least four workers: a one-GPU box lists device 0 four times) in one C call -- the
        # subtree split of MerkleTree::par_new at the C ABI (tf_merkle_build_multi with fewer trees than devices)
        nl = 1 << 20
        nd = int(tf.lib().tf_device_count())
        devs = [g % nd for g in range(max(4, nd))]
        lv = np.random.default_rng(13).integers(0, P, size=5 * nl, dtype=np.uint64)
        one = tf.MerkleTree.build_batch(lv, nl)
        many = tf.MerkleTree.build_batch(lv, nl, devices=devs)
        rec["single_tree_multi_call"] = {"devices": devs, "subtrees": int(tf.lib().tf_merkle_multi_subtrees(nl, 1, len(devs))),
                                         "every_node_same_as_single_device": bool(np.array_equal(one, many)),
                                         "what": "tf_merkle_build_multi, ONE tree of 2^20 leaves cut into subtrees (util_types/merkle_tree.rs:165-212, :247-275), host-resident"}
    except Exception as e:
        rec["single_tree_multi_call"] = {"error": repr(e)}
    return rec


def side_measurements(tf, torch, dev):
    """The callers on either side of the path, device-resident; reported, not part of any headline value."""
    extra = {}
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        off = tf.BFieldElement.new(7)

        def rnd(numel, seed):
            t = torch.empty(numel, dtype=torch.int64, device=dev)
            tf.device.fill_random(t, seed)
            return t

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
        pa, pb = rnd(256 * nh, 6), rnd(256 * nh, 7)
        po = torch.empty(256 * (2 * nh - 1), dtype=torch.int64, device=dev)
        ms = _timed(lambda: tf.device.poly_mul(pa, nh, pb, nh, po, batch=256))
        extra["fast_multiply_256x_2p19_by_2p19"] = {"ms": round(ms, 3), "output_coefficients_per_s": round(256 * (2 * nh - 1) / ms * 1e3, 1)}
        del pa, pb, po
        lv = rnd(128 * (1 << 18), 8)
        lo = torch.empty(128 * (1 << 21), dtype=torch.int64, device=dev)
        ms = _timed(lambda: tf.device.lde(lv, 1 << 18, tf.BFieldElement.new(1), lo, 1 << 21, off, batch=128))
        extra["lde_128x_2p18_to_2p21"] = {"ms": round(ms, 3), "g_points_per_s": round(128 * (1 << 21) / ms / 1e6, 3)}
        # the 128 codewords are now a column-major table of 2^21 rows: hash every row, build the tree
        tn = torch.empty(10 * (1 << 21), dtype=torch.int64, device=dev)
        ms = _timed(lambda: tf.device.merkle_from_columns(lo, 1 << 21, 128, tn))
        extra["merkle_from_columns_2p21_rows_x128"] = {"ms": round(ms, 3), "rows_per_s": round((1 << 21) / ms * 1e3, 1)}
        del lv, lo, tn
        # the zerofier tree: batch evaluation at the reference's complexity, interpolation, and both over a prepared tree
        for log_n in (16, 20):
            n = 1 << log_n
            dom, cf = rnd(n, 9), rnd(n, 10)
            vals, back = torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int64, device=dev)
            ms_e = _timed(lambda: tf.device.batch_evaluate(cf, n, dom, vals), reps=3)
            ms_i = _timed(lambda: tf.device.interpolate(dom, vals, back), reps=3)
            ok = bool(torch.equal(back, cf))
            with tf.device.ZerofierTree(dom) as tree:
                ms_te = _timed(lambda: tree.batch_evaluate(cf, n, vals), reps=3)
                ms_ti = _timed(lambda: tree.interpolate(vals, back), reps=3)
            extra[f"zerofier_tree_2p{log_n}_points"] = {"batch_evaluate_ms": round(ms_e, 3), "interpolate_ms": round(ms_i, 3),
                                                       "prepared_tree_evaluate_ms": round(ms_te, 3), "prepared_tree_interpolate_ms": round(ms_ti, 3),
                                                       "evaluate_then_interpolate_is_identity": ok}
            del dom, cf, vals, back
        # a table of codewords extrapolated to many points: every codeword's chunks walk the zerofier tree together
        cw, pts = rnd(64 << 16, 11), rnd(1 << 14, 12)
        ex = torch.empty(64 << 14, dtype=torch.int64, device=dev)
        ms = _timed(lambda: tf.device.coset_extrapolate(off, cw, 1 << 16, pts, ex, batch=64), reps=3)
        extra["coset_extrapolate_64x_2p16_to_2p14_points"] = {"ms": round(ms, 3)}
        del cw, pts, ex
        # the reference's own call shape: ONE slice per call (math/ntt.rs:67-82), device-resident, back-to-back calls
        lat = {}
        for log_n in (10, 12, 16, 18, 20):
            sx = rnd(1 << log_n, 13)
            us = _timed(lambda: tf.device.ntt_(sx, 1 << log_n), reps=100) * 1e3
            lat[f"2p{log_n}"] = round(us, 1)
        extra["single_slice_ntt_us"] = lat
        dom, cf = rnd(1 << 12, 14), rnd(1 << 12, 15)
        vals, back = torch.empty(1 << 12, dtype=torch.int64, device=dev), torch.empty(1 << 12, dtype=torch.int64, device=dev)
        with tf.device.ZerofierTree(dom) as tree:
            us_e = _timed(lambda: tree.batch_evaluate(cf, 1 << 12, vals), reps=50) * 1e3
            us_i = _timed(lambda: tree.interpolate(vals, back), reps=50) * 1e3
        us_o = _timed(lambda: tf.device.interpolate(dom, vals, back), reps=20) * 1e3  # the reference's call shape: tree built inside the call
        extra["zerofier_tree_2p12_points"] = {"prepared_tree_evaluate_us": round(us_e, 1), "prepared_tree_interpolate_us": round(us_i, 1),
                                              "one_shot_interpolate_us": round(us_o, 1),
                                              "evaluate_then_interpolate_is_identity": bool(torch.equal(back, cf))}
        del dom, cf, vals, back
        # PCIe-inclusive figure of the host-pointer entry point (pageable numpy buffers, 32 x 2^20 BFE = 256 MiB each way)
        import numpy as _np

        hb = 32
        hx = _np.random.default_rng(5).integers(0, 2 ** 63, size=hb * (1 << 20), dtype=_np.uint64)
        tf.ntt(hx, batch=hb)
        t0 = time.perf_counter()
        tf.ntt(hx, batch=hb)
        dt = time.perf_counter() - t0
        extra["host_pointer_ntt_32x2p20"] = {"ms": round(dt * 1e3, 2), "gfelts_per_s": round(hb * (1 << 20) / dt / 1e9, 3),
                                             "note": "tf_ntt_bfe on pageable host memory: H2D + 2 passes + D2H, synchronous"}
        # the same host-resident batch through the multi-device entry point (tf_ntt_bfe_multi): every visible GPU, and -- the
        # shape a one-GPU box can show -- two and four workers on device 0 (separate streams: copies of one slice beside compute
        # of another).  The words must be those of the single-device call.
        want = hx.copy()                       # hx has been transformed twice by the single-device entry point
        multi = {}
        ndev = int(tf.lib().tf_device_count())
        for label, devs in (("all_devices", "all"), ("two_workers_on_device_0", [0, 0]), ("four_workers_on_device_0", [0, 0, 0, 0])):
            hy = _np.random.default_rng(5).integers(0, 2 ** 63, size=hb * (1 << 20), dtype=_np.uint64)
            tf.ntt(hy, batch=hb, devices=devs)
            t0 = time.perf_counter()
            tf.ntt(hy, batch=hb, devices=devs)
            dt = time.perf_counter() - t0
            multi[label] = {"ms": round(dt * 1e3, 2), "gfelts_per_s": round(hb * (1 << 20) / dt / 1e9, 3), "same_words_as_single_device": bool(_np.array_equal(hy, want))}
        multi["visible_devices"] = ndev
        multi["note"] = "tf_ntt_bfe_multi, 32 x 2^20 BFE on pageable host memory, contiguous slices, one worker thread + stream per listed device"
        extra["host_pointer_multi"] = multi
    except Exception as e:  # side measurements never invalidate the headline line
        extra["error"] = repr(e)
    return extra


if __name__ == "__main__":
    main()
