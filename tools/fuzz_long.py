#!/usr/bin/env python3
"""One-off extended randomised parity run (the pytest fuzz tests are the short form): every entry point against the oracle."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import twenty_first_amd as tf
from oracle import tfo as oracle
P = 0xFFFFFFFF00000001
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rng = random.Random(seed)
KINDS = ["short", "ntt", "ntt", "coset", "coset", "interp", "mul", "merkle", "varlen", "eval", "extrap", "lde", "auth", "square", "mulb", "nttu", "interpu",
         "zerofier", "lagrange", "treeeval", "cdiv", "xoff", "trace", "handle", "bary"]
if len(sys.argv) > 3:
    KINDS = sys.argv[3].split(",")
import torch
def dev_words(a): return torch.from_numpy(a.view(np.int64)).cuda()
FILL = np.uint64(0xFFFFFFFFFFFFFFFF)
counts = {}
t_end = time.time() + budget
def bump(k): counts[k] = counts.get(k, 0) + 1
while time.time() < t_end:
    kind = rng.choice(KINDS)
    width = rng.choice([1, 3])
    if kind == "ntt":
        log_n = rng.randint(0, 23)
        n = 1 << log_n
        batch = rng.randint(1, max(1, min(67, (1 << 23) // (n * width))))
        inverse = rng.random() < 0.5
        x = oracle.fill_random(n * width * batch, rng.getrandbits(40))
        got = x.copy(); tf.ntt(got, width=width, batch=batch, _inverse=inverse)
        assert np.array_equal(got, oracle.ntt(x, width=width, inverse=inverse, batch=batch, threads=16)), (kind, log_n, width, batch, inverse)
    elif kind == "short":  # batches of short transforms: the wave-private tile kernel (n <= 32, BFieldElement <= 64), ragged ends
        n = 1 << rng.randint(1, 6)
        batch = rng.choice([rng.randint(1, 300), rng.randint(300, 70000), (2048 // n) * rng.randint(1, 40) + rng.randint(-1, 1)])
        batch = max(1, batch)
        inverse = rng.random() < 0.5
        x = oracle.fill_random(n * width * batch, rng.getrandbits(40))
        got = x.copy(); tf.ntt(got, width=width, batch=batch, _inverse=inverse)
        assert np.array_equal(got, oracle.ntt(x, width=width, inverse=inverse, batch=batch, threads=16)), (kind, n, width, batch, inverse)
    elif kind == "coset":
        log_order = rng.randint(0, 22)
        order = 1 << log_order
        n_coeffs = rng.choice([rng.randint(0, order), order >> rng.randint(0, 4), (order >> rng.randint(1, 3)) + rng.randint(0, 3)])
        n_coeffs = min(order, max(0, n_coeffs))
        batch = rng.randint(1, 3) if order * width <= (1 << 21) else 1
        off = oracle.bfe_new(rng.randrange(1, P))
        c = oracle.fill_random(n_coeffs * width * batch, rng.getrandbits(40))
        got = tf.fast_coset_evaluate(c, off, order, width=width, batch=batch)
        for b in range(batch):
            want = oracle.coset_evaluate(c[b * n_coeffs * width:(b + 1) * n_coeffs * width], off, order, width=width)
            assert np.array_equal(got[b * order * width:(b + 1) * order * width], want), (kind, order, n_coeffs, width, batch, b)
    elif kind == "interp":
        log_n = rng.randint(0, 21)
        n = 1 << log_n
        off = oracle.bfe_new(rng.randrange(1, P))
        v = oracle.fill_random(n * width, rng.getrandbits(40))
        assert np.array_equal(tf.fast_coset_interpolate(v, off, width=width), oracle.coset_interpolate(v, off, width=width)), (kind, log_n, width)
    elif kind == "mul":
        big = rng.random() < 0.25
        na, nb = (rng.randint(1, 70000), rng.randint(1, 70000)) if big else (rng.randint(1, 3000), rng.randint(1, 3000))
        a = oracle.fill_random(na * width, rng.getrandbits(40)); b = oracle.fill_random(nb * width, rng.getrandbits(40))
        got = tf.fast_multiply(a, b, width=width)
        assert np.array_equal(got, oracle.poly_mul(a, b, width=width)), (kind, na, nb, width)
    elif kind == "mulb":  # packed batch of products on device, unaligned output pointer
        log_hi = rng.choice([11, 13, 15, 16, 17, 18, 20])
        na, nb = rng.randint(1, 1 << (log_hi - 1)), rng.randint(1, 1 << (log_hi - 1))
        batch, shift = rng.randint(1, 5), rng.randint(0, 15)
        a = oracle.fill_random(na * width * batch, rng.getrandbits(40)); b = oracle.fill_random(nb * width * batch, rng.getrandbits(40))
        n_out = (na + nb - 1) * width
        out = torch.full((n_out * batch + shift + 40,), -1, dtype=torch.int64, device="cuda")
        tf.device.poly_mul(dev_words(a), na, dev_words(b), nb, out[shift:shift + n_out * batch], batch=batch, width=width)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint64)
        assert np.all(got[:shift] == FILL) and np.all(got[shift + n_out * batch:] == FILL), (kind, "guard", na, nb, width, batch, shift)
        for k in range(batch):
            want = oracle.poly_mul(a[k * na * width:(k + 1) * na * width], b[k * nb * width:(k + 1) * nb * width], width=width)
            assert np.array_equal(got[shift + k * n_out:shift + (k + 1) * n_out], want), (kind, na, nb, width, batch, shift, k)
    elif kind in ("nttu", "interpu"):  # in place / out of place on unaligned device slices
        log_n = rng.randint(5, 22)
        n = 1 << log_n
        batch = rng.randint(1, max(1, min(9, (1 << 22) // (n * width))))
        shift, inverse = rng.randint(1, 15), rng.random() < 0.5
        words = n * width * batch
        x = oracle.fill_random(words, rng.getrandbits(40))
        buf = torch.full((words + shift + 40,), -1, dtype=torch.int64, device="cuda")
        buf[shift:shift + words] = dev_words(x)
        if kind == "nttu":
            tf.device.ntt_(buf[shift:shift + words], n, batch=batch, width=width, inverse=inverse)
            want = oracle.ntt(x, width=width, inverse=inverse, batch=batch, threads=16)
            res = buf
        else:
            off = oracle.bfe_new(rng.randrange(1, P))
            res = torch.full((words + shift + 40,), -1, dtype=torch.int64, device="cuda")
            tf.device.coset_interpolate(buf[shift:shift + words], n, off, res[shift:shift + words], batch=batch, width=width)
            want = np.concatenate([oracle.coset_interpolate(x[k * n * width:(k + 1) * n * width], off, width=width) for k in range(batch)])
        torch.cuda.synchronize()
        got = res.cpu().numpy().view(np.uint64)
        assert np.all(got[:shift] == FILL) and np.all(got[shift + words:] == FILL), (kind, "guard", log_n, width, batch, shift)
        assert np.array_equal(got[shift:shift + words], want), (kind, log_n, width, batch, shift, inverse)
    elif kind == "merkle":
        h = rng.randint(0, 18)
        n = 1 << h
        batch = rng.randint(1, 3) if h < 15 else 1
        leaves = oracle.fill_random(5 * n * batch, rng.getrandbits(40))
        got = tf.MerkleTree.build_batch(leaves, n)
        for b in range(batch):
            assert np.array_equal(got[b].reshape(-1), oracle.merkle_build(leaves[5 * n * b:5 * n * (b + 1)], threads=16)), (kind, h, batch)
    elif kind == "varlen":
        row_len, n_rows = rng.randint(0, 120), rng.choice([1, 2, 17, 500, 40000])
        rows = oracle.fill_random(max(1, n_rows * row_len), rng.getrandbits(40))[: n_rows * row_len]
        if row_len:
            assert np.array_equal(tf.Tip5.hash_varlen_rows(rows, row_len), oracle.hash_varlen_rows(rows, row_len)), (kind, row_len, n_rows)
    elif kind == "eval":
        nc, m = rng.choice([0, 1, 5, 700, 1024, 5000]), rng.randint(1, 200)
        c = oracle.fill_random(max(1, nc) * width, rng.getrandbits(40))[: nc * width]
        pts = oracle.fill_random(m * width, rng.getrandbits(40))
        poly = tf.Polynomial(c, width=width)
        if poly.degree() >= 0:
            got = poly.batch_evaluate(pts).reshape(m, width)
            i = rng.randrange(m)
            want = oracle.poly_eval(poly.coefficients, int(pts[i]))[:1] if width == 1 else oracle.poly_eval_xfe_point(poly.coefficients, pts[3 * i:3 * i + 3])
            assert np.array_equal(got[i], want), (kind, nc, m, width)
    elif kind == "extrap":
        log_n, m, batch = rng.randint(0, 14), rng.randint(1, 20), rng.randint(1, 3)
        n = 1 << log_n
        off = oracle.bfe_new(rng.randrange(1, P))
        cw = oracle.fill_random(batch * n * width, rng.getrandbits(40)); pts = oracle.fill_random(m * width, rng.getrandbits(40))
        got = tf.Polynomial.batch_coset_extrapolate(off, n, cw, pts, width=width).reshape(batch, m, width)
        b, i = rng.randrange(batch), rng.randrange(m)
        co = oracle.coset_interpolate(cw[b * n * width:(b + 1) * n * width], off, width=width)
        want = oracle.poly_eval(co, int(pts[i]))[:1] if width == 1 else oracle.poly_eval_xfe_point(co, pts[3 * i:3 * i + 3])
        assert np.array_equal(got[b, i], want), (kind, log_n, m, batch, width)
    elif kind == "lde":
        import torch
        log_n, blow = rng.randint(0, 19), rng.randint(0, 3)
        n, m = 1 << log_n, 1 << (log_n + blow)
        batch = rng.randint(1, 3) if m * width <= (1 << 20) else 1
        o_in, o_out = oracle.bfe_new(rng.randrange(1, P)), oracle.bfe_new(rng.randrange(1, P))
        v = oracle.fill_random(batch * n * width, rng.getrandbits(40))
        dv = torch.from_numpy(v.view(np.int64)).cuda()
        ext = torch.empty(batch * m * width, dtype=torch.int64, device="cuda")
        tf.device.lde(dv, n, o_in, ext, m, o_out, batch=batch, width=width)
        torch.cuda.synchronize()
        got = ext.cpu().numpy().view(np.uint64)
        for b in range(batch):
            co = oracle.coset_interpolate(v[b * n * width:(b + 1) * n * width], o_in, width=width)
            assert np.array_equal(got[b * m * width:(b + 1) * m * width], oracle.coset_evaluate(co, o_out, m, width=width)), (kind, log_n, blow, width, batch)
    elif kind == "auth":
        import torch
        h = rng.randint(0, 14)
        n = 1 << h
        leaves = oracle.fill_random(5 * n, rng.getrandbits(40))
        dn = torch.empty(10 * n, dtype=torch.int64, device="cuda")
        tf.device.merkle_build(torch.from_numpy(leaves.view(np.int64)).cuda(), n, dn)
        torch.cuda.synchronize()
        idx = [rng.randrange(n) for _ in range(rng.randint(0, 40))]
        want_idx = oracle.auth_structure_indices(n, idx)
        nodes = oracle.merkle_build(leaves).reshape(2 * n, 5)
        got = tf.device.authentication_structure(dn, n, idx)
        assert np.array_equal(got, nodes[np.asarray(want_idx, dtype=np.int64)].reshape(-1, 5)), (kind, h, len(idx))
    elif kind == "square":
        na = rng.randint(1, 60000) if rng.random() < 0.25 else rng.randint(1, 4000)
        a = oracle.fill_random(na * width, rng.getrandbits(40))
        assert np.array_equal(tf.fast_square(a, width=width), oracle.poly_mul(a, a, width=width)), (kind, na, width)
    elif kind == "zerofier":
        n = rng.choice([rng.randint(0, 40), rng.randint(100, 700), rng.randint(900, 3000)])
        r = oracle.fill_random(max(1, n) * width, rng.getrandbits(40))[: n * width]
        if n > 2 and rng.random() < 0.3:
            r[width: 2 * width] = r[:width]
        assert np.array_equal(tf.Polynomial.zerofier(r, width=width).coefficients, tf.Polynomial(oracle.zerofier(r, width), width=width).coefficients), (kind, n, width)
    elif kind == "lagrange":
        n = rng.choice([rng.randint(1, 40), rng.randint(100, 700), rng.randint(900, 2200)])
        rows = rng.randint(1, 3)
        d = oracle.fill_random(n * width, rng.getrandbits(40))
        vals = [oracle.fill_random(n * width, rng.getrandbits(40)) for _ in range(rows)]
        polys = tf.Polynomial.batch_fast_interpolate(d, vals, width=width)
        k = rng.randrange(rows)
        assert np.array_equal(polys[k].coefficients, tf.Polynomial(oracle.lagrange_interpolate(d, vals[k], width), width=width).coefficients), (kind, n, width, rows)
    elif kind == "treeeval":  # the zerofier-tree route against Horner and the oracle
        n, m = rng.randint(2, 20000), rng.randint(2 * (256 if width == 1 else 128), 9000)
        c = oracle.fill_random(n * width, rng.getrandbits(40)); pts = oracle.fill_random(m * width, rng.getrandbits(40))
        try:
            tf.lib().tf_set_batch_eval_route(2)
            got = tf.Polynomial(c, width=width).batch_evaluate(pts).reshape(m, width)
        finally:
            tf.lib().tf_set_batch_eval_route(0)
        for i in (0, rng.randrange(m), m - 1):
            want = oracle.poly_eval(c, int(pts[i]))[:1] if width == 1 else oracle.poly_eval_xfe_point(c, pts[3 * i:3 * i + 3])
            assert np.array_equal(got[i], want), (kind, n, m, width, i)
    elif kind == "cdiv":
        nq, nb = rng.choice([(rng.randint(1, 50), rng.randint(1, 50)), (rng.randint(1, 3000), rng.randint(1, 3000)), (rng.randint(1, 40000), rng.randint(500, 40000))])
        q, b = oracle.fill_random(nq, rng.getrandbits(40)), oracle.fill_random(nb, rng.getrandbits(40))
        if rng.random() < 0.3:
            b[0] = 0
        if not q[-1] or not b[-1]:
            continue
        a = oracle.poly_mul(q, b)
        assert np.array_equal(tf.Polynomial(a).clean_divide(tf.Polynomial(b)).coefficients, q), (kind, nq, nb)
        if nq * nb < (1 << 21):
            assert np.array_equal(oracle.clean_divide(a, b, 0), q), (kind, "oracle", nq, nb)
    elif kind == "xoff":
        log_order = rng.randint(0, 14)
        order = 1 << log_order
        n_coeffs = rng.randint(0, order)
        off = oracle.fill_random(3, rng.getrandbits(40))
        c = oracle.fill_random(max(1, n_coeffs) * 3, rng.getrandbits(40))[: 3 * n_coeffs]
        got = tf.fast_coset_evaluate(c, off, order, width=3)
        assert np.array_equal(got, oracle.coset_evaluate_xfe_offset(c, off, order)), (kind, order, n_coeffs)
        assert np.array_equal(tf.fast_coset_interpolate(got, off, width=3), oracle.coset_interpolate_xfe_offset(got, off)), (kind, "interp", order)
    elif kind == "trace":
        count = rng.choice([1, 5, 300, 5000])
        s0 = oracle.fill_random(16 * count, rng.getrandbits(40))
        s = s0.copy()
        tr = tf.Tip5.trace_states(s)
        i = rng.randrange(count)
        want, end = oracle.tip5_trace(s0[16 * i: 16 * i + 16])
        assert np.array_equal(tr[i], want) and np.array_equal(s[16 * i: 16 * i + 16], end), (kind, count, i)
    elif kind == "bary":
        log_n, batch = rng.randint(0, 12), rng.randint(1, 5)
        n = 1 << log_n
        cw = oracle.fill_random(n * width * batch, rng.getrandbits(40))
        x = oracle.fill_random(3, rng.getrandbits(40)) if rng.random() < 0.7 else np.array([oracle.bfe_new(rng.randrange(2, P))], dtype=np.uint64)
        if x.size == 1 and oracle.bfe_mod_pow(int(x[0]), n) == oracle.bfe_new(1):
            continue
        got = tf.barycentric_evaluate(cw, x, width=width, batch=batch).reshape(batch, -1)
        b = rng.randrange(batch)
        want = oracle.barycentric_evaluate(cw[b * n * width:(b + 1) * n * width], x, width)
        assert np.array_equal(got[b], want[: got.shape[1]]), (kind, log_n, width, batch, x.size)
    elif kind == "handle":
        n = rng.choice([rng.randint(1, 300), rng.randint(300, 4000)])
        d = oracle.fill_random(n * width, rng.getrandbits(40))
        with tf.ZerofierTree(d, width=width) as tree:
            f = tf.Polynomial(oracle.fill_random(rng.randint(1, 3 * n) * width, rng.getrandbits(40)), width=width)
            vals = tree.batch_evaluate(f)
            i = rng.randrange(n)
            want = oracle.poly_eval(f.coefficients, int(d[i]))[:1] if width == 1 else oracle.poly_eval_xfe_point(f.coefficients, d[3 * i:3 * i + 3])
            assert np.array_equal(vals[i * width:(i + 1) * width], want), (kind, n, width)
            g = tf.Polynomial(oracle.fill_random(n * width, rng.getrandbits(40)), width=width)
            back = tree.interpolate([tree.batch_evaluate(g)])[0]
            assert np.array_equal(back.coefficients, g.coefficients), (kind, "round trip", n, width)
    bump(kind)
print(f"seed {seed}: all matched the oracle: " + ", ".join(f"{k} {v}" for k, v in sorted(counts.items())), flush=True)
