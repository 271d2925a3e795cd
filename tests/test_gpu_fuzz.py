"""Randomised shapes through every entry point of the hot path (seeded, so reproducible): sizes, batches,
widths, directions, offsets and coefficient counts drawn at random and compared bit-for-bit with the oracle."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def test_fuzz_ntt_shapes(tf, oracle):
    rng = random.Random(20260928)
    for it in range(60):
        log_n = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 17, 19, 20, 21])
        n = 1 << log_n
        width = rng.choice([1, 3])
        max_batch = max(1, min(40, (1 << 21) // (n * width)))
        batch = rng.randint(1, max_batch)
        inverse = rng.random() < 0.5
        x = oracle.fill_random(n * width * batch, rng.getrandbits(40))
        want = oracle.ntt(x, width=width, inverse=inverse, batch=batch, threads=8)
        got = x.copy()
        tf.ntt(got, width=width, batch=batch, _inverse=inverse)
        assert np.array_equal(got, want), (it, log_n, width, batch, inverse)


def test_fuzz_coset_shapes(tf, oracle):
    rng = random.Random(7)
    for it in range(40):
        log_order = rng.choice([0, 1, 3, 5, 8, 10, 11, 12, 15, 18, 20])
        order = 1 << log_order
        n_coeffs = rng.randint(0, order)
        width = rng.choice([1, 3])
        off = oracle.bfe_new(rng.randrange(1, P))
        c = oracle.fill_random(n_coeffs * width, rng.getrandbits(40))
        got = tf.fast_coset_evaluate(c, off, order, width=width)
        want = oracle.coset_evaluate(c, off, order, width=width)
        assert np.array_equal(got, want), (it, order, n_coeffs, width)
        if n_coeffs == order and order > 0:
            back = tf.fast_coset_interpolate(got, off, width=width)
            assert np.array_equal(back, c)


def test_fuzz_merkle_and_hash_shapes(tf, oracle):
    rng = random.Random(99)
    for it in range(30):
        h = rng.randint(0, 13)
        n = 1 << h
        batch = rng.randint(1, 4)
        leaves = oracle.fill_random(5 * n * batch, rng.getrandbits(40))
        got = tf.MerkleTree.build_batch(leaves, n)
        for b in range(batch):
            assert np.array_equal(got[b].reshape(-1), oracle.merkle_build(leaves[5 * n * b:5 * n * (b + 1)]))
        row_len = rng.randint(0, 45)
        n_rows = rng.randint(1, 600)
        rows = oracle.fill_random(max(1, n_rows * row_len), rng.getrandbits(40))[: n_rows * row_len]
        if row_len:
            assert np.array_equal(tf.Tip5.hash_varlen_rows(rows, row_len), oracle.hash_varlen_rows(rows, row_len))


def test_reentrant_from_many_host_threads(tf, oracle):
    """The ABI is called concurrently from host threads (the reference's ntt is called from rayon workers,
    math/ntt.rs:250-274): 8 threads, mixed entry points and sizes, every result checked."""
    import threading

    errors = []

    def worker(k):
        try:
            rng = random.Random(1000 + k)
            for it in range(6):
                log_n = rng.choice([5, 9, 10, 12, 14, 16])
                n = 1 << log_n
                x = oracle.fill_random(n * 2, rng.getrandbits(40))
                got = x.copy()
                tf.ntt(got, batch=2, _inverse=bool(it & 1))
                if not np.array_equal(got, oracle.ntt(x, batch=2, inverse=bool(it & 1))):
                    errors.append(("ntt", k, it))
                leaves = oracle.fill_random(5 * 512, rng.getrandbits(40))
                if not np.array_equal(tf.MerkleTree.par_new(leaves).nodes.reshape(-1), oracle.merkle_build(leaves)):
                    errors.append(("merkle", k, it))
                c = oracle.fill_random(300, rng.getrandbits(40))
                off = oracle.bfe_new(rng.randrange(1, P))  # > 16 distinct offsets overall: exercises the uncached path
                if not np.array_equal(tf.fast_coset_evaluate(c, off, 512), oracle.coset_evaluate(c, off, 512)):
                    errors.append(("coset", k, it))
                # the callers of the second half of round 2: interpolation through the tree, clean division, barycentric evaluation
                m = rng.choice([40, 300, 700])
                d, v = oracle.fill_random(m, rng.getrandbits(40)), oracle.fill_random(m, rng.getrandbits(40))
                f = tf.Polynomial.interpolate(d, v)
                if not np.array_equal(f.coefficients, tf.Polynomial(oracle.lagrange_interpolate(d, v)).coefficients):
                    errors.append(("interpolate", k, it))
                q, b = oracle.fill_random(rng.randint(1, 900), rng.getrandbits(40)), oracle.fill_random(rng.randint(1, 900), rng.getrandbits(40))
                if q[-1] and b[-1] and not np.array_equal(tf.Polynomial(oracle.poly_mul(q, b)).clean_divide(tf.Polynomial(b)).coefficients, q):
                    errors.append(("clean_divide", k, it))
                cw, xp = oracle.fill_random(256, rng.getrandbits(40)), oracle.fill_random(3, rng.getrandbits(40))
                if not np.array_equal(tf.barycentric_evaluate(cw, xp), oracle.barycentric_evaluate(cw, xp)):
                    errors.append(("barycentric", k, it))
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
