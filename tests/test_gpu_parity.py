"""Parity of the HIP path (through the C ABI of libtf_hip.so) against the KAT-pinned CPU oracle.

Bit-exact comparison of raw Montgomery words -- all arithmetic on this path is integer.
Run with `pytest -m gpu` on an MI355X box.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
P = 0xFFFFFFFF00000001
MAX = P - 1


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(tf):
    assert tf.lib().tf_device_count() > 0, "no HIP device visible: the product has no CPU fallback"


# ------------------------------------------------------------------ reference KATs through the GPU path

def test_kat_ntt_basic(tf, oracle):
    """math/ntt.rs:424-445, :448-469, :512-560 on the GPU"""
    x = oracle.to_raw([1, 4, 0, 0])
    y = x.copy()
    tf.ntt(y)
    assert list(y) == list(oracle.to_raw([5, 1125899906842625, 18446744069414584318, 18445618169507741698]))
    tf.intt(y)
    assert list(y) == list(x)
    m = oracle.to_raw([MAX, 0, 0, 0])
    tf.ntt(m)
    assert list(m) == list(oracle.to_raw([MAX] * 4))
    x32 = oracle.to_raw([1, 4, 0, 0, 0, 0, 0, 0] * 4)
    expected = [0] * 32
    for i, v in enumerate([20, 18446744069146148869, 4503599627370500, 18446726477228544005,
                           18446744069414584309, 268435460, 18442240469787213829, 17592186040324]):
        expected[4 * i] = v
    y = x32.copy()
    tf.ntt(y)
    assert list(y) == list(oracle.to_raw(expected))
    tf.intt(y)
    assert list(y) == list(x32)


def test_kat_xfield_basic_ntt(tf, oracle):
    """math/ntt.rs:398-421"""
    x = oracle.to_raw([1, 0, 0] + [0, 0, 0] * 3)
    y = x.copy()
    tf.ntt(y, width=3)
    assert list(y) == list(oracle.to_raw([1, 0, 0] * 4))
    tf.intt(y, width=3)
    assert list(y) == list(x)


def test_kat_tip5_snapshots(tf, oracle):
    """tip5/mod.rs:1294-1306, :1309-1325, :1328-1362, :1146-1206, mmr_accumulator.rs:1038-1046 on the GPU"""
    from tests.test_oracle_kat import DEGENERATE_IN, DEGENERATE_OUT, SNAPSHOT_OUT, SNAPSHOT_STATE

    out = tf.Tip5.permutation(np.array(SNAPSHOT_STATE, dtype=np.uint64))
    assert [int(v) for v in out[:5]] == SNAPSHOT_OUT
    out = tf.Tip5.permutation(oracle.to_raw(DEGENERATE_IN))
    assert list(out) == list(oracle.to_raw(DEGENERATE_OUT))
    pre = np.zeros(10, dtype=np.uint64)
    for i in range(6):
        pre[i:i + 5] = tf.Tip5.hash_10(pre)
    assert tf.Digest.to_hex(tf.Tip5.hash_10(pre)) == (
        "109cc2fe453bd9962f754b96d8f5b919b60af030940a275f5540da195fef65ee651c1b6fa19b2c6a")
    acc = [0] * 5
    for i in range(20):
        d = tf.Tip5.hash_varlen(oracle.to_raw(list(range(i))) if i else np.zeros(0, np.uint64))
        acc = [oracle.bfe_add(a, int(b)) for a, b in zip(acc, d)]
    assert tf.Digest.to_hex(acc) == "efbafa86622a9c69652f8a1c4ffd734f021ad23a0a8085412a877de0f9170b18ea4ff69b6fff9a03"
    assert tf.Digest.to_hex(tf.Tip5.hash_10(np.zeros(10, dtype=np.uint64))) == (
        "cd65052100640f0d27e5654f97c47e49899add2f265967ccbefee7264e9bc08f588542d9dc3d5ac5")


def test_merkle_root_goldens(tf, oracle):
    gold = json.load(open(os.path.join(HERE, "golden", "merkle_roots.json")))
    for h, hexroot in gold["test_tree_of_height_roots"].items():
        leaves = np.concatenate([oracle.hash_varlen(oracle.to_raw([i])) for i in range(1 << int(h))])
        tree = tf.MerkleTree.par_new(leaves)
        assert tf.Digest.to_hex(tree.root()) == hexroot
        assert tf.Digest.to_hex(tf.MerkleTree.par_frugal_root(leaves)) == hexroot


# ------------------------------------------------------------------ NTT vs oracle

@pytest.mark.parametrize("log_n", list(range(0, 15)) + [16, 18, 20, 21, 22, 24, 25, 26])
@pytest.mark.parametrize("inverse", [False, True])
def test_ntt_bfe_matches_oracle(tf, oracle, log_n, inverse):
    n = 1 << log_n
    batch = 5 if log_n <= 12 else (3 if log_n <= 18 else 1)
    x = oracle.fill_random(n * batch, 1000 + log_n)
    want = oracle.ntt(x, inverse=inverse, batch=batch, threads=8)
    got = x.copy()
    tf.ntt(got, batch=batch, _inverse=inverse)
    assert np.array_equal(got, want)
    assert (got < np.uint64(P)).all()


@pytest.mark.parametrize("log_n", [0, 1, 2, 4, 5, 6, 9, 10, 11, 13, 16, 20, 21, 23])
@pytest.mark.parametrize("inverse", [False, True])
def test_ntt_xfe_matches_oracle(tf, oracle, log_n, inverse):
    n = 1 << log_n
    batch = 3 if log_n <= 13 else 1
    x = oracle.fill_random(3 * n * batch, 2000 + log_n)
    want = oracle.ntt(x, width=3, inverse=inverse, batch=batch, threads=8)
    got = x.copy()
    tf.ntt(got, width=3, batch=batch, _inverse=inverse)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64])
@pytest.mark.parametrize("width", [1, 3])
def test_short_transforms_in_wave_private_tiles(tf, oracle, n, width):
    """Batches of transforms of at most 32 points (ntt_rows32w_kernel: a wave moves 64 rows of 32 elements -- 21 element rows of an
    XFieldElement batch -- through its own LDS; 64-point BFieldElement transforms on lane pairs with one DIF stage across the pair):
    batches that end inside a row, inside a tile, one past a tile, both directions; the handful-of-transforms calls below 4096 words
    stay on ntt_tiny_kernel (64 points: the latency-shaped kernel; XFieldElement: the row pass) and are compared all the same."""
    for batch in (1, 3, 21, 22, 63, 64, 65, 127, 2048 // n, 2048 // n + 1, 4096 // n + 3, 1000, (1 << 16) + 5):
        for inverse in (False, True):
            x = oracle.fill_random(n * batch * width, 7000 + 13 * n + batch)
            want = oracle.ntt(x, width=width, inverse=inverse, batch=batch, threads=4)
            got = x.copy()
            tf.ntt(got, width=width, batch=batch, _inverse=inverse)
            assert np.array_equal(got, want), (n, width, batch, inverse)


@pytest.mark.parametrize("n,batch", [(32, 1), (32, 257), (64, 100), (1024, 1), (1024, 33), (2048, 9), (4096, 17), (1 << 15, 7)])
def test_ntt_ragged_batches(tf, oracle, n, batch):
    """tile edges: batch counts that do not divide the per-workgroup tile"""
    x = oracle.fill_random(n * batch, 31 * n + batch)
    want = oracle.ntt(x, batch=batch, threads=8)
    got = x.copy()
    tf.ntt(got, batch=batch)
    assert np.array_equal(got, want)
    tf.intt(got, batch=batch)
    assert np.array_equal(got, x)


def test_ntt_edge_values(tf, oracle):
    """all-zero, all-(p-1), single spike: extreme operands of the shift/reduce paths"""
    for n in [64, 1024, 4096, 1 << 16]:
        for fill in ("zero", "max", "spike", "alt"):
            if fill == "zero":
                x = np.zeros(n, dtype=np.uint64)
            elif fill == "max":
                x = np.full(n, oracle.bfe_new(MAX), dtype=np.uint64)
            elif fill == "spike":
                x = np.zeros(n, dtype=np.uint64)
                x[n - 1] = np.uint64(P - 1)  # raw p-1
            else:
                x = np.array([P - 1, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1] * (n // 4), dtype=np.uint64)
            want = oracle.ntt(x)
            got = x.copy()
            tf.ntt(got)
            assert np.array_equal(got, want), (n, fill)
            want_i = oracle.intt(x)
            got = x.copy()
            tf.intt(got)
            assert np.array_equal(got, want_i), (n, fill)


def test_ntt_panics(tf):
    """math/ntt.rs:135-140: not a power of two -> panic; empty and length-1 slices are fine"""
    for n in [3, 5, 6, 12, 1000, 3 << 10]:
        with pytest.raises(tf.NttPanic) as e:
            tf.ntt(np.zeros(n, dtype=np.uint64))
        assert e.value.code == 4
    tf.ntt(np.zeros(0, dtype=np.uint64))
    one = np.array([12345], dtype=np.uint64)
    tf.ntt(one)
    tf.intt(one)
    assert int(one[0]) == 12345


# ------------------------------------------------------------------ coset evaluation

@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("n_coeffs,order", [(1, 1), (1, 2), (3, 4), (11, 16), (16, 16), (20, 32), (700, 1024), (1024, 1024),
                                            (1500, 2048), (5000, 8192), (1 << 16, 1 << 16), (300000, 1 << 20)])
def test_coset_evaluate_matches_oracle(tf, oracle, width, n_coeffs, order):
    batch = 2 if order <= 8192 else 1
    c = oracle.fill_random(n_coeffs * width * batch, 77 + n_coeffs)
    offset = oracle.bfe_new(7)  # benches/polynomial_coset.rs:20
    got = tf.fast_coset_evaluate(c, offset, order, width=width, batch=batch).reshape(batch, -1)
    for b in range(batch):
        want = oracle.coset_evaluate(c[b * n_coeffs * width:(b + 1) * n_coeffs * width], offset, order, width=width)
        assert np.array_equal(got[b], want)


def test_coset_evaluate_other_offsets_and_polynomial_api(tf, oracle):
    c = oracle.fill_random(100, 5)
    for off_val in [1, 2, 7, MAX, 1234567890123456789]:
        off = oracle.bfe_new(off_val)
        assert np.array_equal(tf.fast_coset_evaluate(c, off, 128), oracle.coset_evaluate(c, off, 128))
    poly = tf.Polynomial(np.concatenate([c, np.zeros(60, np.uint64)]))  # leading zeros are not degree
    assert poly.degree() == 99
    assert np.array_equal(poly.fast_coset_evaluate(oracle.bfe_new(7), 128), oracle.coset_evaluate(c, oracle.bfe_new(7), 128))
    with pytest.raises(tf.NttPanic):  # polynomial.rs:1388-1392
        poly.fast_coset_evaluate(oracle.bfe_new(7), 64)
    with pytest.raises(tf.NttPanic):
        tf.fast_coset_evaluate(c, oracle.bfe_new(7), 96)
    zero = tf.fast_coset_evaluate(np.zeros(0, np.uint64), oracle.bfe_new(7), 16)
    assert not zero.any() and zero.size == 16


# ------------------------------------------------------------------ Tip5

def test_tip5_random_batches(tf, oracle):
    x = oracle.fill_random(10 * 20011, 42)
    assert np.array_equal(tf.Tip5.hash_pairs(x), oracle.hash_pairs(x))
    st = oracle.fill_random(16 * 1000, 43)
    got = st.copy()
    tf.Tip5.permute_states(got)
    want = np.concatenate([oracle.tip5_permutation(st[16 * i:16 * i + 16]) for i in range(1000)])
    assert np.array_equal(got, want)
    l, r = x[:5], x[5:10]
    assert np.array_equal(tf.Tip5.hash_pair(l, r), oracle.hash_pair(l, r))


def test_tip5_degenerate_words_after_lookup(tf, oracle):
    """states whose first four words look up to >= p patterns (tip5/mod.rs:222-242)"""
    rows = []
    for b in [0x00, 0xFE, 0xFF, 0x01, 0x80]:
        w = int.from_bytes(bytes([b] * 8), "little") % P
        rows.append([w] * 16)
    st = np.array(rows, dtype=np.uint64).reshape(-1)
    got = st.copy()
    tf.Tip5.permute_states(got)
    want = np.concatenate([oracle.tip5_permutation(st[16 * i:16 * i + 16]) for i in range(len(rows))])
    assert np.array_equal(got, want)
    assert (got < np.uint64(P)).all()


@pytest.mark.parametrize("row_len", [0, 1, 5, 9, 10, 11, 19, 20, 21, 37, 100])
def test_hash_varlen_rows(tf, oracle, row_len):
    n_rows = 301
    rows = oracle.fill_random(n_rows * row_len, 900 + row_len)
    got = tf.Tip5.hash_varlen_rows(rows, row_len) if row_len else np.concatenate(
        [tf.Tip5.hash_varlen(np.zeros(0, np.uint64)) for _ in range(3)])
    if row_len:
        assert np.array_equal(got, oracle.hash_varlen_rows(rows, row_len))
    else:
        assert np.array_equal(got, np.concatenate([oracle.hash_varlen(np.zeros(0, np.uint64))] * 3))


# ------------------------------------------------------------------ Merkle

@pytest.mark.parametrize("height", list(range(0, 17)))
def test_merkle_build_matches_oracle(tf, oracle, height):
    n = 1 << height
    leaves = oracle.fill_random(5 * n, 300 + height)
    tree = tf.MerkleTree.par_new(leaves)
    want = oracle.merkle_build(leaves, threads=8).reshape(2 * n, 5)
    assert np.array_equal(tree.nodes, want)
    assert not tree.nodes[0].any()
    assert tree.num_leafs() == n and tree.height() == height
    assert np.array_equal(tf.MerkleTree.par_frugal_root(leaves), want[1])
    assert np.array_equal(tf.MerkleTree.sequential_frugal_root(leaves), want[1])


def test_merkle_batch_of_trees(tf, oracle):
    # (the levels near the root run as subtrees, merkle_subtree_kernel: one, two and three launches, with and without a node array)
    # (a subtree workgroup hashes a level on row PAIRS where it has the rows to spare -- only while every workgroup has a compute unit of its
    #  own: batches of more than 256 small trees keep the 16-lane rows)
    for n, batch in [(1, 3), (8, 5), (64, 9), (128, 5), (256, 3), (512, 3), (2048, 4), (4096, 7), (16384, 2), (65536, 3), (64, 300), (32, 257), (1024, 70)]:
        leaves = oracle.fill_random(5 * n * batch, 17 * n + batch)
        got = tf.MerkleTree.build_batch(leaves, n)
        roots = tf.MerkleTree.roots_batch(leaves, n)
        for b in range(batch):
            want = oracle.merkle_build(leaves[5 * n * b:5 * n * (b + 1)]).reshape(2 * n, 5)
            assert np.array_equal(got[b], want)
            assert np.array_equal(roots[b], want[1])


def test_merkle_errors(tf, oracle):
    """util_types/merkle_tree.rs:393-410, :933-965, :299-309, :332-335"""
    with pytest.raises(tf.MerkleTreeError) as e:
        tf.MerkleTree.par_new(np.zeros(0, np.uint64))
    assert e.value.variant == "TooFewLeafs"
    with pytest.raises(tf.MerkleTreeError) as e:
        tf.MerkleTree.sequential_frugal_root(np.zeros(0, np.uint64))
    assert e.value.variant == "TooFewLeafs"
    with pytest.raises(tf.MerkleTreeError) as e:
        tf.MerkleTree.par_frugal_root(np.zeros(0, np.uint64))
    assert e.value.variant == "IncorrectNumberOfLeafs"
    for n in [3, 5, 6, 7, 12, 1000]:
        leaves = oracle.fill_random(5 * n, 1)
        for fn in (tf.MerkleTree.par_new, tf.MerkleTree.par_frugal_root, tf.MerkleTree.sequential_frugal_root):
            with pytest.raises(tf.MerkleTreeError) as e:
                fn(leaves)
            assert e.value.variant == "IncorrectNumberOfLeafs"


# ------------------------------------------------------------------ device-pointer API, BASELINE sizes

def _to_dev(a):
    import torch

    return torch.from_numpy(a.view(np.int64)).cuda()


def _to_host(t):
    return t.cpu().numpy().view(np.uint64)


def test_device_api_small(tf, oracle):
    import torch

    n, batch = 1 << 12, 6
    x = oracle.fill_random(n * batch, 5)
    d = _to_dev(x)
    tf.device.ntt_(d, n, batch=batch)
    torch.cuda.synchronize()
    assert np.array_equal(_to_host(d), oracle.ntt(x, batch=batch, threads=8))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tf.device.ntt_(d, n, batch=batch, inverse=True)
    s.synchronize()
    assert np.array_equal(_to_host(d), x)
    leaves = oracle.fill_random(5 * 2048, 6)
    dl = _to_dev(leaves)
    nodes = torch.empty(10 * 2048, dtype=torch.int64, device="cuda")
    root = torch.empty(5, dtype=torch.int64, device="cuda")
    tf.device.merkle_build(dl, 2048, nodes)
    tf.device.merkle_root(dl, 2048, root)
    torch.cuda.synchronize()
    want = oracle.merkle_build(leaves)
    assert np.array_equal(_to_host(nodes), want)
    assert np.array_equal(_to_host(root), want[5:10])


def test_config2_256x2pow20_roundtrip_and_samples(tf, oracle):
    """BASELINE config 2: 256 x 2^20 BFE on one GPU.  Full-size properties: intt(ntt(x)) == x for all
    2^28 elements; sampled transforms are compared word for word with the oracle."""
    import torch

    n, batch = 1 << 20, 256
    chunk = 16
    d = torch.empty(n * batch, dtype=torch.int64, device="cuda")
    host_chunks = {}
    for c0 in range(0, batch, chunk):
        h = oracle.fill_random(n * chunk, 0x7F210002 + c0)
        if c0 in (0, 128, 240):
            host_chunks[c0] = h
        d[c0 * n:(c0 + chunk) * n] = _to_dev(h)
    orig = d.clone()
    tf.device.ntt_(d, n, batch=batch)
    torch.cuda.synchronize()
    # 32 of the 256 transforms word for word (chunks 0 and 240 entirely: the oracle runs one transform per thread)
    for c0, h in host_chunks.items():
        picks = range(chunk) if c0 in (0, 240) else (0, chunk - 1)
        want = oracle.ntt(h, batch=chunk, threads=chunk)
        for b in picks:
            got = _to_host(d[(c0 + b) * n:(c0 + b + 1) * n])
            assert np.array_equal(got, want[b * n:(b + 1) * n]), (c0, b)
    tf.device.ntt_(d, n, batch=batch, inverse=True)
    torch.cuda.synchronize()
    assert torch.equal(d, orig)


def test_config3_merkle_2pow24(tf, oracle):
    """BASELINE config 3: 2^24-leaf tree on one GPU; root and every node match the CPU oracle
    (oracle runs the par_new restatement on the host cores)."""
    import torch

    n = 1 << 24
    leaves = oracle.fill_random(5 * n, 0x7F210003)
    dl = _to_dev(leaves)
    nodes = torch.empty(10 * n, dtype=torch.int64, device="cuda")
    tf.device.merkle_build(dl, n, nodes)
    torch.cuda.synchronize()
    got = _to_host(nodes)
    threads = min(64, os.cpu_count() or 8)
    want = oracle.merkle_build(leaves, threads=threads)
    assert np.array_equal(got[5:10], want[5:10]), "root mismatch"
    assert np.array_equal(got, want)


def test_config4_xfe_coset_eval_2pow22(tf, oracle):
    """BASELINE config 4 shape (XFE, 2^22 coefficients, order 2^22, offset 7) on a batch of 2;
    compared word for word with the oracle."""
    import torch

    n, batch = 1 << 22, 2
    c = oracle.fill_random(3 * n * batch, 0x7F210004)
    dc = _to_dev(c)
    out = torch.empty(3 * n * batch, dtype=torch.int64, device="cuda")
    off = oracle.bfe_new(7)
    tf.device.coset_evaluate(dc, n, off, out, n, batch=batch, width=3)
    torch.cuda.synchronize()
    got = _to_host(out)
    for b in range(batch):
        want = oracle.coset_evaluate(c[3 * n * b:3 * n * (b + 1)], off, n, width=3)
        assert np.array_equal(got[3 * n * b:3 * n * (b + 1)], want)


@pytest.mark.parametrize("passes,log_n,width,batch", [(3, 15, 1, 3), (3, 18, 3, 2), (4, 20, 1, 2), (4, 21, 3, 1), (4, 23, 1, 1)])
def test_deeper_pass_plans_match_oracle(tf, oracle, passes, log_n, width, batch):
    """the three- and four-pass plans (n > 2^20, n = 2^31) forced at sizes the oracle finishes quickly; forward,
    inverse, coset evaluation and interpolation all run through the same planner"""
    n = 1 << log_n
    x = oracle.fill_random(batch * n * width, 600 + log_n)
    lib = tf._lib.lib()
    lib.tf_set_ntt_min_passes(passes)
    try:
        y = x.copy()
        tf.ntt(y, width=width, batch=batch)
        assert np.array_equal(y, oracle.ntt(x, width=width, batch=batch, threads=8))
        tf.intt(y, width=width, batch=batch)
        assert np.array_equal(y, x)
        off = oracle.bfe_new(7)
        one = x[:n * width]
        ev = tf.fast_coset_evaluate(one[: (n // 2 + 3) * width], off, n, width=width)
        assert np.array_equal(ev, oracle.coset_evaluate(one[: (n // 2 + 3) * width], off, n, width=width))
        assert np.array_equal(tf.fast_coset_interpolate(one, off, width=width), oracle.coset_interpolate(one, off, width=width))
    finally:
        lib.tf_set_ntt_min_passes(0)


@pytest.mark.parametrize("log_n,width,batch", [(21, 1, 2), (21, 3, 1), (22, 1, 1), (22, 3, 2)])
def test_two_pass_plan_for_2p21_2p22_matches_oracle(tf, oracle, log_n, width, batch):
    """2^21 / 2^22 points in two global passes (a 2048-point pass = pairs of 1024-point workgroups that share their input,
    DESIGN 4.1b, tf_set_ntt_two_pass(1) -- or, mode 3, its first pass as ONE workgroup per 2048-row x 8-column tile that loads and
    scales every element once, ntt_col2048_kernel): forward, inverse, coset evaluation (full and zero-padded coefficient lists)
    and interpolation against the oracle's radix-2 sweeps (math/ntt.rs:153-228, polynomial.rs:1374-1399, :1907-1918), and word
    for word against the three-pass plan."""
    import ctypes as C
    n = 1 << log_n
    x = oracle.fill_random(batch * n * width, 2100 + log_n + width)
    lib = tf._lib.lib()
    radix = (C.c_int * 4)()
    got = {}
    try:
        for mode in (3, 1, 0):
            lib.tf_set_ntt_two_pass(mode)
            passes = lib.tf_ntt_plan(n, width, radix)  # (an A/B switch in the environment may rule the two-pass plan out: then three)
            assert passes == 3 if mode == 0 else passes in (2, 3)
            if mode == 3 and passes == 2:
                assert list(radix)[:2] == [11, log_n - 11]
            y = x.copy()
            tf.ntt(y, width=width, batch=batch)
            fwd = y.copy()
            tf.intt(y, width=width, batch=batch)
            assert np.array_equal(y, x)
            off = oracle.bfe_new(7)
            one = x[:n * width]
            ev_full = tf.fast_coset_evaluate(one, off, n, width=width)
            ev_pad = tf.fast_coset_evaluate(one[: (n // 2 + 3) * width], off, n, width=width)
            ev_short = tf.fast_coset_evaluate(one[: (n - 5) * width], off, n, width=width)
            ci = tf.fast_coset_interpolate(one, off, width=width)
            got[mode] = (fwd, ev_full, ev_pad, ev_short, ci)
    finally:
        lib.tf_set_ntt_two_pass(-1)
    for a, b, c8 in zip(got[0], got[1], got[3]):
        assert np.array_equal(a, b)
        assert np.array_equal(a, c8)
    fwd, ev_full, ev_pad, ev_short, ci = got[1]
    assert np.array_equal(fwd, oracle.ntt(x, width=width, batch=batch, threads=8))
    off = oracle.bfe_new(7)
    one = x[:n * width]
    assert np.array_equal(ev_full, oracle.coset_evaluate(one, off, n, width=width))
    assert np.array_equal(ev_pad, oracle.coset_evaluate(one[: (n // 2 + 3) * width], off, n, width=width))
    assert np.array_equal(ev_short, oracle.coset_evaluate(one[: (n - 5) * width], off, n, width=width))
    assert np.array_equal(ci, oracle.coset_interpolate(one, off, width=width))


@pytest.mark.parametrize("width,batch", [(1, 3), (3, 2)])
def test_radix4_last_pass_plan_for_2p22_matches_oracle(tf, oracle, width, batch):
    """2^22 points as 1024 x 4096 with the 4096-point last pass run as four 1024-point classes per tile (ntt_kernels.h PRE4: a measured
    loss, so the laboratory library only -- tf_set_ntt_two_pass(2); the product library treats mode 2 as mode 1): forward NTT and
    coset evaluation with full, zero-padded and ragged coefficient lists against the oracle's radix-2 sweeps (math/ntt.rs:153-228,
    polynomial.rs:1374-1399) and word for word against the 2048 x 2048 plan (mode 1) and the three-pass plan (mode 0); also
    device-resident with several polynomials per call and from an unaligned output pointer."""
    import torch

    n = 1 << 22
    x = oracle.fill_random(batch * n * width, 2300 + width)
    lib = tf._lib.lib()
    off = oracle.bfe_new(7)
    one = x[:n * width]
    got = {}
    try:
        for mode in (2, 1, 0, -1):
            lib.tf_set_ntt_two_pass(mode)
            y = x.copy()
            tf.ntt(y, width=width, batch=batch)
            ev_full = tf.fast_coset_evaluate(one, off, n, width=width)
            ev_pad = tf.fast_coset_evaluate(one[: (n // 2 + 3) * width], off, n, width=width)
            ev_short = tf.fast_coset_evaluate(one[: (n - 5) * width], off, n, width=width)
            # several polynomials per call, device-resident, output 8 bytes off a cache line
            c = torch.from_numpy(x.view(np.int64)).cuda()
            o = torch.empty(batch * n * width + 1, dtype=torch.int64, device="cuda")
            tf.device.coset_evaluate(c, n, off, o[1:], n, batch=batch, width=width)
            torch.cuda.synchronize()
            got[mode] = (y, ev_full, ev_pad, ev_short, o[1:].cpu().numpy().view(np.uint64))
    finally:
        lib.tf_set_ntt_two_pass(-1)
    for mode in (1, 0, -1):
        for a, b in zip(got[2], got[mode]):
            assert np.array_equal(a, b), mode
    fwd, ev_full, ev_pad, ev_short, ev_batch = got[2]
    assert np.array_equal(fwd, oracle.ntt(x, width=width, batch=batch, threads=8))
    assert np.array_equal(ev_full, oracle.coset_evaluate(one, off, n, width=width))
    assert np.array_equal(ev_pad, oracle.coset_evaluate(one[: (n // 2 + 3) * width], off, n, width=width))
    assert np.array_equal(ev_short, oracle.coset_evaluate(one[: (n - 5) * width], off, n, width=width))
    assert np.array_equal(ev_batch, oracle.coset_evaluate_batch(x, off, n, batch, width=width, threads=batch))


@pytest.mark.parametrize("length", [0, 1, 9, 10, 11, 25, 40])
def test_sponge_absorb_squeeze_matches_oracle(tf, oracle, length):
    """impl Sponge for Tip5 (tip5/mod.rs:677-699) + pad_and_absorb_all (sponge.rs:41-55), three sponges stepped together"""
    batch = 3
    data = oracle.fill_random(batch * max(length, 1), 700 + length).reshape(batch, -1)[:, :length]
    sp = tf.Tip5Sponge.init(batch)
    sp.pad_and_absorb_all(data)
    for b in range(batch):
        # the digest of hash_varlen is the first five words of the state after padding and absorbing (mod.rs:617-623)
        assert np.array_equal(sp.state[b, :5], oracle.hash_varlen(data[b]))
    # squeeze twice: rate part, then permutation (mod.rs:693-698)
    want = sp.state.copy()
    for _ in range(2):
        got = sp.squeeze()
        for b in range(batch):
            assert np.array_equal(got[b], want[b, :10])
            want[b] = oracle.tip5_permutation(want[b])
        assert np.array_equal(sp.state, want)
    # a further absorb overwrites the rate part only (mod.rs:684-691)
    chunk = oracle.fill_random(batch * 10, 800 + length).reshape(batch, 10)
    sp.absorb(chunk)
    for b in range(batch):
        assert np.array_equal(sp.state[b], oracle.absorb(want[b], chunk[b]))
    fixed = tf.Tip5Sponge(1, fixed_length=True)
    assert list(fixed.state[0]) == [0] * 10 + [0xFFFFFFFF] * 6  # Tip5::new(Domain::FixedLength), mod.rs:511-526


@pytest.mark.parametrize("count", [1, 7, 8, 9, 15, 17, 2040, 2047, 2048, 2049, 2063, 8192, 8193, 8207, 8208, 40000, 65599])
def test_tip5_both_kernel_shapes_match_oracle(tf, oracle, count):
    """launches of <= 2^13 permutation chains run 16 lanes per permutation -- 32 lanes (a row pair, the circulant's rotation terms split over
    the two rows) up to 8 chains per compute unit, i.e. 2048 on an MI355X: counts around that switch and counts that leave a workgroup of
    8 / 16 chains ragged --, larger ones in the matrix-pipe form (4 lanes per permutation, 16 permutations per wave: counts that are not
    multiples of 16 leave clamped lanes in the last wave): permutation, hash_pair and hash_varlen on both sides of every switch, every
    output word"""
    states = oracle.fill_random(count * 16, 900 + count)
    got = states.copy()
    tf.Tip5.permute_states(got)
    idx = range(count) if count <= 8300 else sorted(set([0, 1, 15, 16, count // 2, count - 17, count - 16, count - 2, count - 1]))
    for i in idx:
        assert np.array_equal(got[16 * i:16 * i + 16], oracle.tip5_permutation(states[16 * i:16 * i + 16]))
    pairs = states[:count * 10]
    assert np.array_equal(tf.Tip5.hash_pairs(pairs), oracle.hash_pairs(pairs))
    for row_len in (0, 7, 10, 13, 29):
        rows = oracle.fill_random(count * row_len, 901 + count)
        assert np.array_equal(tf.Tip5.hash_varlen_rows(rows, row_len), oracle.hash_varlen_rows(rows, row_len))


def test_tip5_matrix_pipe_extreme_words(tf, oracle):
    """the matrix-pipe MDS is exact f64 arithmetic on 32-bit halves: states of the largest canonical words (p - 1, 2^32 - 1 halves),
    zeros and single non-zero words, in a launch large enough for that kernel"""
    count = 8192 + 64
    p = (1 << 64) - (1 << 32) + 1
    states = oracle.fill_random(count * 16, 4242).reshape(count, 16)
    states[0, :] = p - 1
    states[1, :] = 0
    states[2, :] = 0xFFFFFFFF
    states[3, :] = 0xFFFFFFFF00000000
    for k in range(16):
        states[4 + k, :] = 0
        states[4 + k, k] = p - 1
    states[20, ::2] = p - 1
    states[21, 1::2] = 0xFFFFFFFEFFFFFFFF
    flat = states.reshape(-1).copy()
    got = flat.copy()
    tf.Tip5.permute_states(got)
    for i in list(range(24)) + [count - 1]:
        assert np.array_equal(got[16 * i:16 * i + 16], oracle.tip5_permutation(flat[16 * i:16 * i + 16]))


@pytest.mark.parametrize("count", [1, 33, 9000])
def test_tip5_trace_matrix_pipe(tf, oracle, count):
    """Tip5::trace (mod.rs:538-548) runs in the matrix-pipe form at every size: all six states of every permutation"""
    import torch

    states = oracle.fill_random(count * 16, 950 + count)
    d = torch.from_numpy(states.view(np.int64)).cuda()
    tr = torch.empty(count * 96, dtype=torch.int64, device="cuda")
    tf.device.tip5_trace_(d, tr)
    torch.cuda.synchronize()
    got_tr, got_st = tr.cpu().numpy().view(np.uint64), d.cpu().numpy().view(np.uint64)
    for i in (range(count) if count < 100 else [0, 15, 16, 4500, count - 9, count - 1]):
        want, after = oracle.tip5_trace(states[16 * i:16 * i + 16])
        assert np.array_equal(got_tr[96 * i:96 * i + 96], want.reshape(-1))
        assert np.array_equal(got_st[16 * i:16 * i + 16], after)


def test_hash_varlen_one_long_input(tf, oracle):
    """a single long input is a sequential absorb chain (tip5/mod.rs:617-623): 16-lane path, 1000 permutations"""
    data = oracle.fill_random(9999, 77)
    assert np.array_equal(tf.Tip5.hash_varlen(data), oracle.hash_varlen(data))


def _dev_random(numel, seed):
    """canonical raw words generated on the device (uniform below 2^62 < p): full-size property tests only"""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randint(0, 1 << 62, (numel,), dtype=torch.int64, device="cuda", generator=g)


def test_config4_full_size_64_xfe_polys_roundtrip(tf, oracle):
    """BASELINE config 4 at full size: 64 XFE polynomials x 2^22 coefficients on the coset 7 * <w>.  Properties over all
    2^28 points: interpolation returns the coefficients; evaluation is linear (f + g); one polynomial is checked word
    for word in test_config4_xfe_coset_eval_2pow22."""
    import torch

    n, batch = 1 << 22, 64
    off = oracle.bfe_new(7)
    c = _dev_random(3 * n * batch, 44)
    ev = torch.empty_like(c)
    tf.device.coset_evaluate(c, n, off, ev, n, batch=batch, width=3)
    back = torch.empty_like(c)
    tf.device.coset_interpolate(ev, n, off, back, batch=batch, width=3)
    torch.cuda.synchronize()
    assert torch.equal(back, c)
    # linearity on the first two polynomials: eval(f) + eval(g) == eval(f + g), field addition done by the oracle on a sample
    f, g = _to_host(c[:3 * 4096]), _to_host(c[3 * n:3 * n + 3 * 4096])
    idx = np.arange(0, 3 * n, 3 * 4099)[:64]
    h = _to_dev(np.array([oracle.bfe_add(int(a), int(b)) for a, b in zip(f, g)], dtype=np.uint64))
    # low-degree pair: the first 4096 coefficients of f and g and their sum, evaluated on the full 2^22 coset
    e = [torch.empty(3 * n, dtype=torch.int64, device="cuda") for _ in range(3)]
    for src, dst in ((c[:3 * 4096], e[0]), (c[3 * n:3 * n + 3 * 4096], e[1]), (h, e[2])):
        tf.device.coset_evaluate(src.contiguous(), 4096, off, dst, n, batch=1, width=3)
    torch.cuda.synchronize()
    ef, eg, eh = (_to_host(t) for t in e)
    for i in idx:
        assert int(eh[i]) == oracle.bfe_add(int(ef[i]), int(eg[i]))


def test_config5_one_rank_of_eight(tf, oracle):
    """BASELINE config 5 as rank 0 of 8 sees it (sharding.shard_range): 512 of the 4096 x 2^20 NTTs and 32 of the 256 trees
    of 2^20 leaves.  Round trip over the whole shard; first and last transform and tree against the oracle."""
    import torch
    from twenty_first_amd import sharding

    n = 1 << 20
    lo, hi = sharding.shard_range(4096, 8, 0)
    batch = hi - lo
    assert batch == 512
    d = _dev_random(n * batch, 55)
    orig = d.clone()
    tf.device.ntt_(d, n, batch=batch)
    torch.cuda.synchronize()
    for b in (0, batch - 1):
        assert np.array_equal(_to_host(d[b * n:(b + 1) * n]), oracle.ntt(_to_host(orig[b * n:(b + 1) * n])))
    tf.device.ntt_(d, n, batch=batch, inverse=True)
    torch.cuda.synchronize()
    assert torch.equal(d, orig)
    del d, orig
    tlo, thi = sharding.shard_range(256, 8, 0)
    trees = thi - tlo
    assert trees == 32
    leaves = _dev_random(5 * n * trees, 56)
    nodes = torch.empty(10 * n * trees, dtype=torch.int64, device="cuda")
    roots = torch.empty(5 * trees, dtype=torch.int64, device="cuda")
    tf.device.merkle_build(leaves, n, nodes, batch=trees)
    tf.device.merkle_root(leaves, n, roots, batch=trees)
    torch.cuda.synchronize()
    threads = min(64, os.cpu_count() or 8)
    for t in (0, trees - 1):
        want = oracle.merkle_build(_to_host(leaves[5 * n * t:5 * n * (t + 1)]), threads=threads)
        assert np.array_equal(_to_host(nodes[10 * n * t:10 * n * (t + 1)]), want)
    got_roots = _to_host(roots).reshape(trees, 5)
    all_nodes = nodes.view(trees, 2 * n, 5)
    assert np.array_equal(got_roots, _to_host(all_nodes[:, 1, :].contiguous().view(-1)).reshape(trees, 5))  # frugal root == node 1


def test_every_entry_point_validates_its_arguments(tf):
    """Raw C-ABI calls: null pointers come back as TF_ERR_NULL_POINTER (7), bad lengths as the reference's panics, and
    zero-sized work is a successful no-op -- never a crash (the reference panics or returns Err in the same places)."""
    import ctypes as C

    lib = tf._lib.lib()
    buf = (C.c_uint64 * 4096)()
    out = (C.c_uint64 * 16384)()
    NULL = None
    # ntt / intt
    assert lib.tf_ntt_bfe(NULL, 8, 1, 0) == 7 and lib.tf_ntt_xfe(NULL, 8, 1, 1) == 7
    assert lib.tf_ntt_bfe(buf, 12, 1, 0) == 4 and lib.tf_ntt_bfe(buf, 0, 5, 0) == 0 and lib.tf_ntt_bfe(buf, 8, 0, 0) == 0
    assert lib.tf_ntt_bfe_dev(NULL, 8, 1, 0, None) == 7
    # coset evaluation / interpolation / extrapolation
    assert lib.tf_coset_eval_bfe(buf, 9, 7, out, 8, 1) == 6       # order <= degree  (polynomial.rs:1388)
    assert lib.tf_coset_eval_bfe(buf, 4, 7, out, 12, 1) == 4
    assert lib.tf_coset_eval_bfe(NULL, 4, 7, out, 8, 1) == 7 and lib.tf_coset_eval_xfe(buf, 4, 7, NULL, 8, 1) == 7
    assert lib.tf_coset_eval_bfe(NULL, 0, 7, out, 8, 1) == 0      # the zero polynomial needs no coefficient pointer
    assert lib.tf_coset_interpolate_bfe(buf, 12, 7, out, 1) == 4 and lib.tf_coset_interpolate_bfe(buf, 8, 0, out, 1) == 12
    assert lib.tf_coset_interpolate_xfe(NULL, 8, 7, out, 1) == 7
    assert lib.tf_coset_extrapolate_bfe(7, buf, 0, 1, buf, 2, out) == 4 and lib.tf_coset_extrapolate_bfe(7, buf, 24, 1, buf, 2, out) == 4
    assert lib.tf_coset_extrapolate_bfe(0, buf, 8, 1, buf, 2, out) == 12 and lib.tf_coset_extrapolate_xfe(7, NULL, 8, 1, buf, 2, out) == 7
    assert lib.tf_coset_extrapolate_bfe(7, buf, 8, 1, buf, 0, out) == 0
    # products / evaluation
    assert lib.tf_poly_mul_bfe(NULL, 3, buf, 3, out, 1) == 7 and lib.tf_poly_mul_xfe(buf, 3, buf, 3, NULL, 1) == 7
    assert lib.tf_poly_square_bfe(NULL, 3, out, 1) == 7
    assert lib.tf_poly_batch_evaluate_bfe(NULL, 3, buf, 2, out) == 7 and lib.tf_poly_batch_evaluate_xfe(buf, 3, buf, 2, NULL) == 7
    assert lib.tf_poly_batch_evaluate_bfe(NULL, 0, buf, 2, out) == 0 and lib.tf_poly_batch_evaluate_bfe(buf, 3, NULL, 0, out) == 0
    # Tip5 / Merkle
    assert lib.tf_tip5_permute(NULL, 1) == 7 and lib.tf_tip5_permute(NULL, 0) == 0
    assert lib.tf_tip5_hash_pairs(NULL, out, 1) == 7 and lib.tf_tip5_hash_pairs(buf, NULL, 1) == 7
    assert lib.tf_tip5_hash_varlen_rows(NULL, 3, 2, out) == 7 and lib.tf_tip5_hash_varlen_rows(NULL, 0, 2, out) == 0
    assert lib.tf_merkle_build(buf, 0, out, 1) == 1 and lib.tf_merkle_build(buf, 6, out, 1) == 2 and lib.tf_merkle_build(NULL, 8, out, 1) == 7
    assert lib.tf_merkle_root(buf, 6, out, 1) == 2 and lib.tf_merkle_root(buf, 8, NULL, 1) == 7
    assert lib.tf_merkle_from_rows(NULL, 3, 8, out, 1) == 7 and lib.tf_merkle_from_rows(buf, 3, 6, out, 1) == 2
    cnt = C.c_size_t(0)
    idx = (C.c_uint64 * 2)(0, 9)
    assert lib.tf_merkle_auth_structure_indices(8, idx, 2, out, 64, C.byref(cnt)) == 11   # leaf 9 of 8
    assert lib.tf_merkle_auth_structure_indices(12, idx, 1, out, 64, C.byref(cnt)) == 2
    assert lib.tf_merkle_auth_structure_indices(8, idx, 1, out, 0, C.byref(cnt)) == 0 and cnt.value == 3   # sizing call: count only
    assert lib.tf_merkle_auth_structure_indices(8, idx, 1, None, 64, C.byref(cnt)) == 0 and cnt.value == 3  # NULL buffer: sizing call too
    assert lib.tf_merkle_auth_structure_indices(8, idx, 1, out, 2, C.byref(cnt)) == 13 and cnt.value == 3  # TF_ERR_BUFFER_TOO_SMALL, nothing written
    assert lib.tf_status_string(7) == b"TF_ERR_NULL_POINTER" and lib.tf_status_string(13) == b"TF_ERR_BUFFER_TOO_SMALL"
    # after all that the device still works
    x = np.arange(8, dtype=np.uint64)
    y = x.copy()
    tf.ntt(y)
    tf.intt(y)
    assert np.array_equal(x, y)


@pytest.mark.parametrize("log_n,width", [(27, 1), (26, 3)])
def test_huge_single_transforms_match_oracle(tf, oracle, log_n, width):
    """The largest single transforms the suite can afford (tools/check_huge.py covers 2^28 .. 2^31 outside pytest): a 2^27-point
    BFieldElement slice (1 GiB, three passes) and a 2^26-point XFieldElement slice (1.5 GiB), forward word for word against the
    oracle (math/ntt.rs:67-82), then the inverse round trip."""
    import torch

    n = 1 << log_n
    x = oracle.fill_random(n * width, 0xABC + log_n)
    want = oracle.ntt(x, width=width)
    d = _to_dev(x)
    tf.device.ntt_(d, n, width=width)
    torch.cuda.synchronize()
    assert np.array_equal(_to_host(d), want)
    del want
    tf.device.ntt_(d, n, width=width, inverse=True)
    torch.cuda.synchronize()
    assert np.array_equal(_to_host(d), x)


def test_device_fill_random_is_the_oracle_sequence(tf, oracle):
    """tf_debug_fill_random_dev (bench.py's synthetic inputs) == tfo.fill_random, also from an offset"""
    import torch

    t = torch.empty(100_003, dtype=torch.int64, device="cuda")
    tf.device.fill_random(t, 0x7F210002)
    torch.cuda.synchronize()
    want = oracle.fill_random(200_000, 0x7F210002)
    assert np.array_equal(_to_host(t), want[:100_003])
    tf.device.fill_random(t, 0x7F210002, first_index=77_777)
    torch.cuda.synchronize()
    assert np.array_equal(_to_host(t), want[77_777:77_777 + 100_003])
    assert int(_to_host(t).max()) < P


@pytest.mark.parametrize("tile_mib,pipe", [(8, 2), (16, 3), (8, 4), (64, 1)])
def test_pipelined_batch_tiles_are_bit_exact(tf, oracle, tile_mib, pipe):
    """TF_NTT_TILE_BYTES x TF_NTT_PIPE: the batch tiles of a multi-pass transform dealt to side streams (fork/join with events
    on the caller's stream) give the same words as the one-stream plan, for in-place transforms and coset evaluations, and the
    next call on the caller's stream sees the finished result."""
    import torch

    L = tf.lib()
    n, batch = 1 << 16, 40  # 20 MiB per call: 3 .. 5 tiles of 8 MiB, a ragged last tile with 16 MiB
    x = oracle.fill_random(n * batch, 31)
    want = oracle.ntt(x, batch=batch, threads=8)
    try:
        L.tf_set_ntt_tile_bytes(tile_mib << 20)
        L.tf_set_ntt_pipe(pipe)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            d = _to_dev(x)
            tf.device.ntt_(d, n, batch=batch)
            e = d.clone()                       # enqueued behind the join on the same stream
            tf.device.ntt_(d, n, batch=batch, inverse=True)
        s.synchronize()
        assert np.array_equal(_to_host(e), want)
        assert np.array_equal(_to_host(d), x)
        c = oracle.fill_random(3 * 1000 * 6, 32)
        out = torch.empty(3 * 4096 * 6, dtype=torch.int64, device="cuda")
        tf.device.coset_evaluate(_to_dev(c), 1000, oracle.bfe_new(7), out, 4096, batch=6, width=3)
        torch.cuda.synchronize()
        for b in range(6):
            assert np.array_equal(_to_host(out[b * 3 * 4096:(b + 1) * 3 * 4096]), oracle.coset_evaluate(c[b * 3000:(b + 1) * 3000], oracle.bfe_new(7), 4096, width=3))
    finally:
        L.tf_set_ntt_tile_bytes(0)
        L.tf_set_ntt_pipe(0)  # automatic


def test_wrapper_size_checks_raise_before_the_abi(tf):
    """device.py validates every tensor against (n, batch, width): a wrong size is a ValueError, never an out-of-bounds access"""
    import torch

    z = lambda k: torch.zeros(k, dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        tf.device.tip5_hash_varlen_rows(z(99), 10, z(50))          # rows != n_rows * row_len
    with pytest.raises(ValueError):
        tf.device.merkle_root(z(5 * 8), 8, z(4))                   # root_out too small
    with pytest.raises(ValueError):
        tf.device.coset_interpolate(z(64), 64, 1, z(32))           # out too small
    with pytest.raises(ValueError):
        tf.device.hadamard(z(30), z(33), z(30), width=3)
    with pytest.raises(ValueError):
        tf.device.poly_mul(z(16), 8, z(16), 8, z(29), batch=2)     # out != batch * (na + nb - 1)
    with pytest.raises(ValueError):
        tf.device.lde(z(64), 64, 1, z(100), 128, 1)
    with pytest.raises(ValueError):
        tf.device.merkle_from_rows(z(70), 10, 8, z(80))
    with pytest.raises(ValueError):
        tf.device.hash_table_rows(z(100), 16, 4, z(80), col_stride=8)   # col_stride < n_rows * width
    with pytest.raises(ValueError):
        tf.device.merkle_from_columns(z(63), 16, 4, z(160))
    with pytest.raises(ValueError):
        tf.device.tip5_permute_(z(40))                              # not a multiple of 16 words
    with pytest.raises(ValueError):
        tf.device.ntt_(z(64), 64, width=2)


def test_free_functions_validate_and_trim(tf, oracle):
    """fast_multiply / fast_square reject sizes that are not batch * n * width; fast_coset_evaluate compares
    `order` with the degree (math/polynomial.rs:1388): high-order zero coefficients do not count"""
    with pytest.raises(ValueError):
        tf.fast_multiply(np.zeros(7, np.uint64), np.zeros(8, np.uint64), batch=2)
    with pytest.raises(ValueError):
        tf.fast_square(np.zeros(7, np.uint64), width=3)
    c = oracle.fill_random(3 * 40, 8)
    padded = np.concatenate([c, np.zeros(3 * 30, np.uint64)])   # 70 coefficients, the top 30 zero: degree 39 < order 64
    got = tf.fast_coset_evaluate(padded, oracle.bfe_new(7), 64, width=3)
    assert np.array_equal(got, oracle.coset_evaluate(c, oracle.bfe_new(7), 64, width=3))
    with pytest.raises(tf.NttPanic):
        tf.fast_coset_evaluate(oracle.fill_random(3 * 70, 9), oracle.bfe_new(7), 64, width=3)   # degree 69 >= order


@pytest.mark.parametrize("count", [1, 3, 64, 1000, 1 << 14])
def test_tip5_trace_matches_oracle(tf, oracle, count):
    """Tip5::trace (tip5/mod.rs:538-548): count x 6 x 16 words, row 0 the input state, row 5 the permutation's output
    (test :1557-1565), every row equal to the oracle's; the states end permuted like `&mut self`."""
    import torch

    s0 = oracle.fill_random(16 * count, 2100 + count)
    s = s0.copy()
    trace = tf.Tip5.trace_states(s)
    assert trace.shape == (count, 6, 16)
    assert np.array_equal(trace[:, 0, :].reshape(-1), s0)
    assert np.array_equal(trace[:, 5, :].reshape(-1), s)
    perm = s0.copy()
    tf.Tip5.permute_states(perm)
    assert np.array_equal(perm, s)
    for i in sorted({0, count // 2, count - 1}):
        want, _ = oracle.tip5_trace(s0[16 * i: 16 * i + 16])
        assert np.array_equal(trace[i], want)
    ds = torch.from_numpy(s0.view(np.int64).copy()).cuda()
    dt = torch.empty(96 * count, dtype=torch.int64, device="cuda")
    tf.device.tip5_trace_(ds, dt)
    torch.cuda.synchronize()
    assert np.array_equal(dt.cpu().numpy().view(np.uint64).reshape(count, 6, 16), trace)
    with pytest.raises(ValueError):
        tf.device.tip5_trace_(ds, dt[:-1])


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("log_n", [6, 7, 8, 9, 10, 11, 12])
def test_latency_shaped_transform_matches_oracle(tf, oracle, log_n, width):
    """The 8-elements-per-thread kernel for calls with little work (ntt_lat_kernel, tf_set_ntt_latency_kernel): forced on and
    off, forward / inverse / zero-padded coset evaluation input / ragged batch, word for word against the oracle's radix-2
    sweeps (math/ntt.rs:153-228) and against the pass kernels."""
    n = 1 << log_n
    lib = tf._lib.lib()
    batch = 5
    x = oracle.fill_random(batch * n * width, 3000 + log_n + width)
    want = oracle.ntt(x, width=width, batch=batch, threads=4)
    got = {}
    try:
        for mode in (1, 0):
            lib.tf_set_ntt_latency_kernel(mode)
            y = x.copy()
            tf.ntt(y, width=width, batch=batch)
            assert np.array_equal(y, want), (mode, "forward")
            tf.intt(y, width=width, batch=batch)
            assert np.array_equal(y, x), (mode, "inverse")
            # fast_multiply of n / 2 by n / 2 - 3 coefficients: zero-padded forward transforms of order n, product, inverse
            a, b = x[: (n // 2) * width], x[n * width: n * width + (n // 2 - 3) * width]
            got[mode] = tf.fast_multiply(a, b, width=width)
            off = oracle.bfe_new(7)
            ev = tf.fast_coset_evaluate(a, off, n, width=width)
            assert np.array_equal(ev, oracle.coset_evaluate(a, off, n, width=width)), (mode, "coset")
            # the coset scalings ride in the latency kernel's load / store (round 5): full-length and ragged batches, and back
            for nc, polys in ((n, 1), (n - 3, 3), (max(1, n // 3), 2)):
                c = x[: polys * nc * width]
                evb = tf.fast_coset_evaluate(c, off, n, width=width, batch=polys)
                assert np.array_equal(evb, oracle.coset_evaluate_batch(c, off, n, polys, width=width, threads=polys)), (mode, "coset batch", nc)
                back = tf.fast_coset_interpolate(evb, off, width=width, batch=polys)
                for k in range(polys):
                    assert np.array_equal(back[k * n * width:(k + 1) * n * width], oracle.coset_interpolate(evb[k * n * width:(k + 1) * n * width], off, width=width)), (mode, "interpolate", nc)
                    assert np.array_equal(back[k * n * width: k * n * width + nc * width], c[k * nc * width:(k + 1) * nc * width])
    finally:
        lib.tf_set_ntt_latency_kernel(-1)
    assert np.array_equal(got[0], got[1])
    assert np.array_equal(got[1], oracle.poly_mul(x[: (n // 2) * width], x[n * width: n * width + (n // 2 - 3) * width], width=width))


@pytest.mark.parametrize("log_n,width,batch", [(13, 1, 3), (14, 3, 2), (15, 1, 1), (16, 1, 2), (16, 3, 1), (17, 1, 1), (18, 3, 1), (19, 1, 1), (20, 1, 1)])
def test_latency_shaped_two_pass_plan_matches_oracle(tf, oracle, log_n, width, batch):
    """2^13 .. 2^20 points with little work per call (one slice is the reference's own call shape, math/ntt.rs:67-82): the two
    passes n = N1 N2 on the 8-elements-per-thread stages (ntt_lat2_kernel), forced on and off: forward, inverse, fast_multiply
    (zero-padded forward transforms, product fused into the inverse's load) and coset evaluation against the oracle."""
    n = 1 << log_n
    lib = tf._lib.lib()
    x = oracle.fill_random(batch * n * width, 3100 + log_n + width)
    want = oracle.ntt(x, width=width, batch=batch, threads=8)
    got = {}
    a, b = x[: (n // 2) * width], x[(n // 2) * width: (n // 2) * width + (n // 2 - 3) * width]
    try:
        for mode in (1, 0):
            lib.tf_set_ntt_latency_kernel(mode)
            y = x.copy()
            tf.ntt(y, width=width, batch=batch)
            assert np.array_equal(y, want), (mode, "forward")
            tf.intt(y, width=width, batch=batch)
            assert np.array_equal(y, x), (mode, "inverse")
            got[mode] = tf.fast_multiply(a, b, width=width)
            # coset evaluation / interpolation with the scalings fused into the two latency-shaped passes (round 5)
            off = oracle.bfe_new(7)
            for nc in (n, n - 5):
                c = x[: batch * nc * width]
                ev = tf.fast_coset_evaluate(c, off, n, width=width, batch=batch)
                assert np.array_equal(ev, oracle.coset_evaluate_batch(c, off, n, batch, width=width, threads=batch)), (mode, "coset", nc)
                back = tf.fast_coset_interpolate(ev, off, width=width, batch=batch)
                assert np.array_equal(back[: n * width], oracle.coset_interpolate(ev[: n * width], off, width=width)), (mode, "interpolate", nc)
                for k in range(batch):
                    assert np.array_equal(back[k * n * width: k * n * width + nc * width], c[k * nc * width:(k + 1) * nc * width])
    finally:
        lib.tf_set_ntt_latency_kernel(-1)
    assert np.array_equal(got[0], got[1])
    if log_n <= 16:
        assert np.array_equal(got[1], oracle.poly_mul(a, b, width=width))


def test_chained_column_pass_gives_the_same_words(tf, oracle):
    """tf_set_ntt_chain: the R = 1024 column pass as chains of k tiles per workgroup with the next tile's loads issued inside the
    store phase (an A/B path, slower) -- 128 transforms of 2^20 points (2048 tiles per launch), k = 2, 3, 8, against
    the one-tile kernel word for word, and the first two transforms against the oracle."""
    import torch

    if not tf._lib.is_ab_build():
        pytest.skip("ntt_col1024_chain_kernel is a measured loss: compiled into the laboratory library only (csrc: make ab; TF_HIP_LIBRARY=.../libtf_hip_ab.so)")
    lib = tf._lib.lib()
    n, batch = 1 << 20, 128
    src = torch.empty(n * batch, dtype=torch.int64, device="cuda")
    tf.device.fill_random(src, 7100)
    ref = src.clone()
    tf.device.ntt_(ref, n, batch=batch)
    torch.cuda.synchronize()
    want = oracle.ntt(src[: 2 * n].cpu().numpy().view(np.uint64), batch=2, threads=2)
    assert np.array_equal(ref[: 2 * n].cpu().numpy().view(np.uint64), want)
    try:
        for k in (2, 3, 8):
            lib.tf_set_ntt_chain(k)
            for inverse in (False, True):
                x = (ref if inverse else src).clone()
                tf.device.ntt_(x, n, batch=batch, inverse=inverse)
                torch.cuda.synchronize()
                assert torch.equal(x, src if inverse else ref), (k, inverse)
    finally:
        lib.tf_set_ntt_chain(0)


# ---- one host-resident batch over several GPUs at the C ABI (tf_*_multi, include/tf_hip.h) ------------------------------
# A one-GPU box lists device 0 several times: several worker threads, streams and copies in flight on one device.
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], "all"])
@pytest.mark.parametrize("width", [1, 3])
def test_multi_device_ntt_matches_oracle(tf, oracle, devices, width):
    """tf_ntt_{bfe,xfe}_multi: a ragged batch (7 transforms: slices of 3 + 2 + 2 over three workers), forward and inverse"""
    n, batch = 1 << 12, 7
    x = oracle.fill_random(n * batch * width, 3100 + width)
    got = x.copy()
    tf.ntt(got, width=width, batch=batch, devices=devices)
    assert np.array_equal(got, oracle.ntt(x, width=width, batch=batch, threads=4))
    tf.intt(got, width=width, batch=batch, devices=devices)
    assert np.array_equal(got, x)


def test_multi_device_fewer_units_than_workers(tf, oracle):
    """two transforms over five workers (three empty slices), one unit, an empty batch"""
    n = 1 << 10
    x = oracle.fill_random(2 * n, 3200)
    got = x.copy()
    tf.ntt(got, batch=2, devices=[0, 0, 0, 0, 0])
    assert np.array_equal(got, oracle.ntt(x, batch=2))
    one = x[:n].copy()
    tf.ntt(one, batch=1, devices=[0, 0])
    assert np.array_equal(one, oracle.ntt(x[:n]))
    tf.ntt(np.zeros(0, dtype=np.uint64), batch=0, devices=[0, 0])


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0], "all"])
def test_multi_device_merkle_and_coset_match_oracle(tf, oracle, devices):
    """tf_merkle_{build,root}_multi and tf_coset_eval_{bfe,xfe}_multi: every word of every unit, ragged splits"""
    n, batch = 1 << 9, 5
    leafs = oracle.fill_random(batch * n * 5, 3300)
    nodes = tf.MerkleTree.build_batch(leafs, n, devices=devices)
    roots = tf.MerkleTree.roots_batch(leafs, n, devices=devices)
    for b in range(batch):
        want = oracle.merkle_build(leafs[b * n * 5:(b + 1) * n * 5])
        assert np.array_equal(nodes[b].reshape(-1), want)
        assert np.array_equal(roots[b], want[5:10])
    off = oracle.bfe_new(7)
    for width in (1, 3):
        nc, order, polys = 300, 1 << 10, 6
        c = oracle.fill_random(polys * nc * width, 3400 + width)
        got = tf.fast_coset_evaluate(c, off, order, width=width, batch=polys, devices=devices)
        assert np.array_equal(got, oracle.coset_evaluate_batch(c, off, order, polys, width=width, threads=2))


@pytest.mark.parametrize("log_n,batch,devices", [(20, 1, [0, 0, 0, 0]), (12, 1, [0, 0]), (10, 3, [0] * 8), (3, 1, [0] * 8), (2, 1, [0, 0, 0, 0]),
                                                  (1, 1, [0, 0]), (9, 2, [0, 0, 0, 0, 0])])
def test_multi_device_single_tree_is_split_into_subtrees(tf, oracle, log_n, batch, devices):
    """Fewer trees than listed devices: tf_merkle_{build,root}_multi cut every tree into subtrees the way MerkleTree::par_new cuts it over
    its threads (util_types/merkle_tree.rs:165-212, :247-275), one subtree per worker, the top layers on devices[0]; EVERY node of
    every tree equals the oracle's par_new and the single-device call, nodes[0] is the zero digest, the roots agree."""
    n = 1 << log_n
    S = tf.lib().tf_merkle_multi_subtrees(n, batch, len(devices))
    assert S >= 1 and batch * S <= len(devices) and (S == 1 or n // S >= 2)
    if (log_n, batch) in ((20, 1), (12, 1), (10, 3), (3, 1), (2, 1)):
        assert S > 1  # these shapes do take the split
    leafs = oracle.fill_random(batch * n * 5, 3600 + log_n)
    nodes = tf.MerkleTree.build_batch(leafs, n, devices=devices)
    roots = tf.MerkleTree.roots_batch(leafs, n, devices=devices)
    single = tf.MerkleTree.build_batch(leafs, n)
    assert np.array_equal(nodes, single)
    for b in range(batch):
        want = oracle.merkle_build(leafs[b * n * 5:(b + 1) * n * 5], threads=4)
        assert np.array_equal(nodes[b].reshape(-1), want)
        assert not nodes[b][0].any()
        assert np.array_equal(roots[b], want[5:10])


def test_multi_device_one_2p24_leaf_tree_over_four_workers(tf, oracle):
    """BASELINE configs[2] through the one-call form: ONE 2^24-leaf tree over devices = [0, 0, 0, 0] (four subtrees of 2^22 leaves, two
    top layers on the first device), all 2^25 nodes against the oracle's par_new."""
    n = 1 << 24
    leafs = oracle.fill_random(n * 5, 0x7F210003)
    nodes = tf.MerkleTree.build_batch(leafs, n, devices=[0, 0, 0, 0])
    want = oracle.merkle_build(leafs, threads=min(64, os.cpu_count() or 1))
    assert np.array_equal(nodes.reshape(-1), want)
    assert np.array_equal(tf.MerkleTree.roots_batch(leafs, n, devices=[0, 0, 0, 0])[0], want[5:10])


def test_one_host_thread_round_robin_never_blocks(tf, oracle):
    """BASELINE configs[4]'s shape at the boundary a one-process caller binds, WITHOUT PCIe in the loop ("eight GPUs from
    one thread"; the reference's callers are one process, math/ntt.rs:250-274): one host thread, one device buffer + stream per listed
    device -- device 0 four times here --, tf_set_device(g); tf_ntt_bfe_dev / tf_merkle_build_dev / tf_coset_eval_xfe_dev round-robin, three
    rounds deep.  After tf_prepare_* every call must only ENQUEUE: its stream is still busy when it returns (hipStreamQuery), and the
    host spends a small fraction of the time the queued work takes -- a first-use table build, a hipMemcpyToSymbol or a hipMalloc
    hiding a device-wide synchronisation in any of them would fail one or the other.  Results word for word against the oracle."""
    import time

    import torch

    lib = tf.lib()
    devs = [0, 0, 0, 0]
    n, batch, nl, trees, nc, order, polys = 1 << 20, 48, 1 << 18, 8, 1 << 15, 1 << 16, 24
    off = oracle.bfe_new(7)
    for d in sorted(set(devs)):  # start-up: once per device and shape
        tf.set_device(d)
        assert lib.tf_prepare_ntt(n, batch, 1, 0) == 0
        assert lib.tf_prepare_merkle(nl, trees) == 0
        assert lib.tf_prepare_coset_eval(nc, off, order, polys, 3) == 0
    streams = [torch.cuda.Stream(device=d) for d in devs]
    x = [torch.empty(n * batch, dtype=torch.int64, device=f"cuda:{d}") for d in devs]
    leaves = [torch.empty(5 * nl * trees, dtype=torch.int64, device=f"cuda:{d}") for d in devs]
    nodes = [torch.empty(10 * nl * trees, dtype=torch.int64, device=f"cuda:{d}") for d in devs]
    coeffs = [torch.empty(3 * nc * polys, dtype=torch.int64, device=f"cuda:{d}") for d in devs]
    evals = [torch.empty(3 * order * polys, dtype=torch.int64, device=f"cuda:{d}") for d in devs]
    for g, d in enumerate(devs):
        tf.set_device(d)
        tf.device.fill_random(x[g], 5000 + g, stream=streams[g])
        tf.device.fill_random(leaves[g], 5100 + g, stream=streams[g])
        tf.device.fill_random(coeffs[g], 5200 + g, stream=streams[g])
    for st in streams:
        st.synchronize()
    host_us = []

    def one_round(record):
        for g, d in enumerate(devs):
            tf.set_device(d)
            for kind, call in (("ntt", lambda: tf.device.ntt_(x[g], n, batch=batch, stream=streams[g])),
                               ("merkle_build", lambda: tf.device.merkle_build(leaves[g], nl, nodes[g], batch=trees, stream=streams[g])),
                               ("coset_evaluate", lambda: tf.device.coset_evaluate(coeffs[g], nc, off, evals[g], order, batch=polys, width=3, stream=streams[g]))):
                t = time.perf_counter()
                call()
                host_us.append((round((time.perf_counter() - t) * 1e6), kind, g))
                record.append(not streams[g].query())

    # one untimed round first: with ONE physical device listed four times, four calls in flight need four scratch blocks of that
    # device's cache where tf_prepare_* left one (on four devices each has its own); the blocks are allocated here, once
    one_round([])
    for st in streams:
        st.synchronize()
    pending, rounds = [], 3
    host_us.clear()
    t0 = time.perf_counter()
    for _ in range(rounds):
        one_round(pending)
    t_host = time.perf_counter() - t0
    for st in streams:
        st.synchronize()
    t_all = time.perf_counter() - t0
    tf.set_device(0)
    assert all(pending), f"{pending.count(False)} of {len(pending)} calls returned with their stream idle: something waited for the device"
    assert t_host < 0.5 * t_all, (f"the host spent {t_host * 1e3:.2f} ms enqueueing {t_all * 1e3:.2f} ms of device work; slowest calls (us, kind, slot): "
                                  f"{sorted(host_us, reverse=True)[:6]}")
    for g in (0, len(devs) - 1):  # four forward transforms in a row; the tree; the evaluation
        want = oracle.fill_random(n * 2, 5000 + g)
        for _ in range(rounds + 1):
            want = oracle.ntt(want, batch=2, threads=2)
        assert np.array_equal(x[g][: 2 * n].cpu().numpy().view(np.uint64), want)
        lv = oracle.fill_random(5 * nl * trees, 5100 + g)
        assert np.array_equal(nodes[g][-10 * nl:].cpu().numpy().view(np.uint64), oracle.merkle_build(lv[-5 * nl:], threads=8))
        cf = oracle.fill_random(3 * nc * polys, 5200 + g)
        assert np.array_equal(evals[g][: 3 * order].cpu().numpy().view(np.uint64), oracle.coset_evaluate(cf[: 3 * nc], off, order, width=3))


def test_multi_device_errors_and_current_device(tf, oracle):
    """argument errors as the single-device calls; a device index out of range is TF_ERR_NO_DEVICE and names the index; the
    caller's current device is untouched"""
    assert tf.get_device() == 0
    with pytest.raises(tf.NttPanic):
        tf.ntt(np.zeros(12, dtype=np.uint64), devices=[0, 0])
    with pytest.raises(tf.MerkleTreeError):
        tf.MerkleTree.build_batch(np.zeros(30, dtype=np.uint64), 3, devices=[0, 0])
    with pytest.raises(tf.NttPanic):  # order below the coefficient count
        tf.fast_coset_evaluate(oracle.fill_random(64, 1), oracle.bfe_new(7), 16, batch=2, devices=[0, 0])
    bad = tf.lib().tf_device_count()
    with pytest.raises(tf.TwentyFirstError) as e:
        tf.ntt(np.zeros(16, dtype=np.uint64), batch=2, devices=[0, bad])
    assert e.value.code == 8 and str(bad) in tf.lib().tf_last_error().decode()
    with pytest.raises(tf.TwentyFirstError):
        tf.set_device(bad)
    tf.set_device(0)
    assert tf.get_device() == 0


def test_multi_device_matches_single_device_at_size(tf, oracle):
    """64 transforms of 2^16 points over four workers: the words of the one-call result"""
    n, batch = 1 << 16, 64
    x = oracle.fill_random(n * batch, 3500)
    a, b = x.copy(), x.copy()
    tf.ntt(a, batch=batch)
    tf.ntt(b, batch=batch, devices=[0, 0, 0, 0])
    assert np.array_equal(a, b)
    assert np.array_equal(a[:n], oracle.ntt(x[:n]))
