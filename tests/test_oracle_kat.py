"""Pin the CPU oracle against the reference's own known-answer tests.

Every vector below is *data* copied from a test in /root/reference/twenty-first/src (file:line in
each docstring); the constant tables come from tests/golden/reference_tables.json.  These tests
run without a GPU.
"""
import json
import os
import random

import numpy as np
import pytest

from tests import pyref

P = pyref.P
MAX = P - 1
HERE = os.path.dirname(os.path.abspath(__file__))
TABLES = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))


def new(o, v):
    return o.bfe_new(v)


# ------------------------------------------------------------------ BFieldElement

def test_fixed_mul(oracle):
    """math/b_field_element.rs:1498-1514 (test_fixed_mul)"""
    o = oracle
    assert o.bfe_mul(new(o, 2779336007265862836), new(o, 8146517303801474933)) == new(o, 1857758653037316764)
    assert o.bfe_mul(new(o, 9223372036854775808), new(o, 9223372036854775808)) == new(o, 18446744068340842497)


def test_fixed_inverse(oracle):
    """math/b_field_element.rs:1479-1487"""
    o = oracle
    assert o.bfe_inverse(new(o, 8561862112314395584)) == new(o, 17307602810081694772)
    assert o.bfe_inverse(0) == 0  # inverse_or_zero, traits.rs:39-45


def test_fixed_modpow(oracle):
    """math/b_field_element.rs:1490-1495"""
    o = oracle
    assert o.bfe_mod_pow(new(o, 7808276826625786800), 16608971246357572739) == new(o, 2288673415394035783)


def test_mod_pow(oracle):
    """math/b_field_element.rs:1358-1371"""
    o = oracle
    one = new(o, 1)
    assert one == 0xFFFFFFFF  # ONE.raw, b_field_element.rs:707-709
    assert o.bfe_mod_pow(new(o, 281474976710656), 4) == one
    assert o.bfe_mod_pow(new(o, 281474976710656), 5) == new(o, 281474976710656)
    assert o.bfe_mod_pow(new(o, 18446744069414584320), 2) == one
    assert o.bfe_mod_pow(new(o, 18446744069397807105), 8) == one
    assert o.bfe_mod_pow(new(o, 2625919085333925275), 10) == one
    assert o.bfe_mod_pow(new(o, 281474976645120), 12) == one
    assert o.bfe_mod_pow(new(o, 0), 0) == one


def test_primitive_roots(oracle):
    """math/b_field_element.rs:43-78 table and :1374-1386 (get_primitive_root_of_unity_test)"""
    o = oracle
    one = new(o, 1)
    for i in range(1, 33):
        n = 1 << i
        root = o.primitive_root(n)
        assert root == new(o, int(TABLES["primitive_roots"][str(n)]))
        assert o.bfe_mod_pow(root, n) == one
        assert o.bfe_mod_pow(root, n // 2) != one
        # omega_n = 7^((p-1)/n)
        assert o.bfe_value(root) == pyref.root_of_unity(n)
    assert o.primitive_root(0) == one and o.primitive_root(1) == one
    assert o.primitive_root(3) == 0


def test_add_sub_wrap_around_and_neg(oracle):
    """math/b_field_element.rs:1270-1280 and :1283-1294"""
    o = oracle
    four = new(o, 4)
    s = o.bfe_add(new(o, MAX), four)
    assert s == new(o, 3)
    assert o.bfe_sub(s, four) == new(o, MAX)
    assert o.bfe_neg(0) == 0
    assert o.bfe_value(o.bfe_neg(new(o, 1))) == MAX
    mx = new(o, MAX)
    mp1 = o.bfe_add(mx, new(o, 1))
    mp2 = o.bfe_add(mp1, new(o, 1))
    assert o.bfe_neg(mp1) == 0
    assert o.bfe_neg(mp2) == mx


def test_constants(oracle):
    """R2 (b_field_element.rs:229), MINUS_TWO_INVERSE (:232, test :1634-1636), montyred (:357-370)"""
    o = oracle
    assert o.bfe_inverse(new(o, P - 2)) == new(o, 0x7FFFFFFF80000000)
    assert o.bfe_value(new(o, 12345)) == 12345
    assert o.lib().tfo_montyred(0xFFFFFFFE00000001, 0) == 0xFFFFFFFF  # montyred(R2) = R mod p = ONE.raw
    rng = random.Random(1)
    for _ in range(200):
        a, b = rng.randrange(P), rng.randrange(P)
        ra, rb = new(o, a), new(o, b)
        assert ra == pyref.to_raw(a)
        assert o.bfe_value(o.bfe_mul(ra, rb)) == a * b % P
        assert o.bfe_value(o.bfe_add(ra, rb)) == (a + b) % P
        assert o.bfe_value(o.bfe_sub(ra, rb)) == (a - b) % P
        if a:
            assert o.bfe_value(o.bfe_inverse(ra)) == pow(a, P - 2, P)


def test_degenerate_add(oracle):
    """tip5/mod.rs:1098-1142: Add with a degenerate (>= p) lhs and a small enough rhs is canonical."""
    o = oracle
    rng = random.Random(2)
    for _ in range(200):
        a = rng.randrange(P, 1 << 64)
        b = rng.randrange(0, P + 2 - (1 << 32))
        assert o.bfe_add(a, b) < P
    for c in TABLES["round_constants"]:
        assert new(o, int(c)) < P + 2 - (1 << 32)


# ------------------------------------------------------------------ XFieldElement

def xfe(o, vals):
    return [new(o, v) for v in vals]


def test_x_field_mul(oracle):
    """math/x_field_element.rs:924-964 (x_field_mul_test)"""
    o = oracle
    cases = [
        ([2, 0, 0], [3, 0, 0], [6, 0, 0]),
        ([0, 3, 0], [0, 3, 0], [0, 0, 9]),
        ([125, 0, 0], [0, 0, 5], [0, 0, 625]),
        ([0, 0, 1], [0, 0, 1], [0, MAX, 1]),
        ([0, 1, 0], [0, 0, 1], [MAX, 1, 0]),
        ([13, 2, 3], [19, 0, 5], [237, 33, 137]),
    ]
    for a, b, c in cases:
        assert list(o.xfe_mul(xfe(o, a), xfe(o, b))) == xfe(o, c)


def test_x_field_add_sub(oracle):
    """math/x_field_element.rs:858-921"""
    o = oracle
    add_cases = [
        ([2, 0, 0], [3, 0, 0], [5, 0, 0]),
        ([0, 5, 0], [0, 7, 0], [0, 12, 0]),
        ([0, 0, 14], [0, 0, 23], [0, 0, 37]),
        ([0, 0, MAX], [0, 0, 23], [0, 0, 22]),
        ([MAX - 2, 12, 4], [2, 45000, MAX - 3], [MAX, 45012, 0]),
    ]
    for a, b, c in add_cases:
        assert list(o.xfe_add(xfe(o, a), xfe(o, b))) == xfe(o, c)
    sub_cases = [  # (minuend, subtrahend, diff)
        ([3, 0, 0], [2, 0, 0], [1, 0, 0]),
        ([0, 7, 0], [0, 5, 0], [0, 2, 0]),
        ([0, 0, 23], [0, 0, 14], [0, 0, 9]),
        ([0, 0, 23], [0, 0, MAX], [0, 0, 24]),
        ([2, 45000, MAX - 3], [MAX - 2, 12, 4], [5, 44988, MAX - 7]),
    ]
    for a, b, c in sub_cases:
        assert list(o.xfe_sub(xfe(o, a), xfe(o, b))) == xfe(o, c)


def test_xfe_mod_pow_static(oracle):
    """math/x_field_element.rs:1251-1256 : 3^n lifted into the extension field"""
    o = oracle
    acc = xfe(o, [1, 0, 0])
    three = xfe(o, [3, 0, 0])
    for expected in [1, 3, 9, 27, 81, 243]:
        assert list(acc) == xfe(o, [expected, 0, 0])
        acc = o.xfe_mul(acc, three)


# ------------------------------------------------------------------ NTT

def test_bfield_basic_ntt(oracle):
    """math/ntt.rs:424-445 (bfield_basic_test_of_chu_ntt)"""
    o = oracle
    x = o.to_raw([1, 4, 0, 0])
    y = o.ntt(x)
    assert list(y) == list(o.to_raw([5, 1125899906842625, 18446744069414584318, 18445618169507741698]))
    assert list(o.intt(y)) == list(x)


def test_bfield_max_value_ntt(oracle):
    """math/ntt.rs:448-469"""
    o = oracle
    x = o.to_raw([MAX, 0, 0, 0])
    y = o.ntt(x)
    assert list(y) == list(o.to_raw([MAX] * 4))
    assert list(o.intt(y)) == list(x)


def test_xfield_basic_ntt(oracle):
    """math/ntt.rs:398-421"""
    o = oracle
    x = o.to_raw([1, 0, 0] + [0, 0, 0] * 3)
    y = o.ntt(x, width=3)
    assert list(y) == list(o.to_raw([1, 0, 0] * 4))
    assert list(o.intt(y, width=3)) == list(x)


def test_ntt_length_32(oracle):
    """math/ntt.rs:512-560 (b_field_ntt_with_length_32)"""
    o = oracle
    x = o.to_raw([1, 4, 0, 0, 0, 0, 0, 0] * 4)
    expected = [0] * 32
    for i, v in enumerate(
        [20, 18446744069146148869, 4503599627370500, 18446726477228544005, 18446744069414584309, 268435460,
         18442240469787213829, 17592186040324]
    ):
        expected[4 * i] = v
    y = o.ntt(x)
    assert list(y) == list(o.to_raw(expected))
    assert list(o.intt(y)) == list(x)


@pytest.mark.parametrize("log_n", range(0, 9))
def test_ntt_matches_naive_dft_and_evaluation(oracle, log_n):
    """math/ntt.rs:563-579 (test_compare_ntt_to_eval) + independent python-int DFT"""
    o = oracle
    n = 1 << log_n
    rng = random.Random(100 + log_n)
    vals = [rng.randrange(P) for _ in range(n)]
    x = o.to_raw(vals)
    y = o.ntt(x)
    assert [int(v) for v in o.to_values(y)] == pyref.dft(vals)
    if n > 1:
        omega = o.primitive_root(n)
        for i in [0, 1, n // 2, n - 1]:
            assert int(o.poly_eval(x, o.bfe_mod_pow(omega, i))[0]) == int(y[i])
    back = o.intt(y)
    assert list(back) == list(x)
    assert [int(v) for v in o.to_values(o.intt(x))] == pyref.dft(vals, inverse=True)


@pytest.mark.parametrize("log_n", [1, 3, 6])
def test_xfe_ntt_is_three_bfe_columns(oracle, log_n):
    """math/x_field_element.rs:1271-1286 and SURVEY 7a: XFE ntt == 3 interleaved BFE transforms"""
    o = oracle
    n = 1 << log_n
    x = o.fill_random(3 * n, 7 + log_n)
    y = o.ntt(x, width=3)
    for c in range(3):
        assert list(y[c::3]) == list(o.ntt(x[c::3].copy()))
    assert list(o.intt(y, width=3)) == list(x)


def test_ntt_bad_lengths(oracle):
    """math/ntt.rs:135-140 : panics unless len is 0 or a power of two"""
    o = oracle
    assert o.ntt(np.zeros(0, np.uint64)).size == 0
    one = o.fill_random(1, 3)
    assert list(o.ntt(one)) == list(one) and list(o.intt(one)) == list(one)
    for n in [3, 5, 6, 12, 1000]:
        with pytest.raises(o.OraclePanic):
            o.ntt_raw_len(np.zeros(n, np.uint64), n)


def test_ntt_batch_threads(oracle):
    o = oracle
    x = o.fill_random(8 * 256, 11)
    a = o.ntt(x, batch=8)
    b = o.ntt(x, batch=8, threads=4)
    assert (a == b).all()
    for i in range(8):
        assert (a[i * 256:(i + 1) * 256] == o.ntt(x[i * 256:(i + 1) * 256])).all()


# ------------------------------------------------------------------ Polynomial

@pytest.mark.parametrize("width", [1, 3])
def test_coset_evaluate_matches_horner(oracle, width):
    """math/polynomial.rs:3646-3662 : fast_coset_evaluate == evaluation on {offset * w^i}"""
    o = oracle
    n_coeffs, order = 11, 16
    c = o.fill_random(n_coeffs * width, 21)
    offset = new(o, 7)  # benches/polynomial_coset.rs:20
    ev = o.coset_evaluate(c, offset, order, width=width).reshape(order, width)
    omega = o.primitive_root(order)
    for i in range(order):
        point = o.bfe_mul(offset, o.bfe_mod_pow(omega, i))
        assert list(o.poly_eval(c, point, width=width)) == list(ev[i])
    back = o.coset_interpolate(ev.reshape(-1), offset, width=width)
    assert list(back[: n_coeffs * width]) == list(c)
    assert not back[n_coeffs * width:].any()


def test_coset_evaluate_order_check(oracle):
    """math/polynomial.rs:1388-1392 : panics unless order > degree (leading zeros do not count)"""
    o = oracle
    c = o.fill_random(9, 5)
    with pytest.raises(o.OraclePanic):
        o.coset_evaluate(c, new(o, 7), 8)
    c[8] = 0  # degree 7 now
    o.coset_evaluate(c, new(o, 7), 8)
    with pytest.raises(o.OraclePanic):
        o.coset_evaluate(c[:4], new(o, 7), 6)  # order not a power of two


# ------------------------------------------------------------------ Tip5

def test_lookup_table_and_round_constants(oracle):
    """tip5/mod.rs:1035-1053 (lookup_table_is_correct), :50-64, :1022-1026"""
    assert TABLES["lookup_table"] == pyref.LOOKUP
    assert TABLES["mds_matrix_first_column"] == pyref.MDS_COL


def test_hash10_snapshot(oracle):
    """tip5/mod.rs:1294-1306 (hash10_test_vectors_snapshot)"""
    o = oracle
    pre = np.zeros(10, dtype=np.uint64)
    for i in range(6):
        d = o.hash_10(pre)
        pre[i:i + 5] = d
    assert o.digest_hex(o.hash_10(pre)) == (
        "109cc2fe453bd9962f754b96d8f5b919b60af030940a275f5540da195fef65ee651c1b6fa19b2c6a"
    )


def test_hash_varlen_vectors(oracle):
    """tip5/mod.rs:1309-1325 (hash_varlen_test_vectors)"""
    o = oracle
    acc = [0] * 5
    for i in range(20):
        d = o.hash_varlen(o.to_raw(list(range(i))) if i else np.zeros(0, np.uint64))
        acc = [o.bfe_add(a, int(b)) for a, b in zip(acc, d)]
    assert o.digest_hex(acc) == "efbafa86622a9c69652f8a1c4ffd734f021ad23a0a8085412a877de0f9170b18ea4ff69b6fff9a03"


SNAPSHOT_STATE = [
    0x0000000FFFFFFFF0, 0x00000000FFFFFFFF, 0x00000000FFFFFFFF, 0x00000028FFFFFFD7,
    0x00000006FFFFFFF9, 0x00000002FFFFFFFD, 0x00000000FFFFFFFF, 0x00000030FFFFFFCF,
    0x00000397FFFFFC68, 0x0000000FFFFFFFF0, 0x316BFB7236382123, 0x216F521B66EF83F5,
    0x5689D7B363F52DF0, 0xEB2F59E3AEAE25FC, 0xB08299D277CBB4DC, 0xCBE3D9FDC5349140,
]
SNAPSHOT_OUT = [0x15D38EA929F6632A, 0xF988E509FF738BB4, 0x48BCDFAE88A2E9F3, 0x87339E832DAAC02A, 0x511E41268150FDAC]


def test_permutation_snapshot(oracle):
    """tip5/mod.rs:1328-1362 (snapshot): raw Montgomery words in and out"""
    o = oracle
    out = o.tip5_permutation(np.array(SNAPSHOT_STATE, dtype=np.uint64))
    assert [int(v) for v in out[:5]] == SNAPSHOT_OUT
    out_naive = o.tip5_permutation(np.array(SNAPSHOT_STATE, dtype=np.uint64), naive=True)
    assert (out == out_naive).all()


DEGENERATE_IN = [
    0x1063C4BF5D8BB0DD, 0xDB6275D371FE05D0, 0xDE58CAE30144CDAE, 0xC774E64681D3622E,
    0xC4A947D10A5AA466, 0xDA5577A00A913151, 0xE80E978B3836DCD0, 0x8DD161F0A3AC00C2,
    0x6857F251A9C0F693, 0x4923A3683046178E, 0x6E6FC54A9B81010B, 0xCB84FA5BB9FAEC36,
    0x93CBF9DB4C5CB1EA, 0xF215D9B92DC87266, 0x88F09783D2AE3C57, 0x6D29F9CE94A90B71,
]
DEGENERATE_OUT = [
    0xA5D32D629E60D72E, 0x5516EF90D2773D74, 0x65D3FA1CDE45F6CB, 0x7BF0E725DFA5906B,
    0x67A2DB4B141B90E9, 0x91DB162D32309083, 0xEFEC1D00146A05C9, 0xCCA0D6566BCA8186,
    0x405BAEB5B3F87F02, 0xD897015870278F76, 0xD4B2EE4810AAC7D1, 0x27B451E706A5C2FC,
    0xE9B4177F0A0EFFE4, 0x0C60DEF0F2C5287F, 0x703AA06D327CCC34, 0x536F23550EBF98F1,
]


def test_tip5_recovers_from_degenerate(oracle):
    """tip5/mod.rs:1146-1206 (values are canonical, passed through BFieldElement::new)"""
    o = oracle
    st = o.to_raw(DEGENERATE_IN)
    out = o.tip5_permutation(st)
    assert list(out) == list(o.to_raw(DEGENERATE_OUT))
    assert list(o.tip5_permutation(st, naive=True)) == list(out)


def test_hasher_trait_snapshot(oracle):
    """tip5/mod.rs:1526-1531 + Hasher::write :706-720: absorb b"hello world" without padding"""
    o = oracle
    data = b"hello world"
    elems = [int.from_bytes(data[i:i + 8], "little") for i in range(0, len(data), 8)]
    buf = o.to_raw(elems + [0] * (10 - len(elems)))
    st = o.absorb(np.zeros(16, dtype=np.uint64), buf)
    assert o.bfe_value(int(st[0])) == 2267905471610932299


def test_bag_peaks_empty_snapshot(oracle):
    """util_types/mmr/mmr_accumulator.rs:1038-1046 (empty case) = hash_10([0; 10]) (:379-391)"""
    o = oracle
    assert o.digest_hex(o.hash_10(np.zeros(10, dtype=np.uint64))) == (
        "cd65052100640f0d27e5654f97c47e49899add2f265967ccbefee7264e9bc08f588542d9dc3d5ac5"
    )


def test_tip5_vs_python_math(oracle):
    """tip5/naive.rs:93-105 style differential test against the pure-python restatement."""
    o = oracle
    rcs = [int(c) for c in TABLES["round_constants"]]
    rng = random.Random(9)
    for _ in range(10):
        vals = [rng.randrange(P) for _ in range(16)]
        expect = pyref.tip5_permutation(vals, rcs)
        got = o.tip5_permutation(o.to_raw(vals))
        assert [int(v) for v in o.to_values(got)] == expect
        assert all(int(v) < P for v in got)


def test_hash_pair_is_hash_10(oracle):
    """tip5/mod.rs:577-586 vs :559-569"""
    o = oracle
    x = o.fill_random(10, 77)
    assert list(o.hash_pair(x[:5], x[5:])) == list(o.hash_10(x))
    assert list(o.hash_pairs(np.concatenate([x, x]))) == list(o.hash_10(x)) * 2


def test_hash_varlen_padding(oracle):
    """util_types/sponge.rs:41-55: pad 1,0,...; a full last chunk still gets a padding block"""
    o = oracle
    x = o.fill_random(25, 5)
    one = new(o, 1)
    for ln in [0, 1, 9, 10, 11, 19, 20, 25]:
        st = np.zeros(16, dtype=np.uint64)
        full = ln // 10
        for c in range(full):
            st = o.absorb(st, x[10 * c:10 * c + 10])
        last = np.zeros(10, dtype=np.uint64)
        rem = ln - 10 * full
        last[:rem] = x[10 * full:ln]
        last[rem] = one
        st = o.absorb(st, last)
        assert list(o.hash_varlen(x[:ln])) == list(st[:5])
    rows = o.fill_random(6 * 13, 8)
    got = o.hash_varlen_rows(rows, 13).reshape(6, 5)
    for i in range(6):
        assert list(got[i]) == list(o.hash_varlen(rows[13 * i:13 * i + 13]))


# ------------------------------------------------------------------ MerkleTree

def tree_leaves(o, height):
    """util_types/merkle_tree.rs:980-987 (test_tree_of_height): leaf i = hash_varlen([i])"""
    return np.concatenate([o.hash_varlen(o.to_raw([i])) for i in range(1 << height)])


@pytest.mark.parametrize("height", range(0, 8))
def test_merkle_structure(oracle, height):
    """util_types/merkle_tree.rs:85-88,:149-222,:393-429: heap layout, nodes[0] dummy, root = nodes[1];
    sequential == parallel for all cutoffs (:1059-1128); frugal root == full root (:1089-1116)"""
    o = oracle
    leaves = tree_leaves(o, height)
    n = 1 << height
    nodes = o.merkle_build(leaves).reshape(2 * n, 5)
    assert not nodes[0].any()
    assert (nodes[n:].reshape(-1) == leaves).all()
    for i in range(1, n):
        assert list(nodes[i]) == list(o.hash_pair(nodes[2 * i], nodes[2 * i + 1]))
    for threads, cutoff in [(1, 2), (2, 2), (4, 4), (8, 16), (3, 2), (8, 512)]:
        assert (o.merkle_build(leaves, threads=threads, cutoff=cutoff).reshape(2 * n, 5) == nodes).all()
    assert list(o.merkle_frugal_root(leaves)) == list(nodes[1])


def test_merkle_errors(oracle):
    """util_types/merkle_tree.rs:393-410,:933-965"""
    o = oracle
    with pytest.raises(o.OraclePanic) as e:
        o.merkle_build(np.zeros(0, np.uint64))
    assert e.value.code == 1  # TooFewLeafs
    for n in [3, 5, 6, 7, 12]:
        with pytest.raises(o.OraclePanic) as e:
            o.merkle_build(o.fill_random(5 * n, 1))
        assert e.value.code == 2  # IncorrectNumberOfLeafs
        with pytest.raises(o.OraclePanic) as e:
            o.merkle_frugal_root(o.fill_random(5 * n, 1))
        assert e.value.code == 2


def test_merkle_root_goldens(oracle):
    """No literal Merkle root exists in the reference; these goldens were generated by
    this KAT-pinned oracle (tests/golden/make_merkle_goldens.py) and guard against regressions."""
    o = oracle
    path = os.path.join(HERE, "golden", "merkle_roots.json")
    gold = json.load(open(path))
    for h, hexroot in gold["test_tree_of_height_roots"].items():
        leaves = tree_leaves(o, int(h))
        nodes = o.merkle_build(leaves).reshape(-1, 5)
        assert o.digest_hex(nodes[1]) == hexroot


def test_mds_matrix_mul_methods_agree(oracle):
    """tip5/mod.rs:1509-1523 (test_mds_matrix_mul_methods_agree): mds_cyclomul == mds_generated on arbitrary states -- restated
    over the oracle's three MDS forms (plain circulant sum, the cyclomul recursion :753-1019, and the deferred-halving wrapping-u64
    graph that generated_function :256-506 unrolls), on 10^5 random canonical states AND on degenerate words (>= p, all-ones,
    0xffffffff halves): the raw output words must agree bit for bit, degenerate results included."""
    o = oracle
    rng = np.random.default_rng(20260928)
    n = 100_000
    states = rng.integers(0, P, size=(n, 16), dtype=np.uint64)
    special = np.array([0, 1, 0xFFFFFFFF, 0x100000000, P - 1, P, P + 1, 2 ** 64 - 1, 0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF,
                        0x00000000FFFFFFFF, 0x8000000080000000], dtype=np.uint64)
    # degenerate rows: every word drawn from the special set / fully random 64-bit words (not canonical)
    states[: n // 10] = special[rng.integers(0, special.size, size=(n // 10, 16))]
    states[n // 10: n // 5] = rng.integers(0, 2 ** 64, size=(n // 10, 16), dtype=np.uint64)
    for i in range(n):
        a, b, c = o.tip5_mds(states[i], 0), o.tip5_mds(states[i], 1), o.tip5_mds(states[i], 2)
        assert (a == b).all() and (a == c).all(), (i, states[i])
    # the constants the reference's unrolled graph hard-codes for the fully split components (node_64 / node_67, :283-284)
    assert o.tip5_mds_graph_constants() == (524757, 52427)


def test_mds_degenerate_words_are_reproduced(oracle):
    """tip5/mod.rs:217-242: the reduction at the end of mds_generated can leave a degenerate word (>= p) -- the reference's own
    example is s_hi = 0, s_lo = P -- which only the round-constant addition repairs.  A state [x, 0, ..., 0] with
    1108 * x in [p, 2^64) puts output 1 (= M[1] * x, an integer below 2^64, so s_hi = 0) exactly there: all three MDS forms must
    hand back that degenerate word unchanged, and the full permutation must still agree with the naive field version."""
    o = oracle
    M = [61402, 1108, 28750, 33823, 7454, 43244, 53865, 12034, 56951, 27521, 41351, 40901, 12021, 59689, 26798, 17845]
    x = -(-P // 1108)  # ceil(p / 1108)
    for delta in (0, 1, 5):
        st = np.zeros(16, dtype=np.uint64)
        st[0] = x + delta
        outs = [o.tip5_mds(st, m) for m in (0, 1, 2)]
        assert (outs[0] == outs[1]).all() and (outs[0] == outs[2]).all()
        assert int(outs[2][1]) == 1108 * (x + delta) and int(outs[2][1]) >= P   # degenerate, not reduced
        for r in range(16):
            assert int(outs[2][r]) % P == (M[r] * (x + delta)) % P
    st = np.zeros(16, dtype=np.uint64)
    st[0] = x
    assert (o.tip5_permutation(st) == o.tip5_permutation(st, naive=True)).all()


def test_mds_linearity_and_circulancy(oracle):
    """tip5/mod.rs:1391-1455 (test_linearity_of_mds, test_mds_circulancy) on the MDS form the permutation runs: as field maps,
    mds(a u + b v) = a mds(u) + b mds(v), and the image of the unit vector e_0 is the first column."""
    o = oracle
    rng = np.random.default_rng(5)
    M = [61402, 1108, 28750, 33823, 7454, 43244, 53865, 12034, 56951, 27521, 41351, 40901, 12021, 59689, 26798, 17845]  # tip5/mod.rs:154-157
    e0 = np.zeros(16, dtype=np.uint64)
    e0[0] = o.bfe_new(1)
    col = o.tip5_mds(e0, 2)
    assert [o.bfe_value(int(v) % P) for v in col] == M
    for _ in range(200):
        u = rng.integers(0, P, size=16, dtype=np.uint64)
        v = rng.integers(0, P, size=16, dtype=np.uint64)
        a, b = int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64))
        w = np.array([o.bfe_add(o.bfe_mul(a, int(x)), o.bfe_mul(b, int(y))) for x, y in zip(u, v)], dtype=np.uint64)
        mu, mv, mw = o.tip5_mds(u, 2), o.tip5_mds(v, 2), o.tip5_mds(w, 2)
        for k in range(16):
            assert int(mw[k]) % P == o.bfe_add(o.bfe_mul(a, int(mu[k]) % P), o.bfe_mul(b, int(mv[k]) % P))


def test_zerofier_and_interpolation_doc_examples(oracle):
    """Polynomial::zerofier doc example (math/polynomial.rs:1426-1434) and Polynomial::interpolate doc example (:1490-1497)
    through the oracle's smart_zerofier (:1462-1475) / lagrange_interpolate (:1565-1606); interpolation then evaluation is the
    identity (:3602-3612), one point gives the constant (:3562-3570), repeated points panic (:3554-3560)."""
    tfo = oracle
    P = 0xFFFFFFFF00000001
    z = tfo.zerofier(tfo.to_raw([2, 4, 6]))
    assert [int(v) for v in tfo.to_values(z)] == [(-48) % P, 44, (-12) % P, 1]
    for root in (2, 4, 6):
        assert int(tfo.poly_eval(z, tfo.bfe_new(root))[0]) == 0
    for other in (0, 1, 3, 5):
        assert int(tfo.poly_eval(z, tfo.bfe_new(other))[0]) != 0
    assert [int(v) for v in tfo.to_values(tfo.zerofier(np.zeros(0, dtype=np.uint64)))] == [1]
    f = tfo.lagrange_interpolate(tfo.to_raw([0, 1, 2, 3]), tfo.to_raw([1, 3, 5, 7]))
    assert [int(v) for v in tfo.to_values(f)] == [1, 2, 0, 0]
    assert int(tfo.to_values(tfo.poly_eval(f, tfo.bfe_new(4)))[0]) == 9
    assert [int(v) for v in tfo.to_values(tfo.lagrange_interpolate(tfo.to_raw([5]), tfo.to_raw([42])))] == [42]
    with pytest.raises(tfo.OraclePanic):
        tfo.lagrange_interpolate(tfo.to_raw([1, 1]), tfo.to_raw([1, 2]))
    for width in (1, 3):
        n = 23
        d, v = tfo.fill_random(n * width, 5), tfo.fill_random(n * width, 6)
        c = tfo.lagrange_interpolate(d, v, width)
        zz = tfo.zerofier(d, width)
        for i in range(n):
            if width == 1:
                assert int(tfo.poly_eval(c, int(d[i]))[0]) == int(v[i])
                assert int(tfo.poly_eval(zz, int(d[i]))[0]) == 0
            else:
                assert np.array_equal(tfo.poly_eval_xfe_point(c, d[3 * i: 3 * i + 3]), v[3 * i: 3 * i + 3])
                assert not tfo.poly_eval_xfe_point(zz, d[3 * i: 3 * i + 3]).any()


def test_xfe_inverse_is_the_inverse(oracle):
    """XFieldElement::inverse (x_field_element.rs:371-379): a * a^-1 = 1 (tests :1294-1300), lifted base-field elements invert
    like BFieldElement::inverse, zero panics."""
    tfo = oracle
    one = np.array([tfo.bfe_new(1), 0, 0], dtype=np.uint64)
    for seed in range(20):
        a = tfo.fill_random(3, 900 + seed)
        assert np.array_equal(tfo.xfe_mul(a, tfo.xfe_inverse(a)), one)
    b = np.array([tfo.bfe_new(12345), 0, 0], dtype=np.uint64)
    assert int(tfo.xfe_inverse(b)[0]) == tfo.bfe_inverse(tfo.bfe_new(12345)) and not tfo.xfe_inverse(b)[1:].any()
    for sparse in ([0, 1, 0], [0, 0, 1], [1, 0, 1]):
        a = tfo.to_raw(sparse)
        assert np.array_equal(tfo.xfe_mul(a, tfo.xfe_inverse(a)), one)
    with pytest.raises(tfo.OraclePanic):
        tfo.xfe_inverse(np.zeros(3, dtype=np.uint64))


def test_division_restatements(oracle):
    """Polynomial::naive_divide (math/polynomial.rs:552-600) and clean_divide (:2358-2411) in the oracle: a = q b + r with
    deg r < deg b (:3789-3803 style), clean_divide == divide's quotient on clean divisions through both of its routes (:3707-3719),
    divisors with zero as a (multiple) root (:3722-3772), and the panics."""
    tfo = oracle
    for na, nb in ((100, 13), (13, 100), (64, 64), (1, 1), (300, 299)):
        a, b = tfo.fill_random(na, 10 + na), tfo.fill_random(nb, 11 + nb)
        q, r = tfo.naive_divide(a, b)
        assert r.size < nb
        back = tfo.poly_mul(q, b) if q.size else np.zeros(0, dtype=np.uint64)
        back = np.concatenate([back, np.zeros(max(na - back.size, 0), dtype=np.uint64)])
        for i in range(r.size):
            back[i] = tfo.bfe_add(int(back[i]), int(r[i]))
        assert np.array_equal(back[:na], a) and not back[na:].any()
    # (x + 1)(x + 2)(x + 3) / (x + 2) = x^2 + 4 x + 3
    prod = tfo.poly_mul(tfo.poly_mul(tfo.to_raw([1, 1]), tfo.to_raw([2, 1])), tfo.to_raw([3, 1]))
    for cutoff in (0, 1 << 9):
        assert [int(v) for v in tfo.to_values(tfo.clean_divide(prod, tfo.to_raw([2, 1]), cutoff))] == [3, 4, 1]
    for nq, nb in ((5, 3), (100, 40), (700, 600), (3, 1000)):
        q, b = tfo.fill_random(nq, 20 + nq), tfo.fill_random(nb, 21 + nb)
        b[0] = 0 if nb > 3 else b[0]
        a = tfo.poly_mul(q, b)
        assert np.array_equal(tfo.clean_divide(a, b, 0), q) and np.array_equal(tfo.clean_divide(a, b), q)
        assert np.array_equal(tfo.naive_divide(a, b)[0], q) and tfo.naive_divide(a, b)[1].size == 0
    with pytest.raises(tfo.OraclePanic):
        tfo.clean_divide(prod, np.zeros(2, dtype=np.uint64))
    with pytest.raises(tfo.OraclePanic):
        tfo.clean_divide(tfo.to_raw([1, 0, 1]), tfo.to_raw([1, 1]), 0)  # x^2 + 1 is not a multiple of x + 1


def test_tip5_trace_starts_with_the_state_and_ends_with_the_permutation(oracle):
    """tip5/mod.rs:1557-1565: trace()[0] is the initial state, trace()[5] the permuted one; every row is one round further
    (checked against the naive permutation's end state, tip5/naive.rs, which shares no code with the round function)."""
    for seed in range(5):
        s = oracle.fill_random(16, 4000 + seed)
        trace, end = oracle.tip5_trace(s)
        assert np.array_equal(trace[0], s)
        assert np.array_equal(trace[5], oracle.tip5_permutation(s)) and np.array_equal(end, trace[5])
        assert np.array_equal(trace[5], oracle.tip5_permutation(s, naive=True))
        assert len({trace[r].tobytes() for r in range(6)}) == 6


def test_barycentric_evaluation_equals_polynomial_evaluation(oracle):
    """math/polynomial.rs:4593-4616 (polynomial_evaluation_and_barycentric_evaluation_are_equivalent) and :4630-4638 (all four
    type combinations), through the oracle's restatement of barycentric_evaluate (:2609-2637)."""
    tfo = oracle
    for width in (1, 3):
        for log_n in (0, 1, 4, 7):
            n = 1 << log_n
            c = tfo.fill_random(n * width, 60 + log_n)
            cw = tfo.ntt(c, width=width)
            for x in (tfo.fill_random(3, 61), np.array([tfo.bfe_new(424242), 0, 0], dtype=np.uint64)):
                lifted = c
                if width == 1:
                    lifted = np.zeros(3 * n, dtype=np.uint64)
                    lifted[0::3] = c
                assert np.array_equal(tfo.barycentric_evaluate(cw, x, width), tfo.poly_eval_xfe_point(lifted, x))
    with pytest.raises(tfo.OraclePanic):
        tfo.barycentric_evaluate(tfo.fill_random(12, 1), tfo.fill_random(3, 2))      # not a power of two
    with pytest.raises(tfo.OraclePanic):
        tfo.barycentric_evaluate(tfo.fill_random(8, 1), np.array([tfo.bfe_new(1), 0, 0], dtype=np.uint64))  # 1 is in every subgroup


def _poly_golden_cases():
    return json.load(open(os.path.join(HERE, "golden", "poly_goldens.json")))["cases"]


def _poly_golden_eval(case, ops):
    """Runs one golden case through `ops` (the oracle here, the HIP path in tests/test_gpu_next_rows.py); returns raw words."""
    o, fr = case["op"], oracle_fill
    if o == "zerofier":
        return ops["zerofier"](fr(case["n"] * case["width"], case["seed"]), case["width"])
    if o == "interpolate":
        return ops["interpolate"](fr(case["n"] * case["width"], case["domain_seed"]), fr(case["n"] * case["width"], case["values_seed"]), case["width"])
    if o == "barycentric_evaluate":
        return ops["barycentric"](fr(case["n"] * case["width"], case["codeword_seed"]), fr(3, case["indeterminate_seed"]), case["width"])
    if o == "clean_divide":
        return ops["clean_divide"](ops["to_raw"](case["dividend"]), fr(case["nb"], case["divisor_seed"]))
    if o == "coset_evaluate_xfe_offset":
        return ops["coset_xoff"](fr(3 * case["n_coeffs"], case["coeffs_seed"]), fr(3, case["offset_seed"]), case["order"])
    if o == "tip5_trace":
        return ops["trace"](fr(16, case["state_seed"]))
    raise AssertionError(o)


oracle_fill = None


def test_poly_goldens_reproduced_by_the_oracle(oracle):
    """tests/golden/poly_goldens.json (tests/golden/make_poly_goldens.py): the committed vectors of zerofier, interpolation, clean
    division, barycentric evaluation, the extension-field offset and Tip5::trace -- a change of the oracle shows up here."""
    global oracle_fill
    oracle_fill = oracle.fill_random
    ops = {"zerofier": oracle.zerofier, "interpolate": oracle.lagrange_interpolate, "barycentric": oracle.barycentric_evaluate,
           "clean_divide": lambda a, b: oracle.clean_divide(a, b, 0), "coset_xoff": oracle.coset_evaluate_xfe_offset,
           "trace": lambda s: oracle.tip5_trace(s)[0], "to_raw": oracle.to_raw}
    cases = _poly_golden_cases()
    assert len(cases) == 9
    for case in cases:
        got = np.asarray(_poly_golden_eval(case, ops), dtype=np.uint64).reshape(-1)
        assert [int(v) for v in oracle.to_values(got)] == case["out"], case["op"]


# ---- the independent fixture: tests/golden/poly_goldens_pyref.json, generated by tests/pyref.py alone (pure-Python schoolbook
# definitions; tests/golden/make_poly_goldens_pyref.py).  Inputs and outputs are explicit canonical values.
def pyref_golden_cases():
    return json.load(open(os.path.join(HERE, "golden", "poly_goldens_pyref.json")))["cases"]


def pyref_golden_eval(case, ops):
    """One case of the pyref fixture through `ops` (the oracle here, the HIP path in tests/test_gpu_next_rows.py).  `ops` functions
    take and return raw Montgomery words; returns the canonical values, padded with zeros to the fixture's length (the routes
    return trimmed polynomials)."""
    raw = ops["to_raw"]
    o, w = case["op"], case.get("width", 1)
    if o == "zerofier":
        got = ops["zerofier"](raw(case["roots"]), w)
    elif o == "interpolate":
        got = ops["interpolate"](raw(case["domain"]), raw(case["values"]), w)
    elif o == "barycentric_evaluate":
        got = ops["barycentric"](raw(case["codeword"]), raw(case["indeterminate"]), w)
    elif o == "coset_evaluate":
        got = ops["coset"](raw(case["coeffs"]), int(raw([case["offset"]])[0]), case["order"], w)
    elif o == "coset_evaluate_xfe_offset":
        got = ops["coset_xoff"](raw(case["coeffs"]), raw(case["offset"]), case["order"])
    elif o == "ntt":
        got = ops["ntt"](raw(case["in"]), w)
    elif o == "multiply":
        got = ops["multiply"](raw(case["a"]), raw(case["b"]), w)
    elif o == "clean_divide":
        got = ops["clean_divide"](raw(case["dividend"]), raw(case["divisor"]))
    else:
        raise AssertionError(o)
    vals = [int(v) for v in ops["to_values"](np.asarray(got, dtype=np.uint64).reshape(-1))]
    return vals + [0] * (len(case["out"]) - len(vals))


def test_pyref_splitmix_is_the_oracles_fill_random(oracle):

    for seed, first in ((0x7F210002, 0), (5, 1 << 20), (0xDEADBEEF, 12345)):
        want = [int(v) for v in oracle.to_values(oracle.fill_random(64, seed, first_index=first))]
        assert pyref.splitmix_values(64, seed, first_index=first) == want


def test_pyref_goldens_reproduced_by_the_oracle(oracle):
    """The oracle's fast routes (zerofier / interpolation through trees, NTT-based division and coset evaluation, the barycentric
    formula) against the pure-Python schoolbook definitions: 49 committed cases, BFieldElement and XFieldElement."""
    ops = {"to_raw": oracle.to_raw, "to_values": oracle.to_values, "zerofier": oracle.zerofier, "interpolate": oracle.lagrange_interpolate,
           "barycentric": oracle.barycentric_evaluate, "coset": lambda c, off, order, w: oracle.coset_evaluate(c, off, order, width=w),
           "coset_xoff": oracle.coset_evaluate_xfe_offset, "ntt": lambda x, w: oracle.ntt(x, width=w),
           "multiply": lambda a, b, w: oracle.poly_mul(a, b, width=w), "clean_divide": lambda a, b: oracle.clean_divide(a, b, 0)}
    cases = pyref_golden_cases()
    assert len(cases) == 49
    for case in cases:
        assert pyref_golden_eval(case, ops) == case["out"], (case["op"], case.get("width"), case.get("n"))


def test_pyref_golden_file_is_what_pyref_generates(tmp_path):
    """The committed fixture is exactly the generator's output (a hand edit or a stale file shows up here)."""
    import subprocess
    import sys

    want = open(os.path.join(HERE, "golden", "poly_goldens_pyref.json")).read()
    gen = os.path.join(HERE, "golden", "make_poly_goldens_pyref.py")
    src = open(gen).read().replace('os.path.join(HERE, "poly_goldens_pyref.json")', repr(str(tmp_path / "out.json")))
    script = tmp_path / "gen.py"
    script.write_text(src.replace("HERE = os.path.dirname(os.path.abspath(__file__))", f"HERE = {os.path.join(HERE, 'golden')!r}"))
    subprocess.check_call([sys.executable, str(script)])
    assert open(tmp_path / "out.json").read() == want
