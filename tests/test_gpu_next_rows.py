"""GPU parity for the "next" rows of the scope table: coset interpolation, Hadamard /
fast_multiply, low-degree extension, rows -> Merkle tree, authentication structures.  Bit-exact vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def _to_dev(a):
    import torch

    return torch.from_numpy(a.view(np.int64)).cuda()


def _to_host(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("log_n", [0, 1, 3, 5, 8, 10, 11, 14, 16, 20, 21])
def test_coset_interpolate_matches_oracle(tf, oracle, width, log_n):
    """math/polynomial.rs:1907-1918; and interpolate(evaluate(f)) == f (tests :3665-3680 style)"""
    n = 1 << log_n
    batch = 3 if log_n <= 14 else 1
    v = oracle.fill_random(n * width * batch, 400 + log_n)
    off = oracle.bfe_new(7)
    got = tf.fast_coset_interpolate(v, off, width=width, batch=batch).reshape(batch, -1)
    for b in range(batch):
        want = oracle.coset_interpolate(v[b * n * width:(b + 1) * n * width], off, width=width)
        assert np.array_equal(got[b], want)
    back = tf.fast_coset_evaluate(got.reshape(-1), off, n, width=width, batch=batch)
    assert np.array_equal(back, v)


def test_coset_interpolate_errors_and_polynomial_api(tf, oracle):
    v = oracle.fill_random(16, 1)
    with pytest.raises(tf.NttPanic) as e:
        tf.fast_coset_interpolate(v, 0)  # offset.inverse() of zero
    assert e.value.code == 12
    with pytest.raises(tf.NttPanic):
        tf.fast_coset_interpolate(v[:12], oracle.bfe_new(7))
    for off_val in [1, 2, 7, P - 1]:
        off = oracle.bfe_new(off_val)
        p = tf.Polynomial.fast_coset_interpolate(off, v)
        want = oracle.coset_interpolate(v, off)
        assert np.array_equal(p.coefficients, want[: p.coefficients.size]) and not want[p.coefficients.size:].any()


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("na,nb", [(1, 1), (2, 3), (7, 5), (16, 16), (17, 16), (100, 29), (300, 513), (4096, 4097)])
def test_fast_multiply_matches_oracle(tf, oracle, width, na, nb):
    """math/polynomial.rs:900-932 vs the literal restatement; small cases also vs schoolbook"""
    batch = 2 if na * nb < 100000 else 1
    a = oracle.fill_random(na * width * batch, 50 + na)
    b = oracle.fill_random(nb * width * batch, 60 + nb)
    got = tf.fast_multiply(a, b, width=width, batch=batch).reshape(batch, -1)
    for k in range(batch):
        ak, bk = a[k * na * width:(k + 1) * na * width], b[k * nb * width:(k + 1) * nb * width]
        assert np.array_equal(got[k], oracle.poly_mul(ak, bk, width=width))
        if na * nb <= 4096:
            assert np.array_equal(got[k], oracle.poly_mul(ak, bk, width=width, naive=True))


def test_polynomial_fast_multiply_api(tf, oracle):
    a = tf.Polynomial(np.concatenate([oracle.fill_random(9, 3), np.zeros(4, np.uint64)]))
    b = tf.Polynomial(oracle.fill_random(5, 4))
    prod = a.fast_multiply(b)
    assert prod.degree() == 12
    assert np.array_equal(prod.coefficients, oracle.poly_mul(a.coefficients, b.coefficients, naive=True))
    zero = tf.Polynomial(np.zeros(3, np.uint64))
    assert a.fast_multiply(zero).degree() == -1  # polynomial.rs:907-909


@pytest.mark.parametrize("width", [1, 3])
def test_hadamard_and_lde_device(tf, oracle, width):
    import torch

    n = 1 << 10
    a = oracle.fill_random(n * width, 11)
    b = oracle.fill_random(n * width, 12)
    da, db = _to_dev(a), _to_dev(b)
    out = torch.empty_like(da)
    tf.device.hadamard(da, db, out, width=width)
    torch.cuda.synchronize()
    assert np.array_equal(_to_host(out), oracle.hadamard(a, b, width=width))
    # low-degree extension: values on the order-n coset of offset 1 -> order-4n coset of offset 7
    m = 4 * n
    ext = torch.empty(m * width, dtype=torch.int64, device="cuda")
    one, seven = oracle.bfe_new(1), oracle.bfe_new(7)
    tf.device.lde(da, n, one, ext, m, seven, width=width)
    torch.cuda.synchronize()
    coeffs = oracle.coset_interpolate(a, one, width=width)
    assert np.array_equal(_to_host(ext), oracle.coset_evaluate(coeffs, seven, m, width=width))
    # device fast_multiply
    prod = torch.empty((2 * n - 1) * width, dtype=torch.int64, device="cuda")
    tf.device.poly_mul(da, n, db, n, prod, width=width)
    torch.cuda.synchronize()
    assert np.array_equal(_to_host(prod), oracle.poly_mul(a, b, width=width))


@pytest.mark.parametrize("n_rows,row_len", [(1, 3), (2, 0), (8, 1), (64, 10), (256, 23), (512, 9), (4096, 30)])
def test_merkle_from_rows(tf, oracle, n_rows, row_len):
    rows = oracle.fill_random(n_rows * row_len, 70 + n_rows)
    if row_len == 0:
        with_rows = np.zeros(0, dtype=np.uint64)
        nodes = np.empty(10 * n_rows, dtype=np.uint64)
        import ctypes as C

        rc = tf.lib().tf_merkle_from_rows(C.c_void_p(0), 0, n_rows, C.c_void_p(nodes.ctypes.data), 1)
        assert rc == 0
        leaves = np.concatenate([oracle.hash_varlen(with_rows)] * n_rows)
        assert np.array_equal(nodes, oracle.merkle_build(leaves))
        return
    tree = tf.MerkleTree.from_rows(rows, row_len)
    want = oracle.merkle_from_rows(rows, row_len).reshape(2 * n_rows, 5)
    assert np.array_equal(tree.nodes, want)
    with pytest.raises(tf.MerkleTreeError):
        tf.MerkleTree.from_rows(oracle.fill_random(3 * 4, 1), 4)


def test_authentication_structure(tf, oracle):
    """util_types/merkle_tree.rs:449-504, :604-622 (doc example: leafs 0 and 2 of 8 -> nodes [11, 9, 3])"""
    import random

    import torch

    assert list(tf.MerkleTree.authentication_structure_node_indices(8, [0, 2])) == [11, 9, 3]
    assert list(tf.MerkleTree.authentication_structure_node_indices(8, [])) == []
    with pytest.raises(tf.MerkleTreeError) as e:
        tf.MerkleTree.authentication_structure_node_indices(8, [8])
    assert e.value.variant == "LeafIndexInvalid"
    with pytest.raises(tf.MerkleTreeError) as e:
        tf.MerkleTree.authentication_structure_node_indices(12, [1])
    assert e.value.variant == "IncorrectNumberOfLeafs"
    rng = random.Random(5)
    for height in [0, 1, 4, 10]:
        n = 1 << height
        leaves = oracle.fill_random(5 * n, 900 + height)
        dn = torch.empty(10 * n, dtype=torch.int64, device="cuda")
        tf.device.merkle_build(_to_dev(leaves), n, dn)
        torch.cuda.synchronize()
        host_nodes = oracle.merkle_build(leaves).reshape(2 * n, 5)
        tree = tf.MerkleTree(host_nodes)
        for k in [0, 1, 3, 17]:
            idx = [rng.randrange(n) for _ in range(k)]
            want_idx = oracle.auth_structure_indices(n, idx)
            assert np.array_equal(tf.MerkleTree.authentication_structure_node_indices(n, idx), want_idx)
            got = tf.device.authentication_structure(dn, n, idx)
            assert np.array_equal(got, host_nodes[want_idx.astype(np.int64)])
            assert np.array_equal(tree.authentication_structure(idx), got)


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("na", [1, 2, 9, 64, 1000, 5000])
def test_fast_square_matches_oracle(tf, oracle, width, na):
    """math/polynomial.rs:780-798: square == product with itself"""
    a = oracle.fill_random(na * width * 2, 300 + na)
    got = tf.fast_square(a, width=width, batch=2).reshape(2, -1)
    for k in range(2):
        ak = a[k * na * width:(k + 1) * na * width]
        assert np.array_equal(got[k], oracle.poly_mul(ak, ak, width=width))
    p = tf.Polynomial(a[: na * width], width=width)
    assert np.array_equal(p.fast_square().coefficients, tf.Polynomial(oracle.poly_mul(p.coefficients, p.coefficients, width=width), width=width).coefficients)


@pytest.mark.parametrize("n_coeffs,n_points", [(0, 5), (1, 1), (7, 300), (100, 1000), (1025, 4097)])
def test_batch_evaluate_matches_horner(tf, oracle, n_coeffs, n_points):
    """math/polynomial.rs:1840-1878: values of f on an arbitrary domain, BFE and XFE"""
    c = oracle.fill_random(n_coeffs, 11 + n_coeffs)
    pts = oracle.fill_random(n_points, 13 + n_points)
    got = tf.Polynomial(c).batch_evaluate(pts) if n_coeffs else np.zeros(n_points, np.uint64)
    if n_coeffs:
        pc = tf.Polynomial(c).coefficients
        for i in list(range(min(n_points, 40))) + [n_points - 1]:
            assert int(got[i]) == int(oracle.poly_eval(pc, int(pts[i]))[0])
    cx = oracle.fill_random(3 * max(n_coeffs, 1), 17 + n_coeffs)
    px = oracle.fill_random(3 * n_points, 19 + n_points)
    gx = tf.Polynomial(cx, width=3).batch_evaluate(px).reshape(-1, 3)
    pcx = tf.Polynomial(cx, width=3).coefficients
    for i in list(range(min(n_points, 25))) + [n_points - 1]:
        assert np.array_equal(gx[i], oracle.poly_eval_xfe_point(pcx, px[3 * i:3 * i + 3]))
    # the coset evaluation is the batch evaluation on the explicit coset (polynomial.rs:3646-3662)
    if 0 < n_coeffs <= 128:
        order, off = 128, oracle.bfe_new(7)
        omega = oracle.primitive_root(order)
        coset = np.array([oracle.bfe_mul(off, oracle.bfe_mod_pow(omega, i)) for i in range(order)], dtype=np.uint64)
        assert np.array_equal(tf.Polynomial(c).batch_evaluate(coset), tf.fast_coset_evaluate(tf.Polynomial(c).coefficients, off, order))


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("n_coeffs,n_points", [(1024, 3), (4096 + 5, 17), (1 << 16, 2)])
def test_batch_evaluate_long_polynomials(tf, oracle, width, n_coeffs, n_points):
    """the coefficient-split kernel (n_coeffs >= 1024), ragged tail included"""
    c = oracle.fill_random(n_coeffs * width, 23 + n_coeffs)
    pts = oracle.fill_random(n_points * width, 29 + n_points)
    poly = tf.Polynomial(c, width=width)
    got = poly.batch_evaluate(pts).reshape(n_points, width)
    for i in range(n_points):
        if width == 1:
            assert int(got[i, 0]) == int(oracle.poly_eval(poly.coefficients, int(pts[i]))[0])
        else:
            assert np.array_equal(got[i], oracle.poly_eval_xfe_point(poly.coefficients, pts[3 * i:3 * i + 3]))


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("log_n,batch,n_points", [(0, 1, 4), (5, 2, 2), (8, 3, 33), (12, 2, 5), (16, 1, 3)])
def test_coset_extrapolate(tf, oracle, width, log_n, batch, n_points):
    """math/polynomial.rs:2117-2208: values of the coset interpolants at new points (doc example :2183-2195 included below)"""
    n = 1 << log_n
    off = oracle.bfe_new(7)
    cw = oracle.fill_random(batch * n * width, 31 + log_n)
    pts = oracle.fill_random(n_points * width, 37 + log_n)
    got = tf.Polynomial.batch_coset_extrapolate(off, n, cw, pts, width=width).reshape(batch, n_points, width)
    for b in range(batch):
        coeffs = oracle.coset_interpolate(cw[b * n * width:(b + 1) * n * width], off, width=width)
        for i in range(n_points):
            if width == 1:
                assert int(got[b, i, 0]) == int(oracle.poly_eval(coeffs, int(pts[i]))[0])
            else:
                assert np.array_equal(got[b, i], oracle.poly_eval_xfe_point(coeffs, pts[3 * i:3 * i + 3]))
    one = tf.Polynomial.coset_extrapolate(off, cw[:n * width], pts, width=width)
    assert np.array_equal(one.reshape(n_points, width), got[0])


def test_coset_extrapolate_doc_example_and_panics(tf, oracle):
    """polynomial.rs:2183-2195: constant codewords extrapolate to the constant; :2194 panics on a non-power-of-two length"""
    n = 32
    cw = np.concatenate([oracle.to_raw([3] * n), oracle.to_raw([2] * n)])
    pts = oracle.to_raw([0, 1])
    got = tf.Polynomial.batch_coset_extrapolate(oracle.bfe_new(7), n, cw, pts)
    assert list(got) == list(oracle.to_raw([3, 3, 2, 2]))
    with pytest.raises(tf.NttPanic):
        tf.Polynomial.batch_coset_extrapolate(oracle.bfe_new(7), 24, oracle.to_raw([1] * 24), pts)
    with pytest.raises(tf.TwentyFirstError):
        tf.Polynomial.batch_coset_extrapolate(0, n, cw, pts)  # offset.inverse() panics on zero
    # points on the coset itself give the codeword back
    omega = oracle.primitive_root(n)
    coset = np.array([oracle.bfe_mul(oracle.bfe_new(7), oracle.bfe_mod_pow(omega, i)) for i in range(n)], dtype=np.uint64)
    c1 = oracle.fill_random(n, 41)
    assert np.array_equal(tf.Polynomial.coset_extrapolate(oracle.bfe_new(7), c1, coset), c1)


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("n_coeffs,order,batch,force", [(1 << 11, 1 << 15, 3, 3), ((1 << 12) + 5, 1 << 16, 2, 3), (1500, 1 << 15, 1, 3),
                                                        (1 << 14, 1 << 15, 2, 3), (1 << 16, 1 << 19, 2, 0), (1 << 18, 1 << 21, 1, 0),
                                                        (1 << 20, 1 << 22, 1, 0), ((1 << 20) + 1, 1 << 22, 1, 0)])
def test_blown_up_coset_evaluation(tf, oracle, width, n_coeffs, order, batch, force):
    """fast_coset_evaluate with order >= 2 * degree (math/polynomial.rs:1374-1399).  When the padded coefficient length len
    needs a global pass fewer than the order, the device evaluates order / len interleaved cosets of length len; the
    small cases force that plan through tf_set_ntt_min_passes (three passes from 2^15).  Same words as the oracle's
    zero-pad-and-transform either way."""
    off = oracle.bfe_new(7)
    c = oracle.fill_random(batch * n_coeffs * width, 1000 + n_coeffs % 97)
    lib = tf._lib.lib()
    lib.tf_set_ntt_min_passes(force)
    try:
        got = tf.fast_coset_evaluate(c, off, order, width=width, batch=batch)
    finally:
        lib.tf_set_ntt_min_passes(0)
    for b in range(batch):
        want = oracle.coset_evaluate(c[b * n_coeffs * width:(b + 1) * n_coeffs * width], off, order, width=width)
        assert np.array_equal(got[b * order * width:(b + 1) * order * width], want)


@pytest.mark.parametrize("count,shift", [(1, 0), (2, 0), (7, 0), (1025, 0), (1024, 1), (4097, 1)])
def test_hadamard_bfe_alignment_and_odd_counts(tf, oracle, count, shift):
    """pointwise BFE product (math/polynomial.rs:920-925): the 16-byte path, its odd tail, and 8-byte-aligned operands"""
    import torch

    a = oracle.fill_random(count + shift, 50 + count)
    b = oracle.fill_random(count + shift, 51 + count)
    da = torch.from_numpy(a.view(np.int64)).cuda()[shift:]
    db = torch.from_numpy(b.view(np.int64)).cuda()[shift:]
    out = torch.empty(count + shift, dtype=torch.int64, device="cuda")[shift:]
    tf.device.hadamard(da, db, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), oracle.hadamard(a[shift:], b[shift:]))


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("n_rows,n_cols", [(1, 1), (8, 0), (8, 3), (64, 10), (256, 7), (1 << 11, 5), (1 << 12, 11), (1 << 16, 4)])
def test_rows_of_column_major_tables(tf, oracle, width, n_rows, n_cols):
    """hash_varlen (tip5/mod.rs:617-623) of the rows of a column-major table (XFE rows flattened as
    x_field_element.rs:217-231) and the tree over them; every kernel shape (a row pair per table row up to 8 rows per compute unit --
    2^11 on an MI355X --, 16 lanes per row up to 2^13 rows, the matrix-pipe form above)"""
    cols = oracle.fill_random(max(1, n_cols * n_rows * width), 60 + n_rows + n_cols)[: n_cols * n_rows * width]
    # the same table row-major on the host: row i = [col_0[i], col_1[i], ...]
    rows = cols.reshape(n_cols, n_rows, width).transpose(1, 0, 2).reshape(-1) if n_cols else np.zeros(0, dtype=np.uint64)
    row_len = n_cols * width
    want = oracle.hash_varlen_rows(rows, row_len) if row_len else np.concatenate([oracle.hash_varlen(np.zeros(0, np.uint64))] * n_rows)
    got = tf.Tip5.hash_table_rows(cols, n_rows, width=width) if n_cols else None
    if n_cols:
        assert np.array_equal(got, want)
        tree = tf.MerkleTree.from_columns(cols, n_rows, width=width)
        assert np.array_equal(tree.nodes.reshape(-1), oracle.merkle_build(want))


def test_column_major_tables_on_device_with_stride_and_batch(tf, oracle):
    """two tables of 5 XFE columns x 2^10 rows with padded columns (col_stride > n_rows * width), device-resident"""
    import torch

    n_rows, n_cols, width, pad, batch = 1 << 10, 5, 3, 16, 2
    cs = n_rows * width + pad
    raw = oracle.fill_random(batch * n_cols * cs, 71)
    dt = torch.from_numpy(raw.view(np.int64)).cuda()
    nodes = torch.empty(batch * 10 * n_rows, dtype=torch.int64, device="cuda")
    digs = torch.empty(batch * 5 * n_rows, dtype=torch.int64, device="cuda")
    tf.device.hash_table_rows(dt, n_rows, n_cols, digs, width=width, col_stride=cs, batch=batch)
    tf.device.merkle_from_columns(dt, n_rows, n_cols, nodes, width=width, col_stride=cs, batch=batch)
    torch.cuda.synchronize()
    gd, gn = digs.cpu().numpy().view(np.uint64), nodes.cpu().numpy().view(np.uint64)
    for b in range(batch):
        t = raw[b * n_cols * cs:(b + 1) * n_cols * cs].reshape(n_cols, cs)[:, : n_rows * width].reshape(n_cols, n_rows, width)
        rows = t.transpose(1, 0, 2).reshape(-1)
        want = oracle.hash_varlen_rows(rows, n_cols * width)
        assert np.array_equal(gd[b * 5 * n_rows:(b + 1) * 5 * n_rows], want)
        assert np.array_equal(gn[b * 10 * n_rows:(b + 1) * 10 * n_rows], oracle.merkle_build(want))


@pytest.mark.parametrize("width", [1, 3])
def test_column_major_tables_odd_row_counts_in_batches(tf, oracle, width):
    """hash_table_rows of a batch of tables whose row count is NOT a power of two, with more rows in total than the 16-lane
    kernels take: the matrix-pipe kernel finds (table, row) of an item by division, and its last wave is ragged"""
    import torch

    n_rows, n_cols, batch = 3001, 4, 5
    cs = n_rows * width + 3
    raw = oracle.fill_random(batch * n_cols * cs, 72 + width)
    dt = torch.from_numpy(raw.view(np.int64)).cuda()
    digs = torch.empty(batch * 5 * n_rows, dtype=torch.int64, device="cuda")
    tf.device.hash_table_rows(dt, n_rows, n_cols, digs, width=width, col_stride=cs, batch=batch)
    torch.cuda.synchronize()
    gd = digs.cpu().numpy().view(np.uint64)
    for b in range(batch):
        t = raw[b * n_cols * cs:(b + 1) * n_cols * cs].reshape(n_cols, cs)[:, : n_rows * width].reshape(n_cols, n_rows, width)
        rows = t.transpose(1, 0, 2).reshape(-1)
        assert np.array_equal(gd[b * 5 * n_rows:(b + 1) * 5 * n_rows], oracle.hash_varlen_rows(rows, n_cols * width))


@pytest.mark.parametrize("width", [1, 3])
def test_device_resident_commitment_pipeline(tf, oracle, width):
    """The callers on either side of the path chained in HBM: 6 columns of 2^10 values on the coset
    1 * <w> -> low-degree extension to 2^12 points on 7 * <w'> (one codeword per column) -> hash_varlen of every row of that
    column-major table -> Merkle tree -> authentication structure; every stage against the oracle."""
    import torch

    n, m, n_cols = 1 << 10, 1 << 12, 6
    one, seven = oracle.bfe_new(1), oracle.bfe_new(7)
    vals = oracle.fill_random(n_cols * n * width, 88 + width)
    d_vals = torch.from_numpy(vals.view(np.int64)).cuda()
    ext = torch.empty(n_cols * m * width, dtype=torch.int64, device="cuda")
    nodes = torch.empty(10 * m, dtype=torch.int64, device="cuda")
    tf.device.lde(d_vals, n, one, ext, m, seven, batch=n_cols, width=width)
    tf.device.merkle_from_columns(ext, m, n_cols, nodes, width=width)
    torch.cuda.synchronize()
    want_cols = []
    for j in range(n_cols):
        co = oracle.coset_interpolate(vals[j * n * width:(j + 1) * n * width], one, width=width)
        want_cols.append(oracle.coset_evaluate(co, seven, m, width=width))
    want_ext = np.concatenate(want_cols)
    assert np.array_equal(ext.cpu().numpy().view(np.uint64), want_ext)
    rows = want_ext.reshape(n_cols, m, width).transpose(1, 0, 2).reshape(-1)
    want_nodes = oracle.merkle_build(oracle.hash_varlen_rows(rows, n_cols * width))
    assert np.array_equal(nodes.cpu().numpy().view(np.uint64), want_nodes)
    idx = [5, 77, 4095, 2048]
    got_auth = tf.device.authentication_structure(nodes, m, idx)
    want_idx = oracle.auth_structure_indices(m, idx)
    assert np.array_equal(got_auth, want_nodes.reshape(2 * m, 5)[np.asarray(want_idx, dtype=np.int64)])


@pytest.mark.parametrize("width,na,nb", [(1, 20000, 12769), (1, 40000, 50000), (1, (1 << 20) + 5, 1 << 20), (3, 300000, 290001), (3, 9000, 9000)])
def test_fast_multiply_large_products(tf, oracle, width, na, nb):
    """products whose transform length is >= 2^15: zero padding in the first pass, (BFE) the pointwise product on the
    inverse's first load, and the truncation to na + nb - 1 coefficients inside its last pass"""
    a = oracle.fill_random(na * width, 3 + na)
    b = oracle.fill_random(nb * width, 4 + nb)
    assert np.array_equal(tf.fast_multiply(a, b, width=width), oracle.poly_mul(a, b, width=width))
    if na <= 50000:
        assert np.array_equal(tf.fast_square(a, width=width), oracle.poly_mul(a, a, width=width))


@pytest.mark.parametrize("width,na,nb,batch,shift", [(1, 20000, 12769, 5, 0), (1, 33000, 33001, 4, 3), (1, 70000, 61073, 3, 0),
                                                     (1, 1 << 19, (1 << 19) - 15, 3, 0), (1, 600000, 500003, 2, 7),
                                                     (3, 20000, 12769, 5, 0), (3, 16001, 16000, 3, 5), (3, 1 << 19, (1 << 19) - 15, 2, 0),
                                                     (3, 600000, 440001, 2, 1)])
def test_fast_multiply_batched_products_with_unaligned_outputs(tf, oracle, width, na, nb, batch, shift):
    """products in a batch: entry b of the output starts at word b * (na + nb - 1) * width (+ an unaligned caller pointer), so the
    R = 1024 last pass of the inverse transform shifts its (word-granular) tile boundaries per entry to keep its stores on
    cache lines; XFE orders 2^15 and 2^20 take that kernel too"""
    import torch

    a = oracle.fill_random(na * batch * width, 7 + na)
    b = oracle.fill_random(nb * batch * width, 8 + nb)
    n_out = (na + nb - 1) * width
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    out = torch.full((n_out * batch + shift + 64,), -1, dtype=torch.int64, device="cuda")
    tf.device.poly_mul(da, na, db, nb, out[shift:shift + n_out * batch], batch=batch, width=width)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    assert np.all(got[:shift] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(got[shift + n_out * batch:] == np.uint64(0xFFFFFFFFFFFFFFFF))
    for k in range(batch):
        want = oracle.poly_mul(a[k * na * width:(k + 1) * na * width], b[k * nb * width:(k + 1) * nb * width], width=width)
        assert np.array_equal(got[shift + k * n_out:shift + (k + 1) * n_out], want), k


@pytest.mark.parametrize("width,log_n,batch,shift", [(1, 15, 3, 5), (1, 17, 2, 1), (1, 20, 2, 15), (1, 21, 1, 9), (1, 23, 1, 8),
                                                     (3, 15, 3, 5), (3, 13, 2, 1), (3, 20, 2, 7), (3, 23, 1, 2), (3, 24, 1, 0),
                                                     (1, 21, 2, 9), (3, 21, 1, 3), (1, 22, 1, 11), (3, 22, 1, 5),   # the two-pass plan (PRE2)
                                                     (1, 9, 3, 7), (3, 12, 1, 1), (1, 16, 1, 13), (3, 14, 2, 6)])     # the latency-shaped kernels
@pytest.mark.parametrize("inverse", [False, True])
def test_ntt_on_unaligned_device_pointer(tf, oracle, width, log_n, batch, shift, inverse):
    """ntt / intt (math/ntt.rs:67-125) in place on a device slice that does not start on a 128-byte line; the XFE lengths are
    the ones whose last pass is the R = 1024 kernel with word-granular tiles (2^15, 2^20, >= 2^23) plus one that is not"""
    import torch

    n = 1 << log_n
    words = n * batch * width
    x = oracle.fill_random(words, 90 + log_n)
    buf = torch.full((words + shift + 32,), -1, dtype=torch.int64, device="cuda")
    buf[shift:shift + words] = torch.from_numpy(x.view(np.int64)).cuda()
    tf.device.ntt_(buf[shift:shift + words], n, batch=batch, width=width, inverse=inverse)
    torch.cuda.synchronize()
    got = buf.cpu().numpy().view(np.uint64)
    assert np.all(got[:shift] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(got[shift + words:] == np.uint64(0xFFFFFFFFFFFFFFFF))
    assert np.array_equal(got[shift:shift + words], oracle.ntt(x, width=width, inverse=inverse, batch=batch, threads=8))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("width,log_n,batch", [(1, 15, 3), (1, 16, 1), (1, 18, 2), (1, 20, 1), (1, 21, 1), (1, 22, 1),
                                               (3, 11, 5), (3, 13, 2), (3, 15, 1), (3, 17, 3), (3, 20, 1), (3, 21, 1)])
def test_both_launch_geometries_match_oracle(tf, oracle, mode, width, log_n, batch):
    """Calls with little work are planned with 256-thread workgroups and the generic last pass, large ones with 512-thread
    workgroups and the R = 1024 kernel (tf_set_ntt_small_launch: 0 = never, 1 = always): both are the same transform
    (math/ntt.rs:67-125), as are the coset evaluation / interpolation built on them (polynomial.rs:1374-1399, :1907-1918)"""
    lib = tf._lib.lib()
    n = 1 << log_n
    x = oracle.fill_random(n * width * batch, 1234 + log_n + width)
    off = oracle.bfe_new(0x1234567 + log_n)
    lib.tf_set_ntt_small_launch(mode)
    try:
        fwd = x.copy(); tf.ntt(fwd, width=width, batch=batch)
        inv = x.copy(); tf.intt(inv, width=width, batch=batch)
        ev = tf.fast_coset_evaluate(x[: (n // 2 + 3) * width], off, n, width=width)
        ip = tf.fast_coset_interpolate(x[: n * width], off, width=width)
    finally:
        lib.tf_set_ntt_small_launch(-1)
    assert np.array_equal(fwd, oracle.ntt(x, width=width, batch=batch, threads=8))
    assert np.array_equal(inv, oracle.ntt(x, width=width, inverse=True, batch=batch, threads=8))
    assert np.array_equal(ev, oracle.coset_evaluate(x[: (n // 2 + 3) * width], off, n, width=width))
    assert np.array_equal(ip, oracle.coset_interpolate(x[: n * width], off, width=width))


@pytest.mark.parametrize("width,n,m", [(1, 3000, 2048), (1, 9000, 5000), (1, 20000, 4096), (1, 1 << 14, 1 << 13), (1, 700, 2100),
                                       (3, 1000, 777), (3, 2048, 2048), (3, 300, 512), (3, 5000, 1100)])
def test_zerofier_tree_batch_evaluate_matches_horner_and_oracle(tf, oracle, width, n, m):
    """SURVEY 8(f4) at the reference's complexity (math/polynomial.rs:1840-1894, math/zerofier_tree.rs): the zerofier-tree route
    of tf_poly_batch_evaluate_* -- leaves of 256 (BFE) / 128 (XFE) points, levels in the transform domain, remainders by power-series inverses, chunks when the polynomial is
    longer than the padded point count -- returns the same words as the Horner route and as the oracle's Horner; point counts
    that are not powers of two, polynomials shorter and longer than the point count, duplicate and zero points."""
    import torch

    L = tf.lib()
    c = oracle.fill_random(n * width, 900 + n)
    pts = oracle.fill_random(m * width, 901 + m)
    pts[: width] = 0                      # the point 0
    pts[width: 2 * width] = pts[2 * width: 3 * width]  # a duplicate
    dc, dp = _to_dev(c), _to_dev(pts)
    out_t = torch.empty(m * width, dtype=torch.int64, device="cuda")
    out_h = torch.empty(m * width, dtype=torch.int64, device="cuda")
    try:
        L.tf_set_batch_eval_route(2)
        tf.device.batch_evaluate(dc, n, dp, out_t, width=width)
        L.tf_set_batch_eval_route(1)
        tf.device.batch_evaluate(dc, n, dp, out_h, width=width)
    finally:
        L.tf_set_batch_eval_route(0)
    torch.cuda.synchronize()
    got_t, got_h = _to_host(out_t), _to_host(out_h)
    assert np.array_equal(got_t, got_h)
    for i in list(range(5)) + [m // 2, m - 1]:
        if width == 1:
            want = oracle.poly_eval(c, int(pts[i]))
        else:
            want = oracle.poly_eval_xfe_point(c, pts[3 * i: 3 * i + 3])
        assert np.array_equal(got_t[i * width:(i + 1) * width], np.asarray(want).reshape(-1)), i


def test_zerofier_tree_at_m_equals_n_2pow16(tf, oracle):
    """parity of the tree route against Horner for m = n = 2^16 (BFieldElement), and the automatic
    route picks the tree there; the two timings are recorded by tools/batch_eval_sweep.py."""
    import torch

    L = tf.lib()
    n = m = 1 << 16
    dc = torch.empty(n, dtype=torch.int64, device="cuda")
    dp = torch.empty(m, dtype=torch.int64, device="cuda")
    tf.device.fill_random(dc, 11)
    tf.device.fill_random(dp, 12)
    outs = []
    try:
        for route in (2, 1, 0):
            o = torch.empty(m, dtype=torch.int64, device="cuda")
            L.tf_set_batch_eval_route(route)
            tf.device.batch_evaluate(dc, n, dp, o)
            outs.append(o)
    finally:
        L.tf_set_batch_eval_route(0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    c = _to_host(dc)
    p = _to_host(dp)
    for i in (0, 1, 12345, m - 1):
        assert int(_to_host(outs[0])[i]) == int(np.asarray(oracle.poly_eval(c, int(p[i]))).reshape(-1)[0])


# ---- zerofier and interpolation through the zerofier tree (math/polynomial.rs:1435-1838) ---------------------------------------
def _distinct_points(oracle, n, width, seed):
    pts = oracle.fill_random(n * width, seed)
    if n > 1:
        pts[width: 2 * width] = 0  # the point 0 is a legal domain point
    return pts


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 127, 128, 129, 255, 256, 257, 511, 513, 1023, 1024, 1025, 2049, 3000, 5000])
def test_zerofier_matches_oracle(tf, oracle, width, n):
    """Polynomial::zerofier (polynomial.rs:1435-1441): the root of the device's padded tree against the oracle's smart_zerofier
    (:1462-1475), word for word -- sizes around the leaf sizes (1024 BFE / 256 XFE) and the padded tree sizes, repeated and
    zero roots; monic (:3485-3488)."""
    r = oracle.fill_random(n * width, 1300 + n)
    if n > 3:
        r[: width] = 0
        r[width: 2 * width] = r[2 * width: 3 * width]  # a repeated root is allowed
    z = np.empty((n + 1) * width, dtype=np.uint64)
    fn = tf.lib().tf_poly_zerofier_bfe if width == 1 else tf.lib().tf_poly_zerofier_xfe
    import ctypes as C

    assert fn(C.c_void_p(r.ctypes.data) if n else C.c_void_p(0), n, C.c_void_p(z.ctypes.data)) == 0
    assert np.array_equal(z, oracle.zerofier(r, width))
    assert int(z[n * width]) == oracle.bfe_new(1) and not z[n * width + 1:].any()
    assert tf.Polynomial.zerofier(r, width=width).degree() == n


@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("n", [1, 2, 4, 17, 127, 128, 129, 255, 256, 257, 512, 513, 1024, 1025, 2051, 3000])
def test_interpolate_matches_lagrange_oracle(tf, oracle, width, n):
    """Polynomial::interpolate / fast_interpolate (polynomial.rs:1502-1654) against the oracle's lagrange_interpolate (:1565-1606),
    the identity the reference tests at :3572-3582; the interpolant evaluates back to the values (:3602-3612)."""
    d = _distinct_points(oracle, n, width, 1400 + n)
    v = oracle.fill_random(n * width, 1401 + n)
    got = np.empty(n * width, dtype=np.uint64)
    fn = tf.lib().tf_poly_interpolate_bfe if width == 1 else tf.lib().tf_poly_interpolate_xfe
    import ctypes as C

    assert fn(C.c_void_p(d.ctypes.data), C.c_void_p(v.ctypes.data), n, 1, C.c_void_p(got.ctypes.data)) == 0
    assert np.array_equal(got, oracle.lagrange_interpolate(d, v, width))
    assert np.array_equal(tf.Polynomial(got, width=width).batch_evaluate(d), v)


def test_interpolate_doc_examples_and_panics(tf, oracle):
    """polynomial.rs:1490-1497 (doc example), :3522-3570 (no points, unequal lengths, repeated points, one point)."""
    P_ = tf.Polynomial
    f = P_.interpolate(oracle.to_raw([0, 1, 2, 3]), oracle.to_raw([1, 3, 5, 7]))
    assert f.degree() == 1 and np.array_equal(f.coefficients, oracle.to_raw([1, 2]))
    assert np.array_equal(f.batch_evaluate(oracle.to_raw([4])), oracle.to_raw([9]))
    z = P_.zerofier(oracle.to_raw([2, 4, 6]))
    assert z.degree() == 3 and not z.batch_evaluate(oracle.to_raw([2, 4, 6])).any()
    assert z.batch_evaluate(oracle.to_raw([0, 1, 3, 5])).all()
    assert np.array_equal(P_.zerofier(np.zeros(0, dtype=np.uint64)).coefficients, oracle.to_raw([1]))
    assert np.array_equal(P_.interpolate(oracle.to_raw([5]), oracle.to_raw([42])).coefficients, oracle.to_raw([42]))
    with pytest.raises(tf.NttPanic) as e:
        P_.interpolate(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64))
    assert e.value.code == 14
    with pytest.raises(tf.NttPanic):
        P_.interpolate(oracle.to_raw([1, 2]), oracle.to_raw([1]))
    for n in (2, 300, 1500, 5000):  # a repeated domain point: inside one leaf and across leaves
        d = oracle.fill_random(n, 77)
        d[n - 1] = d[n // 3]
        with pytest.raises(tf.NttPanic) as e:
            P_.interpolate(d, oracle.fill_random(n, 78))
        assert e.value.code == 12
    dx = oracle.fill_random(3 * 700, 79)
    dx[3 * 699: 3 * 700] = dx[0:3]
    with pytest.raises(tf.NttPanic) as e:
        P_.interpolate(dx, oracle.fill_random(3 * 700, 80), width=3)
    assert e.value.code == 12
    import ctypes as C

    assert tf.lib().tf_poly_interpolate_bfe(C.c_void_p(0), C.c_void_p(0), 0, 1, C.c_void_p(0)) == 14
    assert tf.lib().tf_status_string(14) == b"TF_ERR_EMPTY_DOMAIN"


@pytest.mark.parametrize("width,n,rows", [(1, 700, 3), (1, 2500, 4), (3, 300, 2), (3, 1100, 3)])
def test_batch_fast_interpolate_rows_share_the_domain(tf, oracle, width, n, rows):
    """batch_fast_interpolate (polynomial.rs:1703-1838) equals interpolate row by row (:3614-3663)."""
    d = _distinct_points(oracle, n, width, 1500 + n)
    vals = [oracle.fill_random(n * width, 1501 + n + r) for r in range(rows)]
    polys = tf.Polynomial.batch_fast_interpolate(d, vals, width=width)
    assert len(polys) == rows
    for r in range(rows):
        want = oracle.lagrange_interpolate(d, vals[r], width)
        assert np.array_equal(polys[r].coefficients, tf.Polynomial(want, width=width).coefficients)
        assert np.array_equal(polys[r].coefficients, tf.Polynomial.interpolate(d, vals[r], width=width).coefficients)


@pytest.mark.parametrize("width,log_n", [(1, 10), (1, 13), (1, 16), (3, 9), (3, 14)])
def test_interpolation_on_a_coset_equals_fast_coset_interpolate(tf, oracle, width, log_n):
    """polynomial.rs:3665-3680: interpolating through the points of a coset gives fast_coset_interpolate's polynomial -- here the
    tree interpolation against the NTT path of the same library, at sizes the O(n^2) oracle does not reach; and the interpolant
    of f's values on arbitrary points is f (deg f < n)."""
    import torch

    n = 1 << log_n
    off = oracle.bfe_new(7)
    w = tf.BFieldElement.primitive_root_of_unity(n)
    pts = np.zeros(n * width, dtype=np.uint64)
    x = off
    for i in range(n):
        pts[i * width] = x
        x = oracle.bfe_mul(x, w)
    v = oracle.fill_random(n * width, 1600 + log_n)
    want = tf.fast_coset_interpolate(v, off, width=width)
    got = tf.Polynomial.interpolate(pts, v, width=width)
    assert np.array_equal(got.coefficients, tf.Polynomial(want, width=width).coefficients)
    # device-resident round trip on random points: evaluate f, interpolate the values, get f back (untrimmed coefficients)
    f = _to_dev(oracle.fill_random(n * width, 1601 + log_n))
    dom = torch.empty(n * width, dtype=torch.int64, device="cuda")
    tf.device.fill_random(dom, 1602 + log_n)
    vals = torch.empty_like(dom)
    back = torch.empty_like(dom)
    tf.device.batch_evaluate(f, n, dom, vals, width=width)
    tf.device.interpolate(dom, vals, back, rows=1, width=width)
    torch.cuda.synchronize()
    assert torch.equal(back, f)
    z = torch.empty((n + 1) * width, dtype=torch.int64, device="cuda")
    tf.device.zerofier(dom, z, width=width)
    zv = torch.empty_like(dom)
    tf.device.batch_evaluate(z, n + 1, dom, zv, width=width)
    torch.cuda.synchronize()
    assert not zv.any().item()


# ---- clean division (math/polynomial.rs:2358-2411) ------------------------------------------------------------------------------
@pytest.mark.parametrize("nq,nb", [(1, 1), (5, 3), (3, 5), (100, 40), (700, 600), (513, 513), (1500, 520), (3, 1000), (5000, 4097),
                                   (1 << 15, (1 << 15) + 1), ((1 << 18) + 3, 1 << 17)])
def test_clean_divide_matches_oracle(tf, oracle, nq, nb):
    """Polynomial::clean_divide against the oracle's restatement of it (both of the reference's routes: naive_divide below
    divisor degree 512 and the extension-field coset above, :2360-2364) -- the property test :3707-3719 with a = q * b."""
    q, b = oracle.fill_random(nq, 1700 + nq), oracle.fill_random(nb, 1701 + nb)
    a = oracle.poly_mul(q, b)
    got = tf.Polynomial(a).clean_divide(tf.Polynomial(b))
    assert np.array_equal(got.coefficients, q)
    if nq * nb <= 1 << 22:
        assert np.array_equal(oracle.clean_divide(a, b, 0), q) and np.array_equal(oracle.clean_divide(a, b), q)
        assert np.array_equal(oracle.naive_divide(a, b)[0], q)


def test_clean_divide_roots_at_zero_and_panics(tf, oracle):
    """polynomial.rs:3722-3787 (divisor with 0 as a single / multiple / only root, roots 0..9), :4433 (monomials), and the panics:
    zero divisor, unclean division, dividend of lower degree."""
    P_ = tf.Polynomial
    one = oracle.bfe_new(1)
    for k in (1, 2, 7):  # divisor x^k
        q = oracle.fill_random(40, 50 + k)
        xk = np.zeros(k + 1, dtype=np.uint64)
        xk[k] = one
        a = np.concatenate([np.zeros(k, dtype=np.uint64), q])
        assert np.array_equal(P_(a).clean_divide(P_(xk)).coefficients, q)
    b = oracle.fill_random(600, 60)
    b[0] = 0
    b[1] = 0
    q = oracle.fill_random(700, 61)
    a = oracle.poly_mul(q, b)
    assert np.array_equal(P_(a).clean_divide(P_(b)).coefficients, q)
    assert np.array_equal(oracle.clean_divide(a, b, 0), q)
    z = P_.zerofier(oracle.to_raw(list(range(10))))  # roots 0 through 9
    q = oracle.fill_random(33, 62)
    a = oracle.poly_mul(q, z.coefficients)
    assert np.array_equal(P_(a).clean_divide(z).coefficients, q)
    # monomial / smaller monomial (:4433-4450)
    hi = np.zeros(10, dtype=np.uint64)
    hi[9] = oracle.bfe_new(6)
    lo = np.zeros(4, dtype=np.uint64)
    lo[3] = oracle.bfe_new(3)
    got = P_(hi).clean_divide(P_(lo)).coefficients
    assert got.size == 7 and int(got[6]) == oracle.bfe_new(2) and not got[:6].any()
    # zero dividend: zero quotient
    assert P_(np.zeros(0, dtype=np.uint64)).clean_divide(P_(b)).degree() == -1
    with pytest.raises(tf.NttPanic) as e:
        P_(a).clean_divide(P_(np.zeros(3, dtype=np.uint64)))
    assert e.value.code == 15
    for na_, nb_ in ((900, 600), (40, 7)):  # unclean: one coefficient of a clean product perturbed; x does not divide 1 + ...
        bb, qq = oracle.fill_random(nb_, 63), oracle.fill_random(na_ - nb_ + 1, 64)
        aa = oracle.poly_mul(qq, bb)
        aa[na_ // 2] ^= np.uint64(1)
        with pytest.raises(tf.NttPanic) as e:
            P_(aa).clean_divide(P_(bb))
        assert e.value.code == 16
        with pytest.raises(oracle.OraclePanic):
            oracle.clean_divide(aa, bb, 0)
    with pytest.raises(tf.NttPanic) as e:
        P_(oracle.fill_random(5, 65)).clean_divide(P_(oracle.fill_random(9, 66)))
    assert e.value.code == 16
    bz = oracle.fill_random(30, 67)
    bz[0] = 0
    with pytest.raises(tf.NttPanic) as e:  # :2374 assert!(dividend_coefficients[0].is_zero())
        P_(oracle.fill_random(60, 68)).clean_divide(P_(bz))
    assert e.value.code == 16
    assert tf.lib().tf_status_string(15) == b"TF_ERR_DIVISION_BY_ZERO" and tf.lib().tf_status_string(16) == b"TF_ERR_DIVISION_NOT_CLEAN"


def test_clean_divide_device_resident_large(tf, oracle):
    """2^20-coefficient quotient times a 2^20-coefficient divisor, all in HBM: fast_multiply then clean_divide gives the factor back."""
    import torch

    nq = nb = 1 << 20
    q = torch.empty(nq, dtype=torch.int64, device="cuda")
    b = torch.empty(nb, dtype=torch.int64, device="cuda")
    tf.device.fill_random(q, 71)
    tf.device.fill_random(b, 72)
    a = torch.empty(nq + nb - 1, dtype=torch.int64, device="cuda")
    tf.device.poly_mul(q, nq, b, nb, a)
    out = torch.empty(nq, dtype=torch.int64, device="cuda")
    tf.device.clean_divide(a, b, out)
    torch.cuda.synchronize()
    assert torch.equal(out, q)


@pytest.mark.parametrize("n_coeffs,order,batch", [(1, 1, 1), (3, 8, 2), (100, 128, 3), (1024, 1024, 1), (3000, 4096, 2), (1 << 15, 1 << 16, 1)])
def test_coset_evaluation_with_an_extension_field_offset(tf, oracle, n_coeffs, order, batch):
    """fast_coset_evaluate / fast_coset_interpolate with S = XFieldElement (math/polynomial.rs:1374-1378, :1907-1911; the
    reference tests the scaling with offsets of either field at :2921-2944): bit-exact against the oracle's sequential power
    chain, and interpolate inverts evaluate."""
    off = oracle.fill_random(3, 1800 + n_coeffs)
    c = oracle.fill_random(3 * n_coeffs * batch, 1801 + n_coeffs)
    got = tf.fast_coset_evaluate(c, off, order, width=3, batch=batch).reshape(batch, -1)
    for b in range(batch):
        want = oracle.coset_evaluate_xfe_offset(c[3 * n_coeffs * b: 3 * n_coeffs * (b + 1)], off, order)
        assert np.array_equal(got[b], want)
    back = tf.fast_coset_interpolate(got.reshape(-1), off, width=3, batch=batch).reshape(batch, -1)
    for b in range(batch):
        assert np.array_equal(back[b, : 3 * n_coeffs], c[3 * n_coeffs * b: 3 * n_coeffs * (b + 1)]) and not back[b, 3 * n_coeffs:].any()
        assert np.array_equal(back[b], oracle.coset_interpolate_xfe_offset(got[b], off))
    # an offset in the base field, given as an XFieldElement, gives the base-field path's values
    lifted = np.array([oracle.bfe_new(7), 0, 0], dtype=np.uint64)
    assert np.array_equal(tf.fast_coset_evaluate(c, lifted, order, width=3, batch=batch),
                          tf.fast_coset_evaluate(c, oracle.bfe_new(7), order, width=3, batch=batch))


def test_extension_field_offset_panics(tf, oracle):
    c = oracle.fill_random(3 * 16, 5)
    off = oracle.fill_random(3, 6)
    with pytest.raises(tf.NttPanic) as e:
        tf.fast_coset_evaluate(c, off, 8, width=3)      # order <= degree (:1388-1392)
    assert e.value.code == 6
    with pytest.raises(tf.NttPanic) as e:
        tf.fast_coset_evaluate(c[: 3 * 5], off, 12, width=3)  # order not a power of two
    assert e.value.code == 4
    with pytest.raises(tf.NttPanic) as e:
        tf.fast_coset_interpolate(c, np.zeros(3, dtype=np.uint64), width=3)  # the zero offset has no inverse
    assert e.value.code == 12
    with pytest.raises(TypeError):
        tf.fast_coset_evaluate(oracle.fill_random(16, 7), off, 16, width=1)  # BFieldElement * XFieldElement is not a BFieldElement


# ---- a zerofier tree kept across calls (math/zerofier_tree.rs) ----------------------------------------------------------------
@pytest.mark.parametrize("width,n", [(1, 0), (1, 1), (1, 100), (1, 256), (1, 700), (1, 5000), (3, 1), (3, 128), (3, 300), (3, 2100)])
def test_zerofier_tree_handle_matches_the_one_shot_calls(tf, oracle, width, n):
    """ZerofierTree::new_from_domain + zerofier (zerofier_tree.rs:66-99) and divide_and_conquer_batch_evaluate
    (polynomial.rs:1882-1894) over a tree kept in HBM: the same words as the one-shot entry points and as the oracle; the tree can be
    empty (:136-138); interpolation over the prepared tree equals Polynomial::interpolate; repeated use gives the same result."""
    d = _distinct_points(oracle, n, width, 1900 + n)
    with tf.ZerofierTree.new_from_domain(d, width=width) as tree:
        assert tree.num_points == n
        z = tree.zerofier()
        assert np.array_equal(z.coefficients, tf.Polynomial(oracle.zerofier(d, width), width=width).coefficients)
        for n_coeffs in (0, 1, 5, max(n // 2, 1), n + 3, 3 * n + 1):
            f = tf.Polynomial(oracle.fill_random(n_coeffs * width, 1901 + n_coeffs), width=width)
            got = tree.batch_evaluate(f)
            if n:
                assert np.array_equal(got, f.batch_evaluate(d))
                for i in (0, n // 2, n - 1):
                    want = oracle.poly_eval(f.coefficients, int(d[i])) if width == 1 else oracle.poly_eval_xfe_point(f.coefficients, d[3 * i: 3 * i + 3])
                    assert np.array_equal(got[i * width:(i + 1) * width], np.asarray(want).reshape(-1))
            else:
                assert got.size == 0
        if n == 0:
            with pytest.raises(tf.NttPanic) as e:
                tree.interpolate([np.zeros(0, dtype=np.uint64)])
            assert e.value.code == 14
            return
        vals = [oracle.fill_random(n * width, 1902 + r) for r in range(3)]
        for attempt in range(2):  # the second pass reuses the cached weights
            polys = tree.interpolate(vals)
            for r in range(3):
                assert np.array_equal(polys[r].coefficients, tf.Polynomial.interpolate(d, vals[r], width=width).coefficients)
        if n <= 700:
            assert np.array_equal(polys[0].coefficients, tf.Polynomial(oracle.lagrange_interpolate(d, vals[0], width), width=width).coefficients)


def test_zerofier_tree_handle_on_device_buffers_and_errors(tf, oracle):
    import ctypes as C

    import torch

    n, rows = 1 << 14, 4
    dom = torch.empty(n, dtype=torch.int64, device="cuda")
    tf.device.fill_random(dom, 31)
    side = torch.cuda.Stream()
    with tf.device.ZerofierTree(dom) as tree:
        f = torch.empty(2 * n, dtype=torch.int64, device="cuda")       # two polynomials of n coefficients
        tf.device.fill_random(f, 32)
        vals = torch.empty(2 * n, dtype=torch.int64, device="cuda")
        tree.batch_evaluate(f, n, vals, batch=2)
        one = torch.empty(n, dtype=torch.int64, device="cuda")
        tf.device.batch_evaluate(f[n:], n, dom, one)
        torch.cuda.synchronize()
        assert torch.equal(vals[n:], one)
        back = torch.empty(2 * n, dtype=torch.int64, device="cuda")
        with torch.cuda.stream(side):                                   # a second stream on the same handle
            tree.interpolate(vals, back, rows=2, stream=side)
        z = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        tree.zerofier(z)
        torch.cuda.synchronize()
        assert torch.equal(back, f)
        zz = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        tf.device.zerofier(dom, zz)
        torch.cuda.synchronize()
        assert torch.equal(z, zz)
        with pytest.raises(ValueError):
            tree.batch_evaluate(f, n, vals[:n], batch=2)
    # a repeated point: the tree builds (a zerofier may have repeated roots), the interpolation panics
    d = oracle.fill_random(600, 33)
    d[599] = d[7]
    with tf.ZerofierTree(d) as tree:
        assert tree.zerofier().degree() == 600
        with pytest.raises(tf.NttPanic) as e:
            tree.interpolate([oracle.fill_random(600, 34)])
        assert e.value.code == 12
    lib = tf.lib()
    assert lib.tf_zerofier_tree_num_points(C.c_void_p(0)) == 0 and lib.tf_zerofier_tree_width(C.c_void_p(0)) == 0
    assert lib.tf_zerofier_tree_zerofier(C.c_void_p(0), C.c_void_p(0)) == 7
    lib.tf_zerofier_tree_free(C.c_void_p(0))  # freeing nothing is fine
    closed = tf.ZerofierTree(oracle.fill_random(10, 35))
    closed.close()
    closed.close()
    with pytest.raises(ValueError):
        closed.zerofier()


# ---- barycentric evaluation (math/polynomial.rs:2609-2637) ---------------------------------------------------------------------
@pytest.mark.parametrize("width", [1, 3])
@pytest.mark.parametrize("log_n,batch", [(0, 1), (1, 2), (3, 3), (6, 5), (11, 4), (12, 1), (16, 3), (20, 1)])
def test_barycentric_evaluate_matches_oracle_and_the_polynomial(tf, oracle, width, log_n, batch):
    """barycentric_evaluate for a batch of codewords at one out-of-domain point, XFieldElement and BFieldElement indeterminates, against
    the oracle's restatement (up to 2^12 points) and against the polynomial itself (the reference's property test :4593-4616:
    codeword = the polynomial's values on the subgroup, barycentric value = polynomial.evaluate)."""
    n = 1 << log_n
    coeffs = oracle.fill_random(n * width * batch, 2200 + log_n)
    cw = coeffs.copy()
    tf.ntt(cw, width=width, batch=batch)  # values on <w_n>, natural order
    for x in (oracle.fill_random(3, 2201 + log_n), np.array([oracle.bfe_new(987654321)], dtype=np.uint64)):
        got = tf.barycentric_evaluate(cw, x, width=width, batch=batch)
        base_field = width == 1 and x.size == 1
        got = got.reshape(batch, 1 if base_field else 3)
        x3 = np.zeros(3, dtype=np.uint64)
        x3[: x.size] = x
        for b in sorted({0, batch - 1}):
            c_b = coeffs[b * n * width:(b + 1) * n * width]
            if width == 1:
                lifted = np.zeros(3 * n, dtype=np.uint64)
                lifted[0::3] = c_b
            else:
                lifted = c_b
            want = oracle.poly_eval_xfe_point(lifted, x3)
            assert np.array_equal(got[b], want[:1] if base_field else want), (b, x.size)
            if log_n <= 12:
                o = oracle.barycentric_evaluate(cw[b * n * width:(b + 1) * n * width], x, width)
                assert np.array_equal(got[b], o[:1] if base_field else o)
                if base_field:
                    assert not o[1:].any()


def test_barycentric_evaluate_on_device_and_panics(tf, oracle):
    import ctypes as C

    import torch

    n, batch = 1 << 14, 300
    cw = torch.empty(n * batch, dtype=torch.int64, device="cuda")
    tf.device.fill_random(cw, 41)
    x = oracle.fill_random(3, 42)
    out = torch.empty(3 * batch, dtype=torch.int64, device="cuda")
    tf.device.barycentric_evaluate(cw, n, x, out, batch=batch)
    # the same values through coset_extrapolate (offset 1: the subgroup itself) on lifted codewords' interpolants
    ref = tf.Polynomial.batch_coset_extrapolate(oracle.bfe_new(1), n, _to_host(cw)[: 3 * n], np.array([oracle.bfe_new(5)], dtype=np.uint64))
    b3 = tf.barycentric_evaluate(_to_host(cw)[: 3 * n], oracle.bfe_new(5), batch=3)
    torch.cuda.synchronize()
    assert np.array_equal(b3, ref)
    got = _to_host(out).reshape(batch, 3)
    for b in (0, 17, batch - 1):
        assert np.array_equal(got[b], tf.barycentric_evaluate(_to_host(cw)[b * n:(b + 1) * n], x))
    with pytest.raises(tf.NttPanic) as e:
        tf.barycentric_evaluate(oracle.fill_random(12, 1), x)          # length not a power of two
    assert e.value.code == 4
    w = tf.BFieldElement.primitive_root_of_unity(16)
    inside = oracle.bfe_mod_pow(w, 5)
    with pytest.raises(tf.NttPanic) as e:
        tf.barycentric_evaluate(oracle.fill_random(16, 2), inside)     # the indeterminate is a domain point
    assert e.value.code == 12
    assert tf.barycentric_evaluate(oracle.fill_random(16, 2), np.array([inside, 1, 0], dtype=np.uint64)).size == 3  # same limb 0, outside the base field
    with pytest.raises(tf.NttPanic) as e:
        tf.barycentric_evaluate(np.zeros(0, dtype=np.uint64), x)       # no points: the denominator is zero
    assert e.value.code == 12
    with pytest.raises(ValueError):
        tf.device.barycentric_evaluate(cw, n, x, out[:-1], batch=batch)


@pytest.mark.parametrize("width,log_n", [(1, 22), (3, 21)])
def test_tree_round_trips_at_millions_of_points(tf, oracle, width, log_n):
    """Size-independent properties at sizes the O(n^2) oracle cannot reach: evaluate -> interpolate is the identity on 2^22 (BFE) /
    2^21 (XFE) random points, and the zerofier of the points vanishes on all of them (polynomial.rs:3470-3483, :3602-3612)."""
    import torch

    n = 1 << log_n
    dom = torch.empty(n * width, dtype=torch.int64, device="cuda")
    f = torch.empty(n * width, dtype=torch.int64, device="cuda")
    tf.device.fill_random(dom, 51)
    tf.device.fill_random(f, 52)
    vals, back = torch.empty_like(f), torch.empty_like(f)
    tf.device.batch_evaluate(f, n, dom, vals, width=width)
    tf.device.interpolate(dom, vals, back, width=width)
    z = torch.empty((n + 1) * width, dtype=torch.int64, device="cuda")
    tf.device.zerofier(dom, z, width=width)
    zv = torch.empty_like(dom)
    tf.device.batch_evaluate(z, n + 1, dom, zv, width=width)
    torch.cuda.synchronize()
    assert torch.equal(back, f)
    assert not zv.any().item()
    i = 12345
    c, p = _to_host(f), _to_host(dom)
    want = oracle.poly_eval(c, int(p[i])) if width == 1 else oracle.poly_eval_xfe_point(c, p[3 * i: 3 * i + 3])
    assert np.array_equal(_to_host(vals)[i * width:(i + 1) * width], np.asarray(want).reshape(-1))


def test_horner_route_with_more_points_than_one_launch_takes(tf, oracle):
    """A short polynomial at 2^23 + 5 points stays on the Horner route, which walks the points in slabs of 2^22 (a launch is
    limited to 2^32 - 1 threads); sampled points across the slab boundaries against the oracle, two polynomials in the batch."""
    import torch

    m = (1 << 23) + 5
    pts = torch.empty(m, dtype=torch.int64, device="cuda")
    tf.device.fill_random(pts, 61)
    c = oracle.fill_random(2 * 1500, 62)   # two polynomials of 1500 coefficients: the workgroup-per-point kernel
    dc = _to_dev(c)
    out = torch.empty(2 * m, dtype=torch.int64, device="cuda")
    lib = tf.lib()
    lib.tf_set_batch_eval_route(1)
    try:
        short = torch.empty(m, dtype=torch.int64, device="cuda")
        tf.device.batch_evaluate(dc[:7], 7, pts, short)   # lane-per-point kernel, 7 coefficients
        for b in range(2):
            tf.device.batch_evaluate(dc[b * 1500:(b + 1) * 1500], 1500, pts, out[b * m:(b + 1) * m])
    finally:
        lib.tf_set_batch_eval_route(0)
    torch.cuda.synchronize()
    hp = _to_host(pts)
    for i in (0, (1 << 22) - 1, 1 << 22, (1 << 23) - 1, 1 << 23, m - 1):
        assert int(_to_host(short[i:i + 1])[0]) == int(oracle.poly_eval(c[:7], int(hp[i]))[0])
        for b in range(2):
            assert int(_to_host(out[b * m + i: b * m + i + 1])[0]) == int(oracle.poly_eval(c[b * 1500:(b + 1) * 1500], int(hp[i]))[0])


def test_poly_goldens_reproduced_by_the_device(tf, oracle):
    """tests/golden/poly_goldens.json through the HIP path: the committed vectors (canonical values) of zerofier, interpolate,
    clean_divide, barycentric_evaluate, fast_coset_evaluate with an XFieldElement offset and Tip5::trace."""
    import json
    import os

    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poly_goldens.json")))["cases"]
    fr = oracle.fill_random

    def pad(p, n_elems, width):  # the device API returns trimmed polynomials: pad back to the fixture's fixed length
        c = p.coefficients
        return np.concatenate([c, np.zeros(n_elems * width - c.size, dtype=np.uint64)])

    for case in cases:
        o, w = case["op"], case.get("width", 1)
        if o == "zerofier":
            got = pad(tf.Polynomial.zerofier(fr(case["n"] * w, case["seed"]), width=w), case["n"] + 1, w)
        elif o == "interpolate":
            got = pad(tf.Polynomial.interpolate(fr(case["n"] * w, case["domain_seed"]), fr(case["n"] * w, case["values_seed"]), width=w), case["n"], w)
        elif o == "barycentric_evaluate":
            got = tf.barycentric_evaluate(fr(case["n"] * w, case["codeword_seed"]), fr(3, case["indeterminate_seed"]), width=w)
        elif o == "clean_divide":
            got = pad(tf.Polynomial(oracle.to_raw(case["dividend"])).clean_divide(tf.Polynomial(fr(case["nb"], case["divisor_seed"]))), case["nq"], 1)
        elif o == "coset_evaluate_xfe_offset":
            got = tf.fast_coset_evaluate(fr(3 * case["n_coeffs"], case["coeffs_seed"]), fr(3, case["offset_seed"]), case["order"], width=3)
        elif o == "tip5_trace":
            got = tf.Tip5.trace_states(fr(16, case["state_seed"]).copy()).reshape(-1)
        else:
            raise AssertionError(o)
        assert [int(v) for v in oracle.to_values(np.asarray(got, dtype=np.uint64).reshape(-1))] == case["out"], o


def test_pyref_goldens_reproduced_by_the_device(tf, oracle):
    """tests/golden/poly_goldens_pyref.json through the HIP path.  The fixture was generated by tests/pyref.py alone -- pure-Python
    schoolbook definitions, no C oracle (tests/golden/make_poly_goldens_pyref.py) -- with explicit canonical inputs and outputs, so
    this is device-vs-independent, not device-vs-oracle: zerofier, interpolate, barycentric_evaluate, fast_coset_evaluate with
    base-field and extension-field offsets, ntt, fast_multiply (BFieldElement and XFieldElement), clean_divide."""
    from tests.test_oracle_kat import pyref_golden_cases, pyref_golden_eval

    def ntt(x, w):
        y = np.array(x, dtype=np.uint64)
        tf.ntt(y, width=w)
        return y

    ops = {"to_raw": oracle.to_raw, "to_values": oracle.to_values,
           "zerofier": lambda r, w: tf.Polynomial.zerofier(r, width=w).coefficients,
           "interpolate": lambda d, v, w: tf.Polynomial.interpolate(d, v, width=w).coefficients,
           "barycentric": lambda cw, x, w: tf.barycentric_evaluate(cw, x, width=w),
           "coset": lambda c, off, order, w: tf.fast_coset_evaluate(c, off, order, width=w),
           "coset_xoff": lambda c, off, order: tf.fast_coset_evaluate(c, off, order, width=3),
           "ntt": ntt,
           "multiply": lambda a, b, w: tf.fast_multiply(a, b, width=w),
           "clean_divide": lambda a, b: tf.Polynomial(a).clean_divide(tf.Polynomial(b)).coefficients}
    cases = pyref_golden_cases()
    assert len(cases) == 49
    for case in cases:
        assert pyref_golden_eval(case, ops) == case["out"], (case["op"], case.get("width"), case.get("n"))


def test_zerofier_tree_handle_serves_concurrent_host_threads(tf, oracle):
    """One prepared tree, four host threads, each on its own stream, evaluating and interpolating at the same time (ctypes releases
    the GIL during the calls; the first interpolation takes the handle's lock to compute the weights): every thread gets the
    words the single-threaded calls give."""
    import threading

    import torch

    n, rounds = 1 << 13, 6
    dom = torch.empty(n, dtype=torch.int64, device="cuda")
    tf.device.fill_random(dom, 71)
    polys = [torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(4)]
    for k, p in enumerate(polys):
        tf.device.fill_random(p, 72 + k)
    want_vals = []
    for p in polys:
        v = torch.empty(n, dtype=torch.int64, device="cuda")
        tf.device.batch_evaluate(p, n, dom, v)
        want_vals.append(v)
    torch.cuda.synchronize()
    errors = []
    with tf.device.ZerofierTree(dom) as tree:
        def worker(k):
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    for _ in range(rounds):
                        v = torch.empty(n, dtype=torch.int64, device="cuda")
                        b = torch.empty(n, dtype=torch.int64, device="cuda")
                        tree.batch_evaluate(polys[k], n, v, stream=st)
                        tree.interpolate(v, b, stream=st)
                        st.synchronize()
                        if not torch.equal(v, want_vals[k]) or not torch.equal(b, polys[k]):
                            errors.append(k)
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert not errors, errors


@pytest.mark.gpu
def test_concurrent_tree_walks_stress_gives_single_threaded_words():
    """Six host threads on their own streams walking prepared trees of 2^8 .. 2^14 points for a few seconds
    (tools/stress_threads.py, a fresh process).  Round 3 found wrong words here about once in 10^4 calls: the runtime's
    hipMemPoolReuseFollowEventDependencies policy handed a stream blocks another stream was still using once the two had
    exchanged a scratch block (profiles/r03_pool_reuse.txt); the library's temporaries now come from a pool with that policy off."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_threads.py"), "6", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all words match" in r.stdout, (r.stdout[-600:], r.stderr[-600:])


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [7, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("rows", [1, 3])
def test_one_launch_per_level_walks_match_the_separate_kernels(tf, oracle, log_n, rows):
    """tree_down_level_kernel / tree_up_level_kernel (a whole level of the walk in one launch, round 3) against the same walk
    with TF_TREE_NO_LEVEL-style separate kernels is covered by the switch matrix; here: against the oracle's Horner values at
    sampled points and the round trip, for every level size the kernels are instantiated for (2d = 64 .. 4096), one and
    several rows / units per call."""
    import torch

    n = 1 << log_n
    dom = oracle.fill_random(n, 4100 + log_n)
    coeffs = oracle.fill_random(rows * n, 4200 + log_n)
    d_dom, d_c = _to_dev(dom), _to_dev(coeffs)
    vals = torch.empty(rows * n, dtype=torch.int64, device="cuda")
    back = torch.empty(rows * n, dtype=torch.int64, device="cuda")
    with tf.device.ZerofierTree(d_dom) as tree:
        tree.batch_evaluate(d_c, n, vals, batch=rows)
        tree.interpolate(vals, back, rows=rows)
    torch.cuda.synchronize()
    got = _to_host(vals).reshape(rows, n)
    for r in range(rows):
        for i in sorted({0, 1, n // 3, n - 1}):
            assert int(got[r, i]) == int(oracle.poly_eval(coeffs[r * n:(r + 1) * n], int(dom[i]))[0]), (r, i)
    assert np.array_equal(_to_host(back), coeffs)
    # several units walking together (a polynomial longer than the point count) through the one-shot call
    long_c = oracle.fill_random(4 * n, 4300 + log_n)
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    tf.lib().tf_set_batch_eval_route(2)
    try:
        tf.device.batch_evaluate(_to_dev(long_c), 4 * n, d_dom, out)
        torch.cuda.synchronize()
    finally:
        tf.lib().tf_set_batch_eval_route(0)
    got = _to_host(out)
    for i in sorted({0, n // 2, n - 1}):
        assert int(got[i]) == int(oracle.poly_eval(long_c, int(dom[i]))[0]), i


@pytest.mark.parametrize("n_coeffs,n_points,batch", [(0, 3, 1), (1, 1, 1), (7, 5, 3), (1024, 2, 2), (5000, 9, 4), (1 << 16, 3, 2)])
def test_base_field_polynomials_at_extension_field_points(tf, oracle, n_coeffs, n_points, batch):
    """Polynomial<BFieldElement>::evaluate::<XFieldElement, XFieldElement> (math/polynomial.rs:309-320), batched over polynomials
    and points: against the oracle's Horner on the lifted coefficients; a point in the base field gives the base-field value
    lifted (the doc example :296-307)."""
    import torch

    c = oracle.fill_random(max(1, n_coeffs * batch), 2300 + n_coeffs)[: n_coeffs * batch]
    pts = oracle.fill_random(3 * n_points, 2301 + n_points)
    pts[1:3] = 0  # the first point is a lifted base-field element
    dc = _to_dev(c) if c.size else torch.empty(0, dtype=torch.int64, device="cuda")
    out = torch.empty(3 * batch * n_points, dtype=torch.int64, device="cuda")
    tf.device.evaluate_bfe_at_xfe(dc, n_coeffs, _to_dev(pts), out, batch=batch)
    torch.cuda.synchronize()
    got = _to_host(out).reshape(batch, n_points, 3)
    for b in range(batch):
        cb = c[b * n_coeffs:(b + 1) * n_coeffs]
        lifted = np.zeros(3 * n_coeffs, dtype=np.uint64)
        lifted[0::3] = cb
        for i in sorted({0, n_points // 2, n_points - 1}):
            assert np.array_equal(got[b, i], oracle.poly_eval_xfe_point(lifted, pts[3 * i: 3 * i + 3])), (b, i)
        if n_coeffs:
            assert int(got[b, 0, 0]) == int(oracle.poly_eval(cb, int(pts[0]))[0]) and not got[b, 0, 1:].any()
    if batch == 1 or n_coeffs == 7:
        p = tf.Polynomial(c[:n_coeffs])
        assert np.array_equal(p.evaluate_at_xfe_points(pts).reshape(n_points, 3), got[0])
    # the doc example (:296-307): 2 + 5 x + 12 x^2 is 19 at 1 and, evaluated into the extension field at 2, xfe!(60)
    ex = tf.Polynomial(oracle.to_raw([2, 5, 12]))
    at = np.array([oracle.bfe_new(1), 0, 0, oracle.bfe_new(2), 0, 0], dtype=np.uint64)
    assert [int(v) for v in oracle.to_values(ex.evaluate_at_xfe_points(at))] == [19, 0, 0, 60, 0, 0]


@pytest.mark.parametrize("width,n_points,n_coeffs,batch", [(1, 700, 2500, 5), (1, 3000, 9000, 3), (1, 600, 40000, 2), (3, 300, 1000, 4), (3, 1100, 2048, 3)])
def test_many_polynomials_and_chunks_walk_the_tree_together(tf, oracle, width, n_points, n_coeffs, batch):
    """All chunks of all polynomials of a call ("units") go down the tree in one walk per slab of units: `batch` polynomials longer
    than the padded point count through a prepared tree, and coset_extrapolate with the tree route forced, against Horner and the oracle."""
    import torch

    d = oracle.fill_random(n_points * width, 2400 + n_points)
    c = oracle.fill_random(batch * n_coeffs * width, 2401 + n_coeffs)
    dd, dc = _to_dev(d), _to_dev(c)
    out = torch.empty(batch * n_points * width, dtype=torch.int64, device="cuda")
    with tf.device.ZerofierTree(dd, width=width) as tree:
        tree.batch_evaluate(dc, n_coeffs, out, batch=batch)
    torch.cuda.synchronize()
    got = _to_host(out).reshape(batch, n_points, width)
    for b in range(batch):
        one = torch.empty(n_points * width, dtype=torch.int64, device="cuda")
        tf.lib().tf_set_batch_eval_route(1)
        try:
            tf.device.batch_evaluate(dc[b * n_coeffs * width:(b + 1) * n_coeffs * width], n_coeffs, dd, one, width=width)
        finally:
            tf.lib().tf_set_batch_eval_route(0)
        torch.cuda.synchronize()
        assert np.array_equal(got[b].reshape(-1), _to_host(one)), b
    cb = c[(batch - 1) * n_coeffs * width:]
    i = n_points - 1
    want = oracle.poly_eval(cb, int(d[i])) if width == 1 else oracle.poly_eval_xfe_point(cb, d[3 * i: 3 * i + 3])
    assert np.array_equal(got[batch - 1, i], np.asarray(want).reshape(-1))
    # coset_extrapolate: `batch` codewords of 2^12 at the same points, tree route forced vs Horner
    n = 1 << 12
    cw = oracle.fill_random(batch * n * width, 2402)
    off = oracle.bfe_new(7)
    res = []
    for route in (2, 1):
        tf.lib().tf_set_batch_eval_route(route)
        try:
            res.append(tf.Polynomial.batch_coset_extrapolate(off, n, cw, d, width=width))
        finally:
            tf.lib().tf_set_batch_eval_route(0)
    assert np.array_equal(res[0], res[1])


@pytest.mark.parametrize("nq,nb,batch", [(5, 3, 4), (300, 700, 7), (3000, 1100, 3), (1 << 15, 1 << 14, 5)])
def test_many_dividends_over_one_divisor(tf, oracle, nq, nb, batch):
    """tf_poly_clean_divide_many_bfe: `batch` products q_k * b divided by the same b give the q_k back (the quotients of a table of
    numerators over one zerofier); a shorter dividend is zero padded; one unclean row fails the call."""
    import torch

    b = oracle.fill_random(nb, 2500 + nb)
    qs = [oracle.fill_random(nq - (k % 2), 2501 + k) for k in range(batch)]   # every other quotient one coefficient shorter
    na = nq + nb - 1
    rows = []
    for q in qs:
        a = oracle.poly_mul(q, b)
        rows.append(np.concatenate([a, np.zeros(na - a.size, dtype=np.uint64)]))
    a_all = np.ascontiguousarray(np.concatenate(rows))
    out = torch.empty(batch * nq, dtype=torch.int64, device="cuda")
    tf.device.clean_divide_many(_to_dev(a_all), na, _to_dev(b), out, batch)
    torch.cuda.synchronize()
    got = _to_host(out).reshape(batch, nq)
    for k, q in enumerate(qs):
        assert np.array_equal(got[k, : q.size], q) and not got[k, q.size:].any(), k
    bad = a_all.copy()
    bad[(batch - 1) * na + 1] ^= np.uint64(1)
    with pytest.raises(tf.NttPanic) as e:
        tf.device.clean_divide_many(_to_dev(bad), na, _to_dev(b), out, batch)
    assert e.value.code == 16
    with pytest.raises(ValueError):
        tf.device.clean_divide_many(_to_dev(a_all), na, _to_dev(b), out[:-1], batch)


@pytest.mark.parametrize("width,na,nb,batch", [(1, 3, 2, 3), (1, 5, 9, 2), (1, 700, 300, 5), (1, 40000, 25000, 3), (3, 6, 4, 2), (3, 2000, 900, 4)])
def test_many_polynomials_times_one_shared_polynomial(tf, oracle, width, na, nb, batch):
    """tf_poly_mul_shared_*_dev: `batch` polynomials times ONE polynomial (its transform computed once and broadcast) equal the
    products the oracle's fast_multiply gives one by one (math/polynomial.rs:900-932)."""
    import torch

    a = oracle.fill_random(batch * na * width, 2600 + na)
    b = oracle.fill_random(nb * width, 2601 + nb)
    n_out = na + nb - 1
    out = torch.empty(batch * n_out * width, dtype=torch.int64, device="cuda")
    tf.device.poly_mul_shared(_to_dev(a), na, _to_dev(b), out, batch, width=width)
    torch.cuda.synchronize()
    got = _to_host(out).reshape(batch, -1)
    for k in range(batch):
        assert np.array_equal(got[k], oracle.poly_mul(a[k * na * width:(k + 1) * na * width], b, width=width)), k
    with pytest.raises(ValueError):
        tf.device.poly_mul_shared(_to_dev(a), na, _to_dev(b), out[:-1], batch, width=width)


def test_out_of_domain_row_and_quotients_compose(tf, oracle):
    """The callers of the second half of round 2 chained the way a prover uses them, all device-resident: 8 base-field columns of
    degree < 2^12; their out-of-domain row at an extension-field point through barycentric_evaluate on the trace-domain codewords
    equals evaluate on the coefficients; the quotients (f_k(X) - f_k(x0)) / (X - x0) for a base-field x0 come from ONE
    clean_divide_many call, and multiplying them back by (X - x0) (one shared factor) returns the numerators."""
    import torch

    n, cols = 1 << 12, 8
    coeffs = torch.empty(cols * n, dtype=torch.int64, device="cuda")
    tf.device.fill_random(coeffs, 81)
    cw = coeffs.clone()
    tf.device.ntt_(cw, n, batch=cols)                       # the columns' values on the subgroup <w_n>
    x = oracle.fill_random(3, 82)                           # out-of-domain point in the extension field
    row_b = torch.empty(3 * cols, dtype=torch.int64, device="cuda")
    tf.device.barycentric_evaluate(cw, n, x, row_b, batch=cols)
    row_e = torch.empty(3 * cols, dtype=torch.int64, device="cuda")
    tf.device.evaluate_bfe_at_xfe(coeffs, n, _to_dev(x), row_e, batch=cols)
    torch.cuda.synchronize()
    assert torch.equal(row_b, row_e)
    # quotients at a base-field point
    x0 = oracle.bfe_new(123456789)
    h = _to_host(coeffs).reshape(cols, n)
    f_x0 = [int(oracle.poly_eval(h[k], x0)[0]) for k in range(cols)]
    numer = h.copy()
    for k in range(cols):
        numer[k, 0] = oracle.bfe_sub(int(numer[k, 0]), f_x0[k])            # f_k - f_k(x0): divisible by X - x0
    divisor = np.array([oracle.bfe_neg(x0), oracle.bfe_new(1)], dtype=np.uint64)
    q = torch.empty(cols * (n - 1), dtype=torch.int64, device="cuda")
    tf.device.clean_divide_many(_to_dev(np.ascontiguousarray(numer.reshape(-1))), n, _to_dev(divisor), q, cols)
    back = torch.empty(cols * n, dtype=torch.int64, device="cuda")
    tf.device.poly_mul_shared(q, n - 1, _to_dev(divisor), back, cols)
    torch.cuda.synchronize()
    assert np.array_equal(_to_host(back).reshape(cols, n), numer)
    qh = _to_host(q).reshape(cols, n - 1)
    assert np.array_equal(qh[3], oracle.naive_divide(numer[3], divisor)[0])
    # a numerator that does not vanish at x0 is reported as unclean
    with pytest.raises(tf.NttPanic) as e:
        tf.device.clean_divide_many(coeffs, n, _to_dev(divisor), q, cols)
    assert e.value.code == 16


# ---- enqueue-and-return variants (tf_*_dev_async): the panic cases go to a status word in HBM, nothing blocks the host ----------
def test_async_chain_interpolate_divide_evaluate_never_synchronises(tf, oracle):
    """interpolate -> clean_divide -> batch_evaluate (+ a tree built and used asynchronously) enqueued on ONE stream behind a long
    transform: the stream is still busy when the last call returns (torch's Stream.query() = hipStreamQuery), i.e. none of the
    calls waited for the device; same words as the blocking entry points, status 0.  Then the three panic cases of the reference
    (repeated point traits.rs:106; unclean division polynomial.rs:2410; the repeated point again through a tree handle) arrive
    as status codes 12 / 16 / 12."""
    import torch

    n = 1 << 12
    dom = torch.empty(n, dtype=torch.int64, device="cuda")
    vals = torch.empty(n, dtype=torch.int64, device="cuda")
    div = torch.empty(n // 4, dtype=torch.int64, device="cuda")
    tf.device.fill_random(dom, 4101)
    tf.device.fill_random(vals, 4102)
    tf.device.fill_random(div, 4103)
    # blocking reference run (also warms every table cache, whose first build synchronises)
    coeffs = torch.empty(n, dtype=torch.int64, device="cuda")
    tf.device.interpolate(dom, vals, coeffs)
    prod = torch.empty(n + n // 4 - 1, dtype=torch.int64, device="cuda")
    tf.device.poly_mul(coeffs, n, div, n // 4, prod)
    quot = torch.empty(n, dtype=torch.int64, device="cuda")
    tf.device.clean_divide(prod, div, quot)
    ev = torch.empty(n, dtype=torch.int64, device="cuda")
    tf.device.batch_evaluate(quot, n, dom, ev)
    with tf.device.ZerofierTree(dom) as t0:
        c_tree = torch.empty(n, dtype=torch.int64, device="cuda")
        t0.interpolate(vals, c_tree)
    big = torch.empty(256 << 20, dtype=torch.int64, device="cuda")
    tf.device.fill_random(big, 4104)
    tf.device.ntt_(big, 1 << 20, batch=256)
    torch.cuda.synchronize()
    assert torch.equal(quot, coeffs) and torch.equal(ev, vals) and torch.equal(c_tree, coeffs)

    s = torch.cuda.Stream()
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    c2, q2, e2, c3 = (torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(4))
    p2 = torch.empty_like(prod)
    with torch.cuda.stream(s):
        for _ in range(20):                                   # ~40 ms of queued work in front of the chain
            tf.device.ntt_(big, 1 << 20, batch=256, stream=s)
        tf.device.interpolate(dom, vals, c2, stream=s, status=status)
        tf.device.poly_mul(c2, n, div, n // 4, p2, stream=s)
        tf.device.clean_divide(p2, div, q2, stream=s, status=status)
        tf.device.batch_evaluate(q2, n, dom, e2, stream=s)
        tree = tf.device.ZerofierTree(dom, stream=s, asynchronous=True)
        tree.interpolate(vals, c3, stream=s, status=status)
        pending = not s.query()
    s.synchronize()
    assert pending, "an entry point of the chain waited for the device"
    assert int(status.item()) == 0
    assert torch.equal(c2, coeffs) and torch.equal(q2, coeffs) and torch.equal(e2, vals) and torch.equal(c3, coeffs)
    tree.close()

    # the panic cases, asynchronously
    bad_dom = dom.clone()
    bad_dom[n - 1] = bad_dom[5]
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    tf.device.interpolate(bad_dom, vals, c2, status=st)
    torch.cuda.synchronize()
    assert int(st.item()) == 12                               # TF_ERR_INVERSE_OF_ZERO
    st.zero_()
    p2.copy_(prod)
    p2[0] += 1                                                # no longer a multiple of the divisor
    tf.device.clean_divide(p2, div, q2, status=st)
    torch.cuda.synchronize()
    assert int(st.item()) == 16                               # TF_ERR_DIVISION_NOT_CLEAN
    tf.device.interpolate(bad_dom, vals, c2, status=st)       # the first error stays
    torch.cuda.synchronize()
    assert int(st.item()) == 16
    st.zero_()
    with tf.device.ZerofierTree(bad_dom, asynchronous=True) as bt:
        bt.interpolate(vals, c2, status=st)
        torch.cuda.synchronize()
        assert int(st.item()) == 12
        st.zero_()
        bt.interpolate(vals, c2, status=st)                   # weights already there: the handle still reports them bad
        torch.cuda.synchronize()
        assert int(st.item()) == 12
        # ... and so does a BLOCKING call on the same handle: the asynchronous first call left its verdict on the device only; the
        # blocking path reads it back once instead of returning TF_OK with garbage coefficients
        for _ in range(2):
            with pytest.raises(tf.TwentyFirstError) as ei:
                bt.interpolate(vals, c2)
            assert ei.value.code == 12
    with tf.device.ZerofierTree(dom, asynchronous=True) as gt:  # a good domain: async weights, then the blocking call succeeds
        gt.interpolate(vals, c3, status=st)
        gt.interpolate(vals, c2)
        torch.cuda.synchronize()
        assert int(st.item()) == 12 or int(st.item()) == 0       # (st still holds the sticky 12 from above unless it was cleared)
        assert torch.equal(c2, coeffs) and torch.equal(c3, coeffs)


def test_clean_divide_by_a_divisor_with_a_root_on_the_division_coset(tf, oracle):
    """x^3 - x + 1 is the minimal polynomial of X = x, the offset of the reference's division coset (polynomial.rs:2383): its
    transform has a zero there.  The reference divides by such a small divisor on its naive route (:2360-2364) and succeeds; the
    device repeats the division on the coset (x + 1) * <w> and returns the same quotient as the oracle's long division."""
    P_ = tf.Polynomial
    one, neg1 = oracle.bfe_new(1), oracle.bfe_new((1 << 64) - (1 << 32))
    b = np.array([one, neg1, 0, one], dtype=np.uint64)  # 1 - x + x^3
    for nq in (1, 5, 300, 5000):
        q = oracle.fill_random(nq, 900 + nq)
        a = oracle.poly_mul(q, b)
        assert np.array_equal(P_(a).clean_divide(P_(b)).coefficients, q)
        assert np.array_equal(oracle.clean_divide(a, b, 512), q)  # the reference's route for this divisor: naive_divide
    # ... times another factor, and an unclean dividend still reports so
    b2 = oracle.poly_mul(b, oracle.fill_random(40, 950))
    q = oracle.fill_random(100, 951)
    a = oracle.poly_mul(q, b2)
    assert np.array_equal(P_(a).clean_divide(P_(b2)).coefficients, q)
    a[3] ^= np.uint64(1)
    with pytest.raises(tf.NttPanic) as e:
        P_(a).clean_divide(P_(b2))
    assert e.value.code == 16


def test_release_caches_and_recompute(tf, oracle):
    """tf_release_caches drops the scratch blocks and the large cached tables of the device; the next calls rebuild them and
    give the same words."""
    n, batch = 1 << 16, 4
    x = oracle.fill_random(n * batch, 77)
    want = oracle.ntt(x, batch=batch, threads=4)
    y = x.copy()
    tf.ntt(y, batch=batch)
    assert np.array_equal(y, want)
    ev = tf.fast_coset_evaluate(x[:n], oracle.bfe_new(7), 2 * n)
    assert tf.lib().tf_release_caches() == 0
    y = x.copy()
    tf.ntt(y, batch=batch)
    assert np.array_equal(y, want)
    assert np.array_equal(tf.fast_coset_evaluate(x[:n], oracle.bfe_new(7), 2 * n), ev)
    assert tf.lib().tf_release_caches() == 0


def test_two_pass_plan_over_several_batch_tiles(tf, oracle):
    """tf_set_ntt_tile_bytes smaller than the call: a batch of 2^21-point transforms goes through the two-pass plan in several
    equal tiles (here 5 slices as 2 + 2 + 1 ... the planner evens them out); same words as the oracle, forward and inverse,
    BFieldElement and XFieldElement."""
    lib = tf._lib.lib()
    keep = lib.tf_get_ntt_tile_bytes()
    try:
        for width, batch in ((1, 5), (3, 3)):
            n = 1 << 21
            lib.tf_set_ntt_tile_bytes(2 * n * width * 8)
            x = oracle.fill_random(n * width * batch, 5100 + width)
            y = x.copy()
            tf.ntt(y, width=width, batch=batch)
            assert np.array_equal(y, oracle.ntt(x, width=width, batch=batch, threads=8))
            tf.intt(y, width=width, batch=batch)
            assert np.array_equal(y, x)
    finally:
        lib.tf_set_ntt_tile_bytes(keep)
