"""Device-pointer API: the same operations on data already resident in HBM.

torch is used only as plumbing -- to own device memory and streams.  Tensors are 1-D contiguous
int64 (or uint64) CUDA tensors holding raw Montgomery words; work is enqueued on torch's current
stream (or the given one) through the tf_*_dev entry points and is NOT synchronised here.
"""
from __future__ import annotations

import ctypes as C

from . import _lib


def _chk(rc, where):
    from . import _check

    _check(rc, where)


def _t(t, name="tensor"):
    import torch

    if not isinstance(t, torch.Tensor) or not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.int64, torch.uint64):
        raise TypeError(f"{name} must be a contiguous CUDA int64/uint64 tensor of raw Montgomery words")
    return t


def _stream(stream):
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _need(cond: bool, what: str) -> None:
    """Every wrapper checks its tensors against (n, batch, width) BEFORE the C ABI sees a raw pointer: a wrong size is a
    ValueError here, never an out-of-bounds HBM access there."""
    if not cond:
        raise ValueError(what)


def _width(width: int) -> int:
    _need(width in (1, 3), "width must be 1 (BFieldElement) or 3 (XFieldElement)")
    return width


def fill_random(out, seed: int, first_index: int = 0, stream=None) -> None:
    """Synthetic inputs: out[i] = BFieldElement::new(splitmix64(seed ^ (first_index + i)) mod p), generated on
    the device; the oracle's tfo.fill_random(count, seed) is the same sequence."""
    out = _t(out, "out")
    _chk(_lib.lib().tf_debug_fill_random_dev(_p(out), out.numel(), C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(first_index),
                                             _stream(stream)), "fill_random")


def ntt_(x, n: int, batch: int = 1, width: int = 1, inverse: bool = False, stream=None) -> None:
    """In place on device: `batch` slices of n elements (math/ntt.rs:67-82, :109-125)."""
    x = _t(x, "x")
    _width(width)
    if x.numel() != n * batch * width:
        raise ValueError("tensor size is not batch * n * width")
    fn = _lib.lib().tf_ntt_bfe_dev if width == 1 else _lib.lib().tf_ntt_xfe_dev
    _chk(fn(_p(x), n, batch, int(inverse), _stream(stream)), "intt" if inverse else "ntt")


def coset_evaluate(coeffs, n_coeffs: int, offset_raw: int, out, order: int, batch: int = 1, width: int = 1, stream=None) -> None:
    """math/polynomial.rs:1374-1399 on device buffers (out: batch * order * width words)."""
    coeffs, out = _t(coeffs, "coeffs"), _t(out, "out")
    _width(width)
    if coeffs.numel() != n_coeffs * batch * width or out.numel() != order * batch * width:
        raise ValueError("buffer sizes do not match n_coeffs/order/batch/width")
    fn = _lib.lib().tf_coset_eval_bfe_dev if width == 1 else _lib.lib().tf_coset_eval_xfe_dev
    _chk(fn(_p(coeffs), n_coeffs, C.c_uint64(offset_raw), _p(out), order, batch, _stream(stream)), "fast_coset_evaluate")


def tip5_permute_(states, stream=None) -> None:
    states = _t(states, "states")
    _need(states.numel() % 16 == 0, "states must hold 16 words per Tip5 state")
    _chk(_lib.lib().tf_tip5_permute_dev(_p(states), states.numel() // 16, _stream(stream)), "Tip5::permutation")


def tip5_trace_(states, trace, stream=None) -> None:
    """Tip5::trace (tip5/mod.rs:538-548) on device buffers: trace = count x 6 x 16 words, states permuted in place."""
    states, trace = _t(states, "states"), _t(trace, "trace")
    _need(states.numel() % 16 == 0 and trace.numel() == 6 * states.numel(), "trace must hold 6 x 16 words per Tip5 state")
    _chk(_lib.lib().tf_tip5_trace_dev(_p(states), _p(trace), states.numel() // 16, _stream(stream)), "Tip5::trace")


def tip5_hash_pairs(inp, out, stream=None) -> None:
    inp, out = _t(inp, "in"), _t(out, "out")
    _need(inp.numel() % 10 == 0, "in must hold 10 words per pair of digests")
    count = inp.numel() // 10
    if out.numel() != count * 5:
        raise ValueError("out must hold 5 words per input pair")
    _chk(_lib.lib().tf_tip5_hash_pairs_dev(_p(inp), _p(out), count, _stream(stream)), "Tip5::hash_pair")


def tip5_hash_varlen_rows(rows, row_len: int, out, stream=None) -> None:
    rows, out = _t(rows, "rows"), _t(out, "out")
    _need(out.numel() % 5 == 0, "out must hold 5 words per row")
    n_rows = out.numel() // 5
    _need(row_len >= 0 and rows.numel() == n_rows * row_len, "rows must hold n_rows * row_len words (n_rows = out.numel() / 5)")
    _chk(_lib.lib().tf_tip5_hash_varlen_rows_dev(_p(rows), row_len, n_rows, _p(out), _stream(stream)), "Tip5::hash_varlen")


def merkle_build(leaves, n_leaves: int, nodes_out, batch: int = 1, stream=None) -> None:
    """util_types/merkle_tree.rs:165-212: nodes_out = batch x 2n digests, heap layout."""
    leaves, nodes_out = _t(leaves, "leaves"), _t(nodes_out, "nodes_out")
    if leaves.numel() != batch * n_leaves * 5 or nodes_out.numel() != batch * n_leaves * 10:
        raise ValueError("buffer sizes do not match n_leaves/batch")
    _chk(_lib.lib().tf_merkle_build_dev(_p(leaves), n_leaves, _p(nodes_out), batch, _stream(stream)), "MerkleTree::par_new")


def merkle_root(leaves, n_leaves: int, root_out, batch: int = 1, stream=None) -> None:
    leaves, root_out = _t(leaves, "leaves"), _t(root_out, "root_out")
    _need(leaves.numel() == batch * n_leaves * 5 and root_out.numel() == batch * 5, "buffer sizes do not match n_leaves/batch")
    _chk(_lib.lib().tf_merkle_root_dev(_p(leaves), n_leaves, _p(root_out), batch, _stream(stream)), "MerkleTree::par_frugal_root")


# ---- the callers on either side of the path, kept in HBM --------------------------------

def coset_interpolate(values, n: int, offset_raw: int, out, batch: int = 1, width: int = 1, stream=None) -> None:
    """math/polynomial.rs:1907-1918 on device buffers (out: batch * n * width words; may alias values)."""
    values, out = _t(values, "values"), _t(out, "out")
    _width(width)
    _need(values.numel() == n * batch * width and out.numel() == n * batch * width, "buffer sizes do not match n/batch/width")
    fn = _lib.lib().tf_coset_interpolate_bfe_dev if width == 1 else _lib.lib().tf_coset_interpolate_xfe_dev
    _chk(fn(_p(values), n, C.c_uint64(offset_raw), _p(out), batch, _stream(stream)), "fast_coset_interpolate")


def hadamard(a, b, out, width: int = 1, stream=None) -> None:
    """Pointwise field product (math/polynomial.rs:920-925); out may alias a or b."""
    a, b, out = _t(a, "a"), _t(b, "b"), _t(out, "out")
    _width(width)
    _need(a.numel() % width == 0 and b.numel() == a.numel() and out.numel() == a.numel(), "a, b and out must hold the same number of elements")
    fn = _lib.lib().tf_hadamard_bfe_dev if width == 1 else _lib.lib().tf_hadamard_xfe_dev
    _chk(fn(_p(a), _p(b), _p(out), a.numel() // width, _stream(stream)), "hadamard")


def poly_mul(a, na: int, b, nb: int, out, batch: int = 1, width: int = 1, stream=None) -> None:
    """Polynomial::fast_multiply on device (math/polynomial.rs:900-932): out = batch x (na + nb - 1) coefficients."""
    a, b, out = _t(a, "a"), _t(b, "b"), _t(out, "out")
    _width(width)
    _need(a.numel() == na * batch * width and b.numel() == nb * batch * width, "operand sizes do not match na/nb/batch/width")
    _need(not (na and nb) or out.numel() == (na + nb - 1) * batch * width, "out must hold batch * (na + nb - 1) coefficients")
    fn = _lib.lib().tf_poly_mul_bfe_dev if width == 1 else _lib.lib().tf_poly_mul_xfe_dev
    _chk(fn(_p(a), na, _p(b), nb, _p(out), batch, _stream(stream)), "fast_multiply")


def poly_mul_shared(a, na: int, b, out, batch: int, width: int = 1, stream=None) -> None:
    """`batch` polynomials of na coefficients each times ONE polynomial b (tf_poly_mul_shared_*_dev): out = batch x (na + nb - 1)."""
    a, b, out = _t(a, "a"), _t(b, "b"), _t(out, "out")
    w = _width(width)
    _need(b.numel() % w == 0 and a.numel() == batch * na * w, "a must hold batch * na elements, b whole elements")
    nb = b.numel() // w
    _need(nb >= 1 and out.numel() == batch * (na + nb - 1) * w, "out must hold batch * (na + nb - 1) coefficients")
    fn = _lib.lib().tf_poly_mul_shared_bfe_dev if width == 1 else _lib.lib().tf_poly_mul_shared_xfe_dev
    _chk(fn(_p(a), na, batch, _p(b), nb, _p(out), _stream(stream)), "fast_multiply")


def lde(values, n: int, offset_in_raw: int, out, m: int, offset_out_raw: int, batch: int = 1, width: int = 1, stream=None) -> None:
    """Low-degree extension: interpolate on {offset_in w_n^i}, evaluate on {offset_out w_m^i}; coefficients stay in HBM."""
    values, out = _t(values, "values"), _t(out, "out")
    _width(width)
    _need(values.numel() == n * batch * width and out.numel() == m * batch * width, "buffer sizes do not match n/m/batch/width")
    fn = _lib.lib().tf_lde_bfe_dev if width == 1 else _lib.lib().tf_lde_xfe_dev
    _chk(fn(_p(values), n, C.c_uint64(offset_in_raw), _p(out), m, C.c_uint64(offset_out_raw), batch, _stream(stream)), "lde")


def merkle_from_rows(rows, row_len: int, n_rows: int, nodes_out, batch: int = 1, stream=None) -> None:
    """hash_varlen of every row -> leaf level -> tree, without the leaves leaving HBM."""
    rows, nodes_out = _t(rows, "rows"), _t(nodes_out, "nodes_out")
    _need(rows.numel() == batch * n_rows * row_len and nodes_out.numel() == batch * n_rows * 10, "buffer sizes do not match row_len/n_rows/batch")
    _chk(_lib.lib().tf_merkle_from_rows_dev(_p(rows), row_len, n_rows, _p(nodes_out), batch, _stream(stream)), "MerkleTree::par_new")


def authentication_structure(nodes, num_leafs: int, leaf_indices):
    """util_types/merkle_tree.rs:614-622 from a device-resident node array: returns a (k, 5) numpy array."""
    import numpy as np

    nodes = _t(nodes, "nodes")
    _need(nodes.numel() == num_leafs * 10, "nodes must hold 2 * num_leafs digests")
    li = np.ascontiguousarray(leaf_indices, dtype=np.uint64).reshape(-1)
    cap = max(1, li.size * 66)
    out = np.empty(cap * 5, dtype=np.uint64)
    cnt = C.c_size_t(0)
    rc = _lib.lib().tf_merkle_authentication_structure_dev(_p(nodes), num_leafs, C.c_void_p(li.ctypes.data) if li.size else C.c_void_p(0),
                                                           li.size, C.c_void_p(out.ctypes.data), cap, C.byref(cnt), _stream(None))
    _chk(rc, "MerkleTree::authentication_structure")
    return out[: cnt.value * 5].reshape(-1, 5).copy()


def batch_evaluate(coeffs, n_coeffs: int, points, out, width: int = 1, stream=None) -> None:
    """Polynomial::batch_evaluate (math/polynomial.rs:1840-1878) on device buffers: out[i] = f(points[i])."""
    coeffs, points, out = _t(coeffs, "coeffs"), _t(points, "points"), _t(out, "out")
    if points.numel() != out.numel() or coeffs.numel() != n_coeffs * width:
        raise ValueError("buffer sizes do not match n_coeffs/points/width")
    fn = _lib.lib().tf_poly_batch_evaluate_bfe_dev if width == 1 else _lib.lib().tf_poly_batch_evaluate_xfe_dev
    _chk(fn(_p(coeffs), n_coeffs, _p(points), points.numel() // width, _p(out), _stream(stream)), "batch_evaluate")


def evaluate_bfe_at_xfe(coeffs, n_coeffs: int, points, out, batch: int = 1, stream=None) -> None:
    """Polynomial<BFieldElement>::evaluate with XFieldElement indeterminates (math/polynomial.rs:309-320) on device buffers: `batch`
    base-field polynomials at the XFieldElement points -> out[(b * n_points + i) * 3]."""
    coeffs, points, out = _t(coeffs, "coeffs"), _t(points, "points"), _t(out, "out")
    _need(points.numel() % 3 == 0 and coeffs.numel() == batch * n_coeffs and out.numel() == batch * points.numel(),
          "coeffs = batch * n_coeffs words, points = n_points XFieldElements, out = batch * n_points XFieldElements")
    _chk(_lib.lib().tf_poly_evaluate_bfe_at_xfe_dev(_p(coeffs), n_coeffs, batch, _p(points), points.numel() // 3, _p(out), _stream(stream)), "evaluate")


def _status(status):
    """`status`: a one-element int32 CUDA tensor the *_dev_async entry points report the reference's panic cases through
    (first non-zero tf_status code wins); with it a wrapper enqueues and returns without ever synchronising the stream."""
    import torch

    if not isinstance(status, torch.Tensor) or not status.is_cuda or status.dtype != torch.int32 or status.numel() != 1:
        raise TypeError("status must be a one-element int32 CUDA tensor")
    return C.c_void_p(status.data_ptr())


def clean_divide(a, b, out, stream=None, status=None) -> None:
    """Polynomial::<BFieldElement>::clean_divide (math/polynomial.rs:2358-2411) on device buffers: a, b normalised coefficient
    arrays, out = the na - nb + 1 quotient coefficients.  status: see _status (no host synchronisation)."""
    a, b, out = _t(a, "a"), _t(b, "b"), _t(out, "out")
    _need(a.numel() >= b.numel() and out.numel() == a.numel() - b.numel() + 1, "out must hold na - nb + 1 coefficients")
    if status is not None:
        _chk(_lib.lib().tf_poly_clean_divide_bfe_dev_async(_p(a), a.numel(), _p(b), b.numel(), _p(out), _stream(stream), _status(status)), "clean_divide")
        return
    _chk(_lib.lib().tf_poly_clean_divide_bfe_dev(_p(a), a.numel(), _p(b), b.numel(), _p(out), _stream(stream)), "clean_divide")


def clean_divide_many(a, na: int, b, out, batch: int, stream=None, status=None) -> None:
    """`batch` dividends of na coefficients each over one divisor (tf_poly_clean_divide_many_bfe_dev): out = batch x (na - nb + 1)."""
    a, b, out = _t(a, "a"), _t(b, "b"), _t(out, "out")
    _need(a.numel() == batch * na and na >= b.numel() and out.numel() == batch * (na - b.numel() + 1), "a = batch * na, out = batch * (na - nb + 1) coefficients")
    if status is not None:
        _chk(_lib.lib().tf_poly_clean_divide_many_bfe_dev_async(_p(a), na, batch, _p(b), b.numel(), _p(out), _stream(stream), _status(status)), "clean_divide")
        return
    _chk(_lib.lib().tf_poly_clean_divide_many_bfe_dev(_p(a), na, batch, _p(b), b.numel(), _p(out), _stream(stream)), "clean_divide")


def zerofier(roots, out, width: int = 1, stream=None) -> None:
    """Polynomial::zerofier (math/polynomial.rs:1435-1441) on device buffers: out = the n + 1 coefficients of prod (x - roots[i])."""
    roots, out = _t(roots, "roots"), _t(out, "out")
    _need(roots.numel() % width == 0, "roots must hold whole elements")
    n = roots.numel() // width
    _need(out.numel() == (n + 1) * width, "out must hold n_roots + 1 coefficients")
    fn = _lib.lib().tf_poly_zerofier_bfe_dev if width == 1 else _lib.lib().tf_poly_zerofier_xfe_dev
    _chk(fn(_p(roots), n, _p(out), _stream(stream)), "zerofier")


def interpolate(domain, values, out, rows: int = 1, width: int = 1, stream=None, status=None) -> None:
    """Polynomial::interpolate / batch_fast_interpolate (math/polynomial.rs:1502-1838) on device buffers: `rows` value rows over
    one domain -> rows x n coefficients (untrimmed).  status: see _status (no host synchronisation)."""
    domain, values, out = _t(domain, "domain"), _t(values, "values"), _t(out, "out")
    _need(domain.numel() % width == 0, "domain must hold whole elements")
    n = domain.numel() // width
    _need(values.numel() == rows * n * width and out.numel() == rows * n * width, "values / out must hold rows * n elements")
    if status is not None:
        fn = _lib.lib().tf_poly_interpolate_bfe_dev_async if width == 1 else _lib.lib().tf_poly_interpolate_xfe_dev_async
        _chk(fn(_p(domain), _p(values), n, rows, _p(out), _stream(stream), _status(status)), "interpolate")
        return
    fn = _lib.lib().tf_poly_interpolate_bfe_dev if width == 1 else _lib.lib().tf_poly_interpolate_xfe_dev
    _chk(fn(_p(domain), _p(values), n, rows, _p(out), _stream(stream)), "interpolate")


def barycentric_evaluate(codewords, n: int, indeterminate, out, batch: int = 1, width: int = 1, stream=None) -> None:
    """barycentric_evaluate (math/polynomial.rs:2609-2637) on device buffers: `batch` codewords of n elements at one indeterminate
    (3 raw words, host side) -> out = batch x 3 words."""
    import numpy as np

    codewords, out = _t(codewords, "codewords"), _t(out, "out")
    _need(codewords.numel() == batch * n * _width(width) and out.numel() == 3 * batch, "codewords = batch * n elements, out = batch XFieldElements")
    x = np.zeros(3, dtype=np.uint64)
    xi = np.asarray(indeterminate, dtype=np.uint64).reshape(-1)
    _need(xi.size in (1, 3), "the indeterminate is one raw word or three")
    x[: xi.size] = xi
    fn = _lib.lib().tf_barycentric_evaluate_bfe_dev if width == 1 else _lib.lib().tf_barycentric_evaluate_xfe_dev
    _chk(fn(_p(codewords), n, batch, C.c_void_p(x.ctypes.data), _p(out), _stream(stream)), "barycentric_evaluate")


class ZerofierTree:
    """math/zerofier_tree.rs on device buffers: the tree of a device-resident domain, kept in HBM across calls
    (tf_zerofier_tree_* of include/tf_hip.h).  close() (or the context manager) releases the device memory."""

    def __init__(self, domain, width: int = 1, stream=None, asynchronous: bool = False):
        """asynchronous: return without waiting for the build (tf_zerofier_tree_new_*_dev_async): until the caller synchronises,
        the tree may only be used on the stream it was built on."""
        domain = _t(domain, "domain")
        _need(domain.numel() % _width(width) == 0, "domain must hold whole elements")
        self.width = width
        self.num_points = domain.numel() // width
        self._h = C.c_void_p(0)
        self._free = _lib.lib().tf_zerofier_tree_free
        if asynchronous:
            fn = _lib.lib().tf_zerofier_tree_new_bfe_dev_async if width == 1 else _lib.lib().tf_zerofier_tree_new_xfe_dev_async
        else:
            fn = _lib.lib().tf_zerofier_tree_new_bfe_dev if width == 1 else _lib.lib().tf_zerofier_tree_new_xfe_dev
        _chk(fn(_p(domain), self.num_points, _stream(stream), C.byref(self._h)), "ZerofierTree::new_from_domain")

    def close(self) -> None:
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._free(h)  # bound at construction: still callable while the interpreter shuts down
            h.value = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def zerofier(self, out, stream=None) -> None:
        out = _t(out, "out")
        _need(out.numel() == (self.num_points + 1) * self.width, "out must hold n + 1 coefficients")
        _chk(_lib.lib().tf_zerofier_tree_zerofier_dev(self._h, _p(out), _stream(stream)), "ZerofierTree::zerofier")

    def batch_evaluate(self, coeffs, n_coeffs: int, out, batch: int = 1, stream=None) -> None:
        coeffs, out = _t(coeffs, "coeffs"), _t(out, "out")
        _need(coeffs.numel() == batch * n_coeffs * self.width, "coeffs must hold batch * n_coeffs elements")
        _need(out.numel() == batch * self.num_points * self.width, "out must hold batch * n_points elements")
        _chk(_lib.lib().tf_zerofier_tree_batch_evaluate_dev(self._h, _p(coeffs), n_coeffs, batch, _p(out), _stream(stream)),
             "divide_and_conquer_batch_evaluate")

    def interpolate(self, values, out, rows: int = 1, stream=None, status=None) -> None:
        values, out = _t(values, "values"), _t(out, "out")
        _need(values.numel() == rows * self.num_points * self.width and out.numel() == values.numel(), "values / out must hold rows * n elements")
        if status is not None:
            _chk(_lib.lib().tf_zerofier_tree_interpolate_dev_async(self._h, _p(values), rows, _p(out), _stream(stream), _status(status)), "interpolate")
            return
        _chk(_lib.lib().tf_zerofier_tree_interpolate_dev(self._h, _p(values), rows, _p(out), _stream(stream)), "interpolate")


def coset_extrapolate(offset_raw: int, codewords, n: int, points, out, batch: int = 1, width: int = 1, stream=None) -> None:
    """Polynomial::batch_coset_extrapolate (math/polynomial.rs:2196-2208) on device buffers:
    out[(b * n_points + i) * width] = interpolant_b(points[i])."""
    codewords, points, out = _t(codewords, "codewords"), _t(points, "points"), _t(out, "out")
    n_points = points.numel() // width
    if codewords.numel() != batch * n * width or out.numel() != batch * n_points * width:
        raise ValueError("buffer sizes do not match n/batch/points/width")
    fn = _lib.lib().tf_coset_extrapolate_bfe_dev if width == 1 else _lib.lib().tf_coset_extrapolate_xfe_dev
    _chk(fn(C.c_uint64(offset_raw), _p(codewords), n, batch, _p(points), n_points, _p(out), _stream(stream)), "batch_coset_extrapolate")


def hash_table_rows(table, n_rows: int, n_cols: int, out, width: int = 1, col_stride=None, batch: int = 1, stream=None) -> None:
    """hash_varlen of every row of `batch` column-major tables resident in HBM (column j at table + j * col_stride words)."""
    table, out = _t(table, "table"), _t(out, "out")
    _width(width)
    cs = n_rows * width if col_stride is None else col_stride
    _need(cs >= n_rows * width, "col_stride must be at least n_rows * width words")
    _need(table.numel() >= ((batch * n_cols - 1) * cs + n_rows * width if batch * n_cols else 0), "table is smaller than batch * n_cols columns")
    _need(out.numel() == batch * n_rows * 5, "out must hold 5 words per row and table")
    _chk(_lib.lib().tf_tip5_hash_table_rows_dev(_p(table), n_rows, n_cols, width, cs, _p(out), batch, _stream(stream)), "Tip5::hash_varlen")


def merkle_from_columns(table, n_rows: int, n_cols: int, nodes_out, width: int = 1, col_stride=None, batch: int = 1, stream=None) -> None:
    """Rows of column-major tables -> leaves -> trees (nodes_out: batch x 2 n_rows digests), all in HBM."""
    table, nodes_out = _t(table, "table"), _t(nodes_out, "nodes_out")
    _width(width)
    cs = n_rows * width if col_stride is None else col_stride
    _need(cs >= n_rows * width, "col_stride must be at least n_rows * width words")
    _need(table.numel() >= ((batch * n_cols - 1) * cs + n_rows * width if batch * n_cols else 0), "table is smaller than batch * n_cols columns")
    _need(nodes_out.numel() == batch * n_rows * 10, "nodes_out must hold batch x 2 n_rows digests")
    _chk(_lib.lib().tf_merkle_from_columns_dev(_p(table), n_rows, n_cols, width, cs, _p(nodes_out), batch, _stream(stream)), "MerkleTree::par_new")
