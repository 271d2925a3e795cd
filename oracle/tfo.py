"""ctypes binding of the CPU oracle (oracle/libtf_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (twenty-first_amd/) never imports this.

All arrays are numpy uint64 arrays of raw Montgomery words (BFE 1 word, XFE 3, Digest 5).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtf_oracle.so")

P = 0xFFFFFFFF00000001


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "tf_oracle.c")
    hdr = os.path.join(_HERE, "tf_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(f) > os.path.getmtime(_SO) for f in (src, hdr)
    )
    if _SO.endswith("_native.so"):
        return _SO  # built by use_native_build()
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtf_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
build_flags = "-O3 -march=x86-64-v2 (portable build, oracle/Makefile)"


def use_native_build() -> str:
    """bench.py's cpu_baseline leg: time the restatement built with -O3 -march=native ON THE HOST IT RUNS ON.
    Must be called before the first lib(); falls back to the portable build when the compiler is missing.  Returns the flags."""
    global _SO, build_flags
    if _lib is not None:
        return build_flags
    try:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtf_oracle_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _SO = os.path.join(_HERE, "libtf_oracle_native.so")
        build_flags = "-O3 -march=native (built on this host for the timed leg)"
    except Exception:
        pass
    return build_flags


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u64, sz, i32 = C.c_uint64, C.c_size_t, C.c_int
        pu = C.POINTER(C.c_uint64)
        sig = {
            "tfo_montyred": (u64, [u64, u64]),
            "tfo_bfe_new": (u64, [u64]),
            "tfo_bfe_value": (u64, [u64]),
            "tfo_bfe_add": (u64, [u64, u64]),
            "tfo_bfe_sub": (u64, [u64, u64]),
            "tfo_bfe_neg": (u64, [u64]),
            "tfo_bfe_mul": (u64, [u64, u64]),
            "tfo_bfe_mod_pow": (u64, [u64, u64]),
            "tfo_bfe_inverse": (u64, [u64]),
            "tfo_bfe_primitive_root": (u64, [u64]),
            "tfo_xfe_add": (None, [pu, pu, pu]),
            "tfo_xfe_sub": (None, [pu, pu, pu]),
            "tfo_xfe_mul": (None, [pu, pu, pu]),
            "tfo_xfe_mul_bfe": (None, [pu, u64, pu]),
            "tfo_ntt": (i32, [pu, sz, i32]),
            "tfo_intt": (i32, [pu, sz, i32]),
            "tfo_ntt_batch": (i32, [pu, sz, sz, i32, i32, i32]),
            "tfo_poly_scale": (None, [pu, sz, i32, u64]),
            "tfo_coset_evaluate": (i32, [pu, sz, i32, u64, pu, sz]),
            "tfo_coset_evaluate_batch": (i32, [pu, sz, i32, u64, pu, sz, sz, i32]),
            "tfo_coset_interpolate": (i32, [pu, sz, i32, u64, pu]),
            "tfo_poly_eval": (None, [pu, sz, i32, u64, pu]),
            "tfo_tip5_permutation": (None, [pu]),
            "tfo_tip5_permutation_naive": (None, [pu]),
            "tfo_tip5_trace": (None, [pu, pu]),
            "tfo_tip5_mds": (None, [pu, i32]),
            "tfo_tip5_mds_graph_constants": (None, [pu]),
            "tfo_tip5_hash_10": (None, [pu, pu]),
            "tfo_tip5_hash_pair": (None, [pu, pu, pu]),
            "tfo_tip5_hash_varlen": (None, [pu, sz, pu]),
            "tfo_tip5_absorb": (None, [pu, pu]),
            "tfo_tip5_hash_pairs": (None, [pu, pu, sz]),
            "tfo_tip5_hash_varlen_rows": (None, [pu, sz, sz, pu]),
            "tfo_tip5_hash_varlen_rows_par": (None, [pu, sz, sz, pu, C.c_int]),
            "tfo_merkle_build": (i32, [pu, sz, pu]),
            "tfo_merkle_build_par": (i32, [pu, sz, pu, i32, sz]),
            "tfo_merkle_frugal_root": (i32, [pu, sz, pu]),
            "tfo_poly_mul_naive": (None, [pu, sz, pu, sz, i32, pu]),
            "tfo_poly_mul_fast": (i32, [pu, sz, pu, sz, i32, pu]),
            "tfo_auth_structure_indices": (i32, [sz, pu, sz, pu, sz, C.POINTER(C.c_size_t)]),
            "tfo_poly_eval_xfe_point": (None, [pu, sz, pu, pu]),
            "tfo_xfe_inverse": (i32, [pu, pu]),
            "tfo_poly_zerofier": (None, [pu, sz, i32, pu]),
            "tfo_poly_lagrange_interpolate": (i32, [pu, pu, sz, i32, pu]),
            "tfo_poly_naive_divide_bfe": (i32, [pu, sz, pu, sz, pu, pu]),
            "tfo_poly_scale_xfe": (None, [pu, sz, pu]),
            "tfo_barycentric_evaluate": (i32, [pu, sz, i32, pu, pu]),
            "tfo_poly_clean_divide_bfe": (i32, [pu, sz, pu, sz, sz, pu]),
            "tfo_fill_random": (None, [pu, sz, u64]),
            "tfo_fill_random_from": (None, [pu, sz, u64, u64]),
            "tfo_digest_to_hex": (None, [pu, C.c_char_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _arr(x, n=None) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


# ---- scalars -------------------------------------------------------------------------

def bfe_new(v: int) -> int:
    return lib().tfo_bfe_new(v % (1 << 64))


def bfe_value(raw: int) -> int:
    return lib().tfo_bfe_value(raw)


def bfe_add(a, b):
    return lib().tfo_bfe_add(a, b)


def bfe_sub(a, b):
    return lib().tfo_bfe_sub(a, b)


def bfe_neg(a):
    return lib().tfo_bfe_neg(a)


def bfe_mul(a, b):
    return lib().tfo_bfe_mul(a, b)


def bfe_mod_pow(a, e):
    return lib().tfo_bfe_mod_pow(a, e)


def bfe_inverse(a):
    return lib().tfo_bfe_inverse(a)


def primitive_root(n: int) -> int:
    return lib().tfo_bfe_primitive_root(n)


def to_raw(values) -> np.ndarray:
    """canonical values -> raw Montgomery words (vectorised through python ints; small inputs)."""
    flat = [bfe_new(int(v)) for v in np.asarray(values, dtype=object).ravel()]
    return np.array(flat, dtype=np.uint64).reshape(np.shape(values))


def to_values(raw) -> np.ndarray:
    flat = [bfe_value(int(v)) for v in np.asarray(raw, dtype=np.uint64).ravel()]
    return np.array(flat, dtype=np.uint64).reshape(np.shape(raw))


def xfe_mul(a, b):
    a, b = _arr(a, 3), _arr(b, 3)
    out = np.zeros(3, dtype=np.uint64)
    lib().tfo_xfe_mul(_p(a), _p(b), _p(out))
    return out


def xfe_add(a, b):
    a, b = _arr(a, 3), _arr(b, 3)
    out = np.zeros(3, dtype=np.uint64)
    lib().tfo_xfe_add(_p(a), _p(b), _p(out))
    return out


def xfe_sub(a, b):
    a, b = _arr(a, 3), _arr(b, 3)
    out = np.zeros(3, dtype=np.uint64)
    lib().tfo_xfe_sub(_p(a), _p(b), _p(out))
    return out


# ---- NTT -----------------------------------------------------------------------------

class OraclePanic(Exception):
    """The reference would panic / return an error here; .code is the oracle's return code."""

    def __init__(self, code):
        super().__init__(f"oracle error code {code}")
        self.code = code


def ntt(x, width: int = 1, inverse: bool = False, batch: int = 1, threads: int = 1) -> np.ndarray:
    """Returns a transformed copy.  x holds batch * n * width words."""
    x = _arr(x).copy().reshape(-1)
    assert x.size % (batch * width) == 0
    n = x.size // (batch * width) if batch else 0
    rc = lib().tfo_ntt_batch(_p(x), n, batch, width, int(inverse), threads)
    if rc:
        raise OraclePanic(rc)
    return x


def intt(x, width: int = 1, batch: int = 1, threads: int = 1) -> np.ndarray:
    return ntt(x, width=width, inverse=True, batch=batch, threads=threads)


def ntt_raw_len(x: np.ndarray, n: int, width: int = 1, inverse: bool = False):
    """Call the single-slice entry point with an explicit length (for error-path tests)."""
    x = _arr(x).copy()
    fn = lib().tfo_intt if inverse else lib().tfo_ntt
    rc = fn(_p(x), n, width)
    if rc:
        raise OraclePanic(rc)
    return x


def coset_evaluate(coeffs, offset_raw: int, order: int, width: int = 1) -> np.ndarray:
    c = _arr(coeffs).reshape(-1)
    n_coeffs = c.size // width
    out = np.zeros(order * width, dtype=np.uint64)
    rc = lib().tfo_coset_evaluate(_p(c) if c.size else _p(np.zeros(1, np.uint64)), n_coeffs, width, offset_raw, _p(out) if out.size else _p(np.zeros(1, np.uint64)), order)
    if rc:
        raise OraclePanic(rc)
    return out


def coset_evaluate_batch(coeffs, offset_raw: int, order: int, batch: int, width: int = 1, threads: int = 1) -> np.ndarray:
    """`batch` contiguous polynomials of equal length, one polynomial per thread."""
    c = _arr(coeffs).reshape(-1)
    assert batch > 0 and c.size % (batch * width) == 0
    n_coeffs = c.size // (batch * width)
    out = np.zeros(batch * order * width, dtype=np.uint64)
    rc = lib().tfo_coset_evaluate_batch(_p(c), n_coeffs, width, offset_raw, _p(out), order, batch, threads)
    if rc:
        raise OraclePanic(rc)
    return out


def coset_interpolate(values, offset_raw: int, width: int = 1) -> np.ndarray:
    v = _arr(values).reshape(-1)
    n = v.size // width
    out = np.zeros_like(v)
    rc = lib().tfo_coset_interpolate(_p(v), n, width, offset_raw, _p(out))
    if rc:
        raise OraclePanic(rc)
    return out


def poly_eval(coeffs, point_raw: int, width: int = 1) -> np.ndarray:
    c = _arr(coeffs).reshape(-1)
    out = np.zeros(width, dtype=np.uint64)
    lib().tfo_poly_eval(_p(c), c.size // width, width, point_raw, _p(out))
    return out


def poly_scale(coeffs, alpha_raw: int, width: int = 1) -> np.ndarray:
    c = _arr(coeffs).copy().reshape(-1)
    lib().tfo_poly_scale(_p(c), c.size // width, width, alpha_raw)
    return c


# ---- Tip5 ----------------------------------------------------------------------------

def tip5_permutation(state, naive: bool = False) -> np.ndarray:
    s = _arr(state, 16).copy()
    (lib().tfo_tip5_permutation_naive if naive else lib().tfo_tip5_permutation)(_p(s))
    return s


def tip5_trace(state):
    """Tip5::trace (tip5/mod.rs:538-548): (6 x 16 trace, permuted state)."""
    s = _arr(state, 16).copy()
    t = np.zeros(96, dtype=np.uint64)
    lib().tfo_tip5_trace(_p(s), _p(t))
    return t.reshape(6, 16), s


def tip5_mds(state, method: int) -> np.ndarray:
    """One MDS layer on 16 raw words.  method 0: plain circulant sum; 1: Tip5::mds_cyclomul restatement (tip5/mod.rs:753-1019);
    2: the shape of mds_generated / generated_function (:210-506) -- what tfo_tip5_permutation runs."""
    s = _arr(state, 16).copy()
    lib().tfo_tip5_mds(_p(s), method)
    return s


def tip5_mds_graph_constants():
    out = np.zeros(2, dtype=np.uint64)
    lib().tfo_tip5_mds_graph_constants(_p(out))
    return int(out[0]), int(out[1])


def hash_10(inp) -> np.ndarray:
    i = _arr(inp, 10)
    out = np.zeros(5, dtype=np.uint64)
    lib().tfo_tip5_hash_10(_p(i), _p(out))
    return out


def hash_pair(l, r) -> np.ndarray:
    l, r = _arr(l, 5), _arr(r, 5)
    out = np.zeros(5, dtype=np.uint64)
    lib().tfo_tip5_hash_pair(_p(l), _p(r), _p(out))
    return out


def hash_varlen(inp) -> np.ndarray:
    i = _arr(inp).reshape(-1)
    out = np.zeros(5, dtype=np.uint64)
    buf = i if i.size else np.zeros(1, dtype=np.uint64)
    lib().tfo_tip5_hash_varlen(_p(buf), i.size, _p(out))
    return out


def absorb(state, inp) -> np.ndarray:
    s = _arr(state, 16).copy()
    lib().tfo_tip5_absorb(_p(s), _p(_arr(inp, 10)))
    return s


def hash_pairs(inp) -> np.ndarray:
    i = _arr(inp).reshape(-1)
    count = i.size // 10
    out = np.zeros(count * 5, dtype=np.uint64)
    if count:
        lib().tfo_tip5_hash_pairs(_p(i), _p(out), count)
    return out


def hash_varlen_rows(rows, row_len: int, threads: int = 1) -> np.ndarray:
    r = _arr(rows).reshape(-1)
    n_rows = r.size // row_len if row_len else 0
    out = np.zeros(n_rows * 5, dtype=np.uint64)
    if n_rows and threads > 1:
        lib().tfo_tip5_hash_varlen_rows_par(_p(r), row_len, n_rows, _p(out), threads)
    elif n_rows:
        lib().tfo_tip5_hash_varlen_rows(_p(r), row_len, n_rows, _p(out))
    return out


# ---- Merkle --------------------------------------------------------------------------

def merkle_build(leaves, threads: int = 0, cutoff: int = 512) -> np.ndarray:
    """Full node array (2n digests).  threads=0: sequential_new; >0: par_new restatement."""
    l = _arr(leaves).reshape(-1)
    n = l.size // 5
    nodes = np.zeros(max(10 * n, 1), dtype=np.uint64)
    buf = l if l.size else np.zeros(1, dtype=np.uint64)
    if threads > 0:
        rc = lib().tfo_merkle_build_par(_p(buf), n, _p(nodes), threads, cutoff)
    else:
        rc = lib().tfo_merkle_build(_p(buf), n, _p(nodes))
    if rc:
        raise OraclePanic(rc)
    return nodes[: 10 * n]


def merkle_frugal_root(leaves) -> np.ndarray:
    l = _arr(leaves).reshape(-1)
    n = l.size // 5
    root = np.zeros(5, dtype=np.uint64)
    buf = l if l.size else np.zeros(1, dtype=np.uint64)
    rc = lib().tfo_merkle_frugal_root(_p(buf), n, _p(root))
    if rc:
        raise OraclePanic(rc)
    return root


# ---- "next" rows ---------------------------------------------------------------------

def poly_mul(a, b, width: int = 1, naive: bool = False) -> np.ndarray:
    a, b = _arr(a).reshape(-1), _arr(b).reshape(-1)
    na, nb = a.size // width, b.size // width
    if na == 0 or nb == 0:
        return np.zeros(0, dtype=np.uint64)
    out = np.zeros((na + nb - 1) * width, dtype=np.uint64)
    if naive:
        lib().tfo_poly_mul_naive(_p(a), na, _p(b), nb, width, _p(out))
    else:
        rc = lib().tfo_poly_mul_fast(_p(a), na, _p(b), nb, width, _p(out))
        if rc:
            raise OraclePanic(rc)
    return out


def hadamard(a, b, width: int = 1) -> np.ndarray:
    a, b = _arr(a).reshape(-1), _arr(b).reshape(-1)
    if width == 1:
        return np.array([bfe_mul(int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint64)
    return np.concatenate([xfe_mul(a[3 * i:3 * i + 3], b[3 * i:3 * i + 3]) for i in range(a.size // 3)])


def auth_structure_indices(num_leafs: int, leaf_indices) -> np.ndarray:
    li = _arr(leaf_indices).reshape(-1)
    cap = max(1, li.size * 70)
    out = np.zeros(cap, dtype=np.uint64)
    cnt = C.c_size_t(0)
    buf = li if li.size else np.zeros(1, dtype=np.uint64)
    rc = lib().tfo_auth_structure_indices(num_leafs, _p(buf), li.size, _p(out), cap, C.byref(cnt))
    if rc:
        raise OraclePanic(rc)
    return out[: cnt.value].copy()


def poly_eval_xfe_point(coeffs, point) -> np.ndarray:
    c = _arr(coeffs).reshape(-1)
    pt = _arr(point, 3)
    out = np.zeros(3, dtype=np.uint64)
    buf = c if c.size else np.zeros(3, dtype=np.uint64)
    lib().tfo_poly_eval_xfe_point(_p(buf), c.size // 3, _p(pt), _p(out))
    return out


def xfe_inverse(a) -> np.ndarray:
    out = np.zeros(3, dtype=np.uint64)
    if lib().tfo_xfe_inverse(_p(_arr(a, 3)), _p(out)):
        raise OraclePanic(12)
    return out


def zerofier(roots, width: int = 1) -> np.ndarray:
    """Polynomial::smart_zerofier (math/polynomial.rs:1462-1475): n + 1 coefficients."""
    r = _arr(roots).reshape(-1)
    n = r.size // width
    out = np.zeros((n + 1) * width, dtype=np.uint64)
    buf = r if r.size else np.zeros(width, dtype=np.uint64)
    lib().tfo_poly_zerofier(_p(buf), n, width, _p(out))
    return out


def lagrange_interpolate(domain, values, width: int = 1) -> np.ndarray:
    """Polynomial::lagrange_interpolate (math/polynomial.rs:1565-1606): n coefficients, untrimmed."""
    d, v = _arr(domain).reshape(-1), _arr(values).reshape(-1)
    assert d.size == v.size
    n = d.size // width
    out = np.zeros(max(n, 1) * width, dtype=np.uint64)
    rc = lib().tfo_poly_lagrange_interpolate(_p(d if d.size else out), _p(v if v.size else out), n, width, _p(out))
    if rc:
        raise OraclePanic(12 if rc == 1 else 14)
    return out[: n * width]


def _trim(c: np.ndarray) -> np.ndarray:
    n = c.size
    while n and not c[n - 1]:
        n -= 1
    return c[:n]


def naive_divide(a, b):
    """Polynomial::naive_divide (math/polynomial.rs:552-600) over BFieldElement: (quotient, remainder), trimmed."""
    a, b = _trim(_arr(a).reshape(-1)), _trim(_arr(b).reshape(-1))
    quot = np.zeros(max(a.size - b.size + 1, 1), dtype=np.uint64)
    rem = np.zeros(max(a.size, 1), dtype=np.uint64)
    pad = np.zeros(1, dtype=np.uint64)
    rc = lib().tfo_poly_naive_divide_bfe(_p(a if a.size else pad), a.size, _p(b if b.size else pad), b.size, _p(quot), _p(rem))
    if rc:
        raise OraclePanic(15)
    return _trim(quot[: max(a.size - b.size + 1, 0)]), _trim(rem[: a.size])


def clean_divide(a, b, cutoff: int = 1 << 9) -> np.ndarray:
    """Polynomial::<BFieldElement>::clean_divide (math/polynomial.rs:2358-2411); `cutoff` is CLEAN_DIVIDE_CUTOFF_THRESHOLD
    (0 = the cfg(test) value: always the coset route).  Trimmed quotient."""
    a, b = _trim(_arr(a).reshape(-1)), _trim(_arr(b).reshape(-1))
    out = np.zeros(max(a.size - b.size + 1, 1), dtype=np.uint64)
    pad = np.zeros(1, dtype=np.uint64)
    rc = lib().tfo_poly_clean_divide_bfe(_p(a if a.size else pad), a.size, _p(b if b.size else pad), b.size, cutoff, _p(out))
    if rc:
        raise OraclePanic({1: 15, 2: 12, 3: 16, 4: 16}[rc])
    return _trim(out[: max(a.size - b.size + 1, 0)])


def coset_evaluate_xfe_offset(coeffs, offset, order: int) -> np.ndarray:
    """fast_coset_evaluate with S = XFieldElement (math/polynomial.rs:1374-1399): scale by the offset, zero-pad, ntt."""
    c = _arr(coeffs).reshape(-1)
    if c.size // 3 > order:
        raise OraclePanic(6)
    buf = np.zeros(3 * order, dtype=np.uint64)
    buf[: c.size] = c
    lib().tfo_poly_scale_xfe(_p(buf), c.size // 3, _p(_arr(offset, 3)))
    return ntt(buf, width=3)


def coset_interpolate_xfe_offset(values, offset) -> np.ndarray:
    """fast_coset_interpolate with S = XFieldElement (math/polynomial.rs:1907-1918): intt, scale by the inverse offset."""
    v = intt(_arr(values).reshape(-1), width=3)
    lib().tfo_poly_scale_xfe(_p(v), v.size // 3, _p(xfe_inverse(offset)))
    return v


def barycentric_evaluate(codeword, indeterminate, width: int = 1) -> np.ndarray:
    """barycentric_evaluate (math/polynomial.rs:2609-2637); the indeterminate as 1 (BFieldElement, lifted) or 3 raw words; the
    result as an XFieldElement (limbs 1, 2 zero when everything is in the base field)."""
    c = _arr(codeword).reshape(-1)
    x = np.zeros(3, dtype=np.uint64)
    xi = np.asarray(indeterminate, dtype=np.uint64).reshape(-1)
    x[: xi.size] = xi
    out = np.zeros(3, dtype=np.uint64)
    rc = lib().tfo_barycentric_evaluate(_p(c if c.size else out), c.size // width, width, _p(x), _p(out))
    if rc:
        raise OraclePanic(4 if rc == 1 else 12)
    return out


def merkle_from_rows(rows, row_len: int) -> np.ndarray:
    return merkle_build(hash_varlen_rows(rows, row_len))


# ---- helpers -------------------------------------------------------------------------

def fill_random(count: int, seed: int, first_index: int = 0) -> np.ndarray:
    """Elements [first_index, first_index + count) of the counter-based sequence of `seed`."""
    out = np.zeros(count, dtype=np.uint64)
    if count:
        lib().tfo_fill_random_from(_p(out), count, seed, first_index)
    return out


def digest_hex(d) -> str:
    d = _arr(d, 5)
    buf = C.create_string_buffer(81)
    lib().tfo_digest_to_hex(_p(d), buf)
    return buf.value.decode()
