"""twenty_first_amd -- host-side mirror of the reference's API for the hot path, on top of the
C ABI of libtf_hip.so (include/tf_hip.h).  Python is only the harness language here (tests,
bench); the C++ mirror of the same interface is host/twenty_first.hpp.

Names, argument meaning and error behaviour follow the Rust crate (paths relative to
/root/reference/twenty-first/src/):

    ntt / intt                       math/ntt.rs:67-82, :109-125
    Polynomial.fast_coset_evaluate   math/polynomial.rs:1374-1399
    Tip5.hash_10 / hash_pair / hash_varlen / permutation   tip5/mod.rs:529-623
    MerkleTree.par_new / sequential_new / par_frugal_root / sequential_frugal_root
                                     util_types/merkle_tree.rs:149-364
    MerkleTreeError                  util_types/merkle_tree.rs:933-965

Data are numpy uint64 arrays of RAW Montgomery words exactly as the Rust types lay them out
(BFieldElement 1 word, XFieldElement 3, Digest 5); `BFieldElement.new/value` convert single values.
Every operation runs on the GPU through the C ABI; there is no CPU fallback -- a missing library
or device raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

__all__ = [
    "P", "BFieldElement", "ntt", "intt", "Polynomial", "ZerofierTree", "barycentric_evaluate", "fast_coset_evaluate", "fast_coset_interpolate", "fast_multiply", "fast_square", "Tip5", "Tip5Sponge", "Digest", "MerkleTree",
    "MerkleTreeError", "TwentyFirstError", "NttPanic", "lib", "device", "set_device", "get_device", "shard_range",
]

P = 0xFFFFFFFF00000001  # BFieldElement::P, math/b_field_element.rs:225
_R = 1 << 64
_R_INV = pow(_R, P - 2, P)


def lib():
    return _lib.lib()


# ----------------------------------------------------------------------------- errors
class TwentyFirstError(RuntimeError):
    def __init__(self, code: int, where: str = ""):
        name = lib().tf_status_string(code).decode()
        detail = lib().tf_last_error().decode()
        msg = f"{where}: {name}" if where else name
        if detail and (8 <= code <= 10 or code == 18):  # tf_last_error() is the text of a HIP failure / of an exception caught at the ABI; the other codes carry none
            msg += f" ({detail})"
        super().__init__(msg)
        self.code = code


class NttPanic(TwentyFirstError):
    """Raised where the reference panics (math/ntt.rs:135-140, math/polynomial.rs:1388-1392)."""


class MerkleTreeError(TwentyFirstError):
    """util_types/merkle_tree.rs:933-965; .variant is the Rust variant name."""

    VARIANTS = {1: "TooFewLeafs", 2: "IncorrectNumberOfLeafs", 3: "TreeTooHigh", 11: "LeafIndexInvalid"}

    def __init__(self, code: int, where: str = ""):
        super().__init__(code, where)
        self.variant = self.VARIANTS.get(code, "Unknown")


def _check(rc: int, where: str):
    if rc == 0:
        return
    if rc in (1, 2, 3, 11):
        raise MerkleTreeError(rc, where)
    if rc in (4, 5, 6, 12, 14, 15, 16):
        raise NttPanic(rc, where)
    raise TwentyFirstError(rc, where)


# ----------------------------------------------------------------------------- scalars
class BFieldElement:
    """Conversions between canonical values and raw Montgomery words (host-side convenience;
    math/b_field_element.rs:235-237, :248-250)."""

    P = P
    MAX = P - 1
    ZERO_RAW = 0             # :691-693
    ONE_RAW = 0xFFFFFFFF     # :707-709 (2^64 mod p)

    @staticmethod
    def new(value: int) -> int:
        return (value % (1 << 64)) % P * _R % P

    @staticmethod
    def value(raw: int) -> int:
        return int(raw) * _R_INV % P

    @staticmethod
    def generator() -> int:
        return BFieldElement.new(7)  # :312-314

    @staticmethod
    def primitive_root_of_unity(n: int):
        """:814-818; raw word, or None if n is not 0/1/a power of two <= 2^32."""
        if n in (0, 1):
            return BFieldElement.new(1)
        if n & (n - 1) or n > (1 << 32):
            return None
        return BFieldElement.new(pow(7, (P - 1) // n, P))


def _words(x, name="x") -> np.ndarray:
    if not isinstance(x, np.ndarray) or x.dtype != np.uint64 or not x.flags["C_CONTIGUOUS"]:
        raise TypeError(f"{name} must be a C-contiguous numpy uint64 array of raw Montgomery words")
    return x


def _ptr(a: np.ndarray):
    return C.c_void_p(a.ctypes.data) if a.size else C.c_void_p(0)


# ----------------------------------------------------------------------------- devices
def set_device(device: int) -> None:
    """Select the GPU every later call of this host thread runs on (include/tf_hip.h: tf_set_device)."""
    _check(lib().tf_set_device(int(device)), "set_device")


def get_device() -> int:
    d = C.c_int(0)
    _check(lib().tf_get_device(C.byref(d)), "get_device")
    return d.value


def shard_range(total_units: int, n_shards: int, shard: int):
    """[lo, hi) of slice `shard` when `total_units` independent units are split over `n_shards` devices: the rule of the
    multi-device entry points (tf_shard_range), the same as sharding.shard_range."""
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    _check(lib().tf_shard_range(total_units, n_shards, shard, C.byref(lo), C.byref(hi)), "shard_range")
    return lo.value, hi.value


def _devices(devices):
    """devices="all" -> (NULL, 0): every visible GPU; a sequence of indices -> (int array, count).  The same index may repeat."""
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError('devices must be "all" or a sequence of device indices')
        return C.cast(C.c_void_p(0), C.POINTER(C.c_int)), 0
    ids = [int(d) for d in devices]
    if not ids:
        raise ValueError("devices must not be empty")
    return (C.c_int * len(ids))(*ids), len(ids)


# ----------------------------------------------------------------------------- NTT
def ntt(x: np.ndarray, width: int = 1, batch: int = 1, _inverse: bool = False, devices=None) -> None:
    """In-place NTT of `batch` contiguous slices (math/ntt.rs:67-82).  width 1 = BFieldElement slice,
    3 = XFieldElement slice.  Raises NttPanic where the reference panics.
    devices: None = the current GPU; "all" or a list of device indices = the batch split over those GPUs (tf_ntt_*_multi)."""
    x = _words(x)
    if width not in (1, 3):
        raise ValueError("width must be 1 (BFieldElement) or 3 (XFieldElement)")
    if batch < 0 or (batch and x.size % (batch * width)):
        raise ValueError("array size is not batch * n * width")
    n = x.size // (batch * width) if batch else 0
    if devices is not None:
        ids, k = _devices(devices)
        fn = lib().tf_ntt_bfe_multi if width == 1 else lib().tf_ntt_xfe_multi
        _check(fn(_ptr(x), n, batch, int(_inverse), ids, k), "intt" if _inverse else "ntt")
        return
    fn = lib().tf_ntt_bfe if width == 1 else lib().tf_ntt_xfe
    _check(fn(_ptr(x), n, batch, int(_inverse)), "intt" if _inverse else "ntt")


def intt(x: np.ndarray, width: int = 1, batch: int = 1, devices=None) -> None:
    """In-place inverse NTT (math/ntt.rs:109-125)."""
    ntt(x, width=width, batch=batch, _inverse=True, devices=devices)


def _xfe_offset(offset, width: int):
    """None for a BFieldElement offset (one raw word), the 3 raw words for an XFieldElement offset (XFE polynomials only:
    FF: Mul<S, Output = FF>, math/polynomial.rs:1376)."""
    if isinstance(offset, (int, np.integer)):
        return None
    off = np.ascontiguousarray(offset, dtype=np.uint64).reshape(-1)
    if off.size == 1:
        return None
    if off.size != 3 or width != 3:
        raise TypeError("an XFieldElement offset (3 raw words) needs XFieldElement coefficients")
    return off


def fast_coset_evaluate(coeffs: np.ndarray, offset_raw, order: int, width: int = 1, batch: int = 1, devices=None) -> np.ndarray:
    """`batch` polynomials of equal length -> `batch` x `order` evaluations (math/polynomial.rs:1374-1399).
    The reference compares `order` with the DEGREE (:1388): high-order zero coefficients common to the whole batch are
    trimmed here before the length reaches the C ABI, as Polynomial::degree() does for one polynomial."""
    coeffs = _words(coeffs, "coeffs")
    if batch and coeffs.size % (batch * width):
        raise ValueError("coeffs size is not batch * n_coeffs * width")
    n_coeffs = coeffs.size // (batch * width) if batch else 0
    if batch and n_coeffs > order:
        c3 = coeffs.reshape(batch, n_coeffs, width)
        live = np.nonzero(c3.any(axis=(0, 2)))[0]
        keep = int(live[-1]) + 1 if live.size else 0
        if keep < n_coeffs:
            coeffs = np.ascontiguousarray(c3[:, :keep, :]).reshape(-1)
            n_coeffs = keep
    out = np.empty(batch * order * width, dtype=np.uint64)
    xoff = _xfe_offset(offset_raw, width)
    if xoff is not None:  # S = XFieldElement (:1374-1378)
        if devices is not None:  # (this used to run on the current GPU only, without saying so)
            raise NotImplementedError("fast_coset_evaluate with an XFieldElement offset is a single-device call: there is no tf_coset_eval_xfe_xoffset_multi")
        _check(lib().tf_coset_eval_xfe_xoffset(_ptr(coeffs), n_coeffs, _ptr(xoff), _ptr(out), order, batch), "fast_coset_evaluate")
        return out
    offset_raw = int(np.asarray(offset_raw).reshape(-1)[0])
    if devices is not None:  # the polynomials of the batch split over several GPUs (tf_coset_eval_*_multi)
        ids, k = _devices(devices)
        fn = lib().tf_coset_eval_bfe_multi if width == 1 else lib().tf_coset_eval_xfe_multi
        _check(fn(_ptr(coeffs), n_coeffs, C.c_uint64(offset_raw), _ptr(out), order, batch, ids, k), "fast_coset_evaluate")
        return out
    fn = lib().tf_coset_eval_bfe if width == 1 else lib().tf_coset_eval_xfe
    _check(fn(_ptr(coeffs), n_coeffs, C.c_uint64(offset_raw), _ptr(out), order, batch), "fast_coset_evaluate")
    return out


def fast_coset_interpolate(values: np.ndarray, offset_raw, width: int = 1, batch: int = 1) -> np.ndarray:
    """`batch` x n evaluations on {offset * w^i} -> `batch` x n coefficients (math/polynomial.rs:1907-1918)."""
    values = _words(values, "values")
    if batch and values.size % (batch * width):
        raise ValueError("values size is not batch * n * width")
    n = values.size // (batch * width) if batch else 0
    out = np.empty_like(values)
    xoff = _xfe_offset(offset_raw, width)
    if xoff is not None:  # S = XFieldElement (:1907-1911)
        _check(lib().tf_coset_interpolate_xfe_xoffset(_ptr(values), n, _ptr(xoff), _ptr(out), batch), "fast_coset_interpolate")
        return out
    offset_raw = int(np.asarray(offset_raw).reshape(-1)[0])
    fn = lib().tf_coset_interpolate_bfe if width == 1 else lib().tf_coset_interpolate_xfe
    _check(fn(_ptr(values), n, C.c_uint64(offset_raw), _ptr(out), batch), "fast_coset_interpolate")
    return out


def fast_multiply(a: np.ndarray, b: np.ndarray, width: int = 1, batch: int = 1) -> np.ndarray:
    """Coefficient arrays of `batch` polynomial pairs -> `batch` x (na + nb - 1) product coefficients
    (math/polynomial.rs:900-932, untrimmed)."""
    a, b = _words(a, "a"), _words(b, "b")
    if batch and (a.size % (batch * width) or b.size % (batch * width)):
        raise ValueError("operand size is not batch * n_coeffs * width")
    na = a.size // (batch * width) if batch else 0
    nb = b.size // (batch * width) if batch else 0
    if na == 0 or nb == 0:
        return np.zeros(0, dtype=np.uint64)
    out = np.empty(batch * (na + nb - 1) * width, dtype=np.uint64)
    fn = lib().tf_poly_mul_bfe if width == 1 else lib().tf_poly_mul_xfe
    _check(fn(_ptr(a), na, _ptr(b), nb, _ptr(out), batch), "fast_multiply")
    return out


def fast_square(a: np.ndarray, width: int = 1, batch: int = 1) -> np.ndarray:
    """`batch` polynomials of na coefficients -> `batch` x (2 na - 1) coefficients of their squares
    (math/polynomial.rs:780-798, untrimmed)."""
    a = _words(a, "a")
    if batch and a.size % (batch * width):
        raise ValueError("operand size is not batch * n_coeffs * width")
    na = a.size // (batch * width) if batch else 0
    if na == 0:
        return np.zeros(0, dtype=np.uint64)
    out = np.empty(batch * (2 * na - 1) * width, dtype=np.uint64)
    fn = lib().tf_poly_square_bfe if width == 1 else lib().tf_poly_square_xfe
    _check(fn(_ptr(a), na, _ptr(out), batch), "fast_square")
    return out


class Polynomial:
    """Coefficients low -> high degree (math/polynomial.rs:78-84); only the hot-path members."""

    def __init__(self, coefficients: np.ndarray, width: int = 1):
        c = _words(np.ascontiguousarray(coefficients, dtype=np.uint64), "coefficients").reshape(-1)
        self.width = width
        n = c.size // width
        c = c.reshape(n, width)
        while n and not c[n - 1].any():  # Polynomial::new normalises: leading zeros do not count (:degree)
            n -= 1
        self.coefficients = np.ascontiguousarray(c[:n]).reshape(-1)

    def degree(self) -> int:
        return self.coefficients.size // self.width - 1

    def fast_coset_evaluate(self, offset_raw: int, order: int) -> np.ndarray:
        if order <= self.degree():  # :1388-1392
            raise NttPanic(6, "fast_coset_evaluate")
        return fast_coset_evaluate(self.coefficients, offset_raw, order, width=self.width, batch=1)

    @classmethod
    def fast_coset_interpolate(cls, offset_raw: int, values: np.ndarray, width: int = 1) -> "Polynomial":
        """math/polynomial.rs:1907-1918; panics (NttPanic) unless len(values) is a power of two."""
        v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1)
        return cls(fast_coset_interpolate(v, offset_raw, width=width), width=width)

    def batch_evaluate(self, domain: np.ndarray) -> np.ndarray:
        """math/polynomial.rs:1840-1852: f at every point of `domain` (points of the same field as the coefficients)."""
        pts = _words(np.ascontiguousarray(domain, dtype=np.uint64).reshape(-1), "domain")
        n_points = pts.size // self.width
        out = np.empty(n_points * self.width, dtype=np.uint64)
        fn = lib().tf_poly_batch_evaluate_bfe if self.width == 1 else lib().tf_poly_batch_evaluate_xfe
        _check(fn(_ptr(self.coefficients), self.coefficients.size // self.width, _ptr(pts), n_points, _ptr(out)), "batch_evaluate")
        return out

    def evaluate_at_xfe_points(self, points: np.ndarray) -> np.ndarray:
        """Polynomial<BFieldElement>::evaluate::<XFieldElement, XFieldElement> (math/polynomial.rs:309-320) at every point of
        `points` (3 raw words each): n_points x 3 raw words."""
        if self.width != 1:
            return self.batch_evaluate(points)
        pts = _words(np.ascontiguousarray(points, dtype=np.uint64).reshape(-1), "points")
        if pts.size % 3:
            raise ValueError("points must hold whole XFieldElements")
        out = np.zeros(pts.size, dtype=np.uint64)
        _check(lib().tf_poly_evaluate_bfe_at_xfe(_ptr(self.coefficients), self.coefficients.size, 1, _ptr(pts), pts.size // 3, _ptr(out)), "evaluate")
        return out

    def clean_divide(self, divisor: "Polynomial") -> "Polynomial":
        """math/polynomial.rs:2358-2411 (BFieldElement only): self / divisor for a division known to be clean.  Panics (NttPanic)
        on a zero divisor and on an unclean division."""
        if self.width != 1 or divisor.width != 1:
            raise TypeError("clean_divide is defined for Polynomial<BFieldElement> (polynomial.rs:2333)")
        na, nb = self.coefficients.size, divisor.coefficients.size
        out = np.zeros(max(na - nb + 1, 0), dtype=np.uint64)
        _check(lib().tf_poly_clean_divide_bfe(_ptr(self.coefficients), na, _ptr(divisor.coefficients), nb, _ptr(out)), "clean_divide")
        return Polynomial(out)

    @classmethod
    def zerofier(cls, roots: np.ndarray, width: int = 1) -> "Polynomial":
        """math/polynomial.rs:1435-1441 (and par_zerofier :1444-1459): the monic polynomial with exactly these roots."""
        r = _words(np.ascontiguousarray(roots, dtype=np.uint64).reshape(-1), "roots")
        if r.size % width:
            raise ValueError("roots size is not n * width")
        n = r.size // width
        out = np.empty((n + 1) * width, dtype=np.uint64)
        fn = lib().tf_poly_zerofier_bfe if width == 1 else lib().tf_poly_zerofier_xfe
        _check(fn(_ptr(r), n, _ptr(out)), "zerofier")
        return cls(out, width=width)

    @classmethod
    def interpolate(cls, domain: np.ndarray, values: np.ndarray, width: int = 1) -> "Polynomial":
        """math/polynomial.rs:1502-1520 (and par_interpolate :1525-1545, fast_interpolate :1611-1654): the lowest-degree
        polynomial through (domain[i], values[i]).  Panics (NttPanic) on an empty domain, on unequal lengths and on repeated
        domain points."""
        return cls.batch_fast_interpolate(domain, [values], width=width)[0]

    @classmethod
    def batch_fast_interpolate(cls, domain: np.ndarray, values_matrix, width: int = 1) -> list:
        """math/polynomial.rs:1703-1731: one interpolant per row of `values_matrix` over the same domain."""
        d = _words(np.ascontiguousarray(domain, dtype=np.uint64).reshape(-1), "domain")
        if d.size % width:
            raise ValueError("domain size is not n * width")
        n = d.size // width
        if n == 0:
            raise NttPanic(14, "interpolate")  # "interpolation must happen through more than zero points" (:1503-1506)
        rows = [np.ascontiguousarray(v, dtype=np.uint64).reshape(-1) for v in values_matrix]
        for v in rows:
            if v.size != n * width:
                raise NttPanic(14, "interpolate: the domain and values lists have to be of equal length")  # :1507-1511
        if not rows:
            return []
        vals = np.ascontiguousarray(np.concatenate(rows))
        out = np.empty(len(rows) * n * width, dtype=np.uint64)
        fn = lib().tf_poly_interpolate_bfe if width == 1 else lib().tf_poly_interpolate_xfe
        _check(fn(_ptr(d), _ptr(vals), n, len(rows), _ptr(out)), "interpolate")
        return [cls(out[i * n * width:(i + 1) * n * width], width=width) for i in range(len(rows))]

    @staticmethod
    def batch_coset_extrapolate(domain_offset_raw: int, codeword_length: int, codewords: np.ndarray, points: np.ndarray,
                                width: int = 1) -> np.ndarray:
        """math/polynomial.rs:2196-2208 (and par_ :2262): every codeword's interpolant at every point, codeword-major.
        Panics (NttPanic) unless codeword_length is a power of two."""
        cw = _words(np.ascontiguousarray(codewords, dtype=np.uint64).reshape(-1), "codewords")
        pts = _words(np.ascontiguousarray(points, dtype=np.uint64).reshape(-1), "points")
        n = int(codeword_length)
        batch = cw.size // (n * width) if n else 0
        n_points = pts.size // width
        out = np.empty(batch * n_points * width, dtype=np.uint64)
        fn = lib().tf_coset_extrapolate_bfe if width == 1 else lib().tf_coset_extrapolate_xfe
        _check(fn(C.c_uint64(domain_offset_raw), _ptr(cw), n, batch, _ptr(pts), n_points, _ptr(out)), "batch_coset_extrapolate")
        return out

    @staticmethod
    def coset_extrapolate(domain_offset_raw: int, codeword: np.ndarray, points: np.ndarray, width: int = 1) -> np.ndarray:
        """math/polynomial.rs:2117-2128"""
        cw = np.ascontiguousarray(codeword, dtype=np.uint64).reshape(-1)
        return Polynomial.batch_coset_extrapolate(domain_offset_raw, cw.size // width, cw, points, width=width)

    def fast_square(self) -> "Polynomial":
        """math/polynomial.rs:780-798"""
        if self.degree() < 0:
            return Polynomial(np.zeros(0, dtype=np.uint64), width=self.width)
        return Polynomial(fast_square(self.coefficients, width=self.width), width=self.width)

    def fast_multiply(self, other: "Polynomial") -> "Polynomial":
        """math/polynomial.rs:900-932 (same field on both sides)."""
        if self.width != other.width:
            raise ValueError("mixed-field products stay on the caller's side")
        if self.degree() < 0 or other.degree() < 0:
            return Polynomial(np.zeros(0, dtype=np.uint64), width=self.width)
        return Polynomial(fast_multiply(self.coefficients, other.coefficients, width=self.width), width=self.width)


def barycentric_evaluate(codewords: np.ndarray, indeterminate, width: int = 1, batch: int = 1) -> np.ndarray:
    """math/polynomial.rs:2609-2637 for `batch` codewords of equal length at one indeterminate (an int / one raw word: BFieldElement;
    three raw words: XFieldElement).  Returns batch x 3 raw words (XFieldElements), or batch words when codewords and indeterminate
    are both in the base field (the reference's BFieldElement result).  Panics (NttPanic) unless the length is a power of two, and
    when the indeterminate lies in the evaluation domain."""
    cw = _words(np.ascontiguousarray(codewords, dtype=np.uint64).reshape(-1), "codewords")
    if width not in (1, 3) or batch < 0 or (batch and cw.size % (batch * width)):
        raise ValueError("codewords size is not batch * n * width")
    n = cw.size // (batch * width) if batch else 0
    xi = np.asarray([indeterminate] if isinstance(indeterminate, (int, np.integer)) else indeterminate, dtype=np.uint64).reshape(-1)
    if xi.size not in (1, 3):
        raise ValueError("the indeterminate is one raw word (BFieldElement) or three (XFieldElement)")
    x = np.zeros(3, dtype=np.uint64)
    x[: xi.size] = xi
    out = np.zeros(3 * batch, dtype=np.uint64)
    fn = lib().tf_barycentric_evaluate_bfe if width == 1 else lib().tf_barycentric_evaluate_xfe
    _check(fn(_ptr(cw), n, batch, _ptr(x), _ptr(out)), "barycentric_evaluate")
    if width == 1 and xi.size == 1:
        return np.ascontiguousarray(out.reshape(batch, 3)[:, 0])
    return out


class ZerofierTree:
    """math/zerofier_tree.rs: the tree of vanishing polynomials of a domain, built once on the device and kept there
    (`ZerofierTree::new_from_domain` :66-87), for `Polynomial::divide_and_conquer_batch_evaluate` (polynomial.rs:1882-1894) and for
    interpolation over the same domain.  Holds device memory: use as a context manager or call close()."""

    def __init__(self, domain: np.ndarray, width: int = 1):
        d = _words(np.ascontiguousarray(domain, dtype=np.uint64).reshape(-1), "domain")
        if width not in (1, 3) or d.size % width:
            raise ValueError("domain size is not n * width")
        self.width = width
        self.num_points = d.size // width
        self._h = C.c_void_p(0)
        self._free = lib().tf_zerofier_tree_free
        fn = lib().tf_zerofier_tree_new_bfe if width == 1 else lib().tf_zerofier_tree_new_xfe
        _check(fn(_ptr(d), self.num_points, C.byref(self._h)), "ZerofierTree::new_from_domain")

    @classmethod
    def new_from_domain(cls, domain: np.ndarray, width: int = 1) -> "ZerofierTree":
        return cls(domain, width)

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

    @property
    def handle(self):
        if not self._h.value:
            raise ValueError("the tree has been closed")
        return self._h

    def zerofier(self) -> "Polynomial":
        """:93-99"""
        out = np.empty((self.num_points + 1) * self.width, dtype=np.uint64)
        _check(lib().tf_zerofier_tree_zerofier(self.handle, _ptr(out)), "ZerofierTree::zerofier")
        return Polynomial(out, width=self.width)

    def batch_evaluate(self, polynomial: "Polynomial") -> np.ndarray:
        """polynomial.divide_and_conquer_batch_evaluate(&tree) (polynomial.rs:1882-1894): the values on the tree's domain."""
        if polynomial.width != self.width:
            raise TypeError("the polynomial and the tree are over different fields")
        out = np.zeros(self.num_points * self.width, dtype=np.uint64)
        c = polynomial.coefficients
        _check(lib().tf_zerofier_tree_batch_evaluate(self.handle, _ptr(c), c.size // self.width, 1, _ptr(out)), "divide_and_conquer_batch_evaluate")
        return out

    def interpolate(self, values_matrix) -> list:
        """One interpolant per value row over the tree's domain (the memoised form of batch_fast_interpolate, polynomial.rs:1703-1838)."""
        n, w = self.num_points, self.width
        if n == 0:
            raise NttPanic(14, "interpolate")
        rows = [np.ascontiguousarray(v, dtype=np.uint64).reshape(-1) for v in values_matrix]
        for v in rows:
            if v.size != n * w:
                raise NttPanic(14, "interpolate: the domain and values lists have to be of equal length")
        if not rows:
            return []
        vals = np.ascontiguousarray(np.concatenate(rows))
        out = np.empty(len(rows) * n * w, dtype=np.uint64)
        _check(lib().tf_zerofier_tree_interpolate(self.handle, _ptr(vals), len(rows), _ptr(out)), "interpolate")
        return [Polynomial(out[i * n * w:(i + 1) * n * w], width=w) for i in range(len(rows))]


# ----------------------------------------------------------------------------- Tip5
class Digest:
    LEN = 5  # tip5/digest.rs:49
    BYTES = 40

    @staticmethod
    def to_hex(d) -> str:
        """Canonical values as little-endian bytes (tip5/digest.rs:85-90, :144-153)."""
        return b"".join(BFieldElement.value(int(v)).to_bytes(8, "little") for v in d).hex()


class Tip5:
    RATE = 10
    STATE_SIZE = 16

    @staticmethod
    def permute_states(states: np.ndarray) -> None:
        """In-place Tip5::permutation on count x 16 raw words (tip5/mod.rs:529-533)."""
        states = _words(states, "states")
        if states.size % 16:
            raise ValueError("states must hold a multiple of 16 words")
        _check(lib().tf_tip5_permute(_ptr(states), states.size // 16), "Tip5::permutation")

    @staticmethod
    def trace_states(states: np.ndarray) -> np.ndarray:
        """Tip5::trace (tip5/mod.rs:538-548) for count x 16 raw words: returns count x 6 x 16 words (the state before the permutation
        and after each round); `states` ends permuted, like `&mut self`."""
        states = _words(states, "states")
        if states.size % 16:
            raise ValueError("states must hold a multiple of 16 words")
        count = states.size // 16
        trace = np.empty(count * 96, dtype=np.uint64)
        _check(lib().tf_tip5_trace(_ptr(states), _ptr(trace), count), "Tip5::trace")
        return trace.reshape(count, 6, 16)

    @staticmethod
    def permutation(state: np.ndarray) -> np.ndarray:
        s = np.ascontiguousarray(state, dtype=np.uint64).copy()
        Tip5.permute_states(s)
        return s

    @staticmethod
    def hash_pairs(inp: np.ndarray) -> np.ndarray:
        """count x 10 words -> count x 5 words; row i is hash_10 of its 10 words (tip5/mod.rs:559-586)."""
        inp = _words(inp, "in")
        if inp.size % 10:
            raise ValueError("input must hold a multiple of 10 words")
        out = np.empty(inp.size // 2, dtype=np.uint64)
        _check(lib().tf_tip5_hash_pairs(_ptr(inp), _ptr(out), inp.size // 10), "Tip5::hash_pair")
        return out

    @staticmethod
    def hash_10(inp) -> np.ndarray:
        a = np.ascontiguousarray(inp, dtype=np.uint64)
        if a.size != 10:
            raise ValueError("hash_10 takes exactly 10 elements")
        return Tip5.hash_pairs(a)

    @staticmethod
    def hash_pair(left, right) -> np.ndarray:
        a = np.concatenate([np.asarray(left, dtype=np.uint64), np.asarray(right, dtype=np.uint64)])
        if a.size != 10:
            raise ValueError("hash_pair takes two digests")
        return Tip5.hash_pairs(a)

    @staticmethod
    def hash_varlen_rows(rows: np.ndarray, row_len: int) -> np.ndarray:
        rows = _words(rows, "rows")
        if row_len < 0 or (row_len and rows.size % row_len):
            raise ValueError("rows size is not n_rows * row_len")
        n_rows = rows.size // row_len if row_len else 0
        out = np.empty(n_rows * 5, dtype=np.uint64)
        _check(lib().tf_tip5_hash_varlen_rows(_ptr(rows), row_len, n_rows, _ptr(out)), "Tip5::hash_varlen")
        return out

    @staticmethod
    def hash_table_rows(columns: np.ndarray, n_rows: int, width: int = 1) -> np.ndarray:
        """hash_varlen of every row of a column-major table (columns back to back, n_rows elements of `width` words each):
        n_rows x 5 words."""
        t = _words(np.ascontiguousarray(columns, dtype=np.uint64).reshape(-1), "columns")
        col_words = n_rows * width
        n_cols = t.size // col_words if col_words else 0
        out = np.empty(n_rows * 5, dtype=np.uint64)
        _check(lib().tf_tip5_hash_table_rows(_ptr(t), n_rows, n_cols, width, col_words, _ptr(out), 1), "Tip5::hash_varlen")
        return out

    @staticmethod
    def hash_varlen(inp) -> np.ndarray:
        """tip5/mod.rs:617-623.  (One row; the empty input is one row of length 0.)"""
        a = np.ascontiguousarray(inp, dtype=np.uint64).reshape(-1)
        out = np.empty(5, dtype=np.uint64)
        _check(lib().tf_tip5_hash_varlen_rows(_ptr(a), a.size, 1, _ptr(out)), "Tip5::hash_varlen")
        return out


class Tip5Sponge:
    """`batch` independent Tip5 sponges stepped together (impl Sponge for Tip5, tip5/mod.rs:677-699; trait
    util_types/sponge.rs:33-55).  The state lives in a (batch, 16) array of raw words; every absorb / squeeze is one
    batched Tip5::permutation on the device."""

    RATE = Tip5.RATE

    def __init__(self, batch: int = 1, fixed_length: bool = False):
        self.state = np.zeros((batch, Tip5.STATE_SIZE), dtype=np.uint64)  # Domain::VariableLength (tip5/mod.rs:511-526)
        if fixed_length:
            self.state[:, Tip5.RATE:] = BFieldElement.ONE_RAW

    @classmethod
    def init(cls, batch: int = 1) -> "Tip5Sponge":
        return cls(batch)

    def _permute(self) -> None:
        flat = self.state.reshape(-1)
        Tip5.permute_states(flat)

    def absorb(self, inp) -> None:
        """Overwrite-mode absorb of RATE elements per sponge (tip5/mod.rs:684-691)."""
        a = np.ascontiguousarray(inp, dtype=np.uint64).reshape(self.state.shape[0], Tip5.RATE)
        self.state[:, :Tip5.RATE] = a
        self._permute()

    def squeeze(self) -> np.ndarray:
        """tip5/mod.rs:693-698: the rate part, then one permutation."""
        out = self.state[:, :Tip5.RATE].copy()
        self._permute()
        return out

    def pad_and_absorb_all(self, inp) -> None:
        """util_types/sponge.rs:41-55: full chunks, then the remainder padded with 1, 0, 0, ...  (equal lengths per sponge)."""
        b = self.state.shape[0]
        a = np.ascontiguousarray(inp, dtype=np.uint64).reshape(b, -1)
        full = a.shape[1] // Tip5.RATE
        for c in range(full):
            self.absorb(a[:, c * Tip5.RATE:(c + 1) * Tip5.RATE])
        rem = a.shape[1] - full * Tip5.RATE
        last = np.zeros((b, Tip5.RATE), dtype=np.uint64)
        last[:, :rem] = a[:, full * Tip5.RATE:]
        last[:, rem] = BFieldElement.ONE_RAW
        self.absorb(last)


# ----------------------------------------------------------------------------- Merkle tree
class MerkleTree:
    """nodes: (2n, 5) raw words in the reference's heap layout (util_types/merkle_tree.rs:85-88):
    nodes[0] dummy, nodes[1] root, leaves at nodes[n:]."""

    ROOT_INDEX = 1

    def __init__(self, nodes: np.ndarray):
        self.nodes = nodes

    @classmethod
    def par_new(cls, leafs: np.ndarray) -> "MerkleTree":  # :165-212
        leafs = _words(np.ascontiguousarray(leafs, dtype=np.uint64).reshape(-1), "leafs")
        if leafs.size % 5:
            raise ValueError("leafs must hold whole digests (5 words each)")
        n = leafs.size // 5
        nodes = np.empty(max(10 * n, 1), dtype=np.uint64)
        _check(lib().tf_merkle_build(_ptr(leafs), n, _ptr(nodes), 1), "MerkleTree::par_new")
        return cls(nodes[: 10 * n].reshape(2 * n, 5))

    sequential_new = par_new  # :149-153 -- same result by construction (tests :1059-1087)

    @staticmethod
    def build_batch(leafs: np.ndarray, n_leafs: int, devices=None) -> np.ndarray:
        """`batch` independent trees of n_leafs leaves each -> (batch, 2n, 5); devices: as ntt()."""
        leafs = _words(leafs, "leafs")
        if n_leafs <= 0 or leafs.size % (5 * n_leafs):
            if n_leafs == 0:
                raise MerkleTreeError(1, "MerkleTree::par_new")
            raise ValueError("leafs size is not batch * n_leafs * 5")
        batch = leafs.size // (5 * n_leafs)
        nodes = np.empty(batch * 10 * n_leafs, dtype=np.uint64)
        if devices is not None:
            ids, k = _devices(devices)
            _check(lib().tf_merkle_build_multi(_ptr(leafs), n_leafs, _ptr(nodes), batch, ids, k), "MerkleTree::par_new")
        else:
            _check(lib().tf_merkle_build(_ptr(leafs), n_leafs, _ptr(nodes), batch), "MerkleTree::par_new")
        return nodes.reshape(batch, 2 * n_leafs, 5)

    @staticmethod
    def sequential_frugal_root(leafs: np.ndarray) -> np.ndarray:  # :299-309
        leafs = _words(np.ascontiguousarray(leafs, dtype=np.uint64).reshape(-1), "leafs")
        n = leafs.size // 5
        root = np.empty(5, dtype=np.uint64)
        _check(lib().tf_merkle_root(_ptr(leafs), n, _ptr(root), 1), "MerkleTree::sequential_frugal_root")
        return root

    @staticmethod
    def par_frugal_root(leafs: np.ndarray) -> np.ndarray:  # :332-364
        a = np.asarray(leafs)
        if a.size == 0:  # is_power_of_two() is false for 0  (:333-335)
            raise MerkleTreeError(2, "MerkleTree::par_frugal_root")
        return MerkleTree.sequential_frugal_root(leafs)

    @staticmethod
    def roots_batch(leafs: np.ndarray, n_leafs: int, devices=None) -> np.ndarray:
        leafs = _words(leafs, "leafs")
        batch = leafs.size // (5 * n_leafs) if n_leafs else 0
        roots = np.empty(max(batch, 1) * 5, dtype=np.uint64)
        if devices is not None:
            ids, k = _devices(devices)
            _check(lib().tf_merkle_root_multi(_ptr(leafs), n_leafs, _ptr(roots), batch, ids, k), "MerkleTree::par_frugal_root")
        else:
            _check(lib().tf_merkle_root(_ptr(leafs), n_leafs, _ptr(roots), batch), "MerkleTree::par_frugal_root")
        return roots[: batch * 5].reshape(batch, 5)

    @classmethod
    def from_rows(cls, rows: np.ndarray, row_len: int) -> "MerkleTree":
        """Leaves = Tip5::hash_varlen of every row (tip5/mod.rs:617-623), then par_new -- one device pipeline."""
        rows = _words(np.ascontiguousarray(rows, dtype=np.uint64).reshape(-1), "rows")
        n = rows.size // row_len if row_len else 0
        nodes = np.empty(max(10 * n, 1), dtype=np.uint64)
        _check(lib().tf_merkle_from_rows(_ptr(rows), row_len, n, _ptr(nodes), 1), "MerkleTree::par_new")
        return cls(nodes[: 10 * n].reshape(2 * n, 5))

    @classmethod
    def from_columns(cls, columns: np.ndarray, n_rows: int, width: int = 1) -> "MerkleTree":
        """Leaves = Tip5::hash_varlen of every ROW of a column-major table (one codeword per column, n_rows elements of
        `width` words each, columns back to back) -- the layout a batch of coset evaluations produces."""
        t = _words(np.ascontiguousarray(columns, dtype=np.uint64).reshape(-1), "columns")
        col_words = n_rows * width
        n_cols = t.size // col_words if col_words else 0
        nodes = np.empty(max(10 * n_rows, 1), dtype=np.uint64)
        _check(lib().tf_merkle_from_columns(_ptr(t), n_rows, n_cols, width, col_words, _ptr(nodes), 1), "MerkleTree::par_new")
        return cls(nodes[: 10 * n_rows].reshape(2 * n_rows, 5))

    @staticmethod
    def authentication_structure_node_indices(num_leafs: int, leaf_indices) -> np.ndarray:
        """util_types/merkle_tree.rs:449-504 (descending node indices); raises MerkleTreeError variants."""
        li = np.ascontiguousarray(leaf_indices, dtype=np.uint64).reshape(-1)
        cap = max(1, li.size * 66)
        out = np.empty(cap, dtype=np.uint64)
        cnt = C.c_size_t(0)
        rc = lib().tf_merkle_auth_structure_indices(num_leafs, _ptr(li), li.size, _ptr(out), cap, C.byref(cnt))
        if rc == 11:
            raise MerkleTreeError(rc, "MerkleTree::authentication_structure")
        _check(rc, "MerkleTree::authentication_structure")
        return out[: cnt.value].copy()

    def authentication_structure(self, leaf_indices) -> np.ndarray:
        """util_types/merkle_tree.rs:614-622: the digests at authentication_structure_node_indices, in that order."""
        idx = self.authentication_structure_node_indices(self.num_leafs(), leaf_indices)
        return self.nodes[idx.astype(np.int64)]

    def root(self) -> np.ndarray:  # :624-626
        return self.nodes[1]

    def num_leafs(self) -> int:  # :628-631
        return self.nodes.shape[0] // 2

    def height(self) -> int:  # :633-636
        return self.num_leafs().bit_length() - 1

    def node(self, i: int):  # :638-645
        return self.nodes[i] if 0 <= i < self.nodes.shape[0] else None

    def leafs(self) -> np.ndarray:  # :647-652
        return self.nodes[self.num_leafs():]

    def leaf(self, i: int):  # :654-661
        n = self.num_leafs()
        return self.nodes[n + i] if 0 <= i < n else None


from . import device  # noqa: E402  (torch device-pointer API)
