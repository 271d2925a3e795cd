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


def ntt_(x, n: int, batch: int = 1, width: int = 1, inverse: bool = False, stream=None) -> None:
    """In place on device: `batch` slices of n elements (math/ntt.rs:67-82, :109-125)."""
    x = _t(x, "x")
    if x.numel() != n * batch * width:
        raise ValueError("tensor size is not batch * n * width")
    fn = _lib.lib().tf_ntt_bfe_dev if width == 1 else _lib.lib().tf_ntt_xfe_dev
    _chk(fn(_p(x), n, batch, int(inverse), _stream(stream)), "intt" if inverse else "ntt")


def coset_evaluate(coeffs, n_coeffs: int, offset_raw: int, out, order: int, batch: int = 1, width: int = 1, stream=None) -> None:
    """math/polynomial.rs:1374-1399 on device buffers (out: batch * order * width words)."""
    coeffs, out = _t(coeffs, "coeffs"), _t(out, "out")
    if coeffs.numel() != n_coeffs * batch * width or out.numel() != order * batch * width:
        raise ValueError("buffer sizes do not match n_coeffs/order/batch/width")
    fn = _lib.lib().tf_coset_eval_bfe_dev if width == 1 else _lib.lib().tf_coset_eval_xfe_dev
    _chk(fn(_p(coeffs), n_coeffs, C.c_uint64(offset_raw), _p(out), order, batch, _stream(stream)), "fast_coset_evaluate")


def tip5_permute_(states, stream=None) -> None:
    states = _t(states, "states")
    _chk(_lib.lib().tf_tip5_permute_dev(_p(states), states.numel() // 16, _stream(stream)), "Tip5::permutation")


def tip5_hash_pairs(inp, out, stream=None) -> None:
    inp, out = _t(inp, "in"), _t(out, "out")
    count = inp.numel() // 10
    if out.numel() != count * 5:
        raise ValueError("out must hold 5 words per input pair")
    _chk(_lib.lib().tf_tip5_hash_pairs_dev(_p(inp), _p(out), count, _stream(stream)), "Tip5::hash_pair")


def tip5_hash_varlen_rows(rows, row_len: int, out, stream=None) -> None:
    rows, out = _t(rows, "rows"), _t(out, "out")
    n_rows = out.numel() // 5
    _chk(_lib.lib().tf_tip5_hash_varlen_rows_dev(_p(rows), row_len, n_rows, _p(out), _stream(stream)), "Tip5::hash_varlen")


def merkle_build(leaves, n_leaves: int, nodes_out, batch: int = 1, stream=None) -> None:
    """util_types/merkle_tree.rs:165-212: nodes_out = batch x 2n digests, heap layout."""
    leaves, nodes_out = _t(leaves, "leaves"), _t(nodes_out, "nodes_out")
    if leaves.numel() != batch * n_leaves * 5 or nodes_out.numel() != batch * n_leaves * 10:
        raise ValueError("buffer sizes do not match n_leaves/batch")
    _chk(_lib.lib().tf_merkle_build_dev(_p(leaves), n_leaves, _p(nodes_out), batch, _stream(stream)), "MerkleTree::par_new")


def merkle_root(leaves, n_leaves: int, root_out, batch: int = 1, stream=None) -> None:
    leaves, root_out = _t(leaves, "leaves"), _t(root_out, "root_out")
    _chk(_lib.lib().tf_merkle_root_dev(_p(leaves), n_leaves, _p(root_out), batch, _stream(stream)), "MerkleTree::par_frugal_root")
