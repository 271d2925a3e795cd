"""ctypes loader for libtf_hip.so (the C ABI of include/tf_hip.h).

The library is the product: if it is missing or does not load, importing this module raises --
there is no Python/CPU fallback for any operation.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("TF_HIP_LIBRARY") or os.path.join(_HERE, "libtf_hip.so")  # override: A/B builds only
HEADER = os.path.join(os.path.dirname(_HERE), "include", "tf_hip.h")

_u64p = C.POINTER(C.c_uint64)
_sz = C.c_size_t
_vp = C.c_void_p

# name -> (restype, argtypes); must list every function declared in include/tf_hip.h
SIGNATURES = {
    "tf_status_string": (C.c_char_p, [C.c_int]),
    "tf_last_error": (C.c_char_p, []),
    "tf_version": (C.c_int, []),
    "tf_source_hash": (C.c_char_p, []),
    "tf_device_count": (C.c_int, []),
    "tf_set_device": (C.c_int, [C.c_int]),
    "tf_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "tf_shard_range": (C.c_int, [_sz, C.c_int, C.c_int, C.POINTER(_sz), C.POINTER(_sz)]),
    "tf_merkle_multi_subtrees": (C.c_int, [_sz, _sz, C.c_int]),
    "tf_prepare_ntt": (C.c_int, [_sz, _sz, C.c_int, C.c_int]),
    "tf_prepare_coset_eval": (C.c_int, [_sz, C.c_uint64, _sz, _sz, C.c_int]),
    "tf_prepare_merkle": (C.c_int, [_sz, _sz]),
    "tf_merkle_subtree_layer_range": (C.c_int, [_sz, _sz, _sz, C.c_uint, C.POINTER(_sz), C.POINTER(_sz)]),
    "tf_ntt_bfe_multi": (C.c_int, [_vp, _sz, _sz, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "tf_ntt_xfe_multi": (C.c_int, [_vp, _sz, _sz, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "tf_coset_eval_bfe_multi": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, _sz, C.POINTER(C.c_int), C.c_int]),
    "tf_coset_eval_xfe_multi": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, _sz, C.POINTER(C.c_int), C.c_int]),
    "tf_merkle_build_multi": (C.c_int, [_vp, _sz, _vp, _sz, C.POINTER(C.c_int), C.c_int]),
    "tf_merkle_root_multi": (C.c_int, [_vp, _sz, _vp, _sz, C.POINTER(C.c_int), C.c_int]),
    "tf_ntt_bfe": (C.c_int, [_vp, _sz, _sz, C.c_int]),
    "tf_ntt_xfe": (C.c_int, [_vp, _sz, _sz, C.c_int]),
    "tf_ntt_bfe_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp]),
    "tf_ntt_xfe_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp]),
    "tf_coset_eval_bfe": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, _sz]),
    "tf_coset_eval_xfe": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, _sz]),
    "tf_coset_eval_bfe_dev": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, _sz, _vp]),
    "tf_coset_eval_xfe_dev": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, _sz, _vp]),
    "tf_tip5_permute": (C.c_int, [_vp, _sz]),
    "tf_tip5_trace": (C.c_int, [_vp, _vp, _sz]),
    "tf_tip5_trace_dev": (C.c_int, [_vp, _vp, _sz, _vp]),
    "tf_tip5_hash_pairs": (C.c_int, [_vp, _vp, _sz]),
    "tf_tip5_hash_varlen_rows": (C.c_int, [_vp, _sz, _sz, _vp]),
    "tf_tip5_permute_dev": (C.c_int, [_vp, _sz, _vp]),
    "tf_tip5_hash_pairs_dev": (C.c_int, [_vp, _vp, _sz, _vp]),
    "tf_tip5_hash_varlen_rows_dev": (C.c_int, [_vp, _sz, _sz, _vp, _vp]),
    "tf_merkle_build": (C.c_int, [_vp, _sz, _vp, _sz]),
    "tf_merkle_root": (C.c_int, [_vp, _sz, _vp, _sz]),
    "tf_merkle_build_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp]),
    "tf_merkle_root_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp]),
    "tf_coset_interpolate_bfe": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz]),
    "tf_coset_interpolate_xfe": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz]),
    "tf_coset_interpolate_bfe_dev": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, _vp]),
    "tf_coset_interpolate_xfe_dev": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, _vp]),
    "tf_hadamard_bfe_dev": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "tf_hadamard_xfe_dev": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "tf_poly_mul_bfe": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _sz]),
    "tf_poly_mul_xfe": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _sz]),
    "tf_poly_mul_bfe_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _sz, _vp]),
    "tf_poly_mul_xfe_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _sz, _vp]),
    "tf_poly_square_bfe": (C.c_int, [_vp, _sz, _vp, _sz]),
    "tf_poly_square_xfe": (C.c_int, [_vp, _sz, _vp, _sz]),
    "tf_poly_square_bfe_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp]),
    "tf_poly_square_xfe_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp]),
    "tf_lde_bfe_dev": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, C.c_uint64, _sz, _vp]),
    "tf_lde_xfe_dev": (C.c_int, [_vp, _sz, C.c_uint64, _vp, _sz, C.c_uint64, _sz, _vp]),
    "tf_poly_batch_evaluate_bfe": (C.c_int, [_vp, _sz, _vp, _sz, _vp]),
    "tf_poly_batch_evaluate_xfe": (C.c_int, [_vp, _sz, _vp, _sz, _vp]),
    "tf_poly_batch_evaluate_bfe_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "tf_poly_batch_evaluate_xfe_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "tf_coset_eval_xfe_xoffset": (C.c_int, [_vp, _sz, _vp, _vp, _sz, _sz]),
    "tf_coset_interpolate_xfe_xoffset": (C.c_int, [_vp, _sz, _vp, _vp, _sz]),
    "tf_coset_eval_xfe_xoffset_dev": (C.c_int, [_vp, _sz, _vp, _vp, _sz, _sz, _vp]),
    "tf_coset_interpolate_xfe_xoffset_dev": (C.c_int, [_vp, _sz, _vp, _vp, _sz, _vp]),
    "tf_poly_evaluate_bfe_at_xfe": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp]),
    "tf_poly_evaluate_bfe_at_xfe_dev": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "tf_barycentric_evaluate_bfe": (C.c_int, [_vp, _sz, _sz, _vp, _vp]),
    "tf_barycentric_evaluate_xfe": (C.c_int, [_vp, _sz, _sz, _vp, _vp]),
    "tf_barycentric_evaluate_bfe_dev": (C.c_int, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "tf_barycentric_evaluate_xfe_dev": (C.c_int, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "tf_poly_mul_shared_bfe_dev": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "tf_poly_mul_shared_xfe_dev": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "tf_poly_clean_divide_bfe": (C.c_int, [_vp, _sz, _vp, _sz, _vp]),
    "tf_poly_clean_divide_many_bfe": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp]),
    "tf_poly_clean_divide_many_bfe_dev": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "tf_poly_clean_divide_bfe_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "tf_zerofier_tree_new_bfe": (C.c_int, [_vp, _sz, C.POINTER(C.c_void_p)]),
    "tf_zerofier_tree_new_xfe": (C.c_int, [_vp, _sz, C.POINTER(C.c_void_p)]),
    "tf_zerofier_tree_new_bfe_dev": (C.c_int, [_vp, _sz, _vp, C.POINTER(C.c_void_p)]),
    "tf_zerofier_tree_new_xfe_dev": (C.c_int, [_vp, _sz, _vp, C.POINTER(C.c_void_p)]),
    "tf_zerofier_tree_free": (None, [_vp]),
    "tf_zerofier_tree_num_points": (_sz, [_vp]),
    "tf_zerofier_tree_width": (C.c_int, [_vp]),
    "tf_zerofier_tree_zerofier": (C.c_int, [_vp, _vp]),
    "tf_zerofier_tree_batch_evaluate": (C.c_int, [_vp, _vp, _sz, _sz, _vp]),
    "tf_zerofier_tree_interpolate": (C.c_int, [_vp, _vp, _sz, _vp]),
    "tf_zerofier_tree_zerofier_dev": (C.c_int, [_vp, _vp, _vp]),
    "tf_zerofier_tree_batch_evaluate_dev": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "tf_zerofier_tree_interpolate_dev": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "tf_poly_zerofier_bfe": (C.c_int, [_vp, _sz, _vp]),
    "tf_poly_zerofier_xfe": (C.c_int, [_vp, _sz, _vp]),
    "tf_poly_zerofier_bfe_dev": (C.c_int, [_vp, _sz, _vp, _vp]),
    "tf_poly_zerofier_xfe_dev": (C.c_int, [_vp, _sz, _vp, _vp]),
    "tf_poly_interpolate_bfe": (C.c_int, [_vp, _vp, _sz, _sz, _vp]),
    "tf_poly_interpolate_xfe": (C.c_int, [_vp, _vp, _sz, _sz, _vp]),
    "tf_poly_interpolate_bfe_dev": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "tf_poly_interpolate_xfe_dev": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "tf_coset_extrapolate_bfe": (C.c_int, [C.c_uint64, _vp, _sz, _sz, _vp, _sz, _vp]),
    "tf_coset_extrapolate_xfe": (C.c_int, [C.c_uint64, _vp, _sz, _sz, _vp, _sz, _vp]),
    "tf_coset_extrapolate_bfe_dev": (C.c_int, [C.c_uint64, _vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "tf_coset_extrapolate_xfe_dev": (C.c_int, [C.c_uint64, _vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "tf_tip5_hash_table_rows": (C.c_int, [_vp, _sz, _sz, C.c_int, _sz, _vp, _sz]),
    "tf_tip5_hash_table_rows_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, _sz, _vp, _sz, _vp]),
    "tf_merkle_from_columns": (C.c_int, [_vp, _sz, _sz, C.c_int, _sz, _vp, _sz]),
    "tf_merkle_from_columns_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, _sz, _vp, _sz, _vp]),
    "tf_merkle_from_rows": (C.c_int, [_vp, _sz, _sz, _vp, _sz]),
    "tf_merkle_from_rows_dev": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp]),
    "tf_merkle_auth_structure_indices": (C.c_int, [_sz, _vp, _sz, _vp, _sz, C.POINTER(C.c_size_t)]),
    "tf_merkle_authentication_structure_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _sz, C.POINTER(C.c_size_t), _vp]),
    "tf_ntt_launch_count": (C.c_int, [_sz, _sz, C.c_int]),
    "tf_ntt_plan": (C.c_int, [_sz, C.c_int, C.POINTER(C.c_int)]),
    "tf_batch_eval_plan": (C.c_int, [_sz, _sz, _sz, C.c_int]),
    "tf_poly_interpolate_bfe_dev_async": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "tf_poly_interpolate_xfe_dev_async": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "tf_poly_clean_divide_bfe_dev_async": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "tf_poly_clean_divide_many_bfe_dev_async": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp, _vp, _vp]),
    "tf_zerofier_tree_new_bfe_dev_async": (C.c_int, [_vp, _sz, _vp, C.POINTER(C.c_void_p)]),
    "tf_zerofier_tree_new_xfe_dev_async": (C.c_int, [_vp, _sz, _vp, C.POINTER(C.c_void_p)]),
    "tf_zerofier_tree_interpolate_dev_async": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "tf_release_caches": (C.c_int, []),
    "tf_set_ntt_small_launch": (None, [C.c_int]),
    "tf_set_ntt_two_pass": (None, [C.c_int]),
    "tf_set_ntt_latency_kernel": (None, [C.c_int]),
    "tf_debug_sclk_mhz": (C.c_double, []),
    "tf_set_ntt_tile_bytes": (None, [_sz]),
    "tf_set_ntt_min_passes": (None, [C.c_int]),
    "tf_get_ntt_tile_bytes": (_sz, []),
    "tf_set_ntt_pipe": (None, [C.c_int]),
    "tf_get_ntt_pipe": (C.c_int, []),
    "tf_set_batch_eval_route": (None, [C.c_int]),
    "tf_debug_fill_random_dev": (C.c_int, [_vp, _sz, C.c_uint64, C.c_uint64, _vp]),
}


# declared under #ifdef TF_AB_BUILD in the header: present in the laboratory library (csrc: make ab, selected with
# TF_HIP_LIBRARY=.../libtf_hip_ab.so) only -- the product library neither exports them nor contains the kernels behind them
AB_SIGNATURES = {
    "tf_set_ntt_chain": (None, [C.c_int]),
    "tf_debug_stamps": (C.c_int, [_vp, _sz]),
    "tf_set_ntt_nt": (None, [C.c_int]),
}


def build(force: bool = False) -> str:
    """Compile libtf_hip.so for gfx950 with hipcc (works without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    args = ["make", "-C", csrc]
    if force:
        args.append("-B")
    args.append(f"-j{min(8, os.cpu_count() or 1)}")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return SO_PATH


_lib = None


def _load_hip_runtime() -> None:
    """libtf_hip.so is linked without DT_NEEDED on libamdhip64 (csrc/Makefile) so that the process
    holds exactly ONE HIP runtime: the one torch ships when torch is used for device memory/streams
    (two runtimes in one process cannot share a device), otherwise /opt/rocm's."""
    candidates = []
    try:
        import torch  # noqa: F401  (maps torch/lib/libamdhip64.so)

        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:  # torch absent: fine, the library itself never needs it
        pass
    candidates += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    errors = []
    for path in candidates:
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return
        except OSError as e:
            errors.append(f"{path}: {e}")
    raise ImportError("cannot load a HIP runtime (libamdhip64): " + "; ".join(errors))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        _load_hip_runtime()
        L = C.CDLL(SO_PATH)
        older = os.environ.get("TF_HIP_ALLOW_OLDER_LIBRARY") == "1"  # tools/ab.sh only: time an earlier build through this harness
        for name, (res, args) in SIGNATURES.items():
            if older and not hasattr(L, name):
                continue
            fn = getattr(L, name)  # AttributeError if the .so does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in AB_SIGNATURES.items():
            if hasattr(L, name):
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib


def is_ab_build() -> bool:
    """True when the loaded library is the laboratory build (every A/B switch and measured-loser kernel compiled in)."""
    return lib().tf_source_hash().decode().endswith("-ab")
