"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares,
fails loudly without a device, and the host-side logic (conversions, sharding) is right."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tf_hip.h")


def declared_functions(ab=False):
    """Function names the header declares: the product's (outside #ifdef TF_AB_BUILD) or the laboratory build's extras."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    ab_blocks = re.findall(r"#ifdef TF_AB_BUILD(.*?)#endif", text, flags=re.S)
    if ab:
        text = "\n".join(ab_blocks)
    else:
        text = re.sub(r"#ifdef TF_AB_BUILD.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for required in ["tf_ntt_bfe", "tf_ntt_xfe", "tf_coset_eval_bfe", "tf_coset_eval_xfe", "tf_tip5_permute",
                     "tf_tip5_hash_pairs", "tf_tip5_hash_varlen_rows", "tf_merkle_build", "tf_merkle_root"]:
        assert required in names and required + "_dev" in names


def test_library_exports_every_declared_symbol(tf):
    from twenty_first_amd import _lib

    lib = tf.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libtf_hip.so does not export {n}"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    # the laboratory hooks are declared under #ifdef TF_AB_BUILD and the product library does not export them
    ab_names = declared_functions(ab=True)
    assert sorted(_lib.AB_SIGNATURES) == ab_names
    if not _lib.is_ab_build():
        for n in ab_names:
            assert not hasattr(lib, n), f"the product library exports the laboratory hook {n}"
    assert len(lib.tf_source_hash()) >= 16
    assert lib.tf_version() >= 1001
    assert lib.tf_status_string(2) == b"TF_ERR_INCORRECT_NUMBER_OF_LEAFS"


def test_library_is_native_gfx950_code():
    """the product is a HIP code object for gfx950, not a Python fallback"""
    so = os.path.join(ROOT, "twenty-first_amd", "libtf_hip.so")
    blob = open(so, "rb").read()
    assert b"gfx950" in blob
    assert b"ntt_pass_kernel" in blob and b"tip5_hash_pairs_mx_kernel" in blob and b"tip5_hash_pairs_coop_kernel" in blob


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "twenty-first_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "tf_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_fails_loudly_without_a_device(tf):
    if tf.lib().tf_device_count() > 0:
        pytest.skip("a GPU is present")
    x = np.zeros(8, dtype=np.uint64)
    with pytest.raises(tf.TwentyFirstError) as e:
        tf.ntt(x)
    assert e.value.code == 8  # TF_ERR_NO_DEVICE
    with pytest.raises(tf.TwentyFirstError):
        tf.Tip5.hash_pairs(np.zeros(10, dtype=np.uint64))
    with pytest.raises(tf.TwentyFirstError):
        tf.MerkleTree.par_new(np.zeros(10, dtype=np.uint64))
    # argument errors are reported before any device work, exactly where the reference panics / errors
    with pytest.raises(tf.NttPanic):
        tf.ntt(np.zeros(6, dtype=np.uint64))
    with pytest.raises(tf.MerkleTreeError) as e:
        tf.MerkleTree.par_new(np.zeros(15, dtype=np.uint64))
    assert e.value.variant == "IncorrectNumberOfLeafs"
    with pytest.raises(tf.MerkleTreeError) as e:
        tf.MerkleTree.par_new(np.zeros(0, dtype=np.uint64))
    assert e.value.variant == "TooFewLeafs"


def test_bfield_conversions_match_oracle(tf, oracle):
    import random

    rng = random.Random(4)
    for _ in range(200):
        v = rng.randrange(tf.P)
        assert tf.BFieldElement.new(v) == oracle.bfe_new(v)
        assert tf.BFieldElement.value(oracle.bfe_new(v)) == v
    for n in [0, 1, 2, 4, 1 << 10, 1 << 20, 1 << 32]:
        assert tf.BFieldElement.primitive_root_of_unity(n) == oracle.primitive_root(n)
    assert tf.BFieldElement.primitive_root_of_unity(3) is None
    assert tf.BFieldElement.generator() == oracle.bfe_new(7)
    d = oracle.fill_random(5, 9)
    assert tf.Digest.to_hex(d) == oracle.digest_hex(d)


def test_polynomial_degree_and_order_check(tf):
    p = tf.Polynomial(np.array([5, 0, 7, 0, 0], dtype=np.uint64))
    assert p.degree() == 2
    assert tf.Polynomial(np.zeros(4, dtype=np.uint64)).degree() == -1
    with pytest.raises(tf.NttPanic):  # polynomial.rs:1388-1392, raised host-side before touching the device
        p.fast_coset_evaluate(tf.BFieldElement.new(7), 2)


def test_shard_ranges():
    from twenty_first_amd.sharding import all_shards, shard_range

    for total in [0, 1, 7, 8, 256, 4096, 4099]:
        for world in [1, 2, 3, 4, 8]:
            shards = all_shards(total, world)
            assert shards[0][0] == 0 and shards[-1][1] == total
            for (a, b), (c, d) in zip(shards, shards[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in shards]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(4096, 8, 3) == (1536, 2048)  # BASELINE config 5: 4096 NTTs over 8 GPUs
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_c_abi_shard_range_is_the_python_split(tf):
    """tf_shard_range (the split of the tf_*_multi entry points, include/tf_hip.h) == sharding.shard_range, ragged batches included"""
    from twenty_first_amd.sharding import shard_range

    for total in [0, 1, 2, 7, 8, 9, 255, 256, 257, 4096, 4099, (1 << 40) + 5]:
        for world in [1, 2, 3, 4, 7, 8, 16]:
            got = [tf.shard_range(total, world, r) for r in range(world)]
            assert got == [shard_range(total, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == total
    import ctypes as C
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    lib = tf.lib()
    assert lib.tf_shard_range(10, 0, 0, C.byref(lo), C.byref(hi)) == 17     # no shards: TF_ERR_INVALID_ARGUMENT (not NO_DEVICE)
    assert lib.tf_shard_range(10, 2, 2, C.byref(lo), C.byref(hi)) == 17     # shard out of range
    assert lib.tf_shard_range(10, 2, 1, None, C.byref(hi)) == 7             # TF_ERR_NULL_POINTER
    assert lib.tf_status_string(17) == b"TF_ERR_INVALID_ARGUMENT"


def test_single_tree_subtree_split_is_the_reference_s(tf):
    """tf_merkle_{build,root}_multi with fewer trees than devices cut every tree into subtrees.  The node-index run of every layer of every
    subtree (tf_merkle_subtree_layer_range) must be the slicing of MerkleTree::subtrees_mut, restated here from
    util_types/merkle_tree.rs:247-275: skip num_trees nodes, then for layer 0, 1, ... hand every tree in turn the next 2^layer nodes.
    Together the layers tile [num_trees, 2 n) exactly once (the reference's debug_assert!(nodes.is_empty())), as its own test
    merkle_subtrees_are_sliced_correctly (:1537-1592) checks.  Also the number of subtrees the multi call picks (host logic)."""
    import ctypes as C
    lib = tf.lib()
    lo, hi = C.c_size_t(0), C.c_size_t(0)

    def reference_slicing(num_leafs, num_trees):  # merkle_tree.rs:247-275
        height = num_leafs.bit_length() - 1
        sub_height = height - (num_trees.bit_length() - 1)
        layers = [[] for _ in range(num_trees)]
        at = num_trees  # nodes_to_skip: includes the dummy node at index 0
        for layer_idx in range(sub_height + 1):
            for tree in range(num_trees):
                layers[tree].append((at, at + (1 << layer_idx)))
                at += 1 << layer_idx
        assert at == 2 * num_leafs
        return layers

    for log_n in range(1, 9):
        n = 1 << log_n
        for log_s in range(0, log_n + 1):
            S = 1 << log_s
            want = reference_slicing(n, S)
            seen = []
            for sub in range(S):
                for layer, (a, b) in enumerate(want[sub]):
                    assert lib.tf_merkle_subtree_layer_range(n, S, sub, layer, C.byref(lo), C.byref(hi)) == 0
                    assert (lo.value, hi.value) == (a, b)
                    seen.extend(range(a, b))
                assert lib.tf_merkle_subtree_layer_range(n, S, sub, len(want[sub]), C.byref(lo), C.byref(hi)) == 17  # no such layer
            assert sorted(seen) == list(range(S, 2 * n))
            assert lib.tf_merkle_subtree_layer_range(n, S, S, 0, C.byref(lo), C.byref(hi)) == 17  # no such subtree
    assert lib.tf_merkle_subtree_layer_range(8, 3, 0, 0, C.byref(lo), C.byref(hi)) == 17      # not a power of two
    assert lib.tf_merkle_subtree_layer_range(12, 2, 0, 0, C.byref(lo), C.byref(hi)) == 2      # leaf-count errors as tf_merkle_build
    assert lib.tf_merkle_subtree_layer_range(0, 1, 0, 0, C.byref(lo), C.byref(hi)) == 1
    # subtrees per tree: the largest power of two with batch * S <= devices and at least two leaves per subtree (merkle_tree.rs:182)
    assert lib.tf_merkle_multi_subtrees(1 << 24, 1, 8) == 8          # BASELINE configs[2] on eight GPUs
    assert lib.tf_merkle_multi_subtrees(1 << 24, 1, 7) == 4
    assert lib.tf_merkle_multi_subtrees(1 << 20, 2, 8) == 4
    assert lib.tf_merkle_multi_subtrees(1 << 20, 3, 8) == 2
    assert lib.tf_merkle_multi_subtrees(1 << 20, 8, 8) == 1          # a tree per device: the batch split
    assert lib.tf_merkle_multi_subtrees(1 << 20, 256, 8) == 1
    assert lib.tf_merkle_multi_subtrees(4, 1, 8) == 2                # two leaves per subtree at least
    assert lib.tf_merkle_multi_subtrees(2, 1, 8) == 1 and lib.tf_merkle_multi_subtrees(1, 1, 8) == 1
    assert lib.tf_merkle_multi_subtrees(1 << 20, 0, 8) == 1


def test_multi_device_entry_points_without_a_device(tf):
    """argument errors of the tf_*_multi calls are those of the single-device calls and come first; with no GPU the rest is
    TF_ERR_NO_DEVICE -- never a CPU fallback"""
    if tf.lib().tf_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(tf.NttPanic):
        tf.ntt(np.zeros(6, dtype=np.uint64), devices="all")
    with pytest.raises(tf.MerkleTreeError) as e:
        tf.MerkleTree.build_batch(np.zeros(15, dtype=np.uint64), 3, devices=[0])
    assert e.value.variant == "IncorrectNumberOfLeafs"
    for call in (lambda: tf.ntt(np.zeros(8, dtype=np.uint64), devices="all"),
                 lambda: tf.ntt(np.zeros(8, dtype=np.uint64), devices=[0, 0]),
                 lambda: tf.fast_coset_evaluate(np.zeros(8, dtype=np.uint64), 7, 8, devices=[0]),
                 lambda: tf.MerkleTree.roots_batch(np.zeros(20, dtype=np.uint64), 2, devices="all"),
                 lambda: tf.set_device(0)):
        with pytest.raises(tf.TwentyFirstError) as e:
            call()
        assert e.value.code == 8  # TF_ERR_NO_DEVICE


def test_length_checks_precede_any_device_work(tf):
    """math/ntt.rs:134-139: the length panics are decided on the host before a device is touched, so they hold here too"""
    import ctypes as C
    lib = tf._lib.lib()
    buf = (C.c_uint64 * 4)()
    assert lib.tf_ntt_bfe_dev(buf, 12, 1, 0, None) == 4           # neither 0 nor a power of two
    assert lib.tf_ntt_bfe_dev(buf, 1 << 32, 1, 0, None) == 5      # longer than u32::MAX
    assert lib.tf_ntt_xfe_dev(buf, 1 << 33, 1, 1, None) == 5
    assert lib.tf_ntt_launch_count(1 << 31, 1, 1) == 4            # 2^31 is in range: three column passes + one last pass
    assert lib.tf_ntt_launch_count(1 << 20, 256, 1) == 2
    assert lib.tf_ntt_launch_count(1 << 32, 1, 1) == 0


def test_planner_splits_are_valid_for_every_length(tf):
    """Host logic, no device: for every length ntt accepts (math/ntt.rs:134-139) and both element widths the pass plan has
    radices in [2^5, 2^10] that multiply to n (single-pass lengths: one radix = n), also with the deeper plans forced."""
    import ctypes as C
    lib = tf._lib.lib()
    radix = (C.c_int * 4)()
    assert lib.tf_ntt_plan(12, 1, radix) == 0 and lib.tf_ntt_plan(1 << 32, 1, radix) == 0 and lib.tf_ntt_plan(1, 1, radix) == 0
    assert lib.tf_ntt_plan(1 << 20, 2, radix) == 0
    for force in (0, 3, 4):
        lib.tf_set_ntt_min_passes(force)
        try:
            for width in (1, 3):
                for log_n in range(1, 32):
                    passes = lib.tf_ntt_plan(1 << log_n, width, radix)
                    r = list(radix)
                    assert 1 <= passes <= 4 and sum(r[:passes]) == log_n and all(v == 0 for v in r[passes:]), (force, width, log_n, r)
                    if passes > 1:
                        # 2^21 / 2^22: two passes, a radix of 2^11 being a pass of paired 1024-point halves
                        two_pass = force == 0 and log_n in (21, 22)
                        assert all(5 <= v <= (11 if two_pass else 10) for v in r[:passes]), (force, width, log_n, r)
                        assert passes >= (2 if log_n <= 20 or two_pass else 3 if log_n <= 30 else 4)
                        if two_pass:
                            assert passes == 2 and r[1] == 11, (width, log_n, r)
                    else:
                        assert log_n <= 10 or (log_n <= (14 if width == 1 else 12) and force == 0)  # whole transform per workgroup
                    if force and log_n >= 5 * force:
                        assert passes >= force, (force, width, log_n, r)
        finally:
            lib.tf_set_ntt_min_passes(0)
    # the headline transform: two passes of radix 1024; launches per call follow the plan
    assert lib.tf_ntt_plan(1 << 20, 1, radix) == 2 and list(radix)[:2] == [10, 10]
    assert lib.tf_ntt_launch_count(1 << 13, 1000, 1) == 1 and lib.tf_ntt_launch_count(1 << 13, 1000, 3) == 2
    # batches of short transforms are ONE launch whatever the batch (a grid-stride walk of wave-private tiles, round 5)
    for n in (2, 4, 8, 16, 32):
        assert lib.tf_ntt_launch_count(n, 1 << 24, 1) == 1 and lib.tf_ntt_launch_count(n, 1 << 22, 3) == 1
    assert lib.tf_ntt_launch_count(64, 1 << 22, 1) == 1


def test_batch_evaluation_router_is_host_logic_with_the_measured_crossovers(tf):
    """tf_batch_eval_plan (host logic, no device): the router of tf_poly_batch_evaluate_* picks the zerofier tree where it measured
    faster than Horner (profiles/r03_batch_eval_fine_w*.txt) and Horner elsewhere; the test hook forces either route; a tree never
    applies below two leaves."""
    lib = tf._lib.lib()
    plan = lib.tf_batch_eval_plan
    lib.tf_set_batch_eval_route(0)
    try:
        # (n_coeffs, n_points, batch, width) -> route; 1 = Horner, 2 = tree
        # (round 3: profiles/r03_batch_eval_fine_w1_b.txt / _w3_b.txt -- the latency-shaped transforms and the one-launch-per-level
        #  build and walks moved every crossover down; Horner has a floor of n dependent steps however few the points)
        measured = [((1 << 12, 1 << 12, 1, 1), 1), ((1 << 14, 1 << 12, 1, 1), 1), ((1 << 14, 1 << 14, 1, 1), 1), ((1 << 14, 1 << 16, 1, 1), 1),
                    ((1 << 16, 1 << 13, 1, 1), 1), ((1 << 16, 1 << 14, 1, 1), 2), ((1 << 16, 1 << 15, 1, 1), 2), ((1 << 16, 1 << 16, 1, 1), 2),
                    ((1 << 18, 1 << 11, 1, 1), 1), ((1 << 18, 1 << 12, 1, 1), 2), ((1 << 18, 1 << 14, 1, 1), 2), ((1 << 18, 1 << 16, 1, 1), 2),
                    ((1 << 18, 1 << 18, 1, 1), 2), ((1 << 20, 1 << 9, 1, 1), 2), ((1 << 20, 1 << 10, 1, 1), 2), ((1 << 20, 1 << 12, 1, 1), 2), ((1 << 20, 1 << 20, 1, 1), 2), ((1 << 22, 1 << 16, 1, 1), 2),
                    ((1 << 12, 1 << 14, 1, 3), 1), ((1 << 14, 1 << 12, 1, 3), 1), ((1 << 14, 1 << 14, 1, 3), 2), ((1 << 16, 1 << 10, 1, 3), 1),
                    ((1 << 16, 1 << 12, 1, 3), 2), ((1 << 16, 1 << 13, 1, 3), 2), ((1 << 16, 1 << 14, 1, 3), 2),
                    ((1 << 16, 1 << 16, 1, 3), 2), ((1 << 18, 1 << 14, 1, 3), 2), ((1 << 20, 1 << 20, 1, 3), 2)]
        for shape, want in measured:
            assert plan(*shape) == want, shape
        assert plan(1 << 20, 100, 1, 1) == 1 and plan(5, 1 << 20, 1, 1) == 1 and plan(0, 1 << 20, 1, 1) == 1   # few points / a short polynomial
        assert plan(1 << 26, 1 << 10, 1, 1) == 2 and plan(1 << 16, 1 << 13, 64, 1) == 2  # many chunks / many polynomials walk the tree together
        assert plan(1 << 16, 1 << 16, 1, 2) == 0
        prev = 1                                   # at n = m the decision is monotone in the size
        for log in range(8, 24):
            r = plan(1 << log, 1 << log, 1, 1)
            assert r >= prev, log
            prev = r
        lib.tf_set_batch_eval_route(1)
        assert plan(1 << 20, 1 << 20, 1, 1) == 1
        lib.tf_set_batch_eval_route(2)
        assert plan(1 << 12, 1 << 12, 1, 1) == 2 and plan(1 << 12, 511, 1, 1) == 1 and plan(1 << 12, 255, 1, 3) == 1
    finally:
        lib.tf_set_batch_eval_route(0)


# ---------------------------------------------------------------------------------- no C++ exception crosses the C ABI (csrc/tf_guard.h)
CSRC = os.path.join(ROOT, "twenty-first_amd", "csrc")
# int-returning entry points whose value is not a status (they cannot fail and own nothing that throws)
NOT_A_STATUS = {"tf_version", "tf_device_count", "tf_get_ntt_pipe", "tf_batch_eval_plan", "tf_ntt_plan", "tf_ntt_launch_count",
                "tf_zerofier_tree_width", "tf_merkle_multi_subtrees"}


def test_every_status_returning_entry_point_is_a_function_try_block():
    """Structure check on the two translation units that define extern "C" functions: each `int tf_xxx(...)` whose value is a status opens
    with `try {` and closes with TF_ABI_CATCH, so std::bad_alloc / std::system_error / ... come back as TF_ERR_OUT_OF_MEMORY /
    TF_ERR_INTERNAL instead of unwinding into a C or Rust caller."""
    seen = set()
    for unit in ("tf_abi.hip", "tf_multi.hip"):
        text = open(os.path.join(CSRC, unit)).read()
        for m in re.finditer(r"^int (tf_[a-z0-9_]+)\(", text, flags=re.M):
            name = m.group(1)
            depth, j = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(text[j], 0)
                j += 1
            opens = text[j:j + 8].split()
            seen.add(name)
            if name in NOT_A_STATUS:
                assert opens[0] == "{", name
            else:
                assert opens[:2] == ["try", "{"], f"{name} is not guarded"
        assert text.count(" TF_ABI_CATCH") == len([n for n in re.findall(r"^int (tf_[a-z0-9_]+)\(", text, flags=re.M) if n not in NOT_A_STATUS])
    declared_ints = set(re.findall(r"^int\s+(tf_[a-z0-9_]+)\s*\(", re.sub(r"#ifdef TF_AB_BUILD.*?#endif", "", open(HEADER).read(), flags=re.S), flags=re.M))
    assert declared_ints <= seen, sorted(declared_ints - seen)


def test_abi_guard_turns_exceptions_into_statuses(tmp_path, tf):
    """The mechanism itself, compiled with g++ alone (tf_guard.h depends on the status codes only): bad_alloc / length_error ->
    TF_ERR_OUT_OF_MEMORY (10), any other exception -> TF_ERR_INTERNAL (18), no exception -> the body's own value."""
    src = tmp_path / "guard_check.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <system_error>
#include "include/tf_hip.h"
#include "twenty-first_amd/csrc/tf_guard.h"
static std::string g_msg;
namespace tfi { int abi_caught(const char* what, int status) noexcept { try { g_msg = what; } catch (...) {} return status; } }
extern "C" int f_ok(int v) try { return v; } TF_ABI_CATCH
extern "C" int f_alloc(void) try { throw std::bad_alloc(); } TF_ABI_CATCH
extern "C" int f_len(void) try { std::vector<int> v; v.reserve(v.max_size() + 1); return TF_OK; } TF_ABI_CATCH
extern "C" int f_sys(void) try { throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "thread"); } TF_ABI_CATCH
extern "C" int f_int(void) try { throw 42; } TF_ABI_CATCH
int main() {
    if (f_ok(TF_ERR_NULL_POINTER) != TF_ERR_NULL_POINTER || f_ok(TF_OK) != TF_OK) return 1;
    if (f_alloc() != TF_ERR_OUT_OF_MEMORY || g_msg.find("bad_alloc") == std::string::npos) return 2;
    if (f_len() != TF_ERR_OUT_OF_MEMORY) return 3;
    if (f_sys() != TF_ERR_INTERNAL || g_msg.find("thread") == std::string::npos) return 4;
    if (f_int() != TF_ERR_INTERNAL || g_msg != "unknown C++ exception") return 5;
    std::puts("guard ok");
    return 0;
}
''')
    import subprocess
    exe = tmp_path / "guard_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, "-o", str(exe), str(src)])
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "guard ok"
    lib = tf.lib()
    assert lib.tf_status_string(18) == b"TF_ERR_INTERNAL" and lib.tf_status_string(10) == b"TF_ERR_OUT_OF_MEMORY"
