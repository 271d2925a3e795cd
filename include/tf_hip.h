/*
 * tf_hip.h -- C ABI of libtf_hip.so: the MI355X (gfx950) backend for the data-parallel hot path of
 * Neptune-Crypto/twenty-first (Goldilocks NTT/iNTT, fast_coset_evaluate, Tip5, Merkle builder).
 *
 * The reference has no FFI of its own (it is a pure-Rust crate); each entry point below replaces the
 * body of one Rust function, cited as file:line relative to twenty-first/src/.  INTEGRATION.md shows
 * the Rust `extern "C"` shim a maintainer would add.
 *
 * Data contract (identical to the reference's in-memory layout, so `&mut [BFieldElement]` can be
 * passed as `*mut u64` without conversion):
 *   BFieldElement  = 1 little-endian u64: x * 2^64 mod p, canonical (< p)   math/b_field_element.rs:84-86 (#[repr(transparent)])
 *   XFieldElement  = 3 consecutive BFieldElements [c0, c1, c2]              math/x_field_element.rs:56-59 (#[repr(transparent)])
 *   Digest         = 5 consecutive BFieldElements                           tip5/digest.rs:29
 *   Tip5 state     = 16 consecutive BFieldElements                          tip5/mod.rs:159-165
 * Inputs must be canonical (true for anything built through BFieldElement::new); outputs always are.
 * All pointers 8-byte aligned.  Results are bit-identical to the reference's CPU path.
 *
 * Two flavours of every entry point:
 *   tf_xxx(...)            host pointers; copies in, runs on the current HIP device, copies out, returns when done.
 *   tf_xxx_dev(..., stream) device pointers (memory of the current HIP device); work is enqueued on `stream`
 *                          (a hipStream_t, NULL = default stream) and the call returns without synchronising.
 * Every function is re-entrant and may be called concurrently from many host threads
 * (the reference's ntt is called from rayon workers, math/ntt.rs:250-274).
 *
 * There is NO CPU fallback: without a usable HIP device every call returns TF_ERR_NO_DEVICE.
 *
 * Return value: 0 on success, otherwise one of the codes below.  Codes 1-3 are the reference's
 * MerkleTreeError variants (util_types/merkle_tree.rs:933-965); codes 4-6 replace panics.
 * No C++ exception leaves the library: every status-returning entry point catches what its host-side C++ (tables, caches,
 * worker threads) may throw and returns TF_ERR_OUT_OF_MEMORY (a failed host allocation) or TF_ERR_INTERNAL, with the
 * message in tf_last_error() (csrc/tf_guard.h) -- a Rust `extern "C"` caller never sees an unwind.
 */
#ifndef TF_HIP_H
#define TF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum tf_status {
    TF_OK = 0,
    TF_ERR_TOO_FEW_LEAFS = 1,              /* MerkleTreeError::TooFewLeafs            merkle_tree.rs:394-396 */
    TF_ERR_INCORRECT_NUMBER_OF_LEAFS = 2,  /* MerkleTreeError::IncorrectNumberOfLeafs merkle_tree.rs:398-401 */
    TF_ERR_TREE_TOO_HIGH = 3,              /* MerkleTreeError::TreeTooHigh (allocation failure) :405-410 */
    TF_ERR_LEN_NOT_POWER_OF_TWO = 4,       /* ntt/intt panic: assert!(len == 0 || len.is_power_of_two()) math/ntt.rs:137 */
    TF_ERR_LEN_TOO_LARGE = 5,              /* ntt/intt panic: len > u32::MAX (math/ntt.rs:134-139), i.e. a power of two above 2^31 */
    TF_ERR_ORDER_NOT_ABOVE_DEGREE = 6,     /* fast_coset_evaluate panic: order <= degree  math/polynomial.rs:1388-1392 */
    TF_ERR_NULL_POINTER = 7,
    TF_ERR_NO_DEVICE = 8,                  /* no HIP device / HIP runtime unusable */
    TF_ERR_HIP = 9,                        /* a HIP call failed; see tf_last_error() */
    TF_ERR_OUT_OF_MEMORY = 10,
    TF_ERR_LEAF_INDEX_INVALID = 11,        /* MerkleTreeError::LeafIndexInvalid  merkle_tree.rs:486-488 */
    TF_ERR_INVERSE_OF_ZERO = 12,           /* offset.inverse() of zero panics    b_field_element.rs:264-268 */
    TF_ERR_BUFFER_TOO_SMALL = 13,
    TF_ERR_EMPTY_DOMAIN = 14,              /* interpolate panic: "interpolation must happen through more than zero points"  math/polynomial.rs:1503-1506 */
    TF_ERR_DIVISION_BY_ZERO = 15,          /* naive_divide panic: "divisor should be non-zero"  math/polynomial.rs:556-559 */
    TF_ERR_DIVISION_NOT_CLEAN = 16,        /* clean_divide panic: the quotient does not come back to the base field  math/polynomial.rs:2374, :2410 */
    TF_ERR_INVALID_ARGUMENT = 17,          /* an index / count argument of a host-logic helper (tf_shard_range, tf_merkle_subtree_layer_range) is out of range */
    TF_ERR_INTERNAL = 18                   /* a C++ exception other than an allocation failure was caught at the ABI (csrc/tf_guard.h); see tf_last_error() */
};

/* Human-readable name of a status code. */
const char *tf_status_string(int status);
/* Text of the last HIP failure on the calling thread ("" if none). */
const char *tf_last_error(void);
/* Library/ABI version (major * 1000 + minor). */
int tf_version(void);
/* First 16 hex digits of the SHA-256 of the sources this library was built from (csrc/Makefile); "-ab" appended for the
 * laboratory build (TF_AB_BUILD).  Stored rocprof records carry it so that a figure is never quoted for another build. */
const char *tf_source_hash(void);
/* Number of visible HIP devices (0 if the runtime is unusable). */
int tf_device_count(void);
/* What the library keeps in HBM for speed, per device and for the life of the process: work space between the passes of the
 * multi-pass transforms (at most 12 GiB), inter-pass twiddle tables (at most 4 GiB), coset power tables (at most 1 GiB), and
 * the freed blocks of the memory pool its stream-ordered temporaries come from (a pool of the library's own per device: the
 * application's default pool and its attributes are left alone.  Only if the runtime refuses to create a pool does the library
 * fall back to the device's default pool, and then it changes ONE attribute of it -- hipMemPoolReuseFollowEventDependencies
 * off, which correctness needs when several streams share the library's scratch cache -- and nothing else).
 * tf_release_caches() waits for the current device and frees all of it (the next call rebuilds what it needs); meant for hosts
 * that share the GPU with other users of its memory.  Call it when no other thread is inside the library on that device: a
 * call in flight on another host thread may hold a pointer to a table this frees. */
int tf_release_caches(void);

/* ---------------------------------------------------------------------------------------------
 * Devices.  Every entry point of this header runs on the calling thread's CURRENT HIP device (hipSetDevice semantics: per host
 * thread, default 0); the library keeps its tables, pools and streams per device, and any number of host threads may be inside it
 * on the same or on different devices.  tf_set_device / tf_get_device are that selector for callers that do not link the HIP
 * runtime themselves (a Rust crate binds only this library).
 * Errors: device < 0 or >= tf_device_count() -> TF_ERR_NO_DEVICE.
 */
int tf_set_device(int device);
int tf_get_device(int *device);

/* ---------------------------------------------------------------------------------------------
 * Warm-up.  The FIRST call of a shape on a device builds that shape's tables, opens kernel attributes, uploads the Tip5 constants and
 * creates the library's pool / side streams / scratch on that device -- steps that wait for the device and may synchronise it as a whole.
 * tf_prepare_* runs the shape once on zeroed scratch of the same size on the CURRENT device and returns when it is done; afterwards every
 * tf_*_dev call of that shape (same n, batch, width, direction / offset) on that device only enqueues work on the caller's stream.  A host
 * thread that drives several GPUs round-robin ("eight GPUs from one thread") calls these once per device at start-up.
 * Errors: as the call they prepare; width not 1 / 3 -> TF_ERR_INVALID_ARGUMENT; scratch allocation -> TF_ERR_OUT_OF_MEMORY. */
int tf_prepare_ntt(size_t n, size_t batch, int width, int inverse);
int tf_prepare_coset_eval(size_t n_coeffs, uint64_t offset_raw, size_t order, size_t batch, int width);
int tf_prepare_merkle(size_t n_leaves, size_t batch);

/* ---------------------------------------------------------------------------------------------
 * One host-resident batch over several GPUs.      replaces  the rayon fan-out over independent units in the reference's callers:
 *                                                            par_iter over polynomials around ntt / fast_coset_evaluate
 *                                                            (math/ntt.rs:250-274 is the unit), MerkleTree::par_new's
 *                                                            thread split util_types/merkle_tree.rs:165-212
 * Same arguments and results as tf_ntt_{bfe,xfe}, tf_coset_eval_{bfe,xfe}, tf_merkle_{build,root} on HOST pointers, plus a device
 * list: `devices` = n_devices device indices, or NULL for every visible device (n_devices is then ignored).  The `batch` units
 * (transforms, polynomials, trees) are independent; they are cut into n_devices contiguous slices by tf_shard_range's rule
 * (slice g takes units [g*base + min(g, extra), ...) with base = batch / G, extra = batch % G: the first `extra` slices hold one
 * more unit) and one worker thread per slice runs the single-device entry point on devices[g]: its own stream, its own H2D /
 * compute / D2H, all slices concurrently -- a host-resident batch crosses every listed GPU's PCIe link at once.  Nothing is
 * exchanged between devices and no collective library is involved; every result lands in the caller's buffer at its unit's
 * offset, so the output is word for word that of the single-device call.  A device may be listed more than once (several
 * workers, i.e. several copy/compute streams, on one GPU).  The calling thread's current device is not changed.
 * Errors: argument errors exactly as the single-device call (checked before any worker starts); a device index out of range ->
 * TF_ERR_NO_DEVICE; if workers fail, the status of the first failing slice in batch order is returned and tf_last_error() names
 * the device and slice.  tf_shard_range: n_shards <= 0 or shard outside [0, n_shards) -> TF_ERR_INVALID_ARGUMENT.
 *
 * FEWER TREES THAN DEVICES (in particular ONE tree): tf_merkle_{build,root}_multi cut every tree into S subtrees, S the largest power of
 * two with batch * S <= n_devices and at least two leaves per subtree -- exactly the cut MerkleTree::par_new makes over its threads
 * (util_types/merkle_tree.rs:165-212; layer l of subtree s of S is the run [(S + s) 2^l, (S + s + 1) 2^l) of the heap-ordered node
 * array, subtrees_mut :247-275).  Unit u = (tree u / S, subtree u % S); the batch * S units are dealt to the listed devices by
 * tf_shard_range's rule; every worker copies its subtrees' layers straight to their place in the caller's node array; the S subtree
 * roots of a tree (40 bytes each) are gathered on the host and the top log2 S layers are built on devices[0].  Same words as the
 * single-device call.  tf_merkle_multi_subtrees returns S for a shape (host logic, no device touched); tf_merkle_subtree_layer_range
 * the node-index run of one layer of one subtree (errors: TF_ERR_INVALID_ARGUMENT for a subtree / layer that does not exist, the
 * leaf-count errors of tf_merkle_build).
 */
int tf_shard_range(size_t total_units, int n_shards, int shard, size_t *begin, size_t *end);
int tf_merkle_multi_subtrees(size_t n_leaves, size_t batch, int n_devices);
int tf_merkle_subtree_layer_range(size_t n_leaves, size_t n_subtrees, size_t subtree, unsigned layer, size_t *begin, size_t *end);
int tf_ntt_bfe_multi(uint64_t *x, size_t n, size_t batch, int inverse, const int *devices, int n_devices);
int tf_ntt_xfe_multi(uint64_t *x, size_t n, size_t batch, int inverse, const int *devices, int n_devices);
int tf_coset_eval_bfe_multi(const uint64_t *coeffs, size_t n_coeffs, uint64_t offset_raw, uint64_t *out, size_t order, size_t batch,
                            const int *devices, int n_devices);
int tf_coset_eval_xfe_multi(const uint64_t *coeffs, size_t n_coeffs, uint64_t offset_raw, uint64_t *out, size_t order, size_t batch,
                            const int *devices, int n_devices);
int tf_merkle_build_multi(const uint64_t *leaves, size_t n_leaves, uint64_t *nodes_out, size_t batch, const int *devices, int n_devices);
int tf_merkle_root_multi(const uint64_t *leaves, size_t n_leaves, uint64_t *root_out, size_t batch, const int *devices, int n_devices);

/* ---------------------------------------------------------------------------------------------
 * NTT / iNTT            replaces  pub fn ntt<FF>(x: &mut [FF])   math/ntt.rs:67-82
 *                                 pub fn intt<FF>(x: &mut [FF])  math/ntt.rs:109-125
 * x: `batch` contiguous slices of n elements each, transformed in place, natural order in and out.
 * batch = 1 reproduces one Rust call; n = 0 and n = 1 are no-ops exactly as in the reference.
 * inverse != 0 selects intt (w^-1 twiddles and the n^-1 unscale of ntt.rs:220-228).
 * Errors: n not 0/power of two -> TF_ERR_LEN_NOT_POWER_OF_TWO; n > 2^31 -> TF_ERR_LEN_TOO_LARGE.
 */
int tf_ntt_bfe(uint64_t *x, size_t n, size_t batch, int inverse);
int tf_ntt_xfe(uint64_t *x /* 3n words per slice */, size_t n, size_t batch, int inverse);
int tf_ntt_bfe_dev(uint64_t *d_x, size_t n, size_t batch, int inverse, void *stream);
int tf_ntt_xfe_dev(uint64_t *d_x, size_t n, size_t batch, int inverse, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Coset evaluation      replaces  Polynomial::fast_coset_evaluate(&self, offset, order) -> Vec<FF>
 *                                 math/polynomial.rs:1374-1399  (= scale :760-773, zero-pad, ntt)
 * coeffs: `batch` polynomials of n_coeffs coefficients each (low to high degree), read only;
 * out:    `batch` x `order` evaluations: out[i] = f(offset * w_order^i).
 * offset_raw is a BFieldElement (raw Montgomery word); an XFieldElement offset stays on the caller's side.
 * Errors: n_coeffs > order -> TF_ERR_ORDER_NOT_ABOVE_DEGREE (trim leading zero coefficients first, as
 * Polynomial::degree() does); order not a power of two / too large as for tf_ntt_*.
 */
int tf_coset_eval_bfe(const uint64_t *coeffs, size_t n_coeffs, uint64_t offset_raw, uint64_t *out, size_t order, size_t batch);
int tf_coset_eval_xfe(const uint64_t *coeffs, size_t n_coeffs, uint64_t offset_raw, uint64_t *out, size_t order, size_t batch);
int tf_coset_eval_bfe_dev(const uint64_t *d_coeffs, size_t n_coeffs, uint64_t offset_raw, uint64_t *d_out, size_t order, size_t batch, void *stream);
int tf_coset_eval_xfe_dev(const uint64_t *d_coeffs, size_t n_coeffs, uint64_t offset_raw, uint64_t *d_out, size_t order, size_t batch, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Tip5                  replaces  Tip5::permutation  tip5/mod.rs:529-533   (states: count x 16 words, in place)
 *                                 Tip5::hash_10 / hash_pair  :559-586       (in: count x 10 words, out: count x 5)
 *                                 Tip5::hash_varlen  :617-623 (+ sponge.rs:41-55)   one digest per row
 *                                 Tip5::trace  :538-548   trace: count x 6 x 16 words (the state before the permutation and
 *                                                          after each of the 5 rounds); states end permuted, as `&mut self` does
 * Batching is the only reason to cross the boundary; a single hash_pair belongs on the CPU.
 */
int tf_tip5_permute(uint64_t *states, size_t count);
int tf_tip5_trace(uint64_t *states, uint64_t *trace, size_t count);
int tf_tip5_trace_dev(uint64_t *d_states, uint64_t *d_trace, size_t count, void *stream);
int tf_tip5_hash_pairs(const uint64_t *in, uint64_t *out, size_t count);
int tf_tip5_hash_varlen_rows(const uint64_t *rows, size_t row_len, size_t n_rows, uint64_t *out);
int tf_tip5_permute_dev(uint64_t *d_states, size_t count, void *stream);
int tf_tip5_hash_pairs_dev(const uint64_t *d_in, uint64_t *d_out, size_t count, void *stream);
int tf_tip5_hash_varlen_rows_dev(const uint64_t *d_rows, size_t row_len, size_t n_rows, uint64_t *d_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Merkle tree           replaces  MerkleTree::par_new / sequential_new   util_types/merkle_tree.rs:149-212
 *                                 MerkleTree::par_frugal_root / sequential_frugal_root   :299-364
 * leaves:    `batch` x n_leaves digests.
 * nodes_out: `batch` x 2*n_leaves digests in the reference's heap layout: nodes[0] = all-zero dummy,
 *            nodes[1] = root, nodes[i] = hash_pair(nodes[2i], nodes[2i+1]), leaves at nodes[n..2n).
 * root_out:  `batch` digests.
 * Errors: n_leaves == 0 -> TF_ERR_TOO_FEW_LEAFS (as par_new/sequential_new :394-396 and sequential_frugal_root
 * :300-302; par_frugal_root reports IncorrectNumberOfLeafs for 0 leaves, :333-335 -- its shim checks that first);
 * not a power of two -> TF_ERR_INCORRECT_NUMBER_OF_LEAFS; device allocation failure -> TF_ERR_TREE_TOO_HIGH.
 */
int tf_merkle_build(const uint64_t *leaves, size_t n_leaves, uint64_t *nodes_out, size_t batch);
int tf_merkle_root(const uint64_t *leaves, size_t n_leaves, uint64_t *root_out, size_t batch);
int tf_merkle_build_dev(const uint64_t *d_leaves, size_t n_leaves, uint64_t *d_nodes_out, size_t batch, void *stream);
int tf_merkle_root_dev(const uint64_t *d_leaves, size_t n_leaves, uint64_t *d_root_out, size_t batch, void *stream);

/* ---------------------------------------------------------------------------------------------
 * "Next" rows of the scope table: the callers on either side of the path, kept in HBM.
 *
 * Coset interpolation   replaces  Polynomial::fast_coset_interpolate(offset, values)  math/polynomial.rs:1907-1918
 *   values: batch x n evaluations on {offset * w_n^i}; out: batch x n coefficients (intt, then coefficient j
 *   times offset^-j, fused into the last pass).  offset_raw == 0 -> TF_ERR_INVERSE_OF_ZERO.
 * Hadamard product      the pointwise product inside fast_multiply  math/polynomial.rs:920-925
 *   (BFieldElement b_field_element.rs:755-762; XFieldElement x_field_element.rs:512-536); out may alias a or b.
 * Polynomial product    replaces  Polynomial::fast_multiply  math/polynomial.rs:900-932
 *   a: batch x na coefficients, b: batch x nb, out: batch x (na + nb - 1); zero-pad to the next power of two,
 *   ntt both, pointwise product, intt, truncate -- all on the device.  na == 0 or nb == 0: nothing is written.
 * Low-degree extension  fast_coset_interpolate followed by fast_coset_evaluate with the coefficients staying in HBM:
 *   values on {offset_in * w_n^i} -> values on {offset_out * w_m^i}, m >= n, both powers of two.
 * Rows -> Merkle tree   Tip5::hash_varlen of every row (tip5/mod.rs:617-623) written straight into the leaf level
 *   of the tree (util_types/merkle_tree.rs:165-212); rows: batch x n_rows x row_len words; nodes_out as tf_merkle_build.
 * Authentication structure  replaces MerkleTree::authentication_structure_node_indices / authentication_structure
 *   util_types/merkle_tree.rs:449-504, :614-622: node indices (needed minus computable, descending) and the gather of
 *   those digests from a device-resident node array.  *out_count receives the number of nodes; leaf index >=
 *   num_leafs -> TF_ERR_LEAF_INDEX_INVALID; num_leafs not a power of two -> TF_ERR_INCORRECT_NUMBER_OF_LEAFS.
 *   tf_merkle_authentication_structure_dev synchronises `stream` (its result is host data).
 */
int tf_coset_interpolate_bfe(const uint64_t *values, size_t n, uint64_t offset_raw, uint64_t *out, size_t batch);
int tf_coset_interpolate_xfe(const uint64_t *values, size_t n, uint64_t offset_raw, uint64_t *out, size_t batch);
int tf_coset_interpolate_bfe_dev(const uint64_t *d_values, size_t n, uint64_t offset_raw, uint64_t *d_out, size_t batch, void *stream);
int tf_coset_interpolate_xfe_dev(const uint64_t *d_values, size_t n, uint64_t offset_raw, uint64_t *d_out, size_t batch, void *stream);
int tf_hadamard_bfe_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count, void *stream);
int tf_hadamard_xfe_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count, void *stream);
int tf_poly_mul_bfe(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out, size_t batch);
int tf_poly_mul_xfe(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out, size_t batch);
int tf_poly_mul_bfe_dev(const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, uint64_t *d_out, size_t batch, void *stream);
int tf_poly_mul_xfe_dev(const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, uint64_t *d_out, size_t batch, void *stream);
/* Polynomial::fast_square  math/polynomial.rs:780-798: out = batch x (2 na - 1) coefficients (one forward transform). */
int tf_poly_square_bfe(const uint64_t *a, size_t na, uint64_t *out, size_t batch);
int tf_poly_square_xfe(const uint64_t *a, size_t na, uint64_t *out, size_t batch);
int tf_poly_square_bfe_dev(const uint64_t *d_a, size_t na, uint64_t *d_out, size_t batch, void *stream);
int tf_poly_square_xfe_dev(const uint64_t *d_a, size_t na, uint64_t *d_out, size_t batch, void *stream);
/* fast_multiply (math/polynomial.rs:900-932) of `batch` polynomials of na coefficients each (packed) by ONE polynomial b -- a table of
 * numerators times the same zerofier: b is transformed once.  out: batch x (na + nb - 1) coefficients.  Device-resident only. */
int tf_poly_mul_shared_bfe_dev(const uint64_t *d_a, size_t na, size_t batch, const uint64_t *d_b, size_t nb, uint64_t *d_out, void *stream);
int tf_poly_mul_shared_xfe_dev(const uint64_t *d_a, size_t na, size_t batch, const uint64_t *d_b, size_t nb, uint64_t *d_out, void *stream);
int tf_lde_bfe_dev(const uint64_t *d_values, size_t n, uint64_t offset_in_raw, uint64_t *d_out, size_t m, uint64_t offset_out_raw, size_t batch, void *stream);
int tf_lde_xfe_dev(const uint64_t *d_values, size_t n, uint64_t offset_in_raw, uint64_t *d_out, size_t m, uint64_t offset_out_raw, size_t batch, void *stream);
/* Polynomial::batch_evaluate / iterative_batch_evaluate  math/polynomial.rs:1840-1894: out[i] = f(points[i]),
 * points in the same field as the coefficients (bfe: 1 word per point, xfe: 3).  Few points / short polynomials: Horner per
 * point; many points on a long polynomial: remaindering down a zerofier tree built from batched fast_multiply calls
 * (O((n + m) log^2 m), the reference's divide_and_conquer_batch_evaluate :1882-1894 / math/zerofier_tree.rs). */
int tf_poly_batch_evaluate_bfe(const uint64_t *coeffs, size_t n_coeffs, const uint64_t *points, size_t n_points, uint64_t *out);
int tf_poly_batch_evaluate_xfe(const uint64_t *coeffs, size_t n_coeffs, const uint64_t *points, size_t n_points, uint64_t *out);
int tf_poly_batch_evaluate_bfe_dev(const uint64_t *d_coeffs, size_t n_coeffs, const uint64_t *d_points, size_t n_points, uint64_t *d_out, void *stream);
int tf_poly_batch_evaluate_xfe_dev(const uint64_t *d_coeffs, size_t n_coeffs, const uint64_t *d_points, size_t n_points, uint64_t *d_out, void *stream);
/* Polynomial::zerofier / par_zerofier  math/polynomial.rs:1435-1485 (smart_zerofier :1462, fast_zerofier :1478): the monic
 * prod_i (x - roots[i]); out receives n_roots + 1 coefficients, low to high (out[n_roots] = 1; n_roots = 0 gives the constant 1).
 * Repeated roots are allowed.  On the device: the root of the zerofier tree of the batch evaluation above. */
int tf_poly_zerofier_bfe(const uint64_t *roots, size_t n_roots, uint64_t *out);
int tf_poly_zerofier_xfe(const uint64_t *roots, size_t n_roots, uint64_t *out);
int tf_poly_zerofier_bfe_dev(const uint64_t *d_roots, size_t n_roots, uint64_t *d_out, void *stream);
int tf_poly_zerofier_xfe_dev(const uint64_t *d_roots, size_t n_roots, uint64_t *d_out, void *stream);
/* Polynomial::interpolate / par_interpolate / lagrange_interpolate / fast_interpolate  math/polynomial.rs:1502-1701, and
 * batch_fast_interpolate :1703-1838 (`rows` value rows over one domain, the tree and the inverse weights shared as the reference
 * memoises them): out[row * n_points + j] = coefficient j of the unique polynomial of degree < n_points through
 * (domain[i], values[row * n_points + i]).  Always n_points coefficients per row: the reference trims leading zero coefficients
 * in Polynomial::new, the caller does that -- the length here is data independent.
 * Errors: n_points == 0 -> TF_ERR_EMPTY_DOMAIN (:1503-1506); a repeated domain point -> TF_ERR_INVERSE_OF_ZERO (the reference
 * panics dividing by zero: traits.rs:106 / b_field_element.rs:264-268).  The _dev calls synchronise the stream once (the
 * repeated-point check); tf_poly_interpolate_*_dev_async (below) report it through a device status word instead. */
int tf_poly_interpolate_bfe(const uint64_t *domain, const uint64_t *values, size_t n_points, size_t rows, uint64_t *out);
int tf_poly_interpolate_xfe(const uint64_t *domain, const uint64_t *values, size_t n_points, size_t rows, uint64_t *out);
int tf_poly_interpolate_bfe_dev(const uint64_t *d_domain, const uint64_t *d_values, size_t n_points, size_t rows, uint64_t *d_out, void *stream);
int tf_poly_interpolate_xfe_dev(const uint64_t *d_domain, const uint64_t *d_values, size_t n_points, size_t rows, uint64_t *d_out, void *stream);
/* fast_coset_evaluate / fast_coset_interpolate of XFieldElement polynomials with an XFieldElement OFFSET (S = XFieldElement in
 * math/polynomial.rs:1374-1378 and :1907-1911; offset = 3 raw words).  Same arguments, errors and layout as tf_coset_eval_xfe /
 * tf_coset_interpolate_xfe; a zero offset -> TF_ERR_INVERSE_OF_ZERO in the interpolation (x_field_element.rs:371-375).  The
 * scaling by offset^i is a separate device pass here (the reference's docs recommend a base-field offset, :1366-1368, which is
 * the fused fast path). */
int tf_coset_eval_xfe_xoffset(const uint64_t *coeffs, size_t n_coeffs, const uint64_t offset[3], uint64_t *out, size_t order, size_t batch);
int tf_coset_interpolate_xfe_xoffset(const uint64_t *values, size_t n, const uint64_t offset[3], uint64_t *out, size_t batch);
int tf_coset_eval_xfe_xoffset_dev(const uint64_t *d_coeffs, size_t n_coeffs, const uint64_t offset[3], uint64_t *d_out, size_t order, size_t batch, void *stream);
int tf_coset_interpolate_xfe_xoffset_dev(const uint64_t *d_values, size_t n, const uint64_t offset[3], uint64_t *d_out, size_t batch, void *stream);
/* Polynomial<BFieldElement>::evaluate::<XFieldElement, XFieldElement>  math/polynomial.rs:309-320 (the generic evaluate with the
 * indeterminate in the extension field), batched: `batch` base-field polynomials of n_coeffs packed coefficients at n_points
 * XFieldElement points (3 words each) -> out[(b * n_points + i) * 3].  Horner; the coefficients are read as 8-byte words, not
 * lifted. */
int tf_poly_evaluate_bfe_at_xfe(const uint64_t *coeffs, size_t n_coeffs, size_t batch, const uint64_t *points, size_t n_points, uint64_t *out);
int tf_poly_evaluate_bfe_at_xfe_dev(const uint64_t *d_coeffs, size_t n_coeffs, size_t batch, const uint64_t *d_points, size_t n_points, uint64_t *d_out, void *stream);
/* barycentric_evaluate  math/polynomial.rs:2609-2637: out[b] = the value at `indeterminate` of the interpolant of codeword b given
 * on the subgroup of order n (natural order, no offset), for `batch` codewords of n elements at ONE indeterminate -- the
 * out-of-domain evaluation of every column of a table.  indeterminate: 3 raw words (an XFieldElement; a BFieldElement x as
 * [x, 0, 0]); _bfe / _xfe is the codewords' field; out: batch x 3 words (an XFieldElement each; limbs 1 and 2 are zero when
 * codewords and indeterminate are in the base field, the reference's BFieldElement result being limb 0).  One pass over the
 * codewords (8 / 24 bytes per element) against the shared weights d_i / (x - d_i).
 * Errors where the reference panics: n not a power of two -> TF_ERR_LEN_NOT_POWER_OF_TWO (primitive_root_of_unity(..).unwrap(),
 * :2620); the indeterminate inside the subgroup, or n == 0 -> TF_ERR_INVERSE_OF_ZERO (batch_inversion / inverse of zero). */
int tf_barycentric_evaluate_bfe(const uint64_t *codewords, size_t n, size_t batch, const uint64_t indeterminate[3], uint64_t *out);
int tf_barycentric_evaluate_xfe(const uint64_t *codewords, size_t n, size_t batch, const uint64_t indeterminate[3], uint64_t *out);
/* (at most 65 534 x 8 codewords per call: more -> TF_ERR_LEN_TOO_LARGE; split the batch) */
int tf_barycentric_evaluate_bfe_dev(const uint64_t *d_codewords, size_t n, size_t batch, const uint64_t indeterminate[3], uint64_t *d_out, void *stream);
int tf_barycentric_evaluate_xfe_dev(const uint64_t *d_codewords, size_t n, size_t batch, const uint64_t indeterminate[3], uint64_t *d_out, void *stream);
/* Polynomial::<BFieldElement>::clean_divide  math/polynomial.rs:2358-2411: the quotient a / b of a division KNOWN to be clean
 * (b | a), by pointwise division on a coset of the extension field: two forward XFE transforms of order
 * next_power_of_two(na), one inverse.  a, b: normalised coefficient arrays (na, nb count up to the non-zero leading coefficient,
 * as Polynomial::degree does); out receives na - nb + 1 coefficients.
 * Errors (where the reference panics): nb == 0 -> TF_ERR_DIVISION_BY_ZERO; the division is not clean (including
 * 0 < na < nb) -> TF_ERR_DIVISION_NOT_CLEAN (the reference: "might panic or produce a wrong result"; here it is always
 * detected).  na == 0 (zero dividend): TF_OK, nothing written.  The _dev call synchronises its stream once.
 * Where this differs from the reference: (1) the reference takes the coset route only for divisors of degree >= 512 and long
 * division below (:2360-2364); here every divisor takes the coset route.  A divisor with a root ON the coset x * <w_order>
 * (x^3 - x + 1 and its relatives y^3 - w^2i y + w^3i) cannot be inverted there: the blocking calls repeat the division once on
 * the coset (x + 1) * <w_order> (a clean quotient is the same on any coset) and only then report TF_ERR_INVERSE_OF_ZERO, the
 * code of the reference's batch_inversion panic; the _dev_async variant reports it after the first coset.  (2) 0 < na < nb is
 * TF_ERR_DIVISION_NOT_CLEAN, where a release build of the reference returns the zero quotient.  (3) An unclean division is
 * always detected. */
int tf_poly_clean_divide_bfe(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out);
int tf_poly_clean_divide_bfe_dev(const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, uint64_t *d_out, void *stream);
/* The same for `batch` dividends of na coefficients each (packed; na counts up to the longest dividend, shorter ones zero padded)
 * over ONE divisor -- a prover's quotients: many numerators over the same zerofier.  The divisor's transform is inverted once.
 * out: batch x (na - nb + 1) coefficients.  Any unclean row fails the call.  At most 65 535 dividends per call. */
int tf_poly_clean_divide_many_bfe(const uint64_t *a, size_t na, size_t batch, const uint64_t *b, size_t nb, uint64_t *out);
int tf_poly_clean_divide_many_bfe_dev(const uint64_t *d_a, size_t na, size_t batch, const uint64_t *d_b, size_t nb, uint64_t *d_out, void *stream);
/* ZerofierTree  math/zerofier_tree.rs (new_from_domain :66-87, zerofier :93-99) with Polynomial::divide_and_conquer_batch_evaluate
 * math/polynomial.rs:1882-1894: the tree of a domain built ONCE and kept in HBM (levels, the cached level transforms, the root,
 * the domain, and after the first interpolation the inverse weights), for callers that evaluate or interpolate on the same
 * points many times -- the build is 1.4 walks, and an interpolation over a prepared tree is one walk up.  A handle belongs to the
 * device that was current when it was made; every call brings its own stream-ordered work space, so one handle serves concurrent
 * calls on different streams.  An empty domain gives the empty tree (zerofier 1, no values; :136-138).
 * tf_zerofier_tree_new_*: domain of n points (host pointer; _dev: device pointer, the call synchronises `stream` before returning).
 * tf_zerofier_tree_zerofier*: n + 1 coefficients.   tf_zerofier_tree_batch_evaluate*: `batch` polynomials of n_coeffs packed
 * coefficients -> out[(b * n + i) * width] = f_b(domain[i]).   tf_zerofier_tree_interpolate*: `rows` value rows -> rows * n
 * coefficients (errors as tf_poly_interpolate_*; the first call computes the weights and synchronises its stream once). */
typedef struct tf_zerofier_tree tf_zerofier_tree;
int tf_zerofier_tree_new_bfe(const uint64_t *domain, size_t n_points, tf_zerofier_tree **tree);
int tf_zerofier_tree_new_xfe(const uint64_t *domain, size_t n_points, tf_zerofier_tree **tree);
int tf_zerofier_tree_new_bfe_dev(const uint64_t *d_domain, size_t n_points, void *stream, tf_zerofier_tree **tree);
int tf_zerofier_tree_new_xfe_dev(const uint64_t *d_domain, size_t n_points, void *stream, tf_zerofier_tree **tree);
void tf_zerofier_tree_free(tf_zerofier_tree *tree);
size_t tf_zerofier_tree_num_points(const tf_zerofier_tree *tree);
int tf_zerofier_tree_width(const tf_zerofier_tree *tree);   /* 1 = BFieldElement, 3 = XFieldElement */
int tf_zerofier_tree_zerofier(const tf_zerofier_tree *tree, uint64_t *out);
int tf_zerofier_tree_batch_evaluate(const tf_zerofier_tree *tree, const uint64_t *coeffs, size_t n_coeffs, size_t batch, uint64_t *out);
int tf_zerofier_tree_interpolate(tf_zerofier_tree *tree, const uint64_t *values, size_t rows, uint64_t *out);
int tf_zerofier_tree_zerofier_dev(const tf_zerofier_tree *tree, uint64_t *d_out, void *stream);
int tf_zerofier_tree_batch_evaluate_dev(const tf_zerofier_tree *tree, const uint64_t *d_coeffs, size_t n_coeffs, size_t batch, uint64_t *d_out, void *stream);
int tf_zerofier_tree_interpolate_dev(tf_zerofier_tree *tree, const uint64_t *d_values, size_t rows, uint64_t *d_out, void *stream);
/* ---- enqueue-and-return variants of the three _dev entry points above that otherwise synchronise their stream ------------------
 * The reference PANICS on a repeated interpolation point (traits.rs:106), on a divisor with a root on the division coset and on
 * an unclean division (polynomial.rs:2374, :2410).  The plain _dev calls detect these by copying a flag back, i.e. they block the
 * host once; the _async variants never synchronise: they take `d_status`, ONE int in device memory owned by the caller (zero it
 * before the first call of a chain), and a panic case writes its tf_status code there -- the first non-zero code written wins, so
 * one word can serve a whole chain of calls; the output of a call that reported an error is unspecified.  Everything that can be
 * checked on the host (null pointers, lengths, an empty domain, a zero divisor length) is still the return value.
 * tf_zerofier_tree_new_*_dev_async returns the handle without waiting for the build: until the caller has synchronised (or
 * ordered its other streams behind `stream` with an event) the handle may only be used on `stream`; the same holds for the
 * inverse weights the first tf_zerofier_tree_interpolate_dev_async computes.  A handle whose weights met a repeated point keeps
 * reporting TF_ERR_INVERSE_OF_ZERO through d_status on every later asynchronous interpolation. */
int tf_poly_interpolate_bfe_dev_async(const uint64_t *d_domain, const uint64_t *d_values, size_t n_points, size_t rows, uint64_t *d_out, void *stream, int *d_status);
int tf_poly_interpolate_xfe_dev_async(const uint64_t *d_domain, const uint64_t *d_values, size_t n_points, size_t rows, uint64_t *d_out, void *stream, int *d_status);
int tf_poly_clean_divide_bfe_dev_async(const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, uint64_t *d_out, void *stream, int *d_status);
int tf_poly_clean_divide_many_bfe_dev_async(const uint64_t *d_a, size_t na, size_t batch, const uint64_t *d_b, size_t nb, uint64_t *d_out, void *stream, int *d_status);
int tf_zerofier_tree_new_bfe_dev_async(const uint64_t *d_domain, size_t n_points, void *stream, tf_zerofier_tree **tree);
int tf_zerofier_tree_new_xfe_dev_async(const uint64_t *d_domain, size_t n_points, void *stream, tf_zerofier_tree **tree);
int tf_zerofier_tree_interpolate_dev_async(tf_zerofier_tree *tree, const uint64_t *d_values, size_t rows, uint64_t *d_out, void *stream, int *d_status);
/* The route tf_poly_batch_evaluate_* takes for a shape: 1 = Horner, 2 = zerofier tree (0: width not 1 / 3).  Pure host logic (the
 * fitted cost model of the router), checked by the CPU tests; honours tf_set_batch_eval_route / TF_BATCH_EVAL. */
int tf_batch_eval_plan(size_t n_coeffs, size_t n_points, size_t batch, int width);
/* Route of the batch evaluation (test / A-B hook): 0 = automatic (zerofier tree for many points on a long polynomial, Horner
 * otherwise), 1 = always Horner, 2 = the zerofier tree whenever it applies (at least two leaves: 512 points over BFE, 256 over XFE).  Same values either way. */
void tf_set_batch_eval_route(int route);
/* Polynomial::coset_extrapolate :2117-2128 / batch_coset_extrapolate :2196-2208 (and the par_ variant :2262): for each of
 * `batch` codewords of length n (a power of two, else TF_ERR_LEN_NOT_POWER_OF_TWO) given on {offset * w_n^i}, the values of
 * its interpolant at `points` (same field as the codeword): out[(b * n_points + i) * width]. */
int tf_coset_extrapolate_bfe(uint64_t offset_raw, const uint64_t *codewords, size_t n, size_t batch, const uint64_t *points, size_t n_points, uint64_t *out);
int tf_coset_extrapolate_xfe(uint64_t offset_raw, const uint64_t *codewords, size_t n, size_t batch, const uint64_t *points, size_t n_points, uint64_t *out);
int tf_coset_extrapolate_bfe_dev(uint64_t offset_raw, const uint64_t *d_codewords, size_t n, size_t batch, const uint64_t *d_points, size_t n_points, uint64_t *d_out, void *stream);
int tf_coset_extrapolate_xfe_dev(uint64_t offset_raw, const uint64_t *d_codewords, size_t n, size_t batch, const uint64_t *d_points, size_t n_points, uint64_t *d_out, void *stream);
/* Rows of a COLUMN-major table (the layout a batch of coset evaluations leaves behind: one codeword per column).
 * table: `batch` tables of n_cols columns; column j = n_rows elements of `width` words (1 = BFieldElement, 3 = XFieldElement
 * flattened as math/x_field_element.rs:217-231) at table + j * col_stride (col_stride >= n_rows * width, in words; tables are
 * n_cols * col_stride words apart).  Row i = the concatenation over the columns of element i.
 *   tf_tip5_hash_table_rows : digests[(t * n_rows + i) * 5 ..] = Tip5::hash_varlen(row i of table t)   tip5/mod.rs:617-623
 *   tf_merkle_from_columns  : those digests as the leaves of one tree per table (nodes: batch x 2 n_rows digests), errors as
 *                             tf_merkle_build. */
int tf_tip5_hash_table_rows(const uint64_t *table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t *digests, size_t batch);
int tf_tip5_hash_table_rows_dev(const uint64_t *d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t *d_digests, size_t batch, void *stream);
int tf_merkle_from_columns(const uint64_t *table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t *nodes_out, size_t batch);
int tf_merkle_from_columns_dev(const uint64_t *d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t *d_nodes_out, size_t batch, void *stream);
int tf_merkle_from_rows(const uint64_t *rows, size_t row_len, size_t n_rows, uint64_t *nodes_out, size_t batch);
int tf_merkle_from_rows_dev(const uint64_t *d_rows, size_t row_len, size_t n_rows, uint64_t *d_nodes_out, size_t batch, void *stream);
/* *out_count always receives the full count.  out_indices == NULL or capacity == 0 is the sizing call (returns TF_OK);
 * otherwise capacity < count -> TF_ERR_BUFFER_TOO_SMALL and nothing is written (same rule as the _dev sibling below). */
int tf_merkle_auth_structure_indices(size_t num_leafs, const uint64_t *leaf_indices, size_t k, uint64_t *out_indices, size_t capacity, size_t *out_count);
int tf_merkle_authentication_structure_dev(const uint64_t *d_nodes, size_t num_leafs, const uint64_t *leaf_indices, size_t k,
                                           uint64_t *out_digests, size_t capacity_digests, size_t *out_count, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Deployment settings (process-wide).  These two are the ONLY environment variables the product library reads (once, at the
 * first call); every other TF_* switch of DESIGN_HISTORY.md exists in the laboratory build alone (TF_AB_BUILD, below).
 *   TF_NTT_TILE_BYTES : bytes of batch processed between the passes of a multi-pass NTT (scratch size),
 *                       (default 2 GiB: measured on MI355X the pass kernels are VALU-bound and larger
 *                       launches overlap better than Infinity-Cache-sized ones).
 */
void tf_set_ntt_tile_bytes(size_t bytes);
size_t tf_get_ntt_tile_bytes(void);
/*   TF_NTT_PIPE       : K = 1..4 side streams the batch tiles of a multi-pass NTT are dealt to round-robin (each with its own
 *                       scratch tile), so the column pass of tile t + 1 overlaps the transposing pass of tile t and a tile
 *                       sized for the Infinity Cache is re-read out of it; the caller's stream forks/joins with events. */
void tf_set_ntt_pipe(int streams);  /* 0 = automatic (the default: two streams for the two-pass plans of 2^21 / 2^22 points when a call has
                                     * several batch tiles, one otherwise; profiles/r06_pipe_tiles.txt).  All callers of a device share
                                     * its K side streams: concurrent callers stay correct (event forks / joins) but wait for each other */
int tf_get_ntt_pipe(void);

/* ---------------------------------------------------------------------------------------------
 * Test hooks.  The planner picks between code paths by shape (number of passes, tile geometry, latency-shaped kernels, the
 * batch-evaluation route); each hook forces one side so that the parity tests can run BOTH against the oracle at any size.  Every
 * setting produces the same words.  Process-wide, not meant for production callers.
 *
 * Plan at least `passes` (2..4) global passes whenever n >= 32^passes, so the three- and four-pass paths (normally
 * n > 2^22 and n = 2^31) can be checked at small sizes.  0 restores the automatic plan. */
void tf_set_ntt_min_passes(int passes);
/* Calls with little work (<= 2^21 words) are planned with narrower tiles (256-thread workgroups); -1 = automatic
 * (default), 0 = never, 1 = always. */
void tf_set_ntt_small_launch(int mode);
/* Transforms of 2^21 and 2^22 points run in TWO global passes (a 2048-point pass = pairs of 1024-point workgroups sharing their
 * input) instead of three; -1 = automatic (default), 0 = never (the three-pass plan, which also serves the shapes the
 * two-pass plan does not: small launches, truncated products), 1 = the 2048-point pairs whenever the shape supports them.
 * (2: laboratory build only -- every FORWARD 2^22-point transform on the 1024 x 4096 plan whose last pass runs as four 1024-point
 * classes per tile, a measured loss, profiles/r04_c4_plan_ab.txt; the product library treats 2 as 1.)
 * 3 = the two-pass plan with its FIRST pass as one workgroup per 2048-row x 8-column tile (every element loaded and scaled once,
 * DESIGN 4.1b, round 6) wherever the two-pass plan applies; the automatic plan takes that kernel for coset evaluations of 2^22 points,
 * where it measured faster (profiles/r06_c4_cols8_ab.txt). */
void tf_set_ntt_two_pass(int mode);
/* The latency-shaped kernels (8 elements per thread, radix-8 stages through LDS) instead of the 32-elements-per-thread
 * pass kernels: ntt_lat_kernel for 64 .. 4096-point transforms in calls of up to 2^22 words (BFieldElement; 3 * 2^19 words
 * XFieldElement), ntt_lat2_kernel for 2^13 .. 2^20-point transforms in calls below a per-length threshold (tf_ntt.hip:
 * lat_wanted / lat2_wanted hold the measured crossovers).  -1 = automatic (default), 0 = never, 1 = whenever the shape allows. */
void tf_set_ntt_latency_kernel(int mode);
/* Number of transform-kernel launches one plain tf_ntt_*_dev call enqueues for this shape, by the planner's own predicates in
 * the planner's order (tiny / rows / latency-shaped kernels, whole-transform-per-workgroup kernels, tiles x passes of
 * ntt_pass_kernel, the narrow-tile three-pass plan of small 2^21 / 2^22 calls).  Diagnostic; bench.py uses it to turn a
 * HIP-event interval into an average launch duration.  Table builders of a first call are not counted. */
int tf_ntt_launch_count(size_t n, size_t batch, int width);
/* Planner introspection (no device needed): number of global passes of one n-point transform (0 for lengths ntt rejects)
 * and log2 of each pass's radix in log2_radix_out[0..3] (unused entries 0).  The radices multiply to n.  This is the plan of a
 * LARGE call (enough work for 512-thread tiles); a call small enough for the narrow tiles -- e.g. ONE 2^21-point BFieldElement
 * slice -- runs 2^21 / 2^22 points on the three-pass plan instead of {10, 11} / {11, 11}: tf_ntt_launch_count(n, batch, width)
 * is exact for a given batch. */
int tf_ntt_plan(size_t n, int width, int* log2_radix_out);
/* Measurement helper: the shader clock (MHz) the current device is running at right now (one-wave ~0.5 ms spin; < 0 on failure). */
double tf_debug_sclk_mhz(void);

#ifdef TF_AB_BUILD
/* ---------------------------------------------------------------------------------------------
 * Laboratory build only (csrc: make ab -> libtf_hip_ab.so).  The product library neither exports these nor contains the kernels
 * behind them: measured losers and diagnostics that are kept so that every claim of DESIGN_HISTORY.md stays reproducible
 * (tools/switch_matrix.sh runs the GPU suite under each switch against this library).
 *   TF_NTT_NT / tf_set_ntt_nt : bit 0 = non-temporal loads of the caller's input in the first pass, bit 1 = non-temporal stores
 *                               of the result in the last pass of a plain multi-pass transform (generic kernels).
 *   tf_set_ntt_chain          : the R = 1024 column pass as a chain of k tiles per workgroup, the next tile's loads issued inside
 *                               the store phase of the current one (TF_NTT_PERSIST): +11..22 % time, profiles/r03_chain_ab.txt.
 *   tf_debug_stamps           : per-wave phase cycle stamps of the NTT pass kernel (TF_NTT_ABLATE=3, tools/phase_timeline.py).
 *   environment               : TF_NTT_NO_* / TF_NTT_WG_THREADS / TF_NTT_ABLATE / TF_TREE_* / TF_POLY_MUL_NO_FUSE /
 *                               TF_POOL_REUSE_FOLLOW_EVENTS ... (the list is `grep ab_env csrc/`). */
void tf_set_ntt_nt(int mask);
void tf_set_ntt_chain(int tiles_per_workgroup);
int tf_debug_stamps(unsigned long long *host_out, size_t words);
#endif
/* Synthetic inputs for benches/tests: d_out[i] = BFieldElement::new(splitmix64(seed ^ (first_index + i)) mod p),
 * raw Montgomery words, generated on the device (the oracle's tfo_fill_random is the same counter-based sequence). */
int tf_debug_fill_random_dev(uint64_t *d_out, size_t count, uint64_t seed, uint64_t first_index, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TF_HIP_H */
