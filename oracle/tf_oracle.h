/*
 * tf_oracle.h -- CPU oracle for the twenty-first hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the reference algorithms (Neptune-Crypto/twenty-first
 * v2.0.2, paths relative to twenty-first/src/).  It exists to CHECK the HIP product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Nothing under twenty-first_amd/ links, imports or calls it.
 *
 * Pinning: the reference is Rust and cannot be built here (no rustc/cargo); this oracle is
 * pinned by the reference's own known-answer tests (tests/test_oracle_kat.py lists them
 * with file:line).
 *
 * Conventions (same as the reference):
 *   BFE    = one u64 holding x * 2^64 mod p (Montgomery form), always < p   (math/b_field_element.rs:84-86)
 *   XFE    = 3 consecutive BFEs [c0, c1, c2]                                (math/x_field_element.rs:56-59)
 *   Digest = 5 consecutive BFEs                                             (tip5/digest.rs:29)
 *   "raw"  = the Montgomery word;  "value" = canonical representative in [0, p).
 */
#ifndef TF_ORACLE_H
#define TF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFO_P 0xffffffff00000001ULL

/* ---- BFieldElement (math/b_field_element.rs) ---- */
uint64_t tfo_montyred(uint64_t lo, uint64_t hi);       /* :357-370, input = hi*2^64+lo */
uint64_t tfo_bfe_new(uint64_t value);                  /* :235-237 */
uint64_t tfo_bfe_value(uint64_t raw);                  /* :248-250, :334-336 */
uint64_t tfo_bfe_add(uint64_t a, uint64_t b);          /* :711-732 */
uint64_t tfo_bfe_sub(uint64_t a, uint64_t b);          /* :773-795 */
uint64_t tfo_bfe_neg(uint64_t a);                      /* :764-771 */
uint64_t tfo_bfe_mul(uint64_t a, uint64_t b);          /* :755-762 */
uint64_t tfo_bfe_mod_pow(uint64_t base, uint64_t exp); /* :340-353 */
uint64_t tfo_bfe_inverse(uint64_t a);                  /* :254-284; returns 0 for 0 (inverse_or_zero, traits.rs:39-45) */
/* primitive root of unity of order n (n = 0 or a power of two <= 2^32); raw form; 0 if none. :43-78, :814-818 */
uint64_t tfo_bfe_primitive_root(uint64_t n);

/* ---- XFieldElement (math/x_field_element.rs), all on raw words ---- */
void tfo_xfe_add(const uint64_t a[3], const uint64_t b[3], uint64_t out[3]);     /* :479-489 */
void tfo_xfe_sub(const uint64_t a[3], const uint64_t b[3], uint64_t out[3]);     /* :570-577 */
void tfo_xfe_mul(const uint64_t a[3], const uint64_t b[3], uint64_t out[3]);     /* :512-536 */
void tfo_xfe_mul_bfe(const uint64_t a[3], uint64_t b, uint64_t out[3]);          /* :540-548 */

/* ---- NTT (math/ntt.rs).  width = 1 (BFE) or 3 (XFE).  n must be 0 or a power of two <= 2^31.
 *      Returns 0 on success, nonzero where the reference panics (ntt.rs:135-140). ---- */
int tfo_ntt(uint64_t *x, size_t n, int width);   /* :67-82 + :153-215 */
int tfo_intt(uint64_t *x, size_t n, int width);  /* :109-125 + :220-228 */
/* batch of contiguous equal-length transforms, optionally over `threads` OS threads
 * (one transform per thread at a time -- what a rayon caller of ntt() does). */
int tfo_ntt_batch(uint64_t *x, size_t n, size_t batch, int width, int inverse, int threads);

/* ---- Polynomial (math/polynomial.rs) ---- */
/* scale :760-773 : c[i] *= alpha^i  (alpha a BFE) */
void tfo_poly_scale(uint64_t *coeffs, size_t n_coeffs, int width, uint64_t alpha_raw);
/* fast_coset_evaluate :1374-1399 ; out has `order` elements.  Returns nonzero where the reference panics. */
int tfo_coset_evaluate(const uint64_t *coeffs, size_t n_coeffs, int width, uint64_t offset_raw,
                       uint64_t *out, size_t order);
/* `batch` contiguous polynomials, one per thread (bench.py's all-core cpu_baseline of BASELINE configs[3]) */
int tfo_coset_evaluate_batch(const uint64_t *coeffs, size_t n_coeffs, int width, uint64_t offset_raw, uint64_t *out, size_t order,
                             size_t batch, int threads);
/* fast_coset_interpolate :1907-1918 ; values -> coefficients (n of them) */
int tfo_coset_interpolate(const uint64_t *values, size_t n, int width, uint64_t offset_raw, uint64_t *out);
/* Horner evaluation of a BFE/XFE polynomial at a BFE point (used to cross-check NTT == evaluation,
 * ntt.rs:563-579).  out has `width` words. */
void tfo_poly_eval(const uint64_t *coeffs, size_t n_coeffs, int width, uint64_t point_raw, uint64_t *out);

/* ---- Tip5 (tip5/mod.rs, util_types/sponge.rs) ---- */
void tfo_tip5_permutation(uint64_t state[16]);                              /* :529-533, round :175-181 */
void tfo_tip5_permutation_naive(uint64_t state[16]);                        /* tip5/naive.rs:26-76 */
void tfo_tip5_hash_10(const uint64_t in[10], uint64_t out[5]);              /* :559-569 */
void tfo_tip5_hash_pair(const uint64_t l[5], const uint64_t r[5], uint64_t out[5]); /* :577-586 */
void tfo_tip5_hash_varlen(const uint64_t *in, size_t len, uint64_t out[5]); /* :617-623, sponge.rs:41-55 */
void tfo_tip5_absorb(uint64_t state[16], const uint64_t in[10]);            /* :684-691 */
/* batch helpers (plain loops over the above) */
void tfo_tip5_hash_pairs(const uint64_t *in, uint64_t *out, size_t count);
void tfo_tip5_hash_varlen_rows(const uint64_t *rows, size_t row_len, size_t n_rows, uint64_t *out);
void tfo_tip5_hash_varlen_rows_par(const uint64_t *rows, size_t row_len, size_t n_rows, uint64_t *out, int threads);

/* ---- MerkleTree (util_types/merkle_tree.rs) ----
 * error codes: 1 TooFewLeafs, 2 IncorrectNumberOfLeafs, 3 TreeTooHigh (:933-965) */
int tfo_merkle_build(const uint64_t *leaves, size_t n_leaves, uint64_t *nodes /* 10*n words */); /* sequential_new :149-153 */
/* par_new :165-212 -- subtree fan-out over `threads` OS threads, then sequential top. */
int tfo_merkle_build_par(const uint64_t *leaves, size_t n_leaves, uint64_t *nodes, int threads, size_t cutoff);
int tfo_merkle_frugal_root(const uint64_t *leaves, size_t n_leaves, uint64_t root[5]); /* :299-309 via mmr_accumulator.rs:96-115 */

/* ---- "next" rows ---- */
void tfo_poly_mul_naive(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, int width, uint64_t *out); /* schoolbook */
int tfo_poly_mul_fast(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, int width, uint64_t *out);   /* polynomial.rs:900-932 */
int tfo_auth_structure_indices(size_t num_leafs, const uint64_t *leaf_indices, size_t k, uint64_t *out, size_t cap, size_t *count); /* merkle_tree.rs:449-504 */

void tfo_poly_eval_xfe_point(const uint64_t *coeffs, size_t n_coeffs, const uint64_t point[3], uint64_t out[3]); /* polynomial.rs:309-325 */

/* ---- helpers for tests/bench ---- */
uint64_t tfo_splitmix64(uint64_t *state);
/* fill `count` raw BFE words: new(splitmix64(seed ^ (b<<32) ^ i) mod p)   */
void tfo_fill_random(uint64_t *out, size_t count, uint64_t seed);
/* elements [first_index, first_index + count) of the same counter-based sequence (a rank's shard of a job-wide input) */
void tfo_fill_random_from(uint64_t *out, size_t count, uint64_t seed, uint64_t first_index);
/* digest -> 80 lowercase hex chars + NUL: canonical value, little-endian bytes (tip5/digest.rs:85-90,:144-153) */
void tfo_digest_to_hex(const uint64_t d[5], char out[81]);

#ifdef __cplusplus
}
#endif
#endif
