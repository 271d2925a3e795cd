// tip5_kernels.h -- Tip5 permutation / sponge and the Merkle level sweep for gfx950 (device side).
//
// Reference: twenty-first/src/tip5/mod.rs
//   round :175-181 = sbox_layer :184-194 (4 x split_and_lookup :197-207 on the RAW Montgomery bytes,
//   12 x x^7) -> mds_generated :210-253 -> + ROUND_CONSTANTS :68-149;  permutation :529-533;
//   hash_10 :559-569; hash_pair :577-586; hash_varlen :617-623 + util_types/sponge.rs:41-55.
// Merkle: twenty-first/src/util_types/merkle_tree.rs:149-222 (nodes[i] = hash_pair(nodes[2i], nodes[2i+1])).
//
// GPU mapping: one lane = one permutation, whole state in registers (16 x u64), 64-bit integer VALU
// only (no MFMA: there is no dense contraction).  The MDS is the plain integer circulant product
//   out[r] = sum_c M[(r-c) mod 16] * raw[c]        (tip5/naive.rs:54-68, mod.rs:154-157)
// accumulated per 32-bit half with v_mad_u64_u32 (16-bit x 32-bit + 64-bit; a half-sum is < 2^52, the
// same bound mds_generated relies on, mod.rs:244), then reduced with 2^64 = 2^32 - 1 and made canonical
// explicitly after adding the round constant (the reference gets there through a quirk of Add,
// mod.rs:222-242 / :1098-1142; the canonical result is the same word).
#pragma once

#include "gl64.h"

namespace tfk {

using gl::u32;
using gl::u64;

struct Tip5Consts {
    u64 rc[80];        // Montgomery form of ROUND_CONSTANTS (mod.rs:68-149)
    u32 lut[64];       // LOOKUP_TABLE (mod.rs:50-64) packed 4 bytes per word
};
__constant__ Tip5Consts g_tip5;

__device__ __forceinline__ constexpr u32 mds_entry(int i) {
    // MDS_MATRIX_FIRST_COLUMN, mod.rs:154-157
    constexpr u32 col[16] = {61402, 1108, 28750, 33823, 7454, 43244, 53865, 12034,
                             56951, 27521, 41351, 40901, 12021, 59689, 26798, 17845};
    return col[i & 15];
}

__device__ __forceinline__ u32 lookup4(u32 w, const unsigned char* lut) {
    u32 b0 = lut[w & 0xff], b1 = lut[(w >> 8) & 0xff], b2 = lut[(w >> 16) & 0xff], b3 = lut[w >> 24];
    return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

__device__ __forceinline__ void tip5_round(u64 (&s)[16], int round, const unsigned char* lut) {
    // S-box layer (mod.rs:184-194)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32 lo = lookup4((u32)s[i], lut), hi = lookup4((u32)(s[i] >> 32), lut);
        s[i] = ((u64)hi << 32) | lo;  // may be >= p, exactly like the reference
    }
#pragma unroll
    for (int i = 4; i < 16; i += 2) {  // x^7 = x * (x^2 * x^4), two elements per hand-scheduled product pair
        u64 sq0, sq1, qu0, qu1, t0, t1;
        gl::mont_mul2(s[i], s[i], s[i + 1], s[i + 1], sq0, sq1);
        gl::mont_mul2(sq0, sq0, sq1, sq1, qu0, qu1);
        gl::mont_mul2(sq0, qu0, sq1, qu1, t0, t1);
        gl::mont_mul2(s[i], t0, s[i + 1], t1, s[i], s[i + 1]);
    }
    // MDS on 32-bit halves + round constant
    u32 lo[16], hi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        lo[i] = (u32)s[i];
        hi[i] = (u32)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        u64 alo = 0, ahi = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const u32 m = mds_entry(16 + r - c);
            alo += (u64)m * lo[c];
            ahi += (u64)m * hi[c];
        }
        // value + round constant = alo + ahi * 2^32 + rc  (< 2^85)  as three 32-bit carry-chain words (t2 : t1 : t0), then one
        // fold of t2 * 2^64 == t2 * (2^32 - 1) and one conditional subtraction: the canonical word of the reference's
        // reduce-then-add (mod.rs:244-252, :178-180), because every step is exact
        const u64 rc = g_tip5.rc[round * 16 + r];
        unsigned c0, c1, c2, c3, c4;
        const u32 w1 = __builtin_addc((u32)(alo >> 32), (u32)ahi, 0u, &c0);
        const u32 w2 = __builtin_addc((u32)(ahi >> 32), 0u, c0, &c1);
        const u32 t0 = __builtin_addc((u32)alo, (u32)rc, 0u, &c2);
        const u32 t1 = __builtin_addc(w1, (u32)(rc >> 32), c2, &c3);
        const u32 t2 = __builtin_addc(w2, 0u, c3, &c4);  // < 2^22
        const u64 l64 = ((u64)t1 << 32) | t0;
        const u64 t = (u64)t2 * 0xffffffffu + l64;  // true value < 2^64 + 2^54
        const bool ca = t < l64;
        const u64 u = t + gl::EPS;  // t - p (mod 2^64)
        const bool cb = u < t;
        s[r] = (ca | cb) ? u : t;
    }
}

__device__ __forceinline__ void tip5_permutation(u64 (&s)[16], const unsigned char* lut) {
#pragma unroll 1
    for (int r = 0; r < 5; ++r) tip5_round(s, r, lut);
}

__device__ __forceinline__ void stage_lut(unsigned char* lut_lds) {
    // 256-byte table into LDS; every workgroup does this once.
    u32* w = reinterpret_cast<u32*>(lut_lds);
    for (int i = threadIdx.x; i < 64; i += blockDim.x) w[i] = g_tip5.lut[i];
    __syncthreads();
}

// states: count x 16 words, permuted in place (Tip5::permutation, mod.rs:529-533)
__global__ void __launch_bounds__(256) tip5_permute_kernel(u64* states, long long count) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 s[16];
    u64* p = states + i * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = p[k];
    tip5_permutation(s, lut);
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = s[k];
}

// Tip5::trace (mod.rs:538-548): trace[i][0] = the state before the permutation, trace[i][1 + r] = the state after round r;
// states[i] ends as the permuted state.  6 x 16 words per permutation, the rows a hash-table arithmetisation is filled from.
__global__ void __launch_bounds__(256) tip5_trace_kernel(u64* states, u64* trace, long long count) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 s[16];
    u64* p = states + i * 16;
    u64* t = trace + i * 96;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        s[k] = p[k];
        t[k] = s[k];
    }
#pragma unroll 1
    for (int r = 0; r < 5; ++r) {
        tip5_round(s, r, lut);
#pragma unroll
        for (int k = 0; k < 16; ++k) t[16 * (r + 1) + k] = s[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = s[k];
}

// out[i] = hash_10(in[10 i .. 10 i + 10)) = hash_pair(left, right)  (mod.rs:559-586).
// If leaf_copy != null the 10 input words are also copied there (Merkle leaf level, merkle_tree.rs:426).
// Addressing: item i belongs to tree i / per_tree; its input is in + tree * in_ts + 10 * (i % per_tree), etc.
__global__ void __launch_bounds__(256) tip5_hash_pairs_kernel(const u64* in, u64* out, u64* leaf_copy, long long count,
                                                              long long per_tree, long long in_ts, long long out_ts,
                                                              long long copy_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const long long tree = i / per_tree, j = i - tree * per_tree;
    const u64* p = in + tree * in_ts + 10 * j;
    u64 s[16];
#pragma unroll
    for (int k = 0; k < 10; ++k) s[k] = p[k];
    if (leaf_copy) {
        u64* q = leaf_copy + tree * copy_ts + 10 * j;
#pragma unroll
        for (int k = 0; k < 10; ++k) q[k] = s[k];
    }
#pragma unroll
    for (int k = 10; k < 16; ++k) s[k] = gl::ONE;  // Tip5::new(Domain::FixedLength), mod.rs:511-526
    tip5_permutation(s, lut);
    u64* o = out + tree * out_ts + 5 * j;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = s[k];
}

// hash_varlen of n_rows rows of row_len words each (mod.rs:617-623; padding sponge.rs:41-55)
// Row i belongs to tree i / per_tree; its digest goes to out + tree * out_ts + 5 * (i % per_tree)  (out_ts = 0 and
// per_tree = n_rows: a flat digest array; out_ts = 10 n, out = nodes + 5 n: straight into the leaf level of a tree).
__global__ void __launch_bounds__(256) tip5_hash_varlen_rows_kernel(const u64* rows, long long row_len, long long n_rows,
                                                                    u64* out, long long per_tree, long long out_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const u64* p = rows + i * row_len;
    u64 s[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = 0;  // Domain::VariableLength
    long long full = row_len / 10;
    for (long long c = 0; c < full; ++c) {
#pragma unroll
        for (int k = 0; k < 10; ++k) s[k] = p[c * 10 + k];  // overwrite-mode absorb, mod.rs:684-691
        tip5_permutation(s, lut);
    }
    const int rem = (int)(row_len - full * 10);
#pragma unroll
    for (int k = 0; k < 10; ++k) s[k] = (k < rem) ? p[full * 10 + k] : ((k == rem) ? gl::ONE : 0);
    tip5_permutation(s, lut);
    const long long tree = i / per_tree;
    u64* o = out + tree * out_ts + (i - tree * per_tree) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = s[k];
}

// hash_varlen of every ROW of a COLUMN-MAJOR table: column j is n_rows contiguous elements of `width` words (1 =
// BFieldElement, 3 = XFieldElement, flattened as x_field_element.rs:217-231) starting at table + j * col_stride; row i is
// the concatenation over the columns of element i's words.  This is the layout a batch of coset evaluations leaves in HBM
// (one codeword per column), and the lanes -- one row each -- read every column coalesced.
// Word w of a row is word (w % width) of the element in column w / width.
__device__ __forceinline__ u64 table_word(const u64* table, long long i, long long w, int width, long long col_stride) {
    const long long j = width == 1 ? w : (long long)(((unsigned long long)w * 0xAAAAAAABull) >> 33);  // w / 3 for w < 2^31
    const long long k = w - j * width;
    return table[j * col_stride + i * width + k];
}

__global__ void __launch_bounds__(256) tip5_hash_table_rows_kernel(const u64* table, long long n_rows, long long n_cols, int width,
                                                                   long long col_stride, long long table_stride, long long total,
                                                                   u64* out, long long out_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const long long tree = id / n_rows, i = id - tree * n_rows;
    const u64* tb = table + tree * table_stride;
    const long long row_len = n_cols * width;
    u64 s[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = 0;  // Domain::VariableLength
    const long long full = row_len / 10;
    for (long long c = 0; c < full; ++c) {
#pragma unroll
        for (int k = 0; k < 10; ++k) s[k] = table_word(tb, i, c * 10 + k, width, col_stride);  // overwrite-mode absorb, mod.rs:684-691
        tip5_permutation(s, lut);
    }
    const int rem = (int)(row_len - full * 10);
#pragma unroll
    for (int k = 0; k < 10; ++k) s[k] = (k < rem) ? table_word(tb, i, full * 10 + k, width, col_stride) : ((k == rem) ? gl::ONE : 0);
    tip5_permutation(s, lut);
    u64* o = out + tree * out_ts + i * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = s[k];
}

// ---- cooperative form: the 16 lanes of a DPP row hold the 16 state words of ONE permutation ------------------
// The lane-per-permutation kernels above are throughput-optimal but one permutation is a dependent chain of ~8 200
// instructions (~19 us), which is what every level of a Merkle tree with fewer nodes than the GPU has lanes costs.
// Here lane j of a row owns state[j]: the S-box is one lookup or one x^7 per lane, and because the MDS matrix is
// circulant, out[r] = sum_k M[k] * state[(r - k) mod 16] is 16 row rotations (v_mov_b32 row_ror:k) each followed by one
// v_mad_u64_u32 with the same constant M[k] in every lane.  ~1 600 instructions per permutation (the lookup lanes and
// the x^7 lanes take turns): 3x the total work of the lane-per-permutation form, 1/5 of its latency -- measured 4.6 us
// per tree level instead of 19 us.  Used for launches of at most kCoopMaxCount permutation chains (tf_tip5.hip).
template <int K>
__device__ __forceinline__ void mds_coop_terms(u32 lo, u32 hi, u64& alo, u64& ahi) {
    if constexpr (K < 16) {
        const u32 rl = (u32)__builtin_amdgcn_mov_dpp((int)lo, 0x120 + K, 0xf, 0xf, true);  // row_ror:K: lane r <- lane (r - K) mod 16
        const u32 rh = (u32)__builtin_amdgcn_mov_dpp((int)hi, 0x120 + K, 0xf, 0xf, true);
        alo += (u64)mds_entry(K) * rl;
        ahi += (u64)mds_entry(K) * rh;
        mds_coop_terms<K + 1>(lo, hi, alo, ahi);
    }
}

// s = state[j] of the permutation shared by the 16 lanes of this row; all 16 lanes must be active.
__device__ __forceinline__ void tip5_permutation_coop(u64& s, int j, const unsigned char* lut) {
    u64 rcs[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) rcs[r] = g_tip5.rc[r * 16 + j];
#pragma unroll 1
    for (int round = 0; round < 5; ++round) {
        if (j < 4) {  // split_and_lookup (mod.rs:197-207)
            const u32 l = lookup4((u32)s, lut), h = lookup4((u32)(s >> 32), lut);
            s = ((u64)h << 32) | l;
        } else {  // x^7
            const u64 sq = gl::mont_mul(s, s);
            const u64 qu = gl::mont_mul(sq, sq);
            s = gl::mont_mul(s, gl::mont_mul(sq, qu));
        }
        const u32 lo = (u32)s, hi = (u32)(s >> 32);
        u64 alo = (u64)mds_entry(0) * lo, ahi = (u64)mds_entry(0) * hi;
        mds_coop_terms<1>(lo, hi, alo, ahi);
        // same single-fold reduction as tip5_round
        const u64 rc = rcs[round];
        unsigned c0, c1, c2, c3, c4;
        const u32 w1 = __builtin_addc((u32)(alo >> 32), (u32)ahi, 0u, &c0);
        const u32 w2 = __builtin_addc((u32)(ahi >> 32), 0u, c0, &c1);
        const u32 t0 = __builtin_addc((u32)alo, (u32)rc, 0u, &c2);
        const u32 t1 = __builtin_addc(w1, (u32)(rc >> 32), c2, &c3);
        const u32 t2 = __builtin_addc(w2, 0u, c3, &c4);
        const u64 l64 = ((u64)t1 << 32) | t0;
        const u64 t = (u64)t2 * 0xffffffffu + l64;
        const bool ca = t < l64;
        const u64 u = t + gl::EPS;
        const bool cb = u < t;
        s = (ca | cb) ? u : t;
    }
}

// hash_pair per 16-lane row: item i = blockIdx.x * 16 + threadIdx.x / 16.  Same addressing as tip5_hash_pairs_kernel.
__global__ void __launch_bounds__(256) tip5_hash_pairs_coop_kernel(const u64* in, u64* out, long long count, long long per_tree,
                                                                   long long in_ts, long long out_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    const int j = threadIdx.x & 15;
    const long long i = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= count) return;  // whole rows leave together
    const long long tree = i / per_tree, k = i - tree * per_tree;
    u64 s = j < 10 ? in[tree * in_ts + 10 * k + j] : gl::ONE;
    tip5_permutation_coop(s, j, lut);
    if (j < 5) out[tree * out_ts + 5 * k + j] = s;
}

// hash_varlen of row i by the 16 lanes of row-group i (few rows, or one long input: the absorb chain is sequential)
__global__ void __launch_bounds__(256) tip5_hash_varlen_rows_coop_kernel(const u64* rows, long long row_len, long long n_rows,
                                                                         u64* out, long long per_tree, long long out_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    const int j = threadIdx.x & 15;
    const long long i = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= n_rows) return;
    const u64* p = rows + i * row_len;
    u64 s = 0;  // Domain::VariableLength
    const long long full = row_len / 10;
    for (long long c = 0; c < full; ++c) {
        if (j < 10) s = p[c * 10 + j];  // overwrite-mode absorb, mod.rs:684-691
        tip5_permutation_coop(s, j, lut);
    }
    const int rem = (int)(row_len - full * 10);
    if (j < 10) s = (j < rem) ? p[full * 10 + j] : ((j == rem) ? gl::ONE : 0);
    tip5_permutation_coop(s, j, lut);
    const long long tree = i / per_tree;
    if (j < 5) out[tree * out_ts + (i - tree * per_tree) * 5 + j] = s;
}

// the same for few rows: 16 lanes per row
__global__ void __launch_bounds__(256) tip5_hash_table_rows_coop_kernel(const u64* table, long long n_rows, long long n_cols, int width,
                                                                        long long col_stride, long long table_stride, long long total,
                                                                        u64* out, long long out_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    const int j = threadIdx.x & 15;
    const long long id = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (id >= total) return;
    const long long tree = id / n_rows, i = id - tree * n_rows;
    const u64* tb = table + tree * table_stride;
    const long long row_len = n_cols * width;
    u64 s = 0;
    const long long full = row_len / 10;
    for (long long c = 0; c < full; ++c) {
        if (j < 10) s = table_word(tb, i, c * 10 + j, width, col_stride);
        tip5_permutation_coop(s, j, lut);
    }
    const int rem = (int)(row_len - full * 10);
    if (j < 10) s = (j < rem) ? table_word(tb, i, full * 10 + j, width, col_stride) : ((j == rem) ? gl::ONE : 0);
    tip5_permutation_coop(s, j, lut);
    if (j < 5) out[tree * out_ts + i * 5 + j] = s;
}

// Tip5::permutation of state i by row-group i
__global__ void __launch_bounds__(256) tip5_permute_coop_kernel(u64* states, long long count) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    const int j = threadIdx.x & 15;
    const long long i = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= count) return;
    u64 s = states[i * 16 + j];
    tip5_permutation_coop(s, j, lut);
    states[i * 16 + j] = s;
}

// Top of a tree in one workgroup per tree: given level `width` (<= 256 nodes, i.e. nodes[width .. 2 width)),
// compute nodes[1 .. width) level by level through LDS and write them out; also zero nodes[0]
// (merkle_tree.rs:415-419).  Mirrors sequentially_fill_tree (:216-222) below the parallelisation cutoff.
// level_in: pointer to the `width` digests of the starting level for tree 0, stride in_ts words per tree.
// nodes: node array (may be null when only the root is wanted); root_out: 5 words per tree or null.
__global__ void __launch_bounds__(1024) merkle_top_kernel(const u64* level_in, long long in_ts, int width, u64* nodes,
                                                          long long nodes_ts, u64* root_out, const u64* leaves_to_copy,
                                                          long long leaves_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    __shared__ u64 buf[2][256 * 5];
    (void)leaves_ts;  // leaves_to_copy != null only says that level_in IS the leaf level, to be copied into nodes[width..2 width)
    stage_lut(lut);
    const long long tree = blockIdx.x;
    const int t = threadIdx.x, j = t & 15, row = t >> 4;  // 64 rows of 16 lanes: one hash_pair per row at a time
    const u64* src = level_in + tree * in_ts;
    u64* nd = nodes ? nodes + tree * nodes_ts : nullptr;
    for (int k = t; k < width * 5; k += blockDim.x) {
        u64 v = src[k];
        buf[0][k] = v;
        if (leaves_to_copy && nd) nd[(long long)width * 5 + k] = v;  // starting level is the leaf level
    }
    if (nd && t < 5) nd[t] = 0;
    __syncthreads();
    int cur = 0;
    for (int w = width / 2; w >= 1; w /= 2) {
        for (int base = 0; base < w; base += 64) {
            const int i = base + row;
            if (i < w) {  // whole rows take the branch together
                u64 s = j < 10 ? buf[cur][10 * i + j] : gl::ONE;
                tip5_permutation_coop(s, j, lut);
                if (j < 5) {
                    buf[cur ^ 1][5 * i + j] = s;
                    if (nd) nd[(long long)(w + i) * 5 + j] = s;
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (root_out && t < 5) root_out[tree * 5 + t] = buf[cur][t];
}

// out[k] = nodes[idx[k]] for digests (5 words): authentication structures from a device-resident tree
__global__ void __launch_bounds__(256) gather_digests_kernel(const u64* nodes, const unsigned long long* idx, long long count, u64* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * 5) return;
    const long long k = i / 5, w = i - 5 * k;
    out[i] = nodes[idx[k] * 5 + w];
}

}  // namespace tfk
