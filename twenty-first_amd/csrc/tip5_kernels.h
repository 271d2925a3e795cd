// tip5_kernels.h -- Tip5 permutation / sponge and the Merkle level sweep for gfx950 (device side).
//
// Reference: twenty-first/src/tip5/mod.rs
//   round :175-181 = sbox_layer :184-194 (4 x split_and_lookup :197-207 on the RAW Montgomery bytes,
//   12 x x^7) -> mds_generated :210-253 -> + ROUND_CONSTANTS :68-149;  permutation :529-533;
//   hash_10 :559-569; hash_pair :577-586; hash_varlen :617-623 + util_types/sponge.rs:41-55.
// Merkle: twenty-first/src/util_types/merkle_tree.rs:149-222 (nodes[i] = hash_pair(nodes[2i], nodes[2i+1])).
//
// Three formulations of the same round, all producing the reference's words:
//   * matrix-pipe form (throughput; every launch of more than 2^13 permutation chains): FOUR lanes per permutation, the MDS --
//     the one dense contraction on this path, out = M * state with the constant circulant M (tip5/naive.rs:54-68, mod.rs:154-157)
//     -- on the matrix pipe: v_mfma_i32_16x16x64_i8 on byte planes since round 6 (round 5: v_mfma_f64_16x16x4_f64, kept as -DTF_TIP5_I8=0)
//   * cooperative form (latency; small launches and the top of a tree): 16 lanes of a DPP row per permutation, the circulant as
//     16 row rotations + v_mad_u64_u32; where half the rows would idle anyway, a row PAIR per permutation (the rotation terms split
//     over the two rows, v_permlane16_swap_b32 joins them: tip5_permutation_coop2)
//   * lane-per-permutation form (tip5_round below): the whole state in one lane's registers, 512 v_mad_u64_u32 per MDS.  It was
//     the throughput kernel through round 4; the library no longer launches it -- tools/microbench_mds.hip keeps it as the
//     yardstick the matrix-pipe form is measured against (profiles/r05_microbench_mds_mfma.txt).
// In every form a half-sum is reduced with 2^64 = 2^32 - 1 and made canonical explicitly after adding the round constant (the
// reference gets there through a quirk of Add, mod.rs:222-242 / :1098-1142; the canonical result is the same word).
#pragma once

#include <cstddef>
#include <type_traits>

#include "gl64.h"

namespace tfk {

using gl::u32;
using gl::u64;

struct Tip5Consts {
    u64 rc[80];        // Montgomery form of ROUND_CONSTANTS (mod.rs:68-149)
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
    // LOOKUP_TABLE (mod.rs:50-64) into LDS, once per workgroup, from its definition L(x) = ((x + 1)^3 mod 257) - 1
    // (mod.rs:1022-1026; the reference's lookup_table_is_correct test :1035-1053 pins the literal table to it): a dozen integer
    // instructions per entry instead of a trip to memory, so a small launch does not wait for a table before it waits for its input.
    // Every Tip5 kernel runs workgroups of at least 256 threads: thread x writes entry x.
    if (threadIdx.x < 256) {
        const u32 v = threadIdx.x + 1u;
        lut_lds[threadIdx.x] = (unsigned char)((v * v * v) % 257u - 1u);  // (x + 1)^3 mod 257 is never 0 for x < 256
    }
    __syncthreads();
}

// Rows of a COLUMN-MAJOR table (tip5_hash_table_rows_*_kernel): column j is n_rows contiguous elements of `width` words (1 =
// BFieldElement, 3 = XFieldElement, flattened as x_field_element.rs:217-231) starting at table + j * col_stride; row i is
// the concatenation over the columns of element i's words.  This is the layout a batch of coset evaluations leaves in HBM
// (one codeword per column); the 16 columns of a wave are 16 consecutive rows, so every table column is read in 128-byte runs.
// Word w of a row is word (w % width) of the element in column w / width.
__device__ __forceinline__ u64 table_word(const u64* table, long long i, long long w, int width, long long col_stride) {
    const long long j = width == 1 ? w : (long long)(((unsigned long long)w * 0xAAAAAAABull) >> 33);  // w / 3 for w < 2^31
    const long long k = w - j * width;
    return table[j * col_stride + i * width + k];
}

// ---- matrix-pipe form: FOUR lanes hold one permutation, the MDS runs on v_mfma_f64_16x16x4_f64 ---------------------------------
// The MDS layer is the one dense contraction on this path: a constant 16 x 16 matrix times the state (tip5/naive.rs:54-68,
// mod.rs:210-253).  A wave is a 16-column x 4-quarter grid: lane l = (column j = l & 15, quarter q = l >> 4) holds the words
// 4 i + q (i = 0..3) of the permutation in column j, so register 0 of every lane is a split_and_lookup word and registers 1..3 are
// x^7 words -- no divergence in the S-box layer.  For the MDS each 32-bit half of a word becomes an f64 (v_cvt_f64_u32) and is the
// B operand of K-block i (B[k][j]: lane (j, k) -- exactly where the word lives); A is the circulant itself, A_i[r][k] =
// M[(r - 4 i - k) mod 16], one constant per lane and K-block; D[r][j] comes back in lane (j, r & 3) register r >> 2 -- again exactly
// the strided layout (MI355X f64 C/D map: row = (lane >> 4) + 4 * reg).  Products are < 2^48 and a sum of 16 is < 2^51.01, so the
// f64 arithmetic is exact; the accumulator starts at 2^52 + (half of the round constant, adjusted), which makes the raw bits of
// the result  0x433 << 52 | integer sum.  The recombination works on those raw bits: with B = 0x433 << 52,
//     rawlo + 2^32 rawhi = (lo-sum + 2^32 hi-sum + x) + B (1 + 2^32),     x = (rc - B (1 + 2^32)) mod p  split into the two starts,
// so   h1 * (2^32 - 1) + rawlo  (one v_mad_u64_u32; < 2^64: no carry)  + h0 * 2^32  (one add with carry k)  is congruent to
// MDS(state)[r] + rc[r], and one conditional "+ (2^32 - 1)" (carry, or >= p) makes it the canonical word -- the word the
// lane-per-permutation round above produces.  8 MFMA per 16 permutations and round; measured (tools/microbench_mds.hip,
// profiles/r05_microbench_mds_mfma.txt) the f64 matrix pipe does NOT overlap with the vector ALU on gfx950 (it runs at the vector
// f64 rate and blocks VALU issue), so the gain is what the layout saves -- 8 MFMA (512 cycles) replace 128 v_mad_u64_u32 (~600
// cycles) per 16 permutations, the round constant and the 85-bit recombination shrink to 6 instructions per word -- not a second pipe.
#ifndef TF_TIP5_I8
#define TF_TIP5_I8 1  // 1: the MDS on v_mfma_i32_16x16x64_i8 (round 6); 0: on v_mfma_f64_16x16x4_f64 (round 5; the yardstick of tools/microbench_mds.hip)
#endif
typedef double d4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
#if TF_TIP5_I8
// ---- round 6: the MDS on the i8 matrix pipe --------------------------------------------------------------------------------------
// Same lane layout.  The state goes in byte by byte, the matrix entries as three SIGNED base-256 digits (M = m0 + 256 m1 + 65536 m2,
// |m| <= 128), and output plane p = a + b (0..9) collects
//     P_p[r][j] = sum_c sum_{a <= 2} m_a[(r - c) mod 16] * d_{p - a}[c][j],   d_b = (byte b of word c) - 128 = the byte XOR 0x80 as an i8,
// on v_mfma_i32_16x16x64_i8 with K = (c, b) (16 words x 4 bytes), at most 48 non-zero products per plane, |P_p| < 2^20, so
//     MDS(state)[r] = sum_p 256^p P_p + 128 * (sum of the row) * (2^64 - 1) / 255.
//   B operands: NO byte shuffling.  B_lo = the low dwords of the lane's four words XOR 0x80808080 (bytes b = 0..3 at K slot (i, b)), B_hi =
//     the high dwords (bytes 4..7): two v_xor per word and MDS layer, nothing else.  The plane is selected on the CONSTANT side instead:
//   A operand of plane p: digit p - b of M[(pi(r') - (4 i + q)) mod 16] at byte 4 i + b (zero where p - b is not a digit), p = 0..5 -- six
//     constant operands (24 VGPRs) that serve B_lo for planes 0..5 and, as A_{p-4}, B_hi for planes 4..9.  Planes 4 and 5 take both (their
//     second MFMA accumulates onto the first): TWELVE MFMA per layer.  [The first i8 form of this round kept ONE A operand and built a
//     byte window [d_p, d_{p-1}, d_{p-2}, 0] per plane and word with v_perm_b32: ten MFMA but forty permutes per lane and layer -- 12 % of
//     the round's vector instructions, on the pipe that binds it; the matrix pipe has the room (profiles/r06_microbench_mds_i8.txt).]
//     A and B index K by the same function of (lane >> 4, byte), whatever the hardware's is, so only "lane & 15 = row of A / column of B"
//     is assumed; the row permutation pi(r') = 4 (r' & 3) + (r' >> 2) makes D row r' (lane (j, r' >> 2), register r' & 3) the word
//     4 t + q of THIS lane's layout: nothing moves between lanes;
//   C operand (from LDS): 2^21 + byte p of (rc + K1 - K2) mod p, K1 the constant above, K2 = 2^21 sum_p 256^p: every plane comes back
//     non-negative (< 2^22) and the round constant costs nothing.
// Recombination per word: L0 = Q0 + 2^8 Q1 + 2^16 Q2 + 2^24 Q3 (one v_lshl_add_u32, two v_mad_u64_u32), L1 from Q4..Q7, L2 = Q8 + 2^8 Q9;
//   value = L0 + 2^32 L1 + 2^64 L2 = L0 + 2^32 lo(L1) + (2^32 - 1) (hi(L1) + L2) (mod p): one add, one v_mad_u64_u32 (< 2^64), then the same
//   "add to the high word, fold the carry" tail as the f64 form (mx_fold4_tail).  Ten MFMA of 16 cycles instead of eight of 64, and unlike
//   the f64 pipe the i8 pipe runs BESIDE the vector ALU (profiles/r05_mfma_valu_mix.txt).  Measured (tools/microbench_mds.hip,
//   profiles/r06_microbench_mds_i8.txt): MDS layer alone 49.6 -> 73.8 G layers/s, whole permutations 5.57 -> 6.42 G/s (+15 %), every word
//   that of 128-bit arithmetic / of the lane-per-permutation round.
constexpr u32 kMdsCol[16] = {61402, 1108, 28750, 33823, 7454, 43244, 53865, 12034, 56951, 27521, 41351, 40901, 12021, 59689, 26798, 17845};
struct Tip5MxConsts {
    int c[5][10][4][4];  // accumulator starts [round][plane][quarter q][register t] for state word 4 t + q
    int cf[10][4][4];    // round 0 of a FIXED-LENGTH hash (hash_10 / hash_pair): words 12..15 (register 3 of every lane) are the constant 1
                         // there (Tip5::new(Domain::FixedLength), mod.rs:511-526; 1^7 = 1): their bytes are left out of B and their MDS
                         // contribution rides here
    int cz[10][4][4];    // round 0 of the first permutation of a variable-length sponge: words 12..15 are 0, left out of B
    int a[6][64][4];     // the A operand of plane p (0..5) for lane l
};
__constant__ Tip5MxConsts g_tip5_mx;
struct MxA {
    v4i p[6];
};

inline int mds_digit(u32 M, int a) {  // signed base-256 digits of a matrix entry
    const int m0 = (int)(signed char)(M & 0xff);
    const u32 M1 = (u32)((int)M - m0) >> 8;
    const int m1 = (int)(signed char)(M1 & 0xff);
    const u32 M2 = (u32)((int)M1 - m1) >> 8;
    return a == 0 ? m0 : (a == 1 ? m1 : (int)M2);
}
// host side of the table above; rc_mont = Montgomery form of ROUND_CONSTANTS (the g_tip5.rc words)
inline void fill_tip5_mx(Tip5MxConsts& t, const u64* rc_mont) {
    typedef unsigned __int128 u128;
    const u128 ones8 = 0x0101010101010101ULL;  // sum_{b < 8} 256^b
    u128 k2 = 0;
    for (int p = 9; p >= 0; --p) k2 = (k2 * 256 + ((u128)1 << 21)) % gl::P;
    const u64 K2 = (u64)k2;
    // skip = 1: the words 12..15 are not in B; fixed_one = their value is ONE (0x00000000ffffffff, Montgomery 1) instead of 0
    const auto starts = [&](int (*dst)[4][4], const u64* rc16, bool skip, bool fixed_one) {
        for (int q = 0; q < 4; ++q)
            for (int v = 0; v < 4; ++v) {
                const int r = 4 * v + q;
                u64 rowsum = 0, tail = 0;
                for (int c = 0; c < 16; ++c) (c >= 12 ? tail : rowsum) += kMdsCol[(r - c) & 15];
                if (!skip) rowsum += tail;
                u128 adj = (u128)rc16[r] + (u128)(128 * rowsum) % gl::P * (ones8 % gl::P) % gl::P + gl::P - K2;
                if (skip && fixed_one) adj += (u128)tail * 0xffffffffULL;
                const u64 x = (u64)(adj % gl::P);
                for (int p = 0; p < 10; ++p) dst[p][q][v] = (1 << 21) + (p < 8 ? (int)((x >> (8 * p)) & 0xff) : 0);
            }
    };
    for (int round = 0; round < 5; ++round) starts(t.c[round], rc_mont + 16 * round, false, false);
    starts(t.cf, rc_mont, true, true);
    starts(t.cz, rc_mont, true, false);
    for (int p = 0; p < 6; ++p)
        for (int l = 0; l < 64; ++l) {
            const int rp = l & 15, qa = l >> 4, r = 4 * (rp & 3) + (rp >> 2);
            for (int i = 0; i < 4; ++i) {
                const u32 M = kMdsCol[(r - (4 * i + qa)) & 15];
                u32 w = 0;
                for (int b = 0; b < 4; ++b)
                    if (p - b >= 0 && p - b <= 2) w |= ((u32)mds_digit(M, p - b) & 0xff) << (8 * b);
                t.a[p][l][i] = (int)w;
            }
        }
}

struct Tip5MxLds {
    int c[5][10][4][4];
    int cf[10][4][4];
    int cz[10][4][4];
    unsigned char lut[256];
};

__device__ __forceinline__ void stage_mx(Tip5MxLds* l) {
    const int* src = &g_tip5_mx.c[0][0][0][0];
    int* dst = &l->c[0][0][0][0];
    for (int i = threadIdx.x; i < (5 + 2) * 160; i += blockDim.x) dst[i] = src[i];  // c, cf, cz are contiguous in both records
    stage_lut(l->lut);  // ends in __syncthreads()
}
static_assert(offsetof(Tip5MxConsts, cf) == 5 * 160 * 4 && offsetof(Tip5MxConsts, cz) == 6 * 160 * 4 && offsetof(Tip5MxLds, cz) == 6 * 160 * 4, "stage_mx copies c, cf, cz as one run");

__device__ __forceinline__ void mx_a_operands(const Tip5MxLds*, MxA& a) {
#pragma unroll
    for (int p = 0; p < 6; ++p) a.p[p] = *reinterpret_cast<const v4i*>(&g_tip5_mx.a[p][threadIdx.x & 63][0]);
}
#else  // ---- the f64 form of round 5 ------------------------------------------------------------------------------------------------

struct Tip5MxConsts {
    double c[5][4][8];  // accumulator starts [round][quarter q][lo-half reg 0..3 | hi-half reg 0..3] for state word 4 reg + q
    double a[16];       // MDS_MATRIX_FIRST_COLUMN as f64
    double cf[4][8];    // round 0 of a FIXED-LENGTH hash (hash_10 / hash_pair): c[0] plus the MDS contribution of words 12..15, which
                        // are the constant 1 there (Tip5::new(Domain::FixedLength), mod.rs:511-526, and 1^7 = 1)
};
__constant__ Tip5MxConsts g_tip5_mx;

// host side of the table above; rc_mont = Montgomery form of ROUND_CONSTANTS (the g_tip5.rc words)
inline void fill_tip5_mx(Tip5MxConsts& t, const u64* rc_mont) {
    constexpr u32 col[16] = {61402, 1108, 28750, 33823, 7454, 43244, 53865, 12034, 56951, 27521, 41351, 40901, 12021, 59689, 26798, 17845};
    const u64 B = 0x4330000000000000ULL;
    const u64 Bp = B % gl::P;
    const u64 K = (u64)(((unsigned __int128)Bp * ((1ull << 32) + 1)) % gl::P);
    for (int i = 0; i < 16; ++i) t.a[i] = (double)col[i];
    for (int round = 0; round < 5; ++round)
        for (int q = 0; q < 4; ++q)
            for (int v = 0; v < 4; ++v) {
                const u64 rc = rc_mont[round * 16 + 4 * v + q];
                const u64 x = rc >= K ? rc - K : rc + (gl::P - K);
                t.c[round][q][v] = 4503599627370496.0 + (double)(u32)x;
                t.c[round][q][4 + v] = 4503599627370496.0 + (double)(u32)(x >> 32);
            }
    // words 12..15 = ONE = 0x00000000ffffffff (Montgomery form of 1): low half 2^32 - 1, high half 0
    for (int q = 0; q < 4; ++q)
        for (int v = 0; v < 4; ++v) {
            const int r = 4 * v + q;
            u64 m = 0;
            for (int c = 12; c < 16; ++c) m += col[(r - c) & 15];
            t.cf[q][v] = t.c[0][q][v] + (double)(m * 0xffffffffULL);  // < 2^50: exact, and the half-sum bound is that of the full product
            t.cf[q][4 + v] = t.c[0][q][4 + v];
        }
}

struct Tip5MxLds {
    Tip5MxConsts t;
    unsigned char lut[256];
};

__device__ __forceinline__ void stage_mx(Tip5MxLds* l) {
    const double* src = reinterpret_cast<const double*>(&g_tip5_mx);
    double* dst = reinterpret_cast<double*>(&l->t);
    for (int i = threadIdx.x; i < (int)(sizeof(Tip5MxConsts) / 8); i += blockDim.x) dst[i] = src[i];
    stage_lut(l->lut);  // ends in __syncthreads()
}

// per-lane A operands: K-block i, lane (r = l & 15, k = l >> 4) holds M[(r - 4 i - k) mod 16]
struct MxA {
    double v[4];
    __device__ __forceinline__ double operator[](int i) const { return v[i]; }
};
__device__ __forceinline__ void mx_a_operands(const Tip5MxLds* l, MxA& aa) {
    double (&a)[4] = aa.v;
    const int lane = threadIdx.x & 63, r = lane & 15, k = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = l->t.a[(r - 4 * i - k) & 15];
}

#endif  // TF_TIP5_I8

// four accumulator registers of the lo-half and hi-half products -> the four state words of this lane.  Word 0 (register 0: the
// next round's split_and_lookup input, whose bytes must be those of the canonical word) is always made canonical; words 1..3
// (x^7 inputs) only when CANON: in between any 64-bit representative serves -- mont_mul3 accepts them (a Montgomery product of two
// arbitrary 64-bit words is a correct, possibly non-canonical, 64-bit representative: montyred's subtrahend is < p, so at most one
// "+ p" is ever needed), and the MDS is linear in the integer value of a word, so a representative that is p too large changes a
// sum by a multiple of p.  The last round of a permutation runs with CANON: everything that leaves the registers is canonical.
template <bool CANON>
__device__ __forceinline__ void mx_fold4_tail(u32 (&tl)[4], u32 (&th)[4], const u32 (&h0)[4], u64* out);
#if !TF_TIP5_I8
template <bool CANON>
__device__ __forceinline__ void mx_fold4(const d4 dlo, const d4 dhi, u64* out) {
    u32 tl[4], th[4], h0[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const u64 rawlo = (u64)__double_as_longlong(dlo[v]), rawhi = (u64)__double_as_longlong(dhi[v]);
        const u64 t1 = (u64)(u32)(rawhi >> 32) * 0xffffffffu + rawlo;  // < 2^63.1: rawlo < 2^62.1, (rawhi >> 32) < 2^30.1
        tl[v] = (u32)t1;
        th[v] = (u32)(t1 >> 32);
        h0[v] = (u32)rawhi;
    }
    mx_fold4_tail<CANON>(tl, th, h0, out);
}
#endif
// value = (th : tl) + h0 * 2^32, th : tl < 2^63.1 + 2^47: add h0 to the high word (carry k: the value is t + k 2^64 = t + k (2^32 - 1)),
// fold the carry; word 0 canonical always, words 1..3 when CANON
template <bool CANON>
__device__ __forceinline__ void mx_fold4_tail(u32 (&tl)[4], u32 (&th)[4], const u32 (&h0)[4], u64* out) {
    u32 rl[4], rh[4];
    u64 ka, kb, kc, kd, ea, na, nb, nc, nd;
    if constexpr (CANON) {
        u64 eb, ec, ed;
#define TF_MX_STEP(A, B, C, D) A "\n\t" B "\n\t" C "\n\t" D "\n\t"
        asm(TF_MX_STEP("v_add_co_u32_e64 %[tha], %[ka], %[tha], %[h0a]", "v_add_co_u32_e64 %[thb], %[kb], %[thb], %[h0b]",  // th += h0, carry k: the value is t + k 2^64
                       "v_add_co_u32_e64 %[thc], %[kc], %[thc], %[h0c]", "v_add_co_u32_e64 %[thd], %[kd], %[thd], %[h0d]")
            TF_MX_STEP("v_cmp_ne_u32_e64 %[na], 0, %[tla]", "v_cmp_ne_u32_e64 %[nb], 0, %[tlb]", "v_cmp_ne_u32_e64 %[nc], 0, %[tlc]",
                       "v_cmp_ne_u32_e64 %[nd], 0, %[tld]")
            TF_MX_STEP("v_cmp_eq_u32_e64 %[ea], -1, %[tha]", "v_cmp_eq_u32_e64 %[eb], -1, %[thb]", "v_cmp_eq_u32_e64 %[ec], -1, %[thc]",
                       "v_cmp_eq_u32_e64 %[ed], -1, %[thd]")
            TF_MX_STEP("s_and_b64 %[ea], %[ea], %[na]", "s_and_b64 %[eb], %[eb], %[nb]", "s_and_b64 %[ec], %[ec], %[nc]",      // t >= p
                       "s_and_b64 %[ed], %[ed], %[nd]")
            TF_MX_STEP("s_or_b64 %[ea], %[ea], %[ka]", "s_or_b64 %[eb], %[eb], %[kb]", "s_or_b64 %[ec], %[ec], %[kc]",           // ... or carry: add 2^32 - 1
                       "s_or_b64 %[ed], %[ed], %[kd]")
            TF_MX_STEP("v_subbrev_co_u32_e64 %[rla], %[na], 0, %[tla], %[ea]", "v_subbrev_co_u32_e64 %[rlb], %[nb], 0, %[tlb], %[eb]",  // lo -= cond, borrow n
                       "v_subbrev_co_u32_e64 %[rlc], %[nc], 0, %[tlc], %[ec]", "v_subbrev_co_u32_e64 %[rld], %[nd], 0, %[tld], %[ed]")
            TF_MX_STEP("s_andn2_b64 %[ea], %[ea], %[na]", "s_andn2_b64 %[eb], %[eb], %[nb]", "s_andn2_b64 %[ec], %[ec], %[nc]",
                       "s_andn2_b64 %[ed], %[ed], %[nd]")
            "v_addc_co_u32_e64 %[rha], %[na], 0, %[tha], %[ea]\n\t"                                                                  // hi += cond & ~borrow
            "v_addc_co_u32_e64 %[rhb], %[nb], 0, %[thb], %[eb]\n\t"
            "v_addc_co_u32_e64 %[rhc], %[nc], 0, %[thc], %[ec]\n\t"
            "v_addc_co_u32_e64 %[rhd], %[nd], 0, %[thd], %[ed]"
            : [tha] "+v"(th[0]), [thb] "+v"(th[1]), [thc] "+v"(th[2]), [thd] "+v"(th[3]), [rla] "=&v"(rl[0]), [rlb] "=&v"(rl[1]),
              [rlc] "=&v"(rl[2]), [rld] "=&v"(rl[3]), [rha] "=&v"(rh[0]), [rhb] "=&v"(rh[1]), [rhc] "=&v"(rh[2]), [rhd] "=&v"(rh[3]),
              [ka] "=&s"(ka), [kb] "=&s"(kb), [kc] "=&s"(kc), [kd] "=&s"(kd), [ea] "=&s"(ea), [eb] "=&s"(eb), [ec] "=&s"(ec), [ed] "=&s"(ed),
              [na] "=&s"(na), [nb] "=&s"(nb), [nc] "=&s"(nc), [nd] "=&s"(nd)
            : [tla] "v"(tl[0]), [tlb] "v"(tl[1]), [tlc] "v"(tl[2]), [tld] "v"(tl[3]), [h0a] "v"(h0[0]), [h0b] "v"(h0[1]), [h0c] "v"(h0[2]),
              [h0d] "v"(h0[3])
            : "scc");
#undef TF_MX_STEP
    } else {
        // chain a as above; chains b..d only fold the carry (t + k 2^64 = t + k (2^32 - 1), which cannot carry again: t < 2^63.1 then)
        asm("v_add_co_u32_e64 %[tha], %[ka], %[tha], %[h0a]\n\t"
            "v_add_co_u32_e64 %[thb], %[kb], %[thb], %[h0b]\n\t"
            "v_add_co_u32_e64 %[thc], %[kc], %[thc], %[h0c]\n\t"
            "v_add_co_u32_e64 %[thd], %[kd], %[thd], %[h0d]\n\t"
            "v_cmp_ne_u32_e64 %[na], 0, %[tla]\n\t"
            "v_cmp_eq_u32_e64 %[ea], -1, %[tha]\n\t"
            "v_subbrev_co_u32_e64 %[rlb], %[nb], 0, %[tlb], %[kb]\n\t"   // lo -= k, borrow n
            "v_subbrev_co_u32_e64 %[rlc], %[nc], 0, %[tlc], %[kc]\n\t"
            "v_subbrev_co_u32_e64 %[rld], %[nd], 0, %[tld], %[kd]\n\t"
            "s_and_b64 %[ea], %[ea], %[na]\n\t"
            "s_or_b64 %[ea], %[ea], %[ka]\n\t"
            "s_andn2_b64 %[kb], %[kb], %[nb]\n\t"
            "s_andn2_b64 %[kc], %[kc], %[nc]\n\t"
            "s_andn2_b64 %[kd], %[kd], %[nd]\n\t"
            "v_subbrev_co_u32_e64 %[rla], %[na], 0, %[tla], %[ea]\n\t"
            "v_addc_co_u32_e64 %[rhb], %[nb], 0, %[thb], %[kb]\n\t"       // hi += k & ~borrow
            "v_addc_co_u32_e64 %[rhc], %[nc], 0, %[thc], %[kc]\n\t"
            "v_addc_co_u32_e64 %[rhd], %[nd], 0, %[thd], %[kd]\n\t"
            "s_andn2_b64 %[ea], %[ea], %[na]\n\t"
            "v_addc_co_u32_e64 %[rha], %[na], 0, %[tha], %[ea]"
            : [tha] "+v"(th[0]), [thb] "+v"(th[1]), [thc] "+v"(th[2]), [thd] "+v"(th[3]), [rla] "=&v"(rl[0]), [rlb] "=&v"(rl[1]),
              [rlc] "=&v"(rl[2]), [rld] "=&v"(rl[3]), [rha] "=&v"(rh[0]), [rhb] "=&v"(rh[1]), [rhc] "=&v"(rh[2]), [rhd] "=&v"(rh[3]),
              [ka] "=&s"(ka), [kb] "=&s"(kb), [kc] "=&s"(kc), [kd] "=&s"(kd), [ea] "=&s"(ea), [na] "=&s"(na), [nb] "=&s"(nb), [nc] "=&s"(nc),
              [nd] "=&s"(nd)
            : [tla] "v"(tl[0]), [tlb] "v"(tl[1]), [tlc] "v"(tl[2]), [tld] "v"(tl[3]), [h0a] "v"(h0[0]), [h0b] "v"(h0[1]), [h0c] "v"(h0[2]),
              [h0d] "v"(h0[3])
            : "scc");
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) out[v] = ((u64)rh[v] << 32) | rl[v];
}

// One round on NS permutations per lane quartet: s[4 n + i] = word 4 i + q of the permutation in column j of column block n.
// the canonical recombination of words 0 and 1 only: the last round of hash_10 / hash_pair, whose digest is state words 0..4
// (registers 0 of the four quarters and register 1 of quarter 0)
__device__ __forceinline__ void mx_fold2_tail(u32 (&tl)[2], u32 (&th)[2], const u32 (&h0)[2], u64* out);
#if !TF_TIP5_I8
__device__ __forceinline__ void mx_fold2_canon(const d4 dlo, const d4 dhi, u64* out) {
    u32 tl[2], th[2], h0[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const u64 rawlo = (u64)__double_as_longlong(dlo[v]), rawhi = (u64)__double_as_longlong(dhi[v]);
        const u64 t1 = (u64)(u32)(rawhi >> 32) * 0xffffffffu + rawlo;
        tl[v] = (u32)t1;
        th[v] = (u32)(t1 >> 32);
        h0[v] = (u32)rawhi;
    }
    mx_fold2_tail(tl, th, h0, out);
}
#endif
__device__ __forceinline__ void mx_fold2_tail(u32 (&tl)[2], u32 (&th)[2], const u32 (&h0)[2], u64* out) {
    u32 rl[2], rh[2];
    u64 ka, kb, ea, eb, na, nb;
    asm("v_add_co_u32_e64 %[tha], %[ka], %[tha], %[h0a]\n\t"
        "v_add_co_u32_e64 %[thb], %[kb], %[thb], %[h0b]\n\t"
        "v_cmp_ne_u32_e64 %[na], 0, %[tla]\n\t"
        "v_cmp_ne_u32_e64 %[nb], 0, %[tlb]\n\t"
        "v_cmp_eq_u32_e64 %[ea], -1, %[tha]\n\t"
        "v_cmp_eq_u32_e64 %[eb], -1, %[thb]\n\t"
        "s_and_b64 %[ea], %[ea], %[na]\n\t"
        "s_and_b64 %[eb], %[eb], %[nb]\n\t"
        "s_or_b64 %[ea], %[ea], %[ka]\n\t"
        "s_or_b64 %[eb], %[eb], %[kb]\n\t"
        "v_subbrev_co_u32_e64 %[rla], %[na], 0, %[tla], %[ea]\n\t"
        "v_subbrev_co_u32_e64 %[rlb], %[nb], 0, %[tlb], %[eb]\n\t"
        "s_andn2_b64 %[ea], %[ea], %[na]\n\t"
        "s_andn2_b64 %[eb], %[eb], %[nb]\n\t"
        "v_addc_co_u32_e64 %[rha], %[na], 0, %[tha], %[ea]\n\t"
        "v_addc_co_u32_e64 %[rhb], %[nb], 0, %[thb], %[eb]"
        : [tha] "+v"(th[0]), [thb] "+v"(th[1]), [rla] "=&v"(rl[0]), [rlb] "=&v"(rl[1]), [rha] "=&v"(rh[0]), [rhb] "=&v"(rh[1]), [ka] "=&s"(ka),
          [kb] "=&s"(kb), [ea] "=&s"(ea), [eb] "=&s"(eb), [na] "=&s"(na), [nb] "=&s"(nb)
        : [tla] "v"(tl[0]), [tlb] "v"(tl[1]), [h0a] "v"(h0[0]), [h0b] "v"(h0[1])
        : "scc");
    out[0] = ((u64)rh[0] << 32) | rl[0];
    out[1] = ((u64)rh[1] << 32) | rl[1];
}

// FIXED0 = 1: round 0 of a fixed-length hash, where state words 10..15 are the constant 1.  Register 3 of every lane (words 12..15)
// then needs no x^7 (1^7 = 1) and no MFMA: its MDS contribution is a constant that rides in the accumulator start (Tip5MxConsts::cf).
// FIXED0 = 2: round 0 of the FIRST permutation of a variable-length sponge, whose capacity words 10..15 are still 0 (0^7 = 0, no
// contribution at all).  The same words as the general round on such a state -- one Montgomery chain of three and 2 of the 8 MFMA
// saved in one round of five.
template <int NS, bool LAST, int FIXED0 = 0, bool DIGEST = false>
__device__ __forceinline__ void tip5_round_mx(u64 (&s)[4 * NS], int round, const Tip5MxLds* l, const MxA& a, int q) {
#pragma unroll
    for (int n = 0; n < NS; ++n) {  // split_and_lookup (mod.rs:197-207): words 0..3 = register 0 of the four quarters
        const u32 lo = lookup4((u32)s[4 * n], l->lut), hi = lookup4((u32)(s[4 * n] >> 32), l->lut);
        s[4 * n] = ((u64)hi << 32) | lo;
    }
#pragma unroll
    for (int n = 0; n < NS; ++n) {  // x^7 = x * (x^2 * x^4) on registers 1..3
        if constexpr (FIXED0) {
            u64 &x0 = s[4 * n + 1], &x1 = s[4 * n + 2], sq0, sq1, qu0, qu1, t0, t1;
            gl::mont_mul2(x0, x0, x1, x1, sq0, sq1);
            gl::mont_mul2(sq0, sq0, sq1, sq1, qu0, qu1);
            gl::mont_mul2(sq0, qu0, sq1, qu1, t0, t1);
            gl::mont_mul2(x0, t0, x1, t1, x0, x1);
        } else {
            u64 x[3] = {s[4 * n + 1], s[4 * n + 2], s[4 * n + 3]}, sq[3], qu[3], t[3];
            gl::mont_mul3(x, x, sq);
            gl::mont_mul3(sq, sq, qu);
            gl::mont_mul3(sq, qu, t);
            gl::mont_mul3(x, t, x);
            s[4 * n + 1] = x[0], s[4 * n + 2] = x[1], s[4 * n + 3] = x[2];
        }
    }
#if TF_TIP5_I8
    // ---- the MDS on the i8 matrix pipe (see Tip5MxConsts): ten planes, twelve MFMA
    const int* cp = FIXED0 == 1 ? &l->cf[0][q][0] : (FIXED0 == 2 ? &l->cz[0][q][0] : &l->c[round][0][q][0]);  // [plane][q][t]: 16 ints per plane
    constexpr int NW = FIXED0 ? 3 : 4;  // registers that go into B (FIXED0: register 3 -- words 12..15 -- is a constant, folded into the starts)
    v4i blo[NS], bhi[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            blo[n][i] = i < NW ? (int)((u32)s[4 * n + i] ^ 0x80808080u) : 0;
            bhi[n][i] = i < NW ? (int)((u32)(s[4 * n + i] >> 32) ^ 0x80808080u) : 0;
        }
    v4i d[NS][10];
#pragma unroll
    for (int p = 0; p < 10; ++p) {
        const v4i c = *reinterpret_cast<const v4i*>(cp + p * 16);
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            if (p < 6) {
                d[n][p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.p[p], blo[n], c, 0, 0, 0);
                if (p >= 4) d[n][p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.p[p - 4], bhi[n], d[n][p], 0, 0, 0);
            } else {
                d[n][p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.p[p - 4], bhi[n], c, 0, 0, 0);
            }
        }
    }
    // ---- recombination: (th : tl) = L0 + (2^32 - 1) (hi(L1) + L2), h0 = lo(L1), then the shared tail
    // x * y + z as ONE v_mad_u64_u32.  NOT inline assembly: these are the first readers of the MFMA results, and the compiler only counts
    // the wait states between a matrix instruction and a reader it can see (an asm block reading d[] too early returned garbage in the
    // two-permutations-per-quartet build of tools/microbench_mds.hip).  The multipliers are made opaque instead (an SGPR the optimiser
    // cannot see through), so that the products by 2^16 / 2^24 are not strength-reduced into 64-bit shift-and-add sequences.
    u32 k16 = 1u << 16, k24 = 1u << 24, kff = 0xffffffffu;
    asm volatile("" : "+s"(k16), "+s"(k24), "+s"(kff));
    const auto mad = [](u32 x, u32 y, u64 z) { return (u64)x * y + z; };
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        constexpr int NT = DIGEST ? 2 : 4;  // (words 8..15 of the final state are not part of a digest)
        u32 tl[NT], th[NT], h0[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const auto Q = [&](int p) { return (u32)d[n][p][t]; };
            u64 L0 = mad(Q(2), k16, (u64)((Q(1) << 8) + Q(0)));
            L0 = mad(Q(3), k24, L0);
            u64 L1 = mad(Q(6), k16, (u64)((Q(5) << 8) + Q(4)));
            L1 = mad(Q(7), k24, L1);
            const u32 hsum = (u32)(L1 >> 32) + ((Q(9) << 8) + Q(8));  // < 2^15 + 2^31
            const u64 u = mad(hsum, kff, L0);                           // < 2^64: hsum (2^32 - 1) < 2^63.1, L0 < 2^47
            tl[t] = (u32)u, th[t] = (u32)(u >> 32), h0[t] = (u32)L1;
        }
        if constexpr (DIGEST) mx_fold2_tail(tl, th, h0, &s[4 * n]);
        else mx_fold4_tail<LAST>(tl, th, h0, &s[4 * n]);
    }
#else
    const d4* cp = reinterpret_cast<const d4*>(FIXED0 == 1 ? &l->t.cf[q][0] : &l->t.c[round][q][0]);
    const d4 c_lo = cp[0], c_hi = cp[1];
    d4 dlo[NS], dhi[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) dlo[n] = c_lo, dhi[n] = c_hi;
#pragma unroll
    for (int i = 0; i < (FIXED0 ? 3 : 4); ++i)
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            dlo[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], (double)(u32)s[4 * n + i], dlo[n], 0, 0, 0);
            dhi[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], (double)(u32)(s[4 * n + i] >> 32), dhi[n], 0, 0, 0);
        }
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        if constexpr (DIGEST) mx_fold2_canon(dlo[n], dhi[n], &s[4 * n]);  // (words 8..15 of the final state are not part of a digest)
        else mx_fold4<LAST>(dlo[n], dhi[n], &s[4 * n]);
    }
#endif
}

template <int NS>
__device__ __forceinline__ void tip5_permutation_mx(u64 (&s)[4 * NS], const Tip5MxLds* l, const MxA& a, int q) {
#pragma unroll 1
    for (int r = 0; r < 4; ++r) tip5_round_mx<NS, false>(s, r, l, a, q);
    tip5_round_mx<NS, true>(s, 4, l, a, q);
}

// the permutation of hash_10 / hash_pair: on entry words 10..15 of every state are 1 (register 3 = 1 in every lane, register 2 = 1 in
// quarters 2 and 3); register 3 need not even be initialised by the caller.  On exit only registers 0 and 1 (state words 0..7, of
// which 0..4 are the digest) are defined.
template <int NS>
__device__ __forceinline__ void tip5_permutation_mx_fixed(u64 (&s)[4 * NS], const Tip5MxLds* l, const MxA& a, int q) {
    tip5_round_mx<NS, false, 1>(s, 0, l, a, q);
#pragma unroll 1
    for (int r = 1; r < 4; ++r) tip5_round_mx<NS, false>(s, r, l, a, q);
    tip5_round_mx<NS, true, 0, true>(s, 4, l, a, q);  // only the digest words are finished
}

// the permutations of a variable-length sponge: FIRST = capacity words still 0 on entry, FINAL = only the digest is read afterwards
template <int NS, bool FIRST, bool FINAL>
__device__ __forceinline__ void tip5_permutation_mx_sponge(u64 (&s)[4 * NS], const Tip5MxLds* l, const MxA& a, int q) {
    tip5_round_mx<NS, false, FIRST ? 2 : 0>(s, 0, l, a, q);
#pragma unroll 1
    for (int r = 1; r < 4; ++r) tip5_round_mx<NS, false>(s, r, l, a, q);
    tip5_round_mx<NS, true, 0, FINAL>(s, 4, l, a, q);
}

// item -> (tree, index in its tree).  shift >= 0: per_tree = 2^shift (every Merkle level); shift < 0: any per_tree (a flat call
// has per_tree = count, so the quotient is 0 -- the division is uniform over the launch and only taken by odd table shapes).
__device__ __forceinline__ void split_item(long long i, long long per_tree, int shift, long long& tree, long long& k) {
    if (shift >= 0) {
        tree = i >> shift;
        k = i - (tree << shift);
    } else if (i < per_tree) {
        tree = 0;
        k = i;
    } else {
        tree = i / per_tree;
        k = i - tree * per_tree;
    }
}

// Work distribution of every matrix-pipe kernel: a workgroup is 4 waves; the waves of the grid walk over groups of 16 NS
// consecutive items with a grid stride (the host caps the grid at a few workgroups per CU, so the tables are staged and the A
// operands fetched once per wave, not once per group); lane (j, q) serves item base + 16 n + j for n < NS.  Lanes of items past
// the end stay in the wave (the MFMA wants all 64 lanes) on a clamped item and skip their stores.
#define TF_MX_PROLOGUE()                                                                               \
    __shared__ __attribute__((aligned(32))) Tip5MxLds lds;                                             \
    stage_mx(&lds);                                                                                    \
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;                                   \
    MxA a;                                                                                             \
    mx_a_operands(&lds, a)
#define TF_MX_GROUPS(COUNT)                                                                            \
    for (long long base = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (16 * NS); base < (COUNT); \
         base += (long long)gridDim.x * (4 * 16 * NS))

// out[i] = hash_10(in[10 i .. 10 i + 10)) = hash_pair(left, right)  (mod.rs:559-586), item i belongs to tree i / per_tree; its input is in + tree * in_ts + 10 * (i % per_tree), its
// digest goes to out + tree * out_ts + 5 * (i % per_tree); if leaf_copy != null the 10 input words are also copied to
// leaf_copy + tree * copy_ts + 10 * (i % per_tree) (Merkle leaf level, merkle_tree.rs:426).
template <int NS>
__global__ void __launch_bounds__(256) tip5_hash_pairs_mx_kernel(const u64* in, u64* out, u64* leaf_copy, long long count,
                                                                 long long per_tree, int shift, long long in_ts, long long out_ts,
                                                                 long long copy_ts) {
    // The input of a wave's NEXT group is fetched before the current one is permuted, and the first group's before the tables are
    // staged: a launch of one group per wave (a tree level of 2^14 .. 2^16 pairs) waits for memory once, not twice.
    __shared__ __attribute__((aligned(32))) Tip5MxLds lds;
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const long long first = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (16 * NS), step = (long long)gridDim.x * (4 * 16 * NS);
    u64 pre[3 * NS];
    long long tree[NS], k[NS];
    bool live[NS];
    auto fetch = [&](long long base) {
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const long long item = base + 16 * n + j;
            live[n] = item < count;
            split_item(live[n] ? item : count - 1, per_tree, shift, tree[n], k[n]);
            const u64* p = in + tree[n] * in_ts + 10 * k[n];
            pre[3 * n] = p[q];
            pre[3 * n + 1] = p[4 + q];
            pre[3 * n + 2] = q < 2 ? p[8 + q] : gl::ONE;  // words 10..15 = 1: Tip5::new(Domain::FixedLength), mod.rs:511-526
        }
    };
    if (first < count) fetch(first);
    stage_mx(&lds);
    MxA a;
    mx_a_operands(&lds, a);
    for (long long base = first; base < count; base += step) {
        u64 s[4 * NS];
        long long otree[NS], ok[NS];
        bool olive[NS];
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            s[4 * n] = pre[3 * n];
            s[4 * n + 1] = pre[3 * n + 1];
            s[4 * n + 2] = pre[3 * n + 2];
            s[4 * n + 3] = gl::ONE;
            otree[n] = tree[n], ok[n] = k[n], olive[n] = live[n];
            if (leaf_copy && live[n]) {  // Merkle leaf level, merkle_tree.rs:426
                u64* c = leaf_copy + tree[n] * copy_ts + 10 * k[n];
                c[q] = s[4 * n];
                c[4 + q] = s[4 * n + 1];
                if (q < 2) c[8 + q] = s[4 * n + 2];
            }
        }
        if (base + step < count) fetch(base + step);
        tip5_permutation_mx_fixed<NS>(s, &lds, a, q);
#pragma unroll
        for (int n = 0; n < NS; ++n)
            if (olive[n]) {
                u64* o = out + otree[n] * out_ts + 5 * ok[n];
                o[q] = s[4 * n];
                if (q == 0) o[4] = s[4 * n + 1];
            }
    }
}

// The sponge of hash_varlen (mod.rs:617-623, overwrite-mode absorb :684-691, padding sponge.rs:41-55) over a row whose word w is
// WORD(w); shared by the row-major and the column-major kernels below.  Rate word w = 4 i + q < 10 lives in register i of quarter q.
// (Fetching the NEXT chunk before the current permutation was measured as a loss here: 16-25 more registers, one wave per SIMD fewer,
// -1.3 % on 2^21 rows of 128 columns -- profiles/r05_mx_prefetch_ab.txt.  The launches are large; other waves hide the latency.)
#define TF_MX_SPONGE(ROW_LEN, WORD)                                                                                     \
    u64 s[4 * NS];                                                                                                      \
    _Pragma("unroll") for (int t = 0; t < 4 * NS; ++t) s[t] = 0; /* Domain::VariableLength */                           \
    const long long full = (ROW_LEN) / 10;                                                                              \
    for (long long c = 0; c < full; ++c) {                                                                              \
        _Pragma("unroll") for (int n = 0; n < NS; ++n) {                                                                \
            s[4 * n] = WORD(n, c * 10 + q);                                                                             \
            s[4 * n + 1] = WORD(n, c * 10 + 4 + q);                                                                     \
            if (q < 2) s[4 * n + 2] = WORD(n, c * 10 + 8 + q);                                                          \
        }                                                                                                               \
        if (c == 0) tip5_permutation_mx_sponge<NS, true, false>(s, &lds, a, q); /* capacity still zero */               \
        else tip5_permutation_mx<NS>(s, &lds, a, q);                                                                    \
    }                                                                                                                   \
    const int rem = (int)((ROW_LEN) - full * 10);                                                                       \
    _Pragma("unroll") for (int n = 0; n < NS; ++n) {                                                                    \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                                 \
            const int w = 4 * i + q;                                                                                    \
            if (w < 10) s[4 * n + i] = (w < rem) ? WORD(n, full * 10 + w) : ((w == rem) ? gl::ONE : 0);                 \
        }                                                                                                               \
    }                                                                                                                   \
    if (full == 0) tip5_permutation_mx_sponge<NS, true, true>(s, &lds, a, q);                                           \
    else tip5_permutation_mx_sponge<NS, false, true>(s, &lds, a, q)

// hash_varlen of n_rows rows of row_len words each.  Row i belongs to tree i / per_tree; its digest goes to out + tree * out_ts +
// 5 * (i % per_tree)  (out_ts = 0 and per_tree = n_rows: a flat digest array; out_ts = 10 n, out = nodes + 5 n: straight into the
// leaf level of a tree).
template <int NS>
__global__ void __launch_bounds__(256) tip5_hash_varlen_rows_mx_kernel(const u64* rows, long long row_len, long long n_rows, u64* out,
                                                                       long long per_tree, int shift, long long out_ts) {
    TF_MX_PROLOGUE();
    TF_MX_GROUPS(n_rows) {
    const u64* p[NS];
    long long item[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        item[n] = base + 16 * n + j;
        p[n] = rows + (item[n] < n_rows ? item[n] : n_rows - 1) * row_len;
    }
#define TF_MX_ROW_WORD(n, w) p[n][w]
    TF_MX_SPONGE(row_len, TF_MX_ROW_WORD);
#undef TF_MX_ROW_WORD
#pragma unroll
    for (int n = 0; n < NS; ++n)
        if (item[n] < n_rows) {
            long long tree, k;
            split_item(item[n], per_tree, shift, tree, k);
            u64* o = out + tree * out_ts + 5 * k;
            o[q] = s[4 * n];
            if (q == 0) o[4] = s[4 * n + 1];
        }
    }
}

// hash_varlen of every row of `batch` column-major tables (layout: table_word above); row i of table t -> out + t * out_ts + 5 i.
template <int NS>
__global__ void __launch_bounds__(256) tip5_hash_table_rows_mx_kernel(const u64* table, long long n_rows, int shift, long long n_cols,
                                                                      int width, long long col_stride, long long table_stride,
                                                                      long long total, u64* out, long long out_ts) {
    TF_MX_PROLOGUE();
    const long long row_len = n_cols * width;
    TF_MX_GROUPS(total) {
    const u64* tb[NS];
    long long tree[NS], row[NS];
    bool live[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const long long id = base + 16 * n + j;
        live[n] = id < total;
        split_item(live[n] ? id : total - 1, n_rows, shift, tree[n], row[n]);
        tb[n] = table + tree[n] * table_stride;
    }
#define TF_MX_TABLE_WORD(n, w) table_word(tb[n], row[n], w, width, col_stride)
    TF_MX_SPONGE(row_len, TF_MX_TABLE_WORD);
#undef TF_MX_TABLE_WORD
#pragma unroll
    for (int n = 0; n < NS; ++n)
        if (live[n]) {
            u64* o = out + tree[n] * out_ts + 5 * row[n];
            o[q] = s[4 * n];
            if (q == 0) o[4] = s[4 * n + 1];
        }
    }
}

// states: count x 16 words, permuted in place (Tip5::permutation, mod.rs:529-533); with trace != null also Tip5::trace
// (mod.rs:538-548): trace[i][0] = the state before the permutation, trace[i][1 + r] = the state after round r.
template <int NS>
__global__ void __launch_bounds__(256) tip5_permute_mx_kernel(u64* states, u64* trace, long long count) {
    TF_MX_PROLOGUE();
    TF_MX_GROUPS(count) {
    u64 s[4 * NS];
    long long item[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        item[n] = base + 16 * n + j;
        const u64* p = states + (item[n] < count ? item[n] : count - 1) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) s[4 * n + i] = p[4 * i + q];
    }
    if (trace) {
#pragma unroll 1
        for (int r = 0; r < 6; ++r) {
            if (r) tip5_round_mx<NS, true>(s, r - 1, &lds, a, q);  // every traced state is canonical
#pragma unroll
            for (int n = 0; n < NS; ++n)
                if (item[n] < count) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) trace[item[n] * 96 + 16 * r + 4 * i + q] = s[4 * n + i];
                }
        }
    } else {
        tip5_permutation_mx<NS>(s, &lds, a, q);
    }
#pragma unroll
    for (int n = 0; n < NS; ++n)
        if (item[n] < count) {
#pragma unroll
            for (int i = 0; i < 4; ++i) states[item[n] * 16 + 4 * i + q] = s[4 * n + i];
        }
    }
}

// ---- cooperative form: the 16 lanes of a DPP row hold the 16 state words of ONE permutation ------------------
// The lane-per-permutation kernels above are throughput-optimal but one permutation is a dependent chain of ~8 200
// instructions (~19 us), which is what every level of a Merkle tree with fewer nodes than the GPU has lanes costs.
// Here lane j of a row owns state[j]: the S-box is one lookup or one x^7 per lane, and because the MDS matrix is
// circulant, out[r] = sum_k M[k] * state[(r - k) mod 16] is 16 row rotations (v_mov_b32 row_ror:k) each followed by one
// v_mad_u64_u32 with the same constant M[k] in every lane.  ~1 600 instructions per permutation (the lookup lanes and
// the x^7 lanes take turns): 3x the total work of the lane-per-permutation form, 1/5 of its latency -- measured 4.6 us
// per tree level instead of 19 us.  Used for launches of at most kCoopMaxCount permutation chains (tf_tip5.hip).
// the circulant as two interleaved pairs of accumulators (even and odd rotations): four independent v_mad_u64_u32 chains in
// flight instead of two -- one permutation is a latency chain, and a v_mad_u64_u32 that waits for its own previous result stalls
template <int K>
__device__ __forceinline__ void mds_coop_terms(u32 lo, u32 hi, u64 (&alo)[2], u64 (&ahi)[2]) {
    if constexpr (K < 16) {
        const u32 rl = (u32)__builtin_amdgcn_mov_dpp((int)lo, 0x120 + K, 0xf, 0xf, true);  // row_ror:K: lane r <- lane (r - K) mod 16
        const u32 rh = (u32)__builtin_amdgcn_mov_dpp((int)hi, 0x120 + K, 0xf, 0xf, true);
        alo[K & 1] += (u64)mds_entry(K) * rl;
        ahi[K & 1] += (u64)mds_entry(K) * rh;
        mds_coop_terms<K + 1>(lo, hi, alo, ahi);
    }
}

// the five round constants of state word j: fetched once per kernel, before anything waits (the loads fly with the kernel's input)
__device__ __forceinline__ void coop_round_constants(int j, u64 (&rcs)[5]) {
#pragma unroll
    for (int r = 0; r < 5; ++r) rcs[r] = g_tip5.rc[r * 16 + j];
}

// s = state[j] of the permutation shared by the 16 lanes of this row; all 16 lanes must be active.
// Latency shape (one permutation is what a tree level near the root, or a chunk of one long sponge, waits for):
//   * no divergence in the S-box layer: every lane runs BOTH the split_and_lookup of its word and its x^7 and keeps the one its
//     position calls for -- a wave issues both paths either way (every row has lanes of both kinds), but as one instruction stream
//     the eight LDS lookups and their packing fill the wait states of the multiply chain instead of following it;
//   * x^7 = (x^2 * x) * (x^2)^2: three products deep instead of four, the two in the middle as one hand-scheduled pair;
//   * the five rounds unrolled: the round constant is a register, not a select chain.
__device__ __forceinline__ void tip5_permutation_coop(u64& s, int j, const unsigned char* lut, const u64 (&rcs)[5]) {
#pragma unroll
    for (int round = 0; round < 5; ++round) {
        const u32 ll = lookup4((u32)s, lut), lh = lookup4((u32)(s >> 32), lut);  // split_and_lookup (mod.rs:197-207), kept by lanes 0..3
        const u64 sq = gl::mont_mul(s, s);
        u64 cu, qu;
        gl::mont_mul2(sq, s, sq, sq, cu, qu);
        const u64 x7 = gl::mont_mul(cu, qu);
        s = j < 4 ? (((u64)lh << 32) | ll) : x7;
        const u32 lo = (u32)s, hi = (u32)(s >> 32);
        u64 alo[2] = {(u64)mds_entry(0) * lo, 0}, ahi[2] = {(u64)mds_entry(0) * hi, 0};
        mds_coop_terms<1>(lo, hi, alo, ahi);
        asm("" : "+v"(alo[1]), "+v"(ahi[1]));  // (keeps the compiler from folding the odd chains back into the even ones)
        const u64 slo = alo[0] + alo[1], shi = ahi[0] + ahi[1];  // < 2^52 each
        // same single-fold reduction as tip5_round
        const u64 rc = rcs[round];
        unsigned c0, c1, c2, c3, c4;
        const u32 w1 = __builtin_addc((u32)(slo >> 32), (u32)shi, 0u, &c0);
        const u32 w2 = __builtin_addc((u32)(shi >> 32), 0u, c0, &c1);
        const u32 t0 = __builtin_addc((u32)slo, (u32)rc, 0u, &c2);
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

// ---- the same permutation on TWO rows (32 lanes) -- round 6, for launches that leave half the chip's rows idle anyway ----------------
// One permutation is an issue chain of ~920 cycles a round, and the circulant is half of it (16 row rotations + 32 v_mad_u64_u32).  Both
// rows of a pair hold the state and run the S-box layer side by side (one instruction stream: no extra time); row h takes the eight
// rotation terms k = 8 h .. 8 h + 7 -- the state rotated by 8 first in row 1, the matrix entries M[k + 8 h] as per-lane registers -- and
// gfx950's v_permlane16_swap_b32 joins the two partial sums (four swaps, two 64-bit additions): 16 rotations + 16 products per lane
// instead of 32 + 32.  Measured (tools/microbench_coop2.hip, profiles/r06_microbench_coop2.txt: a chain of 20 000 dependent permutations
// in one wave, word for word the 16-lane form and the lane-per-permutation round): 2.01 -> 1.68 us per permutation.  It halves the
// permutations per wave, so it is used where the rows would otherwise idle: launches of at most 8 chains per compute unit
// (tf_tip5.hip: coop_two_rows) and the levels of a subtree that have at most half as many pairs as the workgroup has rows.
struct CoopHalfMatrix {
    u32 m[8];  // M[k + 8 half], k = 0..7
};
__device__ __forceinline__ void coop_half_matrix(int half, CoopHalfMatrix& hm) {
#pragma unroll
    for (int k = 0; k < 8; ++k) hm.m[k] = half ? mds_entry(k + 8) : mds_entry(k);
}
template <int K>
__device__ __forceinline__ void mds_coop_half_terms(u32 lo, u32 hi, const CoopHalfMatrix& hm, u64 (&alo)[2], u64 (&ahi)[2]) {
    if constexpr (K < 8) {
        const u32 rl = (u32)__builtin_amdgcn_mov_dpp((int)lo, 0x120 + K, 0xf, 0xf, true);  // row_ror:K
        const u32 rh = (u32)__builtin_amdgcn_mov_dpp((int)hi, 0x120 + K, 0xf, 0xf, true);
        alo[K & 1] += (u64)hm.m[K] * rl;
        ahi[K & 1] += (u64)hm.m[K] * rh;
        mds_coop_half_terms<K + 1>(lo, hi, hm, alo, ahi);
    }
}
// x of this row + x of the partner row (lane ^ 16): after  a = b = x;  v_permlane16_swap a, b  the odd row of a and the even row of b have
// changed places -- a = (x_even, x_even), b = (x_odd, x_odd) in both rows of a pair -- so a + b is the sum, in both rows
__device__ __forceinline__ u64 coop_add_partner_row(u64 x) {
    const u32 xl = (u32)x, xh = (u32)(x >> 32);
    const auto l = __builtin_amdgcn_permlane16_swap(xl, xl, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(xh, xh, false, false);
    return (((u64)h[0] << 32) | l[0]) + (((u64)h[1] << 32) | l[1]);
}
// s = state[j] of the permutation shared by the 32 lanes of this row pair (BOTH rows hold it, both must be active); half = (lane >> 4) & 1
__device__ __forceinline__ void tip5_permutation_coop2(u64& s, int j, int half, const unsigned char* lut, const u64 (&rcs)[5], const CoopHalfMatrix& hm) {
#pragma unroll
    for (int round = 0; round < 5; ++round) {
        const u32 ll = lookup4((u32)s, lut), lh = lookup4((u32)(s >> 32), lut);
        const u64 sq = gl::mont_mul(s, s);
        u64 cu, qu;
        gl::mont_mul2(sq, s, sq, sq, cu, qu);
        const u64 x7 = gl::mont_mul(cu, qu);
        s = j < 4 ? (((u64)lh << 32) | ll) : x7;
        u32 lo = (u32)s, hi = (u32)(s >> 32);
        // row 1 works on the state rotated by 8: M[8 + k] state[(r - 8 - k) mod 16]
        const u32 lo8 = (u32)__builtin_amdgcn_mov_dpp((int)lo, 0x128, 0xf, 0xf, true), hi8 = (u32)__builtin_amdgcn_mov_dpp((int)hi, 0x128, 0xf, 0xf, true);
        lo = half ? lo8 : lo;
        hi = half ? hi8 : hi;
        u64 alo[2] = {(u64)hm.m[0] * lo, 0}, ahi[2] = {(u64)hm.m[0] * hi, 0};
        mds_coop_half_terms<1>(lo, hi, hm, alo, ahi);
        asm("" : "+v"(alo[1]), "+v"(ahi[1]));
        const u64 slo = coop_add_partner_row(alo[0] + alo[1]), shi = coop_add_partner_row(ahi[0] + ahi[1]);  // < 2^52 each
        const u64 rc = rcs[round];
        unsigned c0, c1, c2, c3, c4;
        const u32 w1 = __builtin_addc((u32)(slo >> 32), (u32)shi, 0u, &c0);
        const u32 w2 = __builtin_addc((u32)(shi >> 32), 0u, c0, &c1);
        const u32 t0 = __builtin_addc((u32)slo, (u32)rc, 0u, &c2);
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
// ROWS = 1: 16 lanes per chain (tip5_permutation_coop); ROWS = 2: 32 lanes (tip5_permutation_coop2).  A workgroup of 256 threads holds
// 16 / ROWS chains; chain index and half of a thread:
template <int ROWS>
struct CoopGeom {
    static constexpr int kShift = ROWS == 2 ? 5 : 4, kPerBlock = 256 >> kShift;
    __device__ static __forceinline__ long long item() { return (long long)blockIdx.x * kPerBlock + (threadIdx.x >> kShift); }
    __device__ static __forceinline__ int half() { return ROWS == 2 ? (int)((threadIdx.x >> 4) & 1) : 0; }
};
template <int ROWS>
__device__ __forceinline__ void tip5_permutation_coop_n(u64& s, int j, int half, const unsigned char* lut, const u64 (&rcs)[5], const CoopHalfMatrix& hm) {
    if constexpr (ROWS == 2) tip5_permutation_coop2(s, j, half, lut, rcs, hm);
    else tip5_permutation_coop(s, j, lut, rcs);
}

// hash_pair per 16-lane row: item i = blockIdx.x * 16 + threadIdx.x / 16.  Same item i belongs to tree i / per_tree; its input is in + tree * in_ts + 10 * (i % per_tree), its
// digest goes to out + tree * out_ts + 5 * (i % per_tree); if leaf_copy != null the 10 input words are also copied to
// leaf_copy + tree * copy_ts + 10 * (i % per_tree) (Merkle leaf level, merkle_tree.rs:426).
template <int ROWS>
__global__ void __launch_bounds__(256) tip5_hash_pairs_coop_kernel(const u64* in, u64* out, long long count, long long per_tree,
                                                                   int shift, long long in_ts, long long out_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    const int j = threadIdx.x & 15, half = CoopGeom<ROWS>::half();
    const long long i = CoopGeom<ROWS>::item();
    const bool live = i < count;  // whole rows (row pairs) are live or not
    long long tree, k;
    split_item(live ? i : count - 1, per_tree, shift, tree, k);
    u64 s = j < 10 ? in[tree * in_ts + 10 * k + j] : gl::ONE;  // the input and the round constants are in flight while the table is built
    u64 rcs[5];
    coop_round_constants(j, rcs);
    CoopHalfMatrix hm;
    coop_half_matrix(half, hm);
    stage_lut(lut);
    if (!live) return;
    tip5_permutation_coop_n<ROWS>(s, j, half, lut, rcs, hm);
    if (j < 5 && !half) out[tree * out_ts + 5 * k + j] = s;
}

// hash_varlen of row i by the 16 lanes of row-group i (few rows, or one long input: the absorb chain is sequential)
template <int ROWS>
__global__ void __launch_bounds__(256) tip5_hash_varlen_rows_coop_kernel(const u64* rows, long long row_len, long long n_rows,
                                                                         u64* out, long long per_tree, int shift, long long out_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    const int j = threadIdx.x & 15, half = CoopGeom<ROWS>::half();
    const long long i = CoopGeom<ROWS>::item();
    const bool live = i < n_rows;
    const u64* p = rows + (live ? i : n_rows - 1) * row_len;
    u64 s = 0;  // Domain::VariableLength
    // The absorb chain is sequential (mod.rs:617-623): chunk c overwrites the rate words with input words 10 c .. 10 c + 9, the
    // padded last chunk included (sponge.rs:41-55: a one after the input, then zeros).  The next chunk is fetched BEFORE the
    // permutation of the current one, so a long single input pays the memory latency once, not once per permutation.
    const long long chunks = row_len / 10 + 1;
    auto rate_word = [&](long long c) -> u64 {
        const long long w = c * 10 + j;
        return w < row_len ? p[w] : (w == row_len ? gl::ONE : 0);
    };
    u64 nxt = j < 10 ? rate_word(0) : 0;
    u64 rcs[5];
    coop_round_constants(j, rcs);
    CoopHalfMatrix hm;
    coop_half_matrix(half, hm);
    stage_lut(lut);
    if (!live) return;
    for (long long c = 0; c < chunks; ++c) {
        if (j < 10) s = nxt;  // overwrite-mode absorb, mod.rs:684-691
        if (j < 10 && c + 1 < chunks) nxt = rate_word(c + 1);
        tip5_permutation_coop_n<ROWS>(s, j, half, lut, rcs, hm);
    }
    long long tree, k;
    split_item(i, per_tree, shift, tree, k);
    if (j < 5 && !half) out[tree * out_ts + k * 5 + j] = s;
}

// the same for few rows: 16 lanes per row
template <int ROWS>
__global__ void __launch_bounds__(256) tip5_hash_table_rows_coop_kernel(const u64* table, long long n_rows, int shift, long long n_cols,
                                                                        int width, long long col_stride, long long table_stride,
                                                                        long long total, u64* out, long long out_ts) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    const int j = threadIdx.x & 15, half = CoopGeom<ROWS>::half();
    const long long id = CoopGeom<ROWS>::item();
    const bool live = id < total;
    long long tree, i;
    split_item(live ? id : total - 1, n_rows, shift, tree, i);
    const u64* tb = table + tree * table_stride;
    const long long row_len = n_cols * width;
    u64 s = 0;
    const long long chunks = row_len / 10 + 1;  // as tip5_hash_varlen_rows_coop_kernel: the next chunk is in flight during a permutation
    auto rate_word = [&](long long c) -> u64 {
        const long long w = c * 10 + j;
        return w < row_len ? table_word(tb, i, w, width, col_stride) : (w == row_len ? gl::ONE : 0);
    };
    u64 nxt = j < 10 ? rate_word(0) : 0;
    u64 rcs[5];
    coop_round_constants(j, rcs);
    CoopHalfMatrix hm;
    coop_half_matrix(half, hm);
    stage_lut(lut);
    if (!live) return;
    for (long long c = 0; c < chunks; ++c) {
        if (j < 10) s = nxt;
        if (j < 10 && c + 1 < chunks) nxt = rate_word(c + 1);
        tip5_permutation_coop_n<ROWS>(s, j, half, lut, rcs, hm);
    }
    if (j < 5 && !half) out[tree * out_ts + i * 5 + j] = s;
}

// Tip5::permutation of state i by row-group i
template <int ROWS>
__global__ void __launch_bounds__(256) tip5_permute_coop_kernel(u64* states, long long count) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    const int j = threadIdx.x & 15, half = CoopGeom<ROWS>::half();
    const long long i = CoopGeom<ROWS>::item();
    const bool live = i < count;
    u64 s = states[(live ? i : count - 1) * 16 + j];
    u64 rcs[5];
    coop_round_constants(j, rcs);
    CoopHalfMatrix hm;
    coop_half_matrix(half, hm);
    stage_lut(lut);
    if (!live) return;
    tip5_permutation_coop_n<ROWS>(s, j, half, lut, rcs, hm);
    if (!half) states[i * 16 + j] = s;
}

// A whole subtree in one workgroup (sequentially_fill_tree, merkle_tree.rs:216-222, below the parallelisation cutoff): the workgroup
// takes `chunk` consecutive nodes of a level of `w` nodes (chunk c of tree `tree`: nodes[w + c chunk .. w + (c + 1) chunk)), computes
// the log2(chunk) levels above them through LDS -- one hash_pair per 16-lane row, a barrier per level, no launch between levels --
// and writes every node it computed to its place in the heap-ordered array (if nd != null) and the subtree's root to `out` (if
// != null).  Near the root of a tree a level is one permutation latency (~2.2 us) whatever its width, so what a level costs is the
// launch around it: a workgroup per subtree pays it once per log2(chunk) levels.
__device__ __forceinline__ void merkle_subtree(const u64* src, int chunk, u64* nd, long long w, long long c, bool copy_input, u64* out,
                                               u64 (*buf)[256 * 5], unsigned char* lut, bool two_rows) {
    const int t = threadIdx.x, j = t & 15, row = t >> 4, rows = blockDim.x >> 4, half = row & 1;
    u64 rcs[5];
    coop_round_constants(j, rcs);  // once for all levels
    CoopHalfMatrix hm;
    coop_half_matrix(half, hm);
    for (int k = t; k < chunk * 5; k += blockDim.x) {
        const u64 v = src[k];
        buf[0][k] = v;
        if (copy_input && nd) nd[(w + c * chunk) * 5 + k] = v;  // the input level is the leaf level (merkle_tree.rs:426)
    }
    stage_lut(lut);  // ends in the barrier that also publishes buf[0]
    int cur = 0;
    long long lw = w;  // nodes in the level being read
    for (int cw = chunk / 2; cw >= 1; cw /= 2) {
        lw /= 2;
        if (two_rows && 2 * cw <= rows) {
            // at most half as many pairs as the workgroup has rows: a row PAIR per hash_pair (tip5_permutation_coop2; rows 2 i and
            // 2 i + 1 are in one wave), the level costs 1.7 us instead of 2.0
            const int i = row >> 1;
            if (i < cw) {  // whole row pairs take the branch together
                u64 s = j < 10 ? buf[cur][10 * i + j] : gl::ONE;
                tip5_permutation_coop2(s, j, half, lut, rcs, hm);
                if (j < 5 && !half) {
                    buf[cur ^ 1][5 * i + j] = s;
                    if (nd) nd[(lw + c * cw + i) * 5 + j] = s;
                }
            }
        } else {
            for (int base = 0; base < cw; base += rows) {
                const int i = base + row;
                if (i < cw) {  // whole rows take the branch together
                    u64 s = j < 10 ? buf[cur][10 * i + j] : gl::ONE;
                    tip5_permutation_coop(s, j, lut, rcs);
                    if (j < 5) {
                        buf[cur ^ 1][5 * i + j] = s;
                        if (nd) nd[(lw + c * cw + i) * 5 + j] = s;
                    }
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (out && t < 5) out[t] = buf[cur][t];
}

// Top of a tree in one workgroup per tree: given the level of `width` (<= 256) nodes, i.e. nodes[width .. 2 width), compute
// nodes[1 .. width) and write them out; also zero nodes[0] (merkle_tree.rs:415-419).
// level_in: pointer to the `width` digests of the starting level for tree 0, stride in_ts words per tree.
// nodes: node array (may be null when only the root is wanted); root_out: 5 words per tree or null.
__global__ void __launch_bounds__(1024) merkle_top_kernel(const u64* level_in, long long in_ts, int width, u64* nodes,
                                                          long long nodes_ts, u64* root_out, const u64* leaves_to_copy,
                                                          long long leaves_ts, int two_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    __shared__ u64 buf[2][256 * 5];
    (void)leaves_ts;  // leaves_to_copy != null only says that level_in IS the leaf level, to be copied into nodes[width..2 width)
    const long long tree = blockIdx.x;
    u64* nd = nodes ? nodes + tree * nodes_ts : nullptr;
    if (nd && threadIdx.x < 5) nd[threadIdx.x] = 0;
    merkle_subtree(level_in + tree * in_ts, width, nd, width, 0, leaves_to_copy != nullptr, root_out ? root_out + tree * 5 : nullptr, buf, lut, two_rows != 0);
}

// The levels between the wide ones (one launch each) and the top: workgroup (tree, c) takes chunk = 2^chunk_log nodes of the level of
// w = chunk << chunks_log nodes and leaves the subtree root in out + tree * out_ts + 5 c (if out != null: the root-only builders keep
// no node array; with one the root is already at nodes[(w >> chunk_log) + c]).
__global__ void __launch_bounds__(1024) merkle_subtree_kernel(const u64* level_in, long long in_ts, int chunk_log, int chunks_log, u64* nodes,
                                                              long long nodes_ts, u64* out, long long out_ts, int copy_input, int two_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    __shared__ u64 buf[2][256 * 5];
    const long long tree = blockIdx.x >> chunks_log, c = blockIdx.x - (tree << chunks_log);
    const int chunk = 1 << chunk_log;
    const long long w = (long long)chunk << chunks_log;
    merkle_subtree(level_in + tree * in_ts + c * chunk * 5, chunk, nodes ? nodes + tree * nodes_ts : nullptr, w, c, copy_input != 0,
                   out ? out + tree * out_ts + 5 * c : nullptr, buf, lut, two_rows != 0);
}

// out[k] = nodes[idx[k]] for digests (5 words): authentication structures from a device-resident tree
__global__ void __launch_bounds__(256) gather_digests_kernel(const u64* nodes, const unsigned long long* idx, long long count, u64* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * 5) return;
    const long long k = i / 5, w = i - 5 * k;
    out[i] = nodes[idx[k] * 5 + w];
}

}  // namespace tfk
