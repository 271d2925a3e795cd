// ntt_kernels.h -- batched Goldilocks NTT passes for gfx950 (device side).
//
// What the reference computes (twenty-first/src/math/ntt.rs:67-82, :109-125, :153-228):
//   out[k] = sum_j x[j] * w_n^(j k)   (natural order in and out; intt uses w^-1 and scales by n^-1)
// as log2(n) radix-2 sweeps over the whole slice.  Here the same transform is a generalized
// Cooley-Tukey factorisation n = N_1 * ... * N_s (s <= 4, N_i <= 1024 -- or 2048, run as pairs of 1024-point workgroups that
// share their input: PRE2 below, the two-pass plan of 2^21 / 2^22 points), one HBM pass per factor:
//
//   pass i (not last):  for every (k_1..k_{i-1}, j_{i+1}..j_s):  DFT over j_i, then multiply by the
//                       inter-pass twiddle w_{N_i...N_s}^(k_i * b_i); data stays in place.
//   last pass:          DFT over j_s and write to the digit-reversed position k_1 + N_1 (k_2 + ...),
//                       so the result is in natural order with no bit-reversal sweep.
//
// Inside a pass a workgroup owns a tile of `nc` independent length-R DFTs ("columns"), R = 32 * P2:
//   step 1: every thread holds 32 rows of one column in registers and runs a radix-32 DIT network
//           whose twiddles are all powers of two (gl::Pow2Mul: shl_fold / shl_monty, no 64x64 multiply);
//   inner twiddle w_R^(g k1) (one Montgomery multiply per element; n^-1 of the inverse folded in);
//   one LDS exchange, done in rounds of `cpr` columns so two workgroups fit the 160 KiB LDS (the thread
//           that owned row group g of a column now owns the outputs k1 = g (mod P2) of that column);
//   step 2: radix-P2 DIT network on the other index, again shift-only;
//   inter-pass twiddle (one Montgomery multiply), coalesced store.
// So an element costs 2-3 general multiplies per n = 2^20 transform instead of 10.
//
// XFE slices ([c0,c1,c2] per element, x_field_element.rs:56-59) are three interleaved BFE columns
// with the same twiddles (ntt.rs:203-207, x_field_element.rs:540-548): L = 3 words per element.
//
// Kernels in this file and the lengths they serve (planner: run_ntt in tf_hip.hip):
//   ntt_tiny_kernel    n <= 16                the reference's radix-2 sweeps, one transform per thread
//   ntt_rows32_kernel  n == 32, BFE           tiles of 512 transforms staged through LDS, one per thread, no exchange
//   ntt_pass_kernel    32 <= n <= 1024        one pass, rows of T whole transforms per workgroup
//                      n > 2^14 (and XFE > 1024): 2-4 passes as described above; the R = 1024 instantiations
//                      (LAST1024: row-major load roles / column-major store roles, stores fused with the last radix-2
//                      level; R1024: constant P2) are the two kernels of the 2^20-point headline transform
//   ntt_block_kernel   2^11 <= n <= 2^14, BFE one workgroup owns a whole transform: radix 32 x 32 x P3 with two LDS
//                      exchanges, so these lengths cost one HBM pass instead of two
//   ntt_lat_kernel     64 <= n <= 4096, calls with little work: 8 elements per thread, radix-8 Stockham stages through LDS
#pragma once

#include "gl64.h"

namespace tfk {

using gl::u32;
using gl::u64;

struct NttPassArgs {
    const u64* in;
    u64* out;
    const u64* inner_tw;   // [P2][32] Montgomery words: w_R^(+-g*k1) (times n^-1 for the last inverse pass); may be null
    const u64* post_tw;    // inter-pass twiddles T[k * tw_rs + b] (Montgomery words) or null
    const u64* pre_scale;  // coset powers S[j] (Montgomery words) or null        (polynomial.rs:760-773)
    long long n_out;       // LAST1024 only: >= 0 = store only output elements j < n_out (truncated product); < 0: all
    const u64* in2;        // SCALE = 1 only: second operand with the layout of `in`, multiplied in on load (or null)
    const u64* post_scale; // interpolation powers offset^-j applied to output element j, or null (polynomial.rs:1907-1918)
    long long js_i0, js_i1, js_i2, js_c, js_k;  // output element index j = i0*js_i0 + i1*js_i1 + i2*js_i2 + (c/L)*js_c + k*js_k (last pass only)
    long long n_coeffs;    // elements present per input polynomial; rows beyond are zero (polynomial.rs:1395); <0: no padding
    long long ps_i1;                         // pre_scale index offset per outer index i1 (blown-up coset evaluation: table c)
    long long ib0, ib1, ib2, ob0, ob1, ob2;  // tile base strides (words)
    long long in_cs_hi, out_cs_hi;           // column c -> (c / L) * cs_hi + (c % L)
    long long in_rs, out_rs;                 // row strides (words)
    long long tw_rs;                         // row stride of post_tw (= B)
    long long ps_rs;                         // pre_scale index j = r * ps_rs + (ps_col ? b : 0)
    u32 d1, d2;                              // tile id = (i0 * d1 + i1) * d2 + i2
    u32 d01;                                 // d0 * d1 (for the XCD-aware order)
    int p2;                                  // log2 P2  (R = 32 << p2)
    int nc;                                  // columns per tile; thread t owns column t % nc, row group t / nc
    int L;                                   // words per element (1 BFE, 3 XFE)
    int col_limit;                           // valid columns along i2: min(nc, col_limit - i2 * nc)
    int ps_col;
    int cpr, nrounds;                        // LDS exchange in `nrounds` rounds of `cpr` columns (bounds the LDS footprint)
    int s1, s2, s3;                          // LDS strides in u64: idx = k1*s1 + g*s2 + (c % cpr)*s3
    int gfast;                               // 1: thread t is (g = t % P2, column t / P2) instead of (t / nc, t % nc)
    int xcd_order;                           // G > 0: XCD-aware tile order in groups of G adjacent column tiles (0: natural order)
    int xcd_colfast;                         // with xcd_order: walk the XCD's column groups fastest (their table slices fit its L2)
    int wtiles;                              // transposing passes: word-granular tiles (see the kernel); LAST1024 always works this way
    int col_shift0, col_shift_i0, col_wrap;  // LAST1024: tile i2 of batch entry i0 covers columns [16 i2 - s, 16 i2 - s + 16) mod
                                             // col_wrap (= N_1), s = (col_shift0 + i0 * col_shift_i0) mod 16, so its 128-byte output
                                             // segments start on cache lines even when the output stride is not a multiple of 16 words
                                             // (truncated products); tile 0 wraps around to the last s columns
    u32 nc_magic;                            // t / nc == umulhi(t, nc_magic) for every t < blockDim (checked by the planner)
    int nt;                                  // bit 0: non-temporal loads of the input, bit 1: non-temporal stores of the output (plain
                                             // transforms only): streams that are touched once stay out of the Infinity Cache, which is
                                             // then left to the scratch tile between the passes (TF_NTT_NT, planner)
    unsigned long long* dbg;                 // MODE 3 only: per-wave cycle stamps (6 per wave)
    // PRE2 instantiations only (a 2048-point pass as two 1024-point passes, see the kernel):
    long long pre2_in_off;                   // words from a row to its partner row 1024 rows further
    long long pre2_out_off;                  // words added to `out` by the odd half (the strides above already step two rows)
    long long pre2_tw_off;                   // ... to post_tw
    long long pre2_js_off;                   // ... to the output element index (SCALE 2)
    const u64* pre2_cp;                      // SCALE 1: -> offset^(index distance of partner rows), multiplies the partner coefficient
    int pre2_map;                            // 0: not a PRE2 launch; 1: half = bit 3 of the block id (a pair shares an XCD); 2: bit 0
};

// ---- buffer addressing for the R = 1024 instantiations --------------------------------------------------------------------
// A global_load/store takes ONE 64-bit address per lane; with 32 row slots per thread the compiler keeps 32 uniform bases and
// forms every address with a 64-bit VALU add (v_lshl_add_u64 / v_mad_u64_u32: ~100 VALU instructions per burst, 7 % of a pass).
// A buffer instruction adds  resource base (4 SGPRs, per workgroup) + 32-bit per-lane offset (ONE VGPR for all slots) + 32-bit
// uniform offset (one SGPR per slot)  in the address unit: no VALU work at all.  The planner only selects these instantiations
// when every slot offset plus thread offset fits 32 bits (fits_buffer_offsets in tf_hip.hip); larger transforms take the
// generic kernel with plain pointers.
typedef unsigned int tf_v2u __attribute__((__vector_size__(2 * sizeof(unsigned int))));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);  // raw buffer, 4 GiB window, no swizzle
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc_n(const void* p, u32 nbytes) {  // loads beyond nbytes return zero
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)nbytes, 0x00020000);
}
#ifndef TF_LOAD_AUX
#define TF_LOAD_AUX 2  // cache-policy bits of the data stream's buffer loads / stores: 2 = nt (streamed once; 0 for an A/B build).  Measured on 256 x 2^20: 1.919 -> 1.845 ms with both (loads alone 1.889, stores alone 1.890, profiles/r02f)
#endif
#ifndef TF_STORE_AUX
#define TF_STORE_AUX 2
#endif
// per-role overrides for A/B builds: the scratch tile between the passes (stores of a column pass, loads of the last pass)
#ifndef TF_AUX_COL_STORE
#define TF_AUX_COL_STORE TF_STORE_AUX
#endif
#ifndef TF_AUX_LAST_LOAD
#define TF_AUX_LAST_LOAD TF_LOAD_AUX
#endif
template <int AUX = TF_LOAD_AUX>
__device__ __forceinline__ u64 buf_load(__amdgpu_buffer_rsrc_t r, u32 voff, u32 soff) {
    const tf_v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    return ((u64)v[1] << 32) | v[0];
}
__device__ __forceinline__ u64 buf_load_tab(__amdgpu_buffer_rsrc_t r, u32 voff, u32 soff) {  // twiddle tables: default policy (they are re-read)
    const tf_v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return ((u64)v[1] << 32) | v[0];
}
template <int AUX = TF_STORE_AUX>
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, u32 voff, u32 soff, u64 x) {
    tf_v2u v;
    v[0] = (u32)x;
    v[1] = (u32)(x >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, AUX);
}

// ---- radix-2^k DIT network with power-of-two twiddles --------------------------------------
// w_{2^l} = 2^(39 * 2^(6-l))  (b_field_element.rs:46-51: w_64 = 2^39, ..., w_2 = 2^96 = -1), order 192.
template <bool INV, int LVL, int J>
struct TwExp {
    static constexpr int fwd = ((39 << (6 - LVL)) * J) % 192;
    static constexpr int value = INV ? (192 - fwd) % 192 : fwd;
};

#ifndef TF_ASM_BFLY
#define TF_ASM_BFLY 1  // 0: the compiler's compare-and-select add/sub (12 VALU per butterfly instead of 10)
#endif
#ifndef TF_LDS_TW
#define TF_LDS_TW 1  // the R = 1024 instantiations stage their [32][32] inner twiddle table in LDS (0: per-thread global loads, A/B build)
#endif
// The inner table w_R^(g k1) is read by row g: 32 words = 256 contiguous bytes per THREAD, i.e. a wave-load touches up to 64
// different cache lines.  Through global memory those 16 dwordx4 loads per thread compete with the data stream for the
// texture-addresser / L1 path (measured: 2.148 -> 2.081 ms per 256 x 2^20 with the column pass alone reading it from LDS,
// profiles/r02b_ab_variants.txt); staged once per workgroup behind the exchange buffer, rows padded to 34 words (272 bytes:
// 16-byte aligned for ds_read_b128, consecutive rows 4 banks apart), the reads are conflict-free LDS traffic.
constexpr int kLdsTwStride = 34;
// exchange geometry of the R1024 instantiation (= what finish_geometry computes for 512 threads and rounds of 8 192 elements)
constexpr int kR1024Nc = 16, kR1024Cpr = 8, kR1024Rounds = 2, kR1024S1 = 264;
// LAST1024 exchange layout: element (k1, g, cc) at k1 * kL1024S1 + cc * kL1024CS + g  (see the kernel)
constexpr int kL1024S1 = 273, kL1024CS = 34;
constexpr int kL1024ExchangeWords = ((31 * kL1024S1 + 7 * kL1024CS + 32 + 1) / 2) * 2;  // 16-byte aligned end
#ifndef TF_LAZY
#define TF_LAZY 1  // 0: every network canonical (A/B build); 1: lazy butterflies in the networks that are followed by a Montgomery product
#endif

// Butterfly I (0 .. 15) of level LVL over 32 register slots: groups of 2^LVL consecutive slots, butterfly j of a group pairs
// slots (base + j, base + j + 2^(LVL-1)) with the twiddle w_{2^LVL}^j (inputs of a group in bit-reversed order, outputs natural).
template <bool INV, int LVL, int I>
struct Bf {
    static constexpr int H = 1 << (LVL - 1);
    static constexpr int j = I % H;
    static constexpr int ia = (I / H) * 2 * H + j;
    static constexpr int ib = ia + H;
    static constexpr int E = TwExp<INV, LVL, j>::value;
    static constexpr bool neg = gl::Pow2Mul<E>::negate;  // the power-of-two product comes back negated: swap the outputs
};

// x * 2^E up to the sign Bf::neg, as a CANONICAL word whatever 64-bit word x is (shl_fold / shl_monty reduce fully).
// E = 0 passes x through: canonical in a canonical network; in a lazy network only level 1 meets that case with canonical inputs
// (loaded words or Montgomery products), the levels above canonicalise the operand first (LAZY_IN).
template <int E, bool LAZY_IN>
__device__ __forceinline__ u64 tw_operand(u64 b) {
    if constexpr (E % 192 == 0) {
        if constexpr (LAZY_IN) return gl::add(b, 0);  // b >= p ? b - p : b   (4 VALU)
        return b;
    } else {
        return gl::Pow2Mul<E>::apply(b);
    }
}

// Two butterflies (I, I + 1) of one level in one block of interleaved carry chains (gl::add_sub2 / gl::add_sub_lazy2).
//   LAZY = false: canonical inputs and outputs (ten VALU per butterfly).
//   LAZY = true:  the first operand of a butterfly may be any 64-bit word congruent to the element, the outputs are such words
//                 (eight VALU per butterfly); the twiddled operand is always canonical (tw_operand).
template <bool INV, int LVL, int I, bool LAZY>
__device__ __forceinline__ void butterfly_pair(u64 (&x)[32]) {
    using B0 = Bf<INV, LVL, I>;
    using B1 = Bf<INV, LVL, I + 1>;
    const u64 v0 = tw_operand<B0::E, LAZY && (LVL > 1)>(x[B0::ib]);
    const u64 v1 = tw_operand<B1::E, LAZY && (LVL > 1)>(x[B1::ib]);
    u64 s0, d0, s1, d1;
    if constexpr (LAZY) gl::add_sub_lazy2(x[B0::ia], v0, x[B1::ia], v1, s0, d0, s1, d1);
    else gl::add_sub2(x[B0::ia], v0, x[B1::ia], v1, s0, d0, s1, d1);
    x[B0::ia] = B0::neg ? d0 : s0;
    x[B0::ib] = B0::neg ? s0 : d0;
    x[B1::ia] = B1::neg ? d1 : s1;
    x[B1::ib] = B1::neg ? s1 : d1;
}

template <int E>
__device__ __forceinline__ void butterfly_pow2(u64& a, u64& b) {
    // (a, b) -> (a + b * 2^E, a - b * 2^E), canonical in and out; the sign of the power-of-two product is folded into add/sub
    const u64 v = gl::Pow2Mul<E>::apply(b);
#if TF_ASM_BFLY
    if constexpr (!gl::Pow2Mul<E>::negate) gl::add_sub(a, v, a, b);
    else gl::add_sub(a, v, b, a);
    return;
#endif
    if constexpr (!gl::Pow2Mul<E>::negate) {
        const u64 s = gl::add(a, v);
        b = gl::sub(a, v);
        a = s;
    } else {
        const u64 s = gl::sub(a, v);
        b = gl::add(a, v);
        a = s;
    }
}

// butterflies [I, END) of level LVL, two at a time
template <bool INV, int LVL, int I, int END, bool LAZY>
struct DitRange {
    static __device__ __forceinline__ void run(u64 (&x)[32]) {
#if TF_ASM_BFLY
        butterfly_pair<INV, LVL, I, LAZY>(x);
#else
        butterfly_pow2<Bf<INV, LVL, I>::E>(x[Bf<INV, LVL, I>::ia], x[Bf<INV, LVL, I>::ib]);
        butterfly_pow2<Bf<INV, LVL, I + 1>::E>(x[Bf<INV, LVL, I + 1>::ia], x[Bf<INV, LVL, I + 1>::ib]);
#endif
        if constexpr (I + 2 < END) DitRange<INV, LVL, I + 2, END, LAZY>::run(x);
    }
};
// Level LVL of a DIT network over all 32 registers.
template <bool INV, int LVL, bool LAZY = false>
__device__ __forceinline__ void dit_level(u64 (&x)[32]) { DitRange<INV, LVL, 0, 16, LAZY && TF_LAZY>::run(x); }

// levels 1..4 restricted to the 16 register slots starting at FIRST (0 or 16): butterflies FIRST/2 .. FIRST/2 + 7 of each level
template <bool INV, int FIRST, bool LAZY = false>
__device__ __forceinline__ void dit_half(u64 (&x)[32]) {
    DitRange<INV, 1, FIRST / 2, FIRST / 2 + 8, LAZY && TF_LAZY>::run(x);
    DitRange<INV, 2, FIRST / 2, FIRST / 2 + 8, LAZY && TF_LAZY>::run(x);
    DitRange<INV, 3, FIRST / 2, FIRST / 2 + 8, LAZY && TF_LAZY>::run(x);
    DitRange<INV, 4, FIRST / 2, FIRST / 2 + 8, LAZY && TF_LAZY>::run(x);
}

// x[q0 .. q0+3] *= w[0 .. 3]  (four Montgomery products, no wait-state nops: gl::mont_mul4)
__device__ __forceinline__ void mul4_inplace(u64 (&x)[32], int q0, u64 w0, u64 w1, u64 w2, u64 w3) {
    const u64 a4[4] = {x[q0], x[q0 + 1], x[q0 + 2], x[q0 + 3]}, b4[4] = {w0, w1, w2, w3};
    u64 r4[4];
    gl::mont_mul4(a4, b4, r4);
    x[q0] = r4[0], x[q0 + 1] = r4[1], x[q0 + 2] = r4[2], x[q0 + 3] = r4[3];
}

// Level 5 of the radix-32 network for the four butterflies (Q0+i, Q0+i+16), i < 4, followed by their eight stores.
// TRUNC: only the slots q < qlim are stored (output truncated to the first n_out elements, fast_multiply).
// SCALED: every output is multiplied by its word of a table laid out like the output (fast_coset_interpolate's offset^-j):
//         sbase + 32 q s_rs_bytes + soff for slot q.
template <bool INV, int Q0, bool TRUNC = false, bool SCALED = false>
__device__ __forceinline__ void tail_p5(u64 (&x)[32], bool act, char* obase, u32 toff, long long out_rs_bytes, int qlim = 32,
                                        const char* sbase = nullptr, u32 soff = 0, long long s_rs_bytes = 0, bool nt = false) {
    DitRange<INV, 5, Q0, Q0 + 4, false>::run(x);  // canonical: these words are stored
    if (act) {
        const __amdgpu_buffer_rsrc_t ro = buf_rsrc(obase);  // LAST1024 only: the planner checked that the offsets fit (see buf_load)
        if constexpr (SCALED) {
            u64 w[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i] = *reinterpret_cast<const u64*>(sbase + (long long)(32 * (Q0 + i)) * s_rs_bytes + soff);
                w[4 + i] = *reinterpret_cast<const u64*>(sbase + (long long)(32 * (Q0 + i + 16)) * s_rs_bytes + soff);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) gl::mont_mul2(x[Q0 + i], w[i], x[Q0 + i + 16], w[4 + i], x[Q0 + i], x[Q0 + i + 16]);
        }
        (void)nt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!TRUNC || Q0 + i < qlim) buf_store(ro, toff, (u32)((long long)(32 * (Q0 + i)) * out_rs_bytes), x[Q0 + i]);
            if (!TRUNC || Q0 + i + 16 < qlim) buf_store(ro, toff, (u32)((long long)(32 * (Q0 + i + 16)) * out_rs_bytes), x[Q0 + i + 16]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ constexpr int brev5(int q) {
    return ((q & 1) << 4) | ((q & 2) << 2) | (q & 4) | ((q & 8) >> 2) | ((q & 16) >> 4);
}

__device__ __forceinline__ u32 div_by_L(u32 v, int L) {  // v / L for L in {1, 3}
    return L == 1 ? v : (u32)(((unsigned long long)v * 0xAAAAAAABull) >> 33);
}

// SCALE = 1: first pass of fast_coset_evaluate -- multiply coefficient j by offset^j on load and read rows beyond
//            n_coeffs as zero (polynomial.rs:760-773, :1394-1395).
// SCALE = 2: last pass of fast_coset_interpolate -- multiply output coefficient j by offset^-j on store
//            (polynomial.rs:1907-1918: intt, then scale by the inverse offset).
// MODE: 0 = product kernel.  1 / 2 are measurement-only ablations (TF_NTT_ABLATE, never the default):
//   1 = no global loads/stores (synthetic operands), 2 = no arithmetic (loads, LDS exchange, stores only).
// PRE2 (R = 1024 instantiations only): the pass is a 2048-point DFT per column.  One radix-2 decimation-in-frequency stage is
//   fused into the load,  y_r = x_r + x_{r+1024},  z_r = (x_r - x_{r+1024}) w_2048^r,  and the 1024-point DFT of y gives the even
//   outputs, that of z the odd ones: every tile is run by TWO workgroups (half 0: y, half 1: z) that read the same 32 Ki input
//   elements and write interleaved output rows.  The pair is dispatched back to back on one XCD, so the second read of a line is
//   served by its L2.  This is what lets 2^21 / 2^22-point transforms run in two HBM round trips with 16-column (128-byte)
//   tiles: 2048 rows x 16 columns do not fit one workgroup next to a second one on the CU.
//   w_2048^r, r = g + 32 i:  w_2048^(32 i) = w_64^i is a power of two (a shift per element), and w_2048^g rides on the inner
//   twiddle: the odd half's table is w_1024^(g k1) w_2048^g = w_2048^(g (2 k1 + 1))  (inner_tw holds both tables, [2][32][32]).
template <bool INV, int Q>
struct Pre2Slot {  // register slot Q holds row g + 32 brev5(Q)
    static constexpr int i = ((Q & 1) << 4) | ((Q & 2) << 2) | (Q & 4) | ((Q & 8) >> 2) | ((Q & 16) >> 4);
    static constexpr int E = TwExp<INV, 6, i>::value;
    static constexpr bool neg = gl::Pow2Mul<E>::negate;
};
// x[Q] = x[Q] -+ w (odd half, the sign of slot Q's power-of-two product folded in) or x[Q] + w (even half), slots Q0 .. Q0+7.
// Even half: the sums feed level 1 of a lazy network, whose second operands (odd slots) must be canonical; the first operands
// may be any representative (one correction instead of a compare-and-select).
template <bool INV, int Q0, int I = 0>
__device__ __forceinline__ void pre2_combine8(u64 (&x)[32], const u64 (&w)[8], bool odd) {
    constexpr int Q = Q0 + I;
    if (odd) {
        x[Q] = Pre2Slot<INV, Q>::neg ? gl::sub(w[I], x[Q]) : gl::sub(x[Q], w[I]);
    } else if constexpr ((Q & 1) || !TF_LAZY) {  // (an all-canonical A/B build, TF_LAZY = 0, keeps every sum canonical)
        x[Q] = gl::add(x[Q], w[I]);
    } else {
        const u64 s = x[Q] + w[I];
        x[Q] = s < w[I] ? s + gl::EPS : s;  // 2^64 = EPS (mod p); a + b - 2^64 + EPS < 2^64 for canonical a, b
    }
    if constexpr (I + 1 < 8) pre2_combine8<INV, Q0, I + 1>(x, w, odd);
}
// x[Q] *= w_64^(brev5 Q) up to the sign pre2_combine8 already took care of: shifts only (slots Q .. END-1; slot 0 has exponent 0)
template <bool INV, int Q = 1, int END = 32>
__device__ __forceinline__ void pre2_shift(u64 (&x)[32]) {
    x[Q] = gl::Pow2Mul<Pre2Slot<INV, Q>::E>::apply(x[Q]);
    if constexpr (Q + 1 < END) pre2_shift<INV, Q + 1, END>(x);
}

template <bool INV, int SCALE, int MODE = 0, bool LAST1024 = false, bool R1024 = false, bool COL = false, bool PRE2 = false>
#ifndef TF_PRIO_LOAD
#define TF_PRIO_LOAD 3
#endif
#ifndef TF_PRIO_STEP2
#define TF_PRIO_STEP2 0
#endif
#ifndef TF_NTT_WAVES
#define TF_NTT_WAVES 4
#endif
__global__ void __launch_bounds__(512, TF_NTT_WAVES) ntt_pass_kernel(const NttPassArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int t = threadIdx.x;
    // LAST1024: R = 1024, no output multiplier (last pass of a forward/inverse NTT).  R1024: R = 1024 with the generic
    // tail (a column pass of a 2^20-point transform): the constant P2 folds the slot index arithmetic at compile time.
    const int p2 = (LAST1024 || R1024) ? 5 : A.p2;
    const int P2 = 1 << p2;
    // Lazy networks (TF_LAZY): step 1 of the R = 1024 instantiations is always followed by the inner-twiddle Montgomery product,
    // step 2 of the column pass by the inter-pass one; a Montgomery product takes any 64-bit representative and returns the
    // canonical word, so these networks run on non-canonical words (8 VALU per butterfly).  Everything else stays canonical.
    // COL: any column pass (inter-pass table present) of a plan whose offsets fit the buffer window -- the R1024 treatment
    // (lazy networks, buffer addressing) with a run-time P2; R1024 is its constant-P2 special case.
    constexpr bool COLP = R1024 || COL;
    constexpr bool LAZY1 = (LAST1024 || COLP) && MODE != 2;
    constexpr bool LAZY2 = COLP && MODE == 0;
    const int L = A.L;

    static_assert(!PRE2 || ((LAST1024 || R1024) && MODE == 0 && SCALE != 3), "PRE2 is a variant of the R = 1024 kernels");
    // PRE2: two workgroups per tile; `half` selects the even (y) or odd (z) outputs
    u32 bid = blockIdx.x, half = 0;
    if constexpr (PRE2) {
        if (A.pre2_map == 1) half = (bid >> 3) & 1u, bid = (bid & 7u) | ((bid >> 4) << 3);
        else half = bid & 1u, bid >>= 1;
    }
    u32 i0, i1, i2;
    if (A.xcd_order) {
        // Blocks are dealt round-robin to the 8 XCDs (b % 8).  Give XCD x the column tiles = x (mod 8) and let
        // it walk all (batch, outer) indices of one column tile back to back, so the slice of the inter-pass
        // twiddle table that tile needs stays in that XCD's L2.  Placement only affects speed.
        // xcd_order = G (1 or 2): groups of G adjacent column tiles stay together; with G = 2 the two 64-byte
        // halves of every 128-byte line are requested back to back from the same XCD (second one hits its L2).
        const u32 G = (u32)A.xcd_order;
        const u32 xcd = bid & 7u, slot = bid >> 3;
        // Order inside an XCD: column groups fastest when the inter-pass table slices of ALL its column tiles fit its L2
        // together (the planner decides: xcd_colfast) -- neighbouring workgroups then stream from different DRAM pages
        // instead of the same column of different batch entries (-1.1 % on 256 x 2^20); batch entries fastest otherwise.
        const u32 ngrp = A.d2 / (8u * G);
        const u32 grp = A.xcd_colfast ? slot % ngrp : slot / (G * A.d01);
        const u32 within = A.xcd_colfast ? slot / ngrp : slot % (G * A.d01);
        i2 = ((grp << 3) | xcd) * G + within % G;
        const u32 rest = within / G;
        i1 = rest % A.d1;
        i0 = rest / A.d1;
    } else {
        const u32 tile = bid;
        i2 = tile % A.d2;
        const u32 rest = tile / A.d2;
        i1 = rest % A.d1;
        i0 = rest / A.d1;
    }
    const u64* in = A.in + (long long)i0 * A.ib0 + (long long)i1 * A.ib1 + (long long)i2 * A.ib2;
    u64* out = A.out + (long long)i0 * A.ob0 + (long long)i1 * A.ob1 + (long long)i2 * A.ob2;
    // LAST1024 tiles are WORD-granular: tile i2 of batch entry i0 covers the word-columns [nc i2 - s, nc i2 - s + nc) of the
    // N_1 * L words that are adjacent on the output side (nc = 16 = one 128-byte line for the transposing pass; an XFieldElement
    // tile may start and end inside an element).  s = cshift is the word address of the entry's first output mod 16, so
    // every 16-word store segment is one whole cache line; tile 0 takes the last s columns of the row instead of the
    // columns below 0 (col_wrap).  Word-column w is limb w % L of element column w / L.
    // The other last-pass instantiations tile the same way when the planner asks (A.wtiles: XFE rows, so that their store
    // segments are whole lines as well), without the shift.
    const bool wt = LAST1024 || A.wtiles;
    const int cshift = LAST1024 ? (int)((A.col_shift0 + i0 * (u32)A.col_shift_i0) & 15u) : 0;
    // R1024: the planner launches this instantiation only with the standard geometry (512 threads, 16 columns, exchange rounds of
    // 8 columns: kR1024* below), so the role arithmetic and every LDS offset of the exchange fold into immediates
    constexpr bool GEO = R1024;
    const int nc_ = GEO ? kR1024Nc : A.nc;
    const int cpr_ = GEO ? kR1024Cpr : A.cpr, nrounds_ = GEO ? kR1024Rounds : A.nrounds;
    const int s1_ = GEO ? kR1024S1 : A.s1, s2_ = GEO ? kR1024Cpr : A.s2, s3_ = GEO ? 1 : A.s3;
    const int col0 = (int)i2 * nc_ - cshift;
    const int ncv = min(nc_, A.col_limit - col0);
    const int ch0 = wt ? (int)div_by_L((u32)max(col0, 0), L) : 0;  // first element column of the tile (uniform)
    if (wt) {
        in = A.in + (long long)i0 * A.ib0 + (long long)i1 * A.ib1 + (long long)ch0 * A.in_cs_hi;
        out = A.out + (long long)i0 * A.ob0 + (long long)i1 * A.ob1 + (long long)ch0 * A.out_cs_hi;
    }
    if constexpr (PRE2) out += (long long)half * A.pre2_out_off;

    // LAST1024 always runs 512 threads in 16 column slots
    // gfast (single-pass transforms, whose "columns" are whole rows of contiguous elements): lanes along the row, g = t % P2
    const int g = (LAST1024 || GEO) ? (t >> 4) : (A.gfast ? (t & (P2 - 1)) : (A.nc == 1 ? t : (int)__umulhi((u32)t, A.nc_magic)));  // t / nc
    const int c = (LAST1024 || GEO) ? (t & 15) : (A.gfast ? (t >> p2) : (t - g * A.nc));                                              // t % nc
    const bool act = c < ncv;
    int ch, cl;  // element column relative to the tile base, limb
    if (wt) {
        const int w = col0 + c + ((LAST1024 && col0 + c < 0) ? A.col_wrap : 0);
        const int e = (int)div_by_L((u32)w, L);
        ch = e - ch0, cl = w - e * L;
    } else {
        ch = (int)div_by_L((u32)c, L), cl = c - ch * L;
    }
    const long long bcol = (long long)div_by_L((u32)max(col0 + c, 0), L);
    // LAST1024 reads rows of 1024 contiguous elements and writes 16 adjacent columns: the two sides want different
    // lane orders, and the LDS exchange between them lets each have its own.  Loads and step 1 run with the lanes along the
    // row: a wave reads 2 columns x 32 consecutive elements (four 128-byte lines) instead of 16 pieces of 32 bytes
    // (measured -2.4 % on the 256 x 2^20 workload); the exchange hands the data to the column-major roles (g, c) for
    // step 2.  Thread bit 3 selects the exchange round (column half) in BOTH roles, so a thread still writes its 32 old
    // values and reads its 32 new ones in the same round: g_in = bits {0,1,2,4,5}, c_in = bits {6,7,8} + 8 * bit 3.
    int g_in = LAST1024 ? ((t & 7) | ((t >> 1) & 0x18)) : g;
    int c_in = LAST1024 ? (((t >> 6) & 7) | (t & 8)) : c;
#ifndef TF_XFE_ROLES
#define TF_XFE_ROLES 1
#endif
    if constexpr (LAST1024 && TF_XFE_ROLES) {
        // XFieldElement rows: consecutive elements of one limb are 24 bytes apart, so the assignment above makes a wave-load
        // touch 768 bytes for 256 it uses (measured: 1.53 x the bytes of the pass fetched on the L2's memory side).  Here the
        // 256 (column, g) pairs of an exchange round are dealt to the round's 256 threads in ADDRESS order instead -- within a
        // row the words (g, limb) are contiguous -- so the 32 lanes of a wave that share a round read 256 contiguous bytes.
        // The round is still thread bit 3, as the exchange needs.  (Not for the wrapped first tile of a shifted row.)
        if (L == 3 && col0 >= 0) {  // uniform
            const int r = (t >> 3) & 1;
            const int v = (t & 7) | ((t >> 4) << 3);  // rank of the thread among the 256 of its round
            const int wc0 = col0 + 8 * r;
            int nk = 3 - (wc0 - 3 * (int)div_by_L((u32)wc0, 3));  // limbs of the first (possibly partial) row in this round
            int start = 0, colbase = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // at most four row segments: (1 | 2 | 3), 3, 3, rest
                if (v >= start + 32 * nk) {
                    start += 32 * nk;
                    colbase += nk;
                    nk = min(3, 8 - colbase);
                }
            }
            const int local = v - start;
            const int gq = nk == 1 ? local : (nk == 2 ? (local >> 1) : (int)__umulhi((u32)local, 0x55555556u));
            g_in = gq;
            c_in = 8 * r + colbase + (local - gq * nk);
        }
    }
    const bool act_in = LAST1024 ? (c_in < ncv) : act;
    int ch_in = ch, cl_in = cl;
    if constexpr (LAST1024) {
        const int w = col0 + c_in + (col0 + c_in < 0 ? A.col_wrap : 0);
        const int e = (int)div_by_L((u32)w, L);
        ch_in = e - ch0, cl_in = w - e * L;
    }

    // Addressing: every global access is  uniform 64-bit base (SGPRs, one per register slot q)  +  32-bit
    // per-thread offset (one VGPR for all 32 slots), so the load and store bursts cost (almost) no vector ALU
    // work.  VALU arbitration favours the OLDER workgroup on a CU, so a young workgroup whose loads needed
    // address arithmetic would not get them issued until the old one finished computing; together with the
    // s_setprio brackets this is what lets one workgroup's memory phase overlap the other's arithmetic.
    unsigned long long stamp[6];
    if constexpr (MODE == 3) stamp[0] = __builtin_readcyclecounter();
    u64 x[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = 0;
    constexpr bool LDS_TW = TF_LDS_TW && MODE == 0;
    u64* const ltw = lds + (LAST1024 ? kL1024ExchangeWords : 32 * s1_);  // the staged inner table [P2][32], behind the exchange buffer
    if constexpr (LDS_TW) {
        if ((LAST1024 || R1024) || A.inner_tw) {  // uniform
            const u64* itw = A.inner_tw + (PRE2 ? (int)half * 1024 : 0);
            for (int i = t; i < 32 * P2; i += blockDim.x) ltw[(i >> 5) * kLdsTwStride + (i & 31)] = itw[i];
            __syncthreads();
        }
    }
    // ------------------------------------------------------------------ load + step 1 (radix 32 over i, rows g + P2*i)
    __builtin_amdgcn_s_setprio(TF_PRIO_LOAD);
    if constexpr (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 32; ++q) x[q] = (u64)(t * 32 + q) * 0x9e3779b97f4a7c15ULL >> 1;
    } else if (PRE2 && act_in) {
        // the tile's 32 Ki input elements: rows r = g + 32 i in x[], their partner rows r + 1024 eight at a time, combined at once
        const u32 toff = (u32)(((long long)ch_in * A.in_cs_hi + cl_in + (long long)g_in * A.in_rs) * 8);
        u32 nrec = 0xffffffffu;
        if constexpr (SCALE == 1) {
            if (A.n_coeffs >= 0) {  // rows beyond the coefficients read as zero: the buffer's record count does it
                const long long rem = (A.n_coeffs * L - (long long)(in - (A.in + (long long)i0 * A.ib0))) * 8;
                nrec = rem <= 0 ? 0u : (u32)min(rem, 0xffffffffll);
            }
        }
        const __amdgpu_buffer_rsrc_t ri = buf_rsrc_n(in, nrec);
        const u32 poff = (u32)(A.pre2_in_off * 8);
        u64 pc = 0;
        if constexpr (SCALE == 1) {
            if (A.pre_scale) pc = *A.pre2_cp;  // uniform
        }
        // (the range check of a raw buffer covers the per-lane offset only, not the scalar one: with zero padding the whole
        //  offset goes through the VGPR -- one v_add_u32 per load)
        constexpr bool CHK = SCALE == 1;
#ifndef TF_PRE2_LOAD_AUX
#define TF_PRE2_LOAD_AUX 0  // default cache policy: the partner workgroup's read of the same lines should find them in the L2
#endif
        constexpr int AUX = TF_PRE2_LOAD_AUX;
#ifndef TF_PRE2_BURST16
#define TF_PRE2_BURST16 2  // 2: 48 loads (slots 0-15, their partners, slots 16-31), then the last 16 partners in flight under the first
                           //    half's scaling / shifts / levels 1-4; 1: two bursts of 32 loads; 0: 32 loads, then the partners eight at a time (A/B)
#endif
        const auto slot_off = [&](int q) { return (u32)((long long)(brev5(q) << p2) * A.in_rs * 8); };
        const auto ld = [&](u32 so) { return CHK ? buf_load<AUX>(ri, toff + so, 0) : buf_load<AUX>(ri, toff, so); };
        const auto scale8 = [&](u64 (&w)[8]) {  // coefficient j + d is scaled by offset^(j + d) = offset^j * c: the common factor follows below
            if constexpr (SCALE == 1) {
                if (A.pre_scale) {
#pragma unroll
                    for (int i = 0; i < 8; i += 4) {
                        const u64 a4[4] = {w[i], w[i + 1], w[i + 2], w[i + 3]}, b4[4] = {pc, pc, pc, pc};
                        u64 r4[4];
                        gl::mont_mul4(a4, b4, r4);
                        w[i] = r4[0], w[i + 1] = r4[1], w[i + 2] = r4[2], w[i + 3] = r4[3];
                    }
                }
            }
        };
        // (the overlapped order costs the coset-scaling instantiation 22 spilled VGPRs and 9 % -- profiles/r03_two_pass_ab.txt -- so
        //  that one keeps the two plain bursts)
        if constexpr (TF_PRE2_BURST16 == 2 && SCALE != 1) {
            const auto scale_half = [&](int h0) {  // coefficient j times offset^j for slots h0 .. h0+15 (as the SCALE 1 block below does for 32)
                if constexpr (SCALE == 1) {
                    if (A.pre_scale) {
                        const u64* ps = A.pre_scale + (long long)i1 * A.ps_i1;
#pragma unroll
                        for (int q0 = 0; q0 < 16; q0 += 8) {
                            u64 w[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const long long ur = (long long)(brev5(h0 + q0 + i) << p2);
                                const long long j = (ur + g) * A.ps_rs + (A.ps_col ? bcol : 0);
                                w[i] = ps[j < A.n_coeffs ? j : 0];
                            }
#pragma unroll
                            for (int i = 0; i < 8; i += 4) mul4_inplace(x, h0 + q0 + i, w[i], w[i + 1], w[i + 2], w[i + 3]);
                        }
                    }
                }
            };
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = ld(slot_off(q));
            u64 wa[8], wb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wa[i] = ld(slot_off(i) + poff);
#pragma unroll
            for (int i = 0; i < 8; ++i) wb[i] = ld(slot_off(8 + i) + poff);
#pragma unroll
            for (int q = 16; q < 32; ++q) x[q] = ld(slot_off(q));
            scale8(wa);
            scale8(wb);
            pre2_combine8<INV, 0>(x, wa, half != 0);
            pre2_combine8<INV, 8>(x, wb, half != 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) wa[i] = ld(slot_off(16 + i) + poff);
#pragma unroll
            for (int i = 0; i < 8; ++i) wb[i] = ld(slot_off(24 + i) + poff);
            __builtin_amdgcn_sched_barrier(0);
            // the first half goes all the way through levels 1-4 while the last 16 partners are in flight
            scale_half(0);
            if (half) pre2_shift<INV, 1, 16>(x);
            dit_half<INV, 0, LAZY1>(x);
            __builtin_amdgcn_sched_barrier(0);
            scale8(wa);
            scale8(wb);
            pre2_combine8<INV, 16>(x, wa, half != 0);
            pre2_combine8<INV, 24>(x, wb, half != 0);
            scale_half(16);
            if (half) pre2_shift<INV, 16, 32>(x);
        } else if constexpr (TF_PRE2_BURST16 != 0) {
#pragma unroll
        for (int h0 = 0; h0 < 32; h0 += 16) {
#pragma unroll
            for (int q = h0; q < h0 + 16; ++q) x[q] = ld(slot_off(q));
            u64 wa[8], wb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wa[i] = ld(slot_off(h0 + i) + poff);
#pragma unroll
            for (int i = 0; i < 8; ++i) wb[i] = ld(slot_off(h0 + 8 + i) + poff);
            scale8(wa);
            scale8(wb);
            if (h0 == 0) {
                pre2_combine8<INV, 0>(x, wa, half != 0);
                pre2_combine8<INV, 8>(x, wb, half != 0);
            } else {
                pre2_combine8<INV, 16>(x, wa, half != 0);
                pre2_combine8<INV, 24>(x, wb, half != 0);
            }
        }
        } else {
#pragma unroll
        for (int q = 0; q < 32; ++q) x[q] = ld(slot_off(q));
#pragma unroll
        for (int q0 = 0; q0 < 32; q0 += 8) {
            u64 w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = ld(slot_off(q0 + i) + poff);
            scale8(w);
            if (q0 == 0) pre2_combine8<INV, 0>(x, w, half != 0);
            else if (q0 == 8) pre2_combine8<INV, 8>(x, w, half != 0);
            else if (q0 == 16) pre2_combine8<INV, 16>(x, w, half != 0);
            else pre2_combine8<INV, 24>(x, w, half != 0);
        }
        }
    } else if (act_in) {
        const u32 toff = (u32)(((long long)ch_in * A.in_cs_hi + cl_in + (long long)g_in * A.in_rs) * 8);
        const char* base = reinterpret_cast<const char*>(in);
        if constexpr ((LAST1024 || COLP) && SCALE != 1) {
            const __amdgpu_buffer_rsrc_t ri = buf_rsrc(in);
#pragma unroll
            for (int q = 0; q < 32; ++q) x[q] = buf_load<LAST1024 ? TF_AUX_LAST_LOAD : TF_LOAD_AUX>(ri, toff, (u32)((long long)(brev5(q) << p2) * A.in_rs * 8));
        } else
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const long long ur = (long long)(brev5(q) << p2);  // uniform part of the row index
            const u64* ptr = reinterpret_cast<const u64*>(base + ur * A.in_rs * 8 + toff);
            if constexpr (SCALE == 1) {
                const long long j = (ur + g) * A.ps_rs + (A.ps_col ? bcol : 0);
                if (A.n_coeffs < 0 || j < A.n_coeffs) x[q] = *ptr;  // rows beyond the coefficients read as zero; scaled below
            } else {
                x[q] = (A.nt & 1) ? __builtin_nontemporal_load(ptr) : *ptr;  // uniform: the compiler emits the burst twice
            }
        }
    }
    if constexpr (SCALE == 1) {
        // coefficient j times offset^j: the 32 data loads above went out as one burst; the scale words follow eight at a
        // time and the products are taken in hand-scheduled pairs (zero rows stay zero: their scale index is clamped).
        // pre_scale == null with n_coeffs >= 0 is plain zero padding (fast_multiply); in2 != null multiplies by a second
        // operand laid out like the input (the pointwise product of fast_multiply, fused into the inverse transform's load).
        if (act_in && A.in2) {
            const char* base2 = reinterpret_cast<const char*>(in + (A.in2 - A.in));
            const u32 toff2 = (u32)(((long long)ch_in * A.in_cs_hi + cl_in + (long long)g_in * A.in_rs) * 8);
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {
                u64 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const long long ur = (long long)(brev5(q0 + i) << p2);
                    w[i] = *reinterpret_cast<const u64*>(base2 + ur * A.in_rs * 8 + toff2);
                }
#pragma unroll
                for (int i = 0; i < 8; i += 4) mul4_inplace(x, q0 + i, w[i], w[i + 1], w[i + 2], w[i + 3]);
            }
        }
        if (act_in && A.pre_scale) {
            const u64* ps = A.pre_scale + (long long)i1 * A.ps_i1;
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {
                u64 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const long long ur = (long long)(brev5(q0 + i) << p2);
                    const long long j = (ur + g) * A.ps_rs + (A.ps_col ? bcol : 0);
                    w[i] = ps[j < A.n_coeffs ? j : 0];
                }
#pragma unroll
                for (int i = 0; i < 8; i += 4) mul4_inplace(x, q0 + i, w[i], w[i + 1], w[i + 2], w[i + 3]);
            }
        }
    }
    if constexpr (PRE2 && !(TF_PRE2_BURST16 == 2 && SCALE != 1)) {
        if (half && act_in) pre2_shift<INV>(x);  // z_r = (x_r - x_{r+1024}) w_64^i; w_2048^g follows with the inner twiddle
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (MODE != 2 && !(PRE2 && TF_PRE2_BURST16 == 2 && SCALE != 1)) {  // (the overlapped PRE2 load above has done this already: zeros stay zeros)
        // Levels 1-4 of the first 16 slots need only the first 16 loads: start on them while the second half of the
        // burst is still in flight (the levels below 5 never mix the two halves).  Measured: 2.56 -> 2.44 ms.
        dit_half<INV, 0, LAZY1>(x);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MODE == 3) {
        stamp[1] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[2] = __builtin_readcyclecounter();
    }
    if constexpr (MODE != 2) {
        dit_half<INV, 16, LAZY1>(x);
        dit_level<INV, 5, LAZY1>(x);
        if ((LAST1024 || R1024) || A.inner_tw) {  // (the R = 1024 instantiations always have an inner table: launch_pass checks;
                                                  //  a lazy COL pass without one is R = 32: the inter-pass product follows directly)
            const u64* tw = LDS_TW ? ltw + g_in * kLdsTwStride : A.inner_tw + g_in * 32;
#pragma unroll
            for (int q = 0; q < 32; q += 4) mul4_inplace(x, q, tw[q], tw[q + 1], tw[q + 2], tw[q + 3]);
        }
    }
    if constexpr (MODE == 3) { asm volatile("" :: "v"(x[0]), "v"(x[31])); stamp[3] = __builtin_readcyclecounter(); }
    // ------------------------------------------------------------------ LDS exchange, `nrounds` rounds of `cpr` columns
    // Element (k1, g) of a column goes from the thread that owns row group g to the thread that owns k1 mod P2.
    // A thread writes its 32 values and reads its 32 new values in the SAME round, so only 32 are ever live.
    if constexpr (LAST1024) {
        // Writers are in the row-major roles (g_in, c_in), readers in the column-major roles (g, c); 8 columns per round.
        // Element (k1, g, cc) lives at k1 * 273 + cc * 34 + g: the active lanes of a writing lane group (consecutive g, one
        // column) cover consecutive words; the active lanes of a reading group (8 columns, same k1 -- and for the 32-lane
        // groups of ds_read_b64 two consecutive k1) fall on different 8-byte bank pairs because 34 = 2 and 273 = 17 (mod 32):
        // cc * 2 + 17 * (k1 & 1) takes 16 different values.  All 64 offsets are immediates on both sides.  (Round 1 used 36 /
        // 289; the tighter pitch leaves room for the staged twiddle table with two workgroups per CU.  kLast1024LdsBytes in
        // tf_hip.hip sizes the buffer.)
        constexpr int S1 = kL1024S1, CS = kL1024CS;
        const int wround = c_in >> 3, wcc = c_in & 7, rround = c >> 3, rcc = c & 7;
        u64* wr = lds + wcc * CS + g_in;
        const u64* rd = lds + g * S1 + rcc * CS;
        const int nr = (A.nc + 7) >> 3;
#pragma unroll 1
        for (int r = 0; r < nr; ++r) {
            if (r) __syncthreads();
            if (wround == r) {
#pragma unroll
                for (int q = 0; q < 32; ++q) wr[q * S1] = x[q];
            }
            __syncthreads();
            if (rround == r) {
#pragma unroll
                for (int q = 0; q < 32; ++q) x[q] = rd[brev5(q)];
            }
        }
    } else if (p2 != 0) {  // R = 32: step 1 is the whole transform and every element stays with its thread
        const int myround = c / cpr_;
        const int cc = c - myround * cpr_;
        u64* wr = lds + g * s2_ + cc * s3_;
        const u64* rd = lds + cc * s3_;
#pragma unroll 1
        for (int r = 0; r < nrounds_; ++r) {
            if (r) __syncthreads();
            if (myround == r) {
#pragma unroll
                for (int q = 0; q < 32; ++q) wr[q * s1_] = x[q];
            }
            __syncthreads();
            if (myround == r) {
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const int s = q >> p2;         // uniform
                    const int gbr = q & (P2 - 1);  // uniform
                    const int gg = p2 ? (int)(__brev((unsigned)gbr) >> (32 - p2)) : 0;
                    const int k1 = g + (s << p2);
                    x[q] = rd[k1 * s1_ + gg * s2_];
                }
            }
        }
    }
    if constexpr (MODE == 3) { asm volatile("" :: "v"(x[0]), "v"(x[31])); stamp[4] = __builtin_readcyclecounter(); }
    // ------------------------------------------------------------------ step 2 (radix P2 over g) + store
    __builtin_amdgcn_s_setprio(TF_PRIO_STEP2);
    if constexpr (LAST1024) {
        // R = 1024 and nothing to multiply: level 5 is fused with the stores, four butterflies (eight outputs) at a
        // time, so the store burst overlaps the end of the arithmetic.  Slot q holds output row k = g + 32 q.
        dit_half<INV, 0>(x);
        dit_half<INV, 16>(x);
        const u32 toff = (u32)(((long long)ch * A.out_cs_hi + cl + (long long)g * A.out_rs) * 8);
        char* base = reinterpret_cast<char*>(out);
        if constexpr (SCALE == 2) {
            // fast_coset_interpolate: output coefficient j times offset^-j; slot q holds element j0 + 32 q js_k (planner: n <= 2^28)
            long long j0 = (long long)i0 * A.js_i0 + (long long)i1 * A.js_i1 + (long long)(ch0 + ch) * A.js_c + (long long)g * A.js_k;
            if constexpr (PRE2) j0 += (long long)half * A.pre2_js_off;
            const char* sb = reinterpret_cast<const char*>(A.post_scale);
            const u32 soff = (u32)(j0 * 8);
            tail_p5<INV, 0, false, true>(x, act, base, toff, A.out_rs * 8, 32, sb, soff, A.js_k * 8);
            tail_p5<INV, 4, false, true>(x, act, base, toff, A.out_rs * 8, 32, sb, soff, A.js_k * 8);
            tail_p5<INV, 8, false, true>(x, act, base, toff, A.out_rs * 8, 32, sb, soff, A.js_k * 8);
            tail_p5<INV, 12, false, true>(x, act, base, toff, A.out_rs * 8, 32, sb, soff, A.js_k * 8);
            return;
        }
        if (A.n_out >= 0) {
            // truncated output (fast_multiply keeps the first n_out coefficients): slot q holds output element
            // j0 + 32 q js_k; the thread stores the slots below its own limit
            const long long j0 = (long long)i0 * A.js_i0 + (long long)i1 * A.js_i1 + (long long)(ch0 + ch) * A.js_c + (long long)g * A.js_k;
            const long long rem = A.n_out - j0, step = 32 * A.js_k;
            const int qlim = rem <= 0 ? 0 : (int)min(32ll, (rem + step - 1) / step);
            tail_p5<INV, 0, true>(x, act, base, toff, A.out_rs * 8, qlim);
            tail_p5<INV, 4, true>(x, act, base, toff, A.out_rs * 8, qlim);
            tail_p5<INV, 8, true>(x, act, base, toff, A.out_rs * 8, qlim);
            tail_p5<INV, 12, true>(x, act, base, toff, A.out_rs * 8, qlim);
            return;
        }
        const bool nts = (A.nt & 2) != 0;
        tail_p5<INV, 0>(x, act, base, toff, A.out_rs * 8, 32, nullptr, 0, 0, nts);
        tail_p5<INV, 4>(x, act, base, toff, A.out_rs * 8, 32, nullptr, 0, 0, nts);
        tail_p5<INV, 8>(x, act, base, toff, A.out_rs * 8, 32, nullptr, 0, 0, nts);
        tail_p5<INV, 12>(x, act, base, toff, A.out_rs * 8, 32, nullptr, 0, 0, nts);
        return;
    }
    if constexpr (MODE != 2) {
        if (p2 >= 1) dit_level<INV, 1, LAZY2>(x);
        if (p2 >= 2) dit_level<INV, 2, LAZY2>(x);
        if (p2 >= 3) dit_level<INV, 3, LAZY2>(x);
        if (p2 >= 4) dit_level<INV, 4, LAZY2>(x);
        if (p2 >= 5) dit_level<INV, 5, LAZY2>(x);
    }
    if constexpr (MODE == 1) {
        u64 acc = 0;
#pragma unroll
        for (int q = 0; q < 32; ++q) acc ^= x[q];
        if (acc == 0x123456789abcdefULL) out[t] = acc;  // keeps the arithmetic live; never true in practice
    } else if (act) {
        const u32 toff = (u32)(((long long)ch * A.out_cs_hi + cl + (long long)g * A.out_rs) * 8);
        char* base = reinterpret_cast<char*>(out);
        if ((COLP && MODE == 0) || (MODE != 2 && A.post_tw)) {  // (R1024 / COL are only launched with an inter-pass table)
            // inter-pass twiddle: 8 table words at a time (bounded register footprint), multiply, store
            const u32 twoff = (u32)(((long long)g * A.tw_rs + bcol) * 8);
            const u64* ptw = A.post_tw;
            if constexpr (PRE2) ptw += (long long)half * A.pre2_tw_off;  // the odd half's rows of the inter-pass table
            const char* tbase = reinterpret_cast<const char*>(ptw);
            const __amdgpu_buffer_rsrc_t rt = buf_rsrc(ptw), ro = buf_rsrc(out);  // used by the R1024 / COL instantiations only
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {
                u64 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = q0 + i;
                    const long long uk = (long long)(((q >> p2) << p2) + ((q & (P2 - 1)) << 5));  // uniform part of k
                    if constexpr (COLP) w[i] = buf_load_tab(rt, twoff, (u32)(uk * A.tw_rs * 8));
                    else w[i] = *reinterpret_cast<const u64*>(tbase + uk * A.tw_rs * 8 + twoff);
                }
#pragma unroll
                for (int i = 0; i < 8; i += 4) {
                    const int q = q0 + i;
                    const u64 a4[4] = {x[q], x[q + 1], x[q + 2], x[q + 3]}, b4[4] = {w[i], w[i + 1], w[i + 2], w[i + 3]};
                    u64 r4[4];
                    gl::mont_mul4(a4, b4, r4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const long long uk = (long long)((((q + e) >> p2) << p2) + (((q + e) & (P2 - 1)) << 5));  // uniform part of k
                        if constexpr (COLP) buf_store<TF_AUX_COL_STORE>(ro, toff, (u32)(uk * A.out_rs * 8), r4[e]);
                        else *reinterpret_cast<u64*>(base + uk * A.out_rs * 8 + toff) = r4[e];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (SCALE == 2) {
            const long long j0 = (long long)i0 * A.js_i0 + (long long)i1 * A.js_i1 + (wt ? (long long)(ch0 + ch) * A.js_c : (long long)i2 * A.js_i2 + (long long)ch * A.js_c) +
                                 (long long)g * A.js_k;
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {
                u64 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = q0 + i;
                    const long long uk = (long long)(((q >> p2) << p2) + ((q & (P2 - 1)) << 5));
                    w[i] = A.post_scale[j0 + uk * A.js_k];
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const int q = q0 + i;
                    const long long uk0 = (long long)(((q >> p2) << p2) + ((q & (P2 - 1)) << 5));
                    const long long uk1 = (long long)((((q + 1) >> p2) << p2) + (((q + 1) & (P2 - 1)) << 5));
                    u64 r0, r1;
                    gl::mont_mul2(x[q], w[i], x[q + 1], w[i + 1], r0, r1);
                    *reinterpret_cast<u64*>(base + uk0 * A.out_rs * 8 + toff) = r0;
                    *reinterpret_cast<u64*>(base + uk1 * A.out_rs * 8 + toff) = r1;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const long long uk = (long long)(((q >> p2) << p2) + ((q & (P2 - 1)) << 5));
                *reinterpret_cast<u64*>(base + uk * A.out_rs * 8 + toff) = x[q];
            }
        }
    }
    if constexpr (MODE == 3) {
        stamp[5] = __builtin_readcyclecounter();
        if (A.dbg && (t & 63) == 0 && blockIdx.x < 4096) {
            unsigned long long* d = A.dbg + ((size_t)blockIdx.x * 8 + (t >> 6)) * 6;
            for (int i = 0; i < 6; ++i) d[i] = stamp[i];
        }
    }
}

// ---- the R = 1024 column pass with the NEXT tile's loads issued inside the store phase -----------------------------------
// ntt_pass_kernel's R1024 instantiation, plain transform, as a loop over `tiles_per_wg` tiles (tile, tile + gridDim.x, ...): the
// eight registers a store group frees are filled at once with the next tile's loads, so a workgroup's load latency (a fifth of a
// wave's lifetime in the one-tile kernel) runs under its own stores and under the partner workgroup's arithmetic instead of in
// front of its first butterfly.  Few tiles per workgroup keep the dispatcher's dynamic balancing (a fully static assignment
// measured slower in round 1).  Same arithmetic, same words.  Selected by TF_NTT_PERSIST = tiles per workgroup (A/B).
template <bool INV>
__global__ void __launch_bounds__(512, TF_NTT_WAVES) ntt_col1024_chain_kernel(const NttPassArgs A, u32 total_tiles, u32 tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int t = threadIdx.x, g = t >> 4, c = t & 15;
    const int L = A.L;
    u64* const ltw = lds + 32 * kR1024S1;
    for (int i = t; i < 1024; i += 512) ltw[(i >> 5) * kLdsTwStride + (i & 31)] = A.inner_tw[i];
    __syncthreads();
    const int ch = (int)div_by_L((u32)c, L), cl = c - ch * L;
    // per-tile quantities: input / output bases, the element column of my word-column, whether my column exists
    struct Tile {
        const u64* in;
        u64* out;
        long long bcol;
        bool act;
    };
    const auto decode = [&](u32 bid) {
        u32 i0, i1, i2;
        if (A.xcd_order) {
            const u32 G = (u32)A.xcd_order, xcd = bid & 7u, slot = bid >> 3, ngrp = A.d2 / (8u * G);
            const u32 grp = A.xcd_colfast ? slot % ngrp : slot / (G * A.d01);
            const u32 within = A.xcd_colfast ? slot / ngrp : slot % (G * A.d01);
            i2 = ((grp << 3) | xcd) * G + within % G;
            const u32 rest = within / G;
            i1 = rest % A.d1;
            i0 = rest / A.d1;
        } else {
            i2 = bid % A.d2;
            const u32 rest = bid / A.d2;
            i1 = rest % A.d1;
            i0 = rest / A.d1;
        }
        Tile T;
        T.in = A.in + (long long)i0 * A.ib0 + (long long)i1 * A.ib1 + (long long)i2 * A.ib2;
        T.out = A.out + (long long)i0 * A.ob0 + (long long)i1 * A.ob1 + (long long)i2 * A.ob2;
        const int col0 = (int)i2 * kR1024Nc;
        T.act = c < min(kR1024Nc, A.col_limit - col0);
        T.bcol = (long long)div_by_L((u32)(col0 + c), L);
        return T;
    };
    const u32 toff_in = (u32)(((long long)ch * A.in_cs_hi + cl + (long long)g * A.in_rs) * 8);
    const u32 toff_out = (u32)(((long long)ch * A.out_cs_hi + cl + (long long)g * A.out_rs) * 8);
    const int myround = c >> 3, cc = c & 7;
    u64* const wr = lds + g * kR1024Cpr + cc;
    const u64* const rd = lds + cc + g * kR1024S1;
    u64 x[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = 0;
    u32 bid = blockIdx.x;
    Tile cur = decode(bid);
    if (cur.act) {
        const __amdgpu_buffer_rsrc_t ri = buf_rsrc(cur.in);
#pragma unroll
        for (int q = 0; q < 32; ++q) x[q] = buf_load<TF_LOAD_AUX>(ri, toff_in, (u32)((long long)(brev5(q) << 5) * A.in_rs * 8));
    }
#pragma unroll 1
    for (u32 k = 0;; ++k) {
        // ---- step 1
        dit_half<INV, 0, true>(x);
        __builtin_amdgcn_sched_barrier(0);
        dit_half<INV, 16, true>(x);
        dit_level<INV, 5, true>(x);
        {
            const u64* tw = ltw + g * kLdsTwStride;
#pragma unroll
            for (int q = 0; q < 32; q += 4) mul4_inplace(x, q, tw[q], tw[q + 1], tw[q + 2], tw[q + 3]);
        }
        // ---- exchange (two rounds of eight columns); the barrier in front also separates it from the previous tile's reads
        if (k) __syncthreads();
#pragma unroll 1
        for (int r = 0; r < kR1024Rounds; ++r) {
            if (r) __syncthreads();
            if (myround == r) {
#pragma unroll
                for (int q = 0; q < 32; ++q) wr[q * kR1024S1] = x[q];
            }
            __syncthreads();
            if (myround == r) {
#pragma unroll
                for (int q = 0; q < 32; ++q) x[q] = rd[brev5(q) * kR1024Cpr];
            }
        }
        // ---- step 2
        dit_level<INV, 1, true>(x);
        dit_level<INV, 2, true>(x);
        dit_level<INV, 3, true>(x);
        dit_level<INV, 4, true>(x);
        dit_level<INV, 5, true>(x);
        // ---- inter-pass twiddles, stores, and the next tile's loads into the registers the stores free
        const u32 nbid = bid + gridDim.x;
        const bool has_next = k + 1 < tiles_per_wg && nbid < total_tiles;  // uniform
        Tile nxt = cur;
        if (has_next) nxt = decode(nbid);
        if (cur.act) {
            const u32 twoff = (u32)(((long long)g * A.tw_rs + cur.bcol) * 8);
            const __amdgpu_buffer_rsrc_t rt = buf_rsrc(A.post_tw), ro = buf_rsrc(cur.out), rn = buf_rsrc(nxt.in);
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {
                u64 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = buf_load_tab(rt, twoff, (u32)((long long)((q0 + i) << 5) * A.tw_rs * 8));
#pragma unroll
                for (int i = 0; i < 8; i += 4) {
                    const int q = q0 + i;
                    const u64 a4[4] = {x[q], x[q + 1], x[q + 2], x[q + 3]}, b4[4] = {w[i], w[i + 1], w[i + 2], w[i + 3]};
                    u64 r4[4];
                    gl::mont_mul4(a4, b4, r4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) buf_store<TF_AUX_COL_STORE>(ro, toff_out, (u32)((long long)((q + e) << 5) * A.out_rs * 8), r4[e]);
                }
                if (has_next && nxt.act) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[q0 + i] = buf_load<TF_LOAD_AUX>(rn, toff_in, (u32)((long long)(brev5(q0 + i) << 5) * A.in_rs * 8));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (has_next && nxt.act) {
            const __amdgpu_buffer_rsrc_t rn = buf_rsrc(nxt.in);
#pragma unroll
            for (int q = 0; q < 32; ++q) x[q] = buf_load<TF_LOAD_AUX>(rn, toff_in, (u32)((long long)(brev5(q) << 5) * A.in_rs * 8));
        }
        if (!has_next) break;
        if (!nxt.act) {
#pragma unroll
            for (int q = 0; q < 32; ++q) x[q] = 0;
        }
        bid = nbid;
        cur = nxt;
    }
}

// ---- 2^11 <= n <= 2^14, contiguous BFieldElement transforms: the WHOLE transform in one workgroup pass --------------
// 16 384 elements are exactly one 512-thread tile, so n = 32 * 32 * P3 (P3 = 2 .. 16) runs as three register stages joined by
// two LDS exchanges and touches HBM once instead of twice:
//   stage A  thread (tr, rest = j2 P3 + j3): radix 32 over j1 (loads at stride n/32, lanes along `rest`: contiguous),
//            times w_n^(k1 rest);
//   exchange 1 (two rounds, by (tr, j3) pair);  stage B  thread (tr, j3, k1): radix 32 over j2, times w_{32 P3}^(k2 j3);
//   exchange 2 (two rounds, by k1 half);        stage C  thread (tr, s, k1): radix P3 over j3 for its 32 / P3 values of k2;
//   store X[k1 + 32 k2 + 1024 k3], lanes along k1: contiguous.
// Both exchange layouts give every half-wave distinct 8-byte bank pairs (strides = 1 mod 32 between the lanes of a role).
struct NttBlockArgs {
    const u64* in;
    u64* out;
    const u64* tw1;   // [32][n / 32]: w_n^(k1 * rest)            (inverse: w^-1)
    const u64* tw2;   // [32][P3]:     w_{32 P3}^(k2 * j3) (* n^-1 for the inverse)
    const u64* pre_scale;   // or null: coefficient j times pre_scale[j] on load (fast_coset_evaluate)
    const u64* post_scale;  // or null: output element k times post_scale[k] on store (fast_coset_interpolate)
    long long n_coeffs;     // < 0: none; else elements j >= n_coeffs read as zero
    long long in_bs, out_bs;  // words between consecutive transforms
    long long total_transforms;
    const u64* in2;         // SCALE 3: second operand laid out like `in`, multiplied in on load (fast_multiply)
    long long n_out;        // SCALE 3: >= 0 = only output elements k < n_out are stored
    int L;                  // words per element.  L = 3 (XFieldElement): the three limbs of an element are three independent
                            // transforms with element stride 3 (ntt.rs:203-207); total_transforms then counts LIMB transforms
                            // (3 per XFieldElement slice) and workgroup slots take them in order.  SCALE 3 is L = 1 only.
};

// SCALE: 0 plain, 1 padding / pre-scale on load (forward), 2 post-scale on store (inverse), 3 pointwise product with a second
// operand on load and truncated store (the inverse transform of fast_multiply) -- separate instantiations so that the plain
// transform keeps its register budget
template <int LOGP3, bool INV, int SCALE = 0>
__global__ void __launch_bounds__(512, 4) ntt_block_kernel(const NttBlockArgs A) {
    constexpr int P3 = 1 << LOGP3, N = 1024 << LOGP3, REST = 32 << LOGP3, T = 16 >> LOGP3;
    constexpr int PS1 = 1056 + 32 / P3;      // exchange 1: pair slot stride (k1 * 33 + j2 inside a slot)
    constexpr int KS2 = 32 * P3 + 1;         // exchange 2: stride between k1 (k2 * P3 + j3 inside)
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int t = threadIdx.x;
    const long long tr0 = (long long)blockIdx.x * T;
    const int nt = (int)min((long long)T, A.total_transforms - tr0);
    u64 x[32];
    // ---- stage A
    const int trA = t / REST, rest = t - trA * REST;
    const bool actA = trA < nt;
    const int es = A.L;  // element stride in words
    {
        const long long ltA = tr0 + trA, slA = ltA / es;  // limb transform -> (slice, limb)
        const u64* src = A.in + slA * A.in_bs + (ltA - slA * es) + (long long)rest * es;
        const long long lim = (SCALE != 1 || A.n_coeffs < 0) ? (long long)N : A.n_coeffs;
#pragma unroll
        for (int q = 0; q < 32; ++q) x[q] = (actA && (SCALE != 1 || brev5(q) * REST + rest < lim)) ? src[(long long)brev5(q) * REST * es] : 0;
        if (SCALE == 3) {
            const u64* src2 = A.in2 + (tr0 + trA) * A.in_bs + rest;  // (SCALE 3 runs with L = 1)
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {
                u64 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = actA ? src2[(long long)brev5(q0 + i) * REST] : 0;
#pragma unroll
                for (int i = 0; i < 8; i += 4) mul4_inplace(x, q0 + i, w[i], w[i + 1], w[i + 2], w[i + 3]);
            }
        }
        if (SCALE == 1 && A.pre_scale) {
            const u64* ps = A.pre_scale + rest;
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {
                u64 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const long long j = brev5(q0 + i) * REST + rest;
                    w[i] = ps[j < lim ? brev5(q0 + i) * REST : -rest];  // clamp to entry 0 beyond the coefficients (value unused: x is 0)
                }
#pragma unroll
                for (int i = 0; i < 8; i += 4) mul4_inplace(x, q0 + i, w[i], w[i + 1], w[i + 2], w[i + 3]);
            }
        }
    }
    dit_half<INV, 0, true>(x);   // lazy networks in stages A and B: a Montgomery product follows (see ntt_pass_kernel)
    dit_half<INV, 16, true>(x);
    dit_level<INV, 5, true>(x);
    {
        const u64* tw = A.tw1 + rest;
#pragma unroll
        for (int q = 4; q < 32; q += 4) mul4_inplace(x, q, tw[q * REST], tw[(q + 1) * REST], tw[(q + 2) * REST], tw[(q + 3) * REST]);
        gl::mont_mul2(x[2], tw[2 * REST], x[3], tw[3 * REST], x[2], x[3]);
        x[1] = gl::mont_mul(x[1], tw[REST]);  // k1 = 0: factor 1
#if TF_LAZY
        x[0] = gl::add(x[0], 0);              // ... so the word is only made canonical
#endif
    }
    // ---- exchange 1: (k1 = q, j2, j3, tr) -> thread (tr, j3, k1) holding j2
    const int j2A = rest >> LOGP3, j3A = rest & (P3 - 1);
    const int pairA = trA * P3 + j3A;                       // 0 .. 15
    // A thread writes its 32 old values and reads its 32 new ones in the SAME round (only 32 are ever live), so both roles
    // must fall into the same round.  The round is pair >> 3: for P3 <= 8 that is a function of tr alone; for P3 = 16
    // (one transform, 16 pairs = j3) the stage-B role is taken from the thread index with bits 3 and 8 swapped, which makes
    // its j3 >> 3 equal to the stage-A role's.
    const int tb = (LOGP3 == 4) ? ((t & ~0x108) | ((t & 8) << 5) | ((t >> 5) & 8)) : t;
    const int trB = tb / REST, j3B = (tb >> 5) & (P3 - 1), k1B = tb & 31;
    const int pairB = trB * P3 + j3B;
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        if (r) __syncthreads();
        if ((pairA >> 3) == r) {
            u64* wr = lds + (pairA & 7) * PS1 + j2A;
#pragma unroll
            for (int q = 0; q < 32; ++q) wr[q * 33] = x[q];
        }
        __syncthreads();
        if ((pairB >> 3) == r) {
            const u64* rd = lds + (pairB & 7) * PS1 + k1B * 33;
#pragma unroll
            for (int q = 0; q < 32; ++q) x[q] = rd[brev5(q)];
        }
    }
    // ---- stage B
    dit_half<INV, 0, true>(x);
    dit_half<INV, 16, true>(x);
    dit_level<INV, 5, true>(x);
    {
        const u64* tw = A.tw2 + j3B;
#pragma unroll
        for (int q = 0; q < 32; q += 4) mul4_inplace(x, q, tw[q * P3], tw[(q + 1) * P3], tw[(q + 2) * P3], tw[(q + 3) * P3]);
    }
    // ---- exchange 2: (k1, k2 = q, j3, tr) -> thread (tr, s, k1) holding k2 in [s * 32 / P3, ..) x all j3
    const int trC = trB, sC = j3B, k1C = k1B;  // same thread index decomposition, s takes j3's bit positions
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        if (r) __syncthreads();
        if ((k1B >> 4) == r) {
            u64* wr = lds + (trB * 16 + (k1B & 15)) * KS2 + j3B;
#pragma unroll
            for (int q = 0; q < 32; ++q) wr[q * P3] = x[q];
        }
        __syncthreads();
        if ((k1C >> 4) == r) {
            const u64* rd = lds + (trC * 16 + (k1C & 15)) * KS2 + sC * (32 / P3) * P3;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int grp = q >> LOGP3, rr = q & (P3 - 1);
                const int j3 = (int)(__brev((unsigned)rr) >> (32 - LOGP3));
                x[q] = rd[grp * P3 + j3];
            }
        }
    }
    // ---- stage C: radix P3 inside groups of P3 slots
    if constexpr (LOGP3 >= 1) dit_level<INV, 1>(x);
    if constexpr (LOGP3 >= 2) dit_level<INV, 2>(x);
    if constexpr (LOGP3 >= 3) dit_level<INV, 3>(x);
    if constexpr (LOGP3 >= 4) dit_level<INV, 4>(x);
    if (trC < nt) {
        const long long ltC = tr0 + trC, slC = ltC / es;
        u64* dst = A.out + slC * A.out_bs + (ltC - slC * es) + (long long)k1C * es;
        if (SCALE == 2 && A.post_scale) {
            // scale and store eight outputs at a time (bounded register footprint)
            const u64* ps = A.post_scale + k1C;
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {
                u64 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = q0 + i, grp = q >> LOGP3, k3 = q & (P3 - 1);
                    w[i] = ps[32 * (sC * (32 / P3) + grp) + 1024 * k3];
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const int qa = q0 + i, qb = q0 + i + 1;
                    u64 r0, r1;
                    gl::mont_mul2(x[qa], w[i], x[qb], w[i + 1], r0, r1);
                    dst[(long long)(32 * (sC * (32 / P3) + (qa >> LOGP3)) + 1024 * (qa & (P3 - 1))) * es] = r0;
                    dst[(long long)(32 * (sC * (32 / P3) + (qb >> LOGP3)) + 1024 * (qb & (P3 - 1))) * es] = r1;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const long long klim = (SCALE == 3 && A.n_out >= 0) ? A.n_out : (long long)N;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int grp = q >> LOGP3, k3 = q & (P3 - 1);
                const int k2 = sC * (32 / P3) + grp;
                if (SCALE != 3 || k1C + 32 * k2 + 1024 * k3 < klim) dst[(long long)(32 * k2 + 1024 * k3) * es] = x[q];
            }
        }
    }
}

// ---- n = 32, contiguous transforms: one transform (one limb of it for XFE) per thread, staged through LDS ---------------
// A thread's 32 elements are contiguous in memory, so direct loads are 8-byte pieces 256 bytes apart (1.94 ms per 2^28 words).
// Here the workgroup streams its tile (512 BFE transforms or 170 XFE transforms = 510 limb-transforms) through LDS in two
// halves: coalesced loads into rows of pitch 33, each thread picks up its row, transforms it in registers, puts it back, and
// the tile leaves with coalesced stores.
struct NttRows32Args {
    const u64* in;
    u64* out;
    long long total_transforms;  // transforms (XFE counts as one)
    u64 scale;                   // Montgomery 32^-1 for the inverse, 0 = none
    int L;
};

template <bool INV>
__global__ void __launch_bounds__(512, 4) ntt_rows32_kernel(const NttRows32Args A) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int t = threadIdx.x, L = A.L;
    const int per_tile = L == 1 ? 512 : 170;                     // transforms per workgroup
    const long long tr0 = (long long)blockIdx.x * per_tile;
    const int ntr = (int)min((long long)per_tile, A.total_transforms - tr0);
    const int nlt = ntr * L;                                     // limb-transforms (threads with work)
    const int half_lt = L == 1 ? 256 : 255;                      // first half: limb-transforms [0, half_lt)
    const u64* src = A.in + tr0 * 32 * L;
    u64* dst = A.out + tr0 * 32 * L;
    const int words = nlt * 32;
    const int split = min(words, half_lt * 32);                  // words of the first half (whole transforms)
    u64 x[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = 0;
    const int myhalf = t >= half_lt ? 1 : 0;
    const int row = t - myhalf * half_lt;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int w0 = h ? split : 0, w1 = h ? words : split;
        if (h) __syncthreads();
        for (int w = w0 + t; w < w1; w += 512) {
            int lt, e;
            if (L == 1) {
                lt = w >> 5;
                e = w & 31;
            } else {
                const int el = (int)__umulhi((u32)w, 0x55555556u), limb = w - 3 * el;
                lt = (el >> 5) * 3 + limb;
                e = el & 31;
            }
            lds[(lt - h * half_lt) * 33 + e] = src[w];
        }
        __syncthreads();
        if (myhalf == h && t < nlt) {
#pragma unroll
            for (int q = 0; q < 32; ++q) x[q] = lds[row * 33 + brev5(q)];
        }
    }
    dit_half<INV, 0, INV>(x);   // the inverse multiplies every word by 32^-1 afterwards: lazy network
    dit_half<INV, 16, INV>(x);
    dit_level<INV, 5, INV>(x);
    if (INV) {
#pragma unroll
        for (int q = 0; q < 32; q += 4) mul4_inplace(x, q, A.scale, A.scale, A.scale, A.scale);
    }
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int w0 = h ? split : 0, w1 = h ? words : split;
        __syncthreads();
        if (myhalf == h && t < nlt) {
#pragma unroll
            for (int q = 0; q < 32; ++q) lds[row * 33 + q] = x[q];
        }
        __syncthreads();
        for (int w = w0 + t; w < w1; w += 512) {
            int lt, e;
            if (L == 1) {
                lt = w >> 5;
                e = w & 31;
            } else {
                const int el = (int)__umulhi((u32)w, 0x55555556u), limb = w - 3 * el;
                lt = (el >> 5) * 3 + limb;
                e = el & 31;
            }
            dst[w] = lds[(lt - h * half_lt) * 33 + e];
        }
    }
}

// ---- 64 <= n <= 4096, LITTLE work per call: the latency-shaped transform -----------------------------------------------
// The pass kernels above give a thread 32 elements: one thread's program is ~3 500 dependent-ish instructions, 15-25 us however
// few transforms a call holds (a tree walk over 2^12 points, one slice of a caller that transforms one polynomial at a time).
// When a call cannot fill the chip anyway this kernel spends threads instead: EIGHT elements per thread, n / 8 threads per
// transform, Stockham autosort stages of radix 8 (shift-only networks, as everywhere: w_8 = 2^24) joined through LDS, one
// general twiddle per element and stage from a table w_n^e -- three or four short steps instead of one long one.
//   stage (radix R, Ns = product of the radices before it), butterfly unit u < n / R:   k = u mod Ns,
//     v[r] = in[u + r n / R] w_{Ns R}^(k r),   V = DFT_R(v),   out[(u / Ns) Ns R + k + r Ns] = V[r]
// (natural order in and out, no bit reversal).  The last stage has radix 8, 4 or 2 (8 / R units per thread).
// XFieldElement slices are three limb transforms of element stride 3 (ntt.rs:203-207).
#ifndef TF_LAT_MUL4
#define TF_LAT_MUL4 1  // 0 (A/B build): the stage twiddles of the latency-shaped kernels as eight single products
#endif
struct NttLatArgs {
    const u64* in;
    u64* out;
    const u64* in2;        // or null: second operand laid out like `in`, multiplied in on load (L = 1 only)
    const u64* tw;         // [2][n]: w_n^(+-e), then n^-1 w_n^(+-e) (the inverse's last stage)
    long long n_coeffs;    // < 0: none; else elements >= n_coeffs read as zero
    long long in_bs, out_bs;  // words between consecutive slices
    long long total;       // limb transforms = batch * L
    u64 ninv;              // Montgomery n^-1 (inverse only)
    int L;
    // ---- the steps of a zerofier-tree walk that used to be kernels of their own, as modifiers of this kernel's load and store
    // (math/zerofier_tree.rs / polynomial.rs:1882-1894 remaindering; the tree code in tf_hip.hip says which step is which)
    int load_mode;         // 0: element idx of slice b is in[b * in_bs + idx * L]
                           // 1: REVERSED: in[(b >> src_shift) * in_bs + (rev_top - idx) * L] for idx < n_coeffs (poly_reverse /
                           //    remainder_rev_high fused into the forward transform that follows them)
                           // 2: (L = 1) the interpolation walk's parent N_l (Z_r + s) + N_r (Z_l + s), s = (-1)^idx, from the children's
                           //    transforms in[2 b], in[2 b + 1] and the level's cached tail transforms th[2 node], th[2 node + 1],
                           //    node = b % parents (interpolant_pointwise_kernel fused into the inverse transform that follows it)
    int src_shift;
    long long rev_top;
    const u64* th;
    long long parents;
    int store_mode;        // 0: all n outputs; 1: only outputs k < keep, stored as  sub_src[(b >> 1) * sub_bs + k * L] - value
                           //    (remainder_finish_kernel fused into the inverse transform in front of it: r = f_low - (q * tail)_low)
    const u64* sub_src;
    long long sub_bs, keep;
};
__host__ __device__ __forceinline__ constexpr int lat_pad(int i) { return i + (i >> 3); }
template <int LOGR>
__device__ __forceinline__ constexpr int lat_brev(int r) {
    int o = 0;
    for (int b = 0; b < LOGR; ++b) o |= ((r >> b) & 1) << (LOGR - 1 - b);
    return o;
}
template <bool INV, int LOGR>
__device__ __forceinline__ void lat_dft(u64 (&x)[32]) {  // 8 >> LOGR independent DFTs of 2^LOGR points on slots 0 .. 7 (bit-reversed in, natural out)
    DitRange<INV, 1, 0, 4, false>::run(x);
    if constexpr (LOGR >= 2) DitRange<INV, 2, 0, 4, false>::run(x);
    if constexpr (LOGR >= 3) DitRange<INV, 3, 0, 4, false>::run(x);
}

template <int LOGN, bool INV>
__global__ void __launch_bounds__(LOGN == 12 ? 512 : 256) ntt_lat_kernel(const NttLatArgs A) {
    constexpr int N = 1 << LOGN, TPT = N / 8, WG = LOGN == 12 ? 512 : 256, T = WG / TPT;
    constexpr int S = (LOGN + 2) / 3;                 // stages; the first S - 1 have radix 8
    constexpr int LOGRL = LOGN - 3 * (S - 1);         // log2 of the last radix (1 .. 3)
    constexpr int BUF = lat_pad(N * T) + 8;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];  // two buffers of BUF words
    const int t = threadIdx.x, tr = t / TPT, j = t - tr * TPT;
    const long long gtr = (long long)blockIdx.x * T + tr;
    const bool act = gtr < A.total;
    const int L = A.L;
    const long long b = act ? gtr / L : 0;
    const int limb = act ? (int)(gtr - b * L) : 0;
    const u64* src = A.in + b * A.in_bs + limb;
    u64* dst = A.out + b * A.out_bs + limb;
    u64 x[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = 0;
    // the general twiddles of every stage after the first, requested before anything else: w_{Ns R}^(k r) = w_n^(k r n / (Ns R))
    u64 tw[S > 1 ? S - 1 : 1][8];
    {
        int Ns = 8;
#pragma unroll
        for (int s = 1; s < S; ++s) {
            const int logr = s + 1 < S ? 3 : LOGRL, R = 1 << logr, U = 8 >> logr;
            const bool last = s + 1 == S;
#pragma unroll
            for (int a = 0; a < U; ++a) {
                const int u = j + a * TPT, k = u & (Ns - 1);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = k * r * (N / (Ns * R));
                    tw[s - 1][a * R + r] = (r == 0) ? 0 : A.tw[((INV && last) ? N : 0) + e];
                }
            }
            Ns *= R;
        }
    }
    // ---- stage 1: from global memory, no twiddles
    if (act) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = j + r * TPT;
            u64 v = 0;
            if (A.load_mode == 1) {
                if (idx < A.n_coeffs) v = (A.in + (b >> A.src_shift) * A.in_bs + limb)[(A.rev_top - idx) * L];
            } else if (A.load_mode == 2) {
                const u64* c0 = A.in + 2 * b * A.in_bs;
                const u64* t0 = A.th + 2 * (b % A.parents) * A.in_bs;
                const u64 sgn = (idx & 1) ? gl::P - gl::ONE : gl::ONE;
                const u64 zl = gl::add(t0[idx], sgn), zr = gl::add(t0[A.in_bs + idx], sgn);
                v = gl::add(gl::mont_mul(c0[idx], zr), gl::mont_mul(c0[A.in_bs + idx], zl));
            } else if (A.n_coeffs < 0 || idx < A.n_coeffs) {
                v = src[(long long)idx * L];
                if (A.in2) v = gl::mont_mul(v, (A.in2 + b * A.in_bs + limb)[(long long)idx * L]);
            }
            x[lat_brev<3>(r)] = v;
        }
    }
    lat_dft<INV, 3>(x);
    u64* bufs[2] = {lds + 0, lds + BUF};
    {
        u64* o = bufs[0] + 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[lat_pad(tr * N + j * 8 + r)] = x[r];
    }
    __syncthreads();
    int Ns = 8;
#pragma unroll
    for (int s = 1; s < S; ++s) {
        const bool last = s + 1 == S;
        const int logr = last ? LOGRL : 3, R = 1 << logr, U = 8 >> logr;
        const u64* in = bufs[(s - 1) & 1];
        u64* o = bufs[s & 1];
        u64 v[8];
#pragma unroll
        for (int a = 0; a < U; ++a) {
            const int u = j + a * TPT;
#pragma unroll
            for (int r = 0; r < R; ++r) v[a * R + r] = in[lat_pad(tr * N + u + r * (N / R))];
        }
#if TF_LAT_MUL4
        if (logr == 3) {  // (see lat_xform: two blocks of four interleaved products)
            u64 a0[4] = {v[1], v[2], v[3], v[4]}, b0[4] = {tw[s - 1][1], tw[s - 1][2], tw[s - 1][3], tw[s - 1][4]}, r0[4];
            u64 a1[4] = {v[5], v[6], v[7], v[0]}, b1[4] = {tw[s - 1][5], tw[s - 1][6], tw[s - 1][7], (INV && last) ? A.ninv : gl::ONE}, r1[4];
            gl::mont_mul4(a0, b0, r0);
            gl::mont_mul4(a1, b1, r1);
            x[lat_brev<3>(0)] = (INV && last) ? r1[3] : v[0];
            x[lat_brev<3>(1)] = r0[0], x[lat_brev<3>(2)] = r0[1], x[lat_brev<3>(3)] = r0[2], x[lat_brev<3>(4)] = r0[3];
            x[lat_brev<3>(5)] = r1[0], x[lat_brev<3>(6)] = r1[1], x[lat_brev<3>(7)] = r1[2];
        } else
#endif
#pragma unroll
        for (int a = 0; a < U; ++a) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                u64 w = v[a * R + r];
                if (r) w = gl::mont_mul(w, tw[s - 1][a * R + r]);
                else if (INV && last) w = gl::mont_mul(w, A.ninv);
                const int slot = a * R + (logr == 3 ? lat_brev<3>(r) : (logr == 2 ? lat_brev<2>(r) : r));
                x[slot] = w;
            }
        }
        if (logr == 3) lat_dft<INV, 3>(x);
        else if (logr == 2) lat_dft<INV, 2>(x);
        else lat_dft<INV, 1>(x);
#pragma unroll
        for (int a = 0; a < U; ++a) {
            const int u = j + a * TPT, k = u & (Ns - 1), j0 = (u - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int idx = j0 + r * Ns;
                if (last) {
                    if (act) {
                        if (A.store_mode == 1) {
                            if (idx < A.keep) dst[(long long)idx * L] = gl::sub((A.sub_src + (b >> 1) * A.sub_bs + limb)[(long long)idx * L], x[a * R + r]);
                        } else {
                            dst[(long long)idx * L] = x[a * R + r];
                        }
                    }
                } else {
                    o[lat_pad(tr * N + idx)] = x[a * R + r];
                }
            }
        }
        if (!last) __syncthreads();
        Ns *= R;
    }
}

// ---- a whole LEVEL of a zerofier-tree walk in one launch (BFieldElement, 2d <= 4096, the latency regime) -------------------
// A small walk is a chain of dependent launches, each ~4 us of dispatch + drain around ~2 us of work: the walk down spends four
// transforms per level, the walk up three.  Here one workgroup keeps a line's data in LDS through ALL of a level's transforms
// (the stages of ntt_lat_kernel as a device function whose first-stage load and last-stage store are the caller's lambdas).
template <int LOGN, bool INV, class LoadFn, class StoreFn>
__device__ __forceinline__ void lat_xform(const u64* __restrict__ twtab, u64 ninv, int tr, int j, u64* buf0, u64* buf1, LoadFn load, StoreFn store) {
    constexpr int N = 1 << LOGN, TPT = N / 8;
    constexpr int S = (LOGN + 2) / 3, LOGRL = LOGN - 3 * (S - 1);
    static_assert(S >= 2, "64 points at least");
    u64 x[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = 0;
    u64 tw[S - 1][8];
    {
        int Ns = 8;
#pragma unroll
        for (int s = 1; s < S; ++s) {
            const int logr = s + 1 < S ? 3 : LOGRL, R = 1 << logr, U = 8 >> logr;
            const bool last = s + 1 == S;
#pragma unroll
            for (int a = 0; a < U; ++a) {
                const int u = j + a * TPT, k = u & (Ns - 1);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = k * r * (N / (Ns * R));
                    tw[s - 1][a * R + r] = (r == 0) ? 0 : twtab[((INV && last) ? N : 0) + e];
                }
            }
            Ns *= R;
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) x[lat_brev<3>(r)] = load(r, j + r * TPT);
    lat_dft<INV, 3>(x);
#pragma unroll
    for (int r = 0; r < 8; ++r) buf0[lat_pad(tr * N + j * 8 + r)] = x[r];
    __syncthreads();
    int Ns = 8;
#pragma unroll
    for (int s = 1; s < S; ++s) {
        const bool last = s + 1 == S;
        const int logr = last ? LOGRL : 3, R = 1 << logr, U = 8 >> logr;
        const u64* in = ((s - 1) & 1) ? buf1 : buf0;
        u64* o = (s & 1) ? buf1 : buf0;
        u64 v[8];
#pragma unroll
        for (int a = 0; a < U; ++a) {
            const int u = j + a * TPT;
#pragma unroll
            for (int r = 0; r < R; ++r) v[a * R + r] = in[lat_pad(tr * N + u + r * (N / R))];
        }
#if TF_LAT_MUL4
        if (logr == 3) {
            // eight products in two blocks of four interleaved carry chains (gl::mont_mul4: 15 VALU per product and no wait-state
            // nops, against 18 + nops for products issued one by one); the slot of r = 0 rides along with n^-1 or with one
            u64 a0[4] = {v[1], v[2], v[3], v[4]}, b0[4] = {tw[s - 1][1], tw[s - 1][2], tw[s - 1][3], tw[s - 1][4]}, r0[4];
            u64 a1[4] = {v[5], v[6], v[7], v[0]}, b1[4] = {tw[s - 1][5], tw[s - 1][6], tw[s - 1][7], (INV && last) ? ninv : gl::ONE}, r1[4];
            gl::mont_mul4(a0, b0, r0);
            gl::mont_mul4(a1, b1, r1);
            x[lat_brev<3>(0)] = (INV && last) ? r1[3] : v[0];
            x[lat_brev<3>(1)] = r0[0], x[lat_brev<3>(2)] = r0[1], x[lat_brev<3>(3)] = r0[2], x[lat_brev<3>(4)] = r0[3];
            x[lat_brev<3>(5)] = r1[0], x[lat_brev<3>(6)] = r1[1], x[lat_brev<3>(7)] = r1[2];
        } else
#endif
        {
#pragma unroll
            for (int a = 0; a < U; ++a) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    u64 w = v[a * R + r];
                    if (r) w = gl::mont_mul(w, tw[s - 1][a * R + r]);
                    else if (INV && last) w = gl::mont_mul(w, ninv);
                    const int slot = a * R + (logr == 3 ? lat_brev<3>(r) : (logr == 2 ? lat_brev<2>(r) : r));
                    x[slot] = w;
                }
            }
        }
        if (logr == 3) lat_dft<INV, 3>(x);
        else if (logr == 2) lat_dft<INV, 2>(x);
        else lat_dft<INV, 1>(x);
#pragma unroll
        for (int a = 0; a < U; ++a) {
            const int u = j + a * TPT, k = u & (Ns - 1), j0 = (u - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int idx = j0 + r * Ns;
                if (last) store(idx, x[a * R + r]);
                else o[lat_pad(tr * N + idx)] = x[a * R + r];
            }
        }
        if (!last) __syncthreads();
        Ns *= R;
    }
}

struct TreeLevelArgs {
    const u64* cur;   // down: remainders of the level above (lines / 2 polynomials of 2d coefficients); up: the children's interpolants
    u64* nxt;         // down: lines x d remainders; up: lines x 2d interpolants
    const u64* ghat;  // down only: the level's cached transforms of the reversed-zerofier inverses, [children][2d]
    const u64* that;  // the level's cached tail transforms, [children][2d]
    const u64* tw_f;  // ntt_lat_kernel's tables of order 2d, forward and inverse
    const u64* tw_i;
    u64 ninv;
    long long lines;  // down: units x children; up: rows x parents
    long long per;    // down: children; up: parents  (the cached transforms repeat with this period)
};

// LDS: TWO buffers in all.  A transform run as lat_xform(first, second) leaves one of them unread by its last stage -- `second`
// when the stage count is even, `first` when odd -- so its store lambda writes the result THERE, and the next transform, which
// reads that buffer only in its first stage, runs as lat_xform(other, that one).
template <int LOGN>
struct LatChain {
    static constexpr bool EVEN = (((LOGN + 2) / 3) % 2) == 0;
    u64* first;
    u64* second;
    __device__ __forceinline__ u64* out() const { return EVEN ? second : first; }
    __device__ __forceinline__ void next() {  // the result just written becomes the next transform's `second`
        u64* o = out();
        u64* other = (o == first) ? second : first;
        first = other, second = o;
    }
};

// Walk down (polynomial.rs:1882-1894's remaindering through the tree, math/zerofier_tree.rs): line = one child.
//   rev(q) = rev(f_high) g mod x^d;   r = f_low - (q tail)_low          -- four transforms of order N = 2d, nothing leaves LDS
template <int LOGN>
__global__ void __launch_bounds__(LOGN == 12 ? 512 : 256) tree_down_level_kernel(const TreeLevelArgs A) {
    constexpr int N = 1 << LOGN, D = N / 2, TPT = N / 8, WG = LOGN == 12 ? 512 : 256, T = WG / TPT;
    constexpr int BUF = lat_pad(N * T) + 8;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    LatChain<LOGN> ch{lds, lds + BUF};
    const int t = threadIdx.x, tr = t / TPT, j = t - tr * TPT;
    const long long line = (long long)blockIdx.x * T + tr;
    const bool act = line < A.lines;
    const long long c = act ? (long long)((u32)line % (u32)A.per) : 0;  // (lines < 2^31: the grid is 32-bit)
    const u64* f = A.cur + (act ? (line >> 1) : 0) * N;
    const u64* gh = A.ghat + c * N;
    const u64* th = A.that + c * N;
    u64* dst = A.nxt + (act ? line : 0) * D;
    u64 ghv[8], thv[8];  // the cached transforms at this thread's first-stage indices, requested up front
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int idx = j + r * TPT;
        ghv[r] = gh[idx];
        thv[r] = th[idx];
    }
    u64* o = ch.out();
    lat_xform<LOGN, false>(A.tw_f, 0, tr, j, ch.first, ch.second,
                           [&](int r, int idx) -> u64 { return (act && r < 4) ? f[N - 1 - idx] : 0; },
                           [&](int idx, u64 v) { o[lat_pad(tr * N + idx)] = v; });
    __syncthreads();
    const u64* in = o;
    ch.next(), o = ch.out();
    lat_xform<LOGN, true>(A.tw_i, A.ninv, tr, j, ch.first, ch.second,
                          [&](int r, int idx) -> u64 { return gl::mont_mul(in[lat_pad(tr * N + idx)], ghv[r]); },
                          [&](int idx, u64 v) { if (idx < D) o[lat_pad(tr * N + D - 1 - idx)] = v; });
    __syncthreads();
    in = o;
    ch.next(), o = ch.out();
    lat_xform<LOGN, false>(A.tw_f, 0, tr, j, ch.first, ch.second,
                           [&](int r, int idx) -> u64 { return r < 4 ? in[lat_pad(tr * N + idx)] : 0; },
                           [&](int idx, u64 v) { o[lat_pad(tr * N + idx)] = v; });
    __syncthreads();
    in = o;
    ch.next();
    lat_xform<LOGN, true>(A.tw_i, A.ninv, tr, j, ch.first, ch.second,
                          [&](int r, int idx) -> u64 { return gl::mont_mul(in[lat_pad(tr * N + idx)], thv[r]); },
                          [&](int idx, u64 v) { if (act && idx < D) dst[idx] = gl::sub(f[idx], v); });
}

// Walk up (the interpolation's combination N = N_left Z_right + N_right Z_left, Z = tail + x^d): line = one parent; three LDS
// buffers (both children's transforms are alive when the inverse transform starts).  Measured against two thread groups per
// line transforming the children side by side (1024 threads at 2d = 4096, group 1 idle through the inverse): 122.2 vs 126.0 us
// per prepared-tree interpolation of 2^12 points -- the wider workgroup costs more than the parallel child saves.
template <int LOGN>
__global__ void __launch_bounds__(LOGN == 12 ? 512 : 256) tree_up_level_kernel(const TreeLevelArgs A) {
    constexpr int N = 1 << LOGN, D = N / 2, TPT = N / 8, WG = LOGN == 12 ? 512 : 256, T = WG / TPT;
    constexpr int BUF = lat_pad(N * T) + 8;
    constexpr bool EVEN = LatChain<LOGN>::EVEN;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64* p = lds;
    u64* q = lds + BUF;
    u64* c = lds + 2 * BUF;
    const int t = threadIdx.x, tr = t / TPT, j = t - tr * TPT;
    const long long line = (long long)blockIdx.x * T + tr;
    const bool act = line < A.lines;
    const u32 node = act ? (u32)line % (u32)A.per : 0;
    const u64* c0 = A.cur + (act ? line : 0) * N;  // the two children, d coefficients each, side by side
    const u64* t0 = A.that + 2 * (long long)node * N;
    u64* dst = A.nxt + (act ? line : 0) * N;
    u64 zlv[8], zrv[8];  // Z_left, Z_right transforms at this thread's first-stage indices, requested up front
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int idx = j + r * TPT;
        const u64 sgn = (idx & 1) ? gl::P - gl::ONE : gl::ONE;
        zlv[r] = gl::add(t0[idx], sgn);
        zrv[r] = gl::add(t0[N + idx], sgn);
    }
    u64* a = EVEN ? q : p;  // left child's transform: the buffer lat_xform(p, q)'s last stage does not read
    lat_xform<LOGN, false>(A.tw_f, 0, tr, j, p, q, [&](int r, int idx) -> u64 { return (act && r < 4) ? c0[idx] : 0; },
                           [&](int idx, u64 v) { a[lat_pad(tr * N + idx)] = v; });
    __syncthreads();
    u64* w0 = EVEN ? p : q;  // the right child's transform works in the other two buffers
    u64* b = EVEN ? c : w0;
    lat_xform<LOGN, false>(A.tw_f, 0, tr, j, w0, c, [&](int r, int idx) -> u64 { return (act && r < 4) ? c0[D + idx] : 0; },
                           [&](int idx, u64 v) { b[lat_pad(tr * N + idx)] = v; });
    __syncthreads();
    u64* f0 = EVEN ? w0 : c;  // the inverse transform's first stage writes the one buffer that holds neither
    lat_xform<LOGN, true>(A.tw_i, A.ninv, tr, j, f0, b,
                          [&](int r, int idx) -> u64 {
                              return gl::add(gl::mont_mul(a[lat_pad(tr * N + idx)], zrv[r]), gl::mont_mul(b[lat_pad(tr * N + idx)], zlv[r]));
                          },
                          [&](int idx, u64 v) { if (act) dst[idx] = v; });
}

// ---- a whole level of the zerofier-tree BUILD in one launch (BFieldElement, 2d <= 2048) ---------------------------------------
// Per parent the build runs nine transforms (math/zerofier_tree.rs builds the same products with fast_multiply; the power-series
// inverses are this design's own):
//   phase 1  four transforms of order 2d: both children's tails and inverses  -> That, Ghat (kept for the walks)
//   phase 2  tail_parent = iNTT((TL^ + s)(TR^ + s) - 1),   G = iNTT(GL^ GR^) mod x^d
//   phase 3  two transforms of order 4d: G and H = rev(Z_parent) mod x^2d
//   phase 4  inv_parent = iNTT(G^ (2 - H^ G^)) mod x^2d                       (one Newton step)
// One workgroup slice (4 * 2d / 8 threads) per parent: the four / two transforms of a phase run side by side in thread groups,
// everything between the phases stays in LDS (two buffers; LatChain's rule for where a result lands).  Groups without a
// transform in a phase walk through its barriers on zeros.
struct TreeBuildArgs {
    const u64* tails;   // level l: [children][d]
    const u64* inv;     // level l: [children][d]
    u64* that;          // level l: [children][2d]
    u64* ghat;          // level l: [children][2d]
    u64* ptails;        // level l + 1: [parents][2d]   (null: top level, phase 1 only)
    u64* pinv;          // level l + 1: [parents][2d]
    const u64 *tw_f2, *tw_i2, *tw_f4, *tw_i4;  // ntt_lat_kernel's tables of order 2d and 4d
    u64 ninv2, ninv4;
    long long parents;
};
template <int LOGN2>
struct TreeBuildGeom {
    static constexpr int N2 = 1 << LOGN2, TPT2 = N2 / 8;
    static constexpr int T = 4 * TPT2 >= 256 ? 1 : 256 / (4 * TPT2);  // parents per workgroup
    static constexpr int WG = 4 * TPT2 * T;
    static constexpr int BUF = lat_pad(N2 * 4 * T) + 8;
};
template <int LOGN2>
__global__ void __launch_bounds__(TreeBuildGeom<LOGN2>::WG) tree_build_level_kernel(const TreeBuildArgs A) {
    using G = TreeBuildGeom<LOGN2>;
    constexpr int N2 = G::N2, D = N2 / 2, TPT2 = G::TPT2, T = G::T, BUF = G::BUF, LOGN4 = LOGN2 + 1, N4 = 2 * N2, TPT4 = 2 * TPT2;
    constexpr bool EVEN2 = LatChain<LOGN2>::EVEN, EVEN4 = LatChain<LOGN4>::EVEN;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64* P = lds;
    u64* Q = lds + BUF;
    const int t = threadIdx.x;
    const int g = t / TPT2, j = t - g * TPT2, slot = g >> 2, role = g & 3;
    const int g4 = t / TPT4, j4 = t - g4 * TPT4, role4 = g4 & 1;
    const long long parent = (long long)blockIdx.x * T + slot;
    const bool act = parent < A.parents;
    const long long child = 2 * (act ? parent : 0) + (role & 1);
    // ---- phase 1
    u64* O1 = EVEN2 ? Q : P;
    {
        const u64* src = (role < 2 ? A.tails : A.inv) + child * D;
        u64* dst = (role < 2 ? A.that : A.ghat) + child * N2;
        lat_xform<LOGN2, false>(A.tw_f2, 0, g, j, P, Q, [&](int r, int idx) -> u64 { return (act && r < 4) ? src[idx] : 0; },
                                [&](int idx, u64 v) {
                                    O1[lat_pad(g * N2 + idx)] = v;
                                    if (act) dst[idx] = v;
                                });
    }
    if (!A.ptails) return;
    __syncthreads();
    // ---- phase 2: group 0 the parent's tail, group 2 the product of the children's inverses
    u64* F2 = EVEN2 ? P : Q;           // = the buffer that does not hold O1
    u64* O2 = EVEN2 ? O1 : F2;
    {
        const int gb = g & ~3;
        u64* pt = A.ptails + (act ? parent : 0) * N2;
        lat_xform<LOGN2, true>(A.tw_i2, A.ninv2, g, j, F2, O1,
                               [&](int, int idx) -> u64 {
                                   if (role == 0) {
                                       const u64 sgn = (idx & 1) ? gl::P - gl::ONE : gl::ONE;
                                       const u64 a = gl::add(O1[lat_pad(gb * N2 + idx)], sgn), b = gl::add(O1[lat_pad((gb + 1) * N2 + idx)], sgn);
                                       return gl::sub(gl::mont_mul(a, b), gl::ONE);
                                   }
                                   if (role == 2) return gl::mont_mul(O1[lat_pad((gb + 2) * N2 + idx)], O1[lat_pad((gb + 3) * N2 + idx)]);
                                   return 0;
                               },
                               [&](int idx, u64 v) {
                                   if (role == 0) {
                                       O2[lat_pad(g * N2 + idx)] = v;
                                       if (act) pt[idx] = v;
                                   } else if (role == 2 && idx < D) {
                                       O2[lat_pad(g * N2 + idx)] = v;
                                   }
                               });
    }
    __syncthreads();
    // ---- phase 3 (order 4d, two groups per parent): G^ and H^
    u64* F3 = (O2 == P) ? Q : P;
    u64* O3 = EVEN4 ? O2 : F3;
    {
        const int nb = (g4 & ~1) * 2;  // first 2d-sized region of this parent
        lat_xform<LOGN4, false>(A.tw_f4, 0, g4, j4, F3, O2,
                                [&](int r, int idx) -> u64 {
                                    if (role4 == 0) return idx < D ? O2[lat_pad((nb + 2) * N2 + idx)] : 0;  // G = (g_l g_r) mod x^d
                                    if (idx >= N2) return 0;                                                  // H = rev(Z_parent) mod x^2d
                                    return idx == 0 ? gl::ONE : O2[lat_pad(nb * N2 + N2 - idx)];
                                },
                                [&](int idx, u64 v) { O3[lat_pad(g4 * N4 + idx)] = v; });
    }
    __syncthreads();
    // ---- phase 4 (order 4d): one Newton step, its low 2d coefficients are the parent's inverse
    u64* F4 = (O3 == P) ? Q : P;
    {
        const int gp = g4 & ~1;
        u64* pi = A.pinv + (act ? parent : 0) * N2;
        const u64 two = gl::add(gl::ONE, gl::ONE);
        lat_xform<LOGN4, true>(A.tw_i4, A.ninv4, g4, j4, F4, O3,
                               [&](int, int idx) -> u64 {
                                   if (role4) return 0;
                                   const u64 gh = O3[lat_pad(gp * N4 + idx)], hh = O3[lat_pad((gp + 1) * N4 + idx)];
                                   return gl::mont_mul(gh, gl::sub(two, gl::mont_mul(hh, gh)));
                               },
                               [&](int idx, u64 v) { if (act && role4 == 0 && idx < N2) pi[idx] = v; });
    }
}

// ---- the same over XFieldElement: a line's three limb transforms run SIDE BY SIDE in three thread groups of one workgroup
// (one after the other they would lose to separate launches); the extension-field products between the transforms read all three
// limbs of an element from LDS and every group forms its own limb of the product (x_field_element.rs:512-536).
__device__ __forceinline__ u64 xfe_mul_limb(u64 s0, u64 s1, u64 s2, u64 o0, u64 o1, u64 o2, int limb) {
    if (limb == 0) return gl::sub(gl::sub(gl::mont_mul(s0, o0), gl::mont_mul(s2, o1)), gl::mont_mul(s1, o2));
    if (limb == 1)
        return gl::add(gl::add(gl::sub(gl::add(gl::mont_mul(s1, o0), gl::mont_mul(s0, o1)), gl::mont_mul(s2, o2)), gl::mont_mul(s2, o1)), gl::mont_mul(s1, o2));
    return gl::add(gl::add(gl::add(gl::mont_mul(s2, o0), gl::mont_mul(s1, o1)), gl::mont_mul(s0, o2)), gl::mont_mul(s2, o2));
}
// threads of a workgroup: T lines x 3 limbs x N / 8
template <int LOGN>
struct TreeXfeGeom {
    static constexpr int N = 1 << LOGN, TPT = N / 8;
    static constexpr int T = TPT >= 128 ? 1 : 128 / TPT;
    static constexpr int WG = 3 * TPT * T;
    static constexpr int BUF = lat_pad(N * 3 * T) + 8;
};

template <int LOGN>
__global__ void __launch_bounds__(TreeXfeGeom<LOGN>::WG) tree_down_level_xfe_kernel(const TreeLevelArgs A) {
    using G = TreeXfeGeom<LOGN>;
    constexpr int N = G::N, D = N / 2, TPT = G::TPT, T = G::T, BUF = G::BUF;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    LatChain<LOGN> ch{lds, lds + BUF};
    const int t = threadIdx.x, g = t / TPT, j = t - g * TPT, lw = g / 3, limb = g - 3 * lw, g0 = g - limb;
    const long long line = (long long)blockIdx.x * T + lw;
    const bool act = line < A.lines;
    const long long c = act ? (long long)((u32)line % (u32)A.per) : 0;
    const u64* f = A.cur + (act ? (line >> 1) : 0) * N * 3;
    const u64* gh = A.ghat + c * N * 3;
    const u64* th = A.that + c * N * 3;
    u64* dst = A.nxt + (act ? line : 0) * D * 3;
    u64 cv[8][3];  // the cached transform an extension-field product needs, at this thread's first-stage indices, requested ahead
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) cv[r][k] = gh[(j + r * TPT) * 3 + k];
    u64* o = ch.out();
    lat_xform<LOGN, false>(A.tw_f, 0, g, j, ch.first, ch.second,
                           [&](int r, int idx) -> u64 { return (act && r < 4) ? f[(N - 1 - idx) * 3 + limb] : 0; },
                           [&](int idx, u64 v) { o[lat_pad(g * N + idx)] = v; });
    __syncthreads();
    const u64* in = o;
    ch.next(), o = ch.out();
    lat_xform<LOGN, true>(A.tw_i, A.ninv, g, j, ch.first, ch.second,
                          [&](int r, int idx) -> u64 {
                              return xfe_mul_limb(in[lat_pad(g0 * N + idx)], in[lat_pad((g0 + 1) * N + idx)], in[lat_pad((g0 + 2) * N + idx)],
                                                  cv[r][0], cv[r][1], cv[r][2], limb);
                          },
                          [&](int idx, u64 v) { if (idx < D) o[lat_pad(g * N + D - 1 - idx)] = v; });
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) cv[r][k] = th[(j + r * TPT) * 3 + k];
    __syncthreads();
    in = o;
    ch.next(), o = ch.out();
    lat_xform<LOGN, false>(A.tw_f, 0, g, j, ch.first, ch.second,
                           [&](int r, int idx) -> u64 { return r < 4 ? in[lat_pad(g * N + idx)] : 0; },
                           [&](int idx, u64 v) { o[lat_pad(g * N + idx)] = v; });
    __syncthreads();
    in = o;
    ch.next();
    lat_xform<LOGN, true>(A.tw_i, A.ninv, g, j, ch.first, ch.second,
                          [&](int r, int idx) -> u64 {
                              return xfe_mul_limb(in[lat_pad(g0 * N + idx)], in[lat_pad((g0 + 1) * N + idx)], in[lat_pad((g0 + 2) * N + idx)],
                                                  cv[r][0], cv[r][1], cv[r][2], limb);
                          },
                          [&](int idx, u64 v) { if (act && idx < D) dst[idx * 3 + limb] = gl::sub(f[idx * 3 + limb], v); });
}

template <int LOGN>
__global__ void __launch_bounds__(TreeXfeGeom<LOGN>::WG) tree_up_level_xfe_kernel(const TreeLevelArgs A) {
    using G = TreeXfeGeom<LOGN>;
    constexpr int N = G::N, D = N / 2, TPT = G::TPT, T = G::T, BUF = G::BUF;
    constexpr bool EVEN = LatChain<LOGN>::EVEN;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64* p = lds;
    u64* q = lds + BUF;
    u64* c = lds + 2 * BUF;
    const int t = threadIdx.x, g = t / TPT, j = t - g * TPT, lw = g / 3, limb = g - 3 * lw, g0 = g - limb;
    const long long line = (long long)blockIdx.x * T + lw;
    const bool act = line < A.lines;
    const u32 node = act ? (u32)line % (u32)A.per : 0;
    const u64* c0 = A.cur + (act ? line : 0) * N * 3;
    const u64* t0 = A.that + 2 * (long long)node * N * 3;
    u64* dst = A.nxt + (act ? line : 0) * N * 3;
    // Z_left, Z_right transforms (tail + x^d: (-1)^idx on limb 0) at this thread's first-stage indices (one array per limb: arrays
    // of arrays captured by the lambdas below end up in scratch memory)
    u64 zl0[8], zl1[8], zl2[8], zr0[8], zr1[8], zr2[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int idx = j + r * TPT;
        const u64 sgn = (idx & 1) ? gl::P - gl::ONE : gl::ONE;
        zl0[r] = gl::add(t0[idx * 3], sgn), zl1[r] = t0[idx * 3 + 1], zl2[r] = t0[idx * 3 + 2];
        zr0[r] = gl::add(t0[(N + idx) * 3], sgn), zr1[r] = t0[(N + idx) * 3 + 1], zr2[r] = t0[(N + idx) * 3 + 2];
    }
    u64* a = EVEN ? q : p;
    lat_xform<LOGN, false>(A.tw_f, 0, g, j, p, q, [&](int r, int idx) -> u64 { return (act && r < 4) ? c0[idx * 3 + limb] : 0; },
                           [&](int idx, u64 v) { a[lat_pad(g * N + idx)] = v; });
    __syncthreads();
    u64* w0 = EVEN ? p : q;
    u64* b = EVEN ? c : w0;
    lat_xform<LOGN, false>(A.tw_f, 0, g, j, w0, c, [&](int r, int idx) -> u64 { return (act && r < 4) ? c0[(D + idx) * 3 + limb] : 0; },
                           [&](int idx, u64 v) { b[lat_pad(g * N + idx)] = v; });
    __syncthreads();
    u64* f0 = EVEN ? w0 : c;
    u64 v8[8];  // the combination N_left Z_right + N_right Z_left at this thread's first-stage indices
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int idx = j + r * TPT;
        const u64 x = xfe_mul_limb(a[lat_pad(g0 * N + idx)], a[lat_pad((g0 + 1) * N + idx)], a[lat_pad((g0 + 2) * N + idx)], zr0[r], zr1[r], zr2[r], limb);
        const u64 y = xfe_mul_limb(b[lat_pad(g0 * N + idx)], b[lat_pad((g0 + 1) * N + idx)], b[lat_pad((g0 + 2) * N + idx)], zl0[r], zl1[r], zl2[r], limb);
        v8[r] = gl::add(x, y);
    }
    lat_xform<LOGN, true>(A.tw_i, A.ninv, g, j, f0, b, [&](int r, int) -> u64 { return v8[r]; },
                          [&](int idx, u64 v) { if (act) dst[idx * 3 + limb] = v; });
}

// ---- 2^13 <= n <= 2^20, little work per call: the same eight-elements-per-thread stages as the two passes of n = N1 N2 ------
// (one slice per call is the reference's own call shape: math/ntt.rs:67-82 takes ONE slice).  A "line" is one DFT instance:
//   column pass (LAST = false): line c = word-column c of the N2 L words of a row; element i at  i * es + c;  after the last stage
//       output k is multiplied by the inter-pass twiddle w_n^(k b), b = c / L, and stored where it came from (or into scratch);
//   last pass (LAST = true):    line c = (k1, limb) = (c / L, c % L): input row k1 of N2 contiguous elements, output k at
//       (k1 + N1 k) L + limb -- natural order, no bit reversal.
// cfast: adjacent threads take adjacent lines (the column pass: coalesced both ways); otherwise adjacent threads walk along the line
// (the last pass: contiguous loads, strided 8-byte stores -- a call this small is bound by latency, not by store efficiency).
struct NttLat2Args {
    const u64* in;
    u64* out;
    const u64* in2;            // or null: second operand laid out like `in`, multiplied in on load (first pass, L = 1)
    const u64* tw;             // [2][N]: w_N^(+-e), then scale * w_N^(+-e)
    const u64* post_tw;        // column pass: T[k * tw_rs + b]
    long long n_coeffs;        // column pass: < 0 none; else input element index i * nc_es + c / L >= n_coeffs reads as zero
    long long nc_es;
    long long in_bs, out_bs;   // words between batch entries
    long long lines;           // lines per batch entry
    long long in_es, out_es;   // words between consecutive elements of a line
    long long in_lhi, out_lhi; // line c starts at (c / L) * lhi + (c % L)
    long long tw_rs;
    u64 scale;                 // last pass of an inverse: n^-1 (Montgomery); 0 otherwise
    int L;
    int cfast;
    int tiles_per_entry;       // ceil(lines / T)
};

template <int LOGN, bool INV, bool LAST, int WG = 256>
__global__ void __launch_bounds__(WG) ntt_lat2_kernel(const NttLat2Args A) {
    constexpr int N = 1 << LOGN, TPT = N / 8, T = WG / TPT;
    constexpr int S = (LOGN + 2) / 3, LOGRL = LOGN - 3 * (S - 1);
    constexpr int BUF = lat_pad(N * T) + 8;
    static_assert(LOGN >= 6 && LOGN <= 10, "");
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int t = threadIdx.x;
    const int lc = A.cfast ? t % T : t / TPT;   // line within the tile
    const int j = A.cfast ? t / T : t % TPT;    // butterfly unit within the line
    const long long entry = blockIdx.x / A.tiles_per_entry, tile = blockIdx.x - entry * A.tiles_per_entry;
    const long long c = tile * T + lc;
    const bool act = c < A.lines;
    const int L = A.L;
    const long long chi = act ? c / L : 0;
    const int clo = act ? (int)(c - chi * L) : 0;
    const u64* src = A.in + entry * A.in_bs + chi * A.in_lhi + clo;
    u64* dst = A.out + entry * A.out_bs + chi * A.out_lhi + clo;
    const auto li = [&](int idx) { return A.cfast ? lat_pad(idx * T + lc) : lat_pad(lc * N + idx); };  // LDS index of element idx of my line
    u64 x[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = 0;
    u64 tw[S > 1 ? S - 1 : 1][8];
    {
        int Ns = 8;
#pragma unroll
        for (int s = 1; s < S; ++s) {
            const int logr = s + 1 < S ? 3 : LOGRL, R = 1 << logr, U = 8 >> logr;
            const bool last = s + 1 == S;
#pragma unroll
            for (int a = 0; a < U; ++a) {
                const int u = j + a * TPT, k = u & (Ns - 1);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = k * r * (N / (Ns * R));
                    tw[s - 1][a * R + r] = (r == 0) ? 0 : A.tw[((LAST && INV && last) ? N : 0) + e];
                }
            }
            Ns *= R;
        }
    }
    if (act) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = j + r * TPT;
            u64 v = 0;
            if (LAST || A.n_coeffs < 0 || (long long)idx * A.nc_es + chi < A.n_coeffs) {
                v = src[(long long)idx * A.in_es];
                if (!LAST && A.in2) v = gl::mont_mul(v, (A.in2 + entry * A.in_bs + chi * A.in_lhi + clo)[(long long)idx * A.in_es]);
            }
            x[lat_brev<3>(r)] = v;
        }
    }
    lat_dft<INV, 3>(x);
    u64* bufs[2] = {lds + 0, lds + BUF};
#pragma unroll
    for (int r = 0; r < 8; ++r) bufs[0][li(j * 8 + r)] = x[r];
    __syncthreads();
    int Ns = 8;
#pragma unroll
    for (int s = 1; s < S; ++s) {
        const bool last = s + 1 == S;
        const int logr = last ? LOGRL : 3, R = 1 << logr, U = 8 >> logr;
        const u64* in = bufs[(s - 1) & 1];
        u64* o = bufs[s & 1];
        u64 v[8];
#pragma unroll
        for (int a = 0; a < U; ++a) {
            const int u = j + a * TPT;
#pragma unroll
            for (int r = 0; r < R; ++r) v[a * R + r] = in[li(u + r * (N / R))];
        }
        u64 ptw[8];
        if (last && !LAST && act) {  // inter-pass twiddles of my outputs, requested before the arithmetic
#pragma unroll
            for (int a = 0; a < U; ++a) {
                const int u = j + a * TPT, k = u & (Ns - 1), j0 = (u - k) * R + k;
#pragma unroll
                for (int r = 0; r < R; ++r) ptw[a * R + r] = A.post_tw[(long long)(j0 + r * Ns) * A.tw_rs + chi];
            }
        }
#if TF_LAT_MUL4
        if (logr == 3) {  // (see lat_xform: two blocks of four interleaved products)
            u64 a0[4] = {v[1], v[2], v[3], v[4]}, b0[4] = {tw[s - 1][1], tw[s - 1][2], tw[s - 1][3], tw[s - 1][4]}, r0[4];
            u64 a1[4] = {v[5], v[6], v[7], v[0]}, b1[4] = {tw[s - 1][5], tw[s - 1][6], tw[s - 1][7], (LAST && INV && last) ? A.scale : gl::ONE}, r1[4];
            gl::mont_mul4(a0, b0, r0);
            gl::mont_mul4(a1, b1, r1);
            x[lat_brev<3>(0)] = (LAST && INV && last) ? r1[3] : v[0];
            x[lat_brev<3>(1)] = r0[0], x[lat_brev<3>(2)] = r0[1], x[lat_brev<3>(3)] = r0[2], x[lat_brev<3>(4)] = r0[3];
            x[lat_brev<3>(5)] = r1[0], x[lat_brev<3>(6)] = r1[1], x[lat_brev<3>(7)] = r1[2];
        } else
#endif
#pragma unroll
        for (int a = 0; a < U; ++a) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                u64 w = v[a * R + r];
                if (r) w = gl::mont_mul(w, tw[s - 1][a * R + r]);
                else if (LAST && INV && last) w = gl::mont_mul(w, A.scale);
                const int slot = a * R + (logr == 3 ? lat_brev<3>(r) : (logr == 2 ? lat_brev<2>(r) : r));
                x[slot] = w;
            }
        }
        if (logr == 3) lat_dft<INV, 3>(x);
        else if (logr == 2) lat_dft<INV, 2>(x);
        else lat_dft<INV, 1>(x);
#pragma unroll
        for (int a = 0; a < U; ++a) {
            const int u = j + a * TPT, k = u & (Ns - 1), j0 = (u - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int idx = j0 + r * Ns;
                if (last) {
                    if (act) {
                        u64 val = x[a * R + r];
                        if (!LAST) val = gl::mont_mul(val, ptw[a * R + r]);
                        dst[(long long)idx * A.out_es] = val;
                    }
                } else {
                    o[li(idx)] = x[a * R + r];
                }
            }
        }
        if (!last) __syncthreads();
        Ns *= R;
    }
}

// ---- n <= 16: one thread per (transform, limb); reference-shaped radix-2 loop, tables in global memory.
struct NttTinyArgs {
    const u64* in;
    u64* out;
    const u64* tw;         // stage tables back to back: stage i (m = 2^i) at offset m - 1 (ntt.rs:309-324)
    const u64* pre_scale;  // or null
    const u64* post_scale; // or null: output element j times post_scale[j]
    long long n_coeffs;    // < 0: none
    long long in_bs, out_bs;  // batch strides in words
    long long count;       // batch * L
    u64 scale;             // Montgomery n^-1 for the inverse, 0 = no scaling
    int log_n;
    int L;
};

__global__ void __launch_bounds__(256) ntt_tiny_kernel(const NttTinyArgs A) {
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= A.count) return;
    const int L = A.L, n = 1 << A.log_n;
    const long long b = id / L;
    const int limb = (int)(id - b * L);
    const u64* src = A.in + b * A.in_bs + limb;
    u64* dst = A.out + b * A.out_bs + limb;
    u64 x[16];
    for (int j = 0; j < 16; ++j) {
        if (j < n) {
            int r = (int)(__brev((unsigned)j) >> (32 - A.log_n));
            if (A.log_n == 0) r = 0;
            u64 v = 0;
            if (A.n_coeffs < 0 || r < A.n_coeffs) {
                v = src[(long long)r * L];
                if (A.pre_scale) v = gl::mont_mul(v, A.pre_scale[r]);
            }
            x[j] = v;
        }
    }
    for (int m = 1; m < n; m *= 2) {  // ntt.rs:195-214
        for (int k = 0; k < n; k += 2 * m) {
            for (int j = 0; j < m; ++j) {
                u64 u = x[k + j];
                u64 v = gl::mont_mul(x[k + j + m], A.tw[m - 1 + j]);
                x[k + j] = gl::add(u, v);
                x[k + j + m] = gl::sub(u, v);
            }
        }
    }
    for (int j = 0; j < 16; ++j) {
        if (j < n) {
            u64 v = x[j];
            if (A.scale) v = gl::mont_mul(v, A.scale);
            if (A.post_scale) v = gl::mont_mul(v, A.post_scale[j]);
            dst[(long long)j * L] = v;
        }
    }
}

// ---- table builders ----------------------------------------------------------------------------
// out[k * B + b] = HI[e >> h] * LO[e & (2^h - 1)],  e = (k * b) mod M   (w_M^e split in two small tables)
__global__ void __launch_bounds__(256) build_post_tw_kernel(u64* out, const u64* hi, const u64* lo, int h, long long R,
                                                            long long B) {
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= R * B) return;
    long long k = id / B, b = id - k * B;
    unsigned long long e = ((unsigned long long)k * (unsigned long long)b) % (unsigned long long)(R * B);
    out[id] = gl::mont_mul(hi[e >> h], lo[e & ((1ull << h) - 1)]);
}

// out[c * n + j] = HI_c[j >> h] * LO_c[j & (2^h - 1)]   (base_c^j; grid.y = c; tabs = [c][nhi + nlo] split tables)
__global__ void __launch_bounds__(256) build_pow_tables_kernel(u64* out, const u64* tabs, int h, long long n, long long nhi, long long nlo) {
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const u64* hi = tabs + (long long)blockIdx.y * (nhi + nlo);
    const u64* lo = hi + nhi;
    out[(long long)blockIdx.y * n + id] = gl::mont_mul(hi[id >> h], lo[id & ((1ll << h) - 1)]);
}

// ---- pointwise products (Hadamard) ---------------------------------------------------------------
// out[i] = a[i] * b[i] over BFieldElement (b_field_element.rs:755-762)
__global__ void __launch_bounds__(256) hadamard_bfe_kernel(const u64* a, const u64* b, u64* out, long long count) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // two elements per lane and access (16-byte loads/stores) when the three arrays are 16-byte aligned
    if ((((unsigned long long)a | (unsigned long long)b | (unsigned long long)out) & 15) == 0) {
        typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
        const long long pairs = count >> 1;
        const ull2* a2 = reinterpret_cast<const ull2*>(a);
        const ull2* b2 = reinterpret_cast<const ull2*>(b);
        ull2* o2 = reinterpret_cast<ull2*>(out);
        for (long long k = i; k < pairs; k += stride) {
            const ull2 x = a2[k], y = b2[k];
            u64 r0, r1;
            gl::mont_mul2(x.x, y.x, x.y, y.y, r0, r1);
            ull2 r;
            r.x = r0;
            r.y = r1;
            o2[k] = r;
        }
        if ((count & 1) && i == 0) out[count - 1] = gl::mont_mul(a[count - 1], b[count - 1]);
        return;
    }
    for (; i < count; i += stride) out[i] = gl::mont_mul(a[i], b[i]);
}

// out[i] = a[i] * b[i] over XFieldElement = F_p[x]/(x^3 - x + 1)  (x_field_element.rs:512-536):
// with self = [c, b, a], other = [f, e, d]:  r0 = cf - ae - bd;  r1 = bf + ce - ad + ae + bd;  r2 = af + be + cd + ad
__global__ void __launch_bounds__(256) hadamard_xfe_kernel(const u64* pa, const u64* pb, u64* out, long long count) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        const u64 c = pa[3 * i], b = pa[3 * i + 1], a = pa[3 * i + 2];
        const u64 f = pb[3 * i], e = pb[3 * i + 1], d = pb[3 * i + 2];
        const u64 ae = gl::mont_mul(a, e), bd = gl::mont_mul(b, d), ad = gl::mont_mul(a, d);
        const u64 r0 = gl::sub(gl::sub(gl::mont_mul(c, f), ae), bd);
        const u64 r1 = gl::add(gl::add(gl::sub(gl::add(gl::mont_mul(b, f), gl::mont_mul(c, e)), ad), ae), bd);
        const u64 r2 = gl::add(gl::add(gl::add(gl::mont_mul(a, f), gl::mont_mul(b, e)), gl::mont_mul(c, d)), ad);
        out[3 * i] = r0;
        out[3 * i + 1] = r1;
        out[3 * i + 2] = r2;
    }
}

// dst[b][0..n_dst) = src[b][0..min(n_src, n_dst)) then zeros (resize(order, ZERO), polynomial.rs:913-914); words, not elements
__global__ void __launch_bounds__(256) pad_copy_kernel(const u64* src, u64* dst, long long n_src, long long n_dst, long long batch,
                                                       long long src_stride) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n_dst * batch; i += stride) {
        const long long b = i / n_dst, j = i - b * n_dst;
        dst[i] = j < n_src ? src[b * src_stride + j] : 0;
    }
}

// out[i] = f(points[i]) by Horner's rule, one lane per point (Polynomial::iterative_batch_evaluate,
// polynomial.rs:1876-1878; same values as batch_evaluate :1840-1852).  The coefficient reads are wave-uniform.
__device__ __forceinline__ void xfe_mul(const u64 (&s)[3], const u64 (&o)[3], u64 (&r)[3]) {
    // x_field_element.rs:512-536 with self = [c, b, a], other = [f, e, d]
    const u64 c = s[0], b = s[1], a = s[2], f = o[0], e = o[1], d = o[2];
    const u64 ae = gl::mont_mul(a, e), bd = gl::mont_mul(b, d), ad = gl::mont_mul(a, d);
    r[0] = gl::sub(gl::sub(gl::mont_mul(c, f), ae), bd);
    r[1] = gl::add(gl::add(gl::sub(gl::add(gl::mont_mul(b, f), gl::mont_mul(c, e)), ad), ae), bd);
    r[2] = gl::add(gl::add(gl::add(gl::mont_mul(a, f), gl::mont_mul(b, e)), gl::mont_mul(c, d)), ad);
}

// Field element of width L (1: BFieldElement, 3: XFieldElement) for the evaluation kernels.
template <int L>
__device__ __forceinline__ void fe_mul(const u64 (&a)[L], const u64 (&b)[L], u64 (&r)[L]) {
    if constexpr (L == 1) r[0] = gl::mont_mul(a[0], b[0]);
    else xfe_mul(a, b, r);
}

// Lane per point: the right shape for short polynomials at many points.  grid = (ceil(m / 256), batch).
// CL = words per COEFFICIENT (L: same field as the points; 1 with L = 3: Polynomial<BFieldElement>::evaluate::<XFieldElement, _>,
// polynomial.rs:309-320 -- the base-field coefficient is added to limb 0 of the extension-field accumulator).
template <int L, int CL = L>
__global__ void __launch_bounds__(256) batch_evaluate_kernel(const u64* coeffs, long long n_coeffs, long long poly_stride,
                                                            const u64* points, long long n_points, u64* out, long long out_stride) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_points) return;
    const u64* c = coeffs + (long long)blockIdx.y * poly_stride;
    u64 x[L], acc[L];
#pragma unroll
    for (int k = 0; k < L; ++k) { x[k] = points[L * i + k]; acc[k] = 0; }
    for (long long j = n_coeffs - 1; j >= 0; --j) {
        u64 t[L];
        fe_mul<L>(acc, x, t);
#pragma unroll
        for (int k = 0; k < L; ++k) acc[k] = k < CL ? gl::add(t[k], c[CL * j + k]) : t[k];
    }
    u64* o = out + ((long long)blockIdx.y * out_stride + i) * L;  // out_stride: the points of the whole call (a launch may be a slab of them)
#pragma unroll
    for (int k = 0; k < L; ++k) o[k] = acc[k];
}

// Workgroup per (point, polynomial): thread t runs Horner in X = x^256 over coefficients t, t + 256, ... (coalesced
// reads), scales by x^t, and the 256 partial values are summed through LDS:
//   f(x) = sum_t x^t * sum_j c[t + 256 j] X^j.   grid = (m, batch).
template <int L, int CL = L>
__global__ void __launch_bounds__(256) batch_evaluate_split_kernel(const u64* coeffs, long long n_coeffs, long long poly_stride,
                                                                  const u64* points, long long n_points, u64* out, long long out_stride) {
    __shared__ u64 part[256 * L];
    const int t = threadIdx.x;
    const long long i = blockIdx.x;
    const u64* c = coeffs + (long long)blockIdx.y * poly_stride;
    u64 x[L], X[L], acc[L], pw[L], sq[L], tmp[L];
#pragma unroll
    for (int k = 0; k < L; ++k) { x[k] = points[L * i + k]; X[k] = x[k]; sq[k] = x[k]; acc[k] = 0; pw[k] = k ? 0 : gl::ONE; }
#pragma unroll 1
    for (int b = 0; b < 8; ++b) {  // X = x^256 and pw = x^t by square-and-multiply on the bits of t
        if ((t >> b) & 1) {
            fe_mul<L>(pw, sq, tmp);
#pragma unroll
            for (int k = 0; k < L; ++k) pw[k] = tmp[k];
        }
        fe_mul<L>(sq, sq, tmp);
#pragma unroll
        for (int k = 0; k < L; ++k) sq[k] = tmp[k];
    }
#pragma unroll
    for (int k = 0; k < L; ++k) X[k] = sq[k];
    if (t < n_coeffs) {
        for (long long j = (n_coeffs - 1 - t) >> 8; j >= 0; --j) {
            fe_mul<L>(acc, X, tmp);
            const u64* cj = c + (t + (j << 8)) * CL;
#pragma unroll
            for (int k = 0; k < L; ++k) acc[k] = k < CL ? gl::add(tmp[k], cj[k]) : tmp[k];
        }
        fe_mul<L>(acc, pw, tmp);
#pragma unroll
        for (int k = 0; k < L; ++k) acc[k] = tmp[k];
    }
#pragma unroll
    for (int k = 0; k < L; ++k) part[t * L + k] = acc[k];
    __syncthreads();
#pragma unroll 1
    for (int h = 128; h > 0; h >>= 1) {
        if (t < h) {
#pragma unroll
            for (int k = 0; k < L; ++k) part[t * L + k] = gl::add(part[t * L + k], part[(t + h) * L + k]);
        }
        __syncthreads();
    }
    if (t < L) out[((long long)blockIdx.y * out_stride + i) * L + t] = part[t];
}

// out[0] = shader cycles, out[1] = wall-clock ticks spent in a fixed spin (tf_debug_sclk_mhz)
__global__ void sclk_probe_kernel(unsigned long long* out) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned v = threadIdx.x;
    for (int i = 0; i < 200000; ++i) v = v * 1664525u + 1013904223u;
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0 + (v == 0xdeadbeefu);
        out[1] = w1 - w0;
    }
}

// Synthetic inputs for benches and tests: element i = BFieldElement::new(splitmix64(seed ^ i) mod p), raw
// Montgomery word -- counter-based, so any slice can be regenerated; the oracle's tfo_fill_random is the same sequence.
__global__ void __launch_bounds__(256) fill_random_kernel(u64* out, unsigned long long count, u64 seed, unsigned long long first) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        u64 z = (seed ^ (first + i)) + 0x9e3779b97f4a7c15ULL;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        z ^= z >> 31;
        if (z >= gl::P) z -= gl::P;          // z mod p (z < 2^64 < 2p)
        out[i] = gl::mont_mul(z, gl::R2);    // BFieldElement::new (b_field_element.rs:235-237)
    }
}

// out[k] = nodes[idx[k]] for digests (5 words): authentication structures from a device-resident tree
__global__ void __launch_bounds__(256) gather_digests_kernel(const u64* nodes, const unsigned long long* idx, long long count, u64* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * 5) return;
    const long long k = i / 5, w = i - 5 * k;
    out[i] = nodes[idx[k] * 5 + w];
}

}  // namespace tfk
