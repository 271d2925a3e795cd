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
// Kernels in this file and the lengths they serve (planner: run_ntt in tf_ntt.hip):
//   ntt_tiny_kernel    n <= 16                the reference's radix-2 sweeps, one transform per thread
//   ntt_rows32w_kernel n == 32                wave-private tiles of 64 (limb-)transforms through LDS, one per lane, no barrier
//   ntt_pass_kernel    32 <= n <= 1024        one pass, rows of T whole transforms per workgroup
//                      n > 2^14 (and XFE > 1024): 2-4 passes as described above; the R = 1024 instantiations
//                      (LAST1024: row-major load roles / column-major store roles, stores fused with the last radix-2
//                      level; R1024: constant P2) are the two kernels of the 2^20-point headline transform
//   ntt_block_kernel   2^11 <= n <= 2^14, BFE one workgroup owns a whole transform: radix 32 x 32 x P3 with two LDS
//                      exchanges, so these lengths cost one HBM pass instead of two
//   (the latency-shaped kernels for calls with little work -- ntt_lat_kernel, ntt_lat2_kernel, the tree level kernels: lat_kernels.h)
#pragma once

#include <type_traits>

#include "gl64.h"
#include "ntt_args.h"
#include "ntt_network.h"

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
                                             // (PRE4: the residue class q = bits 3-4 / bits 0-1)
    const u64* post_tw_u;                    // ntt_col2048_kernel: the UNSCALED inter-pass table (rows 64 q are read from it; post_tw may carry offset^b)
    const u64* pre4_stw;                     // PRE4 only: [2][32] Montgomery words w_128^(+-q i), q = 1, 3 (the per-slot part of w_4096^(q c))
};

// ---- buffer addressing for the R = 1024 instantiations --------------------------------------------------------------------
// A global_load/store takes ONE 64-bit address per lane; with 32 row slots per thread the compiler keeps 32 uniform bases and
// forms every address with a 64-bit VALU add (v_lshl_add_u64 / v_mad_u64_u32: ~100 VALU instructions per burst, 7 % of a pass).
// A buffer instruction adds  resource base (4 SGPRs, per workgroup) + 32-bit per-lane offset (ONE VGPR for all slots) + 32-bit
// uniform offset (one SGPR per slot)  in the address unit: no VALU work at all.  The planner only selects these instantiations
// when every slot offset plus thread offset fits 32 bits (fits_buffer_offsets in tf_ntt.hip); larger transforms take the
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

// (the radix-2^k networks with power-of-two twiddles, butterfly blocks and mul4_inplace: ntt_network.h)

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

// PRE4 (LAST1024, forward, no scaling): the pass is a 4096-point DFT per row, run as FOUR 1024-point workgroups per tile that share
//   their input.  One radix-4 decimation-in-frequency stage is fused into the load: with x_m = x[c + 1024 m],
//     class q:  z_q[c] = w_4096^(q c) * sum_m x_m w_4^(q m)   and the 1024-point DFT of z_q gives the outputs k = 4 k' + q
//     q = 0: (x_0 + x_2) + (x_1 + x_3)          q = 2: ((x_0 + x_2) - (x_1 + x_3)) w_64^i
//     q = 1: ((x_0 - x_2) + w_4 (x_1 - x_3)) w_128^i        q = 3: ((x_0 - x_2) - w_4 (x_1 - x_3)) w_128^(3 i)
//   where c = g + 32 i: w_4 and w_64^i are powers of two (shifts); w_128^(q i) is one Montgomery product per element for the odd
//   classes, its 32 words uniform per register slot (A.pre4_stw, scalar loads); w_4096^(q g) rides on the inner twiddle (inner_tw
//   holds the four tables, [4][32][32]).  This is the last pass of the 2^22-point plan 1024 x 4096 that fast_coset_evaluate takes:
//   its FIRST pass (where every coefficient is scaled by offset^j) is then an ordinary 1024-point column pass -- the 2048 x 2048
//   plan scales every coefficient twice there, once per half of a PRE2 pair.
#ifdef TF_AB_BUILD  // measured loss (7.9 vs 7.6 ms on BASELINE configs[3], profiles/r04_c4_plan_ab.txt): laboratory build only
// slots Q0 .. Q0+3 from their four quarter-row words; cls = q (uniform).  Every step adds or subtracts a CANONICAL word (a loaded
// word or a power-of-two product) to an accumulator that may be any representative (gl::add_lazy4 / sub_lazy4: four instructions, four
// slots round-robin, no wait states); what leaves is canonical: a power-of-two product, a Montgomery product (stw), or -- class 0 --
// the odd slots' sums made canonical (the second operands of the lazy network's first level, ntt_network.h), the even slots' left lazy.
template <bool INV, int Q0, int I = 0>
__device__ __forceinline__ void pre4_shift4(u64 (&x)[32], const u64 (&d)[4]) {  // x[Q] = d * w_64^i up to the sign the caller folded in
    x[Q0 + I] = gl::Pow2Mul<Pre2Slot<INV, Q0 + I>::E>::apply(d[I]);
    if constexpr (I + 1 < 4) pre4_shift4<INV, Q0, I + 1>(x, d);
}
template <bool INV, int Q0>
__device__ __forceinline__ void pre4_combine4(u64 (&x)[32], const u64 (&v0)[4], const u64 (&v1)[4], const u64 (&v2)[4], const u64 (&v3)[4], u32 cls,
                                              const u64* stw) {
    u64 t[4], u[4];
    if (cls == 0) {  // ((x_0 + x_2) + x_1) + x_3
        gl::add_lazy4(v0, v2, t);
        gl::add_lazy4(t, v1, u);
        gl::add_lazy4(u, v3, t);
        x[Q0] = t[0], x[Q0 + 1] = gl::add(t[1], 0), x[Q0 + 2] = t[2], x[Q0 + 3] = gl::add(t[3], 0);
    } else if (cls == 2) {  // ((x_0 + x_2) - x_1 - x_3) w_64^i; a slot whose power-of-two product comes back negated takes x_1 + x_3 - x_0 - x_2
        const u64 p0[4] = {Pre2Slot<INV, Q0>::neg ? v1[0] : v0[0], Pre2Slot<INV, Q0 + 1>::neg ? v1[1] : v0[1], Pre2Slot<INV, Q0 + 2>::neg ? v1[2] : v0[2],
                           Pre2Slot<INV, Q0 + 3>::neg ? v1[3] : v0[3]};
        const u64 p1[4] = {Pre2Slot<INV, Q0>::neg ? v3[0] : v2[0], Pre2Slot<INV, Q0 + 1>::neg ? v3[1] : v2[1], Pre2Slot<INV, Q0 + 2>::neg ? v3[2] : v2[2],
                           Pre2Slot<INV, Q0 + 3>::neg ? v3[3] : v2[3]};
        const u64 m0[4] = {Pre2Slot<INV, Q0>::neg ? v0[0] : v1[0], Pre2Slot<INV, Q0 + 1>::neg ? v0[1] : v1[1], Pre2Slot<INV, Q0 + 2>::neg ? v0[2] : v1[2],
                           Pre2Slot<INV, Q0 + 3>::neg ? v0[3] : v1[3]};
        const u64 m1[4] = {Pre2Slot<INV, Q0>::neg ? v2[0] : v3[0], Pre2Slot<INV, Q0 + 1>::neg ? v2[1] : v3[1], Pre2Slot<INV, Q0 + 2>::neg ? v2[2] : v3[2],
                           Pre2Slot<INV, Q0 + 3>::neg ? v2[3] : v3[3]};
        gl::add_lazy4(p0, p1, t);
        gl::sub_lazy4(t, m0, u);
        gl::sub_lazy4(u, m1, t);
        pre4_shift4<INV, Q0>(x, t);
    } else {  // ((x_0 - x_2) +- w_4 (x_1 - x_3)) w_128^(q i)
        constexpr int E4 = TwExp<INV, 2, 1>::value;  // w_4^(+-1)
        gl::sub_lazy4(v0, v2, t);
        gl::sub_lazy4(v1, v3, u);
        const u64 w[4] = {gl::Pow2Mul<E4>::apply(u[0]), gl::Pow2Mul<E4>::apply(u[1]), gl::Pow2Mul<E4>::apply(u[2]), gl::Pow2Mul<E4>::apply(u[3])};
        if ((cls == 1) != gl::Pow2Mul<E4>::negate) gl::add_lazy4(t, w, u);
        else gl::sub_lazy4(t, w, u);
        const u64 b4[4] = {stw[brev5(Q0)], stw[brev5(Q0 + 1)], stw[brev5(Q0 + 2)], stw[brev5(Q0 + 3)]};
        gl::mont_mul4(u, b4, t);
        x[Q0] = t[0], x[Q0 + 1] = t[1], x[Q0 + 2] = t[2], x[Q0 + 3] = t[3];
    }
}
#endif  // TF_AB_BUILD

template <bool INV, int SCALE, int MODE = 0, bool LAST1024 = false, bool R1024 = false, bool COL = false, bool PRE2 = false, bool PRE4 = false>
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
    static_assert(!PRE4 || (LAST1024 && !PRE2 && !INV && MODE == 0 && SCALE == 0), "PRE4 is a variant of the plain forward R = 1024 last pass");
#ifndef TF_AB_BUILD
    static_assert(!PRE4, "the radix-4 last pass is a measured loss: laboratory build only");
#endif
    // PRE2: two workgroups per tile; `half` selects the even (y) or odd (z) outputs.  PRE4: four, `half` is the residue class q
    u32 bid = blockIdx.x, half = 0;
    if constexpr (PRE2) {
        if (A.pre2_map == 1) half = (bid >> 3) & 1u, bid = (bid & 7u) | ((bid >> 4) << 3);
        else half = bid & 1u, bid >>= 1;
    }
    if constexpr (PRE4) {
        if (A.pre2_map == 1) half = (bid >> 3) & 3u, bid = (bid & 7u) | ((bid >> 5) << 3);
        else half = bid & 3u, bid >>= 2;
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
#ifndef TF_XCD_SKEW
#define TF_XCD_SKEW 0  // experiment (tools/build_variant.sh): XCD x takes column tile (x + SKEW * grp) mod 8 of column group grp
#endif
#ifndef TF_NAT_SKEW
#define TF_NAT_SKEW 0  // experiment: the natural order with the column tile rotated by SKEW * (outer index) within its group of 8
#endif
        i2 = ((grp << 3) | (TF_XCD_SKEW ? ((xcd + grp * TF_XCD_SKEW) & 7u) : xcd)) * G + within % G;
        const u32 rest = within / G;
        i1 = rest % A.d1;
        i0 = rest / A.d1;
    } else {
        const u32 tile = bid;
        i2 = tile % A.d2;
        const u32 rest = tile / A.d2;
        i1 = rest % A.d1;
        i0 = rest / A.d1;
        if (TF_NAT_SKEW && (A.d2 & 7u) == 0) i2 = (i2 & ~7u) | ((i2 + (i2 >> 3) * TF_NAT_SKEW + rest * TF_NAT_SKEW) & 7u);
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
    if constexpr (PRE2 || PRE4) out += (long long)half * A.pre2_out_off;

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
            const u64* itw = A.inner_tw + ((PRE2 || PRE4) ? (int)half * 1024 : 0);
            for (int i = t; i < 32 * P2; i += blockDim.x) ltw[(i >> 5) * kLdsTwStride + (i & 31)] = itw[i];
            __syncthreads();
        }
    }
    // ------------------------------------------------------------------ load + step 1 (radix 32 over i, rows g + P2*i)
    __builtin_amdgcn_s_setprio(TF_PRIO_LOAD);
    if constexpr (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 32; ++q) x[q] = (u64)(t * 32 + q) * 0x9e3779b97f4a7c15ULL >> 1;
#ifdef TF_AB_BUILD
    } else if (PRE4 && act_in) {
        // the tile's 64 Ki input elements, four slots at a time: the words c, c + 1024, c + 2048, c + 3072 of the row (c = g + 32 i),
        // combined into class `half`'s z[c]; default cache policy -- the three partner workgroups read the same lines out of the L2
        const u32 toff = (u32)(((long long)ch_in * A.in_cs_hi + cl_in + (long long)g_in * A.in_rs) * 8);
        const __amdgpu_buffer_rsrc_t ri = buf_rsrc(in);
        const u32 poff = (u32)(A.pre2_in_off * 8);
        const auto slot_off = [&](int q) { return (u32)((long long)(brev5(q) << p2) * A.in_rs * 8); };
        const auto ld = [&](u32 so) { return buf_load<0>(ri, toff, so); };
        const u64* stw = A.pre4_stw + (half >> 1) * 32;  // uniform: scalar loads
        const auto group = [&](auto q0c) {
            constexpr int Q0 = decltype(q0c)::value;
            u64 v0[4], v1[4], v2[4], v3[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 so = slot_off(Q0 + i);
                v0[i] = ld(so), v1[i] = ld(so + poff), v2[i] = ld(so + 2 * poff), v3[i] = ld(so + 3 * poff);
            }
            pre4_combine4<INV, Q0>(x, v0, v1, v2, v3, half, stw);
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 4>{});
        group(std::integral_constant<int, 8>{});
        group(std::integral_constant<int, 12>{});
        group(std::integral_constant<int, 16>{});
        group(std::integral_constant<int, 20>{});
        group(std::integral_constant<int, 24>{});
        group(std::integral_constant<int, 28>{});
#endif  // TF_AB_BUILD
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
#ifdef TF_AB_BUILD
                x[q] = (A.nt & 1) ? __builtin_nontemporal_load(ptr) : *ptr;  // uniform: the compiler emits the burst twice
#else
                x[q] = *ptr;  // (the run-time non-temporal variant, tf_set_ntt_nt, is a laboratory switch: TF_AB_BUILD)
#endif
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
        // tf_ntt.hip sizes the buffer.)
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

#ifdef TF_AB_BUILD  // measured loss (2.08 / 2.21 / 2.29 ms against 1.87, profiles/r03_chain_ab.txt): laboratory build only
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

#endif  // TF_AB_BUILD

// ---- a 2048-point COLUMN pass in ONE workgroup: 2048 rows x 8 word-columns (round 6, the first pass of the 2^21 / 2^22 plans) ----
// PRE2 runs a 2048-point pass as two 1024-point workgroups that both load (and, in a coset evaluation, both SCALE) the tile's
// 32 Ki elements.  Here the tile is 2048 rows x 8 word-columns = the same 16 Ki elements a 1024 x 16 tile holds, every element is
// loaded and scaled ONCE, and the extra radix-2 stage sits where the data crosses threads anyway -- on the reading side of the LDS
// exchange:
//   r = g + 64 i (g < 64, i < 32),  k = k1 + 32 k2 (k1 < 32, k2 < 64):   w_2048^(r k) = w_32^(i k1) * w_2048^(g k1) * w_64^(g k2)
//   step 1   thread (g, c): radix 32 over i in registers (shift-only network), times the inner twiddle w_2048^(g k1).  The [64][32]
//            table does not fit the LDS BESIDE the exchange buffer with two workgroups per CU, so it is staged INSIDE it: the table is
//            dead once the products are taken, one barrier separates them from the first exchange write.
//   exchange two rounds of 4 columns: element (k1, g, cc) at k1 * 260 + 4 g + cc.
//   step 2   thread (h, k1, c), h = the wave-uniform top thread bit: the 64-point DFT over g with ONE decimation-in-frequency
//            stage fused into the LDS reads -- a_g' = Y[g'] + Y[g' + 32] (h = 0: outputs k2 = 2 k2'),  (Y[g'] - Y[g' + 32]) w_64^g'
//            (h = 1: k2 = 2 k2' + 1) -- exactly the combination pre2_combine8 / pre2_shift make from global loads; both halves read
//            all 64 words (LDS traffic, not HBM / L2 traffic) and each does half of the butterfly, so no arithmetic is duplicated.
//            Then radix 32 over g' (shift-only), the inter-pass twiddle, and stores to rows k1 + 32 h + 64 k2'.
// Global segments are 64 bytes (8 word-columns): the planner's xcd_order = 2 dispatches the two tiles that share every 128-byte line
// back to back on one XCD.  Same words as every other plan (tests: tf_set_ntt_two_pass(3)).
#ifndef TF_C8_LOAD_AUX
#define TF_C8_LOAD_AUX 0  // default cache policy: the neighbouring tile's read of the other half of every 128-byte line should find it in the L2
#endif
#ifndef TF_C8_STORE_AUX
#define TF_C8_STORE_AUX TF_AUX_COL_STORE
#endif
#ifndef TF_C8_ABLATE
#define TF_C8_ABLATE 0  // measurement builds only (tools/build_variant.sh): 1 no scale-table loads, 2 no inter-pass table loads, 4 no arithmetic networks
#endif
constexpr int kC8Nc = 8, kC8Cpr = 4, kC8Rounds = 2, kC8S1 = 260;
// A workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence over every address space, i.e.
// s_waitcnt vmcnt(0): every global load in flight is waited for at the barrier -- which is what made table words requested ahead of the
// exchange a LOSS (8.6 ms against 7.7 on BASELINE configs[3] with sixteen words prefetched, profiles/r06_c4_cols8_ab.txt).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// The radix-2 stage on the words just read from the exchange buffer (canonical: Montgomery products), slots Q0 .. Q0+7: x = low word
// (g'), w = high word (g' + 32).  Odd half: x -+ w with the sign of the slot's power-of-two twiddle folded in, as lazy differences (the
// shift that follows takes any representative; slot 0 has no shift and is the first operand of a lazy butterfly).  Even half: the sums
// feed level 1 of a lazy network -- its first operands (even slots) lazy, its second operands (odd slots) canonical.
template <bool INV, int Q0>
__device__ __forceinline__ void c8_combine8(u64 (&x)[32], const u64 (&w)[8], bool odd) {
    if (odd) {
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            const bool n0 = h ? Pre2Slot<INV, Q0 + 4>::neg : Pre2Slot<INV, Q0>::neg, n1 = h ? Pre2Slot<INV, Q0 + 5>::neg : Pre2Slot<INV, Q0 + 1>::neg;
            const bool n2 = h ? Pre2Slot<INV, Q0 + 6>::neg : Pre2Slot<INV, Q0 + 2>::neg, n3 = h ? Pre2Slot<INV, Q0 + 7>::neg : Pre2Slot<INV, Q0 + 3>::neg;
            const u64 a4[4] = {n0 ? w[h] : x[Q0 + h], n1 ? w[h + 1] : x[Q0 + h + 1], n2 ? w[h + 2] : x[Q0 + h + 2], n3 ? w[h + 3] : x[Q0 + h + 3]};
            const u64 v4[4] = {n0 ? x[Q0 + h] : w[h], n1 ? x[Q0 + h + 1] : w[h + 1], n2 ? x[Q0 + h + 2] : w[h + 2], n3 ? x[Q0 + h + 3] : w[h + 3]};
            u64 r4[4];
            gl::sub_lazy4(a4, v4, r4);
            x[Q0 + h] = r4[0], x[Q0 + h + 1] = r4[1], x[Q0 + h + 2] = r4[2], x[Q0 + h + 3] = r4[3];
        }
    } else {
        const u64 a4[4] = {x[Q0], x[Q0 + 2], x[Q0 + 4], x[Q0 + 6]}, v4[4] = {w[0], w[2], w[4], w[6]};
        u64 r4[4];
        gl::add_lazy4(a4, v4, r4);
        x[Q0] = r4[0], x[Q0 + 2] = r4[1], x[Q0 + 4] = r4[2], x[Q0 + 6] = r4[3];
        x[Q0 + 1] = gl::add(x[Q0 + 1], w[1]), x[Q0 + 3] = gl::add(x[Q0 + 3], w[3]), x[Q0 + 5] = gl::add(x[Q0 + 5], w[5]), x[Q0 + 7] = gl::add(x[Q0 + 7], w[7]);
    }
}
#ifndef TF_C8_LDS_BARRIER
#define TF_C8_LDS_BARRIER 1
#endif
#if TF_C8_LDS_BARRIER
#define TF_C8_BARRIER lds_barrier
#else
#define TF_C8_BARRIER __syncthreads
#endif
template <bool INV, int SCALE>
__global__ void __launch_bounds__(512, TF_NTT_WAVES) ntt_col2048_kernel(const NttPassArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    static_assert(SCALE == 0 || (SCALE == 1 && !INV), "plain passes, or the scaling / zero-padding first pass of a coset evaluation");
    const int t = threadIdx.x;
    const int L = A.L;
    const int c = t & 7, g = t >> 3;          // step 1: rows g + 64 i of word-column c
    const int k1 = g & 31;                    // step 2: outputs k1 + 32 (2 k2' + half)
    const u32 half = (u32)t >> 8;             // wave-uniform
    u64* const ltw = lds;                     // the inner table [64][32], overlaid on the exchange buffer (dead before the first exchange write)
    u32 i0, i1, i2;
    {
        const u32 bid = blockIdx.x;
        if (A.xcd_order) {  // (see ntt_pass_kernel)
            const u32 G = (u32)A.xcd_order, xcd = bid & 7u, slot = bid >> 3, ngrp = A.d2 / (8u * G);
#ifndef TF_C8_XCD_SKEW
#define TF_C8_XCD_SKEW 3  // XCD x takes column pair (x + 3 grp) mod 8 of column group grp instead of pair x of every group: its tiles then spread over
                           // more of its L2's channels (measured 0.945-0.955 of the PRE2 time against 0.965-0.973 without, profiles/r06_c4_cols8_ab.txt)
#endif
#ifndef TF_C8_COLBLOCK
#define TF_C8_COLBLOCK 1  // column groups an XCD walks interleaved (their inter-pass table slices share its L2); must divide ngrp
#endif
            u32 grp, within;
            if (A.xcd_colfast) {
                grp = slot % ngrp, within = slot / ngrp;
            } else if (TF_C8_COLBLOCK > 1 && ngrp % TF_C8_COLBLOCK == 0) {
                const u32 per = G * A.d01 * TF_C8_COLBLOCK, blk = slot / per, r = slot % per;
                grp = blk * TF_C8_COLBLOCK + r % TF_C8_COLBLOCK, within = r / TF_C8_COLBLOCK;
            } else {
                grp = slot / (G * A.d01), within = slot % (G * A.d01);
            }
            i2 = ((grp << 3) | (TF_C8_XCD_SKEW ? ((xcd + grp * TF_C8_XCD_SKEW) & 7u) : xcd)) * G + within % G;
            const u32 rest = within / G;
            i1 = rest % A.d1;
            i0 = rest / A.d1;
        } else {
            i2 = bid % A.d2;
            const u32 rest = bid / A.d2;
            i1 = rest % A.d1;
            i0 = rest / A.d1;
        }
    }
    const u64* in = A.in + (long long)i0 * A.ib0 + (long long)i1 * A.ib1 + (long long)i2 * A.ib2;
    u64* out = A.out + (long long)i0 * A.ob0 + (long long)i1 * A.ob1 + (long long)i2 * A.ob2;
    const int col0 = (int)i2 * kC8Nc;
    const bool act = c < min(kC8Nc, A.col_limit - col0);
    const int ch = (int)div_by_L((u32)c, L), cl = c - ch * L;
    const long long bcol = (long long)div_by_L((u32)(col0 + c), L);
    // Coset scaling (SCALE 1, polynomial.rs:760-773): coefficient j = (g + 64 i) B + b is multiplied by offset^j = offset^(64 B i) *
    // offset^(B g) * offset^b.  The first factor is uniform per register slot (32 scalar loads of the power table), the second is
    // constant over a thread's 32 inputs, commutes with the radix-32 network over i and is multiplied into the staged inner table
    // (four products per thread), the third is constant along the whole column, commutes with the pass and comes with the inter-pass
    // table (the planner passes T[k B + b] * offset^b).  No per-element vector load of the power table is left: in this kernel they
    // were eight exposed L2 round trips per tile (measured: 0.57 ms of 7.95 on BASELINE configs[3], profiles/r06_c4_cols8_ab.txt).
    // Rows at or beyond n_coeffs are zero (the data buffer's record count), so a clamped table index there changes nothing.
    const u64* ps = nullptr;
    if constexpr (SCALE == 1) ps = A.pre_scale ? A.pre_scale + (long long)i1 * A.ps_i1 : nullptr;
    const auto ps_at = [&](long long j) { return ps[(A.n_coeffs < 0 || j < A.n_coeffs) ? j : 0]; };
    // (uniform index: through the constant address space these are scalar loads -- s_load_dwordx2 -- not 64 lanes reading one word)
    typedef const u64 __attribute__((address_space(4)))* cptr_t;
    const cptr_t cps = (cptr_t)(unsigned long long)ps;
    const auto ps_uniform = [&](long long j) { return cps[(A.n_coeffs < 0 || j < A.n_coeffs) ? j : 0]; };
    // ---- the inner table's words (4 per thread) are requested first, the tile's data right behind them
    u64 st[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) st[m] = A.inner_tw[t + 512 * m];
    if constexpr (SCALE == 1) {
        if (ps) {  // uniform
            u64 sg[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) sg[m] = ps_at((long long)((t + 512 * m) >> 5) * A.ps_rs);
            u64 r4[4];
            gl::mont_mul4(st, sg, r4);
#pragma unroll
            for (int m = 0; m < 4; ++m) st[m] = r4[m];
        }
    }
    u64 x[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = 0;
    // ------------------------------------------------------------------ load (+ scale) + step 1
    __builtin_amdgcn_s_setprio(TF_PRIO_LOAD);
    if (act) {
        const u32 toff = (u32)(((long long)ch * A.in_cs_hi + cl + (long long)g * A.in_rs) * 8);
        if constexpr (SCALE == 1) {
            u32 nrec = 0xffffffffu;
            if (A.n_coeffs >= 0) {  // rows beyond the coefficients read as zero: the buffer's record count does it
                const long long rem = (A.n_coeffs * L - (long long)(in - (A.in + (long long)i0 * A.ib0))) * 8;
                nrec = rem <= 0 ? 0u : (u32)min(rem, 0xffffffffll);
            }
            const __amdgpu_buffer_rsrc_t ri = buf_rsrc_n(in, nrec);
            // (the range check of a raw buffer covers the per-lane offset only: the whole offset goes through the VGPR)
#pragma unroll
            for (int q = 0; q < 32; ++q) x[q] = buf_load<TF_C8_LOAD_AUX>(ri, toff + (u32)((long long)(brev5(q) << 6) * A.in_rs * 8), 0);
        } else {
            const __amdgpu_buffer_rsrc_t ri = buf_rsrc(in);
#pragma unroll
            for (int q = 0; q < 32; ++q) x[q] = buf_load<TF_C8_LOAD_AUX>(ri, toff, (u32)((long long)(brev5(q) << 6) * A.in_rs * 8));
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) ltw[((t + 512 * m) >> 5) * kLdsTwStride + ((t + 512 * m) & 31)] = st[m];
    if constexpr (SCALE == 1) {
        if (ps) {  // uniform; slot q holds row g + 64 brev5(q): the uniform factor offset^(64 B brev5(q))
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 4) {
                const u64 u0 = ps_uniform((long long)(brev5(q0) << 6) * A.ps_rs), u1 = ps_uniform((long long)(brev5(q0 + 1) << 6) * A.ps_rs);
                const u64 u2 = ps_uniform((long long)(brev5(q0 + 2) << 6) * A.ps_rs), u3 = ps_uniform((long long)(brev5(q0 + 3) << 6) * A.ps_rs);
                mul4_inplace(x, q0, u0, u1, u2, u3);
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (!(TF_C8_ABLATE & 4)) {
        dit_half<INV, 0, true>(x);
        __builtin_amdgcn_sched_barrier(0);
        dit_half<INV, 16, true>(x);
        dit_level<INV, 5, true>(x);
    }
    TF_C8_BARRIER();  // the staged table is complete
    {
        const u64* tw = ltw + g * kLdsTwStride;
#pragma unroll
        for (int q = 0; q < 32; q += 4) mul4_inplace(x, q, tw[q], tw[q + 1], tw[q + 2], tw[q + 3]);
    }
    TF_C8_BARRIER();  // every wave has read its table rows: the exchange may overwrite them
    // the inter-pass table's first words are requested before the exchange: they arrive under it and under step 2
    const int krow = k1 + 32 * (int)half;  // slot q holds output row krow + 64 q
    const u32 twoff = (u32)(((long long)krow * A.tw_rs + bcol) * 8);
    const __amdgpu_buffer_rsrc_t rt = buf_rsrc(A.post_tw);
#ifndef TF_C8_UV
#define TF_C8_UV 0  // 1: T[(krow + 64 q) B + b] = T[krow B + b] * T[64 q B + b] -- ONE word per thread from its own row (with offset^b when the pass
                    //    scales) and 32 words from rows 64 q that every thread of the tile shares (32 lines per tile instead of 2 048 32-byte
                    //    segments), one more product per element.  MEASURED as a loss: 7.58-7.75 ms against 7.34-7.36 on BASELINE configs[3]
                    //    (profiles/r06_c4_cols8_ab.txt); 0: one table word per element
#endif
    const __amdgpu_buffer_rsrc_t rtu = buf_rsrc(A.post_tw_u ? A.post_tw_u : A.post_tw);
    const u32 uoff = (u32)(bcol * 8);
    const auto tw_load = [&](int q) {
        if constexpr (TF_C8_UV) return (TF_C8_ABLATE & 2) ? (u64)(q + 5) : buf_load_tab(rtu, uoff, (u32)((long long)(q << 6) * A.tw_rs * 8));
        else return (TF_C8_ABLATE & 2) ? (u64)(q + 5) : buf_load_tab(rt, twoff, (u32)((long long)(q << 6) * A.tw_rs * 8));
    };
    u64 vrow = 0;
    if constexpr (TF_C8_UV) {
        if (act) vrow = buf_load_tab(rt, twoff, 0);
    }
#ifndef TF_C8_PREFETCH
#define TF_C8_PREFETCH 0  // inter-pass table words requested ahead of step 2 (0 / 8 / 16): measured 7.60 / 7.75 / 8.04 ms on BASELINE configs[3] -- more of them in flight is SLOWER (profiles/r06_c4_cols8_ab.txt)
#endif
#ifndef TF_C8_PREFETCH_AT
#define TF_C8_PREFETCH_AT 1  // 0: request them before the exchange, 1: after it (in front of step 2's networks)
#endif
    u64 wa[8], wb[8];
    const auto prefetch = [&]() {
        if constexpr (TF_C8_PREFETCH >= 8) {
            if (act) {
#pragma unroll
                for (int i = 0; i < 8; ++i) wa[i] = tw_load(i);
            }
        }
        if constexpr (TF_C8_PREFETCH >= 16) {
            if (act) {
#pragma unroll
                for (int i = 0; i < 8; ++i) wb[i] = tw_load(8 + i);
            }
        }
    };
    if constexpr (TF_C8_PREFETCH_AT == 0) prefetch();
    // ------------------------------------------------------------------ exchange + the fused radix-2 stage
    {
        const int myround = c >> 2, cc = c & 3;
        u64* const wr = lds + g * kC8Cpr + cc;
        const u64* const rd = lds + k1 * kC8S1 + cc;
#pragma unroll 1
        for (int r = 0; r < kC8Rounds; ++r) {
            if (r) TF_C8_BARRIER();
            if (myround == r) {
#pragma unroll
                for (int q = 0; q < 32; ++q) wr[q * kC8S1] = x[q];
            }
            TF_C8_BARRIER();
            if (myround == r) {
                const auto group = [&](auto q0c) {
                    constexpr int Q0 = decltype(q0c)::value;
                    u64 w[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        x[Q0 + i] = rd[brev5(Q0 + i) * kC8Cpr];
                        w[i] = rd[(brev5(Q0 + i) + 32) * kC8Cpr];
                    }
                    c8_combine8<INV, Q0>(x, w, half != 0);
                };
                group(std::integral_constant<int, 0>{});
                group(std::integral_constant<int, 8>{});
                group(std::integral_constant<int, 16>{});
                group(std::integral_constant<int, 24>{});
            }
        }
    }
    if constexpr (TF_C8_PREFETCH_AT == 1) prefetch();
    if (half) pre2_shift<INV>(x);
    // ------------------------------------------------------------------ step 2 (radix 32 over g') + inter-pass twiddle + store
    __builtin_amdgcn_s_setprio(TF_PRIO_STEP2);
    if constexpr (!(TF_C8_ABLATE & 4)) {
        dit_level<INV, 1, true>(x);
        dit_level<INV, 2, true>(x);
        dit_level<INV, 3, true>(x);
        dit_level<INV, 4, true>(x);
        dit_level<INV, 5, true>(x);
    }
    if (act) {
        const u32 toff = (u32)(((long long)ch * A.out_cs_hi + cl + (long long)krow * A.out_rs) * 8);
        const __amdgpu_buffer_rsrc_t ro = buf_rsrc(out);
        // group G's products and stores; the table words of a later group are requested before them (two buffers, wa / wb)
        const auto mul_store = [&](int q0, const u64 (&w)[8]) {
#pragma unroll
            for (int i = 0; i < 8; i += 4) {
                const int q = q0 + i;
                const u64 a4[4] = {x[q], x[q + 1], x[q + 2], x[q + 3]}, b4[4] = {w[i], w[i + 1], w[i + 2], w[i + 3]};
                u64 r4[4];
                gl::mont_mul4(a4, b4, r4);
                if constexpr (TF_C8_UV) {
                    const u64 c4[4] = {r4[0], r4[1], r4[2], r4[3]}, v4[4] = {vrow, vrow, vrow, vrow};
                    gl::mont_mul4(c4, v4, r4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) buf_store<TF_C8_STORE_AUX>(ro, toff, (u32)((long long)((q + e) << 6) * A.out_rs * 8), r4[e]);
            }
        };
        if constexpr (TF_C8_PREFETCH < 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wa[i] = tw_load(i);
        }
        if constexpr (TF_C8_PREFETCH < 16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wb[i] = tw_load(8 + i);
        }
        mul_store(0, wa);
#pragma unroll
        for (int i = 0; i < 8; ++i) wa[i] = tw_load(16 + i);
        __builtin_amdgcn_sched_barrier(0);
        mul_store(8, wb);
#pragma unroll
        for (int i = 0; i < 8; ++i) wb[i] = tw_load(24 + i);
        __builtin_amdgcn_sched_barrier(0);
        mul_store(16, wa);
        __builtin_amdgcn_sched_barrier(0);
        mul_store(24, wb);
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

struct NttRows32Args {
    const u64* in;
    u64* out;
    long long total_transforms;  // transforms (XFE counts as one)
    u64 scale;                   // Montgomery 32^-1 for the inverse, 0 = none
    int L;
};

#ifdef TF_AB_BUILD
// ---- n = 32, contiguous transforms: one transform (one limb of it for XFE) per thread, staged through LDS ---------------
// A thread's 32 elements are contiguous in memory, so direct loads are 8-byte pieces 256 bytes apart (1.94 ms per 2^28 words).
// Here the workgroup streams its tile (512 BFE transforms or 170 XFE transforms = 510 limb-transforms) through LDS in two
// halves: coalesced loads into rows of pitch 33, each thread picks up its row, transforms it in registers, puts it back, and
// the tile leaves with coalesced stores.

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
#endif

// ---- n <= 64, wave-private tiles ----------------------------------------------------------------------------------------------
// A thread's 32 elements are contiguous in memory, so direct loads would be 8-byte pieces 256 bytes apart (1.94 ms per 2^28 words),
// and a workgroup that moves its tile through LDS in barrier-separated phases (load | pick up rows | transform | put rows back |
// store; round 2's ntt_rows32_kernel, kept in the laboratory build) leaves the memory pipe or the vector ALU idle in every phase:
// 1.21 ms per 2^28 words against the 0.87 ms of one coalesced round trip.  Here a WAVE owns its tile -- 64 BFieldElement transforms,
// or 21 XFieldElement transforms = 63 limb-transforms (a limb-transform is what one lane computes) -- and its own 64 x 33 words of
// LDS: no barrier anywhere, and the loads of the wave's next tile are issued before the current one is touched, so every wave keeps
// 16 KiB in flight while it computes.  Measured 0.867 ms per 2^28 words (profiles/r05_rows32_ab.txt): the floor.
template <int L>
__device__ __forceinline__ int rows32_slot(int w) {  // word w of a tile -> its place (row = limb-transform, column = element) in LDS
    if constexpr (L == 1) {
        return (w >> 5) * 33 + (w & 31);
    } else {
        const int el = (int)__umulhi((u32)w, 0x55555556u), limb = w - 3 * el;  // w / 3, w % 3 (w < 2^31 / 3)
        return ((el >> 5) * 3 + limb) * 33 + (el & 31);
    }
}

// n = 2^LOGN <= 32: a lane's 32 consecutive elements are 32 / n whole transforms -- levels 1 .. LOGN of the network over all 32
// register slots are exactly those transforms side by side (inputs bit-reversed within each group of n slots).
template <int LOGN>
__device__ __forceinline__ constexpr int rows32_src(int q) {  // slot q reads element (q / n) n + brev_LOGN(q % n) of the lane's row
    int r = 0;
    for (int b = 0; b < LOGN; ++b) r |= ((q >> b) & 1) << (LOGN - 1 - b);
    return (q >> LOGN << LOGN) | r;
}
template <bool INV, int LOGN, int LVL = 1>
__device__ __forceinline__ void rows32_network(u64 (&x)[32]) {
    if constexpr (LVL <= LOGN) {
        dit_level<INV, LVL, INV>(x);  // the inverse multiplies every word by n^-1 afterwards: lazy network
        rows32_network<INV, LOGN, LVL + 1>(x);
    }
}

// n = 64 (BFieldElement): two lanes per transform -- lane 2 t + h holds elements 32 h .. 32 h + 31 of transform t.  One DIF stage
// across the lane pair (a quad_perm DPP exchange: a' = a + b in the even lane, b' = (a - b) w_64^i in the odd one), then each lane
// runs the 32-point network on what it holds: the even lane comes out with X[2 k], the odd one with X[2 k + 1].
template <bool INV, int Q = 0>
__device__ __forceinline__ void rows64_dif_stage(u64 (&x)[32], bool odd) {
    if constexpr (Q < 32) {
        const u64 own = x[Q];
        const u32 ol = (u32)__builtin_amdgcn_mov_dpp((int)(u32)own, 0xB1, 0xf, 0xf, true);          // quad_perm [1, 0, 3, 2]: lane ^ 1
        const u32 oh = (u32)__builtin_amdgcn_mov_dpp((int)(u32)(own >> 32), 0xB1, 0xf, 0xf, true);
        const u64 other = ((u64)oh << 32) | ol;
        u64 sum, diff;
        gl::add_sub(odd ? other : own, odd ? own : other, sum, diff);
        x[Q] = odd ? gl::mul_pow2<TwExp<INV, 6, brev5(Q)>::value>(diff) : sum;  // slot Q holds element brev5(Q) of the lane's half
        rows64_dif_stage<INV, Q + 1>(x, odd);
    }
}

template <bool INV, int L, int LOGN>
__global__ void __launch_bounds__(256, 2) ntt_rows32w_kernel(const NttRows32Args A) {
    static_assert(LOGN <= 5 || (LOGN == 6 && L == 1), "n = 64: BFieldElement only");
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int kRows = L == 1 ? 64 : 21, kWords = kRows * 32 * L;  // rows of 32 elements and words per tile (2048 / 2016)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64* my = lds + wave * (64 * 33);
    const long long total_words = A.total_transforms * L << LOGN;
    const long long tiles = (total_words + kWords - 1) / kWords, stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= tiles) return;
    int slot[32];  // where word lane + 64 q of a tile goes (the same for every tile)
#pragma unroll
    for (int q = 0; q < 32; ++q) slot[q] = rows32_slot<L>(lane + 64 * q);
    u64 nxt[32];
    {
        const long long base = tile * kWords;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const long long w = base + lane + 64 * q;
            nxt[q] = (lane + 64 * q < kWords && w < total_words) ? A.in[w] : 0;
        }
    }
    for (; tile < tiles; tile += stride) {
        const long long base = tile * kWords;
        u64 x[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) x[q] = nxt[q];
        if (tile + stride < tiles) {
            const long long nb = (tile + stride) * kWords;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const long long w = nb + lane + 64 * q;
                nxt[q] = (lane + 64 * q < kWords && w < total_words) ? A.in[w] : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 32; ++q)
            if (L == 1 || lane + 64 * q < kWords) my[slot[q]] = x[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 32; ++q) x[q] = my[lane * 33 + rows32_src<LOGN < 5 ? LOGN : 5>(q)];  // (lane 63 of an XFieldElement tile transforms a stale row nobody reads)
        if constexpr (LOGN == 6) rows64_dif_stage<INV>(x, (lane & 1) != 0);
        rows32_network<INV, LOGN < 5 ? LOGN : 5>(x);
        if (INV) {
#pragma unroll
            for (int q = 0; q < 32; q += 4) mul4_inplace(x, q, A.scale, A.scale, A.scale, A.scale);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            if constexpr (LOGN == 6) my[((lane & ~1) + (q >> 4)) * 33 + 2 * (q & 15) + (lane & 1)] = x[q];  // X[2 q + h] of transform lane / 2
            else my[lane * 33 + q] = x[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const long long w = base + lane + 64 * q;
            if (lane + 64 * q < kWords && w < total_words) A.out[w] = my[slot[q]];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// (the latency-shaped kernels -- ntt_lat_kernel, ntt_lat2_kernel, the one-launch-per-level kernels of the zerofier-tree walks: lat_kernels.h, tf_lat.hip)

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

// out[k * B + b] = T[k * B + b] * S[b]: the inter-pass table with the column part offset^b of a coset evaluation's scaling folded in
// (ntt_col2048_kernel; B a power of two)
__global__ void __launch_bounds__(256) scale_post_tw_kernel(u64* out, const u64* T, const u64* S, long long B, long long M) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= M) return;
    out[id] = gl::mont_mul(T[id], S[id & (B - 1)]);
}

// out[c * n + j] = HI_c[j >> h] * LO_c[j & (2^h - 1)]   (base_c^j; grid.y = c; tabs = [c][nhi + nlo] split tables)
__global__ void __launch_bounds__(256) build_pow_tables_kernel(u64* out, const u64* tabs, int h, long long n, long long nhi, long long nlo) {
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const u64* hi = tabs + (long long)blockIdx.y * (nhi + nlo);
    const u64* lo = hi + nhi;
    out[(long long)blockIdx.y * n + id] = gl::mont_mul(hi[id >> h], lo[id & ((1ll << h) - 1)]);
}

// tab[i] = base^(i << shift), i < count, by square-and-multiply in every thread: the two split tables of a TEMPORARY inter-pass table
// (get_post_table: tables beyond the cache budget, transforms of 2^29 points and more), built on the caller's stream -- no host-built
// table, no upload, nothing waits; build_post_tw_kernel above then takes them as it takes the uploaded ones
__global__ void __launch_bounds__(256) build_split_powers_kernel(u64* tab, u64 base, int shift, long long count) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= count) return;
    u64 acc = gl::ONE, sq = base;
    for (int i = 0; i < shift; ++i) sq = gl::mont_mul(sq, sq);  // base^(2^shift)
    for (unsigned long long e = (unsigned long long)id; e; e >>= 1) {
        if (e & 1) acc = gl::mont_mul(acc, sq);
        sq = gl::mont_mul(sq, sq);
    }
    tab[id] = acc;
}

// out[c * n + j] = base_c^j by square-and-multiply in every thread (at most 2 log2 n products per word): the TEMPORARY power tables of
// get_pow_table, which must not leave the caller's stream (the split-table builder above uploads host-built tables and waits for them)
struct PowBases {
    u64 base[64];  // kMaxCosetSplit
};
__global__ void __launch_bounds__(256) build_pow_tables_direct_kernel(u64* out, const PowBases bases, long long n) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    u64 acc = gl::ONE, sq = bases.base[blockIdx.y];
    for (unsigned long long e = (unsigned long long)id; e; e >>= 1) {
        if (e & 1) acc = gl::mont_mul(acc, sq);
        sq = gl::mont_mul(sq, sq);
    }
    out[(long long)blockIdx.y * n + id] = acc;
}

}  // namespace tfk
