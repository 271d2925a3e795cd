// ntt_network.h -- the shift-only radix-2^k networks of the NTT kernels (device code shared by ntt_kernels.h and lat_kernels.h).
//
// 2 has order 192 in the Goldilocks field (2^96 = -1), so every twiddle inside a transform of up to 64 points is a power of two
// (b_field_element.rs:46-51: w_64 = 2^39, w_32 = 2^78, ..., w_2 = 2^96): a butterfly level costs shifts and carry chains
// (gl::Pow2Mul, gl::add_sub_lazy2) instead of Montgomery products.  Templates and inline device functions only.
#pragma once

#include "gl64.h"

namespace tfk {

using gl::u32;
using gl::u64;

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

}  // namespace tfk
