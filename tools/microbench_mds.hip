// microbench_mds.hip -- the Tip5 MDS layer (tip5/mod.rs:210-253) alone, one state per lane, two formulations:
//   A  the kernel's: plain circulant product on the two 32-bit halves of every word, 2 x 256 v_mad_u64_u32 whose addend absorbs
//      every accumulation, then one 85-bit recombination and reduction per output word
//   B  the shape of the reference's generated_function taken one CRT level deep, on three 22-bit limbs (the pre-additions
//      a_i +- a_{i+8} must stay below 32 bits, which two 32-bit halves do not): x^16 - 1 = (x^8 - 1)(x^8 + 1), the cyclic half
//      computed once (64 mads) and used as the addend of both signed negacyclic chains (2 x 64 v_mad_i64_i32) -- no post-additions
//      at all, 3 x 192 = 576 mads + limb split + pre-additions
//   C  (round 5) the FP64 matrix pipe: 4 lanes per permutation (lane l: hash column j = l & 15, quarter q = l >> 4, state words
//      4 i + q in register i), the two 32-bit halves of every word converted with v_cvt_f64_u32 and multiplied by the constant
//      circulant with 8 x v_mfma_f64_16x16x4_f64 per 16 permutations (products < 2^48, sums of 16 < 2^52: exact in f64; the
//      accumulator starts at 2^52 so the mantissa of the result IS the integer sum), then the same 85-bit recombination
// All are checked against 128-bit arithmetic first.  The question: does the Karatsuba / CRT shape beat 512
// multiply-adds on this ISA?  On gfx950 a 64-bit addition costs as much as a multiply-add (profiles/r03_instr_rates.txt), so
// every saved product that needs a post-addition is a wash; the variant that needs none needs a third limb.
//   D  (round 6) the i8 matrix pipe, v_mfma_i32_16x16x64_i8: ten byte planes, three signed digits of the matrix entries (below); adopted
//      in tip5_kernels.h (TF_TIP5_I8 = 1), so "permutation, matrix pipe" of this file measures whichever form the header is built with:
//      -DTF_TIP5_I8=0 gives the f64 yardstick of round 5, the default the product's round
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form [-DTF_TIP5_I8=0] -I twenty-first_amd/csrc -o tools/microbench_mds tools/microbench_mds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "gl64.h"
#include "tip5_kernels.h"  // the shipping lane-per-permutation round (tfk::tip5_permutation) and its constant block
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
using gl::u32;
using gl::u64;
typedef long long i64;

__host__ __device__ constexpr u32 mds_entry_hd(int i) {
    constexpr u32 col[16] = {61402, 1108, 28750, 33823, 7454, 43244, 53865, 12034, 56951, 27521, 41351, 40901, 12021, 59689, 26798, 17845};
    return col[i & 15];
}

// value (85 bits) = alo + ahi * 2^32  ->  canonical word   (the tail of tip5_round without the round constant)
__device__ __forceinline__ u64 fold85(u64 alo, u64 ahi) {
    unsigned c0, c1;
    const u32 w1 = __builtin_addc((u32)(alo >> 32), (u32)ahi, 0u, &c0);
    const u32 w2 = __builtin_addc((u32)(ahi >> 32), 0u, c0, &c1);
    const u64 l64 = ((u64)w1 << 32) | (u32)alo;
    const u64 t = (u64)w2 * 0xffffffffu + l64;
    const bool ca = t < l64;
    const u64 u = t + gl::EPS;
    const bool cb = u < t;
    return (ca | cb) ? u : t;
}

__device__ __forceinline__ void mds_a(u64 (&s)[16]) {
    u32 lo[16], hi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) lo[i] = (u32)s[i], hi[i] = (u32)(s[i] >> 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        u64 alo = 0, ahi = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const u32 m = mds_entry_hd(16 + r - c);
            alo += (u64)m * lo[c];
            ahi += (u64)m * hi[c];
        }
        s[r] = fold85(alo, ahi);
    }
}

// one 22-bit limb vector a[16] -> 2 * (circulant product)[16] as signed 64-bit sums (the halving is folded into the recombination)
__device__ __forceinline__ void crt_limb(const u32 (&a)[16], i64 (&out)[16]) {
    int ap[8], am[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ap[i] = (int)(a[i] + a[i + 8]), am[i] = (int)a[i] - (int)a[i + 8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        i64 p = 0;  // cyclic half: sum_c (M[k] + M[k+8]) a+[c], k = (r - c) mod 8
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = (r - c) & 7;
            p += (i64)(int)(mds_entry_hd(k) + mds_entry_hd(k + 8)) * ap[c];
        }
        i64 q0 = p, q1 = p;  // negacyclic half with both signs, accumulated straight onto the cyclic one
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = (r - c) & 7;
            const int mm = (int)mds_entry_hd(k) - (int)mds_entry_hd(k + 8);
            const int sg = (r - c) < 0 ? -mm : mm;
            q0 += (i64)sg * am[c];
            q1 -= (i64)sg * am[c];
        }
        out[r] = q0;
        out[r + 8] = q1;
    }
}

__device__ __forceinline__ void mds_b(u64 (&s)[16]) {
    u32 l0[16], l1[16], l2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const u32 lo = (u32)s[i], hi = (u32)(s[i] >> 32);
        l0[i] = lo & 0x3fffffu;
        l1[i] = ((lo >> 22) | (hi << 10)) & 0x3fffffu;
        l2[i] = hi >> 12;
    }
    i64 o0[16], o1[16], o2[16];
    crt_limb(l0, o0);
    crt_limb(l1, o1);
    crt_limb(l2, o2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // 2 y = o0 + 2^22 o1 + 2^44 o2  (all non-negative: they are twice the plain sums): y = (o0 >> 1) + 2^21 o1 + 2^43 o2, < 2^85
        const u64 a = (u64)o0[r] >> 1, b = (u64)o1[r], c = (u64)o2[r];  // o0 is even whenever o1, o2 are integers... recombine exactly below
        // exact: 2y is even; form the 86-bit value v = o0 + (o1 << 22) + (o2 << 44) and halve
        unsigned __int128 v = (unsigned __int128)(u64)o0[r] + ((unsigned __int128)b << 22) + ((unsigned __int128)c << 44);
        v >>= 1;
        (void)a;
        const u64 vlo = (u64)v, vhi = (u64)(v >> 64);  // vhi < 2^21
        const u64 t = vhi * 0xffffffffull + vlo;
        const bool ca = t < vlo;
        const u64 u = t + gl::EPS;
        const bool cb = u < t;
        s[r] = (ca | cb) ? u : t;
    }
}

// ---- variant C: FP64 matrix pipe ------------------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr u64 MASK52 = (1ull << 52) - 1;
__device__ __forceinline__ u64 dbits(double d) { return (u64)__double_as_longlong(d); }

// per-lane A operands: K-block i, lane (r = l & 15, k = l >> 4) holds M[(r - 4 i - k) mod 16]
__device__ __forceinline__ void mds_a_operands(double (&a)[4]) {
    const int l = threadIdx.x & 63, r = l & 15, k = l >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (r - 4 * i - k) & 15;
        u32 m = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) m = (e == t) ? mds_entry_hd(t) : m;
        a[i] = (double)m;
    }
}

// s[i] = word 4 i + q of the permutation in column j.  c_lo / c_hi: accumulator start (2^52, or 2^52 + a round-constant half).
__device__ __forceinline__ void mds_c(u64 (&s)[4], const double (&a)[4], d4 c_lo, d4 c_hi) {
    double lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[i] = (double)(u32)s[i], hi[i] = (double)(u32)(s[i] >> 32);
    d4 dlo = c_lo, dhi = c_hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dlo = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], lo[i], dlo, 0, 0, 0);
        dhi = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], hi[i], dhi, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) s[v] = fold85(dbits(dlo[v]) & MASK52, dbits(dhi[v]) & MASK52);
}

template <int V>
__global__ void __launch_bounds__(256) bench(u64* out, int iters, u64 seed) {
    if (V == 2) {
        u64 s[4];
        double a[4];
        mds_a_operands(a);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u64 z = seed + (u64)(blockIdx.x * 256 + threadIdx.x) * 4 + i;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            s[i] = z ^ (z >> 27);
        }
        const double two52 = 4503599627370496.0;
        const d4 c = {two52, two52, two52, two52};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) mds_c(s, a, c, c);
        out[blockIdx.x * 256 + threadIdx.x] = s[0] ^ s[1] ^ s[2] ^ s[3];
        return;
    }
    u64 s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        u64 z = seed + (u64)(blockIdx.x * 256 + threadIdx.x) * 16 + i;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        s[i] = z ^ (z >> 27);
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (V == 0) mds_a(s);
        else mds_b(s);
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= s[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void check(const u64* in, int* bad) {
    u64 a[16], b[16], x[16];
    for (int i = 0; i < 16; ++i) x[i] = a[i] = b[i] = in[(blockIdx.x * blockDim.x + threadIdx.x) * 16 + i];
    mds_a(a);
    mds_b(b);
    for (int r = 0; r < 16; ++r) {
        unsigned __int128 acc = 0;
        for (int c = 0; c < 16; ++c) acc += (unsigned __int128)mds_entry_hd(16 + r - c) * x[c];
        const u64 want = (u64)(acc % gl::P);
        if (a[r] != want) atomicOr(bad, 1);
        if (b[r] != want) atomicOr(bad, 2);
    }
}

// variant C against 128-bit arithmetic: wave w of the grid takes states 16 w .. 16 w + 15
__global__ void __launch_bounds__(256) check_c(const u64* in, int* bad) {
    const int l = threadIdx.x & 63, j = l & 15, q = l >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u64* x = in + (wave * 16 + j) * 16;
    u64 s[4];
    double a[4];
    mds_a_operands(a);
    for (int i = 0; i < 4; ++i) s[i] = x[4 * i + q];
    const double two52 = 4503599627370496.0;
    const d4 c = {two52, two52, two52, two52};
    mds_c(s, a, c, c);
    for (int v = 0; v < 4; ++v) {
        const int r = 4 * v + q;
        unsigned __int128 acc = 0;
        for (int cc = 0; cc < 16; ++cc) acc += (unsigned __int128)mds_entry_hd(16 + r - cc) * x[cc];
        if (s[v] != (u64)(acc % gl::P)) atomicOr(bad, 4);
    }
}


// ---- variant D (round 6): the i8 matrix pipe, v_mfma_i32_16x16x64_i8 ---------------------------------------------------------------
// Same lane layout as C (lane (j, q) holds words 4 i + q of the permutation in column j).  The state is taken byte by byte, the matrix
// entries as three SIGNED base-256 digits (M = m0 + 256 m1 + 65536 m2, |m| <= 128), and output plane p = a + b collects
//     P_p[r][j] = sum_c sum_{a <= 2} m_a[(r - c) mod 16] * d_{p - a}[c][j],     d_b = byte b of the word MINUS 128 (the byte XOR 0x80 as i8),
// one MFMA per plane with K = (c, a): 48 of the 64 products used, |P_p| < 2^20.  MDS(state)[r] = sum_p 256^p P_p + 128 * rowsum * (2^64-1)/255.
//   B operand of plane p, lane (j, q), dword i: bytes [d_p, d_{p-1}, d_{p-2}, 0] of word 4 i + q -- ONE v_perm_b32 with a constant selector
//     on the word XOR 0x80..80 (two v_xor per word and MDS);
//   A operand: digit a of M[(pi(r') - (4 i + q)) mod 16] at byte 4 i + a, the SAME for all ten planes (four VGPRs, constant); the row
//     permutation pi(r') = 4 (r' & 3) + (r' >> 2) makes D row r' (lane (j, r' >> 2), register r' & 3) the word 4 t + q of THIS lane's
//     layout, so nothing moves between lanes (A and B index K by the same function of (lane >> 4, byte), whatever it is);
//   C operand: 2^21 + byte p of (rc + K1 - K2) per plane (from LDS): every plane comes back non-negative and the round constant is free.
// Recombination: five v_lshl_add_u32 pair the planes (16-bit steps), two v_mad_u64_u32 build t0 = A0 + 2^16 A1, t1 = A2 + 2^16 A3, then
//   value = t0 + 2^32 t1 + 2^64 A4 = t0 + 2^32 lo(t1) + (2^32 - 1) (hi(t1) + A4)  (mod p): one add, one v_mad_u64_u32, one carry add, fix.
using tfk::v4i;
__host__ __device__ constexpr int i8_digit(u32 M, int a) {
    int m0 = (int)(signed char)(M & 0xff);
    u32 M1 = (u32)((int)M - m0) >> 8;
    int m1 = (int)(signed char)(M1 & 0xff);
    u32 M2 = (u32)((int)M1 - m1) >> 8;
    return a == 0 ? m0 : (a == 1 ? m1 : (int)M2);
}
__device__ __forceinline__ v4i i8_a_operand() {
    const int l = threadIdx.x & 63, rp = l & 15, qa = l >> 4, r = 4 * (rp & 3) + (rp >> 2);
    v4i a;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (r - (4 * i + qa)) & 15;
        u32 M = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) M = (e == t) ? mds_entry_hd(t) : M;
        const u32 b0 = (u32)i8_digit(M, 0) & 0xff, b1 = (u32)i8_digit(M, 1) & 0xff, b2 = (u32)i8_digit(M, 2) & 0xff;
        a[i] = (int)(b0 | (b1 << 8) | (b2 << 16));
    }
    return a;
}
constexpr u64 I8_ROWSUM = 524757;  // sum of MDS_MATRIX_FIRST_COLUMN
__host__ __device__ inline u64 i8_adjust(u64 rc) {  // (rc + K1 - K2) mod p
    const unsigned __int128 ones8 = 0x0101010101010101ULL;
    const u64 K1 = (u64)(((unsigned __int128)128 * I8_ROWSUM % gl::P) * (ones8 % gl::P) % gl::P);
    unsigned __int128 k2 = 0;
    for (int p = 9; p >= 0; --p) k2 = (k2 * 256 + ((unsigned __int128)1 << 21)) % gl::P;
    const u64 K2 = (u64)k2;
    u64 x = (u64)(((unsigned __int128)rc + K1 + gl::P - K2) % gl::P);
    return x;
}
struct I8Lds {
    int c[5][10][4][4];  // accumulator starts [round][plane][q][t]: 2^21 + byte `plane` of i8_adjust(rc[round][4 t + q])
};
template <int P>
__device__ __forceinline__ u32 i8_window(u32 lo, u32 hi) {  // bytes [P, P-1, P-2, zero] of the 64-bit word hi:lo
    constexpr u32 s0 = (P >= 0 && P <= 7) ? (u32)P : 0x0cu, s1 = (P - 1 >= 0 && P - 1 <= 7) ? (u32)(P - 1) : 0x0cu,
                  s2 = (P - 2 >= 0 && P - 2 <= 7) ? (u32)(P - 2) : 0x0cu;
    return __builtin_amdgcn_perm(hi, lo, s0 | (s1 << 8) | (s2 << 16) | (0x0cu << 24));
}
__device__ __forceinline__ u64 i8_mad(u32 a, u32 b, u64 c) {  // a * b + c, ONE v_mad_u64_u32 (b in an SGPR: no VOP3 literals on gfx950)
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(b), "v"(c));
    return d;
}
template <bool CANON>
__device__ __forceinline__ u64 i8_fold(const u32 (&Q)[10]) {
    // L0 = Q0 + 2^8 Q1 + 2^16 Q2 + 2^24 Q3 (< 2^47), L1 likewise from Q4..Q7, L2 = Q8 + 2^8 Q9 (< 2^31)
    const u32 A0 = (Q[1] << 8) + Q[0];  // (< 2^31)
    u64 L0 = i8_mad(Q[2], 1u << 16, (u64)A0);
    L0 = i8_mad(Q[3], 1u << 24, L0);
    const u32 A2 = (Q[5] << 8) + Q[4];
    u64 L1 = i8_mad(Q[6], 1u << 16, (u64)A2);
    L1 = i8_mad(Q[7], 1u << 24, L1);
    const u32 L2 = (Q[9] << 8) + Q[8];
    const u32 h = (u32)(L1 >> 32) + L2;                      // < 2^15 + 2^31
    const u64 u = i8_mad(h, 0xffffffffu, L0);                // < 2^64 (h (2^32 - 1) < 2^63.1, L0 < 2^47)
    // r = u + (lo32(L1) << 32); a carry out of bit 64 is worth 2^32 - 1
    u32 rl = (u32)u, rh;
    u64 k, n;
    if constexpr (CANON) {
        u64 e;
        asm("v_add_co_u32_e64 %[rh], %[k], %[uh], %[l1]\n\t"
            "v_cmp_ne_u32_e64 %[n], 0, %[rl]\n\t"
            "v_cmp_eq_u32_e64 %[e], -1, %[rh]\n\t"
            "s_and_b64 %[e], %[e], %[n]\n\t"                               // value >= p
            "s_or_b64 %[e], %[e], %[k]\n\t"                                // ... or carry: add 2^32 - 1
            "v_subbrev_co_u32_e64 %[rl], %[n], 0, %[rl], %[e]\n\t"         // lo -= cond, borrow n
            "s_andn2_b64 %[e], %[e], %[n]\n\t"
            "v_addc_co_u32_e64 %[rh], %[n], 0, %[rh], %[e]"                // hi += cond & ~borrow
            : [rh] "=&v"(rh), [rl] "+v"(rl), [k] "=&s"(k), [n] "=&s"(n), [e] "=&s"(e)
            : [uh] "v"((u32)(u >> 32)), [l1] "v"((u32)L1)
            : "scc");
    } else {
        asm("v_add_co_u32_e64 %[rh], %[k], %[uh], %[l1]\n\t"
            "s_nop 0\n\t"
            "v_subbrev_co_u32_e64 %[rl], %[n], 0, %[rl], %[k]\n\t"         // lo -= k, borrow n
            "s_andn2_b64 %[k], %[k], %[n]\n\t"
            "v_addc_co_u32_e64 %[rh], %[n], 0, %[rh], %[k]"                // hi += k & ~borrow
            : [rh] "=&v"(rh), [rl] "+v"(rl), [k] "=&s"(k), [n] "=&s"(n)
            : [uh] "v"((u32)(u >> 32)), [l1] "v"((u32)L1)
            : "scc");
    }
    return ((u64)rh << 32) | rl;
}
// s[i] = word 4 i + q; cp -> the ten accumulator starts of this lane's quarter ([plane][q][t], stride 16 ints per plane)
template <bool CANON0, bool CANON>
__device__ __forceinline__ void mds_i8(u64 (&s)[4], const v4i a, const int* cp) {
    u32 lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[i] = (u32)s[i] ^ 0x80808080u, hi[i] = (u32)(s[i] >> 32) ^ 0x80808080u;
    v4i d[10];
    const auto plane = [&](auto pc) {
        constexpr int P = decltype(pc)::value;
        v4i b;
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = (int)i8_window<P>(lo[i], hi[i]);
        const v4i c = *reinterpret_cast<const v4i*>(cp + P * 16);
        d[P] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    };
    plane(std::integral_constant<int, 0>{}); plane(std::integral_constant<int, 1>{}); plane(std::integral_constant<int, 2>{});
    plane(std::integral_constant<int, 3>{}); plane(std::integral_constant<int, 4>{}); plane(std::integral_constant<int, 5>{});
    plane(std::integral_constant<int, 6>{}); plane(std::integral_constant<int, 7>{}); plane(std::integral_constant<int, 8>{});
    plane(std::integral_constant<int, 9>{});
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        u32 Q[10];
#pragma unroll
        for (int p = 0; p < 10; ++p) Q[p] = (u32)d[p][t];
        s[t] = (t == 0) ? i8_fold<CANON0>(Q) : i8_fold<CANON>(Q);
    }
}
__device__ __forceinline__ void i8_stage(I8Lds* l, const u64* rc_mont /* 80 words or null (zero constants) */) {
    for (int i = threadIdx.x; i < 5 * 10 * 16; i += blockDim.x) {
        const int t = i & 3, q = (i >> 2) & 3, p = (i >> 4) % 10, round = i / 160;
        const u64 adj = i8_adjust(rc_mont ? rc_mont[round * 16 + 4 * t + q] : 0);
        (&l->c[0][0][0][0])[i] = (1 << 21) + (p < 8 ? (int)((adj >> (8 * p)) & 0xff) : 0);
    }
    __syncthreads();
}
// variant D against 128-bit arithmetic (no round constant): wave w of the grid takes states 16 w .. 16 w + 15
__global__ void __launch_bounds__(256) check_d(const u64* in, int* bad) {
    __shared__ I8Lds lds;
    i8_stage(&lds, nullptr);
    const int l = threadIdx.x & 63, j = l & 15, q = l >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u64* x = in + (wave * 16 + j) * 16;
    u64 s[4], s2[4];
    const v4i a = i8_a_operand();
    for (int i = 0; i < 4; ++i) s[i] = s2[i] = x[4 * i + q];
    mds_i8<true, true>(s, a, &lds.c[0][0][q][0]);
    mds_i8<true, false>(s2, a, &lds.c[0][0][q][0]);
    for (int v = 0; v < 4; ++v) {
        const int r = 4 * v + q;
        unsigned __int128 acc = 0;
        for (int cc = 0; cc < 16; ++cc) acc += (unsigned __int128)mds_entry_hd(16 + r - cc) * x[cc];
        if (s[v] != (u64)(acc % gl::P)) atomicOr(bad, 32);
        if (s2[v] % gl::P != (u64)(acc % gl::P)) atomicOr(bad, 64);   // the lazy form: any representative
    }
}
__global__ void __launch_bounds__(256) bench_d(u64* out, int iters, u64 seed) {
    __shared__ I8Lds lds;
    i8_stage(&lds, nullptr);
    u64 s[4];
    const v4i a = i8_a_operand();
    const int q = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u64 z = seed + (u64)(blockIdx.x * 256 + threadIdx.x) * 4 + i;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        s[i] = z ^ (z >> 27);
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) mds_i8<true, false>(s, a, &lds.c[0][0][q][0]);
    out[blockIdx.x * 256 + threadIdx.x] = s[0] ^ s[1] ^ s[2] ^ s[3];
}
// whole permutations with the i8 MDS: the S-box layer of tip5_round_mx (tip5_kernels.h), NS = 1
template <bool LAST>
__device__ __forceinline__ void tip5_round_i8(u64 (&s)[4], int round, const unsigned char* lut, const I8Lds* l, const v4i a, int q) {
    {
        const u32 lo = tfk::lookup4((u32)s[0], lut), hi = tfk::lookup4((u32)(s[0] >> 32), lut);
        s[0] = ((u64)hi << 32) | lo;
    }
    {
        u64 x[3] = {s[1], s[2], s[3]}, sq[3], qu[3], t[3];
        gl::mont_mul3(x, x, sq);
        gl::mont_mul3(sq, sq, qu);
        gl::mont_mul3(sq, qu, t);
        gl::mont_mul3(x, t, x);
        s[1] = x[0], s[2] = x[1], s[3] = x[2];
    }
    mds_i8<true, LAST>(s, a, &l->c[round][0][q][0]);
}
__global__ void __launch_bounds__(256) perm_i8_kernel(u64* states, long long count, int reps, const u64* rc_mont) {
    __shared__ I8Lds lds;
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    tfk::stage_lut(lut);
    i8_stage(&lds, rc_mont);
    const int l = threadIdx.x & 63, j = l & 15, q = l >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave * 16 >= count) return;
    const v4i a = i8_a_operand();
    u64 s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = states[(wave * 16 + j) * 16 + 4 * i + q];
#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 1
        for (int r = 0; r < 4; ++r) tip5_round_i8<false>(s, r, lut, &lds, a, q);
        tip5_round_i8<true>(s, 4, lut, &lds, a, q);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) states[(wave * 16 + j) * 16 + 4 * i + q] = s[i];
}

// ---- raw rates: v_mfma_f64_16x16x4_f64 (independent / dependent accumulators) and v_cvt_f64_u32 -----------------------
template <int CHAINS>
__global__ void __launch_bounds__(256) mfma_rate(double* out, int iters) {
    d4 acc[CHAINS];
    const double a = 1.0 + threadIdx.x, b = 0.5;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0, 0, 0, 0};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / CHAINS * 2; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double t = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) t += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

__global__ void __launch_bounds__(256) cvt_rate(double* out, int iters, u32 seed) {
    u32 x[12];
    double d[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = seed * (i + 3) + threadIdx.x, d[i] = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(x[i]));
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) t += d[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

// mont_mul3 on operands that are NOT canonical (any 64-bit words): the result must be congruent to a b 2^-64 -- what the lazy
// recombination of the matrix-pipe round relies on (tip5_kernels.h, mx_fold4<false>)
__global__ void __launch_bounds__(256) check_mm3(const u64* in, int* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a[3], b[3], r[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = in[i * 16 + k] | 0xffffffff00000000ULL;       // >= p unless the low word is 0
        b[k] = (k == 1) ? a[k] : (in[i * 16 + 3 + k] | ((i & 1) ? 0xffffffff00000000ULL : 0));  // a square, and mixed pairs
    }
    gl::mont_mul3(a, b, r);
    for (int k = 0; k < 3; ++k) {
        const unsigned __int128 prod = (unsigned __int128)(a[k] % gl::P) * (b[k] % gl::P);
        // want = prod * 2^-64 mod p; check r * 2^64 == prod (mod p)
        const u64 lhs = (u64)((((unsigned __int128)(r[k] % gl::P)) << 64) % gl::P), rhs = (u64)(prod % gl::P);
        if (lhs != rhs) atomicOr(bad, 16);
    }
}

// ---- whole permutations: the lane-per-permutation round against the matrix-pipe round of tip5_kernels.h --------------------
// NS = permutations per lane quartet (1: 16 per wave, 2: 32 per wave).
// states: count x 16 words.  A block of 256 threads = 4 waves; a wave takes 16 NS consecutive states.
template <int NS>
__global__ void __launch_bounds__(256) perm_mx_kernel(u64* states, long long count, int reps) {
    __shared__ __attribute__((aligned(32))) tfk::Tip5MxLds lds;
    tfk::stage_mx(&lds);
    const int l = threadIdx.x & 63, j = l & 15, q = l >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave * 16 * NS >= count) return;
    tfk::MxA a;
    tfk::mx_a_operands(&lds, a);
    u64 s[4 * NS];
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[4 * n + i] = states[((wave * NS + n) * 16 + j) * 16 + 4 * i + q];
#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) tfk::tip5_permutation_mx<NS>(s, &lds, a, q);
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) states[((wave * NS + n) * 16 + j) * 16 + 4 * i + q] = s[4 * n + i];
}

__global__ void __launch_bounds__(256) perm_a_kernel(u64* states, long long count, int reps) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    tfk::stage_lut(lut);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 s[16];
    u64* p = states + i * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = p[k];
#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) tfk::tip5_permutation(s, lut);
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = s[k];
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int n = 1 << 14;
    std::vector<u64> h(n * 16);
    u64 st = 99;
    for (auto& v : h) { st += 0x9e3779b97f4a7c15ULL; u64 z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; v = z ^ (z >> 31); }
    for (int i = 0; i < 64; ++i) h[i] = (i & 1) ? 0xffffffffffffffffULL : 0xffffffff00000000ULL;  // extreme words (S-box outputs may exceed p)
    for (int i = 64; i < 128; ++i) h[i] = 0xffffffffffffffffULL;                                    // the largest half-sums
    u64* d_in;
    int* d_bad;
    CK(hipMalloc(&d_in, h.size() * 8));
    CK(hipMalloc(&d_bad, 4));
    CK(hipMemset(d_bad, 0, 4));
    CK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad);
    hipLaunchKernelGGL(check_c, dim3(n / 64), dim3(256), 0, 0, d_in, d_bad);
    hipLaunchKernelGGL(check_d, dim3(n / 64), dim3(256), 0, 0, d_in, d_bad);
    hipLaunchKernelGGL(check_mm3, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad);
    int bad = 0;
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("MDS of %d random / extreme states (and mont_mul3 on words >= p) against 128-bit arithmetic: %s (mask %d: 1 = halves, 2 = three-limb CRT, 4 = f64 MFMA, 16 = mont_mul3 on non-canonical operands, 32 / 64 = i8 MFMA canonical / lazy)\n", n,
           bad ? "MISMATCH" : "all bit-exact", bad);
    u64* d_out;
    CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));
    const int iters = 400, grid = cus * 8;
    const char* names[4] = {"A two 32-bit halves, plain circulant (512 mads)", "B three 22-bit limbs, one CRT level (576 mads)",
                            "C f64 MFMA, 4 lanes per state (8 mfma / 16 states)", "D i8 MFMA, 4 lanes per state (10 mfma / 16 states)"};
    for (int v = 0; v < 4; ++v) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(bench<0>, dim3(grid), dim3(256), 0, 0, d_out, iters, 7ull);
            else if (v == 1) hipLaunchKernelGGL(bench<1>, dim3(grid), dim3(256), 0, 0, d_out, iters, 7ull);
            else if (v == 2) hipLaunchKernelGGL(bench<2>, dim3(grid), dim3(256), 0, 0, d_out, iters * 4, 7ull);
            else hipLaunchKernelGGL(bench_d, dim3(grid), dim3(256), 0, 0, d_out, iters * 4, 7ull);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        // variant C: a thread is a quarter state and runs 4 x the iterations: the same number of MDS layers per launch
        printf("%-52s: %8.3f ms for %d x %d MDS layers = %7.2f G MDS/s\n", names[v], ms, grid * 256, iters,
               (double)grid * 256 * iters / (ms * 1e-3) / 1e9);
    }
    // raw rates
    double* d_d = reinterpret_cast<double*>(d_out);
    for (int w : {1, 2, 4, 8}) {
        const int g = cus * w, it = 2000;
        float ms[3];
        for (int ch = 0; ch < 3; ++ch) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0));
                CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                if (ch == 0) hipLaunchKernelGGL(mfma_rate<1>, dim3(g), dim3(256), 0, 0, d_d, it);
                else if (ch == 1) hipLaunchKernelGGL(mfma_rate<2>, dim3(g), dim3(256), 0, 0, d_d, it);
                else hipLaunchKernelGGL(mfma_rate<4>, dim3(g), dim3(256), 0, 0, d_d, it);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms[ch], e0, e1));
            }
        }
        // 16 mfma per iteration per wave; w waves per SIMD
        printf("v_mfma_f64_16x16x4_f64, %d waves/SIMD: ", w);
        for (int ch = 0; ch < 3; ++ch) {
            const double per_simd = 16.0 * it * w / (ms[ch] * 1e-3);  // mfma / s / SIMD
            printf(" %d chain(s) %6.2f M mfma/s/SIMD (%5.1f ns each, %5.1f TFLOP/s chip)", 1 << ch, per_simd / 1e6, 1e9 / per_simd,
                   per_simd * cus * 4 * 2048 / 1e12);
        }
        printf("\n");
    }
    for (int w : {4, 8}) {
        const int g = cus * w, it = 3000;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(cvt_rate, dim3(g), dim3(256), 0, 0, d_d, it, 3u);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("v_cvt_f64_u32, %d waves/SIMD: %6.3f G wave-instr/s/SIMD\n", w, 48.0 * it * w / (ms * 1e-3) / 1e9);
    }
    // whole permutations (arbitrary canonical round constants: the two kernels must agree word for word)
    {
        tfk::Tip5Consts c;
        u64 z = 12345;
        for (int i = 0; i < 80; ++i) { z = z * 6364136223846793005ULL + 1442695040888963407ULL; c.rc[i] = z % gl::P; }
        CK(hipMemcpyToSymbol(HIP_SYMBOL(tfk::g_tip5), &c, sizeof(c)));
        const long long count = 1ll << 22;
        std::vector<u64> hs((size_t)count * 16);
        for (auto& v : hs) { st += 0x9e3779b97f4a7c15ULL; u64 y = st; y = (y ^ (y >> 30)) * 0xbf58476d1ce4e5b9ULL; y = (y ^ (y >> 27)) * 0x94d049bb133111ebULL; v = (y ^ (y >> 31)) % gl::P; }
        u64 *d_a, *d_c;
        CK(hipMalloc(&d_a, hs.size() * 8));
        CK(hipMalloc(&d_c, hs.size() * 8));
        CK(hipMemcpy(d_a, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_c, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
        tfk::Tip5MxConsts mx;
        tfk::fill_tip5_mx(mx, c.rc);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(tfk::g_tip5_mx), &mx, sizeof(mx)));
        std::vector<u64> ra(hs.size()), rx(hs.size());
        hipLaunchKernelGGL(perm_a_kernel, dim3(count / 256), dim3(256), 0, 0, d_a, count, 3);
        CK(hipMemcpy(ra.data(), d_a, hs.size() * 8, hipMemcpyDeviceToHost));
        for (int ns = 1; ns <= 2; ++ns) {
            CK(hipMemcpy(d_c, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
            if (ns == 1) hipLaunchKernelGGL(perm_mx_kernel<1>, dim3(count / 64), dim3(256), 0, 0, d_c, count, 3);
            else hipLaunchKernelGGL(perm_mx_kernel<2>, dim3(count / 128), dim3(256), 0, 0, d_c, count, 3);
            CK(hipMemcpy(rx.data(), d_c, hs.size() * 8, hipMemcpyDeviceToHost));
            size_t diff = 0;
            for (size_t i = 0; i < hs.size(); ++i) diff += ra[i] != rx[i];
            printf("3 chained permutations of %lld states, matrix-pipe round (%d per lane quartet) against the lane-per-permutation round: %s (%zu words differ)\n",
                   count, ns, diff ? "MISMATCH" : "bit-exact", diff);
            if (diff) bad |= 8;
        }
        u64* d_rc;
        CK(hipMalloc(&d_rc, 80 * 8));
        CK(hipMemcpy(d_rc, c.rc, 80 * 8, hipMemcpyHostToDevice));
        {
            CK(hipMemcpy(d_c, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(perm_i8_kernel, dim3(count / 64), dim3(256), 0, 0, d_c, count, 3, d_rc);
            CK(hipMemcpy(rx.data(), d_c, hs.size() * 8, hipMemcpyDeviceToHost));
            size_t diff = 0;
            for (size_t i = 0; i < hs.size(); ++i) diff += ra[i] != rx[i];
            printf("3 chained permutations of %lld states, i8 matrix-pipe round against the lane-per-permutation round: %s (%zu words differ)\n", count,
                   diff ? "MISMATCH" : "bit-exact", diff);
            if (diff) bad |= 128;
        }
        const char* pn[4] = {"permutation, one lane per state (shipping through round 4)", "permutation, matrix pipe, 4 lanes x 1 state (8 mfma / round / wave)",
                             "permutation, matrix pipe, 4 lanes x 2 states (16 mfma / round / wave)", "permutation, i8 matrix pipe, 4 lanes x 1 state (10 mfma / round / wave)"};
        // reps = 40: one long launch (sustained clocks, set-up amortised); reps = 1: the shape of a Merkle level (set-up, loads and stores per permutation)
        for (int reps : {40, 1}) {
            for (int v = 0; v < 4; ++v) {
                float ms = 0, best = 1e30f;
                for (int rep = 0; rep < (reps == 1 ? 6 : 2); ++rep) {
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0));
                    CK(hipEventCreate(&e1));
                    CK(hipEventRecord(e0));
                    if (v == 0) hipLaunchKernelGGL(perm_a_kernel, dim3(count / 256), dim3(256), 0, 0, d_a, count, reps);
                    else if (v == 1) hipLaunchKernelGGL(perm_mx_kernel<1>, dim3(count / 64), dim3(256), 0, 0, d_c, count, reps);
                    else if (v == 2) hipLaunchKernelGGL(perm_mx_kernel<2>, dim3(count / 128), dim3(256), 0, 0, d_c, count, reps);
                    else hipLaunchKernelGGL(perm_i8_kernel, dim3(count / 64), dim3(256), 0, 0, d_c, count, reps, d_rc);
                    CK(hipEventRecord(e1));
                    CK(hipDeviceSynchronize());
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("%-72s: %8.3f ms for %lld x %d permutations = %7.3f G permutations/s\n", pn[v], best, count, reps, (double)count * reps / (best * 1e-3) / 1e9);
            }
        }
    }
    return bad ? 1 : 0;
}

