// gl64.h -- Goldilocks field (p = 2^64 - 2^32 + 1) arithmetic for gfx950 device code and for
// the host-side table builders of this library.
//
// Representation contract (reference: twenty-first/src/math/b_field_element.rs:84-86, :235-237):
// a BFieldElement is ONE u64 holding x * 2^64 mod p (Montgomery form), always canonical (< p).
// Every function below takes canonical inputs and returns canonical outputs unless it says
// otherwise, so any re-association of the exact field arithmetic is bit-identical to the
// reference.
//
//   add/sub      b_field_element.rs:711-732 / :773-795   (same function, different carry shape)
//   mont_mul     b_field_element.rs:755-762 + montyred :357-370
//   mul_pow2<K>  x * 2^K mod p -- replaces the general multiply for the twiddles inside a
//                radix-<=64 butterfly, which are all powers of two because 2^96 = -1 (mod p):
//                w_64 = 2^39, w_32 = 2^78, w_16 = 2^156, w_8 = 2^120, w_4 = 2^48, w_2 = 2^96
//                (the literals of b_field_element.rs:46-51).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 P = 0xffffffff00000001ULL;
constexpr u64 EPS = 0xffffffffULL;               // 2^32 - 1 = 2^64 mod p
constexpr u64 R2 = 0xfffffffe00000001ULL;        // 2^128 mod p (b_field_element.rs:229)
constexpr u64 ONE = 0xffffffffULL;               // Montgomery form of 1 (b_field_element.rs:707-709)

// a + b mod p.  Valid whenever the true sum is < 2^64 + p and the result is meant canonical:
// in particular for canonical a, b.
GL_HD u64 add(u64 a, u64 b) {
    u64 s = a + b;
    bool c = s < a;
    u64 t = s + EPS;  // s - p (mod 2^64)
    bool c2 = t < s;
    return (c | c2) ? t : s;
}

// a - b mod p for canonical a, b.
GL_HD u64 sub(u64 a, u64 b) {
    u64 d = a - b;
    return (a < b) ? d - EPS : d;
}

GL_HD u64 neg(u64 a) { return a ? P - a : 0; }

#if defined(__HIPCC__)
// (a, v) -> (a + v, a - v) for canonical a, v: ten VALU instructions on two interleaved carry chains.
//   n = p - v;  s = a - n;  d = a - v;  each difference gets +p on borrow as
//   lo += borrow (carry c), hi -= borrow & ~c   (p = 2^64 - 2^32 + 1: +1 on the low word, -1 on the high word),
// with the borrow masks combined on the scalar unit.  The s_nop's keep the two wait states gfx950 wants between a
// VALU writing a carry mask and a VALU reading it.
__device__ __forceinline__ void add_sub(u64 a, u64 v, u64& sum, u64& diff) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), v0 = (u32)v, v1 = (u32)(v >> 32);
    u32 s0, s1, d0, d1, n0, n1;
    u64 ma, mb;
    asm("v_sub_co_u32_e32 %4, vcc, 1, %10\n\t"
        "v_sub_co_u32_e64 %2, %6, %8, %10\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e32 %5, vcc, -1, %11, vcc\n\t"
        "v_subb_co_u32_e64 %3, %6, %9, %11, %6\n\t"
        "v_sub_co_u32_e32 %0, vcc, %8, %4\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %2, %7, 0, %2, %6\n\t"
        "v_subb_co_u32_e32 %1, vcc, %9, %5, vcc\n\t"
        "s_andn2_b64 %6, %6, %7\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %0, %7, 0, %0, vcc\n\t"
        "v_subbrev_co_u32_e64 %3, %6, 0, %3, %6\n\t"
        "s_andn2_b64 vcc, vcc, %7\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "=&v"(s0), "=&v"(s1), "=&v"(d0), "=&v"(d1), "=&v"(n0), "=&v"(n1), "=&s"(ma), "=&s"(mb)
        : "v"(a0), "v"(a1), "v"(v0), "v"(v1)
        : "vcc", "scc");
    sum = ((u64)s1 << 32) | s0;
    diff = ((u64)d1 << 32) | d0;
}
#endif

#if defined(__HIPCC__)
// ---- lazy (non-canonical) butterflies ------------------------------------------------------------------------------------
// Between the reductions of a radix-32 network a value may be ANY u64 congruent to the element (only the word
// that is finally stored has to be the canonical representative).  With  a  arbitrary and  v <= p:
//   s = a + v:  a 64-bit carry c means the true sum is s + 2^64 = s + EPS (mod p); s + EPS cannot carry again because
//               a + v - 2^64 < p (v <= p, a < 2^64), so s + EPS < p + EPS = 2^64.          s -= c; s_hi += c & ~borrow
//   d = a - v:  a borrow b means the true difference is d - 2^64 = d - EPS (mod p); d = a - v + 2^64 >= 2^64 - p = EPS because
//               v <= p, so d - EPS does not borrow again.                                  d_lo += b; d_hi -= b & ~carry
// Eight VALU instructions per butterfly instead of ten (no  n = p - v,  no canonical outputs).  TWO butterflies per block:
// their four carry chains are issued round-robin, so every carry mask is read at least three instructions after it was
// written and the block needs no wait-state s_nop (gfx950 wants two wait states between a VALU writing a carry mask and a VALU
// reading it).
__device__ __forceinline__ void add_sub_lazy2(u64 a, u64 v, u64 c, u64 w, u64& sum0, u64& diff0, u64& sum1, u64& diff1) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), v0 = (u32)v, v1 = (u32)(v >> 32);
    const u32 c0 = (u32)c, c1 = (u32)(c >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    u32 p0, p1, q0, q1, r0, r1, t0, t1;
    u64 mp, mq, mr, mt, np, nq, nr;  // m*: carry / borrow of the 64-bit operation, n*: of the correction's low word (chain t: vcc)
    asm("v_add_co_u32_e64 %[p0], %[mp], %[a0], %[v0]\n\t"               //  1 P1  p0 = a0 + v0
        "v_sub_co_u32_e64 %[q0], %[mq], %[a0], %[v0]\n\t"               //  2 Q1  q0 = a0 - v0
        "v_add_co_u32_e64 %[r0], %[mr], %[c0], %[w0]\n\t"               //  3 R1
        "v_sub_co_u32_e64 %[t0], %[mt], %[c0], %[w0]\n\t"               //  4 T1
        "v_addc_co_u32_e64 %[p1], %[mp], %[a1], %[v1], %[mp]\n\t"       //  5 P2  p1 = a1 + v1 + carry     -> mp = carry of a + v
        "v_subb_co_u32_e64 %[q1], %[mq], %[a1], %[v1], %[mq]\n\t"       //  6 Q2  q1 = a1 - v1 - borrow    -> mq = (a < v)
        "v_addc_co_u32_e64 %[r1], %[mr], %[c1], %[w1], %[mr]\n\t"       //  7 R2
        "v_subb_co_u32_e64 %[t1], %[mt], %[c1], %[w1], %[mt]\n\t"       //  8 T2
        "v_subbrev_co_u32_e64 %[p0], %[np], 0, %[p0], %[mp]\n\t"        //  9 P3  p0 -= carry              -> np = borrow
        "v_addc_co_u32_e64 %[q0], %[nq], 0, %[q0], %[mq]\n\t"           // 10 Q3  q0 += borrow             -> nq = carry
        "v_subbrev_co_u32_e64 %[r0], %[nr], 0, %[r0], %[mr]\n\t"        // 11 R3
        "v_addc_co_u32_e64 %[t0], vcc, 0, %[t0], %[mt]\n\t"             // 12 T3                           -> vcc = carry
        "s_andn2_b64 %[mp], %[mp], %[np]\n\t"
        "s_andn2_b64 %[mq], %[mq], %[nq]\n\t"
        "s_andn2_b64 %[mr], %[mr], %[nr]\n\t"
        "s_andn2_b64 %[mt], %[mt], vcc\n\t"
        "v_addc_co_u32_e64 %[p1], %[np], 0, %[p1], %[mp]\n\t"           // 13 P4  p1 += carry & ~borrow
        "v_subbrev_co_u32_e64 %[q1], %[nq], 0, %[q1], %[mq]\n\t"        // 14 Q4  q1 -= borrow & ~carry
        "v_addc_co_u32_e64 %[r1], %[nr], 0, %[r1], %[mr]\n\t"           // 15 R4
        "v_subbrev_co_u32_e64 %[t1], vcc, 0, %[t1], %[mt]"                // 16 T4
        : [p0] "=&v"(p0), [p1] "=&v"(p1), [q0] "=&v"(q0), [q1] "=&v"(q1), [r0] "=&v"(r0), [r1] "=&v"(r1), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [mp] "=&s"(mp), [mq] "=&s"(mq), [mr] "=&s"(mr), [mt] "=&s"(mt), [np] "=&s"(np), [nq] "=&s"(nq), [nr] "=&s"(nr)
        : [a0] "v"(a0), [a1] "v"(a1), [v0] "v"(v0), [v1] "v"(v1), [c0] "v"(c0), [c1] "v"(c1), [w0] "v"(w0), [w1] "v"(w1)
        : "vcc", "scc");
    sum0 = ((u64)p1 << 32) | p0;
    diff0 = ((u64)q1 << 32) | q0;
    sum1 = ((u64)r1 << 32) | r0;
    diff1 = ((u64)t1 << 32) | t0;
}

// FOUR independent lazy sums  r = a + v  (a any 64-bit word congruent to its element, v <= p; r any such word): the sum half of
// add_sub_lazy2, four VALU instructions each, the four carry chains issued round-robin (no wait-state nops).
__device__ __forceinline__ void add_lazy4(const u64 (&a)[4], const u64 (&v)[4], u64 (&r)[4]) {
    u32 lo[4], hi[4];
    u64 m[4], n[4];  // m: carry / borrow of the 64-bit operation, n: of the correction's low word
    asm("v_add_co_u32_e64 %[l0], %[m0], %[a0], %[v0]\n\t"
        "v_add_co_u32_e64 %[l1], %[m1], %[a1], %[v1]\n\t"
        "v_add_co_u32_e64 %[l2], %[m2], %[a2], %[v2]\n\t"
        "v_add_co_u32_e64 %[l3], %[m3], %[a3], %[v3]\n\t"
        "v_addc_co_u32_e64 %[h0], %[m0], %[b0], %[w0], %[m0]\n\t"
        "v_addc_co_u32_e64 %[h1], %[m1], %[b1], %[w1], %[m1]\n\t"
        "v_addc_co_u32_e64 %[h2], %[m2], %[b2], %[w2], %[m2]\n\t"
        "v_addc_co_u32_e64 %[h3], %[m3], %[b3], %[w3], %[m3]\n\t"
        "v_subbrev_co_u32_e64 %[l0], %[n0], 0, %[l0], %[m0]\n\t"
        "v_subbrev_co_u32_e64 %[l1], %[n1], 0, %[l1], %[m1]\n\t"
        "v_subbrev_co_u32_e64 %[l2], %[n2], 0, %[l2], %[m2]\n\t"
        "v_subbrev_co_u32_e64 %[l3], %[n3], 0, %[l3], %[m3]\n\t"
        "s_andn2_b64 %[m0], %[m0], %[n0]\n\t"
        "s_andn2_b64 %[m1], %[m1], %[n1]\n\t"
        "s_andn2_b64 %[m2], %[m2], %[n2]\n\t"
        "s_andn2_b64 %[m3], %[m3], %[n3]\n\t"
        "v_addc_co_u32_e64 %[h0], %[n0], 0, %[h0], %[m0]\n\t"
        "v_addc_co_u32_e64 %[h1], %[n1], 0, %[h1], %[m1]\n\t"
        "v_addc_co_u32_e64 %[h2], %[n2], 0, %[h2], %[m2]\n\t"
        "v_addc_co_u32_e64 %[h3], %[n3], 0, %[h3], %[m3]"
        : [l0] "=&v"(lo[0]), [h0] "=&v"(hi[0]), [m0] "=&s"(m[0]), [n0] "=&s"(n[0]), [l1] "=&v"(lo[1]), [h1] "=&v"(hi[1]), [m1] "=&s"(m[1]), [n1] "=&s"(n[1]), [l2] "=&v"(lo[2]), [h2] "=&v"(hi[2]), [m2] "=&s"(m[2]), [n2] "=&s"(n[2]), [l3] "=&v"(lo[3]), [h3] "=&v"(hi[3]), [m3] "=&s"(m[3]), [n3] "=&s"(n[3])
        : [a0] "v"((u32)a[0]), [b0] "v"((u32)(a[0] >> 32)), [v0] "v"((u32)v[0]), [w0] "v"((u32)(v[0] >> 32)), [a1] "v"((u32)a[1]), [b1] "v"((u32)(a[1] >> 32)), [v1] "v"((u32)v[1]), [w1] "v"((u32)(v[1] >> 32)), [a2] "v"((u32)a[2]), [b2] "v"((u32)(a[2] >> 32)), [v2] "v"((u32)v[2]), [w2] "v"((u32)(v[2] >> 32)), [a3] "v"((u32)a[3]), [b3] "v"((u32)(a[3] >> 32)), [v3] "v"((u32)v[3]), [w3] "v"((u32)(v[3] >> 32))
        : "scc");
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = ((u64)hi[i] << 32) | lo[i];
}

// FOUR independent lazy differences  r = a - v  (a any representative, v <= p): the difference half of add_sub_lazy2.
__device__ __forceinline__ void sub_lazy4(const u64 (&a)[4], const u64 (&v)[4], u64 (&r)[4]) {
    u32 lo[4], hi[4];
    u64 m[4], n[4];  // m: carry / borrow of the 64-bit operation, n: of the correction's low word
    asm("v_sub_co_u32_e64 %[l0], %[m0], %[a0], %[v0]\n\t"
        "v_sub_co_u32_e64 %[l1], %[m1], %[a1], %[v1]\n\t"
        "v_sub_co_u32_e64 %[l2], %[m2], %[a2], %[v2]\n\t"
        "v_sub_co_u32_e64 %[l3], %[m3], %[a3], %[v3]\n\t"
        "v_subb_co_u32_e64 %[h0], %[m0], %[b0], %[w0], %[m0]\n\t"
        "v_subb_co_u32_e64 %[h1], %[m1], %[b1], %[w1], %[m1]\n\t"
        "v_subb_co_u32_e64 %[h2], %[m2], %[b2], %[w2], %[m2]\n\t"
        "v_subb_co_u32_e64 %[h3], %[m3], %[b3], %[w3], %[m3]\n\t"
        "v_addc_co_u32_e64 %[l0], %[n0], 0, %[l0], %[m0]\n\t"
        "v_addc_co_u32_e64 %[l1], %[n1], 0, %[l1], %[m1]\n\t"
        "v_addc_co_u32_e64 %[l2], %[n2], 0, %[l2], %[m2]\n\t"
        "v_addc_co_u32_e64 %[l3], %[n3], 0, %[l3], %[m3]\n\t"
        "s_andn2_b64 %[m0], %[m0], %[n0]\n\t"
        "s_andn2_b64 %[m1], %[m1], %[n1]\n\t"
        "s_andn2_b64 %[m2], %[m2], %[n2]\n\t"
        "s_andn2_b64 %[m3], %[m3], %[n3]\n\t"
        "v_subbrev_co_u32_e64 %[h0], %[n0], 0, %[h0], %[m0]\n\t"
        "v_subbrev_co_u32_e64 %[h1], %[n1], 0, %[h1], %[m1]\n\t"
        "v_subbrev_co_u32_e64 %[h2], %[n2], 0, %[h2], %[m2]\n\t"
        "v_subbrev_co_u32_e64 %[h3], %[n3], 0, %[h3], %[m3]"
        : [l0] "=&v"(lo[0]), [h0] "=&v"(hi[0]), [m0] "=&s"(m[0]), [n0] "=&s"(n[0]), [l1] "=&v"(lo[1]), [h1] "=&v"(hi[1]), [m1] "=&s"(m[1]), [n1] "=&s"(n[1]), [l2] "=&v"(lo[2]), [h2] "=&v"(hi[2]), [m2] "=&s"(m[2]), [n2] "=&s"(n[2]), [l3] "=&v"(lo[3]), [h3] "=&v"(hi[3]), [m3] "=&s"(m[3]), [n3] "=&s"(n[3])
        : [a0] "v"((u32)a[0]), [b0] "v"((u32)(a[0] >> 32)), [v0] "v"((u32)v[0]), [w0] "v"((u32)(v[0] >> 32)), [a1] "v"((u32)a[1]), [b1] "v"((u32)(a[1] >> 32)), [v1] "v"((u32)v[1]), [w1] "v"((u32)(v[1] >> 32)), [a2] "v"((u32)a[2]), [b2] "v"((u32)(a[2] >> 32)), [v2] "v"((u32)v[2]), [w2] "v"((u32)(v[2] >> 32)), [a3] "v"((u32)a[3]), [b3] "v"((u32)(a[3] >> 32)), [v3] "v"((u32)v[3]), [w3] "v"((u32)(v[3] >> 32))
        : "scc");
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = ((u64)hi[i] << 32) | lo[i];
}

// The canonical butterfly of add_sub for TWO butterflies at once: canonical inputs and outputs, ten VALU instructions each
// as in add_sub, but the six carry chains (n = p - v, a - n, a - v, twice) are issued round-robin, so every carry mask is read
// at least three instructions after it was written and no wait-state s_nop is needed (add_sub spends three per butterfly).
__device__ __forceinline__ void add_sub2(u64 a, u64 v, u64 c, u64 w, u64& sum0, u64& diff0, u64& sum1, u64& diff1) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), v0 = (u32)v, v1 = (u32)(v >> 32);
    const u32 c0 = (u32)c, c1 = (u32)(c >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    u32 p0, p1, q0, q1, r0, r1, t0, t1, n0, n1, o0, o1;
    u64 mA, mA2, mB, mB2, mK, mK2;
    asm("v_sub_co_u32_e64 %[n0], %[mA], 1, %[v0]\n\t"                   //  1 N1   n0 = 1 - v0
        "v_sub_co_u32_e64 %[o0], %[mA2], 1, %[w0]\n\t"                  //  2 N'1
        "v_sub_co_u32_e64 %[q0], %[mB], %[a0], %[v0]\n\t"               //  3 D1   q0 = a0 - v0
        "v_sub_co_u32_e64 %[t0], %[mB2], %[c0], %[w0]\n\t"              //  4 D'1
        "v_subb_co_u32_e64 %[n1], %[mA], -1, %[v1], %[mA]\n\t"          //  5 N2   n1 = 0xffffffff - v1 - borrow
        "v_subb_co_u32_e64 %[o1], %[mA2], -1, %[w1], %[mA2]\n\t"        //  6 N'2
        "v_subb_co_u32_e64 %[q1], %[mB], %[a1], %[v1], %[mB]\n\t"       //  7 D2   q1 = a1 - v1 - borrow   -> mB = (a < v)
        "v_subb_co_u32_e64 %[t1], %[mB2], %[c1], %[w1], %[mB2]\n\t"     //  8 D'2
        "v_sub_co_u32_e64 %[p0], %[mA], %[a0], %[n0]\n\t"               //  9 S1   p0 = a0 - n0
        "v_sub_co_u32_e64 %[r0], %[mA2], %[c0], %[o0]\n\t"              // 10 S'1
        "v_addc_co_u32_e64 %[q0], %[mK], 0, %[q0], %[mB]\n\t"           // 11 D3   q0 += borrow           -> carry mK
        "v_addc_co_u32_e64 %[t0], %[mK2], 0, %[t0], %[mB2]\n\t"         // 12 D'3
        "v_subb_co_u32_e64 %[p1], %[mA], %[a1], %[n1], %[mA]\n\t"       // 13 S2   p1 = a1 - n1 - borrow   -> mA = (a < n)
        "v_subb_co_u32_e64 %[r1], %[mA2], %[c1], %[o1], %[mA2]\n\t"     // 14 S'2
        "s_andn2_b64 %[mB], %[mB], %[mK]\n\t"
        "s_andn2_b64 %[mB2], %[mB2], %[mK2]\n\t"
        "v_addc_co_u32_e64 %[p0], %[mK], 0, %[p0], %[mA]\n\t"           // 15 S3   p0 += borrow           -> carry mK
        "v_addc_co_u32_e64 %[r0], %[mK2], 0, %[r0], %[mA2]\n\t"         // 16 S'3
        "v_subbrev_co_u32_e64 %[q1], vcc, 0, %[q1], %[mB]\n\t"          // 17 D4   q1 -= borrow & ~carry
        "v_subbrev_co_u32_e64 %[t1], vcc, 0, %[t1], %[mB2]\n\t"         // 18 D'4
        "s_andn2_b64 %[mA], %[mA], %[mK]\n\t"
        "s_andn2_b64 %[mA2], %[mA2], %[mK2]\n\t"
        "v_subbrev_co_u32_e64 %[p1], vcc, 0, %[p1], %[mA]\n\t"          // 19 S4
        "v_subbrev_co_u32_e64 %[r1], vcc, 0, %[r1], %[mA2]"               // 20 S'4
        : [p0] "=&v"(p0), [p1] "=&v"(p1), [q0] "=&v"(q0), [q1] "=&v"(q1), [r0] "=&v"(r0), [r1] "=&v"(r1), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [n0] "=&v"(n0), [n1] "=&v"(n1), [o0] "=&v"(o0), [o1] "=&v"(o1), [mA] "=&s"(mA), [mA2] "=&s"(mA2), [mB] "=&s"(mB),
          [mB2] "=&s"(mB2), [mK] "=&s"(mK), [mK2] "=&s"(mK2)
        : [a0] "v"(a0), [a1] "v"(a1), [v0] "v"(v0), [v1] "v"(v1), [c0] "v"(c0), [c1] "v"(c1), [w0] "v"(w0), [w1] "v"(w1)
        : "vcc", "scc");
    sum0 = ((u64)p1 << 32) | p0;
    diff0 = ((u64)q1 << 32) | q0;
    sum1 = ((u64)r1 << 32) | r0;
    diff1 = ((u64)t1 << 32) | t0;
}
#endif

GL_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// Montgomery reduction of the 128-bit value hi*2^64 + lo (b_field_element.rs:357-370).
//   a = lo + (lo << 32) (carry e);  b = a - (a >> 32) - e;  r = hi - b, +p on borrow.
// Written on 32-bit limbs with carry builtins so that hipcc emits one carry chain
// (v_add_co / v_subb_co / v_subbrev_co / v_sub_co / v_subb_co, then one select): 8 VALU instructions.
GL_HD u64 montyred(u64 lo, u64 hi) {
#if defined(__clang__)
    const u32 l0 = (u32)lo, l1 = (u32)(lo >> 32);
    unsigned e, c1, c2, c3, unused;
    const u32 a1 = __builtin_addc(l1, l0, 0u, &e);      // high word of a; its low word is l0
    const u32 b0 = __builtin_subc(l0, a1, e, &c1);      // b = (a1:l0) - a1 - e
    const u32 b1 = __builtin_subc(a1, 0u, c1, &unused);
    const u32 r0 = __builtin_subc((u32)hi, b0, 0u, &c2);
    const u32 r1 = __builtin_subc((u32)(hi >> 32), b1, c2, &c3);
    const u64 r = ((u64)r1 << 32) | r0;
    return c3 ? r + P : r;
#else
    u64 a = lo + (lo << 32);
    u64 e = a < lo;
    u64 b = a - (a >> 32) - e;
    u64 r = hi - b;
    return (hi < b) ? r - EPS : r;
#endif
}

// 64 x 64 -> 128 schoolbook product on 32-bit limbs: four v_mad_u64_u32 (32 x 32 + 64) and one 64-bit add.
GL_HD void mul_wide(u64 a, u64 b, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0;
    const u64 p1 = (u64)a0 * b1 + (p0 >> 32);
    const u64 p2 = (u64)a1 * b0 + (u32)p1;
    hi = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
    lo = (p2 << 32) | (u32)p0;
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    lo = (u64)p;
    hi = (u64)(p >> 64);
#endif
}

// Montgomery product: a * b * 2^-64 mod p  (b_field_element.rs:755-762).  18 VALU instructions on gfx950.
GL_HD u64 mont_mul(u64 a, u64 b) {
    u64 lo, hi;
    mul_wide(a, b, lo, hi);
    return montyred(lo, hi);
}

#if defined(__HIPCC__)
// Two independent Montgomery products, 15 VALU instructions each (mont_mul above compiles to 18: the compiler needs a
// v_mov per 32-bit addend of v_mad_u64_u32 and a compare + two selects for the final "+ p").  Per product:
//   p0 = a0 b0, q = a1 b0, hh = a1 b1               three v_mad_u64_u32 with a zero addend
//   m  = a0 b1 + q, carry cm                         one v_mad_u64_u32, its carry-out kept in an SGPR pair
//   (l1, h0, h1) = the upper three words of p0 + (m << 32) + (hh << 64) + (cm << 96)      four carry-chain adds
//   Montgomery reduction of (h1:h0:l1:p0l) as in montyred, the final "+ p" done as  r0 += c, r1 -= c & ~carry.
// The two carry chains are interleaved (one on vcc, one on an SGPR pair) so that, with one s_nop per step, every
// carry is read at least two wait states after it was written.
#ifndef TF_MONT_CC
#define TF_MONT_CC 0  // 1 (A/B build): the compiler's 18-instruction product instead of the hand-scheduled pair
#endif
__device__ __forceinline__ void mont_mul2(u64 a, u64 b, u64 c, u64 d, u64& ab, u64& cd) {
#if TF_MONT_CC
    ab = mont_mul(a, b);
    cd = mont_mul(c, d);
    return;
#endif
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u32 c0 = (u32)c, c1 = (u32)(c >> 32), d0 = (u32)d, d1 = (u32)(d >> 32);
    const u64 xp = (u64)a0 * b0, xq = (u64)a1 * b0, xh = (u64)a1 * b1;
    const u64 yp = (u64)c0 * d0, yq = (u64)c1 * d0, yh = (u64)c1 * d1;
    u64 xm, xc, ym, yc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(xm), "=s"(xc) : "v"(a0), "v"(b1), "v"(xq));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(ym), "=s"(yc) : "v"(c0), "v"(d1), "v"(yq));
    u32 xr0, xr1, xu, xw, yr0, yr1, yu, yw;
    u64 st, su, sy;
    asm("v_add_co_u32_e32 %2, vcc, %12, %13\n\t"            // X1  l1 = p0h + ml
        "v_add_co_u32_e64 %6, %10, %19, %20\n\t"            // Y1
        "s_nop 0\n\t"
        "v_addc_co_u32_e32 %0, vcc, %15, %14, vcc\n\t"      // X2  h0 = hl + mh + c
        "v_addc_co_u32_e64 %4, %10, %22, %21, %10\n\t"      // Y2
        "s_nop 0\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %16, vcc\n\t"        // X3  h1 = hh + c
        "v_addc_co_u32_e64 %5, %10, 0, %23, %10\n\t"        // Y3
        "v_addc_co_u32_e64 %1, %8, 0, %1, %17\n\t"          // X4  h1 += cm
        "v_addc_co_u32_e64 %5, %9, 0, %5, %24\n\t"          // Y4
        "v_add_co_u32_e32 %2, vcc, %2, %11\n\t"             // X5  a1 = l1 + l0
        "v_add_co_u32_e64 %6, %10, %6, %18\n\t"             // Y5
        "s_nop 0\n\t"
        "v_subb_co_u32_e32 %3, vcc, %11, %2, vcc\n\t"       // X6  b0 = l0 - a1 - e
        "v_subb_co_u32_e64 %7, %10, %18, %6, %10\n\t"       // Y6
        "s_nop 0\n\t"
        "v_subbrev_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"      // X7  b1 = a1 - borrow
        "v_subbrev_co_u32_e64 %6, %10, 0, %6, %10\n\t"      // Y7
        "v_sub_co_u32_e32 %0, vcc, %0, %3\n\t"              // X8  r0 = h0 - b0
        "v_sub_co_u32_e64 %4, %10, %4, %7\n\t"              // Y8
        "s_nop 0\n\t"
        "v_subb_co_u32_e32 %1, vcc, %1, %2, vcc\n\t"        // X9  r1 = h1 - b1 - borrow
        "v_subb_co_u32_e64 %5, %10, %5, %6, %10\n\t"        // Y9
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %0, %8, 0, %0, vcc\n\t"          // X10 r0 += borrow
        "v_addc_co_u32_e64 %4, %9, 0, %4, %10\n\t"          // Y10
        "s_andn2_b64 vcc, vcc, %8\n\t"                      // X11
        "s_andn2_b64 %10, %10, %9\n\t"                      // Y11
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"      // X12 r1 -= borrow & ~carry
        "v_subbrev_co_u32_e64 %5, %10, 0, %5, %10"            // Y12
        : "=&v"(xr0), "=&v"(xr1), "=&v"(xu), "=&v"(xw), "=&v"(yr0), "=&v"(yr1), "=&v"(yu), "=&v"(yw), "=&s"(st), "=&s"(su),
          "=&s"(sy)
        : "v"((u32)xp), "v"((u32)(xp >> 32)), "v"((u32)xm), "v"((u32)(xm >> 32)), "v"((u32)xh), "v"((u32)(xh >> 32)), "s"(xc),
          "v"((u32)yp), "v"((u32)(yp >> 32)), "v"((u32)ym), "v"((u32)(ym >> 32)), "v"((u32)yh), "v"((u32)(yh >> 32)), "s"(yc)
        : "vcc", "scc");
    ab = ((u64)xr1 << 32) | xr0;
    cd = ((u64)yr1 << 32) | yr0;
}
#endif

#if defined(__HIPCC__)
// Four independent Montgomery products per block: the same fifteen VALU instructions per product as mont_mul2, but the four
// carry chains are issued round-robin, so every carry mask is read four instructions after it was written and the block needs
// none of the seven wait-state s_nop per pair that mont_mul2 spends.  Operand contract: ANY two 64-bit words are multiplied
// correctly -- montyred's subtrahend is < p whatever the 128-bit product, so the result is always a valid 64-bit representative of
// a b 2^-64 --; the result is CANONICAL when at least one operand is (the product is then < p 2^64 and montyred returns a word < p).
// The NTT kernels rely on the canonical case (lazy first operand x canonical twiddle), Tip5's x^7 on the general one (both operands
// any representative, result any representative; tools/microbench_mds.hip checks it on operands in [p, 2^64), mask 16).
#ifndef TF_MONT4
#define TF_MONT4 1
#endif
__device__ __forceinline__ void mont_mul4(const u64 (&a)[4], const u64 (&b)[4], u64 (&r)[4]) {
#if !TF_MONT4
    mont_mul2(a[0], b[0], a[1], b[1], r[0], r[1]);
    mont_mul2(a[2], b[2], a[3], b[3], r[2], r[3]);
    return;
#endif
    u64 p[4], h[4], m[4], cm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32 a0 = (u32)a[i], a1 = (u32)(a[i] >> 32), b0 = (u32)b[i], b1 = (u32)(b[i] >> 32);
        p[i] = (u64)a0 * b0;
        const u64 q = (u64)a1 * b0;
        h[i] = (u64)a1 * b1;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m[i]), "=s"(cm[i]) : "v"(a0), "v"(b1), "v"(q));
    }
    u32 r0[4], r1[4], u[4], w[4];
    u64 ca, cb, cc, ka, kb, kc, kd;  // carry masks of chains A..C (chain D: vcc) and the masks of the final correction
#define TF_M4_STEP(A, B, C, D) A "\n\t" B "\n\t" C "\n\t" D "\n\t"
    asm(TF_M4_STEP("v_add_co_u32_e64 %[ua], %[ca], %[pha], %[mla]", "v_add_co_u32_e64 %[ub], %[cb], %[phb], %[mlb]",       //  1  l1 = p0h + ml
                   "v_add_co_u32_e64 %[uc], %[cc], %[phc], %[mlc]", "v_add_co_u32_e32 %[ud], vcc, %[phd], %[mld]")
        TF_M4_STEP("v_addc_co_u32_e64 %[r0a], %[ca], %[hla], %[mha], %[ca]", "v_addc_co_u32_e64 %[r0b], %[cb], %[hlb], %[mhb], %[cb]",  //  2  h0 = hl + mh + c
                   "v_addc_co_u32_e64 %[r0c], %[cc], %[hlc], %[mhc], %[cc]", "v_addc_co_u32_e32 %[r0d], vcc, %[hld], %[mhd], vcc")
        TF_M4_STEP("v_addc_co_u32_e64 %[r1a], %[ca], 0, %[hha], %[ca]", "v_addc_co_u32_e64 %[r1b], %[cb], 0, %[hhb], %[cb]",    //  3  h1 = hh + c
                   "v_addc_co_u32_e64 %[r1c], %[cc], 0, %[hhc], %[cc]", "v_addc_co_u32_e32 %[r1d], vcc, 0, %[hhd], vcc")
        TF_M4_STEP("v_addc_co_u32_e64 %[r1a], %[ka], 0, %[r1a], %[cma]", "v_addc_co_u32_e64 %[r1b], %[kb], 0, %[r1b], %[cmb]",  //  4  h1 += carry of the middle term
                   "v_addc_co_u32_e64 %[r1c], %[kc], 0, %[r1c], %[cmc]", "v_addc_co_u32_e64 %[r1d], %[kd], 0, %[r1d], %[cmd]")
        TF_M4_STEP("v_add_co_u32_e64 %[ua], %[ca], %[ua], %[pla]", "v_add_co_u32_e64 %[ub], %[cb], %[ub], %[plb]",            //  5  a1 = l1 + l0
                   "v_add_co_u32_e64 %[uc], %[cc], %[uc], %[plc]", "v_add_co_u32_e32 %[ud], vcc, %[ud], %[pld]")
        TF_M4_STEP("v_subb_co_u32_e64 %[wa], %[ca], %[pla], %[ua], %[ca]", "v_subb_co_u32_e64 %[wb], %[cb], %[plb], %[ub], %[cb]",      //  6  b0 = l0 - a1 - e
                   "v_subb_co_u32_e64 %[wc], %[cc], %[plc], %[uc], %[cc]", "v_subb_co_u32_e32 %[wd], vcc, %[pld], %[ud], vcc")
        TF_M4_STEP("v_subbrev_co_u32_e64 %[ua], %[ca], 0, %[ua], %[ca]", "v_subbrev_co_u32_e64 %[ub], %[cb], 0, %[ub], %[cb]",  //  7  b1 = a1 - borrow
                   "v_subbrev_co_u32_e64 %[uc], %[cc], 0, %[uc], %[cc]", "v_subbrev_co_u32_e32 %[ud], vcc, 0, %[ud], vcc")
        TF_M4_STEP("v_sub_co_u32_e64 %[r0a], %[ca], %[r0a], %[wa]", "v_sub_co_u32_e64 %[r0b], %[cb], %[r0b], %[wb]",            //  8  r0 = h0 - b0
                   "v_sub_co_u32_e64 %[r0c], %[cc], %[r0c], %[wc]", "v_sub_co_u32_e32 %[r0d], vcc, %[r0d], %[wd]")
        TF_M4_STEP("v_subb_co_u32_e64 %[r1a], %[ca], %[r1a], %[ua], %[ca]", "v_subb_co_u32_e64 %[r1b], %[cb], %[r1b], %[ub], %[cb]",    //  9  r1 = h1 - b1 - borrow
                   "v_subb_co_u32_e64 %[r1c], %[cc], %[r1c], %[uc], %[cc]", "v_subb_co_u32_e32 %[r1d], vcc, %[r1d], %[ud], vcc")
        TF_M4_STEP("v_addc_co_u32_e64 %[r0a], %[ka], 0, %[r0a], %[ca]", "v_addc_co_u32_e64 %[r0b], %[kb], 0, %[r0b], %[cb]",    // 10  r0 += borrow (carry k)
                   "v_addc_co_u32_e64 %[r0c], %[kc], 0, %[r0c], %[cc]", "v_addc_co_u32_e64 %[r0d], %[kd], 0, %[r0d], vcc")
        TF_M4_STEP("s_andn2_b64 %[ca], %[ca], %[ka]", "s_andn2_b64 %[cb], %[cb], %[kb]", "s_andn2_b64 %[cc], %[cc], %[kc]", "s_andn2_b64 vcc, vcc, %[kd]")
        "v_subbrev_co_u32_e64 %[r1a], %[ka], 0, %[r1a], %[ca]\n\t"                                                                // 12  r1 -= borrow & ~k
        "v_subbrev_co_u32_e64 %[r1b], %[kb], 0, %[r1b], %[cb]\n\t"
        "v_subbrev_co_u32_e64 %[r1c], %[kc], 0, %[r1c], %[cc]\n\t"
        "v_subbrev_co_u32_e32 %[r1d], vcc, 0, %[r1d], vcc"
        : [r0a] "=&v"(r0[0]), [r1a] "=&v"(r1[0]), [ua] "=&v"(u[0]), [wa] "=&v"(w[0]), [r0b] "=&v"(r0[1]), [r1b] "=&v"(r1[1]), [ub] "=&v"(u[1]),
          [wb] "=&v"(w[1]), [r0c] "=&v"(r0[2]), [r1c] "=&v"(r1[2]), [uc] "=&v"(u[2]), [wc] "=&v"(w[2]), [r0d] "=&v"(r0[3]), [r1d] "=&v"(r1[3]),
          [ud] "=&v"(u[3]), [wd] "=&v"(w[3]), [ca] "=&s"(ca), [cb] "=&s"(cb), [cc] "=&s"(cc), [ka] "=&s"(ka), [kb] "=&s"(kb), [kc] "=&s"(kc),
          [kd] "=&s"(kd)
        : [pla] "v"((u32)p[0]), [pha] "v"((u32)(p[0] >> 32)), [mla] "v"((u32)m[0]), [mha] "v"((u32)(m[0] >> 32)), [hla] "v"((u32)h[0]),
          [hha] "v"((u32)(h[0] >> 32)), [cma] "s"(cm[0]), [plb] "v"((u32)p[1]), [phb] "v"((u32)(p[1] >> 32)), [mlb] "v"((u32)m[1]),
          [mhb] "v"((u32)(m[1] >> 32)), [hlb] "v"((u32)h[1]), [hhb] "v"((u32)(h[1] >> 32)), [cmb] "s"(cm[1]), [plc] "v"((u32)p[2]),
          [phc] "v"((u32)(p[2] >> 32)), [mlc] "v"((u32)m[2]), [mhc] "v"((u32)(m[2] >> 32)), [hlc] "v"((u32)h[2]), [hhc] "v"((u32)(h[2] >> 32)),
          [cmc] "s"(cm[2]), [pld] "v"((u32)p[3]), [phd] "v"((u32)(p[3] >> 32)), [mld] "v"((u32)m[3]), [mhd] "v"((u32)(m[3] >> 32)),
          [hld] "v"((u32)h[3]), [hhd] "v"((u32)(h[3] >> 32)), [cmd] "s"(cm[3])
        : "vcc", "scc");
#undef TF_M4_STEP
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = ((u64)r1[i] << 32) | r0[i];
}

// Three independent Montgomery products per block (the three x^7 words a lane owns in the matrix-pipe Tip5 layout): mont_mul4
// without its third chain.  Round-robin over three chains still puts two instructions between a carry mask's write and its read,
// which is what gfx950 wants, so this block needs no s_nop either.  Same operand contract as mont_mul4: any 64-bit operands, a valid
// representative out, canonical when one operand is canonical -- tip5_round_mx squares non-canonical words with it.
__device__ __forceinline__ void mont_mul3(const u64 (&a)[3], const u64 (&b)[3], u64 (&r)[3]) {
    u64 p[3], h[3], m[3], cm[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const u32 a0 = (u32)a[i], a1 = (u32)(a[i] >> 32), b0 = (u32)b[i], b1 = (u32)(b[i] >> 32);
        p[i] = (u64)a0 * b0;
        const u64 q = (u64)a1 * b0;
        h[i] = (u64)a1 * b1;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m[i]), "=s"(cm[i]) : "v"(a0), "v"(b1), "v"(q));
    }
    u32 r0[3], r1[3], u[3], w[3];
    u64 ca, cb, ka, kb, kd;  // carry masks of chains A, B (chain D: vcc) and the masks of the final correction
#define TF_M3_STEP(A, B, D) A "\n\t" B "\n\t" D "\n\t"
    asm(TF_M3_STEP("v_add_co_u32_e64 %[ua], %[ca], %[pha], %[mla]", "v_add_co_u32_e64 %[ub], %[cb], %[phb], %[mlb]",
                   "v_add_co_u32_e32 %[ud], vcc, %[phd], %[mld]")
        TF_M3_STEP("v_addc_co_u32_e64 %[r0a], %[ca], %[hla], %[mha], %[ca]", "v_addc_co_u32_e64 %[r0b], %[cb], %[hlb], %[mhb], %[cb]",
                   "v_addc_co_u32_e32 %[r0d], vcc, %[hld], %[mhd], vcc")
        TF_M3_STEP("v_addc_co_u32_e64 %[r1a], %[ca], 0, %[hha], %[ca]", "v_addc_co_u32_e64 %[r1b], %[cb], 0, %[hhb], %[cb]",
                   "v_addc_co_u32_e32 %[r1d], vcc, 0, %[hhd], vcc")
        TF_M3_STEP("v_addc_co_u32_e64 %[r1a], %[ka], 0, %[r1a], %[cma]", "v_addc_co_u32_e64 %[r1b], %[kb], 0, %[r1b], %[cmb]",
                   "v_addc_co_u32_e64 %[r1d], %[kd], 0, %[r1d], %[cmd]")
        TF_M3_STEP("v_add_co_u32_e64 %[ua], %[ca], %[ua], %[pla]", "v_add_co_u32_e64 %[ub], %[cb], %[ub], %[plb]",
                   "v_add_co_u32_e32 %[ud], vcc, %[ud], %[pld]")
        TF_M3_STEP("v_subb_co_u32_e64 %[wa], %[ca], %[pla], %[ua], %[ca]", "v_subb_co_u32_e64 %[wb], %[cb], %[plb], %[ub], %[cb]",
                   "v_subb_co_u32_e32 %[wd], vcc, %[pld], %[ud], vcc")
        TF_M3_STEP("v_subbrev_co_u32_e64 %[ua], %[ca], 0, %[ua], %[ca]", "v_subbrev_co_u32_e64 %[ub], %[cb], 0, %[ub], %[cb]",
                   "v_subbrev_co_u32_e32 %[ud], vcc, 0, %[ud], vcc")
        TF_M3_STEP("v_sub_co_u32_e64 %[r0a], %[ca], %[r0a], %[wa]", "v_sub_co_u32_e64 %[r0b], %[cb], %[r0b], %[wb]",
                   "v_sub_co_u32_e32 %[r0d], vcc, %[r0d], %[wd]")
        TF_M3_STEP("v_subb_co_u32_e64 %[r1a], %[ca], %[r1a], %[ua], %[ca]", "v_subb_co_u32_e64 %[r1b], %[cb], %[r1b], %[ub], %[cb]",
                   "v_subb_co_u32_e32 %[r1d], vcc, %[r1d], %[ud], vcc")
        TF_M3_STEP("v_addc_co_u32_e64 %[r0a], %[ka], 0, %[r0a], %[ca]", "v_addc_co_u32_e64 %[r0b], %[kb], 0, %[r0b], %[cb]",
                   "v_addc_co_u32_e64 %[r0d], %[kd], 0, %[r0d], vcc")
        TF_M3_STEP("s_andn2_b64 %[ca], %[ca], %[ka]", "s_andn2_b64 %[cb], %[cb], %[kb]",
                   "s_andn2_b64 vcc, vcc, %[kd]")
        "v_subbrev_co_u32_e64 %[r1a], %[ka], 0, %[r1a], %[ca]\n\t"                                                                // 12  r1 -= borrow & ~k
        "v_subbrev_co_u32_e64 %[r1b], %[kb], 0, %[r1b], %[cb]\n\t"
        "v_subbrev_co_u32_e32 %[r1d], vcc, 0, %[r1d], vcc"
        : [r0a] "=&v"(r0[0]), [r1a] "=&v"(r1[0]), [ua] "=&v"(u[0]), [wa] "=&v"(w[0]), [r0b] "=&v"(r0[1]), [r1b] "=&v"(r1[1]), [ub] "=&v"(u[1]),
          [wb] "=&v"(w[1]), [r0d] "=&v"(r0[2]), [r1d] "=&v"(r1[2]),
          [ud] "=&v"(u[2]), [wd] "=&v"(w[2]), [ca] "=&s"(ca), [cb] "=&s"(cb), [ka] "=&s"(ka), [kb] "=&s"(kb), [kd] "=&s"(kd)
        : [pla] "v"((u32)p[0]), [pha] "v"((u32)(p[0] >> 32)), [mla] "v"((u32)m[0]), [mha] "v"((u32)(m[0] >> 32)), [hla] "v"((u32)h[0]),
          [hha] "v"((u32)(h[0] >> 32)), [cma] "s"(cm[0]), [plb] "v"((u32)p[1]), [phb] "v"((u32)(p[1] >> 32)), [mlb] "v"((u32)m[1]),
          [mhb] "v"((u32)(m[1] >> 32)), [hlb] "v"((u32)h[1]), [hhb] "v"((u32)(h[1] >> 32)), [cmb] "s"(cm[1]), [pld] "v"((u32)p[2]), [phd] "v"((u32)(p[2] >> 32)), [mld] "v"((u32)m[2]), [mhd] "v"((u32)(m[2] >> 32)),
          [hld] "v"((u32)h[2]), [hhd] "v"((u32)(h[2] >> 32)), [cmd] "s"(cm[2])
        : "vcc", "scc");
#undef TF_M3_STEP
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = ((u64)r1[i] << 32) | r0[i];
}

#endif

GL_HD u64 to_mont(u64 v) { return mont_mul(v, R2); }     // BFieldElement::new  (:235-237)
GL_HD u64 from_mont(u64 raw) { return montyred(raw, 0); } // BFieldElement::value (:248-250)

GL_HD u64 mont_pow(u64 base, u64 exp) {
    u64 acc = ONE;
    for (int i = 63; i >= 0; --i) {
        acc = mont_mul(acc, acc);
        if ((exp >> i) & 1) acc = mont_mul(acc, base);
    }
    return acc;
}

GL_HD u64 mont_inverse(u64 a) { return mont_pow(a, P - 2); }  // 0 -> 0

// ---- multiplication by a power of two (the twiddles inside a radix-<=64 butterfly) -----------------
// Two building blocks, both returning canonical words for ANY 64-bit input x:
//
//   shl_fold<K>(x), 0 < K < 32:   x * 2^K  =  (x << K) + hi * 2^64  =  (x << K) + hi * (2^32 - 1)
//                                 one v_mad_u64_u32 does the fold; 8 VALU instructions.
//   shl_monty<S>(x), 0 <= S < 64: montyred(x << S) = x * 2^(S-64) = x * 2^(S+128) = -(x * 2^(S+32))
//                                 because 2^192 = 1 and 2^96 = -1 (mod p): the Montgomery reduction IS the
//                                 multiplication by 2^(S+32) up to a sign the butterfly absorbs; 10 instructions.
template <int K>
GL_HD u64 shl_fold(u64 x) {
    static_assert(K > 0 && K < 32, "");
    const u64 lo = x << K;
    const u32 hi = (u32)(x >> (64 - K));
    const u64 t = (u64)hi * 0xffffffffu + lo;  // < 2^65
    const bool c = t < lo;
    const u64 u = t + EPS;  // t - p (mod 2^64)
    const bool c2 = u < t;
    return (c | c2) ? u : t;
}

template <int S>
GL_HD u64 shl_monty(u64 x) {
    static_assert(S >= 0 && S < 64, "");
#if defined(__HIP_DEVICE_COMPILE__)
    // the 128-bit shifted value on 32-bit limbs: at most three instructions (shift, funnel shift, shift)
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    u32 l0, l1, h0, h1;
    if constexpr (S == 0) {
        l0 = x0, l1 = x1, h0 = 0, h1 = 0;
    } else if constexpr (S < 32) {
        l0 = x0 << S, l1 = __builtin_amdgcn_alignbit(x1, x0, 32 - S), h0 = x1 >> (32 - S), h1 = 0;
    } else if constexpr (S == 32) {
        l0 = 0, l1 = x0, h0 = x1, h1 = 0;
    } else {
        l0 = 0, l1 = x0 << (S - 32), h0 = __builtin_amdgcn_alignbit(x1, x0, 64 - S), h1 = x1 >> (64 - S);
    }
    return montyred(((u64)l1 << 32) | l0, ((u64)h1 << 32) | h0);
#else
    const u64 lo = x << S;
    const u64 hi = S ? (x >> ((64 - S) & 63)) : 0;
    return montyred(lo, hi);  // x * 2^S < 2^127 < p * 2^64: a valid Montgomery-reduction input
#endif
}

// x * 2^E mod p as (value, sign): E taken mod 192; returns v with  x * 2^E = negate ? -v : v.
template <int E>
struct Pow2Mul {
    static constexpr int e96 = ((E % 192) + 192) % 192 % 96;
    static constexpr bool high = (((E % 192) + 192) % 192) >= 96;  // 2^96 = -1
    static constexpr bool via_monty = e96 >= 32;
    static constexpr bool negate = high != via_monty;
    static GL_HD u64 apply(u64 x) {
        if constexpr (e96 == 0) {
            return x;
        } else if constexpr (e96 < 32) {
            return shl_fold<e96>(x);
        } else {
            return shl_monty<e96 - 32>(x);
        }
    }
};

// x * 2^K mod p (canonical), 0 <= K < 192
template <int K>
GL_HD u64 mul_pow2(u64 x) {
    const u64 v = Pow2Mul<K>::apply(x);
    if constexpr (Pow2Mul<K>::negate) return neg(v);
    return v;
}

}  // namespace gl
