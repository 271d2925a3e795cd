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

GL_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// Montgomery reduction of the 128-bit value hi*2^64 + lo (b_field_element.rs:357-370).
GL_HD u64 montyred(u64 lo, u64 hi) {
    u64 a = lo + (lo << 32);
    u64 e = a < lo;
    u64 b = a - (a >> 32) - e;
    u64 r = hi - b;
    return (hi < b) ? r - EPS : r;
}

// Montgomery product: a * b * 2^-64 mod p.
GL_HD u64 mont_mul(u64 a, u64 b) { return montyred(a * b, mulhi64(a, b)); }

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

// x * 2^K mod p for canonical x, 0 <= K < 96.  (Plain integer power of two: multiplying a
// Montgomery word by it keeps the Montgomery form.)
template <int K>
GL_HD u64 mul_pow2(u64 x) {
    static_assert(K >= 0 && K < 96, "fold 2^96 = -1 into the butterfly sign");
    if constexpr (K == 0) {
        return x;
    } else if constexpr (K < 32) {
        // x * 2^K = lo + hi * 2^64 = lo + hi * EPS   (hi < 2^K)
        u64 lo = x << K;
        u64 hi = x >> (64 - K);
        u64 t = (hi << 32) - hi;  // hi * EPS < 2^63
        return add(lo, t);        // true sum < 2^64 + 2^63: add() still canonicalises
    } else if constexpr (K < 64) {
        // y = x << (K - 32) = y0 + y1 2^32 + y2 2^64 ; y * 2^32 = y0 2^32 + y1 EPS - y2
        constexpr int J = K - 32;
        u64 ylo = x << J;
        u64 y2 = (J == 0) ? 0 : (x >> ((64 - J) & 63));
        u64 y0 = ylo & 0xffffffffULL, y1 = ylo >> 32;
        u64 a = y0 << 32;          // <= p - 1
        u64 b = (y1 << 32) - y1;   // y1 * EPS < p
        return sub(add(a, b), y2);
    } else {
        // y = x << (K - 64) ; y * 2^64 = y0 EPS - y1 - y2 2^32
        constexpr int J = K - 64;
        u64 ylo = x << J;
        u64 y2 = (J == 0) ? 0 : (x >> ((64 - J) & 63));
        u64 y0 = ylo & 0xffffffffULL, y1 = ylo >> 32;
        u64 a = (y0 << 32) - y0;   // y0 * EPS < p
        u64 b = y1 + (y2 << 32);   // < 2^32 + 2^63 < p
        return sub(a, b);
    }
}

}  // namespace gl
