// ntt_kernels.h -- batched Goldilocks NTT passes for gfx950 (device side).
//
// What the reference computes (twenty-first/src/math/ntt.rs:67-82, :109-125, :153-228):
//   out[k] = sum_j x[j] * w_n^(j k)   (natural order in and out; intt uses w^-1 and scales by n^-1)
// as log2(n) radix-2 sweeps over the whole slice.  Here the same transform is a generalized
// Cooley-Tukey factorisation n = N_1 * ... * N_s (s <= 3, N_i <= 1024), one HBM pass per factor:
//
//   pass i (not last):  for every (k_1..k_{i-1}, j_{i+1}..j_s):  DFT over j_i, then multiply by the
//                       inter-pass twiddle w_{N_i...N_s}^(k_i * b_i); data stays in place.
//   last pass:          DFT over j_s and write to the digit-reversed position k_1 + N_1 (k_2 + ...),
//                       so the result is in natural order with no bit-reversal sweep.
//
// Inside a pass a workgroup owns a tile of `nc` independent length-R DFTs ("columns"), R = 32 * P2:
//   step 1: every thread holds 32 rows of one column in registers and runs a radix-32 DIT network
//           whose twiddles are all powers of two (gl::mul_pow2 - shifts, no 64x64 multiply);
//   inner twiddle w_R^(g k1) (one Montgomery multiply per element; n^-1 of the inverse folded in);
//   one LDS exchange (also the transposition that makes the global stores coalesced);
//   step 2: radix-P2 DIT network on the other index, again shift-only;
//   inter-pass twiddle (one Montgomery multiply), coalesced store.
// So an element costs 2-3 general multiplies per n = 2^20 transform instead of 10.
//
// XFE slices ([c0,c1,c2] per element, x_field_element.rs:56-59) are three interleaved BFE columns
// with the same twiddles (ntt.rs:203-207, x_field_element.rs:540-548): L = 3 words per element.
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
    long long n_coeffs;    // elements present per input polynomial; rows beyond are zero (polynomial.rs:1395); <0: no padding
    long long ib0, ib1, ib2, ob0, ob1, ob2;  // tile base strides (words)
    long long in_cs_hi, out_cs_hi;           // column c -> (c / L) * cs_hi + (c % L)
    long long in_rs, out_rs;                 // row strides (words)
    long long tw_rs;                         // row stride of post_tw (= B)
    long long ps_rs;                         // pre_scale index j = r * ps_rs + (ps_col ? b : 0)
    u32 d1, d2;                              // tile id = (i0 * d1 + i1) * d2 + i2
    int p2;                                  // log2 P2  (R = 32 << p2)
    int nc;                                  // columns per tile
    int L;                                   // words per element (1 BFE, 3 XFE)
    int col_limit;                           // valid columns along i2: min(nc, col_limit - i2 * nc)
    int load_rowfast, store_rowfast;         // lane order: 1 = (limb, g) fastest, 0 = column fastest
    int ps_col;
    int s1, s2, s3;                          // LDS strides in u64: idx = k1*s1 + g*s2 + (c/L)*s3 + c%L
};

// ---- radix-2^k DIT network with power-of-two twiddles --------------------------------------
// w_{2^l} = 2^(39 * 2^(6-l))  (b_field_element.rs:46-51: w_64 = 2^39, ..., w_2 = 2^96 = -1), order 192.
template <bool INV, int LVL, int J>
struct TwExp {
    static constexpr int fwd = ((39 << (6 - LVL)) * J) % 192;
    static constexpr int value = INV ? (192 - fwd) % 192 : fwd;
};

template <int E>
__device__ __forceinline__ void butterfly_pow2(u64& a, u64& b) {
    if constexpr (E < 96) {
        u64 v = gl::mul_pow2<E>(b);
        u64 s = gl::add(a, v);
        b = gl::sub(a, v);
        a = s;
    } else {  // 2^E = -2^(E-96)
        u64 v = gl::mul_pow2<E - 96>(b);
        u64 s = gl::sub(a, v);
        b = gl::add(a, v);
        a = s;
    }
}

template <bool INV, int LVL, int BASE, int J>
struct DitInner {
    static __device__ __forceinline__ void run(u64 (&x)[32]) {
        constexpr int H = 1 << (LVL - 1);
        butterfly_pow2<TwExp<INV, LVL, J>::value>(x[BASE + J], x[BASE + J + H]);
        if constexpr (J + 1 < H) DitInner<INV, LVL, BASE, J + 1>::run(x);
    }
};
template <bool INV, int LVL, int BASE>
struct DitGroups {
    static __device__ __forceinline__ void run(u64 (&x)[32]) {
        constexpr int H = 1 << (LVL - 1);
        DitInner<INV, LVL, BASE, 0>::run(x);
        if constexpr (BASE + 2 * H < 32) DitGroups<INV, LVL, BASE + 2 * H>::run(x);
    }
};
// Level LVL of a DIT network over all 32 registers (groups of 2^LVL consecutive slots; inputs of a
// group in bit-reversed order, outputs natural).
template <bool INV, int LVL>
__device__ __forceinline__ void dit_level(u64 (&x)[32]) { DitGroups<INV, LVL, 0>::run(x); }

__device__ __forceinline__ constexpr int brev5(int q) {
    return ((q & 1) << 4) | ((q & 2) << 2) | (q & 4) | ((q & 8) >> 2) | ((q & 16) >> 4);
}

template <bool INV>
__global__ void __launch_bounds__(512) ntt_pass_kernel(const NttPassArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int t = threadIdx.x;
    const int p2 = A.p2;
    const int P2 = 1 << p2;
    const int L = A.L;

    const u32 tile = blockIdx.x;
    const u32 i2 = tile % A.d2;
    const u32 rest = tile / A.d2;
    const u32 i1 = rest % A.d1;
    const u32 i0 = rest / A.d1;
    const u64* in = A.in + (long long)i0 * A.ib0 + (long long)i1 * A.ib1 + (long long)i2 * A.ib2;
    u64* out = A.out + (long long)i0 * A.ob0 + (long long)i1 * A.ob1 + (long long)i2 * A.ob2;
    const int col0 = (int)i2 * A.nc;
    const int ncv = min(A.nc, A.col_limit - col0);

    u64 x[32];
    // ------------------------------------------------------------------ load + step 1
    {
        int c, g;
        if (A.load_rowfast) {
            int limb = t % L, q = t / L;
            g = q & (P2 - 1);
            c = (q >> p2) * L + limb;
        } else {
            c = t % A.nc;
            g = t / A.nc;
        }
        const bool act = c < ncv;
        const int ch = c / L, cl = c - ch * L;
        const u64* src = in + (long long)ch * A.in_cs_hi + cl;
        const long long bcol = A.ps_col ? (long long)((col0 + c) / L) : 0;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int r = g + (brev5(q) << p2);
            u64 v = 0;
            if (act) {
                if (A.pre_scale) {
                    const long long j = (long long)r * A.ps_rs + bcol;
                    if (A.n_coeffs < 0 || j < A.n_coeffs) v = gl::mont_mul(src[(long long)r * A.in_rs], A.pre_scale[j]);
                } else {
                    v = src[(long long)r * A.in_rs];
                }
            }
            x[q] = v;
        }
        dit_level<INV, 1>(x);
        dit_level<INV, 2>(x);
        dit_level<INV, 3>(x);
        dit_level<INV, 4>(x);
        dit_level<INV, 5>(x);
        if (A.inner_tw) {
            const u64* tw = A.inner_tw + g * 32;
#pragma unroll
            for (int q = 0; q < 32; ++q) x[q] = gl::mont_mul(x[q], tw[q]);
        }
        u64* dst = lds + g * A.s2 + ch * A.s3 + cl;
#pragma unroll
        for (int q = 0; q < 32; ++q) dst[q * A.s1] = x[q];
    }
    __syncthreads();
    // ------------------------------------------------------------------ step 2 + store
    {
        int c, g;
        if (A.store_rowfast) {
            int limb = t % L, q = t / L;
            g = q & (P2 - 1);
            c = (q >> p2) * L + limb;
        } else {
            c = t % A.nc;
            g = t / A.nc;
        }
        const bool act = c < ncv;
        const int ch = c / L, cl = c - ch * L;
        const u64* srcl = lds + ch * A.s3 + cl;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int s = q >> p2;                 // uniform
            const int gbr = q & (P2 - 1);          // uniform
            const int gg = p2 ? (int)(__brev((unsigned)gbr) >> (32 - p2)) : 0;
            const int k1 = g + (s << p2);
            x[q] = srcl[k1 * A.s1 + gg * A.s2];
        }
        if (p2 >= 1) dit_level<INV, 1>(x);
        if (p2 >= 2) dit_level<INV, 2>(x);
        if (p2 >= 3) dit_level<INV, 3>(x);
        if (p2 >= 4) dit_level<INV, 4>(x);
        if (p2 >= 5) dit_level<INV, 5>(x);
        u64* dstg = out + (long long)ch * A.out_cs_hi + cl;
        const long long b = (long long)((col0 + c) / L);
        if (act) {
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int s = q >> p2;
                const int k2 = q & (P2 - 1);
                const int k = g + (s << p2) + (k2 << 5);
                u64 v = x[q];
                if (A.post_tw) v = gl::mont_mul(v, A.post_tw[(long long)k * A.tw_rs + b]);
                dstg[(long long)k * A.out_rs] = v;
            }
        }
    }
}

// ---- n <= 16: one thread per (transform, limb); reference-shaped radix-2 loop, tables in global memory.
struct NttTinyArgs {
    const u64* in;
    u64* out;
    const u64* tw;         // stage tables back to back: stage i (m = 2^i) at offset m - 1 (ntt.rs:309-324)
    const u64* pre_scale;  // or null
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

// out[j] = HI[j >> h] * LO[j & (2^h - 1)]   (offset^j)
__global__ void __launch_bounds__(256) build_pow_table_kernel(u64* out, const u64* hi, const u64* lo, int h, long long n) {
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    out[id] = gl::mont_mul(hi[id >> h], lo[id & ((1ll << h) - 1)]);
}

}  // namespace tfk
