// lat_kernels.h -- the latency-shaped transforms of libtf_hip.so (device side; launchers: tf_lat.hip).
//
// The reference's ntt() takes ONE slice (math/ntt.rs:67-82) and a zerofier-tree walk (math/zerofier_tree.rs) transforms a handful
// of short polynomials per level: calls that cannot fill the chip.  A 32-elements-per-thread pass kernel is ~3 500 instructions in
// a row however little work there is; these kernels spend threads instead -- EIGHT elements per thread, radix-8 Stockham stages
// joined through LDS:
//   ntt_lat_kernel            64 <= n <= 4096, a whole transform per workgroup slice
//   ntt_lat2_kernel           2^13 <= n <= 2^20 as the two passes of n = N1 N2 on the same stages
//   tree_down / up_level      a whole LEVEL of a zerofier-tree walk in one launch (BFieldElement, 2d <= 4096)
//   tree_build_level          a whole level of the tree BUILD in one launch (BFieldElement, 2d <= 2048)
#pragma once

#include "gl64.h"
#include "ntt_args.h"
#include "ntt_network.h"

namespace tfk {

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
// (struct NttLatArgs: ntt_args.h)
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
                if (A.pre_scale) v = gl::mont_mul(v, A.pre_scale[idx]);
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
                            dst[(long long)idx * L] = A.post_scale ? gl::mont_mul(x[a * R + r], A.post_scale[idx]) : x[a * R + r];
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

// (struct TreeLevelArgs: ntt_args.h)

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
// (struct TreeBuildArgs: ntt_args.h)
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

#ifdef TF_AB_BUILD  // measured loss (243 -> 258 us evaluate, 160 -> 201 us interpolate at 2^12 points, profiles/r03_tree_level_ab.txt): laboratory build only
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

#endif  // TF_AB_BUILD

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
    const u64* scale_tab;      // or null.  Column pass: input element (idx, column chi) times scale_tab[idx * scale_es + chi] (the offset^j of
                               // fast_coset_evaluate on coefficient j = idx * N2 + chi); last pass: output element likewise (the offset^-j
                               // of fast_coset_interpolate on coefficient j = idx * N1 + row)
    long long scale_es;
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
                if (!LAST && A.scale_tab) v = gl::mont_mul(v, A.scale_tab[(long long)idx * A.scale_es + chi]);
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
                        else if (A.scale_tab) val = gl::mont_mul(val, A.scale_tab[(long long)idx * A.scale_es + chi]);
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

}  // namespace tfk
