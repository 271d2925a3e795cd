// poly_kernels.h -- small kernels of the zerofier-tree batch evaluation.
//
// What the reference computes (twenty-first/src/math/polynomial.rs:1840-1852 batch_evaluate -> divide_and_conquer_batch_evaluate
// :1882-1894 over math/zerofier_tree.rs): f at m arbitrary points by remaindering down a binary tree of zerofiers
// Z_node(x) = prod (x - p_i), leaves evaluated directly.  Every scheme returns exactly f(p_i) (exact field arithmetic), so the
// tree here is laid out for the device instead of mirrored node by node:
//   * leaves of LEAF = 1024 points (one workgroup each: the O(LEAF^2) leaf work is 1/64 of a Horner pass at m = 2^16 and saves
//     two tree levels of launch-bound small products); a level is ONE array [nodes][d] of monic zerofiers stored without their leading 1 ("tails"),
//     so every product / remainder of a level is one batched fast_multiply (tf_hip.hip: zerofier_tree_*);
//   * a remainder f mod Z (deg f < 2d, deg Z = d) is taken with the power-series inverse g of rev(Z) mod x^d:
//     rev(q) = rev(f_high) * g mod x^d,  r = f_low - (q * tail(Z))_low   (two products of size d x d);
//     g of a parent = g_left * g_right (precision d) followed by one Newton step to precision 2d, so the inverses cost three
//     products per level and no division is ever done from scratch;
//   * the reference's reduce_by_ntt_friendly_modulus (:1087-1142) serves the same purpose on the CPU (a cheap first reduction
//     when deg f >> m); here a polynomial longer than the padded point count M is cut into chunks of M coefficients which go
//     down the tree together and are recombined per point with powers of x^M.
// The kernels in this file are the glue between those batched products; field elements are L words (1 BFE, 3 XFE).
#pragma once

#include "ntt_kernels.h"

namespace tfk {

template <int L>
__device__ __forceinline__ void fe_add(const u64 (&a)[L], const u64 (&b)[L], u64 (&r)[L]) {
#pragma unroll
    for (int k = 0; k < L; ++k) r[k] = gl::add(a[k], b[k]);
}
template <int L>
__device__ __forceinline__ void fe_sub(const u64 (&a)[L], const u64 (&b)[L], u64 (&r)[L]) {
#pragma unroll
    for (int k = 0; k < L; ++k) r[k] = gl::sub(a[k], b[k]);
}
template <int L>
__device__ __forceinline__ void fe_load(const u64* p, u64 (&r)[L]) {
#pragma unroll
    for (int k = 0; k < L; ++k) r[k] = p[k];
}
template <int L>
__device__ __forceinline__ void fe_store(u64* p, const u64 (&r)[L]) {
#pragma unroll
    for (int k = 0; k < L; ++k) p[k] = r[k];
}

// One workgroup (d threads, d = leaf size <= kLeafMax) per leaf: tail[j] = coefficient j (j < d) of prod_{k < d} (x - p_k), and
// inv[k] = coefficient k (k < d) of the power-series inverse of rev(Z) = 1 + z_{d-1} x + ... + z_0 x^d.
// Points beyond n_points read as zero (the padded evaluations are never stored).
constexpr int kLeafMax = 1024;  // threads per leaf workgroup = points per leaf
template <int L>
__global__ void __launch_bounds__(kLeafMax) leaf_zerofier_kernel(const u64* points, long long n_points, int d, u64* tails, u64* inv) {
    __shared__ u64 c[kLeafMax * L];    // coefficients 0 .. d - 1 of the running product (the leading 1 leaves the array at the last point)
    __shared__ u64 acc[kLeafMax * L];  // partial sums of the series inversion
    const int t = threadIdx.x;
    const long long leaf = blockIdx.x;
    // product: c = 1; for every point p: c_new[j] = c[j - 1] - p * c[j]
#pragma unroll
    for (int k = 0; k < L; ++k) c[t * L + k] = (t == 0 && k == 0) ? gl::ONE : 0;
    __syncthreads();
    for (int i = 0; i < d; ++i) {
        const long long pi = leaf * d + i;
        u64 p[L], cur[L], prev[L];
#pragma unroll
        for (int k = 0; k < L; ++k) p[k] = pi < n_points ? points[pi * L + k] : 0;
        fe_load<L>(&c[t * L], cur);
#pragma unroll
        for (int k = 0; k < L; ++k) prev[k] = t > 0 ? c[(t - 1) * L + k] : 0;
        __syncthreads();
        u64 pc[L], nv[L];
        fe_mul<L>(p, cur, pc);
        fe_sub<L>(prev, pc, nv);
        fe_store<L>(&c[t * L], nv);
        __syncthreads();
    }
    // the product is monic of degree d: c holds its tail
    {
        u64 v[L];
        fe_load<L>(&c[t * L], v);
        fe_store<L>(tails + (leaf * d + t) * L, v);
    }
    // inverse of h(x) = sum_k h_k x^k, h_k = c[d - k], h_0 = 1:  g_0 = 1, g_k = -sum_{j=1..k} h_j g_{k-j}.
    // Online form: after g_k is known, every later index s > k adds h_{s-k} g_k to its partial sum acc[s].
#pragma unroll
    for (int k = 0; k < L; ++k) acc[t * L + k] = 0;
    __syncthreads();
    for (int k = 0; k < d; ++k) {
        u64 gk[L];
        if (k == 0) {
#pragma unroll
            for (int q = 0; q < L; ++q) gk[q] = q ? 0 : gl::ONE;
        } else {
            u64 s[L], z[L];
            fe_load<L>(&acc[k * L], s);
#pragma unroll
            for (int q = 0; q < L; ++q) z[q] = 0;
            fe_sub<L>(z, s, gk);
        }
        if (t == k) fe_store<L>(inv + (leaf * d + k) * L, gk);
        if (t > k) {
            u64 h[L], prod[L], a[L], r[L];
            fe_load<L>(&c[(d - (t - k)) * L], h);
            fe_mul<L>(h, gk, prod);
            fe_load<L>(&acc[t * L], a);
            fe_add<L>(a, prod, r);
            fe_store<L>(&acc[t * L], r);
        }
        __syncthreads();
    }
}

// tails of the parents from the product P = tail_left * tail_right (2d - 1 coefficients, packed) and the children's tails:
//   (x^d + A)(x^d + B) = x^2d + (A + B) x^d + A B   ->   out[j] = P[j] (j < 2d - 1) + (A + B)[j - d] (j >= d),  j < 2d
template <int L>
__global__ void __launch_bounds__(256) zerofier_combine_kernel(const u64* P, const u64* child_tails, u64* out, long long d, long long n_parents) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents * 2 * d) return;
    const long long node = i / (2 * d), j = i - node * 2 * d;
    u64 v[L];
#pragma unroll
    for (int k = 0; k < L; ++k) v[k] = j < 2 * d - 1 ? P[(node * (2 * d - 1) + j) * L + k] : 0;
    if (j >= d) {
        u64 a[L], b[L], s[L], r[L];
        fe_load<L>(child_tails + ((2 * node) * d + (j - d)) * L, a);
        fe_load<L>(child_tails + ((2 * node + 1) * d + (j - d)) * L, b);
        fe_add<L>(a, b, s);
        fe_add<L>(v, s, r);
        fe_store<L>(out + i * L, r);
    } else {
        fe_store<L>(out + i * L, v);
    }
}

// H = rev(Z) to precision len (= deg Z): H[0] = 1, H[k] = tail[len - k]
template <int L>
__global__ void __launch_bounds__(256) zerofier_reverse_kernel(const u64* tails, u64* H, long long len, long long nodes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes * len) return;
    const long long node = i / len, k = i - node * len;
#pragma unroll
    for (int q = 0; q < L; ++q) H[i * L + q] = k == 0 ? (q ? 0 : gl::ONE) : tails[(node * len + (len - k)) * L + q];
}

// dst[node][k] = src[node][k], k < len  (src polynomials src_stride elements apart): truncation mod x^len
template <int L>
__global__ void __launch_bounds__(256) poly_truncate_kernel(const u64* src, long long src_stride, u64* dst, long long len, long long nodes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes * len) return;
    const long long node = i / len, k = i - node * len;
#pragma unroll
    for (int q = 0; q < L; ++q) dst[i * L + q] = src[(node * src_stride + k) * L + q];
}

// E = 2 - T mod x^len  (T polynomials t_stride elements apart): the Newton step g <- g (2 - h g)
template <int L>
__global__ void __launch_bounds__(256) newton_two_minus_kernel(const u64* T, long long t_stride, u64* E, long long len, long long nodes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes * len) return;
    const long long node = i / len, k = i - node * len;
    u64 tv[L], z[L], r[L];
    fe_load<L>(T + (node * t_stride + k) * L, tv);
#pragma unroll
    for (int q = 0; q < L; ++q) z[q] = (k == 0 && q == 0) ? gl::add(gl::ONE, gl::ONE) : 0;
    fe_sub<L>(z, tv, r);
    fe_store<L>(E + i * L, r);
}

// fr[child][k] = f[child / 2][2d - 1 - k], k < d: the reversed upper halves of the parents' remainders, once per child
template <int L>
__global__ void __launch_bounds__(256) remainder_rev_high_kernel(const u64* f, u64* fr, long long d, long long children) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= children * d) return;
    const long long child = i / d, k = i - child * d;
#pragma unroll
    for (int q = 0; q < L; ++q) fr[i * L + q] = f[((child >> 1) * 2 * d + (2 * d - 1 - k)) * L + q];
}

// q[child][j] = Qr[child][d - 1 - j], j < d  (Qr polynomials qr_stride elements apart)
template <int L>
__global__ void __launch_bounds__(256) poly_reverse_kernel(const u64* Qr, long long qr_stride, u64* q, long long d, long long nodes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes * d) return;
    const long long node = i / d, j = i - node * d;
#pragma unroll
    for (int k = 0; k < L; ++k) q[i * L + k] = Qr[(node * qr_stride + (d - 1 - j)) * L + k];
}

// r[child][j] = f[child / 2][j] - S[child][j], j < d  (S polynomials s_stride elements apart)
template <int L>
__global__ void __launch_bounds__(256) remainder_finish_kernel(const u64* f, const u64* S, long long s_stride, u64* r, long long d, long long children) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= children * d) return;
    const long long child = i / d, j = i - child * d;
    u64 a[L], b[L], v[L];
    fe_load<L>(f + ((child >> 1) * 2 * d + j) * L, a);
    fe_load<L>(S + (child * s_stride + j) * L, b);
    fe_sub<L>(a, b, v);
    fe_store<L>(r + i * L, v);
}

// Leaves: remainder of degree < d per leaf, evaluated at the leaf's d points by Horner (coefficients through LDS).
// vals[leaf * d + t] = r_leaf(points[leaf * d + t]);  grid = leaves, block = d threads
template <int L>
__global__ void __launch_bounds__(kLeafMax) leaf_evaluate_kernel(const u64* rem, const u64* points, long long n_points, int d, u64* vals) {
    __shared__ u64 c[kLeafMax * L];
    const int t = threadIdx.x;
    const long long leaf = blockIdx.x;
#pragma unroll
    for (int k = 0; k < L; ++k) c[t * L + k] = rem[(leaf * d + t) * L + k];
    __syncthreads();
    const long long pi = leaf * d + t;
    if (pi >= n_points) return;
    u64 x[L], a[L];
    fe_load<L>(points + pi * L, x);
#pragma unroll
    for (int k = 0; k < L; ++k) a[k] = 0;
    for (int j = d - 1; j >= 0; --j) {
        u64 m[L], cj[L], r[L];
        fe_mul<L>(a, x, m);
        fe_load<L>(&c[j * L], cj);
        fe_add<L>(m, cj, r);
#pragma unroll
        for (int k = 0; k < L; ++k) a[k] = r[k];
    }
    fe_store<L>(vals + pi * L, a);
}

// chunks of a long polynomial recombined per point: out[i] = sum_k vals[k][i] * (x_i^M)^k  (Horner over k), M = 2^log_m
template <int L>
__global__ void __launch_bounds__(256) chunk_combine_kernel(const u64* vals, long long chunk_stride, int n_chunks, const u64* points,
                                                            long long n_points, int log_m, u64* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_points) return;
    u64 xm[L], acc[L];
    fe_load<L>(points + i * L, xm);
    for (int s = 0; s < log_m; ++s) {
        u64 sq[L];
        fe_mul<L>(xm, xm, sq);
#pragma unroll
        for (int k = 0; k < L; ++k) xm[k] = sq[k];
    }
    fe_load<L>(vals + ((long long)(n_chunks - 1) * chunk_stride + i) * L, acc);
    for (int c = n_chunks - 2; c >= 0; --c) {
        u64 m[L], v[L], r[L];
        fe_mul<L>(acc, xm, m);
        fe_load<L>(vals + ((long long)c * chunk_stride + i) * L, v);
        fe_add<L>(m, v, r);
#pragma unroll
        for (int k = 0; k < L; ++k) acc[k] = r[k];
    }
    fe_store<L>(out + i * L, acc);
}

}  // namespace tfk
