// poly_kernels.h -- small kernels of the zerofier-tree batch evaluation.
//
// What the reference computes (twenty-first/src/math/polynomial.rs:1840-1852 batch_evaluate -> divide_and_conquer_batch_evaluate
// :1882-1894 over math/zerofier_tree.rs): f at m arbitrary points by remaindering down a binary tree of zerofiers
// Z_node(x) = prod (x - p_i), leaves evaluated directly.  Every scheme returns exactly f(p_i) (exact field arithmetic), so the
// tree here is laid out for the device instead of mirrored node by node:
//   * leaves of 256 (BFE) / 128 (XFE) points, one workgroup each (O(leaf^2) work in LDS); a level is ONE array [nodes][d] of
//     monic zerofiers stored without their leading 1 ("tails") beside its forward transforms of order 2d, so every product /
//     remainder of a level is a batched transform, a pointwise kernel and a batched inverse transform (tf_poly.hip: zerofier_tree_*);
//   * a remainder f mod Z (deg f < 2d, deg Z = d) is taken with the power-series inverse g of rev(Z) mod x^d:
//     rev(q) = rev(f_high) * g mod x^d,  r = f_low - (q * tail(Z))_low   (two products of size d x d against cached transforms);
//     g of a parent = g_left * g_right (precision d) followed by one Newton step to precision 2d at order 4d, so no division is
//     ever done from scratch;
//   * the reference's reduce_by_ntt_friendly_modulus (:1087-1142) serves the same purpose on the CPU (a cheap first reduction
//     when deg f >> m); here a polynomial longer than the padded point count M is cut into chunks of M coefficients which go
//     down the tree together and are recombined per point with powers of x^M.
// The kernels in this file are the glue between those batched products; field elements are L words (1 BFE, 3 XFE).
#pragma once

#include "gl64.h"

namespace tfk {

using gl::u32;
using gl::u64;

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
    extern __shared__ u64 leaf_lds[];  // 3 * d * L words (dynamic: a small leaf leaves the LDS to more workgroups)
    u64* c = leaf_lds;                 // coefficients 0 .. d - 1 of the running product (the leading 1 leaves the array at the last point)
    u64* acc = leaf_lds + d * L;       // second product buffer, then the partial sums of the series inversion
    u64* pts = leaf_lds + 2 * d * L;   // the leaf's points
    const int t = threadIdx.x;
    const long long leaf = blockIdx.x;
    // product: c = 1; for every point p: c_new[j] = c[j - 1] - p * c[j].  The points are staged once and the product alternates
    // between two buffers: one barrier and no global load per step (a step is on the critical path d times).
    {
        const long long pi = leaf * d + t;
#pragma unroll
        for (int k = 0; k < L; ++k) {
            c[t * L + k] = (t == 0 && k == 0) ? gl::ONE : 0;
            pts[t * L + k] = pi < n_points ? points[pi * L + k] : 0;
        }
    }
    __syncthreads();
    {
        u64* from = c;
        u64* to = acc;
        for (int i = 0; i < d; ++i) {
            u64 p[L], cur[L], prev[L];
            fe_load<L>(&pts[i * L], p);
            fe_load<L>(&from[t * L], cur);
#pragma unroll
            for (int k = 0; k < L; ++k) prev[k] = t > 0 ? from[(t - 1) * L + k] : 0;
            u64 pc[L], nv[L];
            fe_mul<L>(p, cur, pc);
            fe_sub<L>(prev, pc, nv);
            fe_store<L>(&to[t * L], nv);
            __syncthreads();
            u64* sw = from;
            from = to;
            to = sw;
        }
        if (from != c) {  // d odd: the product ended in the second buffer
            u64 v[L];
            fe_load<L>(&from[t * L], v);
            __syncthreads();
            fe_store<L>(&c[t * L], v);
            __syncthreads();
        }
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

// dst[node][k] = src[node][k], k < len  (src polynomials src_stride elements apart): truncation mod x^len
template <int L>
__global__ void __launch_bounds__(256) poly_truncate_kernel(const u64* src, long long src_stride, u64* dst, long long len, long long nodes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes * len) return;
    const long long node = i / len, k = i - node * len;
#pragma unroll
    for (int q = 0; q < L; ++q) dst[i * L + q] = src[(node * src_stride + k) * L + q];
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

// r[child][j] = f[child / 2][j] - S[child][j], j < d  (S polynomials s_stride elements apart).  When fr_next is given, the upper
// half of r is also written reversed for BOTH grandchildren (the first step of the next level down, remainder_rev_high_kernel with
// d / 2): fr_next[2 child + s][k] = r[child][d - 1 - k], k < d / 2 -- one launch per level less.
template <int L>
__global__ void __launch_bounds__(256) remainder_finish_kernel(const u64* f, const u64* S, long long s_stride, u64* r, long long d, long long children,
                                                               u64* fr_next) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= children * d) return;
    const long long child = i / d, j = i - child * d;
    u64 a[L], b[L], v[L];
    fe_load<L>(f + ((child >> 1) * 2 * d + j) * L, a);
    fe_load<L>(S + (child * s_stride + j) * L, b);
    fe_sub<L>(a, b, v);
    fe_store<L>(r + i * L, v);
    if (fr_next && j >= d / 2) {
        const long long h = d / 2, k = d - 1 - j;
        fe_store<L>(fr_next + ((2 * child) * h + k) * L, v);
        fe_store<L>(fr_next + ((2 * child + 1) * h + k) * L, v);
    }
}

// ---- levels in the transform domain (round 2) -------------------------------------------------------------------------------
// Every level keeps, beside the tails and the inverses, their forward transforms of order 2d ("That", "Ghat": [nodes][2d]); the
// build produces them anyway for the parents, and the walk down (and the interpolation walk up) multiplies against them
// without transforming the tree again.

// Parents' tails from the children's transformed tails: the zerofier x^d + A transforms (order 2d) to That_A[k] + (-1)^k;
// (x^d + A)(x^d + B) = x^2d + tail_parent, and x^2d wraps to 1 at order 2d:  out = (A^ + s_k)(B^ + s_k) - 1,  s_k = (-1)^k.
// The inverse transform of `out` is the parents' [parents][2d] tail array.
template <int L>
__global__ void __launch_bounds__(256) zerofier_pointwise_kernel(const u64* That, u64* out, long long d, long long n_parents) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents * 2 * d) return;
    const long long node = i / (2 * d), k = i - node * 2 * d;
    u64 a[L], b[L], r[L];
    fe_load<L>(That + ((2 * node) * 2 * d + k) * L, a);
    fe_load<L>(That + ((2 * node + 1) * 2 * d + k) * L, b);
    if (k & 1) {
        a[0] = gl::sub(a[0], gl::ONE);
        b[0] = gl::sub(b[0], gl::ONE);
    } else {
        a[0] = gl::add(a[0], gl::ONE);
        b[0] = gl::add(b[0], gl::ONE);
    }
    fe_mul<L>(a, b, r);
    r[0] = gl::sub(r[0], gl::ONE);
    fe_store<L>(out + i * L, r);
}

// out[node][k] = X[2 node][k] * X[2 node + 1][k], k < n: the product of sibling transforms (XFieldElement; over BFieldElement
// the product rides on the inverse transform's load)
template <int L>
__global__ void __launch_bounds__(256) pair_product_kernel(const u64* X, u64* out, long long n, long long n_parents) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents * n) return;
    const long long node = i / n, k = i - node * n;
    u64 a[L], b[L], r[L];
    fe_load<L>(X + ((2 * node) * n + k) * L, a);
    fe_load<L>(X + ((2 * node + 1) * n + k) * L, b);
    fe_mul<L>(a, b, r);
    fe_store<L>(out + i * L, r);
}

// Inputs of the Newton step, both 2d coefficients per parent, in ONE array so that one batched transform of order 4d takes them:
//   B[node]           = G = (g_left g_right mod x^d), zero padded   (S1: [parents][2d], the product's low half is G)
//   B[parents + node] = H = rev(Z_parent): H[0] = 1, H[k] = tail[2d - k]
template <int L>
__global__ void __launch_bounds__(256) newton_inputs_kernel(const u64* S1, const u64* parent_tails, u64* B, long long d, long long n_parents) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents * 2 * d) return;
    const long long node = i / (2 * d), k = i - node * 2 * d;
    u64 g[L], h[L];
#pragma unroll
    for (int q = 0; q < L; ++q) {
        g[q] = k < d ? S1[i * L + q] : 0;
        h[q] = k == 0 ? (q ? 0 : gl::ONE) : parent_tails[(node * 2 * d + (2 * d - k)) * L + q];
    }
    fe_store<L>(B + i * L, g);
    fe_store<L>(B + (n_parents * 2 * d + i) * L, h);
}

// C: transforms of order n = 4d of B's rows.  In place on the G rows:  C[node] <- g^ (2 - h^ g^)   (deg g (2 - h g) < 4d: no wrap;
// its low 2d coefficients are the parent's inverse to precision 2d)
template <int L>
__global__ void __launch_bounds__(256) newton_pointwise_kernel(u64* C, long long n, long long n_parents) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents * n) return;
    u64 g[L], h[L], hg[L], e[L], r[L];
    fe_load<L>(C + i * L, g);
    fe_load<L>(C + (n_parents * n + i) * L, h);
    fe_mul<L>(h, g, hg);
#pragma unroll
    for (int q = 0; q < L; ++q) e[q] = gl::neg(hg[q]);
    e[0] = gl::add(e[0], gl::add(gl::ONE, gl::ONE));
    fe_mul<L>(g, e, r);
    fe_store<L>(C + i * L, r);
}

// Leaves: remainder of degree < d per leaf, evaluated at the leaf's d points by Horner (coefficients through LDS).
// vals[leaf * d + t] = r_leaf(points[(leaf % leaves_per_unit) * d + t]);  grid = units * leaves_per_unit, block = d threads
// (several polynomials -- "units" -- walk the tree together; they share the points)
template <int L>
__global__ void __launch_bounds__(kLeafMax) leaf_evaluate_kernel(const u64* rem, const u64* points, long long n_points, int d, u64* vals,
                                                                 long long leaves_per_unit) {
    extern __shared__ u64 leaf_lds[];  // d * L words
    u64* c = leaf_lds;
    const int t = threadIdx.x;
    const long long leaf = blockIdx.x;
#pragma unroll
    for (int k = 0; k < L; ++k) c[t * L + k] = rem[(leaf * d + t) * L + k];
    __syncthreads();
    const long long pi = (leaf % leaves_per_unit) * d + t;
    if (pi >= n_points) return;
    u64 x[L], a[L];
    fe_load<L>(points + pi * L, x);
    if ((d & 3) == 0) {
        // r(x) = sum_{s < 4} x^s R_s(x^4): four independent Horner chains in x^4 (a single chain is d dependent products, the whole
        // latency of this kernel), recombined with three more products
        u64 x2[L], x4[L], acc4[4][L];
        fe_mul<L>(x, x, x2);
        fe_mul<L>(x2, x2, x4);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
            for (int k = 0; k < L; ++k) acc4[c4][k] = 0;
        for (int j = d - 4; j >= 0; j -= 4) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                u64 m[L], cj[L], r[L];
                fe_mul<L>(acc4[c4], x4, m);
                fe_load<L>(&c[(j + c4) * L], cj);
                fe_add<L>(m, cj, r);
#pragma unroll
                for (int k = 0; k < L; ++k) acc4[c4][k] = r[k];
            }
        }
        u64 t3[L], t2[L], t1[L], u3[L], u2[L];
        fe_mul<L>(acc4[3], x, t3);          // ((R3 x + R2) x + R1) x + R0
        fe_add<L>(t3, acc4[2], u3);
        fe_mul<L>(u3, x, t2);
        fe_add<L>(t2, acc4[1], u2);
        fe_mul<L>(u2, x, t1);
        fe_add<L>(t1, acc4[0], a);
        fe_store<L>(vals + (leaf * d + t) * L, a);
        return;
    }
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
    fe_store<L>(vals + (leaf * d + t) * L, a);
}

// The same leaves for a SMALL walk (few leaves: the chip is mostly idle and a thread's d products in a row are the whole latency):
// S threads per point, thread (point, s) evaluates the block of d / S coefficients  P_s(x) = sum_j c[s d/S + j] x^j  with the same
// four chains, the partial values meet in LDS and thread (point, 0) finishes  r(x) = sum_s y^s P_s(x),  y = x^(d/S).
// block = d * S threads (<= 1024), d / S a power of two and a multiple of 4.
template <int L, int S>
__global__ void __launch_bounds__(kLeafMax) leaf_evaluate_split_kernel(const u64* rem, const u64* points, long long n_points, int d, u64* vals,
                                                                       long long leaves_per_unit) {
    extern __shared__ u64 leaf_lds[];  // d L coefficients, then S d L partial values
    u64* c = leaf_lds;
    u64* part = leaf_lds + (size_t)d * L;
    const int t = threadIdx.x, s = t / d, pt = t - s * d;
    const long long leaf = blockIdx.x;
    for (int i = t; i < d * L; i += d * S) c[i] = rem[leaf * d * L + i];
    __syncthreads();
    const long long pi = (leaf % leaves_per_unit) * d + pt;
    const bool live = pi < n_points;
    u64 x[L], x2[L], x4[L], acc4[4][L];
#pragma unroll
    for (int k = 0; k < L; ++k) x[k] = 0;
    if (live) fe_load<L>(points + pi * L, x);
    fe_mul<L>(x, x, x2);
    fe_mul<L>(x2, x2, x4);
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
        for (int k = 0; k < L; ++k) acc4[c4][k] = 0;
    const int seg = d / S, base = s * seg;
    for (int j = seg - 4; j >= 0; j -= 4) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            u64 m[L], cj[L], r[L];
            fe_mul<L>(acc4[c4], x4, m);
            fe_load<L>(&c[(base + j + c4) * L], cj);
            fe_add<L>(m, cj, r);
#pragma unroll
            for (int k = 0; k < L; ++k) acc4[c4][k] = r[k];
        }
    }
    {
        u64 t3[L], t2[L], t1[L], u3[L], u2[L], a[L];
        fe_mul<L>(acc4[3], x, t3);
        fe_add<L>(t3, acc4[2], u3);
        fe_mul<L>(u3, x, t2);
        fe_add<L>(t2, acc4[1], u2);
        fe_mul<L>(u2, x, t1);
        fe_add<L>(t1, acc4[0], a);
        fe_store<L>(&part[((size_t)s * d + pt) * L], a);
    }
    __syncthreads();
    if (s != 0 || !live) return;
    u64 y[L], a[L];
#pragma unroll
    for (int k = 0; k < L; ++k) y[k] = x4[k];
    for (int e = 4; e < seg; e <<= 1) {  // y = x^seg
        u64 q[L];
        fe_mul<L>(y, y, q);
#pragma unroll
        for (int k = 0; k < L; ++k) y[k] = q[k];
    }
    fe_load<L>(&part[((size_t)(S - 1) * d + pt) * L], a);
#pragma unroll
    for (int q = S - 2; q >= 0; --q) {
        u64 m[L], pq[L], r[L];
        fe_mul<L>(a, y, m);
        fe_load<L>(&part[((size_t)q * d + pt) * L], pq);
        fe_add<L>(m, pq, r);
#pragma unroll
        for (int k = 0; k < L; ++k) a[k] = r[k];
    }
    fe_store<L>(vals + (leaf * d + pt) * L, a);
}

// out[i] = A[i] * B[i mod period]: the transforms of several units against the level's cached transforms (shared by the units)
template <int L>
__global__ void __launch_bounds__(256) product_bcast_kernel(const u64* A, const u64* B, u64* out, long long period, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    u64 a[L], b[L], r[L];
    fe_load<L>(A + i * L, a);
    fe_load<L>(B + (i % period) * L, b);
    fe_mul<L>(a, b, r);
    fe_store<L>(out + i * L, r);
}

// chunks of a long polynomial recombined per point: out[b][i] = sum_k vals[b][k][i] * (x_i^M)^k  (Horner over k), M = 2^log_m;
// grid.y = polynomial b (vals: [b][n_chunks][chunk_stride], out: [b][n_points])
template <int L>
__global__ void __launch_bounds__(256) chunk_combine_kernel(const u64* vals, long long chunk_stride, int n_chunks, const u64* points,
                                                            long long n_points, int log_m, u64* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_points) return;
    vals += (long long)blockIdx.y * n_chunks * chunk_stride * L;
    out += (long long)blockIdx.y * n_points * L;
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

// ---- zerofier and interpolation through the same tree ----------------------------------------------------------------------
// Polynomial::zerofier (polynomial.rs:1435-1485) is the root of the tree; Polynomial::interpolate / fast_interpolate /
// batch_fast_interpolate (:1502-1838) is the tree walked upwards:
//   f = sum_i w_i Z(x) / (x - x_i),  w_i = v_i / Z'(x_i)              (the interpolant is unique: any scheme returns the reference's
//   N_node = sum_{i in node} w_i Z_node / (x - x_i)  ->  N_parent = N_left Z_right + N_right Z_left       coefficients exactly)
// The tree is padded with zero points up to M = leaf << h; with zero weights on the padding every padded quantity is the true
// one times x^(number of padded points below the node), so the zerofier and the interpolant come out of the root shifted up by
// M - n coefficients and are copied out from there.

// D = Z' of the TRUE zerofier Z = (x^M + root_tail) / x^(M - n):  D[j] = (j + 1) Z_{j + 1} for j < n, zero up to M.
template <int L>
__global__ void __launch_bounds__(256) zerofier_derivative_kernel(const u64* root_tail, long long M, long long n, u64* D) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    u64 v[L];
#pragma unroll
    for (int q = 0; q < L; ++q) v[q] = 0;
    if (j < n) {
        const long long idx = j + 1 + (M - n);
        const u64 k = gl::to_mont((u64)(j + 1));
#pragma unroll
        for (int q = 0; q < L; ++q) v[q] = gl::mont_mul(k, idx < M ? root_tail[idx * L + q] : (q ? 0 : gl::ONE));
    }
    fe_store<L>(D + j * L, v);
}

// out[j] = Z_j, j <= n: the true zerofier's n + 1 coefficients from the padded root
template <int L>
__global__ void __launch_bounds__(256) zerofier_unpad_kernel(const u64* root_tail, long long M, long long n, u64* out) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    const long long idx = j + (M - n);
#pragma unroll
    for (int q = 0; q < L; ++q) out[j * L + q] = idx < M ? root_tail[idx * L + q] : (q ? 0 : gl::ONE);
}

// Multiplicative inverse in F_p[x] / (x^3 - x + 1): multiplication by a is the matrix
//   [ a0   -a2      -a1    ]
//   [ a1   a0 + a2  a1 - a2]     (x_field_element.rs:512-536 read as a linear map of the other operand)
//   [ a2   a1       a0 + a2]
// and a^-1 is the solution of M b = (1, 0, 0): the cofactors of the first row over the determinant.
__device__ __forceinline__ bool xfe_inverse(const u64 (&a)[3], u64 (&r)[3]) {
    const u64 s = gl::add(a[0], a[2]), dd = gl::sub(a[1], a[2]);
    const u64 c0 = gl::sub(gl::mont_mul(s, s), gl::mont_mul(dd, a[1]));
    const u64 c1 = gl::sub(gl::mont_mul(dd, a[2]), gl::mont_mul(a[1], s));
    const u64 c2 = gl::sub(gl::mont_mul(a[1], a[1]), gl::mont_mul(s, a[2]));
    const u64 det = gl::sub(gl::sub(gl::mont_mul(a[0], c0), gl::mont_mul(a[2], c1)), gl::mont_mul(a[1], c2));
    const u64 di = gl::mont_inverse(det);
    r[0] = gl::mont_mul(c0, di);
    r[1] = gl::mont_mul(c1, di);
    r[2] = gl::mont_mul(c2, di);
    return det != 0;
}

// w[i] = 1 / x[i], i < n; *flag is raised when some x[i] is zero (a repeated domain point: the reference panics in
// batch_inversion, traits.rs:106)
template <int L>
__global__ void __launch_bounds__(256) fe_inverse_kernel(const u64* x, long long n, u64* w, int* flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 a[L], r[L];
    fe_load<L>(x + i * L, a);
    bool ok;
    if constexpr (L == 1) {
        r[0] = gl::mont_inverse(a[0]);
        ok = a[0] != 0;
    } else {
        ok = xfe_inverse(a, r);
    }
    if (!ok) atomicOr(flag, 1);
    fe_store<L>(w + i * L, r);
}

// The *_dev_async entry points report a panic case of the reference through a status word in DEVICE memory instead of
// synchronising the stream to read a flag: status keeps the FIRST non-zero code written to it (0 = nothing happened).
__global__ void status_merge_kernel(const int* flag, int* status, int code_bit0, int code_other) {
    const int f = *flag;
    if (f) atomicCAS(status, 0, (f & 1) ? code_bit0 : code_other);
}

// targets[row][i] = values[row][i] * w[i] for i < n, zero for n <= i < M   (grid.y = rows; values rows n elements apart)
template <int L>
__global__ void __launch_bounds__(256) interpolation_targets_kernel(const u64* values, const u64* w, long long n, long long M, u64* targets) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const long long row = blockIdx.y;
    u64 r[L];
#pragma unroll
    for (int q = 0; q < L; ++q) r[q] = 0;
    if (i < n) {
        u64 v[L], wi[L];
        fe_load<L>(values + (row * n + i) * L, v);
        fe_load<L>(w + i * L, wi);
        fe_mul<L>(v, wi, r);
    }
    fe_store<L>(targets + (row * M + i) * L, r);
}

// Leaves of the interpolant: N = sum_i w_i prod_{k != i} (x - p_k) over the leaf's d points, built with the zerofier one point at
// a time:  N <- N (x - p) + w Z,  Z <- Z (x - p).   grid = (leaves, rows), block = d threads (thread t owns coefficient t).
template <int L>
// The targets w_i = values[row][i] / Z'(x_i) are formed while the leaf's points are staged (winv = the inverse weights).
__global__ void __launch_bounds__(kLeafMax) leaf_interpolant_kernel(const u64* points, const u64* values, const u64* winv, long long n_points, int d,
                                                                    long long M, u64* N) {
    extern __shared__ u64 leaf_lds[];  // 6 * d * L words: Z and N twice (the step alternates between the copies), points, targets
    const int t = threadIdx.x;
    const long long leaf = blockIdx.x, row = blockIdx.y;
    u64* zb[2] = {leaf_lds, leaf_lds + 2 * d * L};          // Z, coefficients 0 .. d - 1 (its leading 1 is still inside while it is needed)
    u64* nb[2] = {leaf_lds + d * L, leaf_lds + 3 * d * L};  // N
    u64* pts = leaf_lds + 4 * d * L;
    u64* tgt = leaf_lds + 5 * d * L;
    {
        const long long pi = leaf * d + t;
#pragma unroll
        for (int k = 0; k < L; ++k) {
            zb[0][t * L + k] = (t == 0 && k == 0) ? gl::ONE : 0;
            nb[0][t * L + k] = 0;
            pts[t * L + k] = pi < n_points ? points[pi * L + k] : 0;
        }
        u64 tv[L];
#pragma unroll
        for (int k = 0; k < L; ++k) tv[k] = 0;
        if (pi < n_points) {
            u64 v[L], wi[L];
            fe_load<L>(values + (row * n_points + pi) * L, v);
            fe_load<L>(winv + pi * L, wi);
            fe_mul<L>(v, wi, tv);
        }
        fe_store<L>(&tgt[t * L], tv);
    }
    __syncthreads();
    int cur_buf = 0;
    for (int i = 0; i < d; ++i) {  // one barrier and no global load per step: the d steps are the kernel's whole latency
        const u64* c = zb[cur_buf];
        const u64* nn = nb[cur_buf];
        u64 p[L], w[L], cur[L], prev[L], ncur[L], nprev[L];
        fe_load<L>(&pts[i * L], p);
        fe_load<L>(&tgt[i * L], w);
        fe_load<L>(&c[t * L], cur);
        fe_load<L>(&nn[t * L], ncur);
#pragma unroll
        for (int k = 0; k < L; ++k) {
            prev[k] = t > 0 ? c[(t - 1) * L + k] : 0;
            nprev[k] = t > 0 ? nn[(t - 1) * L + k] : 0;
        }
        u64 pc[L], pn[L], wz[L], a[L], b[L], zc[L];
        fe_mul<L>(p, cur, pc);
        fe_mul<L>(p, ncur, pn);
        fe_mul<L>(w, cur, wz);
        fe_sub<L>(nprev, pn, a);
        fe_add<L>(a, wz, b);
        fe_sub<L>(prev, pc, zc);
        fe_store<L>(&nb[cur_buf ^ 1][t * L], b);
        fe_store<L>(&zb[cur_buf ^ 1][t * L], zc);
        __syncthreads();
        cur_buf ^= 1;
    }
    u64 v[L];
    fe_load<L>(&nb[cur_buf][t * L], v);
    fe_store<L>(N + (row * M + leaf * d + t) * L, v);
}

// The same leaf interpolants from the leaf zerofiers the tree already holds (tails of level 0), without the d barrier-separated
// steps:  N = sum_i w_i Z / (x - p_i).   Thread i (group 0) divides Z by (x - p_i) synthetically -- q_{d-1} = 1,
// q_{j-1} = z_j + p_i q_j: one product per step, no barrier -- into column i of a d x d matrix in LDS; then S threads per coefficient
// j add up  sum_i w_i q_{i,j}  over a quarter of the points each (independent products), and group 0 adds the S partial sums.
// grid = (leaves, rows), block = d * S threads;  LDS: z, w (d L each), the matrix d x (d + 1) x L (rows padded: the second phase
// reads it by rows with the lanes along j), S d L partial sums.
template <int L, int S>
__global__ void __launch_bounds__(kLeafMax) leaf_interpolant_div_kernel(const u64* points, const u64* values, const u64* winv, const u64* tails0,
                                                                        long long n_points, int d, long long M, u64* N) {
    extern __shared__ u64 leaf_lds[];
    u64* z = leaf_lds;
    u64* w = z + (size_t)d * L;
    u64* Q = w + (size_t)d * L;
    const int stride = d + 1;
    u64* part = Q + (size_t)d * stride * L;
    const int t = threadIdx.x, s = t / d, j = t - s * d;
    const long long leaf = blockIdx.x, row = blockIdx.y;
    u64 p[L];
#pragma unroll
    for (int k = 0; k < L; ++k) p[k] = 0;
    if (s == 0) {
        const long long pi = leaf * d + j;
        u64 tv[L], zj[L];
#pragma unroll
        for (int k = 0; k < L; ++k) tv[k] = 0;
        if (pi < n_points) {
            u64 v[L], wi[L];
            fe_load<L>(points + pi * L, p);
            fe_load<L>(values + (row * n_points + pi) * L, v);
            fe_load<L>(winv + pi * L, wi);
            fe_mul<L>(v, wi, tv);
        }
        fe_store<L>(&w[j * L], tv);
        fe_load<L>(tails0 + (leaf * d + j) * L, zj);
        fe_store<L>(&z[j * L], zj);
    }
    __syncthreads();
    if (s == 0) {  // column j of Q: the quotient Z / (x - p_j)
        u64 q[L];
#pragma unroll
        for (int k = 0; k < L; ++k) q[k] = k ? 0 : gl::ONE;
        for (int c = d - 1; c >= 0; --c) {
            fe_store<L>(&Q[((size_t)c * stride + j) * L], q);
            u64 zc[L], m[L], r[L];
            fe_load<L>(&z[c * L], zc);
            fe_mul<L>(p, q, m);
            fe_add<L>(zc, m, r);
#pragma unroll
            for (int k = 0; k < L; ++k) q[k] = r[k];
        }
    }
    __syncthreads();
    {
        const int seg = d / S, i0 = s * seg;
        u64 acc[L];
#pragma unroll
        for (int k = 0; k < L; ++k) acc[k] = 0;
        for (int i = i0; i < i0 + seg; ++i) {
            u64 wi[L], qi[L], m[L], r[L];
            fe_load<L>(&w[i * L], wi);
            fe_load<L>(&Q[((size_t)j * stride + i) * L], qi);
            fe_mul<L>(wi, qi, m);
            fe_add<L>(acc, m, r);
#pragma unroll
            for (int k = 0; k < L; ++k) acc[k] = r[k];
        }
        fe_store<L>(&part[((size_t)s * d + j) * L], acc);
    }
    __syncthreads();
    if (s != 0) return;
    u64 a[L];
    fe_load<L>(&part[(size_t)j * L], a);
#pragma unroll
    for (int q = 1; q < S; ++q) {
        u64 b[L], r[L];
        fe_load<L>(&part[((size_t)q * d + j) * L], b);
        fe_add<L>(a, b, r);
#pragma unroll
        for (int k = 0; k < L; ++k) a[k] = r[k];
    }
    fe_store<L>(N + (row * M + leaf * d + j) * L, a);
}

// N_parent = N_left Z_right + N_right Z_left in the transform domain of order 2d (deg N_parent < 2d: no wrap-around).
//   Nh: forward transforms of the children's interpolants, [rows * children][2d]   (d coefficients each, zero padded)
//   Th: forward transforms of the children's zerofier TAILS, [children][2d], shared by the rows; the leading x^d of a zerofier
//       transforms to w_2d^(d k) = (-1)^k, added here
//   out[(row * parents + node) * 2d + k] = Nh_left[k] (Th_right[k] + (-1)^k) + Nh_right[k] (Th_left[k] + (-1)^k)
// The inverse transform of `out`, in place, IS the next level's [rows][M] array.
template <int L>
__global__ void __launch_bounds__(256) interpolant_pointwise_kernel(const u64* Nh, const u64* Th, u64* out, long long d, long long parents,
                                                                    long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * parents * 2 * d) return;
    const long long k = i % (2 * d), gnode = i / (2 * d), node = gnode % parents;
    u64 nl[L], nr[L], zl[L], zr[L], a[L], b[L], r[L];
    fe_load<L>(Nh + ((2 * gnode) * 2 * d + k) * L, nl);
    fe_load<L>(Nh + ((2 * gnode + 1) * 2 * d + k) * L, nr);
    fe_load<L>(Th + ((2 * node) * 2 * d + k) * L, zl);
    fe_load<L>(Th + ((2 * node + 1) * 2 * d + k) * L, zr);
    if (k & 1) {
        zl[0] = gl::sub(zl[0], gl::ONE);
        zr[0] = gl::sub(zr[0], gl::ONE);
    } else {
        zl[0] = gl::add(zl[0], gl::ONE);
        zr[0] = gl::add(zr[0], gl::ONE);
    }
    fe_mul<L>(nl, zr, a);
    fe_mul<L>(nr, zl, b);
    fe_add<L>(a, b, r);
    fe_store<L>(out + i * L, r);
}

// out[row][j] = Nroot[row][j + M - n], j < n: the interpolant's n coefficients from the padded root
template <int L>
__global__ void __launch_bounds__(256) interpolant_unpad_kernel(const u64* Nroot, long long M, long long n, u64* out) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long long row = blockIdx.y;
#pragma unroll
    for (int q = 0; q < L; ++q) out[(row * n + j) * L + q] = Nroot[(row * M + j + (M - n)) * L + q];
}

// ---- clean division over BFieldElement (Polynomial::clean_divide, polynomial.rs:2358-2411) ----------------------------------
// The reference moves dividend and divisor to the coset X * <w> of the EXTENSION field (X = the element x of
// F_p[x]/(x^3 - x + 1): a polynomial over the base field has no roots there unless it has an irreducible cubic factor),
// divides pointwise and interpolates.  Same steps here, every one a device pass.

// base^e by square-and-multiply (e < 2^32)
__device__ __forceinline__ void xfe_pow(const u64 (&base)[3], unsigned long long e, u64 (&r)[3]) {
    u64 sq[3] = {base[0], base[1], base[2]}, t[3];
    r[0] = gl::ONE;
    r[1] = 0;
    r[2] = 0;
    while (e) {
        if (e & 1) {
            xfe_mul(r, sq, t);
            r[0] = t[0], r[1] = t[1], r[2] = t[2];
        }
        e >>= 1;
        if (e) {
            xfe_mul(sq, sq, t);
            sq[0] = t[0], sq[1] = t[1], sq[2] = t[2];
        }
    }
}

// out[row][i] = c[row][i] * base^i (an XFieldElement) for i < n_c, zero up to order: Polynomial::scale with an extension-field offset
// (polynomial.rs:760-773) followed by the zero padding of :2391-2392.  grid.y = row (rows n_c words apart in, 3 * order words apart out)
__global__ void __launch_bounds__(256) lift_scale_kernel(const u64* c, long long n_c, long long order, u64* out, u64 b0, u64 b1, u64 b2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= order) return;
    c += (long long)blockIdx.y * n_c;
    out += (long long)blockIdx.y * order * 3;
    u64 r[3] = {0, 0, 0};
    if (i < n_c) {
        const u64 base[3] = {b0, b1, b2};
        u64 pw[3];
        xfe_pow(base, (unsigned long long)i, pw);
        const u64 ci = c[i];
        r[0] = gl::mont_mul(pw[0], ci);
        r[1] = gl::mont_mul(pw[1], ci);
        r[2] = gl::mont_mul(pw[2], ci);
    }
    fe_store<3>(out + 3 * i, r);
}

// b[i] <- 1 / b[i] in place; flag bit 0 when some b[i] is zero (batch_inversion panics, traits.rs:106)
__global__ void __launch_bounds__(256) xfe_invert_inplace_kernel(u64* b, long long count, int* flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 y[3], inv[3];
    fe_load<3>(b + 3 * i, y);
    if (!xfe_inverse(y, inv)) atomicOr(flag, 1);
    fe_store<3>(b + 3 * i, inv);
}

// out[row][i] = unlift(q[row][i] * base_inv^i) for i < n_q; flag bit 1 when a coefficient does not come back to the base field
// (unlift().unwrap() panics, :2410) or a coefficient beyond the quotient's degree is not zero (the division was not clean).
// grid.y = row (rows 3 * order words apart in, n_q words apart out)
__global__ void __launch_bounds__(256) unscale_unlift_kernel(const u64* q, long long order, long long n_q, u64* out, u64 b0, u64 b1, u64 b2,
                                                            int* flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= order) return;
    q += (long long)blockIdx.y * order * 3;
    out += (long long)blockIdx.y * n_q;
    const u64 base[3] = {b0, b1, b2};
    u64 pw[3], v[3], r[3];
    xfe_pow(base, (unsigned long long)i, pw);
    fe_load<3>(q + 3 * i, v);
    xfe_mul(v, pw, r);
    if (r[1] | r[2] | (i >= n_q ? r[0] : 0)) atomicOr(flag, 2);
    if (i < n_q) out[i] = r[0];
}

// out[b][i] = in[b][i] * base^i (XFieldElement coefficients and base) for i < n_in, zero for n_in <= i < n_out:
// Polynomial::scale with an extension-field offset (polynomial.rs:760-773) plus the zero padding of fast_coset_evaluate (:1396).
// In place when in == out and the geometry matches (every thread owns its element).
__global__ void __launch_bounds__(256) xfe_scale_kernel(const u64* in, long long n_in, long long in_stride, u64* out, long long n_out,
                                                        long long batch, u64 b0, u64 b1, u64 b2) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_out * batch) return;
    const long long b = t / n_out, i = t - b * n_out;
    u64 r[3] = {0, 0, 0};
    if (i < n_in) {
        const u64 base[3] = {b0, b1, b2};
        u64 pw[3], v[3];
        xfe_pow(base, (unsigned long long)i, pw);
        fe_load<3>(in + b * in_stride + 3 * i, v);
        xfe_mul(v, pw, r);
    }
    fe_store<3>(out + 3 * t, r);
}

// ---- barycentric evaluation (polynomial.rs:2609-2637) ------------------------------------------------------------------------
// The interpolant of a codeword on the subgroup <w_n> at a point x outside it:  sum_i c_i d_i / (x - d_i)  over
// sum_i d_i / (x - d_i), d_i = w^i.  The weights d_i / (x - d_i) depend on x alone and are shared by every codeword of a batch;
// a codeword is read exactly once (the path through iNTT + Horner moves five times the bytes).  All arithmetic in the extension
// field; a base-field point or codeword is its lift.

// w[i] = d_i / (x - d_i), d_i = omega^i.  Thread t of a block takes i = base + t + 256 j, j < kBaryPerThread: d by one power and
// kBaryPerThread - 1 products with omega^256, the kBaryPerThread inverses by Montgomery's trick (one inversion per thread).
constexpr int kBaryPerThread = 8;
__global__ void __launch_bounds__(256) barycentric_weights_kernel(long long n, int log_n, u64 omega, u64 omega_256, u64 x0, u64 x1, u64 x2, u64* w) {
    const long long base = (long long)blockIdx.x * (kBaryPerThread * 256) + threadIdx.x;
    u64 d[kBaryPerThread], pre[kBaryPerThread][3];
    {   // omega^base by square-and-multiply over the log_n bits an index has
        u64 acc = gl::ONE, sq = omega;
        for (int b = 0; b < log_n; ++b) {
            if ((base >> b) & 1) acc = gl::mont_mul(acc, sq);
            sq = gl::mont_mul(sq, sq);
        }
        d[0] = acc;
    }
#pragma unroll
    for (int j = 1; j < kBaryPerThread; ++j) d[j] = gl::mont_mul(d[j - 1], omega_256);
    // prefix products of the shifts x - d_j (indices beyond n use the shift 1: they are never stored)
    u64 run[3] = {gl::ONE, 0, 0};
#pragma unroll
    for (int j = 0; j < kBaryPerThread; ++j) {
        pre[j][0] = run[0], pre[j][1] = run[1], pre[j][2] = run[2];
        const bool live = base + 256 * j < n;
        const u64 sh[3] = {live ? gl::sub(x0, d[j]) : gl::ONE, live ? x1 : 0, live ? x2 : 0};
        u64 t[3];
        xfe_mul(run, sh, t);
        run[0] = t[0], run[1] = t[1], run[2] = t[2];
    }
    u64 inv[3];
    xfe_inverse(run, inv);  // x is outside the subgroup (checked on the host): no shift is zero
#pragma unroll
    for (int j = kBaryPerThread - 1; j >= 0; --j) {
        const long long i = base + 256 * j;
        const bool live = i < n;
        const u64 sh[3] = {live ? gl::sub(x0, d[j]) : gl::ONE, live ? x1 : 0, live ? x2 : 0};
        u64 one_over[3], t[3];
        xfe_mul(inv, pre[j], one_over);  // 1 / (x - d_j)
        xfe_mul(inv, sh, t);             // drop this shift from the running inverse
        inv[0] = t[0], inv[1] = t[1], inv[2] = t[2];
        if (live) {
            const u64 r[3] = {gl::mont_mul(one_over[0], d[j]), gl::mont_mul(one_over[1], d[j]), gl::mont_mul(one_over[2], d[j])};
            fe_store<3>(w + 3 * i, r);
        }
    }
}

// partial[(row * n_chunks + chunk) * 3] = sum over the chunk's i of c_row[i] * w[i];  row == batch: the denominator (c = 1).
// grid = (n_chunks, ceil((batch + 1) / rows_per_block)), rows_per_block <= kBaryRows; a chunk is kBaryPerThread * 256 consecutive i, thread t takes i = base + t + 256 j
// (coalesced); the block keeps its weights in registers and walks kBaryRows rows with them.
constexpr int kBaryRows = 8;
template <int CW>
__global__ void __launch_bounds__(256) barycentric_partial_kernel(const u64* codewords, const u64* w, long long n, long long batch, u64* partial,
                                                                  int rows_per_block) {
    __shared__ u64 red[kBaryRows][4][3];  // one slot per (row of the group, wave): a single barrier serves all rows
    const int t = threadIdx.x;
    const long long chunk = blockIdx.x, base = chunk * (kBaryPerThread * 256);
#pragma unroll 1
    for (int rr = 0; rr < rows_per_block; ++rr) {
        const long long row = (long long)blockIdx.y * rows_per_block + rr;
        if (row > batch) break;
        u64 acc[3] = {0, 0, 0};
        // (the weights are re-read per row: they stay in L1 / L2; keeping them in registers instead costs the occupancy this
        // latency-bound loop lives on -- 1.22 vs 0.8 ms for 256 codewords of 2^20 -- and staging them in LDS for 2 .. 16 rows per
        // block, 48 KB a block, measured 1.32 - 1.67 ms against 1.04)
#pragma unroll
        for (int j = 0; j < kBaryPerThread; ++j) {
            const long long i = base + t + 256 * j;
            if (i < n) {
                u64 wi[3], term[3], sum[3];
                fe_load<3>(w + 3 * i, wi);
                if (row == batch) {
                    term[0] = wi[0], term[1] = wi[1], term[2] = wi[2];
                } else if constexpr (CW == 1) {
                    const u64 c = codewords[row * n + i];
                    term[0] = gl::mont_mul(wi[0], c), term[1] = gl::mont_mul(wi[1], c), term[2] = gl::mont_mul(wi[2], c);
                } else {
                    u64 c[3];
                    fe_load<3>(codewords + (row * n + i) * 3, c);
                    xfe_mul(c, wi, term);
                }
                fe_add<3>(acc, term, sum);
                acc[0] = sum[0], acc[1] = sum[1], acc[2] = sum[2];
            }
        }
        // the wave's 64 partial sums by shuffles, one LDS slot per wave
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = gl::add(acc[q], (u64)__shfl_down((unsigned long long)acc[q], off, 64));
        }
        if ((t & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) red[rr][t >> 6][q] = acc[q];
        }
    }
    __syncthreads();
    if (t < rows_per_block) {
        const long long row = (long long)blockIdx.y * rows_per_block + t;
        if (row <= batch) {
            u64 r[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) r[q] = gl::add(gl::add(red[t][0][q], red[t][1][q]), gl::add(red[t][2][q], red[t][3][q]));
            fe_store<3>(partial + (row * gridDim.x + chunk) * 3, r);
        }
    }
}

// out[row] = (sum of the row's partials) / (sum of the denominator's partials); one 64-lane workgroup per codeword
__global__ void __launch_bounds__(64) barycentric_finish_kernel(const u64* partial, long long n_chunks, long long batch, u64* out) {
    const long long row = blockIdx.x;
    const int t = threadIdx.x;
    u64 num[3] = {0, 0, 0}, den[3] = {0, 0, 0};
    for (long long c = t; c < n_chunks; c += 64) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            num[q] = gl::add(num[q], partial[(row * n_chunks + c) * 3 + q]);
            den[q] = gl::add(den[q], partial[(batch * n_chunks + c) * 3 + q]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            num[q] = gl::add(num[q], (u64)__shfl_down((unsigned long long)num[q], off, 64));
            den[q] = gl::add(den[q], (u64)__shfl_down((unsigned long long)den[q], off, 64));
        }
    }
    if (t == 0) {
        u64 dinv[3], r[3];
        xfe_inverse(den, dinv);  // sum_i d_i / (x - d_i) = n / (x^n - 1): non-zero for x outside the subgroup
        xfe_mul(num, dinv, r);
        fe_store<3>(out + 3 * row, r);
    }
}

}  // namespace tfk
