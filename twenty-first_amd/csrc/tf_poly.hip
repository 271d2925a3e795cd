// tf_poly.hip -- the callers on either side of the hot path: orchestration over poly_kernels.h and the
// transforms of tf_ntt.hip.
#include "tf_internal.h"
#include "poly_kernels.h"

namespace tfi {

// ------------------------------------------------------------------------------------ device-resident chain
// fast_coset_interpolate (polynomial.rs:1907-1918): intt, then coefficient j times offset^-j (fused into the last pass).
int coset_interp_dev(const u64* d_values, size_t n, u64 offset_raw, u64* d_out, size_t batch, int L, void* stream) {
    int rc = check_len(n);
    if (rc) return rc;
    if (n == 0 || batch == 0) return TF_OK;
    if (!d_values || !d_out) return TF_ERR_NULL_POINTER;
    if (offset_raw == 0) return TF_ERR_INVERSE_OF_ZERO;  // offset.inverse() panics on zero (b_field_element.rs:264-268)
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    const u64* pw = nullptr;
    bool temp = false;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = get_pow_table(ctx, gl::mont_inverse(offset_raw), n, s, &pw, &temp);
    if (rc) return rc;
    rc = run_ntt(ctx, d_values, d_out, (long long)n * L, (long long)n * L, n, batch, L, true, nullptr, -1, s, pw);
    release_pow_table(ctx, pw, temp, s);
    return rc;
}

int hadamard_dev(const u64* a, const u64* b, u64* out, size_t count, int L, void* stream) {
    if (count == 0) return TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long blocks = std::min<long long>(((long long)count + 255) / 256, 256 * 32);
    if (L == 1)
        hipLaunchKernelGGL(tfk::hadamard_bfe_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, out, (long long)count);
    else
        hipLaunchKernelGGL(tfk::hadamard_xfe_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, out, (long long)count);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// one thread per item, 256-thread blocks (the glue kernels of poly_kernels.h)
template <int L, class K, class... Args>
int launch_1d(K kernel, long long threads, hipStream_t s, Args... args) {
    if (threads <= 0) return TF_OK;
    hipLaunchKernelGGL(kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, args...);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

int pad_copy(const u64* src, u64* dst, long long n_src_words, long long n_dst_words, long long batch, hipStream_t s,
             long long src_stride_words = 0) {
    if (n_dst_words * batch == 0) return TF_OK;
    const long long blocks = std::min<long long>((n_dst_words * batch + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(tfk::pad_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n_src_words, n_dst_words, batch,
                       src_stride_words ? src_stride_words : n_src_words);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// Polynomial::fast_multiply (polynomial.rs:900-932): zero-pad both to order = next_power_of_two(deg a + deg b + 1),
// ntt both, pointwise product, intt, truncate to na + nb - 1 coefficients.  (The reference then trims leading zero
// coefficients in Polynomial::new; the caller does that -- the length here is data independent.)
// a_bs / b_bs: words between consecutive polynomials of the batch (0: packed, na * L / nb * L).
int poly_mul_dev(const u64* a, size_t na, const u64* b, size_t nb, u64* out, size_t batch, int L, void* stream, long long a_bs,
                 long long b_bs) {
    if (batch == 0 || na == 0 || nb == 0) return TF_OK;  // a zero polynomial: empty product
    if (!a_bs) a_bs = (long long)na * L;
    if (!b_bs) b_bs = (long long)nb * L;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    const size_t n_out = na + nb - 1;
    size_t order = 1;
    while (order < n_out) order <<= 1;
    int rc = check_len(order);
    if (rc) return rc;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    u64* tmp = nullptr;
    const size_t half = batch * order * size_t(L);
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&tmp), 2 * half * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(poly_mul)", __FILE__, __LINE__);
    static const bool no_fuse = ab_env("TF_POLY_MUL_NO_FUSE") != nullptr;  // A/B switch
    bool copied = false;
    if (order > 16 && !no_fuse) {
        // zero padding happens in the first pass of each forward transform (rows beyond the coefficients read as zero);
        // over BFieldElement the pointwise product rides on the inverse transform's first load
        rc = run_ntt(ctx, a, tmp, a_bs, (long long)order * L, order, batch, L, false, nullptr, (long long)na, s);
        if (!rc) rc = run_ntt(ctx, b, tmp + half, b_bs, (long long)order * L, order, batch, L, false, nullptr, (long long)nb, s);
        const bool trunc = can_truncate(order, L);  // the inverse's last pass writes the n_out coefficients straight to `out`
        u64* dst = trunc ? out : tmp;
        const long long dst_bs = trunc ? (long long)n_out * L : (long long)order * L;
        if (!rc && L == 1) {
            rc = run_ntt(ctx, tmp, dst, (long long)order, dst_bs, order, batch, 1, true, nullptr, -1, s, nullptr, 1, tmp + half,
                         trunc ? (long long)n_out : -1);
        } else if (!rc) {
            rc = hadamard_dev(tmp, tmp + half, tmp, batch * order, L, s);
            if (!rc) rc = run_ntt(ctx, tmp, dst, (long long)order * L, dst_bs, order, batch, L, true, nullptr, -1, s, nullptr, 1, nullptr,
                                  trunc ? (long long)n_out : -1);
        }
        copied = trunc;
    } else {
        rc = pad_copy(a, tmp, (long long)na * L, (long long)order * L, (long long)batch, s, a_bs);
        if (!rc) rc = pad_copy(b, tmp + half, (long long)nb * L, (long long)order * L, (long long)batch, s, b_bs);
        if (!rc) rc = run_ntt(ctx, tmp, tmp, (long long)order * L, (long long)order * L, order, 2 * batch, L, false, nullptr, -1, s);
        if (!rc) rc = hadamard_dev(tmp, tmp + half, tmp, batch * order, L, s);
        if (!rc) rc = run_ntt(ctx, tmp, tmp, (long long)order * L, (long long)order * L, order, batch, L, true, nullptr, -1, s);
    }
    if (!rc && !copied) rc = pad_copy(tmp, out, (long long)order * L, (long long)n_out * L, (long long)batch, s);
    hipError_t e2 = hipFreeAsync(tmp, s);
    if (rc) return rc;
    if (e2 != hipSuccess) return hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}

// fast_multiply of `batch` polynomials by ONE polynomial b (a table of numerators times the same zerofier; polynomial.rs:900-932
// per product): b is transformed once and its transform broadcast.  out: batch x (na + nb - 1) coefficients.
int poly_mul_shared_dev(const u64* a, size_t na, size_t batch, const u64* b, size_t nb, u64* out, int L, void* stream) {
    if (batch == 0 || na == 0 || nb == 0) return TF_OK;  // a zero polynomial: empty products
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    const size_t n_out = na + nb - 1;
    size_t order = 1;
    while (order < n_out) order <<= 1;
    int rc = check_len(order);
    if (rc) return rc;
    if (order <= 16) {  // tiny products: the plain batched route with b repeated is not worth a special case -- one product at a time
        for (size_t k = 0; k < batch && !rc; ++k) rc = poly_mul_dev(a + k * na * L, na, b, nb, out + k * n_out * L, 1, L, stream);
        return rc;
    }
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    u64* tmp = nullptr;  // batch transforms of a, one of b
    const size_t row = order * size_t(L);
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&tmp), (batch + 1) * row * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(poly_mul_shared)", __FILE__, __LINE__);
    u64* bh = tmp + batch * row;
    rc = run_ntt(ctx, a, tmp, (long long)na * L, (long long)row, order, batch, L, false, nullptr, (long long)na, s);
    if (!rc) rc = run_ntt(ctx, b, bh, (long long)nb * L, (long long)row, order, 1, L, false, nullptr, (long long)nb, s);
    if (!rc) rc = L == 1 ? launch_1d<1>(tfk::product_bcast_kernel<1>, (long long)(batch * order), s, (const u64*)tmp, (const u64*)bh, tmp, (long long)order,
                                        (long long)(batch * order))
                         : launch_1d<3>(tfk::product_bcast_kernel<3>, (long long)(batch * order), s, (const u64*)tmp, (const u64*)bh, tmp, (long long)order,
                                        (long long)(batch * order));
    if (!rc) rc = run_ntt(ctx, tmp, tmp, (long long)row, (long long)row, order, batch, L, true, nullptr, -1, s);
    if (!rc) rc = pad_copy(tmp, out, (long long)row, (long long)(n_out * L), (long long)batch, s);
    hipError_t e2 = hipFreeAsync(tmp, s);
    if (rc) return rc;
    if (e2 != hipSuccess) return hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}

// Polynomial::fast_square (polynomial.rs:780-798): one forward transform instead of two.
int poly_square_dev(const u64* a, size_t na, u64* out, size_t batch, int L, void* stream) {
    if (batch == 0 || na == 0) return TF_OK;
    if (!a || !out) return TF_ERR_NULL_POINTER;
    const size_t n_out = 2 * na - 1;
    size_t order = 1;
    while (order < n_out) order <<= 1;
    int rc = check_len(order);
    if (rc) return rc;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    u64* tmp = nullptr;
    const size_t words = batch * order * size_t(L);
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&tmp), words * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(poly_square)", __FILE__, __LINE__);
    static const bool no_fuse = ab_env("TF_POLY_MUL_NO_FUSE") != nullptr;  // A/B switch
    bool copied = false;
    if (order > 16 && !no_fuse) {
        rc = run_ntt(ctx, a, tmp, (long long)na * L, (long long)order * L, order, batch, L, false, nullptr, (long long)na, s);
        const bool trunc = can_truncate(order, L);
        u64* dst = trunc ? out : tmp;
        const long long dst_bs = trunc ? (long long)n_out * L : (long long)order * L;
        if (!rc && L == 1) {
            rc = run_ntt(ctx, tmp, dst, (long long)order, dst_bs, order, batch, 1, true, nullptr, -1, s, nullptr, 1, tmp,
                         trunc ? (long long)n_out : -1);
        } else if (!rc) {
            rc = hadamard_dev(tmp, tmp, tmp, batch * order, L, s);
            if (!rc) rc = run_ntt(ctx, tmp, dst, (long long)order * L, dst_bs, order, batch, L, true, nullptr, -1, s, nullptr, 1, nullptr,
                                  trunc ? (long long)n_out : -1);
        }
        copied = trunc;
    } else {
        rc = pad_copy(a, tmp, (long long)na * L, (long long)order * L, (long long)batch, s);
        if (!rc) rc = run_ntt(ctx, tmp, tmp, (long long)order * L, (long long)order * L, order, batch, L, false, nullptr, -1, s);
        if (!rc) rc = hadamard_dev(tmp, tmp, tmp, batch * order, L, s);
        if (!rc) rc = run_ntt(ctx, tmp, tmp, (long long)order * L, (long long)order * L, order, batch, L, true, nullptr, -1, s);
    }
    if (!rc && !copied) rc = pad_copy(tmp, out, (long long)order * L, (long long)n_out * L, (long long)batch, s);
    hipError_t e2 = hipFreeAsync(tmp, s);
    if (rc) return rc;
    if (e2 != hipSuccess) return hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}

// Low-degree extension: values on {offset_in * w_n^i} -> values on {offset_out * w_m^i}, m >= n
// (= fast_coset_interpolate then fast_coset_evaluate with the coefficients staying in HBM).
int lde_dev(const u64* values, size_t n, u64 offset_in, u64* out, size_t m, u64 offset_out, size_t batch, int L, void* stream) {
    if (n > m) return TF_ERR_ORDER_NOT_ABOVE_DEGREE;
    int rc = check_len(n);
    if (!rc) rc = check_len(m);
    if (rc) return rc;
    if (m == 0 || batch == 0) return TF_OK;
    if (!out || (n && !values)) return TF_ERR_NULL_POINTER;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        HIPCHK(hipMemsetAsync(out, 0, m * batch * size_t(L) * sizeof(u64), s));
        return TF_OK;
    }
    u64* coeffs = nullptr;
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&coeffs), batch * n * size_t(L) * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(lde)", __FILE__, __LINE__);
    rc = coset_interp_dev(values, n, offset_in, coeffs, batch, L, s);
    if (!rc) rc = coset_eval_dev(coeffs, n, offset_out, out, m, batch, L, s);
    hipError_t e2 = hipFreeAsync(coeffs, s);
    if (rc) return rc;
    if (e2 != hipSuccess) return hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}


// ---- zerofier-tree evaluation (poly_kernels.h has the scheme) --------------------------------------------------------------
// Sub-quadratic counterpart of the Horner kernels for many points on a long polynomial: O((n + m) log^2 m) instead of O(n m).
// `units` polynomials of `len` <= M coefficients each (packed, unit u at F + u * len * L) are evaluated at the n_points points;
// vals[(u * M + i) * L] = unit_u(points[i]).  M = kTreeLeaf * 2^h >= n_points is the padded point count.
// Leaf size: 256 points over BFieldElement, 128 over XFieldElement (nine base-field products per step make the quadratic leaf
// work expensive).  With the levels in the transform domain (10 launches per level to build, 7 to walk) a level costs less than
// the O(leaf^2) work of a bigger leaf on the few workgroups a small tree has: n = m = 2^16 BFE 2.56 ms with 1024-point leaves,
// 2.30 with 512, 2.33 with 256, 2.46 with 128; 2^20 x 2^20 6.7 / 5.7 / 5.4 / 5.5; XFE 2^20 x 2^20 24.1 / 17.0 / 14.0 / 13.6
// (tools/leaf_ab.sh, profiles/r02_leaf_ab.txt).
inline int tree_leaf_log(int L) {  // TF_TREE_LEAF_LOG = 6..10 overrides both fields (A/B: tools/batch_eval_sweep.py)
    static const int forced = [] {
        const char* e = ab_env("TF_TREE_LEAF_LOG");
        const int v = e ? atoi(e) : 0;
        return (v >= 6 && v <= 10) ? v : 0;
    }();
    return forced ? forced : (L == 1 ? 8 : 7);
}
inline int tree_leaf(int L) { return 1 << tree_leaf_log(L); }
// Trees that are also walked UPWARDS (interpolation; the padded trees behind zerofier / interpolate / the ZerofierTree handle) are
// built down to 64-point leaves: the interpolant of a leaf is d sequential steps of ~0.4 us each (leaf_interpolant_kernel), 102 us
// for 256 points, while a level costs ~15 us since the latency-shaped transform -- prepared-tree interpolation of 2^12 points
// 176 -> 122 us, 2^16 434 -> 383, XFieldElement 2^12 358 -> 234 (tools/tree_latency.py, profiles/r03_tree_latency_leaf.txt).
// Evaluations on such a tree stop their walk at the level whose nodes have tree_leaf(L) points (Horner is parallel over the
// points: the bigger leaf is the faster one there).  TF_TREE_INTERP_LEAF_LOG = 6..10 overrides.
inline int tree_interp_leaf(int L) {
    static const int forced = [] {
        const char* e = ab_env("TF_TREE_INTERP_LEAF_LOG");
        const int v = e ? atoi(e) : 0;
        return (v >= 6 && v <= 10) ? v : 0;
    }();
    return 1 << std::min(forced ? forced : 6, tree_leaf_log(L));
}

// widest walk (points in flight) that takes the four-threads-per-point leaf kernels (TF_TREE_LEAF_SPLIT_MAX: sweep hook).  Measured,
// tools/tree_latency.py with the limit lifted: 2^16 points evaluate 726 -> 681 us (XFE), interpolate 265 -> 253 (BFE) but 450 -> 487
// (XFE: 109 KB of LDS per leaf), level at 2^18, XFE interpolation 1.5 x slower at 2^20 -- hence 2^16 / 2^16 / 2^15.
inline long long leaf_split_max() {
    static const long long v = [] {
        const char* e = ab_env("TF_TREE_LEAF_SPLIT_MAX");
        return e ? atoll(e) : (1ll << 15);
    }();
    return v;
}

struct ZerofierTree {
    int leaf = 0;              // points per leaf (tree_leaf(L) for a tree that is only evaluated on, tree_interp_leaf(L) otherwise)
    int h = 0;                 // levels 0 .. h-1 hold zerofiers of degree leaf << level (the root, level h, is never needed)
    long long M = 0;           // padded point count = leaf << h
    std::vector<u64*> tails;   // [level]: (M / d) nodes x d elements
    std::vector<u64*> inv;     // [level]: power-series inverses of the reversed zerofiers, precision d
    std::vector<u64*> That;    // [level]: forward transforms of order 2d of the tails   (M / d) x 2d
    std::vector<u64*> Ghat;    // [level]: forward transforms of order 2d of the inverses
};
constexpr int kTreeLevelArrays = 6;  // M-element arrays per level: tails, inv, That (2), Ghat (2)
constexpr int kTreeWorkArrays = 8;   // M-element arrays of work space shared by the build and the walks

// inverse transform of the pointwise product a^ * b^ (batch entries in_bs words apart in both), L words per element.  Over
// BFieldElement the product rides on the transform's first load; over XFieldElement it is a pass of its own into `out`.
template <int L>
int inverse_of_product(DeviceCtx* ctx, const u64* a_hat, const u64* b_hat, long long in_bs, u64* out, size_t order, size_t batch, bool pairs,
                       hipStream_t s) {
    if constexpr (L == 1) {
        return run_ntt(ctx, a_hat, out, in_bs, (long long)order, order, batch, 1, true, nullptr, -1, s, nullptr, 1, b_hat, -1);
    } else {
        int rc;
        if (pairs)  // a_hat / b_hat are the even / odd rows of one array
            rc = launch_1d<L>(tfk::pair_product_kernel<L>, (long long)(batch * order), s, a_hat, out, (long long)order, (long long)batch);
        else
            rc = hadamard_dev(a_hat, b_hat, out, batch * order, L, s);
        if (rc) return rc;
        return run_ntt(ctx, out, out, (long long)order * L, (long long)order * L, order, batch, L, true, nullptr, -1, s);
    }
}

template <int L>
int zerofier_tree_build(DeviceCtx* ctx, const u64* points, long long n_points, ZerofierTree* T, u64* arena, u64* work, hipStream_t s) {
    // arena: kTreeLevelArrays * h level arrays of M * L words (they stay); work: kTreeWorkArrays * M * L words (only during the build)
    const long long M = T->M;
    const int h = T->h;
    T->tails.resize(h);
    T->inv.resize(h);
    T->That.resize(h);
    T->Ghat.resize(h);
    for (int l = 0; l < h; ++l) {
        u64* base = arena + (long long)(kTreeLevelArrays * l) * M * L;
        T->tails[l] = base;
        T->inv[l] = base + M * L;
        T->That[l] = base + 2 * M * L;
        T->Ghat[l] = base + 4 * M * L;
    }
    if (h == 0) return TF_OK;
    const int kTreeLeaf = T->leaf;
    if (3 * kTreeLeaf * L * sizeof(u64) > 48 * 1024) {  // (only with a leaf size forced through TF_TREE_LEAF_LOG, laboratory build)
        static std::atomic<unsigned long long> done_mask{0};
        if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::leaf_zerofier_kernel<L>), 160 * 1024, done_mask)) return rc_attr;
    }
    hipLaunchKernelGGL(tfk::leaf_zerofier_kernel<L>, dim3((unsigned)(M / kTreeLeaf)), dim3(kTreeLeaf), 3 * kTreeLeaf * L * sizeof(u64), s, points,
                       n_points, kTreeLeaf, T->tails[0], T->inv[0]);
    HIPCHK(hipGetLastError());
    for (int l = 0; l < h; ++l) {
        const long long d = (long long)kTreeLeaf << l, children = M / d, parents = children / 2;
        if (L == 1 && tree_build_level_wanted(2 * d, parents)) {
            // the whole level in one launch: this level's transforms and, unless it is the top one, the parents' tails and inverses
            tfk::TreeBuildArgs a{};
            a.tails = T->tails[l], a.inv = T->inv[l], a.that = T->That[l], a.ghat = T->Ghat[l], a.parents = parents;
            if (l + 1 < h) a.ptails = T->tails[l + 1], a.pinv = T->inv[l + 1];
            int rcl = launch_tree_build_level(ctx, ilog2((size_t)(2 * d)), a, s);
            if (rcl) return rcl;
            continue;
        }
        // transforms of order 2d of this level's tails and inverses: kept for the walks, and the parents are built from them
        // (ONE call: the level's tails and inverses are neighbours in the arena, and so are their transforms)
        static_assert(kTreeLevelArrays == 6, "tails | inv | That (2) | Ghat (2)");
        int rc = run_ntt(ctx, T->tails[l], T->That[l], d * L, 2 * d * L, (size_t)(2 * d), (size_t)(2 * children), L, false, nullptr, d, s);
        if (rc) return rc;
        if (l + 1 == h) break;
        u64* S1 = work;              // parents x 2d    g_left g_right (its low half is G)
        u64* B = work + M * L;       // 2 parents x 2d  Newton inputs G | H
        u64* C = work + 3 * M * L;   // 2 parents x 4d  their transforms; the G rows become g (2 - h g)
        // tails of the parents: (A^ + s)(B^ + s) - 1 pointwise, one inverse transform straight into the level array
        rc = launch_1d<L>(tfk::zerofier_pointwise_kernel<L>, parents * 2 * d, s, (const u64*)T->That[l], T->tails[l + 1], d, parents);
        if (!rc) rc = run_ntt(ctx, T->tails[l + 1], T->tails[l + 1], 2 * d * L, 2 * d * L, (size_t)(2 * d), (size_t)parents, L, true, nullptr, -1, s);
        // inverses of the parents: G = g_left g_right mod x^d, then one Newton step g <- G (2 - rev(Z) G) mod x^2d at order 4d
        if (!rc) rc = inverse_of_product<L>(ctx, T->Ghat[l], T->Ghat[l] + 2 * d * L, 4 * d * L, S1, (size_t)(2 * d), (size_t)parents, true, s);
        if (!rc) rc = launch_1d<L>(tfk::newton_inputs_kernel<L>, parents * 2 * d, s, (const u64*)S1, (const u64*)T->tails[l + 1], B, d, parents);
        if (!rc) rc = run_ntt(ctx, B, C, 2 * d * L, 4 * d * L, (size_t)(4 * d), (size_t)(2 * parents), L, false, nullptr, 2 * d, s);
        if (!rc) rc = launch_1d<L>(tfk::newton_pointwise_kernel<L>, parents * 4 * d, s, C, 4 * d, parents);
        if (!rc) rc = run_ntt(ctx, C, C, 4 * d * L, 4 * d * L, (size_t)(4 * d), (size_t)parents, L, true, nullptr, -1, s);
        if (!rc) rc = launch_1d<L>(tfk::poly_truncate_kernel<L>, parents * 2 * d, s, (const u64*)C, 4 * d, T->inv[l + 1], 2 * d, parents);
        if (rc) return rc;
    }
    return TF_OK;
}

// product of U * batch transforms `a_hat` with the level's cached transforms `b_hat` (shared by the U units), back in the coefficient
// domain in `out`: one unit over BFieldElement rides on the inverse transform's load, otherwise a pointwise pass of its own.
template <int L>
int inverse_of_cached_product(DeviceCtx* ctx, const u64* a_hat, const u64* b_hat, u64* out, size_t order, size_t batch, long long U, hipStream_t s) {
    if (U == 1) return inverse_of_product<L>(ctx, a_hat, b_hat, (long long)order * L, out, order, batch, false, s);
    const long long period = (long long)(batch * order), total = period * U;
    int rc = launch_1d<L>(tfk::product_bcast_kernel<L>, total, s, a_hat, b_hat, out, period, total);
    if (rc) return rc;
    return run_ntt(ctx, out, out, (long long)order * L, (long long)order * L, order, batch * (size_t)U, L, true, nullptr, -1, s);
}

// The elementwise steps of a walk (reverse, remainder, the interpolation's pointwise combination) ride on the load / store of the
// latency-shaped transform next to them whenever that kernel serves the level (round 3): a level of the walk down is 4 launches
// instead of 7, of the walk up 2 instead of 3.  TF_TREE_NO_FUSE keeps them as kernels of their own (A/B, tests).
bool tree_fuse(long long order, long long lines, int L) {
    static const bool off = ab_env("TF_TREE_NO_FUSE") != nullptr;
    if (off || order > 4096 || order < 64 || g_min_passes.load(std::memory_order_relaxed) != 0) return false;
    if (lines >= (1ll << 22)) return false;
    return lat_wanted(ilog2((size_t)order), (size_t)lines, L);
}

// F: U units of exactly M coefficients each (zero padded), walking the tree together; vals: U x M values (the first n_points of
// every unit are meaningful); work: kTreeWorkArrays * U * M * L words.
template <int L>
int zerofier_tree_evaluate(DeviceCtx* ctx, const ZerofierTree& T, const u64* F, const u64* points, long long n_points, u64* vals, u64* work,
                           hipStream_t s, long long U = 1) {
    const int kTreeLeaf = T.leaf;
    const long long M = T.M, UM = U * M;
    // the walk stops at the level whose nodes hold tree_leaf(L) points (a tree built for interpolation has smaller leaves)
    int l_stop = 0;
    while ((kTreeLeaf << l_stop) < tree_leaf(L) && l_stop < T.h) ++l_stop;
    const int eval_leaf = kTreeLeaf << l_stop;
    const u64* cur = F;  // remainders of the level above: U x (M / 2d) polynomials of 2d coefficients
    u64* ping = work;                // U M
    u64* pong = work + UM * L;       // U M
    u64* fr = work + 2 * UM * L;     // U x children x d      reversed upper halves, then the quotients
    u64* Fh = work + 3 * UM * L;     // U x children x 2d     their transforms
    u64* prod = work + 5 * UM * L;   // U x children x 2d     products back in the coefficient domain
    u64* frq = work + 7 * UM * L;    // U x children x d      the next level's reversed upper halves, written by this level's last kernel
    for (int l = T.h - 1; l >= l_stop; --l) {
        const long long d = (long long)kTreeLeaf << l, children = M / d, all = U * children;  // (children is even: global child / 2 = global parent)
        // rev(q) = rev(f_high) g mod x^d   (below the top level the reversed upper halves come from the level above's last kernel)
        int rc = TF_OK;
        u64* nxt = (cur == ping) ? pong : ping;
        if (tree_level_wanted(2 * d, all, L, false)) {
            // the whole level in one launch: a line's four transforms never leave LDS
            tfk::TreeLevelArgs a{};
            a.cur = cur, a.nxt = nxt, a.ghat = T.Ghat[l], a.that = T.That[l], a.lines = all, a.per = children;
            rc = launch_tree_level<false>(ctx, ilog2((size_t)(2 * d)), a, s, L);
            if (rc) return rc;
            cur = nxt;
            continue;
        }
        if (tree_fuse(2 * d, all, L)) {
            // the same steps with the reversals read on load and the remainder formed on store (ntt_lat_kernel's modifiers)
            const int lg = ilog2((size_t)(2 * d));
            tfk::NttLatArgs m{};
            m.load_mode = 1, m.src_shift = 1, m.rev_top = 2 * d - 1;  // line `child` <- reversed upper half of its parent's remainder
            rc = launch_lat(ctx, cur, Fh, 2 * d * L, 2 * d * L, lg, (size_t)all, L, false, d, nullptr, s, &m);
            if (!rc) rc = inverse_of_cached_product<L>(ctx, Fh, T.Ghat[l], prod, (size_t)(2 * d), (size_t)children, U, s);
            m.src_shift = 0, m.rev_top = d - 1;                        // q = the reversed low half of that product
            if (!rc) rc = launch_lat(ctx, prod, Fh, 2 * d * L, 2 * d * L, lg, (size_t)all, L, false, d, nullptr, s, &m);
            // r = f_low - (q tail)_low: the product's inverse transform stores f_low - value for the low d outputs only
            tfk::NttLatArgs st{};
            st.store_mode = 1, st.sub_src = cur, st.sub_bs = 2 * d * L, st.keep = d;
            if (!rc && U == 1 && L == 1) {
                rc = launch_lat(ctx, Fh, nxt, 2 * d, d, lg, (size_t)all, 1, true, -1, T.That[l], s, &st);
            } else if (!rc) {
                if (U == 1) rc = hadamard_dev(Fh, T.That[l], prod, (size_t)(all * 2 * d), L, s);
                else rc = launch_1d<L>(tfk::product_bcast_kernel<L>, all * 2 * d, s, (const u64*)Fh, (const u64*)T.That[l], prod, children * 2 * d, all * 2 * d);
                if (!rc) rc = launch_lat(ctx, prod, nxt, 2 * d * L, d * L, lg, (size_t)all, L, true, -1, nullptr, s, &st);
            }
            if (rc) return rc;
            cur = nxt;
            continue;
        }
        const u64* fr_in = frq;
        if (l == T.h - 1 || tree_fuse(4 * d, all / 2, L) || tree_level_wanted(4 * d, all / 2, L, false)) {  // (a fused level above this one did not write frq)
            rc = launch_1d<L>(tfk::remainder_rev_high_kernel<L>, all * d, s, cur, fr, d, all);
            fr_in = fr;
        }
        if (!rc) rc = run_ntt(ctx, fr_in, Fh, d * L, 2 * d * L, (size_t)(2 * d), (size_t)all, L, false, nullptr, d, s);
        if (!rc) rc = inverse_of_cached_product<L>(ctx, Fh, T.Ghat[l], prod, (size_t)(2 * d), (size_t)children, U, s);
        if (!rc) rc = launch_1d<L>(tfk::poly_reverse_kernel<L>, all * d, s, (const u64*)prod, 2 * d, fr, d, all);
        // r = f_low - (q tail)_low
        if (!rc) rc = run_ntt(ctx, fr, Fh, d * L, 2 * d * L, (size_t)(2 * d), (size_t)all, L, false, nullptr, d, s);
        if (!rc) rc = inverse_of_cached_product<L>(ctx, Fh, T.That[l], prod, (size_t)(2 * d), (size_t)children, U, s);
        if (rc) return rc;
        rc = launch_1d<L>(tfk::remainder_finish_kernel<L>, all * d, s, cur, (const u64*)prod, 2 * d, nxt, d, all, l > l_stop ? frq : (u64*)nullptr);
        if (rc) return rc;
        cur = nxt;
    }
    // few leaves: four threads per point (the chip is idle anyway; a thread's eval_leaf products in a row were 16 of the 116 us of
    // a 2^12-point walk).  TF_TREE_NO_LEAF_SPLIT: A/B switch.
    static const bool no_split = ab_env("TF_TREE_NO_LEAF_SPLIT") != nullptr;
    constexpr int kSplit = 4;
    if (!no_split && UM <= 2 * leaf_split_max() && eval_leaf * kSplit <= 1024 && eval_leaf >= 16 * kSplit) {
        hipLaunchKernelGGL((tfk::leaf_evaluate_split_kernel<L, kSplit>), dim3((unsigned)(UM / eval_leaf)), dim3(eval_leaf * kSplit),
                           (size_t)(1 + kSplit) * eval_leaf * L * sizeof(u64), s, cur, points, n_points, eval_leaf, vals, M / eval_leaf);
    } else {
        hipLaunchKernelGGL(tfk::leaf_evaluate_kernel<L>, dim3((unsigned)(UM / eval_leaf)), dim3(eval_leaf), eval_leaf * L * sizeof(u64), s, cur, points,
                           n_points, eval_leaf, vals, M / eval_leaf);
    }
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// When the tree pays (measured, tools/batch_eval_sweep.py): many points AND a long polynomial.  TF_BATCH_EVAL = horner | tree
// forces a route (A/B, tests).
std::atomic<int> g_batch_eval_route{-1};  // tf_set_batch_eval_route: 0 automatic, 1 Horner, 2 zerofier tree (-1: read TF_BATCH_EVAL)
bool tree_route(size_t n_coeffs, size_t n_points, size_t batch, int L) {
    int route = g_batch_eval_route.load(std::memory_order_relaxed);
    if (route < 0) {
        const char* e = ab_env("TF_BATCH_EVAL");
        route = !e ? 0 : (!strcmp(e, "horner") ? 1 : (!strcmp(e, "tree") ? 2 : 0));
        g_batch_eval_route.store(route, std::memory_order_relaxed);
    }
    const char* force = route == 1 ? "horner" : (route == 2 ? "tree" : nullptr);
    if (force && !strcmp(force, "horner")) return false;
    const size_t kTreeLeaf = (size_t)tree_leaf(L);
    if (n_points < kTreeLeaf * 2 || n_coeffs < 2) return false;
    size_t M = kTreeLeaf;
    while (M < n_points) M <<= 1;
    const size_t units = batch * ((n_coeffs + M - 1) / M);
    if (units > 65536) return false;  // (the walk's arrays are indexed per unit)
    {
        int h = 0;
        for (size_t v = kTreeLeaf; v < M; v <<= 1) ++h;
        // the tree and its build work space, then the walk's: padded coefficients and values (2 units M) + work for a slab of units
        const size_t slab = std::max<size_t>(1, std::min<size_t>(units, (size_t(1) << 25) / M));
        const size_t words = ((size_t)(kTreeLevelArrays * h + kTreeWorkArrays) + 2 * units + (size_t)kTreeWorkArrays * slab) * M * (size_t)L;
        if (words * sizeof(u64) > (size_t(64) << 30)) return false;  // would not fit a sane work space (288 GB of HBM)
    }
    if (force && !strcmp(force, "tree")) return true;
    // Cost model fitted to tools/batch_eval_sweep.py <width> fine on MI355X (profiles/r03_batch_eval_fine_w*.txt), milliseconds:
    //   Horner  n m / 1.4e9            (x 8 over XFieldElement: nine base-field products per step; measured 7 - 10)
    //   tree    build + one walk for the first unit: latency-bound per level up to 2^12 points (0.07 ms a level with one
    //           launch per level of the build and of the walk down, round 3), twice that per level above, plus a throughput term in M beyond 2^16 points;
    //           the units walk TOGETHER, so every further unit adds only its share of the throughput term: 0.03 ms per 2^16
    //           points (0.16 over XFE)
    int levels = 0;
    for (size_t v = kTreeLeaf; v < M; v <<= 1) ++levels;
    // (Horner is one thread per point: however few the points, a polynomial costs its n dependent steps -- 1.0 ns each, 3.1 over
    //  XFieldElement: 2^20 coefficients at 2^9 points 1.09 ms where the product term says 0.38)
    const double horner_ms = std::max((double)batch * (double)n_coeffs * (double)n_points / 1.4e9 * (L == 3 ? 8.0 : 1.0),
                                      (double)n_coeffs * (L == 3 ? 3.1e-6 : 1.0e-6));
    const double m16 = (double)M / 65536.0;
    const double first_ms = L == 3 ? 0.25 + 0.085 * levels + 0.10 * std::max(0, levels - 4) + 0.25 * m16
                                   : 0.09 + 0.07 * levels + 0.09 * std::max(0, levels - 4) + 0.055 * m16;
    const double tree_ms = first_ms + (double)(units - 1) * (L == 3 ? 0.16 : 0.03) * m16;
    return tree_ms < 0.95 * horner_ms;
}

// `batch` polynomials down an existing tree (levels >= 1).  A polynomial longer than M is cut into chunks of M coefficients; all
// chunks of all polynomials ("units") walk the tree TOGETHER, a slab of units at a time: every level is the same handful of
// launches whatever the number of units, and the level's cached transforms are shared.  Work space, the padded coefficients and
// the chunk values are this call's own stream-ordered temporaries, so one tree serves concurrent calls.
template <int L>
int tree_batch_evaluate(DeviceCtx* ctx, const ZerofierTree& T, const u64* points, size_t n_points, const u64* coeffs, size_t n_coeffs,
                        size_t poly_stride, size_t batch, u64* out, hipStream_t s) {
    const long long M = T.M;
    const size_t chunks = std::max<size_t>(1, (n_coeffs + (size_t)M - 1) / (size_t)M);
    const size_t units = batch * chunks, ML = (size_t)M * L;
    // units per walk: 2^25 elements of work per array (TF_TREE_UNIT_SLAB = elements: the A/B and test knob for the slab boundary)
    static const size_t slab_elems = [] { const char* e = ab_env("TF_TREE_UNIT_SLAB"); const long long v = e ? atoll(e) : 0; return v > 0 ? (size_t)v : (size_t(1) << 25); }();
    const size_t slab = std::max<size_t>(1, std::min<size_t>(units, slab_elems / (size_t)M));
    // padded coefficients (units M) + values (units M) + walk work (8 slab M)
    u64* tmp = nullptr;
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&tmp), (2 * units + (size_t)kTreeWorkArrays * slab) * ML * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(zerofier tree walk)", __FILE__, __LINE__);
    u64* padded = tmp;
    u64* vals = padded + units * ML;
    u64* work = vals + units * ML;
    int rc = pad_copy(coeffs, padded, (long long)(n_coeffs * L), (long long)(chunks * ML), (long long)batch, s, (long long)poly_stride);
    for (size_t u0 = 0; u0 < units && !rc; u0 += slab) {
        const size_t nu = std::min(slab, units - u0);
        rc = zerofier_tree_evaluate<L>(ctx, T, padded + u0 * ML, points, (long long)n_points, vals + u0 * ML, work, s, (long long)nu);
    }
    int log_m = 0;
    while ((1ll << log_m) < M) ++log_m;
    for (size_t b0 = 0; b0 < batch && !rc; b0 += 65535) {  // grid.y = polynomial
        const unsigned nb = (unsigned)std::min<size_t>(65535, batch - b0);
        hipLaunchKernelGGL(tfk::chunk_combine_kernel<L>, dim3((unsigned)((n_points + 255) / 256), nb), dim3(256), 0, s, (const u64*)(vals + b0 * chunks * ML),
                           M, (int)chunks, points, (long long)n_points, log_m, out + b0 * n_points * L);
        if (hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
    }
    hipError_t e2 = hipFreeAsync(tmp, s);
    if (rc) return rc;
    if (e2 != hipSuccess) return hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}

template <int L>
int batch_evaluate_tree_t(const u64* coeffs, size_t n_coeffs, size_t poly_stride, size_t batch, const u64* points, size_t n_points,
                          u64* out, hipStream_t s) {
    const int kTreeLeaf = tree_leaf(L);
    ZerofierTree T;
    long long M = kTreeLeaf;
    int h = 0;
    while (M < (long long)n_points) M <<= 1, ++h;
    T.leaf = kTreeLeaf;
    T.M = M;
    T.h = h;
    // the tree (6 h M) and the build's work space (8 M); the walks bring their own
    const size_t words = (size_t)(kTreeLevelArrays * h + kTreeWorkArrays) * (size_t)M * L;
    u64* arena = nullptr;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&arena), words * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(zerofier tree)", __FILE__, __LINE__);
    rc = zerofier_tree_build<L>(ctx, points, (long long)n_points, &T, arena, arena + (size_t)(kTreeLevelArrays * h) * M * L, s);
    if (!rc) rc = tree_batch_evaluate<L>(ctx, T, points, n_points, coeffs, n_coeffs, poly_stride, batch, out, s);
    hipError_t e2 = hipFreeAsync(arena, s);
    if (rc) return rc;
    if (e2 != hipSuccess) return hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}

// ------------------------------------------------------------------------------------ batch evaluation
// Polynomial::batch_evaluate / iterative_batch_evaluate (polynomial.rs:1840-1878): f at arbitrary points of the same
// field.  Exact arithmetic makes every evaluation scheme return the reference's values, so the device uses Horner:
// lane per point for short polynomials, workgroup per (point, polynomial) with a 256-way split of the coefficients
// otherwise.  `batch` polynomials of n_coeffs coefficients (poly_stride words apart) share the points;
// out[(b * n_points + i) * L ..] = f_b(points[i]).
int batch_evaluate_horner(const u64* coeffs, size_t n_coeffs, size_t poly_stride, size_t batch, const u64* points, size_t n_points, u64* out,
                          int L, void* stream, int CL = 0);
int batch_evaluate_dev(const u64* coeffs, size_t n_coeffs, size_t poly_stride, size_t batch, const u64* points, size_t n_points,
                       u64* out, int L, void* stream) {
    if (n_points == 0 || batch == 0) return TF_OK;
    if (!points || !out || (n_coeffs && !coeffs)) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (tree_route(n_coeffs, n_points, batch, L)) {  // many points on a long polynomial: the zerofier tree (same values)
        return L == 1 ? batch_evaluate_tree_t<1>(coeffs, n_coeffs, poly_stride, batch, points, n_points, out, s)
                      : batch_evaluate_tree_t<3>(coeffs, n_coeffs, poly_stride, batch, points, n_points, out, s);
    }
    return batch_evaluate_horner(coeffs, n_coeffs, poly_stride, batch, points, n_points, out, L, stream);
}

// CL: words per coefficient (0 = L; 1 with L = 3: base-field coefficients at extension-field points)
int batch_evaluate_horner(const u64* coeffs, size_t n_coeffs, size_t poly_stride, size_t batch, const u64* points, size_t n_points, u64* out,
                          int L, void* stream, int CL) {
    if (n_points == 0 || batch == 0) return TF_OK;
    const bool mixed = L == 3 && CL == 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool split = n_coeffs >= 1024;
    // grid.y is limited to 65535 and a launch to 2^32 - 1 threads: walk the batch and the points in slabs
    const size_t point_slab = size_t(1) << 22;
    for (size_t b0 = 0; b0 < batch; b0 += 65535) {
        const unsigned nb = (unsigned)std::min<size_t>(65535, batch - b0);
        for (size_t p0 = 0; p0 < n_points; p0 += point_slab) {
            const size_t np = std::min(point_slab, n_points - p0);
            const u64* c = coeffs + b0 * poly_stride;
            const u64* pts = points + p0 * size_t(L);
            u64* o = out + (b0 * n_points + p0) * size_t(L);
            // the kernels index the output of polynomial b at o + b * out_stride: the full point count, not the slab's
            const dim3 grid = split ? dim3((unsigned)np, nb) : dim3((unsigned)((np + 255) / 256), nb);
            if (mixed && split)
                hipLaunchKernelGGL((tfk::batch_evaluate_split_kernel<3, 1>), grid, dim3(256), 0, s, c, (long long)n_coeffs, (long long)poly_stride, pts,
                                   (long long)np, o, (long long)n_points);
            else if (mixed)
                hipLaunchKernelGGL((tfk::batch_evaluate_kernel<3, 1>), grid, dim3(256), 0, s, c, (long long)n_coeffs, (long long)poly_stride, pts,
                                   (long long)np, o, (long long)n_points);
            else if (split && L == 1)
                hipLaunchKernelGGL(tfk::batch_evaluate_split_kernel<1>, grid, dim3(256), 0, s, c, (long long)n_coeffs, (long long)poly_stride, pts,
                                   (long long)np, o, (long long)n_points);
            else if (split)
                hipLaunchKernelGGL(tfk::batch_evaluate_split_kernel<3>, grid, dim3(256), 0, s, c, (long long)n_coeffs, (long long)poly_stride, pts,
                                   (long long)np, o, (long long)n_points);
            else if (L == 1)
                hipLaunchKernelGGL(tfk::batch_evaluate_kernel<1>, grid, dim3(256), 0, s, c, (long long)n_coeffs, (long long)poly_stride, pts,
                                   (long long)np, o, (long long)n_points);
            else
                hipLaunchKernelGGL(tfk::batch_evaluate_kernel<3>, grid, dim3(256), 0, s, c, (long long)n_coeffs, (long long)poly_stride, pts,
                                   (long long)np, o, (long long)n_points);
            HIPCHK(hipGetLastError());
        }
    }
    return TF_OK;
}

// ---------------------------------------------------------------- zerofier and interpolation through the zerofier tree
// Polynomial::zerofier / par_zerofier (polynomial.rs:1435-1485) and Polynomial::interpolate / par_interpolate / fast_interpolate /
// batch_fast_interpolate (:1502-1838), poly_kernels.h has the scheme.  The tree of the padded point set is built once; the
// zerofier is its root, the interpolants of `rows` value rows share the tree and the inverse weights 1 / Z'(x_i) (what the
// reference's batch_fast_interpolate memoises in its two dictionaries, :1723-1731).
struct PaddedTree {
    ZerofierTree T;
    size_t n = 0;             // real points
    bool persistent = false;  // hipMalloc'ed (a caller's handle) instead of a stream-ordered temporary
    u64* arena = nullptr;     // tree levels, root tail, leaf scratch, caller's extra
    u64* root_tail = nullptr; // M L words: x^M + root_tail = prod (x - p_i) * x^(M - n)
    u64* extra = nullptr;     // caller's space behind the tree
};

// Builds the padded tree of `points` (levels, root).  The build's work space is a temporary of the build alone.
template <int L>
int padded_tree_build(const u64* points, size_t n_points, size_t extra_words, PaddedTree* pt, hipStream_t s, bool persistent = false) {
    const int kTreeLeaf = tree_interp_leaf(L);
    long long M = kTreeLeaf;
    int h = 0;
    while (M < (long long)n_points) M <<= 1, ++h;
    pt->T.leaf = kTreeLeaf;
    pt->T.M = M;
    pt->T.h = h;
    pt->n = n_points;
    pt->persistent = persistent;
    // tree (6 h M) + root tail (M) + a scratch inverse for a single leaf (M) + caller's
    const size_t words = (size_t)(kTreeLevelArrays * h + 2) * (size_t)M * L + extra_words;
    const size_t work_words = (size_t)kTreeWorkArrays * (size_t)M * L;
    if ((words + work_words) * sizeof(u64) > (size_t(64) << 30)) return TF_ERR_OUT_OF_MEMORY;
    hipError_t e = persistent ? hipMalloc(reinterpret_cast<void**>(&pt->arena), words * sizeof(u64))
                              : pool_malloc_async(reinterpret_cast<void**>(&pt->arena), words * sizeof(u64), s);
    if (e != hipSuccess) {
        pt->arena = nullptr;
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? TF_ERR_OUT_OF_MEMORY : hip_fail(e, "hipMalloc(zerofier tree)", __FILE__, __LINE__);
    }
    pt->root_tail = pt->arena + (size_t)(kTreeLevelArrays * h) * M * L;
    u64* leaf_inv = pt->root_tail + (size_t)M * L;
    pt->extra = leaf_inv + (size_t)M * L;
    if (h == 0) {  // one leaf: it is the root
        if (3 * kTreeLeaf * L * sizeof(u64) > 48 * 1024) {
            static std::atomic<unsigned long long> done_mask{0};
            if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::leaf_zerofier_kernel<L>), 160 * 1024, done_mask)) return rc_attr;
        }
        hipLaunchKernelGGL(tfk::leaf_zerofier_kernel<L>, dim3(1), dim3(kTreeLeaf), 3 * kTreeLeaf * L * sizeof(u64), s, points, (long long)n_points,
                           kTreeLeaf, pt->root_tail, leaf_inv);
        HIPCHK(hipGetLastError());
        return TF_OK;
    }
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    u64* work = nullptr;
    e = pool_malloc_async(reinterpret_cast<void**>(&work), work_words * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(zerofier tree build)", __FILE__, __LINE__);
    rc = zerofier_tree_build<L>(ctx, points, (long long)n_points, &pt->T, pt->arena, work, s);
    // the root from the transforms of the two nodes of level h - 1 (d = M / 2, order M)
    const long long d = M / 2;
    if (!rc) rc = launch_1d<L>(tfk::zerofier_pointwise_kernel<L>, 2 * d, s, (const u64*)pt->T.That[h - 1], pt->root_tail, d, (long long)1);
    if (!rc) rc = run_ntt(ctx, pt->root_tail, pt->root_tail, 2 * d * L, 2 * d * L, (size_t)(2 * d), 1, L, true, nullptr, -1, s);
    hipError_t e2 = hipFreeAsync(work, s);
    if (!rc && e2 != hipSuccess) rc = hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return rc;
}

int padded_tree_free(PaddedTree* pt, hipStream_t s, int rc) {
    hipError_t e = hipSuccess;
    if (pt->arena) e = pt->persistent ? hipFree(pt->arena) : hipFreeAsync(pt->arena, s);
    pt->arena = nullptr;
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(e, "hipFree(zerofier tree)", __FILE__, __LINE__);
    return TF_OK;
}

template <int L>
int zerofier_dev_t(const u64* roots, size_t n_roots, u64* out, hipStream_t s) {
    PaddedTree pt;
    int rc = padded_tree_build<L>(roots, n_roots, 0, &pt, s);
    if (!rc) rc = launch_1d<L>(tfk::zerofier_unpad_kernel<L>, (long long)n_roots + 1, s, (const u64*)pt.root_tail, pt.T.M, (long long)n_roots, out);
    return padded_tree_free(&pt, s, rc);
}

int zerofier_dev(const u64* roots, size_t n_roots, u64* out, int L, void* stream) {
    if (!out || (n_roots && !roots)) return TF_ERR_NULL_POINTER;
    if (n_roots > (size_t(1) << 30)) return TF_ERR_LEN_TOO_LARGE;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return L == 1 ? zerofier_dev_t<1>(roots, n_roots, out, s) : zerofier_dev_t<3>(roots, n_roots, out, s);
}

// winv[i] = 1 / Z'(x_i), i < n (M L words are written: zero beyond n is NOT guaranteed, the consumers stop at n).  Synchronises
// the stream once: a zero Z'(x_i) is a repeated domain point, where the reference panics (TF_ERR_INVERSE_OF_ZERO).
template <int L>
// d_status != null (the *_dev_async entry points): no synchronisation -- a zero weight denominator is reported by writing
// TF_ERR_INVERSE_OF_ZERO to *d_status (device memory, first error wins) and, if sticky != null, by setting *sticky (a flag
// that outlives the call: a ZerofierTree handle whose weights are bad keeps reporting it).
int tree_inverse_weights(DeviceCtx* ctx, const PaddedTree& pt, const u64* domain, u64* winv, hipStream_t s, int* d_status = nullptr,
                         int* sticky = nullptr) {
    const long long M = pt.T.M;
    const size_t ML = (size_t)M * L, n = pt.n;
    u64* tmp = nullptr;  // derivative (M), its values (M), walk work (8 M), flag
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&tmp), ((2 + kTreeWorkArrays) * ML + 2) * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(interpolation weights)", __FILE__, __LINE__);
    u64* deriv = tmp;
    u64* dz = deriv + ML;
    u64* work = dz + ML;
    int* flag = reinterpret_cast<int*>(work + (size_t)kTreeWorkArrays * ML);
    int rc = launch_1d<L>(tfk::zerofier_derivative_kernel<L>, M, s, (const u64*)pt.root_tail, M, (long long)n, deriv);
    if (!rc) rc = zerofier_tree_evaluate<L>(ctx, pt.T, deriv, domain, (long long)n, dz, work, s);
    if (!rc) {
        e = hipMemsetAsync(flag, 0, sizeof(int), s);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__);
    }
    if (!rc) rc = launch_1d<L>(tfk::fe_inverse_kernel<L>, (long long)n, s, (const u64*)dz, (long long)n, winv, flag);
    if (!rc && d_status) {
        hipLaunchKernelGGL(tfk::status_merge_kernel, dim3(1), dim3(1), 0, s, (const int*)flag, d_status, (int)TF_ERR_INVERSE_OF_ZERO, (int)TF_ERR_INVERSE_OF_ZERO);
        if (sticky) hipLaunchKernelGGL(tfk::status_merge_kernel, dim3(1), dim3(1), 0, s, (const int*)flag, sticky, 1, 1);
        if (hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
    } else if (!rc) {
        int host_flag = 0;
        e = hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = hip_fail(e, "interpolate: weight check", __FILE__, __LINE__);
        else if (host_flag) rc = TF_ERR_INVERSE_OF_ZERO;  // Z'(x_i) = 0: a repeated domain point
    }
    hipError_t e2 = hipFreeAsync(tmp, s);
    if (!rc && e2 != hipSuccess) rc = hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return rc;
}

// The walk up: `rows` value rows -> rows x n coefficients, given the tree and the inverse weights.
template <int L>
int tree_interpolate_rows(DeviceCtx* ctx, const PaddedTree& pt, const u64* domain, const u64* winv, const u64* values, size_t rows, u64* out,
                          hipStream_t s) {
    const int kTreeLeaf = pt.T.leaf;
    const long long M = pt.T.M;
    const size_t ML = (size_t)M * L, n = pt.n;
    const int h = pt.T.h;
    // rows go up the tree in slabs: targets, two interpolant levels and the children's transforms (2 M) per row
    const size_t slab = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(rows, 32768), (size_t(1) << 26) / ML));
    u64* tmp = nullptr;
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&tmp), 5 * slab * ML * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(interpolation rows)", __FILE__, __LINE__);
    u64* targets = tmp;
    u64* na = targets + slab * ML;
    u64* nb = na + slab * ML;
    u64* Nh = nb + slab * ML;
    int rc = TF_OK;
    for (size_t r0 = 0; r0 < rows && !rc; r0 += slab) {
        const size_t nr = std::min(slab, rows - r0);
        // few leaves: the quotient form on the leaf zerofiers the tree holds (no barrier per point; four threads per coefficient in
        // its second phase).  TF_TREE_NO_LEAF_SPLIT: A/B switch
        static const bool no_split = ab_env("TF_TREE_NO_LEAF_SPLIT") != nullptr;
        constexpr int kSplit = 4;
        const size_t div_lds = ((size_t)2 * kTreeLeaf + (size_t)kTreeLeaf * (kTreeLeaf + 1) + (size_t)kSplit * kTreeLeaf) * L * sizeof(u64);
        if (!no_split && (long long)nr * M <= (L == 1 ? 2 : 1) * leaf_split_max() && kTreeLeaf * kSplit <= 1024 && kTreeLeaf >= 4 * kSplit && div_lds <= 144 * 1024) {
            static std::atomic<unsigned long long> done_mask{0};
            if (div_lds > 48 * 1024) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::leaf_interpolant_div_kernel<L, kSplit>), 144 * 1024, done_mask);
            if (!rc)
                hipLaunchKernelGGL((tfk::leaf_interpolant_div_kernel<L, kSplit>), dim3((unsigned)(M / kTreeLeaf), (unsigned)nr), dim3(kTreeLeaf * kSplit),
                                   div_lds, s, domain, values + r0 * n * L, winv, (const u64*)(h > 0 ? pt.T.tails[0] : pt.root_tail) /* a single leaf is the root */, (long long)n, kTreeLeaf, M, na);
        } else {
            if (6 * kTreeLeaf * L * sizeof(u64) > 48 * 1024) {  // only with a leaf size forced through TF_TREE_LEAF_LOG
                static std::atomic<unsigned long long> done_mask{0};
                if (!rc) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::leaf_interpolant_kernel<L>), 160 * 1024, done_mask);
            }
            hipLaunchKernelGGL(tfk::leaf_interpolant_kernel<L>, dim3((unsigned)(M / kTreeLeaf), (unsigned)nr), dim3(kTreeLeaf),
                               6 * kTreeLeaf * L * sizeof(u64), s, domain, values + r0 * n * L, winv, (long long)n, kTreeLeaf, M, na);
        }
        if (hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
        u64* cur = na;
        u64* nxt = nb;
        bool wrote_direct = false;
        for (int l = 0; l < h && !rc; ++l) {
            // one level for all rows of the slab: transforms of order 2d of every child's interpolant against the level's cached
            // tail transforms, the combination N_left Z_right + N_right Z_left pointwise, one inverse transform -- which lands
            // in the next level's layout
            const long long d = (long long)kTreeLeaf << l, children = M / d, parents = children / 2;
            if (tree_level_wanted(2 * d, parents * (long long)nr, L, true)) {
                // the whole level in one launch (both children's transforms, the combination and the inverse transform in LDS)
                const bool direct = l == h - 1 && (long long)n == M;
                tfk::TreeLevelArgs a{};
                a.cur = cur, a.nxt = direct ? out + r0 * n * L : nxt, a.that = pt.T.That[l], a.lines = parents * (long long)nr, a.per = parents;
                rc = launch_tree_level<true>(ctx, ilog2((size_t)(2 * d)), a, s, L);
                wrote_direct = direct;
                std::swap(cur, nxt);
                continue;
            }
            rc = run_ntt(ctx, cur, Nh, d * L, 2 * d * L, (size_t)(2 * d), (size_t)(children * (long long)nr), L, false, nullptr, d, s);
            // (measured, tools/tree_latency.py on one box: prepared-tree interpolation of 2^12 points 129.0 us with the pointwise
            //  kernel, 135.8 us with it fused -- four strided loads and two products per element in front of the transform's first
            //  stage cost more than the 1.5 us a pipelined elementwise launch really adds; off unless TF_TREE_FUSE_INTERP is set)
            static const bool fuse_interp = ab_env("TF_TREE_FUSE_INTERP") != nullptr;
            if (!rc && L == 1 && fuse_interp && tree_fuse(2 * d, parents * (long long)nr, 1)) {
                // the pointwise combination rides on the load of the inverse transform (ntt_lat_kernel, load_mode 2)
                tfk::NttLatArgs m{};
                m.load_mode = 2, m.th = pt.T.That[l], m.parents = parents;
                rc = launch_lat(ctx, Nh, nxt, 2 * d, 2 * d, ilog2((size_t)(2 * d)), (size_t)(parents * (long long)nr), 1, true, -1, nullptr, s, &m);
            } else {
                if (!rc) rc = launch_1d<L>(tfk::interpolant_pointwise_kernel<L>, (long long)nr * parents * 2 * d, s, (const u64*)Nh,
                                           (const u64*)pt.T.That[l], nxt, d, parents, (long long)nr);
                // the root level of an unpadded domain (n = M) transforms straight into the caller's rows: no copy-out launch
                const bool direct = l == h - 1 && (long long)n == M;
                u64* dst = direct ? out + r0 * n * L : nxt;
                wrote_direct = direct;
                if (!rc) rc = run_ntt(ctx, nxt, dst, 2 * d * L, 2 * d * L, (size_t)(2 * d), (size_t)(parents * (long long)nr), L, true, nullptr, -1, s);
            }
            std::swap(cur, nxt);
        }
        if (!rc && !wrote_direct) {
            hipLaunchKernelGGL(tfk::interpolant_unpad_kernel<L>, dim3((unsigned)((n + 255) / 256), (unsigned)nr), dim3(256), 0, s,
                               (const u64*)cur, M, (long long)n, out + r0 * n * L);
            if (hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
        }
    }
    hipError_t e2 = hipFreeAsync(tmp, s);
    if (!rc && e2 != hipSuccess) rc = hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return rc;
}

template <int L>
int interpolate_dev_t(const u64* domain, const u64* values, size_t n, size_t rows, u64* out, hipStream_t s, int* d_status) {
    const int kTreeLeaf = tree_interp_leaf(L);
    long long M = kTreeLeaf;
    while (M < (long long)n) M <<= 1;
    PaddedTree pt;
    int rc = padded_tree_build<L>(domain, n, (size_t)M * L, &pt, s);  // extra: the inverse weights
    DeviceCtx* ctx = nullptr;
    if (!rc) rc = current_ctx(&ctx);
    // (The synchronising entry points wait for the repeated-point check between the weights and the walk up.  Reading it back
    //  once, after the whole call had been enqueued, was tried: isolated-call latency 481.8 vs 479.2 us at 2^12 points, 170.5 vs
    //  167.5 at 2^8 -- the extra status launches cost what the removed bubble saved; not kept.)
    if (!rc) rc = tree_inverse_weights<L>(ctx, pt, domain, pt.extra, s, d_status);
    if (!rc) rc = tree_interpolate_rows<L>(ctx, pt, domain, pt.extra, values, rows, out, s);
    return padded_tree_free(&pt, s, rc);
}

// `rows` value rows of n elements over one domain of n distinct points -> rows x n coefficients (low to high).
int interpolate_dev(const u64* domain, const u64* values, size_t n, size_t rows, u64* out, int L, void* stream, int* d_status) {
    if (n == 0) return TF_ERR_EMPTY_DOMAIN;  // "interpolation must happen through more than zero points" (:1503-1506)
    if (rows == 0) return TF_OK;
    if (!domain || !values || !out) return TF_ERR_NULL_POINTER;
    if (n > (size_t(1) << 30)) return TF_ERR_LEN_TOO_LARGE;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return L == 1 ? interpolate_dev_t<1>(domain, values, n, rows, out, s, d_status) : interpolate_dev_t<3>(domain, values, n, rows, out, s, d_status);
}

// ---- a zerofier tree that outlives the call (math/zerofier_tree.rs: ZerofierTree::new_from_domain, used with
// Polynomial::divide_and_conquer_batch_evaluate, polynomial.rs:1882-1894): levels, cached level transforms, root, the domain and
// -- once an interpolation has asked for them -- the inverse weights stay in HBM; every call brings its own work space, so one
// handle serves concurrent calls on different streams.
struct TreeHandle {
    int L = 1;
    int device = 0;
    PaddedTree pt;
    u64* points = nullptr;  // the domain, M L words (pt.extra)
    u64* winv = nullptr;    // 1 / Z'(x_i), M L words (pt.extra + M L)
    int* bad = nullptr;     // device flag behind the weights: set when an asynchronous weight computation met a repeated point
    std::mutex mu;
    bool have_winv = false;
    bool winv_async = false;   // the weights were enqueued by an asynchronous call: their verdict sits in *bad on the device ...
    int winv_verdict = -1;     // ... until a blocking call has read it back once: -1 unknown, 0 fine, 1 repeated domain point
};

template <int L>
int tree_handle_new_t(const u64* d_domain, size_t n, hipStream_t s, TreeHandle* H, bool async) {
    const int kTreeLeaf = tree_interp_leaf(L);
    long long M = kTreeLeaf;
    while (M < (long long)n) M <<= 1;
    int rc = padded_tree_build<L>(d_domain, n, 2 * (size_t)M * L + 1, &H->pt, s, true);
    if (rc) return rc;
    H->points = H->pt.extra;
    H->winv = H->points + (size_t)M * L;
    H->bad = reinterpret_cast<int*>(H->winv + (size_t)M * L);
    if (hipMemsetAsync(H->bad, 0, sizeof(u64), s) != hipSuccess) return TF_ERR_HIP;
    // the tree was built from the caller's array; the handle keeps its own copy for the leaf evaluations
    hipError_t e = hipMemsetAsync(H->points, 0, (size_t)M * L * sizeof(u64), s);
    if (e == hipSuccess && n) e = hipMemcpyAsync(H->points, d_domain, n * L * sizeof(u64), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && !async) e = hipStreamSynchronize(s);  // the handle may be used from any stream afterwards
    if (e != hipSuccess) return hip_fail(e, "zerofier tree: domain copy", __FILE__, __LINE__);
    return TF_OK;
}

int tree_handle_new(const u64* d_domain, size_t n, int L, void* stream, TreeHandle** out, bool async) {
    if (!out) return TF_ERR_NULL_POINTER;
    *out = nullptr;
    if (n && !d_domain) return TF_ERR_NULL_POINTER;
    if (n > (size_t(1) << 30)) return TF_ERR_LEN_TOO_LARGE;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    std::unique_ptr<TreeHandle> H(new TreeHandle());
    H->L = L;
    if (hipGetDevice(&H->device) != hipSuccess) return TF_ERR_NO_DEVICE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = L == 1 ? tree_handle_new_t<1>(d_domain, n, s, H.get(), async) : tree_handle_new_t<3>(d_domain, n, s, H.get(), async);
    if (rc) {
        (void)hipStreamSynchronize(s);
        (void)padded_tree_free(&H->pt, s, rc);
        return rc;
    }
    *out = H.release();
    return TF_OK;
}

int tree_handle_check(const TreeHandle* H, DeviceCtx** ctx) {
    if (!H) return TF_ERR_NULL_POINTER;
    int rc = current_ctx(ctx);
    if (rc) return rc;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != H->device) {
        t_last_error = "zerofier tree used on a device other than the one it was built on";
        return TF_ERR_HIP;
    }
    return TF_OK;
}

int tree_handle_zerofier(const TreeHandle* H, u64* d_out, void* stream) {
    DeviceCtx* ctx = nullptr;
    int rc = tree_handle_check(H, &ctx);
    if (rc) return rc;
    if (!d_out) return TF_ERR_NULL_POINTER;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n = (long long)H->pt.n;
    return H->L == 1 ? launch_1d<1>(tfk::zerofier_unpad_kernel<1>, n + 1, s, (const u64*)H->pt.root_tail, H->pt.T.M, n, d_out)
                     : launch_1d<3>(tfk::zerofier_unpad_kernel<3>, n + 1, s, (const u64*)H->pt.root_tail, H->pt.T.M, n, d_out);
}

// out[(b * n + i) * L] = f_b(domain[i]); `batch` polynomials of n_coeffs coefficients, packed
int tree_handle_batch_evaluate(const TreeHandle* H, const u64* d_coeffs, size_t n_coeffs, size_t batch, u64* d_out, void* stream) {
    DeviceCtx* ctx = nullptr;
    int rc = tree_handle_check(H, &ctx);
    if (rc) return rc;
    const size_t n = H->pt.n;
    if (n == 0 || batch == 0) return TF_OK;
    if (!d_out || (n_coeffs && !d_coeffs)) return TF_ERR_NULL_POINTER;
    const int L = H->L;
    if (H->pt.T.h == 0 || n_coeffs < 2)  // a single leaf (or a constant): Horner on the handle's copy of the domain
        return batch_evaluate_horner(d_coeffs, n_coeffs, n_coeffs * L, batch, H->points, n, d_out, L, stream);
    {
        // the guards tree_route applies to the one-shot call: the walk indexes its arrays per unit (<= 65 536) and brings
        // 2 units M words of padded coefficients and values plus the slab's work space -- a batch beyond that takes Horner
        const size_t M = (size_t)H->pt.T.M, units = batch * ((n_coeffs + M - 1) / M);
        const size_t slab = std::max<size_t>(1, std::min<size_t>(units, (size_t(1) << 25) / M));
        const size_t words = (2 * units + (size_t)kTreeWorkArrays * slab) * M * (size_t)L;
        if (units > 65536 || words * sizeof(u64) > (size_t(64) << 30))
            return batch_evaluate_horner(d_coeffs, n_coeffs, n_coeffs * L, batch, H->points, n, d_out, L, stream);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    return L == 1 ? tree_batch_evaluate<1>(ctx, H->pt.T, H->points, n, d_coeffs, n_coeffs, n_coeffs, batch, d_out, s)
                  : tree_batch_evaluate<3>(ctx, H->pt.T, H->points, n, d_coeffs, n_coeffs, 3 * n_coeffs, batch, d_out, s);
}

int tree_handle_interpolate(TreeHandle* H, const u64* d_values, size_t rows, u64* d_out, void* stream, int* d_status) {
    DeviceCtx* ctx = nullptr;
    int rc = tree_handle_check(H, &ctx);
    if (rc) return rc;
    if (H->pt.n == 0) return TF_ERR_EMPTY_DOMAIN;
    if (rows == 0) return TF_OK;
    if (!d_values || !d_out) return TF_ERR_NULL_POINTER;
    hipStream_t s = static_cast<hipStream_t>(stream);
    {
        std::lock_guard<std::mutex> lk(H->mu);  // the first interpolation computes the weights (and synchronises its stream)
        if (!H->have_winv) {
            rc = H->L == 1 ? tree_inverse_weights<1>(ctx, H->pt, H->points, H->winv, s, d_status, d_status ? H->bad : nullptr)
                           : tree_inverse_weights<3>(ctx, H->pt, H->points, H->winv, s, d_status, d_status ? H->bad : nullptr);
            if (rc) return rc;
            H->have_winv = true;
            H->winv_async = d_status != nullptr;
        } else if (d_status) {  // weights computed by an earlier asynchronous call: pass its verdict on
            hipLaunchKernelGGL(tfk::status_merge_kernel, dim3(1), dim3(1), 0, s, (const int*)H->bad, d_status, (int)TF_ERR_INVERSE_OF_ZERO, (int)TF_ERR_INVERSE_OF_ZERO);
            if (hipGetLastError() != hipSuccess) return TF_ERR_HIP;
        } else if (H->winv_async) {
            // a BLOCKING call on a handle whose weights an asynchronous call enqueued: that call could only leave the verdict
            // (a repeated domain point, traits.rs:106) in device memory.  Read it back once -- after everything in flight on the
            // device, whichever stream the weights were enqueued on -- and answer as the blocking first call would have.
            if (H->winv_verdict < 0) {
                int bad = 0;
                hipError_t e = hipDeviceSynchronize();
                if (e == hipSuccess) e = hipMemcpy(&bad, H->bad, sizeof(int), hipMemcpyDeviceToHost);
                if (e != hipSuccess) return hip_fail(e, "zerofier tree: reading the weights' verdict", __FILE__, __LINE__);
                H->winv_verdict = bad ? 1 : 0;
            }
            if (H->winv_verdict) return TF_ERR_INVERSE_OF_ZERO;
        }
    }
    return H->L == 1 ? tree_interpolate_rows<1>(ctx, H->pt, H->points, H->winv, d_values, rows, d_out, s)
                     : tree_interpolate_rows<3>(ctx, H->pt, H->points, H->winv, d_values, rows, d_out, s);
}

size_t tree_handle_num_points(const TreeHandle* H) { return H->pt.n; }
int tree_handle_width(const TreeHandle* H) { return H->L; }
void tree_handle_free(TreeHandle* H) {
    if (!H) return;
    int prev = -1;
    const bool switched = hipGetDevice(&prev) == hipSuccess && prev != H->device && hipSetDevice(H->device) == hipSuccess;
    (void)hipDeviceSynchronize();  // calls still in flight on any stream read the tree
    (void)padded_tree_free(&H->pt, nullptr, TF_OK);
    if (switched) (void)hipSetDevice(prev);
    delete H;
}

// fast_coset_evaluate / fast_coset_interpolate with an XFieldElement OFFSET (polynomial.rs:1374-1399, :1907-1918 with
// S = XFieldElement; the docs recommend a BFieldElement offset, :1366-1368, and the fused pre/post-scale tables of the main path
// are base-field): the scaling is its own pass here -- c_i * offset^i by square-and-multiply per coefficient -- around the plain
// XFE transform.
static bool xfe_inverse_host(const u64 (&a)[3], u64 (&r)[3]) {  // the cofactor formula of poly_kernels.h on the host
    const u64 sm = gl::add(a[0], a[2]), dd = gl::sub(a[1], a[2]);
    const u64 c0 = gl::sub(gl::mont_mul(sm, sm), gl::mont_mul(dd, a[1]));
    const u64 c1 = gl::sub(gl::mont_mul(dd, a[2]), gl::mont_mul(a[1], sm));
    const u64 c2 = gl::sub(gl::mont_mul(a[1], a[1]), gl::mont_mul(sm, a[2]));
    const u64 det = gl::sub(gl::sub(gl::mont_mul(a[0], c0), gl::mont_mul(a[2], c1)), gl::mont_mul(a[1], c2));
    const u64 di = gl::mont_inverse(det);
    r[0] = gl::mont_mul(c0, di);
    r[1] = gl::mont_mul(c1, di);
    r[2] = gl::mont_mul(c2, di);
    return det != 0;
}

int coset_eval_xoffset_dev(const u64* d_coeffs, size_t n_coeffs, const u64 offset[3], u64* d_out, size_t order, size_t batch, void* stream) {
    if (n_coeffs > order) return TF_ERR_ORDER_NOT_ABOVE_DEGREE;  // polynomial.rs:1388-1392
    int rc = check_len(order);
    if (rc) return rc;
    if (order == 0 || batch == 0) return TF_OK;
    if (!d_out || !offset || (n_coeffs && !d_coeffs)) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a launch takes at most 2^32 - 1 threads: walk the batch in slabs of at most 2^30 elements
    const size_t slab = std::max<size_t>(1, (size_t(1) << 30) / order);
    for (size_t b0 = 0; b0 < batch; b0 += slab) {
        const long long nb = (long long)std::min(slab, batch - b0), total = (long long)order * nb;
        hipLaunchKernelGGL(tfk::xfe_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_coeffs + b0 * n_coeffs * 3, (long long)n_coeffs,
                           (long long)n_coeffs * 3, d_out + b0 * order * 3, (long long)order, nb, offset[0], offset[1], offset[2]);
        HIPCHK(hipGetLastError());
    }
    return run_ntt(ctx, d_out, d_out, (long long)order * 3, (long long)order * 3, order, batch, 3, false, nullptr, -1, s);
}

int coset_interp_xoffset_dev(const u64* d_values, size_t n, const u64 offset[3], u64* d_out, size_t batch, void* stream) {
    int rc = check_len(n);
    if (rc) return rc;
    if (n == 0 || batch == 0) return TF_OK;
    if (!d_values || !d_out || !offset) return TF_ERR_NULL_POINTER;
    const u64 off[3] = {offset[0], offset[1], offset[2]};
    u64 inv[3];
    if (!xfe_inverse_host(off, inv)) return TF_ERR_INVERSE_OF_ZERO;  // offset.inverse() panics on zero (x_field_element.rs:371-375)
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = run_ntt(ctx, d_values, d_out, (long long)n * 3, (long long)n * 3, n, batch, 3, true, nullptr, -1, s);
    if (rc) return rc;
    const size_t slab = std::max<size_t>(1, (size_t(1) << 30) / n);
    for (size_t b0 = 0; b0 < batch; b0 += slab) {
        const long long nb = (long long)std::min(slab, batch - b0), total = (long long)n * nb;
        u64* o = d_out + b0 * n * 3;
        hipLaunchKernelGGL(tfk::xfe_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const u64*)o, (long long)n, (long long)n * 3, o,
                           (long long)n, nb, inv[0], inv[1], inv[2]);
        HIPCHK(hipGetLastError());
    }
    return TF_OK;
}

// barycentric_evaluate (polynomial.rs:2609-2637) for `batch` codewords of length n (a power of two) at ONE indeterminate
// (3 raw words; a BFieldElement as [x, 0, 0]): out[b] = interpolant_b(x) as an XFieldElement.  cw_width 1 / 3 = the codewords'
// field.  Where the reference panics: n not a power of two (primitive_root_of_unity(..).unwrap()) -> TF_ERR_LEN_NOT_POWER_OF_TWO;
// x inside the subgroup, or n = 0 (batch_inversion / inverse of zero) -> TF_ERR_INVERSE_OF_ZERO.
int barycentric_dev(const u64* codewords, size_t n, size_t batch, int cw_width, const u64 x[3], u64* out, void* stream) {
    int rc = check_len(n);
    if (rc) return rc;
    if (!x) return TF_ERR_NULL_POINTER;
    if (n == 0) return TF_ERR_INVERSE_OF_ZERO;  // the empty sums: denominator.inverse() of zero
    if (x[1] == 0 && x[2] == 0 && gl::mont_pow(x[0], (u64)n) == gl::ONE) return TF_ERR_INVERSE_OF_ZERO;  // x = w^i for some i
    if (batch == 0) return TF_OK;
    if (!codewords || !out) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long per_chunk = (long long)tfk::kBaryPerThread * 256, n_chunks = ((long long)n + per_chunk - 1) / per_chunk;
    u64* tmp = nullptr;  // weights (3 n) + partial sums ((batch + 1) n_chunks 3)
    const size_t words = 3 * n + 3 * (batch + 1) * (size_t)n_chunks;
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&tmp), words * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(barycentric)", __FILE__, __LINE__);
    u64* w = tmp;
    u64* partial = tmp + 3 * n;
    const int log_n = ilog2(n);
    const u64 omega = root_of_unity_mont(log_n);
    hipLaunchKernelGGL(tfk::barycentric_weights_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, (long long)n, log_n, omega, gl::mont_pow(omega, 256),
                       x[0], x[1], x[2], w);
    {
        static const int rows_env = [] { const char* e = ab_env("TF_BARY_ROWS"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= tfk::kBaryRows) ? v : 0; }();
        const int rpb = rows_env ? rows_env : 2;  // rows per block (A/B: tools/barycentric_bench.py; 1 / 2 / 4 within 3 % of each other)
        const long long row_groups = ((long long)batch + 1 + rpb - 1) / rpb;  // the denominator is row `batch`
        if (row_groups > 65535) rc = TF_ERR_LEN_TOO_LARGE;  // more than 65 534 codewords in one call
        else if (cw_width == 1)
            hipLaunchKernelGGL(tfk::barycentric_partial_kernel<1>, dim3((unsigned)n_chunks, (unsigned)row_groups), dim3(256), 0, s, codewords,
                               (const u64*)w, (long long)n, (long long)batch, partial, rpb);
        else
            hipLaunchKernelGGL(tfk::barycentric_partial_kernel<3>, dim3((unsigned)n_chunks, (unsigned)row_groups), dim3(256), 0, s, codewords,
                               (const u64*)w, (long long)n, (long long)batch, partial, rpb);
        if (!rc && hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
    }
    if (!rc) {
        hipLaunchKernelGGL(tfk::barycentric_finish_kernel, dim3((unsigned)batch), dim3(64), 0, s, (const u64*)partial, n_chunks, (long long)batch, out);
        if (hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
    }
    hipError_t e2 = hipFreeAsync(tmp, s);
    if (rc) return rc;
    if (e2 != hipSuccess) return hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}

// Polynomial::<BFieldElement>::clean_divide (polynomial.rs:2358-2411): a / b for b | a, by pointwise division on the coset
// X * <w_order> of the extension field (poly_kernels.h).  a, b: normalised coefficient arrays (non-zero leading coefficient),
// out: na - nb + 1 coefficients.  The reference's factor-x workaround (:2368-2378) changes nothing on this coset (X w^i != 0) and
// is not needed; the naive route it takes for divisors below degree 512 (:2360-2364) returns the same quotient.
// `batch` dividends of na coefficients each (packed) by ONE divisor: the divisor's transform is inverted once and shared -- the
// shape of a prover's quotients (many numerators over the same zerofier).
// d_status != null (the *_dev_async entry points): never synchronises; the two panic cases are written to *d_status on the device.
int clean_divide_dev(const u64* a, size_t na, const u64* b, size_t nb, u64* out, void* stream, size_t batch, int* d_status) {
    if (nb == 0) return TF_ERR_DIVISION_BY_ZERO;                      // naive_divide :556-559 "divisor should be non-zero"
    if (na < nb) return na ? TF_ERR_DIVISION_NOT_CLEAN : TF_OK;       // a non-zero dividend of lower degree: the remainder is the dividend
    if (batch == 0) return TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    if (batch > 65535) return TF_ERR_LEN_TOO_LARGE;
    size_t order = 1;
    while (order < na) order <<= 1;                                   // (dividend.degree() + 1).next_power_of_two() :2388-2389
    int rc = check_len(order);
    if (rc) return rc;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    u64* tmp = nullptr;  // rows 0 .. batch-1: the dividends, row `batch`: the divisor; order XFieldElements each
    const size_t half = order * 3;
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&tmp), ((batch + 1) * half + 2) * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(clean_divide)", __FILE__, __LINE__);
    u64* div = tmp + batch * half;
    int* flag = reinterpret_cast<int*>(tmp + (batch + 1) * half);
    // The division coset is X * <w_order> with X = x, the reference's choice (:2383).  A divisor with a root ON that coset (e.g.
    // x^3 - x + 1 itself, which the reference only meets on its naive route below degree 512) makes the pointwise division
    // impossible there: the blocking call then repeats the division once on the coset (x + 1) * <w_order> -- a clean quotient is
    // the same polynomial on any coset -- before it reports TF_ERR_INVERSE_OF_ZERO.
    for (int attempt = 0; attempt < 2; ++attempt) {
    u64 X[3] = {0, gl::ONE, 0};                                       // XFieldElement::from([0, 1, 0]) :2383
    u64 Xinv[3] = {gl::ONE, 0, gl::neg(gl::ONE)};                     // x (x^2 - 1) = -1  ->  x^-1 = 1 - x^2
    if (attempt == 1) {
        X[0] = gl::ONE;                                               // x + 1
        if (!xfe_inverse_host(X, Xinv)) { rc = TF_ERR_INVERSE_OF_ZERO; break; }
        rc = TF_OK;
    }
    const unsigned blocks = (unsigned)((order + 255) / 256);
    e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__);
    if (!rc) {
        hipLaunchKernelGGL(tfk::lift_scale_kernel, dim3(blocks, (unsigned)batch), dim3(256), 0, s, a, (long long)na, (long long)order, tmp, X[0], X[1], X[2]);
        hipLaunchKernelGGL(tfk::lift_scale_kernel, dim3(blocks, 1), dim3(256), 0, s, b, (long long)nb, (long long)order, div, X[0], X[1], X[2]);
        if (hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
    }
    if (!rc) rc = run_ntt(ctx, tmp, tmp, (long long)half, (long long)half, order, batch + 1, 3, false, nullptr, -1, s);
    if (!rc) {
        hipLaunchKernelGGL(tfk::xfe_invert_inplace_kernel, dim3(blocks), dim3(256), 0, s, div, (long long)order, flag);
        if (hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
    }
    if (!rc) rc = launch_1d<3>(tfk::product_bcast_kernel<3>, (long long)(batch * order), s, (const u64*)tmp, (const u64*)div, tmp, (long long)order,
                               (long long)(batch * order));
    if (!rc) rc = run_ntt(ctx, tmp, tmp, (long long)half, (long long)half, order, batch, 3, true, nullptr, -1, s);
    if (!rc) {
        hipLaunchKernelGGL(tfk::unscale_unlift_kernel, dim3(blocks, (unsigned)batch), dim3(256), 0, s, (const u64*)tmp, (long long)order,
                           (long long)(na - nb + 1), out, Xinv[0], Xinv[1], Xinv[2], flag);
        if (hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
    }
    if (!rc && d_status) {
        hipLaunchKernelGGL(tfk::status_merge_kernel, dim3(1), dim3(1), 0, s, (const int*)flag, d_status, (int)TF_ERR_INVERSE_OF_ZERO, (int)TF_ERR_DIVISION_NOT_CLEAN);
        if (hipGetLastError() != hipSuccess) rc = TF_ERR_HIP;
    } else if (!rc) {
        int host_flag = 0;
        e = hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = hip_fail(e, "clean_divide: flag", __FILE__, __LINE__);
        else if (host_flag & 1) rc = TF_ERR_INVERSE_OF_ZERO;     // a zero of the divisor on the coset: batch_inversion panics
        else if (host_flag & 2) rc = TF_ERR_DIVISION_NOT_CLEAN;  // unlift().unwrap() :2410
    }
    if (rc != TF_ERR_INVERSE_OF_ZERO || d_status) break;         // (the asynchronous variant cannot look at the flag: one coset)
    }
    hipError_t e2 = hipFreeAsync(tmp, s);
    if (rc) return rc;
    if (e2 != hipSuccess) return hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}

// Polynomial::{coset_extrapolate, batch_coset_extrapolate} (polynomial.rs:2117-2331): the values, at `points`, of the
// degree-< n interpolants of `batch` codewords given on the coset {offset * w_n^i}.  Both of the reference's routes
// (naive :2145-2156, fast :2158-2170) compute exactly interpolant(point), which is what this does:
// coset-interpolate on the device, then the batched evaluation above; the coefficients never leave HBM.
int coset_extrapolate_dev(u64 offset_raw, const u64* codewords, size_t n, size_t batch, const u64* points, size_t n_points, u64* out,
                          int L, void* stream) {
    if (n == 0) return TF_ERR_LEN_NOT_POWER_OF_TWO;  // "Panics if the codeword_length is not a power of two" (:2194)
    int rc = check_len(n);
    if (rc) return rc;
    if (batch == 0 || n_points == 0) return TF_OK;
    if (!codewords || !points || !out) return TF_ERR_NULL_POINTER;
    if (offset_raw == 0) return TF_ERR_INVERSE_OF_ZERO;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    u64* coeffs = nullptr;
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&coeffs), batch * n * size_t(L) * sizeof(u64), s);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(coset_extrapolate)", __FILE__, __LINE__);
    rc = coset_interp_dev(codewords, n, offset_raw, coeffs, batch, L, s);
    if (!rc) rc = batch_evaluate_dev(coeffs, n, n * size_t(L), batch, points, n_points, out, L, s);
    hipError_t e2 = hipFreeAsync(coeffs, s);
    if (rc) return rc;
    if (e2 != hipSuccess) return hip_fail(e2, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}

// ------------------------------------------------------------------------------------ authentication structures
// MerkleTree::authentication_structure_node_indices (merkle_tree.rs:449-504): needed minus computable, descending.
int auth_structure_indices(size_t num_leafs, const uint64_t* leaf_indices, size_t k, std::vector<unsigned long long>* out) {
    if (num_leafs == 0 || (num_leafs & (num_leafs - 1))) return TF_ERR_INCORRECT_NUMBER_OF_LEAFS;  // :468-470
    std::set<unsigned long long> needed, computable;
    for (size_t i = 0; i < k; ++i) {
        if (leaf_indices[i] >= num_leafs) return TF_ERR_LEAF_INDEX_INVALID;  // :486-488
        unsigned long long node = leaf_indices[i] + num_leafs;
        while (node > 1) {
            computable.insert(node);
            needed.insert(node ^ 1ull);
            node /= 2;
        }
    }
    out->clear();
    for (auto it = needed.rbegin(); it != needed.rend(); ++it)
        if (!computable.count(*it)) out->push_back(*it);
    return TF_OK;
}


}  // namespace tfi
