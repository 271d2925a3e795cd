// tf_lat.hip -- launchers of the latency-shaped kernels of libtf_hip.so (lat_kernels.h): ntt_lat_kernel / ntt_lat2_kernel for
// calls with little work, and the one-launch-per-level kernels of the zerofier-tree walks and build.  The planner (tf_ntt.hip:
// run_ntt) and the tree orchestration (tf_poly.hip) decide WHEN; the thresholds they ask for live here (lat_wanted, lat2_wanted,
// tree_level_wanted, tree_build_level_wanted) next to the measurements they come from.
#include "tf_internal.h"
#include "lat_kernels.h"

namespace tfi {

// ---- the latency-shaped transform (ntt_lat_kernel): calls with little work, 64 <= n <= 4096
// t[e] = w_n^(+-e), e < n, then n^-1 t[e]
int get_lat_table(DeviceCtx* ctx, int log_n, bool inverse, const u64** out, int scale_log = -1) {
    if (scale_log < 0) scale_log = log_n;  // second half: 2^-scale_log w^e (the n^-1 of the whole transform rides on the last stage)
    const u64 key = make_key(TAG_LAT, log_n, inverse, scale_log, 0);
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = it->second;
        return TF_OK;
    }
    const size_t n = size_t(1) << log_n;
    u64 w = root_of_unity_mont(log_n);
    if (inverse) w = gl::mont_inverse(w);
    const u64 ninv = gl::mont_inverse(gl::to_mont(u64(1) << scale_log));
    std::vector<u64> t(2 * n);
    u64 acc = gl::ONE;
    for (size_t e = 0; e < n; ++e) {
        t[e] = acc;
        t[n + e] = gl::mont_mul(acc, ninv);
        acc = gl::mont_mul(acc, w);
    }
    u64* d = nullptr;
    int rc = upload_table(t, &d);
    if (rc) return rc;
    ctx->tables[key] = d;
    *out = d;
    return TF_OK;
}

template <int LOGN, bool INV>
int launch_lat_t(const tfk::NttLatArgs& a, hipStream_t stream) {
    constexpr int N = 1 << LOGN, WG = LOGN == 12 ? 512 : 256, T = WG / (N / 8);
    constexpr size_t lds = size_t(2) * (tfk::lat_pad(N * T) + 8) * sizeof(u64);
    if constexpr (lds > 48 * 1024) {
        static std::atomic<unsigned long long> done_mask{0};
        if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_lat_kernel<LOGN, INV>), (int)((int)lds), done_mask)) return rc_attr;
    }
    const long long blocks = (a.total + T - 1) / T;
    hipLaunchKernelGGL((tfk::ntt_lat_kernel<LOGN, INV>), dim3((unsigned)blocks), dim3(WG), lds, stream, a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}
template <bool INV>
int launch_lat_dir(int log_n, const tfk::NttLatArgs& a, hipStream_t s) {
    switch (log_n) {
        case 6: return launch_lat_t<6, INV>(a, s);
        case 7: return launch_lat_t<7, INV>(a, s);
        case 8: return launch_lat_t<8, INV>(a, s);
        case 9: return launch_lat_t<9, INV>(a, s);
        case 10: return launch_lat_t<10, INV>(a, s);
        case 11: return launch_lat_t<11, INV>(a, s);
        case 12: return launch_lat_t<12, INV>(a, s);
    }
    return TF_ERR_HIP;
}
// When: the call holds too little work to fill the chip with 32-element threads (measured crossover, tools/lat_sweep.py).
std::atomic<int> g_lat_mode{-1};  // tf_set_ntt_latency_kernel: -1 automatic (TF_NTT_NO_LAT disables), 0 never, 1 whenever the shape allows
bool lat_wanted(int log_n, size_t batch, int L) {
    static const bool off = ab_env("TF_NTT_NO_LAT") != nullptr;  // A/B switch
    // measured crossover against the pass / block kernels (tools/lat_sweep.py, profiles/r03_lat_sweep_*.txt): 2.0 - 2.9 x faster up
    // to 2^20 words per call, level at 2^22 words (BFieldElement) / 1.5 x 2^20 words (XFieldElement: its loads step 24 bytes)
    static const long long env_limit = [] {
        const char* e = ab_env("TF_NTT_LAT_MAX_WORDS");
        return e ? atoll(e) : 0ll;
    }();
    const long long limit = env_limit ? env_limit : (L == 1 ? (1ll << 22) : (3ll << 19));
    const int mode = g_lat_mode.load(std::memory_order_relaxed);
    if (mode == 0 || (mode < 0 && off)) return false;
    if (log_n < 6 || log_n > 12) return false;
    if (mode == 1) return true;
    return (long long)(batch * size_t(L)) << log_n <= limit;
}
// mods (tree walks only): the load / store modifier fields of NttLatArgs; such calls are one launch (batch < 2^22)
int launch_lat(DeviceCtx* ctx, const u64* in, u64* out, long long in_bs, long long out_bs, int log_n, size_t batch, int L, bool inverse,
               long long n_coeffs, const u64* in2, hipStream_t stream, const tfk::NttLatArgs* mods, const u64* pre_scale, const u64* post_scale) {
    const u64* tw = nullptr;
    int rc = get_lat_table(ctx, log_n, inverse, &tw);
    if (rc) return rc;
    const size_t max_batch = size_t(1) << 22;  // 2^31 threads per launch at most
    if (mods && batch > max_batch) return TF_ERR_HIP;
    for (size_t b0 = 0; b0 < batch && !rc; b0 += max_batch) {
        const size_t nb = std::min(max_batch, batch - b0);
        tfk::NttLatArgs a{};
        if (mods) a = *mods;
        a.in = in + (long long)b0 * in_bs;
        a.out = out + (long long)b0 * out_bs;
        a.in2 = in2 ? in2 + (long long)b0 * in_bs : nullptr;
        a.pre_scale = pre_scale;
        a.post_scale = post_scale;
        a.tw = tw;
        a.n_coeffs = n_coeffs;
        a.in_bs = in_bs;
        a.out_bs = out_bs;
        a.total = (long long)nb * L;
        a.ninv = inverse ? gl::mont_inverse(gl::to_mont(u64(1) << log_n)) : 0;
        a.L = L;
        rc = inverse ? launch_lat_dir<true>(log_n, a, stream) : launch_lat_dir<false>(log_n, a, stream);
    }
    return rc;
}

std::atomic<int> g_min_passes{0};  // tf_set_ntt_min_passes

// ---- one launch per LEVEL of a small zerofier-tree walk (tree_down_level_kernel / tree_up_level_kernel, BFieldElement)
template <int LOGN, bool UP>
int launch_tree_level_t(const tfk::TreeLevelArgs& a, hipStream_t stream) {
    constexpr int N = 1 << LOGN, WG = LOGN == 12 ? 512 : 256, T = WG / (N / 8);
    constexpr size_t lds = size_t(UP ? 3 : 2) * (tfk::lat_pad(N * T) + 8) * sizeof(u64);
    const void* fn = UP ? reinterpret_cast<const void*>(&tfk::tree_up_level_kernel<LOGN>) : reinterpret_cast<const void*>(&tfk::tree_down_level_kernel<LOGN>);
    if constexpr (lds > 48 * 1024) {
        static std::atomic<unsigned long long> done_mask{0};
        if (int rc_attr = ensure_dynamic_lds(fn, (int)((int)lds), done_mask)) return rc_attr;
    }
    const long long blocks = (a.lines + T - 1) / T;
    if (UP) hipLaunchKernelGGL((tfk::tree_up_level_kernel<LOGN>), dim3((unsigned)blocks), dim3(WG), lds, stream, a);
    else hipLaunchKernelGGL((tfk::tree_down_level_kernel<LOGN>), dim3((unsigned)blocks), dim3(WG), lds, stream, a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}
#ifdef TF_AB_BUILD
template <int LOGN, bool UP>
int launch_tree_level_xfe_t(const tfk::TreeLevelArgs& a, hipStream_t stream) {
    using G = tfk::TreeXfeGeom<LOGN>;
    constexpr size_t lds = size_t(UP ? 3 : 2) * G::BUF * sizeof(u64);
    static_assert(lds <= 160 * 1024, "level too long for one workgroup");
    const void* fn = UP ? reinterpret_cast<const void*>(&tfk::tree_up_level_xfe_kernel<LOGN>) : reinterpret_cast<const void*>(&tfk::tree_down_level_xfe_kernel<LOGN>);
    if constexpr (lds > 48 * 1024) {
        static std::atomic<unsigned long long> done_mask{0};
        if (int rc_attr = ensure_dynamic_lds(fn, (int)lds, done_mask)) return rc_attr;
    }
    const long long blocks = (a.lines + G::T - 1) / G::T;
    if (UP) hipLaunchKernelGGL((tfk::tree_up_level_xfe_kernel<LOGN>), dim3((unsigned)blocks), dim3(G::WG), lds, stream, a);
    else hipLaunchKernelGGL((tfk::tree_down_level_xfe_kernel<LOGN>), dim3((unsigned)blocks), dim3(G::WG), lds, stream, a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}
#endif
// When: the level's transforms are the latency-shaped kernel's anyway AND the level is small enough that launches, not work,
// are what it costs (measured crossover, tools/tree_latency.py / profiles/r03_tree_level_ab.txt).  TF_TREE_NO_LEVEL: A/B switch.
bool tree_level_wanted(long long order, long long lines, int L, bool up) {
    static const bool off = ab_env("TF_TREE_NO_LEVEL") != nullptr;
    static const bool on_xfe = ab_env("TF_TREE_LEVEL_XFE") != nullptr;  // measured loss, opt-in (below)
    static const long long limit = [] {
        const char* e = ab_env("TF_TREE_LEVEL_MAX_WORDS");
        return e ? atoll(e) : (1ll << 22);  // (every width the latency-shaped transform serves: faster at each, profiles/r03_tree_level_ab.txt)
    }();
    if (off || order < 64 || order > 4096 || g_min_passes.load(std::memory_order_relaxed) != 0) return false;
    // XFieldElement (tree_*_level_xfe_kernel: three thread groups per line; 2d <= 2048 fits a workgroup on the way down, 2d <= 1024
    // on the way up) is a measured LOSS and off unless TF_TREE_LEVEL_XFE is set: prepared tree, 2^12 points, evaluate 243 -> 258 us,
    // interpolate 160 -> 201 us (profiles/r03_tree_level_ab.txt) -- the extension-field products between the transforms are up to
    // five base-field products per element and limb, 8 elements per thread: they lengthen the one instruction stream that bounds a
    // level, where the separate pointwise kernels spread them over one thread per element.
    if (L == 3 && (!on_xfe || order > (up ? 1024 : 2048))) return false;
    if (!lat_wanted(ilog2((size_t)order), (size_t)lines, L)) return false;
    return lines * order * L <= limit;
}
template <bool UP>
int launch_tree_level(DeviceCtx* ctx, int log_n, tfk::TreeLevelArgs a, hipStream_t s, int L) {
    int rc = get_lat_table(ctx, log_n, false, &a.tw_f);
    if (!rc) rc = get_lat_table(ctx, log_n, true, &a.tw_i);
    if (rc) return rc;
    a.ninv = gl::mont_inverse(gl::to_mont(u64(1) << log_n));
#ifdef TF_AB_BUILD
    if (L == 3) {
        switch (log_n) {
            case 6: return launch_tree_level_xfe_t<6, UP>(a, s);
            case 7: return launch_tree_level_xfe_t<7, UP>(a, s);
            case 8: return launch_tree_level_xfe_t<8, UP>(a, s);
            case 9: return launch_tree_level_xfe_t<9, UP>(a, s);
            case 10: return launch_tree_level_xfe_t<10, UP>(a, s);
            case 11:
                if constexpr (!UP) return launch_tree_level_xfe_t<11, false>(a, s);
        }
        return TF_ERR_HIP;
    }
#else
    if (L == 3) return TF_ERR_HIP;  // (tree_level_wanted never says yes: the XFieldElement level kernels are a measured loss, TF_AB_BUILD)
#endif
    switch (log_n) {
        case 6: return launch_tree_level_t<6, UP>(a, s);
        case 7: return launch_tree_level_t<7, UP>(a, s);
        case 8: return launch_tree_level_t<8, UP>(a, s);
        case 9: return launch_tree_level_t<9, UP>(a, s);
        case 10: return launch_tree_level_t<10, UP>(a, s);
        case 11: return launch_tree_level_t<11, UP>(a, s);
        case 12: return launch_tree_level_t<12, UP>(a, s);
    }
    return TF_ERR_HIP;
}
template int launch_tree_level<false>(DeviceCtx*, int, tfk::TreeLevelArgs, hipStream_t, int);
template int launch_tree_level<true>(DeviceCtx*, int, tfk::TreeLevelArgs, hipStream_t, int);

// ---- one launch per level of a small zerofier-tree BUILD (tree_build_level_kernel, BFieldElement, 128 <= 2d <= 2048)
template <int LOGN2>
int launch_tree_build_level_t(const tfk::TreeBuildArgs& a, hipStream_t stream) {
    using G = tfk::TreeBuildGeom<LOGN2>;
    constexpr size_t lds = size_t(2) * G::BUF * sizeof(u64);
    static_assert(lds <= 160 * 1024, "level too long for one workgroup");
    if constexpr (lds > 48 * 1024) {
        static std::atomic<unsigned long long> done_mask{0};
        if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::tree_build_level_kernel<LOGN2>), (int)lds, done_mask)) return rc_attr;
    }
    const long long blocks = (a.parents + G::T - 1) / G::T;
    hipLaunchKernelGGL((tfk::tree_build_level_kernel<LOGN2>), dim3((unsigned)blocks), dim3(G::WG), lds, stream, a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}
// When: a small tree (the build of 2^12 points was 58 launches, 378 of the 655 us of a one-shot interpolation).  TF_TREE_NO_BUILD_LEVEL:
// A/B switch; TF_TREE_BUILD_MAX_WORDS: sweep hook (words of one level's transforms, 2 M).
bool tree_build_level_wanted(long long order, long long parents) {
    static const bool off = ab_env("TF_TREE_NO_BUILD_LEVEL") != nullptr;
    static const long long limit = [] {
        const char* e = ab_env("TF_TREE_BUILD_MAX_WORDS");
        return e ? atoll(e) : (1ll << 22);  // (2^16 / 2^18 words lose 7 % / 5 % on one-shot calls at 2^16 / 2^18 points; 2^20 and 2^22 level)
    }();
    if (off || order < 128 || order > 2048 || g_min_passes.load(std::memory_order_relaxed) != 0) return false;
    if (g_lat_mode.load(std::memory_order_relaxed) == 0 || !lat_wanted(ilog2((size_t)order), (size_t)(2 * parents), 1)) return false;
    return 2 * parents * order <= limit;
}
int launch_tree_build_level(DeviceCtx* ctx, int log_n2, tfk::TreeBuildArgs a, hipStream_t s) {
    int rc = get_lat_table(ctx, log_n2, false, &a.tw_f2);
    if (!rc) rc = get_lat_table(ctx, log_n2, true, &a.tw_i2);
    if (!rc) rc = get_lat_table(ctx, log_n2 + 1, false, &a.tw_f4);
    if (!rc) rc = get_lat_table(ctx, log_n2 + 1, true, &a.tw_i4);
    if (rc) return rc;
    a.ninv2 = gl::mont_inverse(gl::to_mont(u64(1) << log_n2));
    a.ninv4 = gl::mont_inverse(gl::to_mont(u64(2) << log_n2));
    switch (log_n2) {
        case 7: return launch_tree_build_level_t<7>(a, s);
        case 8: return launch_tree_build_level_t<8>(a, s);
        case 9: return launch_tree_build_level_t<9>(a, s);
        case 10: return launch_tree_build_level_t<10>(a, s);
        case 11: return launch_tree_build_level_t<11>(a, s);
    }
    return TF_ERR_HIP;
}

// ---- 2^13 .. 2^20 points with little work: the two passes of n = N1 N2 on the eight-elements-per-thread stages (ntt_lat2_kernel)
template <int LOGN, bool INV, bool LAST, int WG = 256>
int launch_lat2_t(const tfk::NttLat2Args& a, size_t batch, hipStream_t stream) {
    constexpr int N = 1 << LOGN, T = WG / (N / 8);
    constexpr size_t lds = size_t(2) * (tfk::lat_pad(N * T) + 8) * sizeof(u64);
    if constexpr (lds > 48 * 1024) {
        static std::atomic<unsigned long long> done_mask{0};
        if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_lat2_kernel<LOGN, INV, LAST, WG>), (int)((int)lds), done_mask)) return rc_attr;
    }
    tfk::NttLat2Args b = a;
    b.tiles_per_entry = (int)((a.lines + T - 1) / T);
    const long long blocks = (long long)batch * b.tiles_per_entry;
    hipLaunchKernelGGL((tfk::ntt_lat2_kernel<LOGN, INV, LAST, WG>), dim3((unsigned)blocks), dim3(WG), lds, stream, b);
    HIPCHK(hipGetLastError());
    return TF_OK;
}
template <bool INV, bool LAST>
int launch_lat2_dir(int log_n, const tfk::NttLat2Args& a, size_t batch, hipStream_t s) {
    switch (log_n) {
        case 6: return launch_lat2_t<6, INV, LAST>(a, batch, s);
        case 7: return launch_lat2_t<7, INV, LAST>(a, batch, s);
        case 8: return launch_lat2_t<8, INV, LAST>(a, batch, s);
        case 9: return launch_lat2_t<9, INV, LAST>(a, batch, s);
        case 10: {
            // 1024-point lines: 512-thread workgroups take four lines instead of two (32-byte segments on the column side)
            static const bool wide = ab_env("TF_NTT_LAT2_NO_WIDE") == nullptr;  // A/B switch
            return wide ? launch_lat2_t<10, INV, LAST, 512>(a, batch, s) : launch_lat2_t<10, INV, LAST>(a, batch, s);
        }
    }
    return TF_ERR_HIP;
}
bool lat2_wanted(int log_n, size_t batch, int L) {
    static const bool off = ab_env("TF_NTT_NO_LAT") != nullptr || ab_env("TF_NTT_NO_LAT2") != nullptr;  // A/B switches
    static const long long env_limit = [] {
        const char* e = ab_env("TF_NTT_LAT2_MAX_WORDS");
        return e ? atoll(e) : 0ll;
    }();
    const int mode = g_lat_mode.load(std::memory_order_relaxed);
    if (mode == 0 || (mode < 0 && off)) return false;
    if (log_n < 13 || log_n > 20) return false;
    if (mode == 1) return true;
    // measured crossover against the pass / block kernels, words per call (tools/lat_sweep.py 13 20, profiles/r03_lat2_sweep_*.txt):
    // one 2^16-point slice 33 -> 16 us; the win ends where the chip fills, and earlier for the longest lines (a 1024-point line
    // leaves two lines per workgroup: 16-byte segments)
    static const long long lim1[8] = {1ll << 21, 1ll << 21, 1ll << 21, 1ll << 21, 1ll << 21, 1ll << 20, 1ll << 20, 1ll << 20};  // log_n = 13 .. 20
    static const long long lim3[8] = {3ll << 20, 3ll << 20, 3ll << 19, 3ll << 20, 3ll << 19, 3ll << 18, 0, 0};
    const long long limit = env_limit ? env_limit : (L == 1 ? lim1 : lim3)[log_n - 13];
    return (long long)(batch * size_t(L)) << log_n <= limit;
}
int launch_lat2(DeviceCtx* ctx, const u64* in, u64* out, long long in_bs, long long out_bs, int log_n, size_t batch, int L, bool inverse,
                long long n_coeffs, const u64* in2, hipStream_t stream, const u64* pre_scale, const u64* post_scale) {
    const int a1 = (log_n + 1) / 2, a2 = log_n - a1;
    const long long N1 = 1ll << a1, N2 = 1ll << a2, n = 1ll << log_n;
    const u64 *tw1 = nullptr, *tw2 = nullptr, *post = nullptr;
    bool post_temp = false;
    int rc = get_lat_table(ctx, a1, inverse, &tw1, 0);
    if (!rc) rc = get_lat_table(ctx, a2, inverse, &tw2, inverse ? log_n : 0);
    if (!rc) rc = get_post_table(ctx, log_n, a1, inverse, stream, &post, &post_temp);
    if (rc) return rc;
    DeviceCtx::ScratchBlock sblk;
    rc = scratch_acquire(ctx, batch * (size_t)n * L * sizeof(u64), stream, &sblk);
    if (rc) {
        if (post_temp) (void)hipFreeAsync(const_cast<u64*>(post), stream);
        return rc;
    }
    tfk::NttLat2Args c{};  // column pass: the caller's input -> scratch
    c.in = in;
    c.out = sblk.p;
    c.in2 = in2;
    c.tw = tw1;
    c.post_tw = post;
    c.n_coeffs = n_coeffs;
    c.nc_es = N2;
    c.in_bs = in_bs;
    c.out_bs = n * L;
    c.lines = N2 * L;
    c.in_es = c.out_es = N2 * L;
    c.in_lhi = c.out_lhi = L;
    c.tw_rs = N2;
    c.scale_tab = pre_scale;
    c.scale_es = N2;
    c.L = L;
    c.cfast = 1;
    rc = inverse ? launch_lat2_dir<true, false>(a1, c, batch, stream) : launch_lat2_dir<false, false>(a1, c, batch, stream);
    if (!rc) {
        tfk::NttLat2Args r{};  // last pass: rows of the scratch -> natural order in the caller's output
        r.in = sblk.p;
        r.out = out;
        r.tw = tw2;
        r.n_coeffs = -1;
        r.in_bs = n * L;
        r.out_bs = out_bs;
        r.lines = N1 * L;
        r.in_es = L;
        r.in_lhi = N2 * L;
        r.out_es = N1 * L;
        r.out_lhi = L;
        r.scale = inverse ? gl::mont_inverse(gl::to_mont(u64(1) << log_n)) : 0;
        r.scale_tab = post_scale;
        r.scale_es = N1;
        r.L = L;
        r.cfast = 0;
        rc = inverse ? launch_lat2_dir<true, true>(a2, r, batch, stream) : launch_lat2_dir<false, true>(a2, r, batch, stream);
    }
    scratch_release(ctx, sblk, stream);
    if (post_temp) (void)hipFreeAsync(const_cast<u64*>(post), stream);
    return rc;
}


}  // namespace tfi
