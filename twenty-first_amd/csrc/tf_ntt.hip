// tf_ntt.hip -- device context, table caches and the NTT pass planner of libtf_hip.so: every launcher of ntt_kernels.h.
//
// One process drives one or more MI355X devices; all state is per HIP device and guarded by a mutex,
// kernels are enqueued on the caller's stream, and nothing here synchronises the device except the
// one-off construction of a twiddle table.  There is no CPU fallback anywhere in this library.
#include "tf_internal.h"
#include "ntt_kernels.h"

namespace tfi {


// ------------------------------------------------------------------------------------ errors
thread_local std::string t_last_error;

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    t_last_error = buf;
    (void)hipGetLastError();
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver || e == hipErrorNotInitialized)
        return TF_ERR_NO_DEVICE;
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) return TF_ERR_OUT_OF_MEMORY;
    return TF_ERR_HIP;
}

// what TF_ABI_CATCH (tf_guard.h) calls for an exception that reached an entry point
int abi_caught(const char* what, int status) noexcept {
    try {
        t_last_error = std::string("C++ exception at the C ABI: ") + (what ? what : "?");
    } catch (...) {  // (not even the message could be allocated: the status alone reports)
    }
    return status;
}

// ------------------------------------------------------------------------------------ field helpers (host)
// w_n = 7^((p-1)/n): equals every entry of PRIMITIVE_ROOTS (b_field_element.rs:43-78).
u64 root_of_unity_mont(int log_n) { return gl::mont_pow(gl::to_mont(7), (gl::P - 1) >> log_n); }

int ilog2(size_t v) {
    int l = 0;
    while ((size_t(1) << l) < v) ++l;
    return l;
}


// ------------------------------------------------------------------------------------ per-device context
// (struct DeviceCtx: tf_internal.h)
// What the caches may pin for the life of the process; a table that does not fit becomes a stream-ordered temporary
// that is rebuilt per call and released after the pass that reads it.
constexpr size_t kPostCacheBudget = size_t(4) << 30;   // inter-pass twiddles (2^20: 8 MiB, 2^28: 2 GiB)
constexpr size_t kPowCacheBudget = size_t(1) << 30;    // offset^j tables (n_coeffs words each)

DeviceCtx g_ctx[kMaxDevices];

// Grids that are sized from the CU count (launch_rows32, mx_blocks) take the CURRENT device's: round 5 cached the count of whichever
// device made the first call, which on a heterogeneous or partitioned node sized every other device's grids wrong (only
// speed was at stake -- those kernels walk their work with a grid stride).
int device_cus() {
    static std::atomic<int> cus[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        (void)hipGetLastError();
        return 256;
    }
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        n = 256;
    }
    cus[dev].store(n, std::memory_order_relaxed);
    return n;
}

int current_ctx(DeviceCtx** out) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        t_last_error = std::string("no usable HIP device: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return TF_ERR_NO_DEVICE;
    }
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return TF_ERR_NO_DEVICE;
    *out = &g_ctx[dev];
    if (!g_ctx[dev].pool_ready.load(std::memory_order_acquire)) {
        // The library's stream-ordered temporaries come from a pool of its own:
        //  - freed blocks stay cached (release threshold: the default 0 hands memory back to the OS at every synchronisation,
        //    which makes per-call work space expensive);
        //  - hipMemPoolReuseFollowEventDependencies OFF.  The scratch cache below makes callers' streams wait on each other's
        //    events; with that reuse policy on, the runtime then hands a block freed on stream A to stream B while A's kernels
        //    still use it (ROCm 7.0: several host threads walking one zerofier tree on their own streams got wrong words about
        //    once in 10^4 calls, tools/stress_threads.py; every other policy combination ran clean, profiles/r03_pool_reuse.txt).
        // The application's default pool is left alone.
        std::lock_guard<std::mutex> lk(g_ctx[dev].mu);
        if (!g_ctx[dev].pool_ready.load(std::memory_order_relaxed)) {
            hipMemPoolProps props{};
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            hipMemPool_t pool = nullptr;
            bool own = true;
            if (hipMemPoolCreate(&pool, &props) != hipSuccess) {
                // no pool of our own: the device's default one.  It belongs to the application, so only the ONE attribute correctness
                // needs is changed on it (the reuse policy below); its release threshold stays what the application set (tf_hip.h)
                pool = nullptr;
                own = false;
                (void)hipGetLastError();
                (void)hipDeviceGetDefaultMemPool(&pool, dev);
            }
            if (pool) {
                uint64_t thr = UINT64_MAX;
                if (own) (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
                int follow = 0;
                if (const char* e = ab_env("TF_POOL_REUSE_FOLLOW_EVENTS")) follow = atoi(e);  // (to reproduce the failure)
                (void)hipMemPoolSetAttribute(pool, hipMemPoolReuseFollowEventDependencies, &follow);
            }
            (void)hipGetLastError();
            g_ctx[dev].pool = pool;
            g_ctx[dev].pool_ready.store(true, std::memory_order_release);
        }
    }
    return TF_OK;
}

// Every stream-ordered temporary of the library: from the context's pool (hipFreeAsync gives it back, whatever the pool).
hipError_t pool_malloc_async(void** p, size_t bytes, hipStream_t stream) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemPool_t pool = (dev >= 0 && dev < kMaxDevices && g_ctx[dev].pool_ready.load(std::memory_order_acquire)) ? g_ctx[dev].pool : nullptr;
    if (!pool) {  // (an entry point that allocates before it looked its context up)
        DeviceCtx* ctx = nullptr;
        if (current_ctx(&ctx) == TF_OK) pool = ctx->pool;
    }
    return pool ? hipMallocFromPoolAsync(p, bytes, pool, stream) : hipMallocAsync(p, bytes, stream);
}

// Work space between the passes of a multi-pass transform.  Round 1 took it from the stream-ordered pool on every call; a
// call of another size in between (bench.py's parity sample, a coset evaluation) splits the pooled block and the next full-size
// call then pays a fresh 2 GiB device allocation inside its timed path (observed: +120 ms on one step).  Blocks are therefore
// kept whole in a small per-device cache and handed from call to call with an event fence: release records an event on the
// releasing stream, the next taker's stream waits for it -- no host synchronisation, any mix of streams and host threads.
constexpr size_t kScratchCacheBytes = size_t(12) << 30;  // blocks beyond this are freed when they come back
int scratch_acquire(DeviceCtx* ctx, size_t bytes, hipStream_t stream, DeviceCtx::ScratchBlock* out) {
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        int best = -1;
        for (int i = 0; i < (int)ctx->scratch_free.size(); ++i) {
            const auto& b = ctx->scratch_free[i];
            if (b.bytes >= bytes && (best < 0 || b.bytes < ctx->scratch_free[best].bytes)) best = i;
        }
        if (best >= 0 && ctx->scratch_free[best].bytes <= 4 * bytes + (size_t(64) << 20)) {  // do not pin a huge block under a small call
            *out = ctx->scratch_free[best];
            ctx->scratch_free.erase(ctx->scratch_free.begin() + best);
            ctx->scratch_bytes -= out->bytes;
        } else {
            out->p = nullptr;
        }
    }
    if (out->p) {
        hipError_t e = hipStreamWaitEvent(stream, out->ready, 0);
        if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent(scratch)", __FILE__, __LINE__);
        return TF_OK;
    }
    out->bytes = bytes;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&out->p), bytes);
    if (e != hipSuccess) {
        // make room: drop every cached block (after their last users) and retry once
        std::vector<DeviceCtx::ScratchBlock> drop;
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            drop.swap(ctx->scratch_free);
            ctx->scratch_bytes = 0;
        }
        (void)hipGetLastError();
        for (auto& b : drop) {
            (void)hipEventSynchronize(b.ready);
            (void)hipEventDestroy(b.ready);
            (void)hipFree(b.p);
        }
        e = hipMalloc(reinterpret_cast<void**>(&out->p), bytes);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc(ntt scratch)", __FILE__, __LINE__);
    }
    e = hipEventCreateWithFlags(&out->ready, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipFree(out->p);
        return hip_fail(e, "hipEventCreate(scratch)", __FILE__, __LINE__);
    }
    return TF_OK;
}
void scratch_release(DeviceCtx* ctx, DeviceCtx::ScratchBlock blk, hipStream_t stream) {
    if (!blk.p) return;
    if (hipEventRecord(blk.ready, stream) != hipSuccess) {  // cannot fence it: wait, then free
        (void)hipGetLastError();
        (void)hipStreamSynchronize(stream);
        (void)hipEventDestroy(blk.ready);
        (void)hipFree(blk.p);
        return;
    }
    // The block that comes back is the one most likely to be wanted again: it is always kept (unless it alone exceeds the budget) and the
    // OLDEST cached blocks make room for it.  Round 5 dropped the NEW block when the cache was full -- after sixteen blocks of other sizes
    // had accumulated (a long-lived process, the test suite) every release then waited for its stream and every acquire allocated from
    // the device, for ever (0.9 ms of host time per call in test_one_host_thread_round_robin_never_blocks).  The evicted blocks were
    // released long ago, so waiting for their events does not wait for anything.
    std::vector<DeviceCtx::ScratchBlock> evict;
    bool keep;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        keep = blk.bytes <= kScratchCacheBytes;
        if (keep) {
            while (!ctx->scratch_free.empty() && (ctx->scratch_bytes + blk.bytes > kScratchCacheBytes || ctx->scratch_free.size() >= 16)) {
                evict.push_back(ctx->scratch_free.front());  // (release order = vector order: the front is the least recently used)
                ctx->scratch_bytes -= ctx->scratch_free.front().bytes;
                ctx->scratch_free.erase(ctx->scratch_free.begin());
            }
            ctx->scratch_free.push_back(blk);
            ctx->scratch_bytes += blk.bytes;
        }
    }
    for (auto& b : evict) {
        (void)hipEventSynchronize(b.ready);
        (void)hipEventDestroy(b.ready);
        (void)hipFree(b.p);
    }
    if (!keep) {
        (void)hipEventSynchronize(blk.ready);
        (void)hipEventDestroy(blk.ready);
        (void)hipFree(blk.p);
    }
}

// tf_release_caches: give back what the current device's context pins for speed -- the scratch blocks between transform passes
// (up to 12 GiB), the inter-pass twiddle tables (up to 4 GiB) and the coset power tables (up to 1 GiB).  Waits for the device
// first (tables may be read by kernels in flight); the next call rebuilds what it needs.  The small tables (inner twiddles,
// Tip5 constants) stay.
int release_caches(DeviceCtx* ctx) {
    HIPCHK(hipDeviceSynchronize());
    std::vector<DeviceCtx::ScratchBlock> drop;
    std::vector<u64*> tabs;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        drop.swap(ctx->scratch_free);
        ctx->scratch_bytes = 0;
        for (auto it = ctx->tables.begin(); it != ctx->tables.end();) {
            if ((it->first >> 56) == 2 /* TAG_POST */) {
                tabs.push_back(it->second);
                it = ctx->tables.erase(it);
            } else {
                ++it;
            }
        }
        ctx->cached_post_bytes = 0;
        for (auto& kv : ctx->pow_tables) tabs.push_back(kv.second);
        ctx->pow_tables.clear();
        ctx->cached_pow_bytes = 0;
        for (auto& kv : ctx->scaled_post) tabs.push_back(kv.second);
        ctx->scaled_post.clear();
        ctx->cached_scaled_post_bytes = 0;
    }
    for (auto& b : drop) {
        (void)hipEventDestroy(b.ready);
        (void)hipFree(b.p);
    }
    for (u64* t : tabs) (void)hipFree(t);
    if (ctx->pool) (void)hipMemPoolTrimTo(ctx->pool, 0);  // the cached stream-ordered temporaries too
    (void)hipGetLastError();
    return TF_OK;
}

size_t g_tile_bytes = 0;
// Pipelined tiles: with g_pipe = K > 1 the batch tiles of a multi-pass transform are dealt round-robin to K side streams,
// each with its own scratch tile, so that the column pass of tile t + 1 runs beside the transposing pass of tile t: the
// launches' tails fill each other and a tile sized for the 256 MiB Infinity Cache is re-read out of it.  1 = one stream.
std::atomic<int> g_pipe{0};
std::atomic<int> g_nt{-1};  // TF_NTT_NT / tf_set_ntt_nt: bit 0 non-temporal input loads (first pass), bit 1 non-temporal output stores (last pass)
std::once_flag g_env_once;
void read_env() {
    std::call_once(g_env_once, [] {
        if (g_tile_bytes == 0) {
            const char* s = getenv("TF_NTT_TILE_BYTES");
            g_tile_bytes = s ? strtoull(s, nullptr, 10) : (size_t(2048) << 20);
            if (g_tile_bytes == 0) g_tile_bytes = size_t(2048) << 20;
        }
        if (g_pipe.load() == 0) {  // 0 = automatic (run_ntt decides by plan); TF_NTT_PIPE / tf_set_ntt_pipe pin it
            const char* s = getenv("TF_NTT_PIPE");
            if (s && atoi(s) > 0) g_pipe.store(std::min(atoi(s), kMaxPipe));
        }
        if (g_nt.load() < 0) {
            const char* s = ab_env("TF_NTT_NT");
            g_nt.store(s ? (atoi(s) & 3) : 0);
        }
    });
}

int upload_table(const std::vector<u64>& host, u64** dev) {
    u64* d = nullptr;
    HIPCHK(hipMalloc(&d, host.size() * sizeof(u64)));
    hipError_t e = hipMemcpy(d, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(e, "hipMemcpy(table)", __FILE__, __LINE__);
    }
    *dev = d;
    return TF_OK;
}

// split powers: hi[i] = base^(i << h), lo[i] = base^i  (both Montgomery)
void split_powers(u64 base, int log_total, int* h_out, std::vector<u64>* hi, std::vector<u64>* lo) {
    int h = (log_total + 1) / 2;
    size_t nlo = size_t(1) << h, nhi = size_t(1) << (log_total - h);
    lo->resize(nlo);
    hi->resize(nhi);
    u64 acc = gl::ONE;
    for (size_t i = 0; i < nlo; ++i) {
        (*lo)[i] = acc;
        acc = gl::mont_mul(acc, base);
    }
    u64 step = acc;  // base^(2^h)
    acc = gl::ONE;
    for (size_t i = 0; i < nhi; ++i) {
        (*hi)[i] = acc;
        acc = gl::mont_mul(acc, step);
    }
    *h_out = h;
}

// (table keys: tf_internal.h)

// inner[g*32 + k1] = w_R^(+-g*k1) * (scale_log_n ? n^-1 : 1),  R = 32 << p2
// pre = 2 or 4 (a = 10 only): the pass is a 2048- / 4096-point DFT run as `pre` 1024-point workgroups per tile (ntt_kernels.h, PRE2 /
// PRE4); `pre` tables back to back, table q: inner[q * 1024 + g*32 + k1] = w_1024^(+-g k1) * w_{1024 pre}^(+-q g) * scale -- the part
// of the radix-`pre` stage's twiddle that depends on g rides on the inner twiddle.  pre = 4: 64 more words follow,
// stw[(q >> 1) * 32 + i] = w_128^(+-q i), q = 1, 3: the per-slot part of w_4096^(q c), c = g + 32 i (*stw_out).
int get_inner_table(DeviceCtx* ctx, int a, bool inverse, int scale_log_n, const u64** out, int pre = 1, const u64** stw_out = nullptr) {
    const int p2 = a - 5;
    if (stw_out) *stw_out = nullptr;
    if (p2 == 0 && scale_log_n == 0) {
        *out = nullptr;
        return TF_OK;
    }
    const u64 key = make_key(TAG_INNER, a, inverse, scale_log_n, pre > 1 ? pre - 1 : 0);  // (pre = 2 keeps its round-3 key)
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int P2 = 1 << p2;
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = it->second;
        if (stw_out && pre == 4) *stw_out = it->second + size_t(P2) * 32 * 4;
        return TF_OK;
    }
    u64 w = root_of_unity_mont(a);
    if (inverse) w = gl::mont_inverse(w);
    u64 scale = gl::ONE;
    if (scale_log_n) scale = gl::mont_inverse(gl::to_mont(u64(1) << scale_log_n));
    std::vector<u64> t(size_t(P2) * 32);
    u64 wg = gl::ONE;  // w^g
    for (int g = 0; g < P2; ++g) {
        u64 acc = scale;
        for (int k = 0; k < 32; ++k) {
            t[size_t(g) * 32 + k] = acc;
            acc = gl::mont_mul(acc, wg);
        }
        wg = gl::mont_mul(wg, w);
    }
    if (pre > 1) {
        u64 wp = root_of_unity_mont(a + (pre == 4 ? 2 : 1));  // w_{R pre}
        if (inverse) wp = gl::mont_inverse(wp);
        t.resize(size_t(P2) * 32 * pre + (pre == 4 ? 64 : 0));
        u64 wq = wp;  // w_{R pre}^q
        for (int q = 1; q < pre; ++q) {
            u64 wqg = gl::ONE;  // w_{R pre}^(q g)
            for (int g = 0; g < P2; ++g) {
                for (int k = 0; k < 32; ++k) t[size_t(q) * P2 * 32 + size_t(g) * 32 + k] = gl::mont_mul(t[size_t(g) * 32 + k], wqg);
                wqg = gl::mont_mul(wqg, wq);
            }
            wq = gl::mont_mul(wq, wp);
        }
        if (pre == 4) {
            const u64 w128 = gl::mont_pow(wp, u64(P2));  // w_{R pre}^32 = w_{R pre / 32}: w_128 for R = 1024
            for (int h = 0; h < 2; ++h) {
                const u64 step = gl::mont_pow(w128, u64(2 * h + 1));
                u64 acc = gl::ONE;
                for (int i = 0; i < 32; ++i) {
                    t[size_t(P2) * 32 * 4 + size_t(h) * 32 + i] = acc;
                    acc = gl::mont_mul(acc, step);
                }
            }
        }
    }
    u64* d = nullptr;
    int rc = upload_table(t, &d);
    if (rc) return rc;
    ctx->tables[key] = d;
    *out = d;
    if (stw_out && pre == 4) *stw_out = d + size_t(P2) * 32 * 4;
    return TF_OK;
}

// T[k*B + b] = w_M^(+-k*b), k < R = 2^a, b < B = M / R
// Inter-pass twiddles T[k * B + b] = w_M^(k * b), M = 2^log_m = R * B.  Tables up to 2^28 entries (2 GiB) are built once
// and cached; larger ones (single transforms of 2^29 .. 2^31 points) are stream-ordered temporaries: *temp = true and
// the caller releases them with hipFreeAsync after the pass that reads them.
constexpr int kMaxCachedPostLog = 28;
// Tables of up to 2^22 words (32 MiB) are cached whatever the budget says: round 6 found that a process which had once transformed
// 2^27 / 2^28 points (two tables of 1-2 GiB) ran every LATER shape on temporaries -- each call then built its table, and the builder of
// that time uploaded host-computed split tables and waited for the stream (0.85 ms of host time per call of a 2^16-point coset
// evaluation in tests/test_gpu_parity.py::test_one_host_thread_round_robin_never_blocks).  Temporaries are now built entirely on the
// caller's stream (their split tables by square-and-multiply on the device) and nothing waits.
constexpr int kAlwaysCachedPostLog = 22;
int get_post_table(DeviceCtx* ctx, int log_m, int a, bool inverse, hipStream_t stream, const u64** out, bool* temp) {
    *temp = log_m > kMaxCachedPostLog;
    const u64 key = make_key(TAG_POST, log_m, a, inverse, 0);
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (!*temp) {
        auto it = ctx->tables.find(key);
        if (it != ctx->tables.end()) {
            *out = it->second;
            return TF_OK;
        }
        if (log_m > kAlwaysCachedPostLog && ctx->cached_post_bytes + (sizeof(u64) << log_m) > kPostCacheBudget) *temp = true;  // over budget: temporary
    }
    u64 w = root_of_unity_mont(log_m);
    if (inverse) w = gl::mont_inverse(w);
    const long long M = 1ll << log_m, R = 1ll << a, B = M / R;
    const int threads = 256;
    const long long blocks = (M + threads - 1) / threads;
    u64* d = nullptr;
    if (*temp) {
        lk.unlock();
        // hi[i] = w^(i << h), lo[i] = w^i (split_powers), computed ON the device and freed behind the build, all on the caller's stream
        const int h = (log_m + 1) / 2;
        const long long nlo = 1ll << h, nhi = 1ll << (log_m - h);
        u64* d_split = nullptr;
        hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&d_split), size_t(nlo + nhi) * sizeof(u64), stream);
        if (e == hipSuccess) e = pool_malloc_async(reinterpret_cast<void**>(&d), size_t(M) * sizeof(u64), stream);
        if (e != hipSuccess) {
            if (d_split) (void)hipFreeAsync(d_split, stream);
            return hip_fail(e, "pool_malloc_async(twiddle table)", __FILE__, __LINE__);
        }
        hipLaunchKernelGGL(tfk::build_split_powers_kernel, dim3((unsigned)((nhi + 255) / 256)), dim3(256), 0, stream, d_split, w, h, nhi);
        hipLaunchKernelGGL(tfk::build_split_powers_kernel, dim3((unsigned)((nlo + 255) / 256)), dim3(256), 0, stream, d_split + nhi, w, 0, nlo);
        hipLaunchKernelGGL(tfk::build_post_tw_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, d, d_split, d_split + nhi, h, R, B);
        e = hipGetLastError();
        (void)hipFreeAsync(d_split, stream);
        if (e != hipSuccess) {
            (void)hipFreeAsync(d, stream);
            return hip_fail(e, "build_post_tw_kernel (temporary)", __FILE__, __LINE__);
        }
        *out = d;
        return TF_OK;
    }
    int h = 0;
    std::vector<u64> hi, lo;
    split_powers(w, log_m, &h, &hi, &lo);
    u64 *d_hi = nullptr, *d_lo = nullptr;
    int rc = upload_table(hi, &d_hi);
    if (rc) return rc;
    rc = upload_table(lo, &d_lo);
    if (rc) {
        (void)hipFree(d_hi);
        return rc;
    }
    hipError_t e = hipMalloc(&d, size_t(M) * sizeof(u64));
    if (e != hipSuccess) {
        (void)hipFree(d_hi);
        (void)hipFree(d_lo);
        return hip_fail(e, "hipMalloc(twiddle table)", __FILE__, __LINE__);
    }
    hipLaunchKernelGGL(tfk::build_post_tw_kernel, dim3((unsigned)blocks), dim3(threads), 0, hipStream_t(0), d, d_hi, d_lo, h, R, B);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(hipStream_t(0));
    (void)hipFree(d_hi);
    (void)hipFree(d_lo);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(e, "build_post_tw_kernel", __FILE__, __LINE__);
    }
    ctx->tables[key] = d;
    ctx->cached_post_bytes += size_t(M) * sizeof(u64);
    *out = d;
    return TF_OK;
}

// tables of ntt_block_kernel (2^11 <= n <= 2^14): tw1[q * REST + rest] = w_n^(+-q * rest), tw2[k2 * P3 + j3] =
// w_{32 P3}^(+-k2 * j3) (* n^-1 for the inverse)
int get_block_tables(DeviceCtx* ctx, int log_n, bool inverse, const u64** tw1, const u64** tw2) {
    const u64 key1 = make_key(TAG_BLOCK1, log_n, inverse, 0, 0), key2 = make_key(TAG_BLOCK2, log_n, inverse, 0, 0);
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto i1 = ctx->tables.find(key1), i2 = ctx->tables.find(key2);
    if (i1 != ctx->tables.end() && i2 != ctx->tables.end()) {
        *tw1 = i1->second;
        *tw2 = i2->second;
        return TF_OK;
    }
    const int n = 1 << log_n, rest_n = n / 32, p3 = n / 1024;
    u64 w = root_of_unity_mont(log_n);
    if (inverse) w = gl::mont_inverse(w);
    std::vector<u64> t1((size_t)n), t2((size_t)32 * p3);
    u64 wq = gl::ONE;  // w^q
    for (int q = 0; q < 32; ++q) {
        u64 acc = gl::ONE;
        for (int r = 0; r < rest_n; ++r) {
            t1[size_t(q) * rest_n + r] = acc;
            acc = gl::mont_mul(acc, wq);
        }
        wq = gl::mont_mul(wq, w);
    }
    const u64 w32 = gl::mont_pow(w, 32);  // w_{n / 32} = w_{32 P3}
    const u64 scale = inverse ? gl::mont_inverse(gl::to_mont(u64(n))) : gl::ONE;
    u64 wk = gl::ONE;  // w32^k2
    for (int k2 = 0; k2 < 32; ++k2) {
        u64 acc = scale;
        for (int j3 = 0; j3 < p3; ++j3) {
            t2[size_t(k2) * p3 + j3] = acc;
            acc = gl::mont_mul(acc, wk);
        }
        wk = gl::mont_mul(wk, w32);
    }
    u64 *d1 = nullptr, *d2 = nullptr;
    int rc = upload_table(t1, &d1);
    if (rc) return rc;
    rc = upload_table(t2, &d2);
    if (rc) {
        (void)hipFree(d1);
        return rc;
    }
    ctx->tables[key1] = d1;
    ctx->tables[key2] = d2;
    *tw1 = d1;
    *tw2 = d2;
    return TF_OK;
}

// stage tables of the reference (ntt.rs:309-324) for n <= 16
int get_tiny_table(DeviceCtx* ctx, int log_n, bool inverse, const u64** out) {
    const u64 key = make_key(TAG_TINY, log_n, inverse, 0, 0);
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = it->second;
        return TF_OK;
    }
    const int n = 1 << log_n;
    std::vector<u64> t(std::max(1, n - 1), gl::ONE);
    u64 w = root_of_unity_mont(log_n);
    if (inverse) w = gl::mont_inverse(w);
    for (int i = 0; i < log_n; ++i) {
        const int m = 1 << i;
        u64 wm = gl::mont_pow(w, u64(n / (2 * m)));
        u64 acc = gl::ONE;
        for (int j = 0; j < m; ++j) {
            t[m - 1 + j] = acc;
            acc = gl::mont_mul(acc, wm);
        }
    }
    u64* d = nullptr;
    int rc = upload_table(t, &d);
    if (rc) return rc;
    ctx->tables[key] = d;
    *out = d;
    return TF_OK;
}

// offset^j, j < n  (the power chain of Polynomial::scale, polynomial.rs:766-771).
// Up to 16 tables per device are cached for the life of the process; beyond that a table is built into a
// stream-ordered temporary (*temp = true) that the caller releases with hipFreeAsync after its launches, so no
// table another thread may still be using is ever freed.
int build_pow_tables(u64 offset_raw, u64 w, size_t cosets, size_t n, u64* d, hipStream_t s) {
    // table c (c < cosets) = powers of base_c = offset * w^c: out[c * n + j] = base_c^j = HI_c[j >> h] * LO_c[j & (2^h - 1)].
    // The split tables of ALL cosets go up in one allocation and ONE kernel fills every table (grid.y = coset).
    const int log_total = std::max(1, ilog2(n));
    int h = 0;
    std::vector<u64> hi, lo, all;
    size_t nhi = 0, nlo = 0;
    u64 base = offset_raw;
    for (size_t c = 0; c < cosets; ++c) {
        split_powers(base, log_total, &h, &hi, &lo);
        nhi = hi.size(), nlo = lo.size();
        if (c == 0) all.reserve(cosets * (nhi + nlo));
        all.insert(all.end(), hi.begin(), hi.end());
        all.insert(all.end(), lo.begin(), lo.end());
        base = gl::mont_mul(base, w);
    }
    u64* d_all = nullptr;
    int rc = upload_table(all, &d_all);
    if (rc) return rc;
    const int threads = 256;
    const long long blocks = ((long long)n + threads - 1) / threads;
    if (n) {
        hipLaunchKernelGGL(tfk::build_pow_tables_kernel, dim3((unsigned)blocks, (unsigned)cosets), dim3(threads), 0, s, d, d_all, h,
                           (long long)n, (long long)nhi, (long long)nlo);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // the split tables are freed below; the build kernel is microseconds
    (void)hipFree(d_all);
    if (e != hipSuccess) return hip_fail(e, "build_pow_tables_kernel", __FILE__, __LINE__);
    return TF_OK;
}

// cosets = 1: out[j] = offset^j, j < n.  cosets = C > 1 (blown-up coset evaluation, see run_ntt; C <= kMaxCosetSplit): C tables
// back to back, out[c * n + j] = (offset * w_{C * len}^c)^j with len the power-of-two transform length the n coefficients are
// padded to.  At most 16 tables / kPowCacheBudget bytes stay cached; anything beyond is a stream-ordered temporary (*temp).
constexpr size_t kMaxCosetSplit = 64;
int get_pow_table(DeviceCtx* ctx, u64 offset_raw, size_t n, hipStream_t stream, const u64** out, bool* temp, size_t cosets,
                  int log_order) {
    *temp = false;
    std::unique_lock<std::mutex> lk(ctx->mu);
    auto key = std::make_pair(offset_raw, u64(n) | (u64(cosets) << 40) | (u64(log_order) << 56));
    auto it = ctx->pow_tables.find(key);
    if (it != ctx->pow_tables.end()) {
        *out = it->second;
        return TF_OK;
    }
    const size_t words = std::max<size_t>(n * cosets, 1);
    const bool cacheable = ctx->pow_tables.size() < 16 && ctx->cached_pow_bytes + words * sizeof(u64) <= kPowCacheBudget;
    const u64 w = cosets > 1 ? root_of_unity_mont(log_order) : gl::ONE;
    u64* d = nullptr;
    if (cacheable) {
        HIPCHK(hipMalloc(&d, words * sizeof(u64)));
        int rc = build_pow_tables(offset_raw, w, cosets, n, d, 0);
        if (rc) {
            (void)hipFree(d);
            return rc;
        }
        ctx->pow_tables[key] = d;
        ctx->cached_pow_bytes += words * sizeof(u64);
        *out = d;
        return TF_OK;
    }
    lk.unlock();  // a temporary is private to this call: build it without holding the device context
    // ... and without leaving the caller's stream.  Round 6 (tests/test_gpu_parity.py::test_one_host_thread_round_robin_never_blocks, a
    // process whose cache is full): (1) the split-table builder uploads host-built tables and WAITS for the stream -- the temporary is
    // built by one kernel instead, every word by square-and-multiply; (2) hipFreeAsync of a pool block behind the launches blocked the
    // host for the depth of the queue (0.8-1.0 ms per call, measured around that one call) when several streams share the pool -- the
    // temporary therefore lives in a block of the scratch cache (hipMalloc'ed blocks handed on with event fences, no stream-ordered free).
    DeviceCtx::ScratchBlock blk;
    int rc = scratch_acquire(ctx, words * sizeof(u64), stream, &blk);
    if (rc) return rc;
    d = blk.p;
    {
        tfk::PowBases pb{};
        u64 base = offset_raw;
        for (size_t c = 0; c < cosets && c < 64; ++c) {
            pb.base[c] = base;
            base = gl::mont_mul(base, w);
        }
        if (n) hipLaunchKernelGGL(tfk::build_pow_tables_direct_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)cosets), dim3(256), 0, stream, d, pb, (long long)n);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            scratch_release(ctx, blk, stream);
            return hip_fail(e, "build_pow_tables_direct_kernel", __FILE__, __LINE__);
        }
    }
    {
        std::lock_guard<std::mutex> lk2(ctx->mu);
        ctx->temp_pow[d] = blk;
    }
    *temp = true;
    *out = d;
    return TF_OK;
}
void release_pow_table(DeviceCtx* ctx, const u64* table, bool temp, hipStream_t stream) {
    if (!temp || !table) return;
    DeviceCtx::ScratchBlock blk;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->temp_pow.find(table);
        if (it == ctx->temp_pow.end()) return;
        blk = it->second;
        ctx->temp_pow.erase(it);
    }
    scratch_release(ctx, blk, stream);
}

// The inter-pass table of a forward first pass with the COLUMN part of a coset evaluation's scaling folded in: T'[k B + b] =
// w_M^(k b) * offset^b (ntt_col2048_kernel: the factor is constant along a column, so it commutes with the column pass).  One table per
// (offset, M, R); at most four / kScaledPostBudget bytes stay cached (a prover evaluates on one coset), anything beyond is a
// stream-ordered temporary the caller frees after the pass.  Built on `stream`; the first build of a cached table synchronises it once.
constexpr size_t kScaledPostBudget = size_t(512) << 20;
int get_scaled_post_table(DeviceCtx* ctx, int log_m, int a, u64 offset_raw, hipStream_t stream, const u64** out, bool* temp) {
    *temp = false;
    const auto key = std::make_pair(offset_raw, (u64(log_m) << 8) | u64(a));
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->scaled_post.find(key);
        if (it != ctx->scaled_post.end()) {
            *out = it->second;
            return TF_OK;
        }
    }
    const long long M = 1ll << log_m, B = M >> a;
    const u64* T = nullptr;
    bool t_temp = false;
    int rc = get_post_table(ctx, log_m, a, false, stream, &T, &t_temp);
    if (rc) return rc;
    const u64* S = nullptr;
    bool s_temp = false;
    rc = get_pow_table(ctx, offset_raw, (size_t)B, stream, &S, &s_temp);
    if (rc) {
        if (t_temp) (void)hipFreeAsync(const_cast<u64*>(T), stream);
        return rc;
    }
    bool cache;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        cache = ctx->scaled_post.size() < 4 && ctx->cached_scaled_post_bytes + size_t(M) * sizeof(u64) <= kScaledPostBudget;
    }
    u64* d = nullptr;
    hipError_t e = cache ? hipMalloc(&d, size_t(M) * sizeof(u64)) : pool_malloc_async(reinterpret_cast<void**>(&d), size_t(M) * sizeof(u64), stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(tfk::scale_post_tw_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, d, T, S, B, M);
        e = hipGetLastError();
        if (e == hipSuccess && cache) e = hipStreamSynchronize(stream);  // other streams may use the cached table from now on
    }
    if (t_temp) (void)hipFreeAsync(const_cast<u64*>(T), stream);
    release_pow_table(ctx, S, s_temp, stream);
    if (e != hipSuccess) {
        if (d) { if (cache) (void)hipFree(d); else (void)hipFreeAsync(d, stream); }
        return hip_fail(e, "scale_post_tw_kernel", __FILE__, __LINE__);
    }
    if (cache) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->scaled_post.find(key);
        if (it != ctx->scaled_post.end()) {  // another host thread built it meanwhile
            (void)hipFree(d);
            d = it->second;
        } else {
            ctx->scaled_post[key] = d;
            ctx->cached_scaled_post_bytes += size_t(M) * sizeof(u64);
        }
    } else {
        *temp = true;
    }
    *out = d;
    return TF_OK;
}

// ------------------------------------------------------------------------------------ NTT planner
struct Launch {
    tfk::NttPassArgs a;
    unsigned tiles;
    unsigned threads;
    size_t lds_bytes;
    bool bad_geometry = false;  // planner self-check failed: launch_pass refuses the launch
};

int pad_to_residue(int base, int residue) {  // smallest s >= base with s == residue (mod 32)
    int r = ((residue - base) % 32 + 32) % 32;
    return base + r;
}

// Workgroup geometry (tunable for A/B runs through TF_NTT_WG_THREADS = 256 | 512):
//   512 threads: 16 columns per tile (128-byte segments), 64 KiB exchange rounds, 2 workgroups per CU;
//   256 threads:  8 columns per tile (64-byte segments, adjacent tiles paired on one XCD), 32 KiB rounds, 4 per CU.
// A call with too little work to fill the chip with 512-thread tiles (a single slice of <= 2^20 points: 64 tiles for 256 CUs) is
// planned with 256-thread workgroups and the generic last pass instead -- twice as many tiles of half the width: 2^16 44 -> 37 us,
// 2^18 47.5 -> 39.7, 2^20 51.8 -> 43.8 us per call; from 2^22 words per call on the wide tiles win (tools/small_batch.py).
// run_ntt sets the mode for the duration of one call (thread-local: the ABI is re-entrant).
thread_local bool t_small_launch = false;
std::atomic<int> g_small_launch_mode{-1};  // tf_set_ntt_small_launch: -1 automatic, 0 never, 1 always (tests)
int wg_env() {
    static const int v = [] {
        const char* e = ab_env("TF_NTT_WG_THREADS");
        return (e && atoi(e) == 256) ? 256 : ((e && atoi(e) == 512) ? 512 : 0);
    }();
    return v;
}
int wg_threads() {
    if (wg_env()) return wg_env();
    return t_small_launch ? 256 : 512;
}
int round_elems() {
    static const int v = [] {
        const char* e = ab_env("TF_NTT_ROUND_ELEMS");
        const int r = e ? atoi(e) : 0;
        return r >= 1024 ? r : 0;
    }();
    return v ? v : wg_threads() * 16;
}

// thread / LDS geometry shared by all pass types: nc columns, exchanged in rounds of cpr columns
void finish_geometry(Launch* l, int nc, int p2) {
    tfk::NttPassArgs& A = l->a;
    const int P2 = 1 << p2, R = 32 << p2;
    A.p2 = p2;
    A.nc = nc;
    A.cpr = std::max(1, std::min(nc, round_elems() / R));
    A.nrounds = (nc + A.cpr - 1) / A.cpr;
    A.s2 = A.cpr;
    A.s3 = 1;
    A.s1 = pad_to_residue(P2 * A.cpr, A.cpr % 32);  // consecutive k1 rows land cpr banks apart: conflict-free reads
    l->threads = (unsigned)(nc * P2);
    l->lds_bytes = size_t(32) * A.s1 * sizeof(u64);
    A.nc_magic = nc > 1 ? (u32)((u64(1) << 32) / (u64)nc + 1) : 0;  // umulhi(t, magic) == t / nc (nc == 1: kernel uses t)
    for (u32 t = 0; t < l->threads; ++t) {
        const u32 q = (u32)(((u64)t * A.nc_magic) >> 32);
        if (nc != 1 && q != t / (u32)nc) l->bad_geometry = true;  // never observed: the magic is exact for t < 1024, nc <= 1024
    }
}

// Column pass: view [batch][outer][R][B*L words]; DFT along R for each of the B*L word-columns; same position in and out.
// pre2: a = 11, run as pairs of 1024-point halves (ntt_kernels.h, PRE2): the kernel radix is 1024, the rows of a column 2048.
Launch plan_column_pass(const u64* in, u64* out, long long in_bs, long long out_bs, size_t batch, long long outer, int a,
                        long long B, int L, bool pre2 = false) {
    Launch l{};
    tfk::NttPassArgs& A = l.a;
    const int p2 = (pre2 ? a - 1 : a) - 5, P2 = 1 << p2;
    const long long R = 1ll << a, Bw = B * L;
    int nc = (int)std::min<long long>(std::max(1, wg_threads() / P2), Bw);
    A.in = in;
    A.out = out;
    A.L = L;
    A.d1 = (u32)outer;
    A.d2 = (u32)((Bw + nc - 1) / nc);
    A.d01 = (u32)(batch * outer);
    A.ib0 = in_bs;
    A.ib1 = R * Bw;
    A.ib2 = nc;
    A.ob0 = out_bs;
    A.ob1 = R * Bw;
    A.ob2 = nc;
    A.in_cs_hi = L;
    A.out_cs_hi = L;
    A.in_rs = Bw;
    A.out_rs = Bw;
    A.tw_rs = B;
    A.ps_rs = B;
    A.ps_col = 1;
    A.col_limit = (int)Bw;
    A.n_coeffs = -1;
    A.n_out = -1;
    {
        const int G = (nc * (int)sizeof(u64) < 128) ? 2 : 1;  // pair tiles narrower than a 128-byte line
        A.xcd_order = (A.d2 % (8 * G) == 0) ? G : ((A.d2 % 8 == 0) ? 1 : 0);
        // table slice of one column tile = R rows x nc words; an XCD owns d2 / 8 column tiles and has a 4 MiB L2
        if (A.xcd_order) A.xcd_colfast = (size_t(A.d2 / 8) * size_t(R) * nc * sizeof(u64) <= (size_t(2) << 20)) ? 1 : 0;
    }
    finish_geometry(&l, nc, p2);
    l.tiles = (unsigned)(batch * outer * A.d2);
    if (pre2) {
        A.out_rs = 2 * Bw;  // kernel row k' of half h is row 2 k' + h of the 2048
        A.tw_rs = 2 * B;
        A.pre2_in_off = 1024 * Bw;
        A.pre2_out_off = Bw;
        A.pre2_tw_off = B;
        A.pre2_map = (l.tiles % 8 == 0) ? 1 : 2;
        l.tiles *= 2;
    }
    return l;
}

#ifdef TF_AB_BUILD
std::atomic<int> g_ablate_cfg{-1};  // measurement only (TF_NTT_ABLATE=1|2 selects an ablated forward kernel; results are then garbage)
int ablate_mode() {
    int v = g_ablate_cfg.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = ab_env("TF_NTT_ABLATE");
        v = e ? atoi(e) : 0;
        g_ablate_cfg.store(v, std::memory_order_relaxed);
    }
    return v;
}
#else
constexpr int ablate_mode() { return 0; }  // the ablated kernels (no memory / no arithmetic / phase stamps) are laboratory instantiations
#endif
// The R = 1024 instantiations address memory through buffer resources: resource base + 32-bit per-thread offset + 32-bit
// per-slot offset (ntt_kernels.h, buf_load).  A thread's row offset is at most 31 rows, a slot's at most 992 rows: with
// row strides of rs words everything stays below 2^32 bytes when 1024 * rs * 8 (+ the tile's column span) does.
bool fits_buffer_offsets(const Launch& l) {
    const unsigned long long lim = 1ull << 32;
    const unsigned long long col_span = (unsigned long long)std::max(l.a.nc, 16) * 8ull * 3ull;  // columns of a tile, any limb
    const auto ok = [&](long long rs_words, long long cs_hi_words, unsigned long long rows = 1024ull) {
        const unsigned long long cols = (unsigned long long)(cs_hi_words < 0 ? 0 : cs_hi_words) * 8ull * 16ull;  // ch < 16 columns of a tile
        return rows * (unsigned long long)rs_words * 8ull + cols + col_span < lim;
    };
    // (a PRE2 launch also reads the partner rows, 1024 rows further; the one-workgroup 2048-point column pass, p2 = 6, spans 2048 rows
    //  on every side)
    const unsigned long long rows = l.a.p2 == 6 ? 2048ull : 1024ull;
    return ok(l.a.in_rs, l.a.in_cs_hi, l.a.pre2_map ? (l.a.pre4_stw ? 4096ull : 2048ull) : rows) && ok(l.a.out_rs, l.a.out_cs_hi, rows) && ok(l.a.tw_rs, 0, rows);
}

// the specialised R = 1024 last-pass kernel (LAST1024) is available unless an A/B switch or an ablation run disables it
bool last1024_enabled() {
    static const bool off = ab_env("TF_NTT_NO_LAST1024") != nullptr;
    return !off && ablate_mode() == 0 && !t_small_launch;
}

bool col_enabled() {
    static const bool off = ab_env("TF_NTT_NO_COL") != nullptr;  // A/B switch
    return !off;
}

// ... and its variant that multiplies on store (fast_coset_interpolate)
bool scaled_last1024_enabled() {
    static const bool off = ab_env("TF_NTT_NO_SCALED_LAST1024") != nullptr;  // A/B switch
    return !off;
}

// rows (independent DFTs) per tile for the passes whose columns are whole rows of elements
int rows_per_tile(int P2, int L, long long limit) {
    int nc_max = std::max(1, wg_threads() / P2);
    int T = std::max(1, nc_max / L);
    return (int)std::min<long long>(T, limit);
}

// Last pass of a multi-pass transform: rows (k1, rho) of R contiguous elements; DFT along the row;
// output element k of row (k1, rho) goes to  k1 + N1 * (rho + Q * k)  (digit reversal = natural order).
// A tile is T consecutive k1: its T*L word-columns are contiguous on the OUTPUT side.
// split = N2 > 0 (four-pass transforms, one polynomial per launch): rho = k2 * (Q / N2) + k3 on the input side but
// k2 + N2 * k3 on the output side; the kernel's batch index carries k2 and its rho index carries k3.
// words > 0: word-granular tiles of `words` adjacent output WORDS (whole 128-byte lines) instead of T whole elements; for
// XFieldElement rows (24-byte elements) a tile then starts and ends inside an element (NttPassArgs::wtiles).
Launch plan_transpose_pass(const u64* in, u64* out, long long in_bs, long long out_bs, size_t batch, int a, long long N1,
                           long long Q, int L, long long split = 0, int words = 0, int pre = 1) {
    Launch l{};
    tfk::NttPassArgs& A = l.a;
    const int p2 = a - (pre == 4 ? 2 : (pre == 2 ? 1 : 0)) - 5, P2 = 1 << p2;
    const long long R = 1ll << a;  // elements per row (pre = 2 / 4: 2048 / 4096, transformed as two / four interleaved 1024-point classes)
    int T = rows_per_tile(P2, L, N1);
    {
        // the kernel addresses its loads as uniform 64-bit base + 32-bit per-thread byte offset; the offset spans the
        // tile's T rows, Q * R * L words apart: keep it below 2^32 (only binds for n = 2^31)
        const long long row_words = Q * R * L;
        const long long t_max = ((1ll << 29) - (1ll << 16)) / row_words;
        if (t_max < T) T = (int)std::max<long long>(1, t_max);
        if (t_max < words / L + 2) words = 0;  // rows a word-granular tile can span
    }
    const int nc = words ? words : T * L;
    A.in = in;
    A.out = out;
    A.L = L;
    A.d1 = (u32)Q;
    A.d2 = words ? (u32)((N1 * L + words - 1) / words) : (u32)((N1 + T - 1) / T);
    A.wtiles = words ? 1 : 0;
    A.d01 = (u32)(batch * Q);
    A.ib0 = in_bs;
    A.ib1 = R * L;
    A.ib2 = (long long)T * Q * R * L;  // (ib2 / ob2 / js_i2 are not used by word-granular tiles)
    A.in_cs_hi = Q * R * L;
    A.in_rs = L;
    A.ob0 = out_bs;
    A.ob1 = N1 * L;
    A.ob2 = (long long)T * L;
    A.out_cs_hi = L;
    A.out_rs = N1 * Q * L;
    A.col_limit = (int)(N1 * L);
    A.n_coeffs = -1;
    A.n_out = -1;
    A.js_i1 = N1;
    A.js_i2 = T;
    A.js_c = 1;
    A.js_k = N1 * Q;
    A.xcd_order = (nc * (int)sizeof(u64) < 128 && A.d2 % 16 == 0) ? 2 : 0;  // output segments narrower than a line: pair them
    if (split) {
        const long long N3 = Q / split;
        A.d1 = (u32)N3;
        A.ib0 = N3 * R * L;
        A.ob0 = N1 * L;
        A.ob1 = N1 * split * L;
        A.js_i0 = N1;
        A.js_i1 = N1 * split;
        batch = 1;
    }
    finish_geometry(&l, nc, p2);
    l.tiles = (unsigned)(batch * Q * A.d2);
    if (pre > 1) {
        A.pre2_in_off = 1024 * L;
        A.pre2_out_off = A.out_rs;  // output k = pre k' + q
        A.pre2_js_off = A.js_k;
        A.out_rs *= pre;
        A.js_k *= pre;
        A.pre2_map = (l.tiles % 8 == 0) ? 1 : 2;
        l.tiles *= (unsigned)pre;
    }
    return l;
}

// Single pass (32 <= n <= 1024): a tile is T whole transforms, output in natural order at the same place.
Launch plan_row_pass(const u64* in, u64* out, long long in_bs, long long out_bs, size_t batch, int a, int L) {
    Launch l{};
    tfk::NttPassArgs& A = l.a;
    const int p2 = a - 5, P2 = 1 << p2;
    const int T = rows_per_tile(P2, L, (long long)batch);
    const int nc = T * L;
    A.in = in;
    A.out = out;
    A.L = L;
    A.d1 = 1;
    A.d2 = (u32)((batch + T - 1) / T);
    A.d01 = 1;
    A.ib2 = (long long)T * in_bs;
    A.ob2 = (long long)T * out_bs;
    A.in_cs_hi = in_bs;
    A.out_cs_hi = out_bs;
    A.in_rs = L;
    A.out_rs = L;
    A.col_limit = (int)std::min<size_t>(batch * L, 0x7fffffff);
    A.ps_rs = 1;
    A.ps_col = 0;
    A.n_coeffs = -1;
    A.n_out = -1;
    A.js_k = 1;  // single pass: output element index = k
    A.xcd_order = 0;
    finish_geometry(&l, nc, p2);
    static const bool no_gfast = ab_env("TF_NTT_NO_GFAST") != nullptr;  // A/B switch
    if (p2 >= 1 && !no_gfast) {
        // rows are contiguous: put the lanes along the row (8 * P2-byte pieces become P2 times longer).  Exchange layout
        // idx = k1 * s1 + cc * P2 + g with s1 = 1 (mod 32): a half-wave writes 32 consecutive words and reads
        // g' * s1 + cc * P2 = g' + cc * P2 (mod 32), all different.
        A.gfast = 1;
        A.s2 = 1;
        A.s3 = P2;
        A.s1 = pad_to_residue(P2 * A.cpr, 1);
        l.lds_bytes = size_t(32) * A.s1 * sizeof(u64);
    }
    l.tiles = A.d2;
    return l;
}

// One hipFuncSetAttribute per (kernel instantiation, device) to open the dynamic LDS above 48 KiB.  First use is serialised
// under a lock: a launch of the same function from another host thread while the attribute is being set is not safe (seen as a
// rare failure of the four-threads-one-tree test when a kernel's first use fell inside the threaded section).
std::mutex g_func_attr_mutex;
int ensure_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done_mask) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask.load(std::memory_order_acquire) & bit) return TF_OK;
    std::lock_guard<std::mutex> guard(g_func_attr_mutex);
    if (done_mask.load(std::memory_order_acquire) & bit) return TF_OK;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done_mask.fetch_or(bit, std::memory_order_release);
    return TF_OK;
}

template <bool INV, int SCALE, int MODE, bool LAST1024 = false, bool R1024 = false, bool COL = false, bool PRE2 = false, bool PRE4 = false>
int launch_pass_t(const Launch& l, hipStream_t stream) {
    // one attribute call per (instantiation, device): the kernels use up to the full 160 KiB of dynamic LDS
    static std::atomic<unsigned long long> done_mask{0};
    if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_pass_kernel<INV, SCALE, MODE, LAST1024, R1024, COL, PRE2, PRE4>), (int)(160 * 1024), done_mask)) return rc_attr;
    // the R = 1024 column-pass instantiation stages its inner twiddle table behind the exchange buffer (LAST1024: part of
    // kLast1024LdsBytes already)
    const size_t lds_bytes = l.lds_bytes + ((TF_LDS_TW && !LAST1024 && MODE == 0 && l.a.inner_tw) ? (size_t(1) << l.a.p2) * tfk::kLdsTwStride * sizeof(u64) : 0);
    hipLaunchKernelGGL((tfk::ntt_pass_kernel<INV, SCALE, MODE, LAST1024, R1024, COL, PRE2, PRE4>), dim3(l.tiles), dim3(l.threads), lds_bytes, stream,
                       l.a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// the 2048-point column pass in one workgroup (2048 rows x 8 word-columns, ntt_col2048_kernel)
template <bool INV, int SCALE>
int launch_col2048_t(const Launch& l, hipStream_t stream) {
    static std::atomic<unsigned long long> done_mask{0};
    if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_col2048_kernel<INV, SCALE>), (int)(160 * 1024), done_mask)) return rc_attr;
    const size_t lds_bytes = size_t(32) * tfk::kC8S1 * sizeof(u64);  // (the inner table is staged inside the exchange buffer)
    hipLaunchKernelGGL((tfk::ntt_col2048_kernel<INV, SCALE>), dim3(l.tiles), dim3(512), lds_bytes, stream, l.a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

#ifdef TF_AB_BUILD
// the R = 1024 column pass as a chain of `k` tiles per workgroup with the next tile's loads inside the store phase
// (ntt_col1024_chain_kernel); k from TF_NTT_PERSIST / tf_set_ntt_chain (0 or 1: the one-tile kernel)
std::atomic<int> g_chain{-1};
int chain_tiles() {
    int v = g_chain.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = ab_env("TF_NTT_PERSIST");
        v = e ? std::max(0, atoi(e)) : 0;
        g_chain.store(v, std::memory_order_relaxed);
    }
    return v;
}
template <bool INV>
int launch_chain_t(const Launch& l, int k, hipStream_t stream) {
    static std::atomic<unsigned long long> done_mask{0};
    if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_col1024_chain_kernel<INV>), (int)(160 * 1024), done_mask)) return rc_attr;
    const size_t lds_bytes = l.lds_bytes + size_t(32) * tfk::kLdsTwStride * sizeof(u64);
    unsigned grid = (l.tiles + (unsigned)k - 1) / (unsigned)k;
    grid = (grid + 7u) & ~7u;  // block id mod 8 is the XCD: tile, tile + grid, ... stay on one XCD
    grid = std::min(grid, l.tiles);
    hipLaunchKernelGGL((tfk::ntt_col1024_chain_kernel<INV>), dim3(grid), dim3(512), lds_bytes, stream, l.a, l.tiles, (unsigned)k);
    HIPCHK(hipGetLastError());
    return TF_OK;
}
#else
constexpr int chain_tiles() { return 0; }  // (ntt_col1024_chain_kernel: a measured loss, laboratory build only)
#endif

#ifdef TF_AB_BUILD
unsigned long long* g_dbg_buf = nullptr;  // TF_NTT_ABLATE=3: per-wave phase stamps of the last launch (tf_debug_stamps)
#endif
constexpr size_t kLast1024LdsBytes = (size_t(tfk::kL1024ExchangeWords) + (TF_LDS_TW ? 32 * tfk::kLdsTwStride : 0)) * sizeof(u64);

int launch_pass(const Launch& l, bool inverse, hipStream_t stream) {
    if (l.tiles == 0) return TF_OK;
    if (l.bad_geometry) {
        t_last_error = "NTT planner self-check failed (thread-to-column division is not exact for this geometry)";
        return TF_ERR_HIP;
    }
    const int g_ablate = ablate_mode();
    if (l.a.p2 == 5 && !l.a.inner_tw) {  // the R = 1024 instantiations run lazy networks and rely on the product that follows
        t_last_error = "internal: R = 1024 pass without its inner twiddle table";
        return TF_ERR_HIP;
    }
    // the plain R = 1024 last-pass kernel: the only one that truncates its output and shifts its tiles
    const bool fits = fits_buffer_offsets(l);
    // the constant-geometry column pass (R1024) only with exactly the geometry its immediates assume; anything else is COL's
    const bool std_geo = l.threads == 512 && l.a.nc == tfk::kR1024Nc && l.a.cpr == tfk::kR1024Cpr && l.a.nrounds == tfk::kR1024Rounds &&
                         l.a.s1 == tfk::kR1024S1 && l.a.s2 == tfk::kR1024Cpr && l.a.s3 == 1 && !l.a.gfast;
    const bool plain_last1024 = l.a.p2 == 5 && !l.a.post_tw && !l.a.gfast && !l.a.pre_scale && l.a.n_coeffs < 0 && !l.a.in2 &&
                                (!l.a.post_scale || (inverse && scaled_last1024_enabled())) && last1024_enabled() && fits;
    if (l.a.p2 == 6) {
        // a 2048-point column pass in ONE workgroup of 2048 rows x 8 word-columns (ntt_kernels.h, ntt_col2048_kernel): planned by
        // run_ntt only, with exactly the geometry the kernel's immediates assume
        const bool geo = l.threads == 512 && l.a.nc == tfk::kC8Nc && l.a.cpr == tfk::kC8Cpr && l.a.nrounds == tfk::kC8Rounds && l.a.s1 == tfk::kC8S1 &&
                         l.a.s2 == tfk::kC8Cpr && l.a.s3 == 1 && !l.a.gfast && !l.a.wtiles;
        const bool load_work = l.a.pre_scale || l.a.n_coeffs >= 0;
        if (!geo || !fits || g_ablate || !l.a.post_tw || !l.a.inner_tw || l.a.in2 || l.a.n_out >= 0 || l.a.post_scale || l.a.pre2_map ||
            (load_work && inverse) || (l.a.pre_scale && l.a.ps_col)) {
            t_last_error = "internal: one-workgroup 2048-point column pass outside the shapes it supports";
            return TF_ERR_HIP;
        }
        if (load_work) return launch_col2048_t<false, 1>(l, stream);
        return inverse ? launch_col2048_t<true, 0>(l, stream) : launch_col2048_t<false, 0>(l, stream);
    }
    if (l.a.pre2_map && l.a.pre4_stw) {
        // a 4096-point last pass as four 1024-point classes per tile (ntt_kernels.h, PRE4): forward, plain, planned by run_ntt only
        if (l.a.p2 != 5 || !fits || g_ablate || l.a.in2 || l.a.n_out >= 0 || l.a.gfast || l.a.post_tw || l.a.pre_scale || l.a.n_coeffs >= 0 ||
            l.a.post_scale || inverse || l.a.col_shift0 || l.a.col_shift_i0) {
            t_last_error = "internal: radix-4 last pass (PRE4) launch outside the shapes it supports";
            return TF_ERR_HIP;
        }
#ifdef TF_AB_BUILD
        Launch l2 = l;
        l2.lds_bytes = std::max(l.lds_bytes, kLast1024LdsBytes);
        l2.threads = 512;
        return launch_pass_t<false, 0, 0, true, false, false, false, true>(l2, stream);
#else
        t_last_error = "internal: radix-4 last pass (PRE4) planned in the product build";
        return TF_ERR_HIP;
#endif
    }
    if (l.a.pre2_map) {
        // a 2048-point pass as two 1024-point halves per tile (ntt_kernels.h, PRE2): only planned by run_ntt when all of this holds
        const bool column = l.a.post_tw != nullptr;
        if (l.a.p2 != 5 || !fits || g_ablate || l.a.in2 || l.a.n_out >= 0 || l.a.gfast || (column && !std_geo) ||
            (!column && (l.a.pre_scale || l.a.n_coeffs >= 0)) || (l.a.post_scale && (column || !inverse)) ||
            ((l.a.pre_scale || l.a.n_coeffs >= 0) && inverse)) {
            t_last_error = "internal: two-pass (PRE2) launch outside the shapes it supports";
            return TF_ERR_HIP;
        }
        if (column) {
            if (l.a.pre_scale || l.a.n_coeffs >= 0) return launch_pass_t<false, 1, 0, false, true, false, true>(l, stream);
            return inverse ? launch_pass_t<true, 0, 0, false, true, false, true>(l, stream) : launch_pass_t<false, 0, 0, false, true, false, true>(l, stream);
        }
        Launch l2 = l;
        l2.lds_bytes = std::max(l.lds_bytes, kLast1024LdsBytes);
        l2.threads = 512;
        if (l.a.post_scale) return launch_pass_t<true, 2, 0, true, false, false, true>(l2, stream);
        return inverse ? launch_pass_t<true, 0, 0, true, false, false, true>(l2, stream) : launch_pass_t<false, 0, 0, true, false, false, true>(l2, stream);
    }
    if ((l.a.n_out >= 0 || l.a.col_shift0 || l.a.col_shift_i0) && !plain_last1024) {  // anything else would overrun the caller's buffer
        t_last_error = "internal: truncated output or shifted tiles requested from a pass that does not support them";
        return TF_ERR_HIP;
    }
    if (l.a.pre_scale || l.a.n_coeffs >= 0 || l.a.in2) {
        // work on load: coset scaling, zero padding, or the pointwise product with a second operand (forward or inverse;
        // constant-P2 variant for a forward first pass with R = 1024)
        static const bool no_r1024_scale = ab_env("TF_NTT_NO_R1024") != nullptr;
        if (inverse) return launch_pass_t<true, 1, 0>(l, stream);
        if (l.a.p2 == 5 && l.a.post_tw && !no_r1024_scale && fits && std_geo) return launch_pass_t<false, 1, 0, false, true>(l, stream);
        if (l.a.post_tw && fits && !l.a.gfast && col_enabled()) return launch_pass_t<false, 1, 0, false, false, true>(l, stream);
        return launch_pass_t<false, 1, 0>(l, stream);
    }
    if (l.a.post_scale) {  // coset interpolation: inverse, scale on store
        if (plain_last1024) {  // R = 1024 last pass: the specialised kernel with the multiplication in its fused tail
            Launch l2 = l;
            l2.lds_bytes = std::max(l.lds_bytes, kLast1024LdsBytes);
            l2.threads = 512;
            return launch_pass_t<true, 2, 0, true>(l2, stream);
        }
        return launch_pass_t<true, 2, 0>(l, stream);
    }
    // last pass of a plain transform with R = 1024: specialised kernel (constant P2, stores fused with level 5)
    // (not for single-pass transforms: their stores run along the row as well, which only the gfast roles give -- 1.05 vs 1.20 ms)
    const bool last1024 = l.a.p2 == 5 && !l.a.post_tw && last1024_enabled() && !l.a.gfast && fits;
    static const bool no_r1024 = ab_env("TF_NTT_NO_R1024") != nullptr;  // A/B switch
    const bool r1024 = l.a.p2 == 5 && l.a.post_tw && g_ablate == 0 && !no_r1024 && fits && std_geo;  // column pass with R = 1024
    if (last1024) {  // this instantiation lays its exchange buffer out itself (32 x 289 words, ntt_kernels.h)
        Launch l2 = l;
        l2.lds_bytes = std::max(l.lds_bytes, kLast1024LdsBytes);
        l2.threads = 512;  // 16 column slots x 32, also for tiles of 15 word-columns (XFE)
        return inverse ? launch_pass_t<true, 0, 0, true>(l2, stream) : launch_pass_t<false, 0, 0, true>(l2, stream);
    }
    // any other column pass whose offsets fit: the same treatment with a run-time P2 (COL)
    const bool col = l.a.post_tw && !r1024 && fits && !l.a.gfast && g_ablate == 0 && col_enabled();
#ifdef TF_AB_BUILD
    if (r1024 && chain_tiles() > 1 && l.tiles >= 1024 && !l.a.dbg)
        return inverse ? launch_chain_t<true>(l, chain_tiles(), stream) : launch_chain_t<false>(l, chain_tiles(), stream);
#endif
    if (inverse) return r1024 ? launch_pass_t<true, 0, 0, false, true>(l, stream)
                              : (col ? launch_pass_t<true, 0, 0, false, false, true>(l, stream) : launch_pass_t<true, 0, 0>(l, stream));
    if (r1024) return launch_pass_t<false, 0, 0, false, true>(l, stream);
    if (col) return launch_pass_t<false, 0, 0, false, false, true>(l, stream);
#ifdef TF_AB_BUILD
    if (g_ablate == 1) return launch_pass_t<false, 0, 1>(l, stream);
    if (g_ablate == 2) return launch_pass_t<false, 0, 2>(l, stream);
    if (g_ablate == 3) {
        Launch l2 = l;
        l2.a.dbg = g_dbg_buf;
        return launch_pass_t<false, 0, 3>(l2, stream);
    }
#endif
    return launch_pass_t<false, 0, 0>(l, stream);
}

int check_len(size_t n) {
    if (n != 0 && (n & (n - 1))) return TF_ERR_LEN_NOT_POWER_OF_TWO;  // ntt.rs:137
    if (n > (size_t(1) << 31)) return TF_ERR_LEN_TOO_LARGE;           // ntt.rs:134-139: lengths beyond u32::MAX panic
    return TF_OK;
}

#ifdef TF_AB_BUILD
// n = 32, contiguous transforms: LDS-staged rows (ntt_rows32_kernel)
template <bool INV>
int launch_rows32_t(const tfk::NttRows32Args& a, unsigned grid, hipStream_t stream) {
    static std::atomic<unsigned long long> done_mask{0};
    if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_rows32_kernel<INV>), (int)(160 * 1024), done_mask)) return rc_attr;
    hipLaunchKernelGGL((tfk::ntt_rows32_kernel<INV>), dim3(grid), dim3(512), size_t(256) * 33 * sizeof(u64), stream, a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}
#endif

// n <= 32: wave-private tiles (ntt_rows32w_kernel), a grid-stride walk of two workgroups per CU
template <bool INV, int L, int LOGN>
int launch_rows32w_t(const tfk::NttRows32Args& a, unsigned grid, hipStream_t stream) {
    static std::atomic<unsigned long long> done_mask{0};
    if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_rows32w_kernel<INV, L, LOGN>), (int)(160 * 1024), done_mask)) return rc_attr;
    hipLaunchKernelGGL((tfk::ntt_rows32w_kernel<INV, L, LOGN>), dim3(grid), dim3(256), size_t(4) * 64 * 33 * sizeof(u64), stream, a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}
template <bool INV, int L>
int launch_rows32w_n(const tfk::NttRows32Args& a, unsigned grid, int log_n, hipStream_t stream) {
    switch (log_n) {
        case 1: return launch_rows32w_t<INV, L, 1>(a, grid, stream);
        case 2: return launch_rows32w_t<INV, L, 2>(a, grid, stream);
        case 3: return launch_rows32w_t<INV, L, 3>(a, grid, stream);
        case 4: return launch_rows32w_t<INV, L, 4>(a, grid, stream);
        case 6:
            if constexpr (L == 1) return launch_rows32w_t<INV, 1, 6>(a, grid, stream);
            return TF_ERR_NULL_POINTER;  // (not reached: run_ntt sends XFieldElement transforms of 64 points to the row pass)
        default: return launch_rows32w_t<INV, L, 5>(a, grid, stream);
    }
}

// batch contiguous transforms of n = 2^log_n <= 32 points (BFieldElement: <= 64)
int launch_rows32(const u64* in, u64* out, size_t batch, int L, int log_n, bool inverse, hipStream_t stream) {
#ifdef TF_AB_BUILD
    if (log_n == 5 && ab_env("TF_NTT_ROWS32_WG")) {  // round 2's workgroup-tile kernel (1.21 vs 0.87 ms per 2^28 words, profiles/r05_rows32_ab.txt)
        const size_t per_tile = L == 1 ? 512 : 170;
        const size_t max_grid = size_t(1) << 30;
        for (size_t b0 = 0; b0 < batch; b0 += max_grid * per_tile) {
            const size_t nb = std::min(batch - b0, max_grid * per_tile);
            tfk::NttRows32Args a{};
            a.in = in + b0 * 32 * L;
            a.out = out + b0 * 32 * L;
            a.total_transforms = (long long)nb;
            a.scale = inverse ? gl::mont_inverse(gl::to_mont(32)) : 0;
            a.L = L;
            const unsigned grid = (unsigned)((nb + per_tile - 1) / per_tile);
            int rc = inverse ? launch_rows32_t<true>(a, grid, stream) : launch_rows32_t<false>(a, grid, stream);
            if (rc) return rc;
        }
        return TF_OK;
    }
#endif
    const int cus = device_cus();
    tfk::NttRows32Args a{};
    a.in = in;
    a.out = out;
    a.total_transforms = (long long)batch;
    a.scale = inverse ? gl::mont_inverse(gl::to_mont(u64(1) << log_n)) : 0;
    a.L = L;
    const size_t words = (batch << log_n) * size_t(L), per_tile = L == 1 ? 2048 : 2016;  // words per wave and tile
    const size_t tiles = (words + per_tile - 1) / per_tile, wgs = (tiles + 3) / 4;
    const unsigned grid = (unsigned)std::min(wgs, size_t(cus) * 2);  // two workgroups are resident per CU (LDS); they walk the tiles
    if (L == 1) return inverse ? launch_rows32w_n<true, 1>(a, grid, log_n, stream) : launch_rows32w_n<false, 1>(a, grid, log_n, stream);
    return inverse ? launch_rows32w_n<true, 3>(a, grid, log_n, stream) : launch_rows32w_n<false, 3>(a, grid, log_n, stream);
}

// 2^11 <= n <= 2^14, contiguous BFieldElement transforms: whole transform per workgroup (ntt_block_kernel)
template <int LOGP3, bool INV, int SCALE>
int launch_block_t(const tfk::NttBlockArgs& a, unsigned grid, hipStream_t stream) {
    static std::atomic<unsigned long long> done_mask{0};
    if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_block_kernel<LOGP3, INV, SCALE>), (int)(160 * 1024), done_mask)) return rc_attr;
    constexpr int P3 = 1 << LOGP3;
    const size_t lds_bytes = size_t(8) * (1056 + 32 / P3) * sizeof(u64);  // exchange 1 is the larger of the two layouts
    hipLaunchKernelGGL((tfk::ntt_block_kernel<LOGP3, INV, SCALE>), dim3(grid), dim3(512), lds_bytes, stream, a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

int launch_block(DeviceCtx* ctx, const u64* in, u64* out, long long in_bs, long long out_bs, int log_n, size_t batch, bool inverse,
                 const u64* pre_scale, long long n_coeffs, const u64* post_scale, hipStream_t stream, const u64* in2 = nullptr,
                 long long n_out = -1, int L = 1) {
    tfk::NttBlockArgs a{};
    int rc = get_block_tables(ctx, log_n, inverse, &a.tw1, &a.tw2);
    if (rc) return rc;
    a.in = in;
    a.out = out;
    a.pre_scale = pre_scale;
    a.post_scale = post_scale;
    a.n_coeffs = n_coeffs;
    a.in_bs = in_bs;
    a.out_bs = out_bs;
    a.total_transforms = (long long)batch * L;  // limb transforms (ntt_kernels.h)
    a.in2 = in2;
    a.n_out = n_out;
    a.L = L;
    const int lp3 = log_n - 10, T = 16 >> lp3;
    const unsigned grid = (unsigned)((batch * (size_t)L + T - 1) / T);
    const bool scaled_load = pre_scale || n_coeffs >= 0, scaled_store = post_scale != nullptr;
    if (in2) {  // the inverse transform of a product: second operand on load, truncated store
        switch (lp3) {
            case 1: return launch_block_t<1, true, 3>(a, grid, stream);
            case 2: return launch_block_t<2, true, 3>(a, grid, stream);
            case 3: return launch_block_t<3, true, 3>(a, grid, stream);
            default: return launch_block_t<4, true, 3>(a, grid, stream);
        }
    }
    switch (lp3 * 2 + (inverse ? 1 : 0)) {
        case 2: return scaled_load ? launch_block_t<1, false, 1>(a, grid, stream) : launch_block_t<1, false, 0>(a, grid, stream);
        case 3: return scaled_store ? launch_block_t<1, true, 2>(a, grid, stream) : launch_block_t<1, true, 0>(a, grid, stream);
        case 4: return scaled_load ? launch_block_t<2, false, 1>(a, grid, stream) : launch_block_t<2, false, 0>(a, grid, stream);
        case 5: return scaled_store ? launch_block_t<2, true, 2>(a, grid, stream) : launch_block_t<2, true, 0>(a, grid, stream);
        case 6: return scaled_load ? launch_block_t<3, false, 1>(a, grid, stream) : launch_block_t<3, false, 0>(a, grid, stream);
        case 7: return scaled_store ? launch_block_t<3, true, 2>(a, grid, stream) : launch_block_t<3, true, 0>(a, grid, stream);
        case 8: return scaled_load ? launch_block_t<4, false, 1>(a, grid, stream) : launch_block_t<4, false, 0>(a, grid, stream);
        default: return scaled_store ? launch_block_t<4, true, 2>(a, grid, stream) : launch_block_t<4, true, 0>(a, grid, stream);
    }
}

// (launchers of the latency-shaped kernels: tf_lat.hip)

// Experiment switches for tools/split3.py: looked up on every call only when TF_NTT_EXPERIMENT is set at load time
// (the sweep tool changes them while the process runs); otherwise the planner never touches the environment.
const char* exp_env(const char* name) {
    static const bool enabled = ab_env("TF_NTT_EXPERIMENT") != nullptr;
    return enabled ? ab_env(name) : nullptr;
}

int pass_count(int log_n) {  // global passes of a transform with log_n > 10
    int P = log_n <= 20 ? 2 : (log_n <= 30 ? 3 : 4);
    const int want = g_min_passes.load(std::memory_order_relaxed);  // test hook: deeper plans at small sizes
    if (want > P && want <= 4 && log_n >= 5 * want) P = want;
    return P;
}

// Radix split of a multi-pass plan: a[0..P-1], sum = log_n, every a[i] in [5, 10].
void choose_split(int log_n, int P, int L, int (&a)[4]) {
    {
        // The last pass gets the largest radix it can (R = 1024 whenever possible: the specialised kernel with constant P2
        // and stores fused into level 5, 128-byte output segments); the column passes share the rest evenly, larger first.
        // Measured against the even split (tools/split3.py, 2^28 words per call): 2^15 1.92 vs 2.62 ms, 2^18 2.20 vs 2.48,
        // 2^22 3.18 vs 3.49, 2^24 3.31 vs 4.15, 2^26 3.58 vs 3.99.
        // XFieldElement slices (L = 3) follow the same rule for n = 2^15, 2^20 and n >= 2^23 now that the R = 1024 kernel tiles
        // their rows by whole 128-byte lines (tools/xfe_sweep.sh: 2^15 1.62 vs 1.83 ms, 2^23 2.62 vs 2.71, 2^25 2.71 vs 2.86 per
        // 3 * 2^26 words); the sweep still prefers R = 32 for n < 2^15 and R = 512 for 2^16 .. 2^19 (all within 1 % of R = 1024).
        int last = std::min(10, log_n - 5 * (P - 1));
        // (round 5, profiles/r05_xfe_split2_sweep.txt: 2^13 has no better split than (8, 5) -- 1.61 ms against 1.61-1.75 for every other --
        // it is simply the first length that no longer fits one workgroup's LDS; 2^14 prefers (5, 9): 1.325 vs 1.367 ms)
        if (L == 3 && P == 2 && log_n < 20 && log_n != 15) last = log_n < 14 ? 5 : 9;
        a[P - 1] = last;
        int rest = log_n - last;
        if (P == 3 && last == 10 && log_n <= (L == 3 ? 25 : 23)) {
            // (log_n - 15, 5, 10): an exchange-free radix-32 pass in the middle (tools/split3.py: XFE 2^21 2.26 vs 2.33 ms,
            // 2^23 2.34 vs 2.45, 2^25 2.49 vs 2.51 per 3 * 2^26 words; BFE 2^22 2.90 vs 3.04, 2^23 3.01 vs 3.09 per 2^28)
            a[1] = 5;
            a[0] = rest - 5;
            // BFieldElement 2^21 / 2^22: the radix-32 pass first, (5, log_n - 15, 10) -- since the column passes run lazy networks
            // through buffer addressing (COL) the order matters only there: 2^21 2.49 vs 2.71 ms, 2^22 2.54 vs 2.59 ms per 2^28
            // words; from 2^23 on and for XFieldElement slices both orders measure the same (tools/split3_ab.py)
            if (L == 1 && log_n <= 22) a[0] = 5, a[1] = rest - 5;
        } else {
            for (int i = 0; i + 1 < P; ++i) {
                a[i] = (rest + (P - 1 - i) - 1) / (P - 1 - i);
                rest -= a[i];
            }
        }
        if (const char* e2 = exp_env("TF_NTT_SPLIT2")) {  // experiment: a0 for two-pass plans
            const int x0 = atoi(e2);
            if (P == 2 && x0 >= 5 && x0 <= 10 && log_n - x0 >= 5 && log_n - x0 <= 10) a[0] = x0, a[1] = log_n - x0;
        }
        if (const char* e = exp_env("TF_NTT_SPLIT3")) {  // experiment: "a0,a1" for three-pass plans
            int x0 = 0, x1 = 0;
            if (P == 3 && sscanf(e, "%d,%d", &x0, &x1) == 2 && x0 >= 5 && x0 <= 10 && x1 >= 5 && x1 <= 10 && log_n - x0 - x1 >= 5 &&
                log_n - x0 - x1 <= 10) {
                a[0] = x0, a[1] = x1, a[2] = log_n - x0 - x1;
            }
        }
    }
}

// Can the last pass of an n-point transform truncate its output (LAST1024 kernel: two or three passes, last radix 1024)?
bool can_truncate(size_t n, int L) {
    if (n <= 1024 || n > (size_t(1) << 30)) return false;
    static const bool no_block = ab_env("TF_NTT_NO_BLOCK") != nullptr;
    if (n <= (size_t(1) << 14)) return L == 1 && !no_block && g_min_passes.load(std::memory_order_relaxed) == 0;  // the block kernel truncates (BFE product inverse)
    const int log_n = ilog2(n), P = pass_count(log_n);
    int a[4] = {0, 0, 0, 0};
    choose_split(log_n, P, L, a);
    // ... and that kernel addresses its stores through a buffer resource: 1024 * (n L / 1024) * 8 bytes must fit 32 bits (fits_buffer_offsets)
    return P <= 3 && a[P - 1] == 10 && last1024_enabled() && (unsigned long long)n * L * 8 + (1ull << 20) < (1ull << 32);
}

// Whether run_ntt plans this call with the narrow (256-thread) tiles: too little work to fill the chip with 512-thread ones
// (see wg_threads()); a truncating call keeps the R = 1024 last pass its caller planned for.
bool small_launch_for(size_t n, size_t cosets, size_t batch, int L, long long n_out) {
    const int small_mode = g_small_launch_mode.load(std::memory_order_relaxed);
    static const bool no_small = ab_env("TF_NTT_NO_SMALL_LAUNCH") != nullptr;  // A/B switch
    const bool small_call = (unsigned long long)n * cosets * batch * L <= (1ull << 21);
    return n_out < 0 && !wg_env() && (small_mode == 1 || (small_mode < 0 && small_call && !no_small));
}

// Two-pass plans for 2^21 / 2^22 points (a 2048-point pass = pairs of 1024-point workgroups, ntt_kernels.h PRE2).
std::atomic<int> g_pre2_mode{-1};  // tf_set_ntt_two_pass: -1 automatic (TF_NTT_NO_PRE2 disables), 0 never, 1 whenever supported (pairs), 3 (one-workgroup first pass)
#ifndef TF_C8_DEFAULT
#define TF_C8_DEFAULT 1
#endif
constexpr bool kC8Default = TF_C8_DEFAULT != 0;
bool pre2_plan_ok(int log_n, int L, size_t n, size_t cosets, bool has_in2, long long n_out, bool inverse, bool load_work, bool store_scale) {
    static const bool off = ab_env("TF_NTT_NO_PRE2") != nullptr;  // A/B switch
    const int mode = g_pre2_mode.load(std::memory_order_relaxed);
    if (mode == 0 || (mode < 0 && off)) return false;  // (mode 2: the radix-4 plan where it applies, this one elsewhere)
    if (log_n < 21 || log_n > 22 || cosets != 1 || has_in2 || n_out >= 0) return false;
    if ((load_work && inverse) || (store_scale && !inverse)) return false;          // shapes no caller produces
    if (!last1024_enabled() || ablate_mode() != 0 || wg_threads() != 512) return false;
    if (g_min_passes.load(std::memory_order_relaxed) > 2) return false;
    return (unsigned long long)n * L * 8 + (1ull << 20) < (1ull << 32);              // buffer addressing (fits_buffer_offsets)
}
// 2^22 points as 1024 x 4096 with the 4096-point LAST pass run as four 1024-point classes per tile (ntt_kernels.h, PRE4) -- round 4's
// attempt on BASELINE configs[3]: the first pass of a coset evaluation is then an ordinary 1024-point column pass that scales every
// coefficient once, where the 2048 x 2048 plan scales it in both workgroups of a PRE2 pair.  MEASURED (profiles/r04_c4_plan_ab.txt,
// 64 x 2^22 XFE, per 16-polynomial tile): the first pass does get cheaper, 1 194 -> 981 us, but the radix-4 load stage costs the last
// pass more than that, 752 -> 1 047 us (four reads of every input through the L2 in bursts of 16, 27 more instructions per element):
// 7.9 ms against 7.6 -- a loss, so the product never plans it.  Laboratory build: TF_NTT_PRE4 in the environment makes it the
// automatic plan of fast_coset_evaluate at 2^22, tf_set_ntt_two_pass(2) of every forward 2^22-point transform.
bool pre4_plan_ok(int log_n, int L, size_t n, size_t cosets, bool has_in2, long long n_out, bool inverse, bool load_work, bool store_scale) {
#ifdef TF_AB_BUILD
    static const bool on = ab_env("TF_NTT_PRE4") != nullptr;
    const int mode = g_pre2_mode.load(std::memory_order_relaxed);
    if (mode == 0 || mode == 1 || mode == 3 || (mode < 0 && !(on && load_work))) return false;
    if (log_n != 22 || cosets != 1 || has_in2 || n_out >= 0 || inverse || store_scale) return false;
    if (!last1024_enabled() || ablate_mode() != 0 || wg_threads() != 512) return false;
    if (g_min_passes.load(std::memory_order_relaxed) > 2) return false;
    return (unsigned long long)n * L * 8 + (1ull << 20) < (1ull << 32);              // buffer addressing (fits_buffer_offsets)
#else
    (void)log_n, (void)L, (void)n, (void)cosets, (void)has_in2, (void)n_out, (void)inverse, (void)load_work, (void)store_scale;
    return false;
#endif
}
// Round 6: the FIRST pass of a two-pass plan as ONE workgroup per 2048 x 8 tile (ntt_col2048_kernel): every element is loaded (and,
// in a coset evaluation, scaled) once, where a PRE2 pair does both twice.  2^22 = 2048 (this kernel) x 2048 (the PRE2 last pass);
// 2^21 = 2048 (this kernel) x 1024 (the plain R = 1024 last pass, no pair anywhere).  MEASURED (profiles/r06_c4_cols8_ab.txt, same
// process, same words, ms per 2^28 / 3 * 2^26 words): it wins where the first pass SCALES, at 2^22 -- XFE coset evaluation 1.84 vs 1.94
// (BASELINE configs[3]: 7.40 vs 7.80), BFE 2.53 vs 2.65 -- ties on BFE 2^21 / 2^22 plain transforms and loses 2-9 % on XFE plain
// transforms and everything XFE at 2^21 (its 64-byte segments cost more there than the pair's second read).  So the automatic plan
// takes it for coset evaluations of 2^22 points only; tf_set_ntt_two_pass(3) forces it wherever the two-pass plan applies, (1) forces
// the round-3 pairs.
bool c8_first_pass(int log_n, bool scaled) {
    const int mode = g_pre2_mode.load(std::memory_order_relaxed);
    if (mode == 3) return true;
    if (mode >= 0) return false;
    return kC8Default && log_n == 22 && scaled;
}
void pre2_split(int log_n, int (&a)[4]) { pre2_split(log_n, a, c8_first_pass(log_n, false)); }
void pre2_split(int log_n, int (&a)[4], bool c8) {
    // 2^21: the 2048-point pass last (the first pass of a coset evaluation then scales every coefficient once); 2^22: both
    a[0] = log_n == 22 ? 11 : 10, a[1] = 11, a[2] = a[3] = 0;
    if (c8) a[0] = 11, a[1] = log_n - 11;
    if (const char* e = exp_env("TF_NTT_PRE2_FIRST")) {
        if (log_n == 21 && atoi(e)) a[0] = 11, a[1] = 10;
    }
}

// The transform proper.  in/out are device pointers; in == out for ntt/intt, distinct for coset evaluation
// (then pre_scale != null and rows >= n_coeffs read as zero).  in_bs/out_bs: words per polynomial.
// cosets = C > 1 (forward coset evaluation only, n > 1024): the output has C * n points per polynomial,
// out[j * C + c] = (transform of the coefficients scaled by pre_scale[c * n_coeffs + .])[j] -- the evaluation on the coset of
// order C * n done as C transforms of length n whose outputs interleave (w_{Cn}^(jC + c) = w_{Cn}^c * w_n^j).  The first pass
// reads the coefficients once per c and writes rows (k_1, c); from there on it is the ordinary plan with N_1 * C rows.
int run_ntt(DeviceCtx* ctx, const u64* in, u64* out, long long in_bs, long long out_bs, size_t n, size_t batch, int L,
            bool inverse, const u64* pre_scale, long long n_coeffs, hipStream_t stream, const u64* post_scale,
            size_t cosets, const u64* in2, long long n_out, const u64* coset_offset) {
    // n_out >= 0 (only with can_truncate(n, L)): the last pass stores output elements j < n_out only and out_bs may be
    // n_out * L -- the truncation of fast_multiply without a copy; `in` is then used as work space and clobbered
    if (n == 0 || batch == 0) return TF_OK;
    const int log_n = ilog2(n);
    int rc;
    // batches of short contiguous transforms: wave-private tiles, one coalesced round trip (a handful of them stay on the tiny kernel)
    if (log_n >= 1 && (log_n <= 5 || (log_n == 6 && L == 1)) && !pre_scale && !post_scale && n_coeffs < 0 && !in2 && n_out < 0 &&
        in_bs == (long long)n * L && out_bs == (long long)n * L && (log_n == 5 || (batch << log_n) * size_t(L) >= 4096)) {
        static const bool no_rows32 = ab_env("TF_NTT_NO_ROWS32") != nullptr;  // A/B switch
        if (!no_rows32) return launch_rows32(in, out, batch, L, log_n, inverse, stream);
    }
    if (log_n <= 4) {
        const u64* tw = nullptr;
        rc = get_tiny_table(ctx, log_n, inverse, &tw);
        if (rc) return rc;
        tfk::NttTinyArgs A{};
        A.in = in;
        A.out = out;
        A.tw = tw;
        A.pre_scale = pre_scale;
        A.post_scale = post_scale;
        A.n_coeffs = n_coeffs;
        A.in_bs = in_bs;
        A.out_bs = out_bs;
        A.count = (long long)batch * L;
        A.scale = (inverse && log_n > 0) ? gl::mont_inverse(gl::to_mont(u64(n))) : 0;
        A.log_n = log_n;
        A.L = L;
        const long long blocks = (A.count + 255) / 256;
        hipLaunchKernelGGL(tfk::ntt_tiny_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, A);
        HIPCHK(hipGetLastError());
        return TF_OK;
    }
    // (the latency-shaped kernel also takes the coset scalings of fast_coset_evaluate / fast_coset_interpolate: a 2^10-point coset
    //  evaluation is one launch of 8-element threads, 8 us, instead of one 32-element-thread workgroup, 19 us)
    if (n_out < 0 && cosets == 1 && (!in2 || L == 1) && g_min_passes.load(std::memory_order_relaxed) == 0 && !(pre_scale && inverse) &&
        !(post_scale && !inverse) && lat_wanted(log_n, batch, L))
        return launch_lat(ctx, in, out, in_bs, out_bs, log_n, batch, L, inverse, n_coeffs, in2, stream, nullptr, pre_scale, post_scale);
    if (n_out < 0 && cosets == 1 && (!in2 || L == 1) && g_min_passes.load(std::memory_order_relaxed) == 0 && !(pre_scale && inverse) &&
        !(post_scale && !inverse) && lat2_wanted(log_n, batch, L))
        return launch_lat2(ctx, in, out, in_bs, out_bs, log_n, batch, L, inverse, n_coeffs, in2, stream, pre_scale, post_scale);
    if (log_n <= 10) {
        const u64* inner = nullptr;
        rc = get_inner_table(ctx, log_n, inverse, inverse ? log_n : 0, &inner);
        if (rc) return rc;
        // 2^31 columns limit per launch: split huge batches
        const size_t max_batch = size_t(1) << 24;
        for (size_t b0 = 0; b0 < batch; b0 += max_batch) {
            const size_t nb = std::min(max_batch, batch - b0);
            Launch l = plan_row_pass(in + b0 * in_bs, out + b0 * out_bs, in_bs, out_bs, nb, log_n, L);
            l.a.inner_tw = inner;
            l.a.pre_scale = pre_scale;
            l.a.post_scale = post_scale;
            l.a.n_coeffs = n_coeffs;
            l.a.in2 = in2 ? in2 + b0 * in_bs : nullptr;
            rc = launch_pass(l, inverse, stream);
            if (rc) return rc;
        }
        return TF_OK;
    }
    {
        static const bool no_block = ab_env("TF_NTT_NO_BLOCK") != nullptr;  // A/B switch
        const bool product_inverse = in2 && inverse && !pre_scale && !post_scale && n_coeffs < 0;
        // XFieldElement slices take the same kernel as three limb transforms per slice with element stride 3 (round 2;
        // TF_NTT_NO_XFE_BLOCK restores the two-pass plan for an A/B run): one HBM pass instead of two
        static const bool no_xfe_block = ab_env("TF_NTT_NO_XFE_BLOCK") != nullptr;
        // (2^11 and 2^12 only: 1.09 vs 1.46 and 1.33 vs 1.42 ms per 3 * 2^26 words; at 2^13 / 2^14 the limbs of a slice sit in
        // different workgroups and the 24-byte element stride costs more than the second pass saves: 1.52 vs 1.42, 1.78 vs 1.33)
        if (!no_block && log_n >= 11 && log_n <= (L == 1 ? 14 : 12) && (L == 1 || (L == 3 && !no_xfe_block)) && cosets == 1 &&
            g_min_passes.load(std::memory_order_relaxed) == 0 && batch < (size_t(1) << 29)) {
            if (product_inverse && L == 1)
                return launch_block(ctx, in, out, in_bs, out_bs, log_n, batch, true, nullptr, -1, nullptr, stream, in2, n_out);
            if (!in2 && n_out < 0 && !((pre_scale || n_coeffs >= 0) && inverse) && !(post_scale && !inverse))
                return launch_block(ctx, in, out, in_bs, out_bs, log_n, batch, inverse, pre_scale, n_coeffs, post_scale, stream, nullptr, -1, L);
        }
    }
    // multi-pass: n = N_1 * ... * N_P, every N_i = 2^(a_i) <= 1024.  Passes 1 .. P-1 are column passes (DFT over digit i,
    // inter-pass twiddle, same position in and out); the last pass transforms the contiguous rows of N_P elements and
    // scatters output digit k_P to  k_1 + N_1 k_2 + ... + N_1..N_{P-1} k_P  (natural order).
    struct SmallLaunchScope {  // see wg_threads(); a truncating call keeps the R = 1024 last pass its caller planned for
        SmallLaunchScope(bool on) { t_small_launch = on; }
        ~SmallLaunchScope() { t_small_launch = false; }
    };
    SmallLaunchScope small_scope(small_launch_for(n, cosets, batch, L, n_out));
    int a[4] = {0, 0, 0, 0};
    int P = pass_count(log_n);
    choose_split(log_n, P, L, a);
    // 2^21 and 2^22 points in TWO passes: a 2048-point pass runs as pairs of 1024-point workgroups that share their input
    // (ntt_kernels.h, PRE2; a[i] = 11 below).  Plain transforms, coset evaluation (forward) and coset interpolation (inverse).
    bool pre2[4] = {false, false, false, false};
    bool c8 = false;         // the first pass is the one-workgroup 2048-point column pass (ntt_col2048_kernel)
    bool pre4_last = false;  // the last pass is a 4096-point one in four classes (a[P - 1] = 12)
    const u64* pre4_stw = nullptr;
    if (pre4_plan_ok(log_n, L, n, cosets, in2 != nullptr, n_out, inverse, pre_scale != nullptr || n_coeffs >= 0, post_scale != nullptr)) {
        P = 2;
        a[0] = 10, a[1] = 12, a[2] = a[3] = 0;
        pre4_last = true;
    } else if (pre2_plan_ok(log_n, L, n, cosets, in2 != nullptr, n_out, inverse, pre_scale != nullptr || n_coeffs >= 0, post_scale != nullptr)) {
        P = 2;
        // (a scaling first pass on the one-workgroup kernel needs the offset itself, for the inter-pass table with offset^b folded in)
        c8 = c8_first_pass(log_n, pre_scale != nullptr) && (!pre_scale || coset_offset);
        pre2_split(log_n, a, c8);
        pre2[0] = a[0] == 11 && !c8, pre2[1] = a[1] == 11;
    }
    const u64* inner[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < P; ++i) {
        const bool p4 = pre4_last && i == P - 1;
        rc = get_inner_table(ctx, (pre2[i] || p4) ? 10 : a[i], inverse, (i == P - 1 && inverse) ? log_n : 0, &inner[i], p4 ? 4 : (pre2[i] ? 2 : 1),
                             p4 ? &pre4_stw : nullptr);  // n^-1 rides on the last pass
        if (rc) return rc;
    }
    const u64* post[3] = {nullptr, nullptr, nullptr};
    bool post_temp[3] = {false, false, false};
    const u64* post_u = nullptr;  // the plain table beside a scaled post[0] (one-workgroup scaling first pass)
    bool post_u_temp = false;
    auto release_tables = [&]() {
        for (int i = 0; i < 3; ++i)
            if (post_temp[i] && post[i]) (void)hipFreeAsync(const_cast<u64*>(post[i]), stream);
        if (post_u_temp && post_u) (void)hipFreeAsync(const_cast<u64*>(post_u), stream);
    };
    {
        int rest = log_n;
        for (int i = 0; i + 1 < P; ++i) {
            if (i == 0 && c8 && pre_scale) {
                // the scaled table for every thread's own row, the plain one for the rows all threads share (ntt_col2048_kernel, TF_C8_UV)
                rc = get_scaled_post_table(ctx, rest, a[i], *coset_offset, stream, &post[i], &post_temp[i]);
                if (!rc) rc = get_post_table(ctx, rest, a[i], inverse, stream, &post_u, &post_u_temp);
            } else {
                rc = get_post_table(ctx, rest, a[i], inverse, stream, &post[i], &post_temp[i]);
            }
            if (rc) {
                release_tables();
                return rc;
            }
            rest -= a[i];
        }
    }
    read_env();
    static const bool no_col_shift = ab_env("TF_NTT_NO_COL_SHIFT") != nullptr;  // A/B switch
    const size_t poly_bytes = n * cosets * size_t(L) * sizeof(u64);
    size_t tb = std::max<size_t>(1, g_tile_bytes / poly_bytes);
    tb = std::min(tb, batch);
    // pipelined tiles (g_pipe): tile t runs on side stream t % K with scratch tile t % K; the caller's stream forks into
    // the side streams before the first tile and joins them after the last (event edges only, no host synchronisation)
    const size_t ntiles = (batch + tb - 1) / tb;
    tb = (batch + ntiles - 1) / ntiles;  // equal tiles: 64 polynomials at 21 per tile are 4 x 16, not 21 + 21 + 21 + 1
    // Automatic (round 6, profiles/r06_pipe_tiles.txt): two streams for the two-pass plans of 2^21 / 2^22 points -- their first pass is bound
    // by the memory pipe (0.68-0.76 of the issue slots busy) and their last pass by the vector ALU (0.80), so tile t + 1's first pass beside
    // tile t's last pass gains 2.6-3.4 % (configs[3] 7.27-7.36 -> 7.06-7.08 ms, 64 x 2^22 XFE ntt 6.67 -> 6.49) -- and one stream for
    // everything else (1024 x 2^20 BFE: 7.20-7.30 vs 7.24-7.32 ms, both passes VALU-bound).
    int pipe = g_pipe.load(std::memory_order_relaxed);
    if (pipe <= 0) pipe = (P == 2 && log_n >= 21) ? 2 : 1;
    int K = (P < 4) ? (int)std::min<size_t>((size_t)pipe, ntiles) : 1;
    hipStream_t side[kMaxPipe] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxPipe] = {nullptr, nullptr, nullptr, nullptr};
    if (K > 1) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (int i = 0; i < K; ++i) {
            if (!ctx->side[i] && hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                ctx->side[i] = nullptr;
                K = 1;  // no side streams: the plain one-stream plan
                break;
            }
            side[i] = ctx->side[i];
        }
    }
    u64* scratch = nullptr;
    DeviceCtx::ScratchBlock sblk;
    {
        rc = scratch_acquire(ctx, size_t(K) * tb * poly_bytes, stream, &sblk);
        if (rc) {
            release_tables();
            return rc;
        }
        scratch = sblk.p;
    }
    if (K > 1) {
        hipError_t e = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
        for (int i = 0; i < K && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ev_fork, stream);
        for (int i = 0; i < K && e == hipSuccess; ++i) e = hipStreamWaitEvent(side[i], ev_fork, 0);
        if (e != hipSuccess) {
            // nothing has been enqueued on the side streams that touches the scratch: release and report
            if (ev_fork) (void)hipEventDestroy(ev_fork);
            for (int i = 0; i < K; ++i)
                if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
            scratch_release(ctx, sblk, stream);
            release_tables();
            return hip_fail(e, "fork into the tile streams", __FILE__, __LINE__);
        }
    }
    hipStream_t const caller_stream = stream;
    u64* const scratch_base = scratch;
    const long long sbs = (long long)(n * cosets) * L;  // scratch batch stride
    long long N[4];
    for (int i = 0; i < 4; ++i) N[i] = 1ll << a[i];
    size_t tile_no = 0;
    for (size_t b0 = 0; b0 < batch && rc == TF_OK; b0 += tb, ++tile_no) {
        const size_t nb = std::min(tb, batch - b0);
        if (K > 1) {
            stream = side[tile_no % K];
            scratch = scratch_base + (tile_no % K) * tb * (size_t)sbs;
        }
        const u64* tin = in + (long long)b0 * in_bs;
        u64* tout = out + (long long)b0 * out_bs;
        // column passes: the first reads the caller's input, the last writes the scratch tile, the ones between work
        // in place on the output
        const u64* src = tin;
        long long src_bs = in_bs;
        long long outer = 1, B = (long long)n;
        for (int i = 0; i + 1 < P && rc == TF_OK; ++i) {
            B >>= a[i];
            const bool to_scratch = i == P - 2;
            // a truncated output (n_out >= 0) is smaller than the transform: the middle passes then work in place on the
            // INPUT, which the caller gives up (fast_multiply's temporary)
            u64* dst = to_scratch ? scratch : (n_out >= 0 ? const_cast<u64*>(tin) : tout);
            const long long dst_bs = to_scratch ? sbs : (n_out >= 0 ? in_bs : out_bs);
            Launch p = plan_column_pass(src, dst, src_bs, dst_bs, nb, (i == 0) ? (long long)cosets : outer, a[i], B, L, pre2[i]);
            p.a.inner_tw = inner[i];
            p.a.post_tw = post[i];
            p.a.post_tw_u = (i == 0 && post_u) ? post_u : nullptr;
            if (pre2[i]) {
                // partner coefficients are n / 2 apart: offset^(n/2) is word n / 2 of the scale table when the polynomial is that long
                p.a.pre2_cp = pre_scale ? pre_scale + ((long long)(n / 2) < n_coeffs ? (long long)(n / 2) : 0) : nullptr;
            }
            if (i == 0 && src != dst) p.a.nt = g_nt.load(std::memory_order_relaxed) & 1;  // the caller's input is read once
            if (i == 0) {
                p.a.pre_scale = pre_scale;
                if (c8 && pre_scale) p.a.ps_col = 0;  // the column part offset^b of the scaling comes with the inter-pass table
                p.a.n_coeffs = n_coeffs;
                p.a.in2 = in2 ? in2 + (long long)b0 * in_bs : nullptr;
                if (cosets > 1) {  // "outer" index = coset c: same input for every c, output row (k_1, c), scale table c
                    p.a.ib1 = 0;
                    p.a.ob1 = B * L;
                    p.a.out_rs = (long long)cosets * B * L;
                    p.a.ps_i1 = n_coeffs;
                }
                outer = (long long)cosets;
            }
            rc = launch_pass(p, inverse, stream);
            src = dst;
            src_bs = dst_bs;
            outer <<= a[i];
        }
        if (rc) break;
        if (P < 4) {
            // XFieldElement rows through the R = 1024 kernel: word-granular tiles (whole 128-byte lines on the output side)
            static const bool no_words16 = ab_env("TF_NTT_NO_WORDS16") != nullptr;  // A/B switch
            const bool plain1024 = (a[P - 1] == 10 || pre2[P - 1] || pre4_last) && last1024_enabled() && (!post_scale || (inverse && scaled_last1024_enabled() && log_n <= 28));
            // (the other last-pass kernels too: all their thread slots as word-columns, e.g. 32 words = 256 bytes for R = 512)
            int words = 0;
            if (L == 3 && !no_words16) {
                if (plain1024) {
                    words = 16;
                } else {  // share the N_1 * L words of a row evenly among the fewest tiles, in whole lines
                    const long long tot = N[0] * (long long)cosets * L, wmax = std::max(16, wg_threads() >> (a[P - 1] - 5));
                    const long long tiles_per_row = (tot + wmax - 1) / wmax;
                    words = (int)((((tot + tiles_per_row - 1) / tiles_per_row) + 15) / 16 * 16);
                }
            }
            Launch pl = plan_transpose_pass(scratch, tout, sbs, out_bs, nb, a[P - 1], N[0] * (long long)cosets, P == 3 ? N[1] : 1, L, 0,
                                            words, pre4_last ? 4 : (pre2[P - 1] ? 2 : 1));
            pl.a.inner_tw = inner[P - 1];
            pl.a.pre4_stw = pre4_last ? pre4_stw : nullptr;
            pl.a.post_scale = post_scale;
            pl.a.n_out = n_out;
            pl.a.nt = g_nt.load(std::memory_order_relaxed) & 2;  // the result is written once
            if (plain1024 && !pre4_last && pl.a.nc == 16 && !no_col_shift && (unsigned long long)n * L * sizeof(u64) < (1ull << 32)) {
                // The R = 1024 last pass stores 128-byte segments of 16 adjacent output words.  When the output of batch entry
                // b does not start on a cache line (a truncated product: stride n_out = na + nb - 1 words, or a caller's
                // unaligned pointer) every segment would straddle two lines written by workgroups on different XCDs: shift
                // the tile boundaries of entry b by s = (word address of its first output) mod 16 columns so that they fall
                // on lines again; the first tile of a row wraps around to the row's last s columns.  (n * L * 8 < 2^32: the
                // wrapped lanes' 32-bit byte offset spans the whole transform.)  Measured, tools/trunc_align.py: 256 products
                // of 2^19 x 2^19 7.29 -> 7.05 ms, 1024 of 2^17 x 2^17 7.19 -> 6.66, 64 of 2^21 x 2^21 10.25 -> 9.65.
                pl.a.col_shift0 = (int)((reinterpret_cast<uintptr_t>(tout) / sizeof(u64)) & 15);
                pl.a.col_shift_i0 = (int)(out_bs & 15);
                pl.a.col_wrap = pl.a.col_limit;
            }
            rc = launch_pass(pl, inverse, stream);
        } else {
            for (size_t b = 0; b < nb && rc == TF_OK; ++b) {
                Launch pl = plan_transpose_pass(scratch + (long long)b * sbs, tout + (long long)b * out_bs, sbs, out_bs, 1, a[3], N[0],
                                                N[1] * N[2], L, N[1]);
                pl.a.inner_tw = inner[3];
                pl.a.post_scale = post_scale;
                rc = launch_pass(pl, inverse, stream);
            }
        }
    }
    stream = caller_stream;
    scratch = scratch_base;
    if (K > 1) {  // join: whatever was enqueued (also after a failed launch) finishes before the caller's stream goes on
        hipError_t je = hipSuccess;
        for (int i = 0; i < K; ++i) {
            hipError_t e1 = hipEventRecord(ev_join[i], side[i]);
            if (e1 == hipSuccess) e1 = hipStreamWaitEvent(stream, ev_join[i], 0);
            if (e1 != hipSuccess) je = e1;
        }
        (void)hipEventDestroy(ev_fork);
        for (int i = 0; i < K; ++i) (void)hipEventDestroy(ev_join[i]);
        if (je != hipSuccess) {
            (void)hipDeviceSynchronize();  // cannot order the release after the side streams any other way
            scratch_release(ctx, sblk, stream);
            release_tables();
            return hip_fail(je, "join of the tile streams", __FILE__, __LINE__);
        }
    }
    scratch_release(ctx, sblk, stream);
    release_tables();
    if (rc) return rc;
    return TF_OK;
}

int ntt_dev(u64* d_x, size_t n, size_t batch, int L, int inverse, void* stream) {
    int rc = check_len(n);
    if (rc) return rc;
    if (n <= 1 || batch == 0) return TF_OK;  // ntt.rs:170-173 ; length 1 is the identity (n^-1 = 1)
    if (!d_x) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    return run_ntt(ctx, d_x, d_x, (long long)n * L, (long long)n * L, n, batch, L, inverse != 0, nullptr, -1,
                   static_cast<hipStream_t>(stream));
}

int coset_eval_dev(const u64* d_coeffs, size_t n_coeffs, u64 offset_raw, u64* d_out, size_t order, size_t batch, int L,
                   void* stream) {
    if (n_coeffs > order) return TF_ERR_ORDER_NOT_ABOVE_DEGREE;  // polynomial.rs:1388-1392
    int rc = check_len(order);
    if (rc) return rc;
    if (order == 0 || batch == 0) return TF_OK;
    if (!d_out || (n_coeffs && !d_coeffs)) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_coeffs == 0) {  // the zero polynomial evaluates to zero everywhere
        HIPCHK(hipMemsetAsync(d_out, 0, order * batch * size_t(L) * sizeof(u64), s));
        return TF_OK;
    }
    const u64* pw = nullptr;
    bool temp = false;
    // Blown-up evaluation (n_coeffs <= order / 2, the low-degree-extension shape): when the padded coefficient length
    // needs one global pass fewer than the order (len <= 2^20 < order), evaluate on the order / len cosets of the
    // subgroup of size len instead of transforming zeros.  Same values: exact arithmetic.  Measured (tools/lde_shapes.py):
    // 2^18 -> 2^21 2.94 vs 3.47 ms, 2^20 -> 2^23 2.80 vs 3.71 ms per 2^28 points; with equal pass counts the plain plan
    // wins because its first pass skips the zero rows, so it stays the default there.
    size_t len = 1;
    while (len < n_coeffs) len <<= 1;
    static const bool no_split = ab_env("TF_COSET_EVAL_NO_SPLIT") != nullptr;  // A/B switch
    // ADVICE r3 asked whether the comparison should use the EFFECTIVE pass count of the plain plan (orders 2^21 / 2^22 run in two
    // passes under PRE2, so the split then saves no pass).  Built and measured (profiles/r04_lde_split_rule.txt, 2^28 words per
    // call): the split still wins where it matters -- BFE 2^20 -> 2^22 2.05 vs 2.48 ms, 2^19 -> 2^22 2.03 vs 2.40, 2^20 -> 2^21
    // 2.07 vs 2.17; XFE 2^20 -> 2^22 1.01 vs 1.13 -- because its first pass reads n_coeffs rows instead of `order` and none of its
    // passes pays the PRE2 pair's second read and second scaling; only XFE 2^18 / 2^19 -> 2^21 is 3 % faster on the plain plan.
    // The round-3 rule (nominal pass counts) therefore stays; the other one is a laboratory switch.
    static const bool effective_rule = ab_env("TF_COSET_EVAL_EFFECTIVE_PASSES") != nullptr;
    int plain_passes = pass_count(ilog2(order));
    if (effective_rule && !small_launch_for(order, 1, batch, L, -1)) {
        // (wg_threads() is 512 outside a small-launch scope, which is the geometry the big call will be planned with)
        if (pre2_plan_ok(ilog2(order), L, order, 1, false, -1, false, true, false)) plain_passes = 2;
    }
    // (at most kMaxCosetSplit cosets: one scale table of n_coeffs words per coset is built and, while it fits the budget, cached)
    if (!no_split && len > 1024 && len < order && order / len <= kMaxCosetSplit && order <= (size_t(1) << 30) &&
        pass_count(ilog2(len)) < plain_passes) {
        const size_t cosets = order / len;
        rc = get_pow_table(ctx, offset_raw, n_coeffs, s, &pw, &temp, cosets, ilog2(order));
        if (rc) return rc;
        rc = run_ntt(ctx, d_coeffs, d_out, (long long)n_coeffs * L, (long long)order * L, len, batch, L, false, pw,
                     (long long)n_coeffs, s, nullptr, cosets);
    } else {
        rc = get_pow_table(ctx, offset_raw, n_coeffs, s, &pw, &temp);
        if (rc) return rc;
        rc = run_ntt(ctx, d_coeffs, d_out, (long long)n_coeffs * L, (long long)order * L, order, batch, L, false, pw,
                     (long long)n_coeffs, s, nullptr, 1, nullptr, -1, &offset_raw);
    }
    release_pow_table(ctx, pw, temp, s);
    return rc;
}


}  // namespace tfi
