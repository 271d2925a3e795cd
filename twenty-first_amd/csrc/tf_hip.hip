// tf_hip.hip -- host side of libtf_hip.so: C ABI (include/tf_hip.h), pass planner, table cache.
//
// One process drives one or more MI355X devices; all state is per HIP device and guarded by a mutex,
// kernels are enqueued on the caller's stream, and nothing here synchronises the device except the
// one-off construction of a twiddle table.  There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/tf_hip.h"
#include "gl64.h"
#include "ntt_kernels.h"
#include "tip5_kernels.h"
#include "poly_kernels.h"

using gl::u32;
using gl::u64;

namespace {

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
#define HIPCHK(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------ field helpers (host)
// w_n = 7^((p-1)/n): equals every entry of PRIMITIVE_ROOTS (b_field_element.rs:43-78).
u64 root_of_unity_mont(int log_n) { return gl::mont_pow(gl::to_mont(7), (gl::P - 1) >> log_n); }

int ilog2(size_t v) {
    int l = 0;
    while ((size_t(1) << l) < v) ++l;
    return l;
}

// ------------------------------------------------------------------------------------ Tip5 constants
// ROUND_CONSTANTS, tip5/mod.rs:68-149 (canonical values; converted to Montgomery form at upload).
const u64 kRoundConstants[80] = {
    13630775303355457758ULL, 16896927574093233874ULL, 10379449653650130495ULL, 1965408364413093495ULL,
    15232538947090185111ULL, 15892634398091747074ULL, 3989134140024871768ULL,  2851411912127730865ULL,
    8709136439293758776ULL,  3694858669662939734ULL,  12692440244315327141ULL, 10722316166358076749ULL,
    12745429320441639448ULL, 17932424223723990421ULL, 7558102534867937463ULL,  15551047435855531404ULL,
    17532528648579384106ULL, 5216785850422679555ULL,  15418071332095031847ULL, 11921929762955146258ULL,
    9738718993677019874ULL,  3464580399432997147ULL,  13408434769117164050ULL, 264428218649616431ULL,
    4436247869008081381ULL,  4063129435850804221ULL,  2865073155741120117ULL,  5749834437609765994ULL,
    6804196764189408435ULL,  17060469201292988508ULL, 9475383556737206708ULL,  12876344085611465020ULL,
    13835756199368269249ULL, 1648753455944344172ULL,  9836124473569258483ULL,  12867641597107932229ULL,
    11254152636692960595ULL, 16550832737139861108ULL, 11861573970480733262ULL, 1256660473588673495ULL,
    13879506000676455136ULL, 10564103842682358721ULL, 16142842524796397521ULL, 3287098591948630584ULL,
    685911471061284805ULL,   5285298776918878023ULL,  18310953571768047354ULL, 3142266350630002035ULL,
    549990724933663297ULL,   4901984846118077401ULL,  11458643033696775769ULL, 8706785264119212710ULL,
    12521758138015724072ULL, 11877914062416978196ULL, 11333318251134523752ULL, 3933899631278608623ULL,
    16635128972021157924ULL, 10291337173108950450ULL, 4142107155024199350ULL,  16973934533787743537ULL,
    11068111539125175221ULL, 17546769694830203606ULL, 5315217744825068993ULL,  4609594252909613081ULL,
    3350107164315270407ULL,  17715942834299349177ULL, 9600609149219873996ULL,  12894357635820003949ULL,
    4597649658040514631ULL,  7735563950920491847ULL,  1663379455870887181ULL,  13889298103638829706ULL,
    7375530351220884434ULL,  3502022433285269151ULL,  9231805330431056952ULL,  9252272755288523725ULL,
    10014268662326746219ULL, 15565031632950843234ULL, 1209725273521819323ULL,  6024642864597845108ULL,
};

// ------------------------------------------------------------------------------------ per-device context
struct DeviceCtx {
    std::mutex mu;
    std::map<u64, u64*> tables;           // twiddle tables, never freed while the process lives
    std::map<std::pair<u64, u64>, u64*> pow_tables;  // (offset_raw, n) -> offset^j table
    bool tip5_ready = false;               // guarded by mu
    std::atomic<bool> pool_ready{false};  // double-checked under mu
    hipMemPool_t pool = nullptr;           // the library's stream-ordered temporaries (current_ctx); written once before pool_ready
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};  // pipelined tiles (run_ntt); created on first use under mu
    struct ScratchBlock {
        u64* p = nullptr;
        size_t bytes = 0;
        hipEvent_t ready = nullptr;  // recorded on the last user's stream when it gave the block back
    };
    std::vector<ScratchBlock> scratch_free;  // work space of the multi-pass transforms (guarded by mu), see scratch_acquire
    size_t scratch_bytes = 0;                // bytes held by blocks in scratch_free
    size_t cached_post_bytes = 0;          // inter-pass twiddle tables kept for the life of the process (guarded by mu)
    size_t cached_pow_bytes = 0;           // coset power tables kept for the life of the process (guarded by mu)
};
// What the caches may pin for the life of the process; a table that does not fit becomes a stream-ordered temporary
// that is rebuilt per call and released after the pass that reads it.
constexpr size_t kPostCacheBudget = size_t(4) << 30;   // inter-pass twiddles (2^20: 8 MiB, 2^28: 2 GiB)
constexpr size_t kPowCacheBudget = size_t(1) << 30;    // offset^j tables (n_coeffs words each)

constexpr int kMaxDevices = 64;
DeviceCtx g_ctx[kMaxDevices];

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
            if (hipMemPoolCreate(&pool, &props) != hipSuccess) {
                pool = nullptr;
                (void)hipGetLastError();
                (void)hipDeviceGetDefaultMemPool(&pool, dev);  // no pool of our own: the default one, same settings
            }
            if (pool) {
                uint64_t thr = UINT64_MAX;
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
                int follow = 0;
                if (const char* e = getenv("TF_POOL_REUSE_FOLLOW_EVENTS")) follow = atoi(e);  // (to reproduce the failure)
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
    bool keep;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        keep = ctx->scratch_bytes + blk.bytes <= kScratchCacheBytes && ctx->scratch_free.size() < 16;
        if (keep) {
            ctx->scratch_free.push_back(blk);
            ctx->scratch_bytes += blk.bytes;
        }
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
constexpr int kMaxPipe = 4;
std::atomic<int> g_nt{-1};  // TF_NTT_NT / tf_set_ntt_nt: bit 0 non-temporal input loads (first pass), bit 1 non-temporal output stores (last pass)
std::once_flag g_env_once;
void read_env() {
    std::call_once(g_env_once, [] {
        if (g_tile_bytes == 0) {
            const char* s = getenv("TF_NTT_TILE_BYTES");
            g_tile_bytes = s ? strtoull(s, nullptr, 10) : (size_t(2048) << 20);
            if (g_tile_bytes == 0) g_tile_bytes = size_t(2048) << 20;
        }
        if (g_pipe.load() == 0) {
            const char* s = getenv("TF_NTT_PIPE");
            const int k = s ? atoi(s) : 1;
            g_pipe.store(std::min(std::max(k, 1), kMaxPipe));
        }
        if (g_nt.load() < 0) {
            const char* s = getenv("TF_NTT_NT");
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

enum : u64 { TAG_INNER = 1, TAG_POST = 2, TAG_TINY = 3, TAG_BLOCK1 = 4, TAG_BLOCK2 = 5, TAG_LAT = 6 };
u64 make_key(u64 tag, u64 a, u64 b, u64 c, u64 d) { return (tag << 56) | (a << 40) | (b << 24) | (c << 8) | d; }

// inner[g*32 + k1] = w_R^(+-g*k1) * (scale_log_n ? n^-1 : 1),  R = 32 << p2
// pre2 (a = 10 only): a second table follows the first, inner[1024 + g*32 + k1] = w_2048^(+-g (2 k1 + 1)) * scale -- the inner
// twiddles of the odd half of a 2048-point pass (ntt_kernels.h, PRE2)
int get_inner_table(DeviceCtx* ctx, int a, bool inverse, int scale_log_n, const u64** out, bool pre2 = false) {
    const int p2 = a - 5;
    if (p2 == 0 && scale_log_n == 0) {
        *out = nullptr;
        return TF_OK;
    }
    const u64 key = make_key(TAG_INNER, a, inverse, scale_log_n, pre2 ? 1 : 0);
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = it->second;
        return TF_OK;
    }
    const int P2 = 1 << p2;
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
    if (pre2) {
        u64 w2 = root_of_unity_mont(a + 1);
        if (inverse) w2 = gl::mont_inverse(w2);
        t.resize(size_t(P2) * 64);
        u64 w2g = gl::ONE;  // w_{2R}^g
        for (int g = 0; g < P2; ++g) {
            for (int k = 0; k < 32; ++k) t[size_t(P2) * 32 + size_t(g) * 32 + k] = gl::mont_mul(t[size_t(g) * 32 + k], w2g);
            w2g = gl::mont_mul(w2g, w2);
        }
    }
    u64* d = nullptr;
    int rc = upload_table(t, &d);
    if (rc) return rc;
    ctx->tables[key] = d;
    *out = d;
    return TF_OK;
}

// T[k*B + b] = w_M^(+-k*b), k < R = 2^a, b < B = M / R
// Inter-pass twiddles T[k * B + b] = w_M^(k * b), M = 2^log_m = R * B.  Tables up to 2^28 entries (2 GiB) are built once
// and cached; larger ones (single transforms of 2^29 .. 2^31 points) are stream-ordered temporaries: *temp = true and
// the caller releases them with hipFreeAsync after the pass that reads them.
constexpr int kMaxCachedPostLog = 28;
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
        if (ctx->cached_post_bytes + (sizeof(u64) << log_m) > kPostCacheBudget) *temp = true;  // over budget: temporary
    }
    u64 w = root_of_unity_mont(log_m);
    if (inverse) w = gl::mont_inverse(w);
    int h = 0;
    std::vector<u64> hi, lo;
    split_powers(w, log_m, &h, &hi, &lo);
    u64 *d_hi = nullptr, *d_lo = nullptr, *d = nullptr;
    int rc = upload_table(hi, &d_hi);
    if (rc) return rc;
    rc = upload_table(lo, &d_lo);
    if (rc) {
        (void)hipFree(d_hi);
        return rc;
    }
    const long long M = 1ll << log_m, R = 1ll << a, B = M / R;
    if (*temp) {
        lk.unlock();
        hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&d), size_t(M) * sizeof(u64), stream);
        if (e != hipSuccess) {
            (void)hipFree(d_hi);
            (void)hipFree(d_lo);
            return hip_fail(e, "pool_malloc_async(twiddle table)", __FILE__, __LINE__);
        }
    } else {
        hipError_t e = hipMalloc(&d, size_t(M) * sizeof(u64));
        if (e != hipSuccess) {
            (void)hipFree(d_hi);
            (void)hipFree(d_lo);
            return hip_fail(e, "hipMalloc(twiddle table)", __FILE__, __LINE__);
        }
    }
    const int threads = 256;
    const long long blocks = (M + threads - 1) / threads;
    hipStream_t bs = *temp ? stream : hipStream_t(0);
    hipLaunchKernelGGL(tfk::build_post_tw_kernel, dim3((unsigned)blocks), dim3(threads), 0, bs, d, d_hi, d_lo, h, R, B);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(bs);
    (void)hipFree(d_hi);
    (void)hipFree(d_lo);
    if (e != hipSuccess) {
        if (*temp) (void)hipFreeAsync(d, stream); else (void)hipFree(d);
        return hip_fail(e, "build_post_tw_kernel", __FILE__, __LINE__);
    }
    if (!*temp) {
        ctx->tables[key] = d;
        ctx->cached_post_bytes += size_t(M) * sizeof(u64);
    }
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
int get_pow_table(DeviceCtx* ctx, u64 offset_raw, size_t n, hipStream_t stream, const u64** out, bool* temp, size_t cosets = 1,
                  int log_order = 0) {
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
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&d), words * sizeof(u64), stream);
    if (e != hipSuccess) return hip_fail(e, "pool_malloc_async(pow table)", __FILE__, __LINE__);
    int rc = build_pow_tables(offset_raw, w, cosets, n, d, stream);
    if (rc) {
        (void)hipFreeAsync(d, stream);
        return rc;
    }
    *temp = true;
    *out = d;
    return TF_OK;
}

int ensure_tip5(DeviceCtx* ctx) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->tip5_ready) return TF_OK;
    tfk::Tip5Consts c;
    for (int i = 0; i < 80; ++i) c.rc[i] = gl::to_mont(kRoundConstants[i]);
    unsigned char lut[256];
    for (int x = 0; x < 256; ++x) {  // L(x) = ((x+1)^3 mod 257) - 1, tip5/mod.rs:1022-1026 (table :50-64)
        u64 xx = u64(x) + 1;
        lut[x] = (unsigned char)(((xx * xx * xx) + 256) % 257);
    }
    memcpy(c.lut, lut, 256);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(tfk::g_tip5), &c, sizeof(c)));
    HIPCHK(hipDeviceSynchronize());
    ctx->tip5_ready = true;
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
        const char* e = getenv("TF_NTT_WG_THREADS");
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
        const char* e = getenv("TF_NTT_ROUND_ELEMS");
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

std::atomic<int> g_ablate_cfg{-1};  // measurement only (TF_NTT_ABLATE=1|2 selects an ablated forward kernel; results are then garbage)
int ablate_mode() {
    int v = g_ablate_cfg.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("TF_NTT_ABLATE");
        v = e ? atoi(e) : 0;
        g_ablate_cfg.store(v, std::memory_order_relaxed);
    }
    return v;
}
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
    // (a PRE2 launch also reads the partner rows, 1024 rows further)
    return ok(l.a.in_rs, l.a.in_cs_hi, l.a.pre2_map ? 2048ull : 1024ull) && ok(l.a.out_rs, l.a.out_cs_hi) && ok(l.a.tw_rs, 0);
}

// the specialised R = 1024 last-pass kernel (LAST1024) is available unless an A/B switch or an ablation run disables it
bool last1024_enabled() {
    static const bool off = getenv("TF_NTT_NO_LAST1024") != nullptr;
    return !off && ablate_mode() == 0 && !t_small_launch;
}

bool col_enabled() {
    static const bool off = getenv("TF_NTT_NO_COL") != nullptr;  // A/B switch
    return !off;
}

// ... and its variant that multiplies on store (fast_coset_interpolate)
bool scaled_last1024_enabled() {
    static const bool off = getenv("TF_NTT_NO_SCALED_LAST1024") != nullptr;  // A/B switch
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
                           long long Q, int L, long long split = 0, int words = 0, bool pre2 = false) {
    Launch l{};
    tfk::NttPassArgs& A = l.a;
    const int p2 = (pre2 ? a - 1 : a) - 5, P2 = 1 << p2;
    const long long R = 1ll << a;  // elements per row (pre2: 2048, transformed as two interleaved 1024-point halves)
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
    if (pre2) {
        A.pre2_in_off = 1024 * L;
        A.pre2_out_off = A.out_rs;  // output k = 2 k' + h
        A.pre2_js_off = A.js_k;
        A.out_rs *= 2;
        A.js_k *= 2;
        A.pre2_map = (l.tiles % 8 == 0) ? 1 : 2;
        l.tiles *= 2;
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
    static const bool no_gfast = getenv("TF_NTT_NO_GFAST") != nullptr;  // A/B switch
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

template <bool INV, int SCALE, int MODE, bool LAST1024 = false, bool R1024 = false, bool COL = false, bool PRE2 = false>
int launch_pass_t(const Launch& l, hipStream_t stream) {
    // one attribute call per (instantiation, device): the kernels use up to the full 160 KiB of dynamic LDS
    static std::atomic<unsigned long long> done_mask{0};
    if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_pass_kernel<INV, SCALE, MODE, LAST1024, R1024, COL, PRE2>), (int)(160 * 1024), done_mask)) return rc_attr;
    // the R = 1024 column-pass instantiation stages its inner twiddle table behind the exchange buffer (LAST1024: part of
    // kLast1024LdsBytes already)
    const size_t lds_bytes = l.lds_bytes + ((TF_LDS_TW && !LAST1024 && MODE == 0 && l.a.inner_tw) ? (size_t(1) << l.a.p2) * tfk::kLdsTwStride * sizeof(u64) : 0);
    hipLaunchKernelGGL((tfk::ntt_pass_kernel<INV, SCALE, MODE, LAST1024, R1024, COL, PRE2>), dim3(l.tiles), dim3(l.threads), lds_bytes, stream,
                       l.a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// the R = 1024 column pass as a chain of `k` tiles per workgroup with the next tile's loads inside the store phase
// (ntt_col1024_chain_kernel); k from TF_NTT_PERSIST / tf_set_ntt_chain (0 or 1: the one-tile kernel)
std::atomic<int> g_chain{-1};
int chain_tiles() {
    int v = g_chain.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("TF_NTT_PERSIST");
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

unsigned long long* g_dbg_buf = nullptr;  // TF_NTT_ABLATE=3: per-wave phase stamps of the last launch (tf_debug_stamps)
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
        static const bool no_r1024_scale = getenv("TF_NTT_NO_R1024") != nullptr;
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
    static const bool no_r1024 = getenv("TF_NTT_NO_R1024") != nullptr;  // A/B switch
    const bool r1024 = l.a.p2 == 5 && l.a.post_tw && g_ablate == 0 && !no_r1024 && fits && std_geo;  // column pass with R = 1024
    if (last1024) {  // this instantiation lays its exchange buffer out itself (32 x 289 words, ntt_kernels.h)
        Launch l2 = l;
        l2.lds_bytes = std::max(l.lds_bytes, kLast1024LdsBytes);
        l2.threads = 512;  // 16 column slots x 32, also for tiles of 15 word-columns (XFE)
        return inverse ? launch_pass_t<true, 0, 0, true>(l2, stream) : launch_pass_t<false, 0, 0, true>(l2, stream);
    }
    // any other column pass whose offsets fit: the same treatment with a run-time P2 (COL)
    const bool col = l.a.post_tw && !r1024 && fits && !l.a.gfast && g_ablate == 0 && col_enabled();
    if (r1024 && chain_tiles() > 1 && l.tiles >= 1024 && !l.a.dbg)
        return inverse ? launch_chain_t<true>(l, chain_tiles(), stream) : launch_chain_t<false>(l, chain_tiles(), stream);
    if (inverse) return r1024 ? launch_pass_t<true, 0, 0, false, true>(l, stream)
                              : (col ? launch_pass_t<true, 0, 0, false, false, true>(l, stream) : launch_pass_t<true, 0, 0>(l, stream));
    if (r1024) return launch_pass_t<false, 0, 0, false, true>(l, stream);
    if (col) return launch_pass_t<false, 0, 0, false, false, true>(l, stream);
    if (g_ablate == 1) return launch_pass_t<false, 0, 1>(l, stream);
    if (g_ablate == 2) return launch_pass_t<false, 0, 2>(l, stream);
    if (g_ablate == 3) {
        Launch l2 = l;
        l2.a.dbg = g_dbg_buf;
        return launch_pass_t<false, 0, 3>(l2, stream);
    }
    return launch_pass_t<false, 0, 0>(l, stream);
}

int check_len(size_t n) {
    if (n != 0 && (n & (n - 1))) return TF_ERR_LEN_NOT_POWER_OF_TWO;  // ntt.rs:137
    if (n > (size_t(1) << 31)) return TF_ERR_LEN_TOO_LARGE;           // ntt.rs:134-139: lengths beyond u32::MAX panic
    return TF_OK;
}

// n = 32, contiguous transforms: LDS-staged rows (ntt_rows32_kernel)
template <bool INV>
int launch_rows32_t(const tfk::NttRows32Args& a, unsigned grid, hipStream_t stream) {
    static std::atomic<unsigned long long> done_mask{0};
    if (int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::ntt_rows32_kernel<INV>), (int)(160 * 1024), done_mask)) return rc_attr;
    hipLaunchKernelGGL((tfk::ntt_rows32_kernel<INV>), dim3(grid), dim3(512), size_t(256) * 33 * sizeof(u64), stream, a);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

int launch_rows32(const u64* in, u64* out, size_t batch, int L, bool inverse, hipStream_t stream) {
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
    static const bool off = getenv("TF_NTT_NO_LAT") != nullptr;  // A/B switch
    // measured crossover against the pass / block kernels (tools/lat_sweep.py, profiles/r03_lat_sweep_*.txt): 2.0 - 2.9 x faster up
    // to 2^20 words per call, level at 2^22 words (BFieldElement) / 1.5 x 2^20 words (XFieldElement: its loads step 24 bytes)
    static const long long env_limit = [] {
        const char* e = getenv("TF_NTT_LAT_MAX_WORDS");
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
               long long n_coeffs, const u64* in2, hipStream_t stream, const tfk::NttLatArgs* mods = nullptr) {
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
// When: the level's transforms are the latency-shaped kernel's anyway AND the level is small enough that launches, not work,
// are what it costs (measured crossover, tools/tree_latency.py / profiles/r03_tree_level_ab.txt).  TF_TREE_NO_LEVEL: A/B switch.
bool tree_level_wanted(long long order, long long lines, int L = 1, bool up = false) {
    static const bool off = getenv("TF_TREE_NO_LEVEL") != nullptr;
    static const bool on_xfe = getenv("TF_TREE_LEVEL_XFE") != nullptr;  // measured loss, opt-in (below)
    static const long long limit = [] {
        const char* e = getenv("TF_TREE_LEVEL_MAX_WORDS");
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
int launch_tree_level(DeviceCtx* ctx, int log_n, tfk::TreeLevelArgs a, hipStream_t s, int L = 1) {
    int rc = get_lat_table(ctx, log_n, false, &a.tw_f);
    if (!rc) rc = get_lat_table(ctx, log_n, true, &a.tw_i);
    if (rc) return rc;
    a.ninv = gl::mont_inverse(gl::to_mont(u64(1) << log_n));
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
    static const bool off = getenv("TF_TREE_NO_BUILD_LEVEL") != nullptr;
    static const long long limit = [] {
        const char* e = getenv("TF_TREE_BUILD_MAX_WORDS");
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
            static const bool wide = getenv("TF_NTT_LAT2_NO_WIDE") == nullptr;  // A/B switch
            return wide ? launch_lat2_t<10, INV, LAST, 512>(a, batch, s) : launch_lat2_t<10, INV, LAST>(a, batch, s);
        }
    }
    return TF_ERR_HIP;
}
bool lat2_wanted(int log_n, size_t batch, int L) {
    static const bool off = getenv("TF_NTT_NO_LAT") != nullptr || getenv("TF_NTT_NO_LAT2") != nullptr;  // A/B switches
    static const long long env_limit = [] {
        const char* e = getenv("TF_NTT_LAT2_MAX_WORDS");
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
                long long n_coeffs, const u64* in2, hipStream_t stream) {
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
        r.L = L;
        r.cfast = 0;
        rc = inverse ? launch_lat2_dir<true, true>(a2, r, batch, stream) : launch_lat2_dir<false, true>(a2, r, batch, stream);
    }
    scratch_release(ctx, sblk, stream);
    if (post_temp) (void)hipFreeAsync(const_cast<u64*>(post), stream);
    return rc;
}


// Experiment switches for tools/split3.py: looked up on every call only when TF_NTT_EXPERIMENT is set at load time
// (the sweep tool changes them while the process runs); otherwise the planner never touches the environment.
const char* exp_env(const char* name) {
    static const bool enabled = getenv("TF_NTT_EXPERIMENT") != nullptr;
    return enabled ? getenv(name) : nullptr;
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
        if (L == 3 && P == 2 && log_n < 20 && log_n != 15) last = log_n < 15 ? 5 : 9;
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
    static const bool no_block = getenv("TF_NTT_NO_BLOCK") != nullptr;
    if (n <= (size_t(1) << 14)) return L == 1 && !no_block && g_min_passes.load(std::memory_order_relaxed) == 0;  // the block kernel truncates (BFE product inverse)
    const int log_n = ilog2(n), P = pass_count(log_n);
    int a[4] = {0, 0, 0, 0};
    choose_split(log_n, P, L, a);
    // ... and that kernel addresses its stores through a buffer resource: 1024 * (n L / 1024) * 8 bytes must fit 32 bits (fits_buffer_offsets)
    return P <= 3 && a[P - 1] == 10 && last1024_enabled() && (unsigned long long)n * L * 8 + (1ull << 20) < (1ull << 32);
}

// Two-pass plans for 2^21 / 2^22 points (a 2048-point pass = pairs of 1024-point workgroups, ntt_kernels.h PRE2).
std::atomic<int> g_pre2_mode{-1};  // tf_set_ntt_two_pass: -1 automatic (TF_NTT_NO_PRE2 disables), 0 never, 1 whenever supported
bool pre2_plan_ok(int log_n, int L, size_t n, size_t cosets, bool has_in2, long long n_out, bool inverse, bool load_work, bool store_scale) {
    static const bool off = getenv("TF_NTT_NO_PRE2") != nullptr;  // A/B switch
    const int mode = g_pre2_mode.load(std::memory_order_relaxed);
    if (mode == 0 || (mode < 0 && off)) return false;
    if (log_n < 21 || log_n > 22 || cosets != 1 || has_in2 || n_out >= 0) return false;
    if ((load_work && inverse) || (store_scale && !inverse)) return false;          // shapes no caller produces
    if (!last1024_enabled() || ablate_mode() != 0 || wg_threads() != 512) return false;
    if (g_min_passes.load(std::memory_order_relaxed) > 2) return false;
    return (unsigned long long)n * L * 8 + (1ull << 20) < (1ull << 32);              // buffer addressing (fits_buffer_offsets)
}
void pre2_split(int log_n, int (&a)[4]) {
    // 2^21: the 2048-point pass last (the first pass of a coset evaluation then scales every coefficient once); 2^22: both
    a[0] = log_n == 22 ? 11 : 10, a[1] = 11, a[2] = a[3] = 0;
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
            bool inverse, const u64* pre_scale, long long n_coeffs, hipStream_t stream, const u64* post_scale = nullptr,
            size_t cosets = 1, const u64* in2 = nullptr, long long n_out = -1) {
    // n_out >= 0 (only with can_truncate(n, L)): the last pass stores output elements j < n_out only and out_bs may be
    // n_out * L -- the truncation of fast_multiply without a copy; `in` is then used as work space and clobbered
    if (n == 0 || batch == 0) return TF_OK;
    const int log_n = ilog2(n);
    int rc;
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
    if (log_n == 5 && L == 1 && !pre_scale && !post_scale && n_coeffs < 0 && !in2 && in_bs == 32 && out_bs == 32) {  // XFE: 0.98 vs 0.90 ms, not used
        static const bool no_rows32 = getenv("TF_NTT_NO_ROWS32") != nullptr;  // A/B switch
        if (!no_rows32) return launch_rows32(in, out, batch, L, inverse, stream);
    }
    if (!pre_scale && !post_scale && n_out < 0 && cosets == 1 && (!in2 || L == 1) && g_min_passes.load(std::memory_order_relaxed) == 0 &&
        lat_wanted(log_n, batch, L))
        return launch_lat(ctx, in, out, in_bs, out_bs, log_n, batch, L, inverse, n_coeffs, in2, stream);
    if (!pre_scale && !post_scale && n_out < 0 && cosets == 1 && (!in2 || L == 1) && g_min_passes.load(std::memory_order_relaxed) == 0 &&
        lat2_wanted(log_n, batch, L))
        return launch_lat2(ctx, in, out, in_bs, out_bs, log_n, batch, L, inverse, n_coeffs, in2, stream);
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
        static const bool no_block = getenv("TF_NTT_NO_BLOCK") != nullptr;  // A/B switch
        const bool product_inverse = in2 && inverse && !pre_scale && !post_scale && n_coeffs < 0;
        // XFieldElement slices take the same kernel as three limb transforms per slice with element stride 3 (round 2;
        // TF_NTT_NO_XFE_BLOCK restores the two-pass plan for an A/B run): one HBM pass instead of two
        static const bool no_xfe_block = getenv("TF_NTT_NO_XFE_BLOCK") != nullptr;
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
    const int small_mode = g_small_launch_mode.load(std::memory_order_relaxed);
    static const bool no_small = getenv("TF_NTT_NO_SMALL_LAUNCH") != nullptr;  // A/B switch
    const bool small_call = (unsigned long long)n * cosets * batch * L <= (1ull << 21);
    SmallLaunchScope small_scope(n_out < 0 && !wg_env() && (small_mode == 1 || (small_mode < 0 && small_call && !no_small)));
    int a[4] = {0, 0, 0, 0};
    int P = pass_count(log_n);
    choose_split(log_n, P, L, a);
    // 2^21 and 2^22 points in TWO passes: a 2048-point pass runs as pairs of 1024-point workgroups that share their input
    // (ntt_kernels.h, PRE2; a[i] = 11 below).  Plain transforms, coset evaluation (forward) and coset interpolation (inverse).
    bool pre2[4] = {false, false, false, false};
    if (pre2_plan_ok(log_n, L, n, cosets, in2 != nullptr, n_out, inverse, pre_scale != nullptr || n_coeffs >= 0, post_scale != nullptr)) {
        P = 2;
        pre2_split(log_n, a);
        pre2[0] = a[0] == 11, pre2[1] = a[1] == 11;
    }
    const u64* inner[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < P; ++i) {
        rc = get_inner_table(ctx, pre2[i] ? 10 : a[i], inverse, (i == P - 1 && inverse) ? log_n : 0, &inner[i], pre2[i]);  // n^-1 rides on the last pass
        if (rc) return rc;
    }
    const u64* post[3] = {nullptr, nullptr, nullptr};
    bool post_temp[3] = {false, false, false};
    auto release_tables = [&]() {
        for (int i = 0; i < 3; ++i)
            if (post_temp[i] && post[i]) (void)hipFreeAsync(const_cast<u64*>(post[i]), stream);
    };
    {
        int rest = log_n;
        for (int i = 0; i + 1 < P; ++i) {
            rc = get_post_table(ctx, rest, a[i], inverse, stream, &post[i], &post_temp[i]);
            if (rc) {
                release_tables();
                return rc;
            }
            rest -= a[i];
        }
    }
    read_env();
    static const bool no_col_shift = getenv("TF_NTT_NO_COL_SHIFT") != nullptr;  // A/B switch
    const size_t poly_bytes = n * cosets * size_t(L) * sizeof(u64);
    size_t tb = std::max<size_t>(1, g_tile_bytes / poly_bytes);
    tb = std::min(tb, batch);
    // pipelined tiles (g_pipe): tile t runs on side stream t % K with scratch tile t % K; the caller's stream forks into
    // the side streams before the first tile and joins them after the last (event edges only, no host synchronisation)
    const size_t ntiles = (batch + tb - 1) / tb;
    tb = (batch + ntiles - 1) / ntiles;  // equal tiles: 64 polynomials at 21 per tile are 4 x 16, not 21 + 21 + 21 + 1
    int K = (P < 4) ? (int)std::min<size_t>((size_t)g_pipe.load(std::memory_order_relaxed), ntiles) : 1;
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
            if (pre2[i]) {
                // partner coefficients are n / 2 apart: offset^(n/2) is word n / 2 of the scale table when the polynomial is that long
                p.a.pre2_cp = pre_scale ? pre_scale + ((long long)(n / 2) < n_coeffs ? (long long)(n / 2) : 0) : nullptr;
            }
            if (i == 0 && src != dst) p.a.nt = g_nt.load(std::memory_order_relaxed) & 1;  // the caller's input is read once
            if (i == 0) {
                p.a.pre_scale = pre_scale;
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
            static const bool no_words16 = getenv("TF_NTT_NO_WORDS16") != nullptr;  // A/B switch
            const bool plain1024 = (a[P - 1] == 10 || pre2[P - 1]) && last1024_enabled() && (!post_scale || (inverse && scaled_last1024_enabled() && log_n <= 28));
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
                                            words, pre2[P - 1]);
            pl.a.inner_tw = inner[P - 1];
            pl.a.post_scale = post_scale;
            pl.a.n_out = n_out;
            pl.a.nt = g_nt.load(std::memory_order_relaxed) & 2;  // the result is written once
            if (plain1024 && pl.a.nc == 16 && !no_col_shift && (unsigned long long)n * L * sizeof(u64) < (1ull << 32)) {
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
    static const bool no_split = getenv("TF_COSET_EVAL_NO_SPLIT") != nullptr;  // A/B switch
    // (at most kMaxCosetSplit cosets: one scale table of n_coeffs words per coset is built and, while it fits the budget, cached)
    if (!no_split && len > 1024 && len < order && order / len <= kMaxCosetSplit && order <= (size_t(1) << 30) &&
        pass_count(ilog2(len)) < pass_count(ilog2(order))) {
        const size_t cosets = order / len;
        rc = get_pow_table(ctx, offset_raw, n_coeffs, s, &pw, &temp, cosets, ilog2(order));
        if (rc) return rc;
        rc = run_ntt(ctx, d_coeffs, d_out, (long long)n_coeffs * L, (long long)order * L, len, batch, L, false, pw,
                     (long long)n_coeffs, s, nullptr, cosets);
    } else {
        rc = get_pow_table(ctx, offset_raw, n_coeffs, s, &pw, &temp);
        if (rc) return rc;
        rc = run_ntt(ctx, d_coeffs, d_out, (long long)n_coeffs * L, (long long)order * L, order, batch, L, false, pw,
                     (long long)n_coeffs, s);
    }
    if (temp) (void)hipFreeAsync(const_cast<u64*>(pw), s);
    return rc;
}

// ------------------------------------------------------------------------------------ Tip5 / Merkle
// Launches of at most this many permutation chains use the 16-lanes-per-permutation kernels (measured crossover: one
// permutation per lane costs ~19 us however few there are; 2^15 items x 16 lanes = 2 waves per SIMD)
constexpr long long kCoopMaxCount = 1ll << 15;

int tip5_permute_dev(u64* d_states, size_t count, void* stream) {
    if (count == 0) return TF_OK;
    if (!d_states) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    if ((long long)count <= kCoopMaxCount) {
        hipLaunchKernelGGL(tfk::tip5_permute_coop_kernel, dim3((unsigned)((count + 15) / 16)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), d_states, (long long)count);
    } else {
        const long long blocks = ((long long)count + 255) / 256;
        hipLaunchKernelGGL(tfk::tip5_permute_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           d_states, (long long)count);
    }
    HIPCHK(hipGetLastError());
    return TF_OK;
}

int tip5_trace_dev(u64* d_states, u64* d_trace, size_t count, void* stream) {
    if (count == 0) return TF_OK;
    if (!d_states || !d_trace) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    const long long blocks = ((long long)count + 255) / 256;
    hipLaunchKernelGGL(tfk::tip5_trace_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_states, d_trace,
                       (long long)count);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

int launch_hash_pairs(const u64* in, u64* out, u64* leaf_copy, long long count, long long per_tree, long long in_ts,
                      long long out_ts, long long copy_ts, hipStream_t s) {
    if (count == 0) return TF_OK;
    if (count <= kCoopMaxCount && !leaf_copy) {
        // fewer permutations than the GPU has lanes: latency, not throughput, is what this launch costs -> 16 lanes each
        const long long blocks = (count + 15) / 16;
        hipLaunchKernelGGL(tfk::tip5_hash_pairs_coop_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, count, per_tree, in_ts,
                           out_ts);
        HIPCHK(hipGetLastError());
        return TF_OK;
    }
    const long long blocks = (count + 255) / 256;
    hipLaunchKernelGGL(tfk::tip5_hash_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, leaf_copy, count,
                       per_tree, in_ts, out_ts, copy_ts);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// hash_varlen of n_rows rows: few rows (or one long input) are latency-bound -> 16 lanes per row
int launch_hash_varlen_rows(const u64* rows, long long row_len, long long n_rows, u64* out, long long per_tree, long long out_ts,
                            hipStream_t s) {
    if (n_rows == 0) return TF_OK;
    if (n_rows <= kCoopMaxCount) {
        hipLaunchKernelGGL(tfk::tip5_hash_varlen_rows_coop_kernel, dim3((unsigned)((n_rows + 15) / 16)), dim3(256), 0, s, rows, row_len,
                           n_rows, out, per_tree, out_ts);
    } else {
        hipLaunchKernelGGL(tfk::tip5_hash_varlen_rows_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, rows, row_len,
                           n_rows, out, per_tree, out_ts);
    }
    HIPCHK(hipGetLastError());
    return TF_OK;
}

int tip5_hash_pairs_dev(const u64* d_in, u64* d_out, size_t count, void* stream) {
    if (count == 0) return TF_OK;
    if (!d_in || !d_out) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    return launch_hash_pairs(d_in, d_out, nullptr, (long long)count, (long long)count, 0, 0, 0,
                             static_cast<hipStream_t>(stream));
}

int tip5_hash_varlen_rows_dev(const u64* d_rows, size_t row_len, size_t n_rows, u64* d_out, void* stream) {
    if (n_rows == 0) return TF_OK;
    if (!d_out || (row_len && !d_rows)) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    return launch_hash_varlen_rows(d_rows, (long long)row_len, (long long)n_rows, d_out, (long long)n_rows, 0ll,
                                   static_cast<hipStream_t>(stream));
}

int check_leaves(size_t n) {
    if (n == 0) return TF_ERR_TOO_FEW_LEAFS;                  // merkle_tree.rs:394-396
    if (n & (n - 1)) return TF_ERR_INCORRECT_NUMBER_OF_LEAFS;  // :398-401
    return TF_OK;
}

constexpr long long kTopWidth = 256;  // levels of at most this many nodes finish in one workgroup per tree

// Levels above the leaf level for trees whose leaves are already at nodes[n..2n) (nodes[0] gets zeroed).
int merkle_levels_in_place(u64* d_nodes, long long N, size_t batch, hipStream_t s) {
    const long long nodes_ts = 10 * N;
    long long w = N;
    while (w > kTopWidth) {  // nodes[w/2 .. w) from nodes[w .. 2w)
        const long long nw = w / 2;
        int rc = launch_hash_pairs(d_nodes + 5 * w, d_nodes + 5 * nw, nullptr, nw * (long long)batch, nw, nodes_ts, nodes_ts, 0, s);
        if (rc) return rc;
        w = nw;
    }
    hipLaunchKernelGGL(tfk::merkle_top_kernel, dim3((unsigned)batch), dim3(1024), 0, s, d_nodes + 5 * w, nodes_ts, (int)w,
                       d_nodes, nodes_ts, (u64*)nullptr, (const u64*)nullptr, 0ll);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// hash_varlen of the rows of `batch` column-major tables (one codeword per column): digests to out + t * out_ts + 5 * i
int launch_hash_table_rows(const u64* table, long long n_rows, long long n_cols, int width, long long col_stride, long long table_stride,
                           long long batch, u64* out, long long out_ts, hipStream_t s) {
    const long long total = n_rows * batch;
    if (total == 0) return TF_OK;
    if (total <= kCoopMaxCount) {
        hipLaunchKernelGGL(tfk::tip5_hash_table_rows_coop_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s, table, n_rows, n_cols,
                           width, col_stride, table_stride, total, out, out_ts);
    } else {
        hipLaunchKernelGGL(tfk::tip5_hash_table_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, table, n_rows, n_cols,
                           width, col_stride, table_stride, total, out, out_ts);
    }
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// Rows of a COLUMN-major table ("hash_varlen over rows of a column-major table, the producer of leaves")
// -> leaf digests -> Merkle tree.  table: batch x n_cols columns of n_rows elements of `width` words, col_stride words apart.
int hash_table_rows_dev(const u64* d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, u64* d_digests, size_t batch,
                        void* stream) {
    if (width != 1 && width != 3) return TF_ERR_NULL_POINTER;
    if (n_rows == 0 || batch == 0) return TF_OK;
    if (!d_digests || (n_cols && !d_table)) return TF_ERR_NULL_POINTER;
    if (n_cols * size_t(width) >= (size_t(1) << 31) || col_stride < n_rows * size_t(width)) return TF_ERR_LEN_TOO_LARGE;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    return launch_hash_table_rows(d_table, (long long)n_rows, (long long)n_cols, width, (long long)col_stride,
                                  (long long)(n_cols * col_stride), (long long)batch, d_digests, 5ll * (long long)n_rows,
                                  static_cast<hipStream_t>(stream));
}

int merkle_from_columns_dev(const u64* d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, u64* d_nodes, size_t batch,
                            void* stream) {
    if (width != 1 && width != 3) return TF_ERR_NULL_POINTER;
    int rc = check_leaves(n_rows);
    if (rc) return rc;
    if (batch == 0) return TF_OK;
    if (!d_nodes || (n_cols && !d_table)) return TF_ERR_NULL_POINTER;
    if (n_cols * size_t(width) >= (size_t(1) << 31) || col_stride < n_rows * size_t(width)) return TF_ERR_LEN_TOO_LARGE;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long N = (long long)n_rows;
    rc = launch_hash_table_rows(d_table, N, (long long)n_cols, width, (long long)col_stride, (long long)(n_cols * col_stride),
                                (long long)batch, d_nodes + 5 * N, 10 * N, s);
    if (rc) return rc;
    return merkle_levels_in_place(d_nodes, N, batch, s);
}

// Rows of a row-major table -> leaf digests (hash_varlen per row, tip5/mod.rs:617-623) -> Merkle tree, without the
// leaves ever leaving HBM.  rows: batch x n_rows x row_len words.
int merkle_from_rows_dev(const u64* d_rows, size_t row_len, size_t n_rows, u64* d_nodes, size_t batch, void* stream) {
    int rc = check_leaves(n_rows);
    if (rc) return rc;
    if (batch == 0) return TF_OK;
    if (!d_nodes || (row_len && !d_rows)) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long N = (long long)n_rows, total = N * (long long)batch;
    rc = launch_hash_varlen_rows(d_rows, (long long)row_len, total, d_nodes + 5 * N, N, 10 * N, s);
    if (rc) return rc;
    return merkle_levels_in_place(d_nodes, N, batch, s);
}

// nodes layout per tree: 2n digests (merkle_tree.rs:85-88, :393-429).
int merkle_build_dev(const u64* d_leaves, size_t n, u64* d_nodes, size_t batch, void* stream) {
    int rc = check_leaves(n);
    if (rc) return rc;
    if (batch == 0) return TF_OK;
    if (!d_leaves || !d_nodes) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long N = (long long)n, nodes_ts = 10 * N, leaves_ts = 5 * N;
    if (N <= kTopWidth) {
        hipLaunchKernelGGL(tfk::merkle_top_kernel, dim3((unsigned)batch), dim3(1024), 0, s, d_leaves, leaves_ts, (int)N,
                           d_nodes, nodes_ts, (u64*)nullptr, d_leaves, leaves_ts);
        HIPCHK(hipGetLastError());
        return TF_OK;
    }
    // first level: read leaves, write the leaf copy nodes[n..2n) and the parents nodes[n/2..n)
    long long w = N / 2;
    rc = launch_hash_pairs(d_leaves, d_nodes + 5 * w, d_nodes + 5 * N, w * (long long)batch, w, leaves_ts, nodes_ts,
                           nodes_ts, s);
    if (rc) return rc;
    while (w > kTopWidth) {  // nodes[w/2 .. w) from nodes[w .. 2w)
        const long long nw = w / 2;
        rc = launch_hash_pairs(d_nodes + 5 * w, d_nodes + 5 * nw, nullptr, nw * (long long)batch, nw, nodes_ts, nodes_ts, 0,
                               s);
        if (rc) return rc;
        w = nw;
    }
    hipLaunchKernelGGL(tfk::merkle_top_kernel, dim3((unsigned)batch), dim3(1024), 0, s, d_nodes + 5 * w, nodes_ts, (int)w,
                       d_nodes, nodes_ts, (u64*)nullptr, (const u64*)nullptr, 0ll);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

int merkle_root_dev(const u64* d_leaves, size_t n, u64* d_root, size_t batch, void* stream) {
    int rc = check_leaves(n);
    if (rc) return rc;
    if (batch == 0) return TF_OK;
    if (!d_leaves || !d_root) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long N = (long long)n, leaves_ts = 5 * N;
    if (N <= kTopWidth) {
        hipLaunchKernelGGL(tfk::merkle_top_kernel, dim3((unsigned)batch), dim3(1024), 0, s, d_leaves, leaves_ts, (int)N,
                           (u64*)nullptr, 0ll, d_root, (const u64*)nullptr, 0ll);
        HIPCHK(hipGetLastError());
        return TF_OK;
    }
    // ping-pong level buffers: n/2 + n/4 digests per tree
    u64* buf = nullptr;
    const size_t words = size_t(batch) * size_t(5) * size_t(N / 2 + N / 4);
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&buf), words * sizeof(u64), s);
    if (e != hipSuccess) {
        hip_fail(e, "pool_malloc_async(merkle levels)", __FILE__, __LINE__);
        return TF_ERR_TREE_TOO_HIGH;
    }
    u64* a = buf;
    u64* b = buf + size_t(batch) * 5 * size_t(N / 2);
    long long w = N / 2;
    rc = launch_hash_pairs(d_leaves, a, nullptr, w * (long long)batch, w, leaves_ts, 5 * w, 0, s);
    while (rc == TF_OK && w > kTopWidth) {
        const long long nw = w / 2;
        rc = launch_hash_pairs(a, b, nullptr, nw * (long long)batch, nw, 5 * w, 5 * nw, 0, s);
        std::swap(a, b);
        w = nw;
    }
    if (rc == TF_OK) {
        hipLaunchKernelGGL(tfk::merkle_top_kernel, dim3((unsigned)batch), dim3(1024), 0, s, a, 5 * w, (int)w, (u64*)nullptr,
                           0ll, d_root, (const u64*)nullptr, 0ll);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) rc = hip_fail(le, "merkle_top_kernel", __FILE__, __LINE__);
    }
    e = hipFreeAsync(buf, s);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(e, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}

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
    if (temp) (void)hipFreeAsync(const_cast<u64*>(pw), s);
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
int poly_mul_dev(const u64* a, size_t na, const u64* b, size_t nb, u64* out, size_t batch, int L, void* stream, long long a_bs = 0,
                 long long b_bs = 0) {
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
    static const bool no_fuse = getenv("TF_POLY_MUL_NO_FUSE") != nullptr;  // A/B switch
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
    static const bool no_fuse = getenv("TF_POLY_MUL_NO_FUSE") != nullptr;  // A/B switch
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
        const char* e = getenv("TF_TREE_LEAF_LOG");
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
        const char* e = getenv("TF_TREE_INTERP_LEAF_LOG");
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
        const char* e = getenv("TF_TREE_LEAF_SPLIT_MAX");
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
    if (3 * kTreeLeaf * L * sizeof(u64) > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tfk::leaf_zerofier_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * kTreeLeaf * L * sizeof(u64)));
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
    static const bool off = getenv("TF_TREE_NO_FUSE") != nullptr;
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
    static const bool no_split = getenv("TF_TREE_NO_LEAF_SPLIT") != nullptr;
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
        const char* e = getenv("TF_BATCH_EVAL");
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
    static const size_t slab_elems = [] { const char* e = getenv("TF_TREE_UNIT_SLAB"); const long long v = e ? atoll(e) : 0; return v > 0 ? (size_t)v : (size_t(1) << 25); }();
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
        if (3 * kTreeLeaf * L * sizeof(u64) > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tfk::leaf_zerofier_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * kTreeLeaf * L * sizeof(u64)));
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
        static const bool no_split = getenv("TF_TREE_NO_LEAF_SPLIT") != nullptr;
        constexpr int kSplit = 4;
        const size_t div_lds = ((size_t)2 * kTreeLeaf + (size_t)kTreeLeaf * (kTreeLeaf + 1) + (size_t)kSplit * kTreeLeaf) * L * sizeof(u64);
        if (!no_split && (long long)nr * M <= (L == 1 ? 2 : 1) * leaf_split_max() && kTreeLeaf * kSplit <= 1024 && kTreeLeaf >= 4 * kSplit && div_lds <= 144 * 1024) {
            static std::atomic<unsigned long long> done_mask{0};
            if (div_lds > 48 * 1024) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&tfk::leaf_interpolant_div_kernel<L, kSplit>), 144 * 1024, done_mask);
            if (!rc)
                hipLaunchKernelGGL((tfk::leaf_interpolant_div_kernel<L, kSplit>), dim3((unsigned)(M / kTreeLeaf), (unsigned)nr), dim3(kTreeLeaf * kSplit),
                                   div_lds, s, domain, values + r0 * n * L, winv, (const u64*)(h > 0 ? pt.T.tails[0] : pt.root_tail) /* a single leaf is the root */, (long long)n, kTreeLeaf, M, na);
        } else {
            if (6 * kTreeLeaf * L * sizeof(u64) > 48 * 1024)  // only with a leaf size forced through TF_TREE_LEAF_LOG
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tfk::leaf_interpolant_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(6 * kTreeLeaf * L * sizeof(u64)));
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
            static const bool fuse_interp = getenv("TF_TREE_FUSE_INTERP") != nullptr;
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
int interpolate_dev(const u64* domain, const u64* values, size_t n, size_t rows, u64* out, int L, void* stream, int* d_status = nullptr) {
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

int tree_handle_new(const u64* d_domain, size_t n, int L, void* stream, TreeHandle** out, bool async = false) {
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

int tree_handle_interpolate(TreeHandle* H, const u64* d_values, size_t rows, u64* d_out, void* stream, int* d_status = nullptr) {
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
        } else if (d_status) {  // weights computed by an earlier asynchronous call: pass its verdict on
            hipLaunchKernelGGL(tfk::status_merge_kernel, dim3(1), dim3(1), 0, s, (const int*)H->bad, d_status, (int)TF_ERR_INVERSE_OF_ZERO, (int)TF_ERR_INVERSE_OF_ZERO);
            if (hipGetLastError() != hipSuccess) return TF_ERR_HIP;
        }
    }
    return H->L == 1 ? tree_interpolate_rows<1>(ctx, H->pt, H->points, H->winv, d_values, rows, d_out, s)
                     : tree_interpolate_rows<3>(ctx, H->pt, H->points, H->winv, d_values, rows, d_out, s);
}

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
        static const int rows_env = [] { const char* e = getenv("TF_BARY_ROWS"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= tfk::kBaryRows) ? v : 0; }();
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
int clean_divide_dev(const u64* a, size_t na, const u64* b, size_t nb, u64* out, void* stream, size_t batch = 1, int* d_status = nullptr) {
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

// ------------------------------------------------------------------------------------ host-pointer wrappers
struct DevBuf {
    u64* p = nullptr;
    hipStream_t s;
    explicit DevBuf(hipStream_t st) : s(st) {}
    int alloc(size_t words) {
        if (words == 0) return TF_OK;
        hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&p), words * sizeof(u64), s);
        if (e != hipSuccess) return hip_fail(e, "hipMallocAsync", __FILE__, __LINE__);
        return TF_OK;
    }
    ~DevBuf() {
        if (p) (void)hipFreeAsync(p, s);
    }
};

// Host buffers are pageable: the runtime stages such copies, and a staged H2D chunk was observed to land
// AFTER a kernel enqueued behind it on the same stream had already rewritten the destination in place.
// The host-pointer entry points therefore wait for the upload before enqueueing compute.
int h2d(u64* d, const u64* h, size_t words, hipStream_t s) {
    if (!words) return TF_OK;
    HIPCHK(hipMemcpyAsync(d, h, words * sizeof(u64), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return TF_OK;
}

// One private non-blocking stream per (host thread, device) for the host-pointer entry points.
hipStream_t host_stream() {
    thread_local hipStream_t streams[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    if (!streams[dev]) {
        if (hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            streams[dev] = nullptr;
        }
    }
    return streams[dev];
}
int d2h(u64* h, const u64* d, size_t words, hipStream_t s) {
    if (!words) return TF_OK;
    HIPCHK(hipMemcpyAsync(h, d, words * sizeof(u64), hipMemcpyDeviceToHost, s));
    return TF_OK;
}
int sync(hipStream_t s) {
    HIPCHK(hipStreamSynchronize(s));
    return TF_OK;
}

#define TRY(x)            \
    do {                  \
        int rc_ = (x);    \
        if (rc_) return rc_; \
    } while (0)

int ntt_host(u64* x, size_t n, size_t batch, int L, int inverse) {
    TRY(check_len(n));
    if (n <= 1 || batch == 0) return TF_OK;
    if (!x) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf d(s);
    const size_t words = n * batch * L;
    TRY(d.alloc(words));
    TRY(h2d(d.p, x, words, s));
    TRY(ntt_dev(d.p, n, batch, L, inverse, s));
    TRY(d2h(x, d.p, words, s));
    return sync(s);
}

int coset_eval_host(const u64* coeffs, size_t n_coeffs, u64 offset_raw, u64* out, size_t order, size_t batch, int L) {
    if (n_coeffs > order) return TF_ERR_ORDER_NOT_ABOVE_DEGREE;
    TRY(check_len(order));
    if (order == 0 || batch == 0) return TF_OK;
    if (!out || (n_coeffs && !coeffs)) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf din(s), dout(s);
    TRY(din.alloc(n_coeffs * batch * L));
    TRY(dout.alloc(order * batch * L));
    TRY(h2d(din.p, coeffs, n_coeffs * batch * L, s));
    TRY(coset_eval_dev(din.p, n_coeffs, offset_raw, dout.p, order, batch, L, s));
    TRY(d2h(out, dout.p, order * batch * L, s));
    return sync(s);
}

template <class F>
int host_roundtrip(const uint64_t* in1, size_t w1, const uint64_t* in2, size_t w2, uint64_t* out, size_t wo, F&& body) {
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf d1(s), d2(s), dout(s);
    TRY(d1.alloc(w1));
    TRY(d2.alloc(w2));
    TRY(dout.alloc(wo));
    TRY(h2d(d1.p, in1, w1, s));
    TRY(h2d(d2.p, in2, w2, s));
    TRY(body(d1.p, d2.p, dout.p, s));
    TRY(d2h(out, dout.p, wo, s));
    return sync(s);
}

}  // namespace

// ==================================================================================== C ABI
extern "C" {

const char* tf_status_string(int status) {
    switch (status) {
        case TF_OK: return "TF_OK";
        case TF_ERR_TOO_FEW_LEAFS: return "TF_ERR_TOO_FEW_LEAFS";
        case TF_ERR_INCORRECT_NUMBER_OF_LEAFS: return "TF_ERR_INCORRECT_NUMBER_OF_LEAFS";
        case TF_ERR_TREE_TOO_HIGH: return "TF_ERR_TREE_TOO_HIGH";
        case TF_ERR_LEN_NOT_POWER_OF_TWO: return "TF_ERR_LEN_NOT_POWER_OF_TWO";
        case TF_ERR_LEN_TOO_LARGE: return "TF_ERR_LEN_TOO_LARGE";
        case TF_ERR_ORDER_NOT_ABOVE_DEGREE: return "TF_ERR_ORDER_NOT_ABOVE_DEGREE";
        case TF_ERR_NULL_POINTER: return "TF_ERR_NULL_POINTER";
        case TF_ERR_NO_DEVICE: return "TF_ERR_NO_DEVICE";
        case TF_ERR_HIP: return "TF_ERR_HIP";
        case TF_ERR_OUT_OF_MEMORY: return "TF_ERR_OUT_OF_MEMORY";
        case TF_ERR_LEAF_INDEX_INVALID: return "TF_ERR_LEAF_INDEX_INVALID";
        case TF_ERR_INVERSE_OF_ZERO: return "TF_ERR_INVERSE_OF_ZERO";
        case TF_ERR_BUFFER_TOO_SMALL: return "TF_ERR_BUFFER_TOO_SMALL";
        case TF_ERR_EMPTY_DOMAIN: return "TF_ERR_EMPTY_DOMAIN";
        case TF_ERR_DIVISION_BY_ZERO: return "TF_ERR_DIVISION_BY_ZERO";
        case TF_ERR_DIVISION_NOT_CLEAN: return "TF_ERR_DIVISION_NOT_CLEAN";
        default: return "TF_ERR_UNKNOWN";
    }
}

const char* tf_last_error(void) { return t_last_error.c_str(); }
int tf_version(void) { return 1000; }

int tf_release_caches(void) {
    DeviceCtx* ctx = nullptr;
    const int rc = current_ctx(&ctx);
    if (rc) return rc;
    return release_caches(ctx);
}

int tf_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

void tf_set_ntt_tile_bytes(size_t bytes) {
    read_env();
    g_tile_bytes = bytes ? bytes : (size_t(2048) << 20);
}
size_t tf_get_ntt_tile_bytes(void) {
    read_env();
    return g_tile_bytes;
}
void tf_set_batch_eval_route(int route) { g_batch_eval_route.store(route == 1 || route == 2 ? route : 0, std::memory_order_relaxed); }
void tf_set_ntt_nt(int mask) {
    read_env();
    g_nt.store(mask & 3, std::memory_order_relaxed);
}
void tf_set_ntt_pipe(int streams) {
    read_env();
    g_pipe.store(std::min(std::max(streams, 1), kMaxPipe), std::memory_order_relaxed);
}
int tf_get_ntt_pipe(void) {
    read_env();
    return g_pipe.load(std::memory_order_relaxed);
}

// measurement helper (not part of the drop-in boundary): the shader clock the GPU is running at right now, from the ratio of
// the shader-cycle counter to the constant-rate wall clock over a ~0.5 ms spin of one wave.  bench.py records it next to its
// timings so that a run taken while the GPU sits in a low power state can be told from a slow kernel.
double tf_debug_sclk_mhz(void) {
    int dev = 0, wall_khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1.0;
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
    unsigned long long* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), 2 * sizeof(unsigned long long)) != hipSuccess) return -1.0;
    hipLaunchKernelGGL(tfk::sclk_probe_kernel, dim3(1), dim3(64), 0, hipStream_t(0), d);
    unsigned long long h[2] = {0, 0};
    const hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess || h[1] == 0) return -1.0;
    return (double)h[0] / (double)h[1] * (double)wall_khz / 1000.0;
}

// synthetic-input helper (not part of the drop-in boundary): d_out[i] = new(splitmix64(seed ^ (first_index + i)) mod p)
int tf_debug_fill_random_dev(uint64_t* d_out, size_t count, uint64_t seed, uint64_t first_index, void* stream) {
    if (count == 0) return TF_OK;
    if (!d_out) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    const unsigned blocks = (unsigned)std::min<size_t>((count + 255) / 256, size_t(1) << 20);
    hipLaunchKernelGGL(tfk::fill_random_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_out,
                       (unsigned long long)count, (u64)seed, (unsigned long long)first_index);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// measurement helper (not part of the drop-in boundary): allocate / fetch the MODE-3 stamp buffer
int tf_debug_stamps(unsigned long long* host_out, size_t words) {
    if (!g_dbg_buf) {
        if (hipMalloc(reinterpret_cast<void**>(&g_dbg_buf), 4096 * 8 * 6 * 8) != hipSuccess) return TF_ERR_HIP;
        (void)hipMemset(g_dbg_buf, 0, 4096 * 8 * 6 * 8);
    }
    if (host_out && words) {
        if (hipDeviceSynchronize() != hipSuccess) return TF_ERR_HIP;
        if (hipMemcpy(host_out, g_dbg_buf, std::min<size_t>(words, 4096 * 8 * 6) * 8, hipMemcpyDeviceToHost) != hipSuccess) return TF_ERR_HIP;
    }
    return TF_OK;
}

void tf_set_ntt_min_passes(int passes) { g_min_passes.store(passes, std::memory_order_relaxed); }
void tf_set_ntt_chain(int tiles_per_workgroup) { g_chain.store(std::max(0, tiles_per_workgroup), std::memory_order_relaxed); }
void tf_set_ntt_latency_kernel(int mode) { g_lat_mode.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }
void tf_set_ntt_two_pass(int mode) { g_pre2_mode.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }
void tf_set_ntt_small_launch(int mode) { g_small_launch_mode.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }
// The plan of one transform: number of global passes and log2 of each pass's radix (planner introspection for the CPU tests).
// The route tf_poly_batch_evaluate_* takes for this shape: 1 Horner, 2 zerofier tree; 0 for a width that is not 1 / 3.  Host logic
// only (no device is touched), so the CPU tests pin the router.
int tf_batch_eval_plan(size_t n_coeffs, size_t n_points, size_t batch, int width) {
    if (width != 1 && width != 3) return 0;
    return tree_route(n_coeffs, n_points, batch, width) ? 2 : 1;
}
int tf_ntt_plan(size_t n, int width, int* log2_radix_out) {
    if (check_len(n) || n <= 1 || (width != 1 && width != 3) || !log2_radix_out) return 0;
    const int log_n = ilog2(n);
    for (int i = 0; i < 4; ++i) log2_radix_out[i] = 0;
    static const bool no_block = getenv("TF_NTT_NO_BLOCK") != nullptr;
    static const bool no_xfe_block = getenv("TF_NTT_NO_XFE_BLOCK") != nullptr;
    if (log_n <= 10 || (log_n <= (width == 1 ? 14 : (no_xfe_block ? 10 : 12)) && !no_block && g_min_passes.load(std::memory_order_relaxed) == 0)) {
        log2_radix_out[0] = log_n;
        return 1;
    }
    int P = pass_count(log_n);
    int a[4] = {0, 0, 0, 0};
    choose_split(log_n, P, width, a);
    // a plain transform of 2^21 / 2^22 points with enough work for the wide tiles: two passes, 2^11 = a pass of paired 1024-point halves
    if (pre2_plan_ok(log_n, width, n, 1, false, -1, false, false, false)) P = 2, pre2_split(log_n, a);
    for (int i = 0; i < P; ++i) log2_radix_out[i] = a[i];
    return P;
}

int tf_ntt_launch_count(size_t n, size_t batch, int width) {
    if (check_len(n) || n <= 1 || batch == 0 || (width != 1 && width != 3)) return 0;
    const int log_n = ilog2(n);
    if (log_n <= 10) return (int)((batch + (size_t(1) << 24) - 1) >> 24);
    int radix[4];
    if (tf_ntt_plan(n, width, radix) == 1) return 1;  // whole transform per workgroup (BFE 2^11 .. 2^14)
    read_env();
    const size_t poly_bytes = n * size_t(width) * sizeof(u64);
    size_t tb = std::min(std::max<size_t>(1, g_tile_bytes / poly_bytes), batch);
    const size_t tiles = (batch + tb - 1) / tb;
    int P = pass_count(log_n);
    if (P == 4) return (int)(tiles * 3 + batch);  // the last pass of a four-pass plan is launched per polynomial
    const bool small_call = (unsigned long long)n * batch * width <= (1ull << 21) && g_small_launch_mode.load(std::memory_order_relaxed) != 0;
    if (!small_call && pre2_plan_ok(log_n, width, n, 1, false, -1, false, false, false)) P = 2;
    return (int)(tiles * P);
}

int tf_ntt_bfe(uint64_t* x, size_t n, size_t batch, int inverse) { return ntt_host(x, n, batch, 1, inverse); }
int tf_ntt_xfe(uint64_t* x, size_t n, size_t batch, int inverse) { return ntt_host(x, n, batch, 3, inverse); }
int tf_ntt_bfe_dev(uint64_t* d_x, size_t n, size_t batch, int inverse, void* stream) {
    return ntt_dev(d_x, n, batch, 1, inverse, stream);
}
int tf_ntt_xfe_dev(uint64_t* d_x, size_t n, size_t batch, int inverse, void* stream) {
    return ntt_dev(d_x, n, batch, 3, inverse, stream);
}

int tf_coset_eval_bfe(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch) {
    return coset_eval_host(c, nc, off, out, order, batch, 1);
}
int tf_coset_eval_xfe(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch) {
    return coset_eval_host(c, nc, off, out, order, batch, 3);
}
int tf_coset_eval_bfe_dev(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch,
                          void* stream) {
    return coset_eval_dev(c, nc, off, out, order, batch, 1, stream);
}
int tf_coset_eval_xfe_dev(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch,
                          void* stream) {
    return coset_eval_dev(c, nc, off, out, order, batch, 3, stream);
}

int tf_tip5_permute_dev(uint64_t* d_states, size_t count, void* stream) { return tip5_permute_dev(d_states, count, stream); }
int tf_tip5_hash_pairs_dev(const uint64_t* d_in, uint64_t* d_out, size_t count, void* stream) {
    return tip5_hash_pairs_dev(d_in, d_out, count, stream);
}
int tf_tip5_hash_varlen_rows_dev(const uint64_t* d_rows, size_t row_len, size_t n_rows, uint64_t* d_out, void* stream) {
    return tip5_hash_varlen_rows_dev(d_rows, row_len, n_rows, d_out, stream);
}
int tf_merkle_build_dev(const uint64_t* d_leaves, size_t n, uint64_t* d_nodes, size_t batch, void* stream) {
    return merkle_build_dev(d_leaves, n, d_nodes, batch, stream);
}
int tf_merkle_root_dev(const uint64_t* d_leaves, size_t n, uint64_t* d_root, size_t batch, void* stream) {
    return merkle_root_dev(d_leaves, n, d_root, batch, stream);
}

int tf_tip5_trace_dev(uint64_t* d_states, uint64_t* d_trace, size_t count, void* stream) { return tip5_trace_dev(d_states, d_trace, count, stream); }
int tf_tip5_trace(uint64_t* states, uint64_t* trace, size_t count) {
    if (count == 0) return TF_OK;
    if (!states || !trace) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf d(s), t(s);
    TRY(d.alloc(count * 16));
    TRY(t.alloc(count * 96));
    TRY(h2d(d.p, states, count * 16, s));
    TRY(tip5_trace_dev(d.p, t.p, count, s));
    TRY(d2h(states, d.p, count * 16, s));
    TRY(d2h(trace, t.p, count * 96, s));
    return sync(s);
}
int tf_tip5_permute(uint64_t* states, size_t count) {
    if (count == 0) return TF_OK;
    if (!states) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf d(s);
    TRY(d.alloc(count * 16));
    TRY(h2d(d.p, states, count * 16, s));
    TRY(tip5_permute_dev(d.p, count, s));
    TRY(d2h(states, d.p, count * 16, s));
    return sync(s);
}

int tf_tip5_hash_pairs(const uint64_t* in, uint64_t* out, size_t count) {
    if (count == 0) return TF_OK;
    if (!in || !out) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf din(s), dout(s);
    TRY(din.alloc(count * 10));
    TRY(dout.alloc(count * 5));
    TRY(h2d(din.p, in, count * 10, s));
    TRY(tip5_hash_pairs_dev(din.p, dout.p, count, s));
    TRY(d2h(out, dout.p, count * 5, s));
    return sync(s);
}

int tf_tip5_hash_varlen_rows(const uint64_t* rows, size_t row_len, size_t n_rows, uint64_t* out) {
    if (n_rows == 0) return TF_OK;
    if (!out || (row_len && !rows)) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf din(s), dout(s);
    TRY(din.alloc(std::max<size_t>(1, n_rows * row_len)));
    TRY(dout.alloc(n_rows * 5));
    TRY(h2d(din.p, rows, n_rows * row_len, s));
    TRY(tip5_hash_varlen_rows_dev(din.p, row_len, n_rows, dout.p, s));
    TRY(d2h(out, dout.p, n_rows * 5, s));
    return sync(s);
}

int tf_merkle_build(const uint64_t* leaves, size_t n, uint64_t* nodes_out, size_t batch) {
    TRY(check_leaves(n));
    if (batch == 0) return TF_OK;
    if (!leaves || !nodes_out) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf din(s), dout(s);
    if (din.alloc(n * batch * 5) || dout.alloc(n * batch * 10)) return TF_ERR_TREE_TOO_HIGH;  // merkle_tree.rs:405-410
    TRY(h2d(din.p, leaves, n * batch * 5, s));
    TRY(merkle_build_dev(din.p, n, dout.p, batch, s));
    TRY(d2h(nodes_out, dout.p, n * batch * 10, s));
    return sync(s);
}

int tf_merkle_root(const uint64_t* leaves, size_t n, uint64_t* root_out, size_t batch) {
    TRY(check_leaves(n));
    if (batch == 0) return TF_OK;
    if (!leaves || !root_out) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf din(s), dout(s);
    if (din.alloc(n * batch * 5)) return TF_ERR_TREE_TOO_HIGH;
    TRY(dout.alloc(batch * 5));
    TRY(h2d(din.p, leaves, n * batch * 5, s));
    TRY(merkle_root_dev(din.p, n, dout.p, batch, s));
    TRY(d2h(root_out, dout.p, batch * 5, s));
    return sync(s);
}

// ---- SURVEY 8(f1)-(f3) ---------------------------------------------------------------------------
int tf_coset_interpolate_bfe_dev(const uint64_t* v, size_t n, uint64_t off, uint64_t* out, size_t batch, void* stream) {
    return coset_interp_dev(v, n, off, out, batch, 1, stream);
}
int tf_coset_interpolate_xfe_dev(const uint64_t* v, size_t n, uint64_t off, uint64_t* out, size_t batch, void* stream) {
    return coset_interp_dev(v, n, off, out, batch, 3, stream);
}
int tf_hadamard_bfe_dev(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count, void* stream) {
    return hadamard_dev(a, b, out, count, 1, stream);
}
int tf_hadamard_xfe_dev(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count, void* stream) {
    return hadamard_dev(a, b, out, count, 3, stream);
}
int tf_poly_mul_bfe_dev(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t batch, void* stream) {
    return poly_mul_dev(a, na, b, nb, out, batch, 1, stream);
}
int tf_poly_mul_xfe_dev(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t batch, void* stream) {
    return poly_mul_dev(a, na, b, nb, out, batch, 3, stream);
}
int tf_poly_square_bfe_dev(const uint64_t* a, size_t na, uint64_t* out, size_t batch, void* stream) {
    return poly_square_dev(a, na, out, batch, 1, stream);
}
int tf_poly_square_xfe_dev(const uint64_t* a, size_t na, uint64_t* out, size_t batch, void* stream) {
    return poly_square_dev(a, na, out, batch, 3, stream);
}
int tf_lde_bfe_dev(const uint64_t* v, size_t n, uint64_t off_in, uint64_t* out, size_t m, uint64_t off_out, size_t batch, void* stream) {
    return lde_dev(v, n, off_in, out, m, off_out, batch, 1, stream);
}
int tf_lde_xfe_dev(const uint64_t* v, size_t n, uint64_t off_in, uint64_t* out, size_t m, uint64_t off_out, size_t batch, void* stream) {
    return lde_dev(v, n, off_in, out, m, off_out, batch, 3, stream);
}
int tf_poly_batch_evaluate_bfe_dev(const uint64_t* c, size_t nc, const uint64_t* pts, size_t np, uint64_t* out, void* stream) {
    return batch_evaluate_dev(c, nc, nc, 1, pts, np, out, 1, stream);
}
int tf_poly_batch_evaluate_xfe_dev(const uint64_t* c, size_t nc, const uint64_t* pts, size_t np, uint64_t* out, void* stream) {
    return batch_evaluate_dev(c, nc, 3 * nc, 1, pts, np, out, 3, stream);
}
int tf_coset_extrapolate_bfe_dev(uint64_t offset, const uint64_t* cw, size_t n, size_t batch, const uint64_t* pts, size_t np,
                                 uint64_t* out, void* stream) {
    return coset_extrapolate_dev(offset, cw, n, batch, pts, np, out, 1, stream);
}
int tf_coset_extrapolate_xfe_dev(uint64_t offset, const uint64_t* cw, size_t n, size_t batch, const uint64_t* pts, size_t np,
                                 uint64_t* out, void* stream) {
    return coset_extrapolate_dev(offset, cw, n, batch, pts, np, out, 3, stream);
}
int tf_tip5_hash_table_rows_dev(const uint64_t* d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t* d_digests,
                                size_t batch, void* stream) {
    return hash_table_rows_dev(d_table, n_rows, n_cols, width, col_stride, d_digests, batch, stream);
}
int tf_merkle_from_columns_dev(const uint64_t* d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t* d_nodes,
                               size_t batch, void* stream) {
    return merkle_from_columns_dev(d_table, n_rows, n_cols, width, col_stride, d_nodes, batch, stream);
}
int tf_merkle_from_rows_dev(const uint64_t* d_rows, size_t row_len, size_t n_rows, uint64_t* d_nodes, size_t batch, void* stream) {
    return merkle_from_rows_dev(d_rows, row_len, n_rows, d_nodes, batch, stream);
}

int tf_coset_interpolate_bfe(const uint64_t* v, size_t n, uint64_t off, uint64_t* out, size_t batch) {
    TRY(check_len(n));
    if (n == 0 || batch == 0) return TF_OK;
    if (!v || !out) return TF_ERR_NULL_POINTER;
    if (off == 0) return TF_ERR_INVERSE_OF_ZERO;
    return host_roundtrip(v, n * batch, nullptr, 0, out, n * batch,
                          [&](u64* a, u64*, u64* o, hipStream_t s) { return coset_interp_dev(a, n, off, o, batch, 1, s); });
}
int tf_coset_interpolate_xfe(const uint64_t* v, size_t n, uint64_t off, uint64_t* out, size_t batch) {
    TRY(check_len(n));
    if (n == 0 || batch == 0) return TF_OK;
    if (!v || !out) return TF_ERR_NULL_POINTER;
    if (off == 0) return TF_ERR_INVERSE_OF_ZERO;
    return host_roundtrip(v, 3 * n * batch, nullptr, 0, out, 3 * n * batch,
                          [&](u64* a, u64*, u64* o, hipStream_t s) { return coset_interp_dev(a, n, off, o, batch, 3, s); });
}
int tf_poly_mul_bfe(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t batch) {
    if (batch == 0 || na == 0 || nb == 0) return TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, na * batch, b, nb * batch, out, (na + nb - 1) * batch,
                          [&](u64* x, u64* y, u64* o, hipStream_t s) { return poly_mul_dev(x, na, y, nb, o, batch, 1, s); });
}
int tf_poly_mul_xfe(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t batch) {
    if (batch == 0 || na == 0 || nb == 0) return TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, 3 * na * batch, b, 3 * nb * batch, out, 3 * (na + nb - 1) * batch,
                          [&](u64* x, u64* y, u64* o, hipStream_t s) { return poly_mul_dev(x, na, y, nb, o, batch, 3, s); });
}
int tf_poly_square_bfe(const uint64_t* a, size_t na, uint64_t* out, size_t batch) {
    if (batch == 0 || na == 0) return TF_OK;
    if (!a || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, na * batch, nullptr, 0, out, (2 * na - 1) * batch,
                          [&](u64* x, u64*, u64* o, hipStream_t s) { return poly_square_dev(x, na, o, batch, 1, s); });
}
int tf_poly_square_xfe(const uint64_t* a, size_t na, uint64_t* out, size_t batch) {
    if (batch == 0 || na == 0) return TF_OK;
    if (!a || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, 3 * na * batch, nullptr, 0, out, 3 * (2 * na - 1) * batch,
                          [&](u64* x, u64*, u64* o, hipStream_t s) { return poly_square_dev(x, na, o, batch, 3, s); });
}
int tf_poly_batch_evaluate_bfe(const uint64_t* c, size_t nc, const uint64_t* pts, size_t np, uint64_t* out) {
    if (np == 0) return TF_OK;
    if (!pts || !out || (nc && !c)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(c, nc, pts, np, out, np,
                          [&](u64* dc, u64* dp, u64* o, hipStream_t s) { return batch_evaluate_dev(dc, nc, nc, 1, dp, np, o, 1, s); });
}
int tf_poly_batch_evaluate_xfe(const uint64_t* c, size_t nc, const uint64_t* pts, size_t np, uint64_t* out) {
    if (np == 0) return TF_OK;
    if (!pts || !out || (nc && !c)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(c, 3 * nc, pts, 3 * np, out, 3 * np,
                          [&](u64* dc, u64* dp, u64* o, hipStream_t s) { return batch_evaluate_dev(dc, nc, 3 * nc, 1, dp, np, o, 3, s); });
}
int tf_poly_zerofier_bfe_dev(const uint64_t* r, size_t n, uint64_t* out, void* stream) { return zerofier_dev(r, n, out, 1, stream); }
int tf_poly_zerofier_xfe_dev(const uint64_t* r, size_t n, uint64_t* out, void* stream) { return zerofier_dev(r, n, out, 3, stream); }
static int zerofier_host(const uint64_t* r, size_t n, uint64_t* out, int L) {
    if (!out || (n && !r)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(r, n * L, nullptr, 0, out, (n + 1) * L, [&](u64* dr, u64*, u64* o, hipStream_t s) { return zerofier_dev(dr, n, o, L, s); });
}
int tf_poly_zerofier_bfe(const uint64_t* r, size_t n, uint64_t* out) { return zerofier_host(r, n, out, 1); }
int tf_poly_zerofier_xfe(const uint64_t* r, size_t n, uint64_t* out) { return zerofier_host(r, n, out, 3); }
int tf_poly_interpolate_bfe_dev(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, void* stream) {
    return interpolate_dev(d, v, n, rows, out, 1, stream);
}
int tf_poly_interpolate_xfe_dev(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, void* stream) {
    return interpolate_dev(d, v, n, rows, out, 3, stream);
}
static int interpolate_host(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, int L) {
    if (n == 0) return TF_ERR_EMPTY_DOMAIN;
    if (rows == 0) return TF_OK;
    if (!d || !v || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(d, n * L, v, rows * n * L, out, rows * n * L,
                          [&](u64* dd, u64* dv, u64* o, hipStream_t s) { return interpolate_dev(dd, dv, n, rows, o, L, s); });
}
int tf_poly_interpolate_bfe(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out) { return interpolate_host(d, v, n, rows, out, 1); }
int tf_poly_interpolate_xfe(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out) { return interpolate_host(d, v, n, rows, out, 3); }
static int tree_new_any(const uint64_t* domain, size_t n, int L, bool on_device, void* stream, tf_zerofier_tree** tree) {
    if (!tree) return TF_ERR_NULL_POINTER;
    *tree = nullptr;
    TreeHandle* H = nullptr;
    int rc;
    if (on_device) {
        rc = tree_handle_new(domain, n, L, stream, &H);
    } else {
        if (n && !domain) return TF_ERR_NULL_POINTER;
        DeviceCtx* ctx = nullptr;
        rc = current_ctx(&ctx);
        if (rc) return rc;
        hipStream_t s = host_stream();
        DevBuf d(s);
        rc = d.alloc(n * L);
        if (!rc) rc = h2d(d.p, domain, n * L, s);
        if (!rc) rc = tree_handle_new(d.p, n, L, s, &H);
        if (!rc) rc = sync(s);
    }
    if (rc) return rc;
    *tree = reinterpret_cast<tf_zerofier_tree*>(H);
    return TF_OK;
}
static int tree_new_async(const uint64_t* d_domain, size_t n, int L, void* stream, tf_zerofier_tree** tree) {
    if (!tree) return TF_ERR_NULL_POINTER;
    *tree = nullptr;
    TreeHandle* H = nullptr;
    const int rc = tree_handle_new(d_domain, n, L, stream, &H, true);
    if (rc) return rc;
    *tree = reinterpret_cast<tf_zerofier_tree*>(H);
    return TF_OK;
}
static TreeHandle* tree_of(const tf_zerofier_tree* t) { return const_cast<TreeHandle*>(reinterpret_cast<const TreeHandle*>(t)); }
int tf_zerofier_tree_new_bfe(const uint64_t* domain, size_t n, tf_zerofier_tree** tree) { return tree_new_any(domain, n, 1, false, nullptr, tree); }
int tf_zerofier_tree_new_xfe(const uint64_t* domain, size_t n, tf_zerofier_tree** tree) { return tree_new_any(domain, n, 3, false, nullptr, tree); }
int tf_zerofier_tree_new_bfe_dev(const uint64_t* d_domain, size_t n, void* stream, tf_zerofier_tree** tree) {
    return tree_new_any(d_domain, n, 1, true, stream, tree);
}
int tf_zerofier_tree_new_xfe_dev(const uint64_t* d_domain, size_t n, void* stream, tf_zerofier_tree** tree) {
    return tree_new_any(d_domain, n, 3, true, stream, tree);
}
void tf_zerofier_tree_free(tf_zerofier_tree* tree) { tree_handle_free(tree_of(tree)); }
size_t tf_zerofier_tree_num_points(const tf_zerofier_tree* tree) { return tree ? tree_of(tree)->pt.n : 0; }
int tf_zerofier_tree_width(const tf_zerofier_tree* tree) { return tree ? tree_of(tree)->L : 0; }
int tf_zerofier_tree_zerofier_dev(const tf_zerofier_tree* tree, uint64_t* d_out, void* stream) { return tree_handle_zerofier(tree_of(tree), d_out, stream); }
int tf_zerofier_tree_batch_evaluate_dev(const tf_zerofier_tree* tree, const uint64_t* d_coeffs, size_t n_coeffs, size_t batch, uint64_t* d_out,
                                        void* stream) {
    return tree_handle_batch_evaluate(tree_of(tree), d_coeffs, n_coeffs, batch, d_out, stream);
}
int tf_zerofier_tree_interpolate_dev(tf_zerofier_tree* tree, const uint64_t* d_values, size_t rows, uint64_t* d_out, void* stream) {
    return tree_handle_interpolate(tree_of(tree), d_values, rows, d_out, stream);
}
// ---- enqueue-and-return variants (tf_hip.h): panic cases go to *d_status on the device, nothing synchronises
int tf_poly_interpolate_bfe_dev_async(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, void* stream, int* d_status) {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return interpolate_dev(d, v, n, rows, out, 1, stream, d_status);
}
int tf_poly_interpolate_xfe_dev_async(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, void* stream, int* d_status) {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return interpolate_dev(d, v, n, rows, out, 3, stream, d_status);
}
int tf_poly_clean_divide_bfe_dev_async(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, void* stream, int* d_status) {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return clean_divide_dev(a, na, b, nb, out, stream, 1, d_status);
}
int tf_poly_clean_divide_many_bfe_dev_async(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out, void* stream,
                                            int* d_status) {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return clean_divide_dev(a, na, b, nb, out, stream, batch, d_status);
}
int tf_zerofier_tree_new_bfe_dev_async(const uint64_t* d_domain, size_t n, void* stream, tf_zerofier_tree** tree) {
    return tree_new_async(d_domain, n, 1, stream, tree);
}
int tf_zerofier_tree_new_xfe_dev_async(const uint64_t* d_domain, size_t n, void* stream, tf_zerofier_tree** tree) {
    return tree_new_async(d_domain, n, 3, stream, tree);
}
int tf_zerofier_tree_interpolate_dev_async(tf_zerofier_tree* tree, const uint64_t* d_values, size_t rows, uint64_t* d_out, void* stream, int* d_status) {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return tree_handle_interpolate(tree_of(tree), d_values, rows, d_out, stream, d_status);
}
int tf_zerofier_tree_zerofier(const tf_zerofier_tree* tree, uint64_t* out) {
    if (!tree || !out) return TF_ERR_NULL_POINTER;
    TreeHandle* H = tree_of(tree);
    return host_roundtrip(nullptr, 0, nullptr, 0, out, (H->pt.n + 1) * H->L, [&](u64*, u64*, u64* o, hipStream_t s) { return tree_handle_zerofier(H, o, s); });
}
int tf_zerofier_tree_batch_evaluate(const tf_zerofier_tree* tree, const uint64_t* coeffs, size_t n_coeffs, size_t batch, uint64_t* out) {
    if (!tree) return TF_ERR_NULL_POINTER;
    TreeHandle* H = tree_of(tree);
    if (H->pt.n == 0 || batch == 0) return TF_OK;
    if (!out || (n_coeffs && !coeffs)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(coeffs, batch * n_coeffs * H->L, nullptr, 0, out, batch * H->pt.n * H->L,
                          [&](u64* dc, u64*, u64* o, hipStream_t s) { return tree_handle_batch_evaluate(H, dc, n_coeffs, batch, o, s); });
}
int tf_zerofier_tree_interpolate(tf_zerofier_tree* tree, const uint64_t* values, size_t rows, uint64_t* out) {
    if (!tree) return TF_ERR_NULL_POINTER;
    TreeHandle* H = tree_of(tree);
    if (H->pt.n == 0) return TF_ERR_EMPTY_DOMAIN;
    if (rows == 0) return TF_OK;
    if (!values || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(values, rows * H->pt.n * H->L, nullptr, 0, out, rows * H->pt.n * H->L,
                          [&](u64* dv, u64*, u64* o, hipStream_t s) { return tree_handle_interpolate(H, dv, rows, o, s); });
}
int tf_coset_eval_xfe_xoffset_dev(const uint64_t* c, size_t nc, const uint64_t offset[3], uint64_t* out, size_t order, size_t batch, void* stream) {
    return coset_eval_xoffset_dev(c, nc, offset, out, order, batch, stream);
}
int tf_coset_interpolate_xfe_xoffset_dev(const uint64_t* v, size_t n, const uint64_t offset[3], uint64_t* out, size_t batch, void* stream) {
    return coset_interp_xoffset_dev(v, n, offset, out, batch, stream);
}
int tf_coset_eval_xfe_xoffset(const uint64_t* c, size_t nc, const uint64_t offset[3], uint64_t* out, size_t order, size_t batch) {
    if (nc > order) return TF_ERR_ORDER_NOT_ABOVE_DEGREE;
    int rc = check_len(order);
    if (rc) return rc;
    if (order == 0 || batch == 0) return TF_OK;
    if (!out || !offset || (nc && !c)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(c, 3 * nc * batch, nullptr, 0, out, 3 * order * batch,
                          [&](u64* dc, u64*, u64* o, hipStream_t s) { return coset_eval_xoffset_dev(dc, nc, offset, o, order, batch, s); });
}
int tf_coset_interpolate_xfe_xoffset(const uint64_t* v, size_t n, const uint64_t offset[3], uint64_t* out, size_t batch) {
    int rc = check_len(n);
    if (rc) return rc;
    if (n == 0 || batch == 0) return TF_OK;
    if (!v || !out || !offset) return TF_ERR_NULL_POINTER;
    return host_roundtrip(v, 3 * n * batch, nullptr, 0, out, 3 * n * batch,
                          [&](u64* dv, u64*, u64* o, hipStream_t s) { return coset_interp_xoffset_dev(dv, n, offset, o, batch, s); });
}
int tf_barycentric_evaluate_bfe_dev(const uint64_t* cw, size_t n, size_t batch, const uint64_t x[3], uint64_t* out, void* stream) {
    return barycentric_dev(cw, n, batch, 1, x, out, stream);
}
int tf_barycentric_evaluate_xfe_dev(const uint64_t* cw, size_t n, size_t batch, const uint64_t x[3], uint64_t* out, void* stream) {
    return barycentric_dev(cw, n, batch, 3, x, out, stream);
}
static int barycentric_host(const uint64_t* cw, size_t n, size_t batch, int width, const uint64_t x[3], uint64_t* out) {
    int rc = check_len(n);
    if (rc) return rc;
    if (!x) return TF_ERR_NULL_POINTER;
    if (n == 0) return TF_ERR_INVERSE_OF_ZERO;
    if (batch == 0) return barycentric_dev(nullptr, n, 0, width, x, nullptr, nullptr);  // the argument checks alone
    if (!cw || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(cw, batch * n * width, nullptr, 0, out, 3 * batch,
                          [&](u64* dc, u64*, u64* o, hipStream_t s) { return barycentric_dev(dc, n, batch, width, x, o, s); });
}
int tf_barycentric_evaluate_bfe(const uint64_t* cw, size_t n, size_t batch, const uint64_t x[3], uint64_t* out) { return barycentric_host(cw, n, batch, 1, x, out); }
int tf_barycentric_evaluate_xfe(const uint64_t* cw, size_t n, size_t batch, const uint64_t x[3], uint64_t* out) { return barycentric_host(cw, n, batch, 3, x, out); }
// Polynomial<BFieldElement>::evaluate::<XFieldElement, XFieldElement> (polynomial.rs:309-320) for `batch` polynomials at n_points
// extension-field points: out[(b * n_points + i) * 3] = f_b(points[i]).  Horner (the shape of the use: every column polynomial
// of a table at a few out-of-domain points).
int tf_poly_evaluate_bfe_at_xfe_dev(const uint64_t* c, size_t nc, size_t batch, const uint64_t* pts, size_t np, uint64_t* out, void* stream) {
    if (np == 0 || batch == 0) return TF_OK;
    if (!pts || !out || (nc && !c)) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    return batch_evaluate_horner(c, nc, nc, batch, pts, np, out, 3, stream, 1);
}
int tf_poly_evaluate_bfe_at_xfe(const uint64_t* c, size_t nc, size_t batch, const uint64_t* pts, size_t np, uint64_t* out) {
    if (np == 0 || batch == 0) return TF_OK;
    if (!pts || !out || (nc && !c)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(c, batch * nc, pts, 3 * np, out, 3 * batch * np,
                          [&](u64* dc, u64* dp, u64* o, hipStream_t s) { return batch_evaluate_horner(dc, nc, nc, batch, dp, np, o, 3, s, 1); });
}
int tf_poly_clean_divide_bfe_dev(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, void* stream) {
    return clean_divide_dev(a, na, b, nb, out, stream);
}
int tf_poly_mul_shared_bfe_dev(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out, void* stream) {
    return poly_mul_shared_dev(a, na, batch, b, nb, out, 1, stream);
}
int tf_poly_mul_shared_xfe_dev(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out, void* stream) {
    return poly_mul_shared_dev(a, na, batch, b, nb, out, 3, stream);
}
int tf_poly_clean_divide_many_bfe_dev(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out, void* stream) {
    return clean_divide_dev(a, na, b, nb, out, stream, batch);
}
int tf_poly_clean_divide_many_bfe(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out) {
    if (nb == 0) return TF_ERR_DIVISION_BY_ZERO;
    if (na < nb) return na ? TF_ERR_DIVISION_NOT_CLEAN : TF_OK;
    if (batch == 0) return TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, batch * na, b, nb, out, batch * (na - nb + 1),
                          [&](u64* da, u64* db, u64* o, hipStream_t s) { return clean_divide_dev(da, na, db, nb, o, s, batch); });
}
int tf_poly_clean_divide_bfe(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out) {
    if (nb == 0) return TF_ERR_DIVISION_BY_ZERO;
    if (na < nb) return na ? TF_ERR_DIVISION_NOT_CLEAN : TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, na, b, nb, out, na - nb + 1, [&](u64* da, u64* db, u64* o, hipStream_t s) { return clean_divide_dev(da, na, db, nb, o, s); });
}
static int coset_extrapolate_host(uint64_t offset, const uint64_t* cw, size_t n, size_t batch, const uint64_t* pts, size_t np,
                                  uint64_t* out, int L) {
    if (n == 0) return TF_ERR_LEN_NOT_POWER_OF_TWO;
    int rc = check_len(n);
    if (rc) return rc;
    if (batch == 0 || np == 0) return TF_OK;
    if (!cw || !pts || !out) return TF_ERR_NULL_POINTER;
    if (offset == 0) return TF_ERR_INVERSE_OF_ZERO;
    return host_roundtrip(cw, batch * n * L, pts, np * L, out, batch * np * L, [&](u64* dc, u64* dp, u64* o, hipStream_t s) {
        return coset_extrapolate_dev(offset, dc, n, batch, dp, np, o, L, s);
    });
}
int tf_coset_extrapolate_bfe(uint64_t offset, const uint64_t* cw, size_t n, size_t batch, const uint64_t* pts, size_t np, uint64_t* out) {
    return coset_extrapolate_host(offset, cw, n, batch, pts, np, out, 1);
}
int tf_coset_extrapolate_xfe(uint64_t offset, const uint64_t* cw, size_t n, size_t batch, const uint64_t* pts, size_t np, uint64_t* out) {
    return coset_extrapolate_host(offset, cw, n, batch, pts, np, out, 3);
}
int tf_tip5_hash_table_rows(const uint64_t* table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t* digests,
                            size_t batch) {
    if (width != 1 && width != 3) return TF_ERR_NULL_POINTER;
    if (n_rows == 0 || batch == 0) return TF_OK;
    if (!digests || (n_cols && !table)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(table, batch * n_cols * col_stride, nullptr, 0, digests, batch * n_rows * 5,
                          [&](u64* dt, u64*, u64* o, hipStream_t s) { return hash_table_rows_dev(dt, n_rows, n_cols, width, col_stride, o, batch, s); });
}
int tf_merkle_from_columns(const uint64_t* table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t* nodes_out,
                           size_t batch) {
    if (width != 1 && width != 3) return TF_ERR_NULL_POINTER;
    int rc = check_leaves(n_rows);
    if (rc) return rc;
    if (batch == 0) return TF_OK;
    if (!nodes_out || (n_cols && !table)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(table, batch * n_cols * col_stride, nullptr, 0, nodes_out, batch * n_rows * 10,
                          [&](u64* dt, u64*, u64* o, hipStream_t s) { return merkle_from_columns_dev(dt, n_rows, n_cols, width, col_stride, o, batch, s); });
}
int tf_merkle_from_rows(const uint64_t* rows, size_t row_len, size_t n_rows, uint64_t* nodes_out, size_t batch) {
    TRY(check_leaves(n_rows));
    if (batch == 0) return TF_OK;
    if (!nodes_out || (row_len && !rows)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(rows, n_rows * row_len * batch, nullptr, 0, nodes_out, n_rows * batch * 10,
                          [&](u64* r, u64*, u64* o, hipStream_t s) { return merkle_from_rows_dev(r, row_len, n_rows, o, batch, s); });
}

int tf_merkle_auth_structure_indices(size_t num_leafs, const uint64_t* leaf_indices, size_t k, uint64_t* out_indices,
                                     size_t capacity, size_t* out_count) {
    if ((k && !leaf_indices) || !out_count) return TF_ERR_NULL_POINTER;
    std::vector<unsigned long long> idx;
    TRY(auth_structure_indices(num_leafs, leaf_indices, k, &idx));
    *out_count = idx.size();
    if (!out_indices || capacity == 0) return TF_OK;  // sizing call: only the count
    if (capacity < idx.size()) return TF_ERR_BUFFER_TOO_SMALL;  // nothing is written; *out_count says what is needed
    for (size_t i = 0; i < idx.size(); ++i) out_indices[i] = idx[i];
    return TF_OK;
}

int tf_merkle_authentication_structure_dev(const uint64_t* d_nodes, size_t num_leafs, const uint64_t* leaf_indices, size_t k,
                                           uint64_t* out_digests, size_t capacity_digests, size_t* out_count, void* stream) {
    if (!d_nodes || (k && !leaf_indices) || !out_count) return TF_ERR_NULL_POINTER;
    std::vector<unsigned long long> idx;
    TRY(auth_structure_indices(num_leafs, leaf_indices, k, &idx));
    *out_count = idx.size();
    if (idx.empty() || !out_digests) return TF_OK;
    if (capacity_digests < idx.size()) return TF_ERR_BUFFER_TOO_SMALL;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = static_cast<hipStream_t>(stream);
    DevBuf didx(s), dout(s);
    TRY(didx.alloc(idx.size()));
    TRY(dout.alloc(idx.size() * 5));
    TRY(h2d(didx.p, reinterpret_cast<const u64*>(idx.data()), idx.size(), s));
    const long long total = (long long)idx.size() * 5;
    hipLaunchKernelGGL(tfk::gather_digests_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_nodes,
                       reinterpret_cast<const unsigned long long*>(didx.p), (long long)idx.size(), dout.p);
    HIPCHK(hipGetLastError());
    TRY(d2h(out_digests, dout.p, idx.size() * 5, s));
    return sync(s);
}

}  // extern "C"
