// tf_abi.hip -- the C ABI of libtf_hip.so (include/tf_hip.h): host-pointer wrappers and the extern "C" entry points.
#include "tf_internal.h"
#include "aux_kernels.h"

#include <functional>

namespace tfi {

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

// One private non-blocking stream per (host thread, device) for the host-pointer entry points; destroyed with the thread
// (the workers of tf_*_multi are short-lived threads).
struct HostStreams {
    hipStream_t s[kMaxDevices] = {};
    ~HostStreams() {
        for (int d = 0; d < kMaxDevices; ++d)
            if (s[d]) (void)hipStreamDestroy(s[d]);
    }
};
hipStream_t host_stream() {
    thread_local HostStreams streams;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    if (!streams.s[dev]) {
        if (hipStreamCreateWithFlags(&streams.s[dev], hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            streams.s[dev] = nullptr;
        }
    }
    return streams.s[dev];
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


}  // namespace tfi

using namespace tfi;


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
        case TF_ERR_INVALID_ARGUMENT: return "TF_ERR_INVALID_ARGUMENT";
        case TF_ERR_INTERNAL: return "TF_ERR_INTERNAL";
        default: return "TF_ERR_UNKNOWN";
    }
}

const char* tf_last_error(void) { return t_last_error.c_str(); }
int tf_version(void) { return 1001; }
#ifndef TF_SOURCE_HASH
#define TF_SOURCE_HASH "unknown"
#endif
const char* tf_source_hash(void) { return TF_SOURCE_HASH; }

int tf_release_caches(void) try {
    DeviceCtx* ctx = nullptr;
    const int rc = current_ctx(&ctx);
    if (rc) return rc;
    return release_caches(ctx);
} TF_ABI_CATCH

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
#ifdef TF_AB_BUILD
void tf_set_ntt_nt(int mask) {
    read_env();
    g_nt.store(mask & 3, std::memory_order_relaxed);
}
#endif
void tf_set_ntt_pipe(int streams) {
    read_env();
    g_pipe.store(std::min(std::max(streams, 0), kMaxPipe), std::memory_order_relaxed);  // 0 = automatic
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
int tf_debug_fill_random_dev(uint64_t* d_out, size_t count, uint64_t seed, uint64_t first_index, void* stream) try {
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
} TF_ABI_CATCH

#ifdef TF_AB_BUILD
// measurement helper (not part of the drop-in boundary): allocate / fetch the MODE-3 stamp buffer
int tf_debug_stamps(unsigned long long* host_out, size_t words) try {
    if (!g_dbg_buf) {
        if (hipMalloc(reinterpret_cast<void**>(&g_dbg_buf), 4096 * 8 * 6 * 8) != hipSuccess) return TF_ERR_HIP;
        (void)hipMemset(g_dbg_buf, 0, 4096 * 8 * 6 * 8);
    }
    if (host_out && words) {
        if (hipDeviceSynchronize() != hipSuccess) return TF_ERR_HIP;
        if (hipMemcpy(host_out, g_dbg_buf, std::min<size_t>(words, 4096 * 8 * 6) * 8, hipMemcpyDeviceToHost) != hipSuccess) return TF_ERR_HIP;
    }
    return TF_OK;
} TF_ABI_CATCH
#endif

void tf_set_ntt_min_passes(int passes) { g_min_passes.store(passes, std::memory_order_relaxed); }
#ifdef TF_AB_BUILD
void tf_set_ntt_chain(int tiles_per_workgroup) { g_chain.store(std::max(0, tiles_per_workgroup), std::memory_order_relaxed); }
#endif
void tf_set_ntt_latency_kernel(int mode) { g_lat_mode.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }
void tf_set_ntt_two_pass(int mode) { g_pre2_mode.store(mode < 0 ? -1 : (mode > 3 ? 1 : mode), std::memory_order_relaxed); }
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
    static const bool no_block = ab_env("TF_NTT_NO_BLOCK") != nullptr;
    static const bool no_xfe_block = ab_env("TF_NTT_NO_XFE_BLOCK") != nullptr;
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
    // the same predicates, in the same order, as run_ntt (tf_ntt.hip) for a plain in-place call
    if (check_len(n) || n <= 1 || batch == 0 || (width != 1 && width != 3)) return 0;
    const int log_n = ilog2(n);
    read_env();
    if (log_n <= 5 || (log_n == 6 && width == 1 && batch >= 64)) return 1;                                 // ntt_rows32w_kernel (a grid-stride walk) / ntt_tiny_kernel
    if (g_min_passes.load(std::memory_order_relaxed) == 0) {
        if (lat_wanted(log_n, batch, width)) return (int)((batch + (size_t(1) << 22) - 1) >> 22);          // ntt_lat_kernel
        if (lat2_wanted(log_n, batch, width)) return 2;                                                    // ntt_lat2_kernel: column pass + last pass
    }
    if (log_n <= 10) return (int)((batch + (size_t(1) << 24) - 1) >> 24);
    int radix[4];
    if (tf_ntt_plan(n, width, radix) == 1) return 1;  // whole transform per workgroup (BFE 2^11 .. 2^14, XFE 2^11 / 2^12)
    const size_t poly_bytes = n * size_t(width) * sizeof(u64);
    size_t tb = std::min(std::max<size_t>(1, g_tile_bytes / poly_bytes), batch);
    const size_t tiles = (batch + tb - 1) / tb;
    int P = pass_count(log_n);
    if (P == 4) return (int)(tiles * 3 + batch);  // the last pass of a four-pass plan is launched per polynomial
    if (!small_launch_for(n, 1, batch, width, -1) && pre2_plan_ok(log_n, width, n, 1, false, -1, false, false, false)) P = 2;
    return (int)(tiles * P);
}

int tf_ntt_bfe(uint64_t* x, size_t n, size_t batch, int inverse) try { return ntt_host(x, n, batch, 1, inverse); } TF_ABI_CATCH
int tf_ntt_xfe(uint64_t* x, size_t n, size_t batch, int inverse) try { return ntt_host(x, n, batch, 3, inverse); } TF_ABI_CATCH
int tf_ntt_bfe_dev(uint64_t* d_x, size_t n, size_t batch, int inverse, void* stream) try {
    return ntt_dev(d_x, n, batch, 1, inverse, stream);
} TF_ABI_CATCH
int tf_ntt_xfe_dev(uint64_t* d_x, size_t n, size_t batch, int inverse, void* stream) try {
    return ntt_dev(d_x, n, batch, 3, inverse, stream);
} TF_ABI_CATCH

int tf_coset_eval_bfe(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch) try {
    return coset_eval_host(c, nc, off, out, order, batch, 1);
} TF_ABI_CATCH
int tf_coset_eval_xfe(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch) try {
    return coset_eval_host(c, nc, off, out, order, batch, 3);
} TF_ABI_CATCH
int tf_coset_eval_bfe_dev(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch,
                          void* stream) try {
    return coset_eval_dev(c, nc, off, out, order, batch, 1, stream);
} TF_ABI_CATCH
int tf_coset_eval_xfe_dev(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch,
                          void* stream) try {
    return coset_eval_dev(c, nc, off, out, order, batch, 3, stream);
} TF_ABI_CATCH

int tf_tip5_permute_dev(uint64_t* d_states, size_t count, void* stream) try { return tip5_permute_dev(d_states, count, stream); } TF_ABI_CATCH
int tf_tip5_hash_pairs_dev(const uint64_t* d_in, uint64_t* d_out, size_t count, void* stream) try {
    return tip5_hash_pairs_dev(d_in, d_out, count, stream);
} TF_ABI_CATCH
int tf_tip5_hash_varlen_rows_dev(const uint64_t* d_rows, size_t row_len, size_t n_rows, uint64_t* d_out, void* stream) try {
    return tip5_hash_varlen_rows_dev(d_rows, row_len, n_rows, d_out, stream);
} TF_ABI_CATCH
int tf_merkle_build_dev(const uint64_t* d_leaves, size_t n, uint64_t* d_nodes, size_t batch, void* stream) try {
    return merkle_build_dev(d_leaves, n, d_nodes, batch, stream);
} TF_ABI_CATCH
int tf_merkle_root_dev(const uint64_t* d_leaves, size_t n, uint64_t* d_root, size_t batch, void* stream) try {
    return merkle_root_dev(d_leaves, n, d_root, batch, stream);
} TF_ABI_CATCH

int tf_tip5_trace_dev(uint64_t* d_states, uint64_t* d_trace, size_t count, void* stream) try { return tip5_trace_dev(d_states, d_trace, count, stream); } TF_ABI_CATCH
int tf_tip5_trace(uint64_t* states, uint64_t* trace, size_t count) try {
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
} TF_ABI_CATCH
int tf_tip5_permute(uint64_t* states, size_t count) try {
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
} TF_ABI_CATCH

int tf_tip5_hash_pairs(const uint64_t* in, uint64_t* out, size_t count) try {
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
} TF_ABI_CATCH

int tf_tip5_hash_varlen_rows(const uint64_t* rows, size_t row_len, size_t n_rows, uint64_t* out) try {
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
} TF_ABI_CATCH

// ---- warm-up: one blocking call per shape, so that the *_dev calls of that shape never leave the stream ---------------------------
// The first call of a shape on a device builds its twiddle / power tables (hipMalloc + a build kernel the host waits for), opens the
// dynamic LDS of the kernels it launches (hipFuncSetAttribute), uploads the Tip5 constants, creates the library's memory pool, side
// streams and scratch blocks.  Some of those synchronise the whole DEVICE.  A caller that drives several GPUs from one host thread
// ("eight GPUs from one thread") calls tf_prepare_* once per device and shape at start-up; every later *_dev call of that
// shape only enqueues (tests/test_gpu_parity.py::test_one_host_thread_round_robin_never_blocks).  Implementation: the shape itself, once,
// on zeroed scratch of the same size (the plan -- passes, tile sizes, tables -- depends on n, batch and width, so nothing smaller is exact).
static int prepare_run(size_t in_words, size_t out_words, const std::function<int(u64*, u64*, hipStream_t)>& body) {
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf a(s), b(s);
    if (a.alloc(in_words) || b.alloc(out_words)) return TF_ERR_OUT_OF_MEMORY;
    if (in_words) HIPCHK(hipMemsetAsync(a.p, 0, in_words * sizeof(u64), s));
    TRY(body(a.p, b.p, s));
    return sync(s);
}
int tf_prepare_ntt(size_t n, size_t batch, int width, int inverse) try {
    if (width != 1 && width != 3) return TF_ERR_INVALID_ARGUMENT;
    TRY(check_len(n));
    if (n <= 1 || batch == 0) return TF_OK;
    return prepare_run(n * batch * (size_t)width, 0, [=](u64* x, u64*, hipStream_t s) { return ntt_dev(x, n, batch, width, inverse, s); });
} TF_ABI_CATCH
int tf_prepare_coset_eval(size_t n_coeffs, uint64_t offset_raw, size_t order, size_t batch, int width) try {
    if (width != 1 && width != 3) return TF_ERR_INVALID_ARGUMENT;
    if (n_coeffs > order) return TF_ERR_ORDER_NOT_ABOVE_DEGREE;
    TRY(check_len(order));
    if (order == 0 || batch == 0) return TF_OK;
    return prepare_run(n_coeffs * batch * (size_t)width, order * batch * (size_t)width,
                       [=](u64* c, u64* o, hipStream_t s) { return coset_eval_dev(c, n_coeffs, offset_raw, o, order, batch, width, s); });
} TF_ABI_CATCH
int tf_prepare_merkle(size_t n_leaves, size_t batch) try {
    TRY(check_leaves(n_leaves));
    if (batch == 0) return TF_OK;
    return prepare_run(n_leaves * batch * 5, n_leaves * batch * 10, [=](u64* l, u64* nd, hipStream_t s) {
        TRY(merkle_build_dev(l, n_leaves, nd, batch, s));
        return merkle_root_dev(l, n_leaves, nd, batch, s);  // (the root-only route has kernels and scratch of its own)
    });
} TF_ABI_CATCH

int tf_merkle_build(const uint64_t* leaves, size_t n, uint64_t* nodes_out, size_t batch) try {
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
} TF_ABI_CATCH

}  // extern "C"  (closed for one internal helper of tf_multi.hip that needs this unit's host-pointer plumbing)
namespace tfi {
// ONE subtree of a host-resident tree on the calling thread's current device (tf_merkle_{build,root}_multi when there are more
// devices than trees): subtree `sub` of `n_sub` of a tree whose node array starts at nodes_tree (heap order, 2 n words-of-5; or null for
// a root-only build).  `leaves_sub` = its m = n / n_sub leaves.  Layer l of the subtree is the run [(n_sub + sub) 2^l, (n_sub + sub + 1) 2^l)
// of the whole tree's array -- the reference's own slicing, util_types/merkle_tree.rs:247-275 (subtrees_mut) -- so the D2H copies put every
// node where par_new would have written it; the subtree's root also goes to root_out (5 words).
int merkle_subtree_host(const u64* leaves_sub, size_t m, u64* nodes_tree, size_t n_sub, size_t sub, u64* root_out) {
    TRY(check_leaves(m));
    if (!leaves_sub || !root_out) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    TRY(current_ctx(&ctx));
    hipStream_t s = host_stream();
    DevBuf din(s), dout(s);
    if (din.alloc(m * 5)) return TF_ERR_TREE_TOO_HIGH;
    TRY(h2d(din.p, leaves_sub, m * 5, s));
    if (!nodes_tree) {
        TRY(dout.alloc(5));
        TRY(merkle_root_dev(din.p, m, dout.p, 1, s));
        TRY(d2h(root_out, dout.p, 5, s));
        return sync(s);
    }
    if (dout.alloc(m * 10)) return TF_ERR_TREE_TOO_HIGH;
    TRY(merkle_build_dev(din.p, m, dout.p, 1, s));
    for (size_t l = 0; (size_t(1) << l) <= m; ++l) {
        const size_t w = size_t(1) << l;  // nodes of layer l: local heap indices [w, 2 w)
        TRY(d2h(nodes_tree + ((n_sub + sub) << l) * 5, dout.p + w * 5, w * 5, s));
    }
    TRY(d2h(root_out, dout.p + 5, 5, s));
    return sync(s);
}
}  // namespace tfi
extern "C" {

int tf_merkle_root(const uint64_t* leaves, size_t n, uint64_t* root_out, size_t batch) try {
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
} TF_ABI_CATCH

// ---- SURVEY 8(f1)-(f3) ---------------------------------------------------------------------------
int tf_coset_interpolate_bfe_dev(const uint64_t* v, size_t n, uint64_t off, uint64_t* out, size_t batch, void* stream) try {
    return coset_interp_dev(v, n, off, out, batch, 1, stream);
} TF_ABI_CATCH
int tf_coset_interpolate_xfe_dev(const uint64_t* v, size_t n, uint64_t off, uint64_t* out, size_t batch, void* stream) try {
    return coset_interp_dev(v, n, off, out, batch, 3, stream);
} TF_ABI_CATCH
int tf_hadamard_bfe_dev(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count, void* stream) try {
    return hadamard_dev(a, b, out, count, 1, stream);
} TF_ABI_CATCH
int tf_hadamard_xfe_dev(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count, void* stream) try {
    return hadamard_dev(a, b, out, count, 3, stream);
} TF_ABI_CATCH
int tf_poly_mul_bfe_dev(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t batch, void* stream) try {
    return poly_mul_dev(a, na, b, nb, out, batch, 1, stream);
} TF_ABI_CATCH
int tf_poly_mul_xfe_dev(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t batch, void* stream) try {
    return poly_mul_dev(a, na, b, nb, out, batch, 3, stream);
} TF_ABI_CATCH
int tf_poly_square_bfe_dev(const uint64_t* a, size_t na, uint64_t* out, size_t batch, void* stream) try {
    return poly_square_dev(a, na, out, batch, 1, stream);
} TF_ABI_CATCH
int tf_poly_square_xfe_dev(const uint64_t* a, size_t na, uint64_t* out, size_t batch, void* stream) try {
    return poly_square_dev(a, na, out, batch, 3, stream);
} TF_ABI_CATCH
int tf_lde_bfe_dev(const uint64_t* v, size_t n, uint64_t off_in, uint64_t* out, size_t m, uint64_t off_out, size_t batch, void* stream) try {
    return lde_dev(v, n, off_in, out, m, off_out, batch, 1, stream);
} TF_ABI_CATCH
int tf_lde_xfe_dev(const uint64_t* v, size_t n, uint64_t off_in, uint64_t* out, size_t m, uint64_t off_out, size_t batch, void* stream) try {
    return lde_dev(v, n, off_in, out, m, off_out, batch, 3, stream);
} TF_ABI_CATCH
int tf_poly_batch_evaluate_bfe_dev(const uint64_t* c, size_t nc, const uint64_t* pts, size_t np, uint64_t* out, void* stream) try {
    return batch_evaluate_dev(c, nc, nc, 1, pts, np, out, 1, stream);
} TF_ABI_CATCH
int tf_poly_batch_evaluate_xfe_dev(const uint64_t* c, size_t nc, const uint64_t* pts, size_t np, uint64_t* out, void* stream) try {
    return batch_evaluate_dev(c, nc, 3 * nc, 1, pts, np, out, 3, stream);
} TF_ABI_CATCH
int tf_coset_extrapolate_bfe_dev(uint64_t offset, const uint64_t* cw, size_t n, size_t batch, const uint64_t* pts, size_t np,
                                 uint64_t* out, void* stream) try {
    return coset_extrapolate_dev(offset, cw, n, batch, pts, np, out, 1, stream);
} TF_ABI_CATCH
int tf_coset_extrapolate_xfe_dev(uint64_t offset, const uint64_t* cw, size_t n, size_t batch, const uint64_t* pts, size_t np,
                                 uint64_t* out, void* stream) try {
    return coset_extrapolate_dev(offset, cw, n, batch, pts, np, out, 3, stream);
} TF_ABI_CATCH
int tf_tip5_hash_table_rows_dev(const uint64_t* d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t* d_digests,
                                size_t batch, void* stream) try {
    return hash_table_rows_dev(d_table, n_rows, n_cols, width, col_stride, d_digests, batch, stream);
} TF_ABI_CATCH
int tf_merkle_from_columns_dev(const uint64_t* d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t* d_nodes,
                               size_t batch, void* stream) try {
    return merkle_from_columns_dev(d_table, n_rows, n_cols, width, col_stride, d_nodes, batch, stream);
} TF_ABI_CATCH
int tf_merkle_from_rows_dev(const uint64_t* d_rows, size_t row_len, size_t n_rows, uint64_t* d_nodes, size_t batch, void* stream) try {
    return merkle_from_rows_dev(d_rows, row_len, n_rows, d_nodes, batch, stream);
} TF_ABI_CATCH

int tf_coset_interpolate_bfe(const uint64_t* v, size_t n, uint64_t off, uint64_t* out, size_t batch) try {
    TRY(check_len(n));
    if (n == 0 || batch == 0) return TF_OK;
    if (!v || !out) return TF_ERR_NULL_POINTER;
    if (off == 0) return TF_ERR_INVERSE_OF_ZERO;
    return host_roundtrip(v, n * batch, nullptr, 0, out, n * batch,
                          [&](u64* a, u64*, u64* o, hipStream_t s) { return coset_interp_dev(a, n, off, o, batch, 1, s); });
} TF_ABI_CATCH
int tf_coset_interpolate_xfe(const uint64_t* v, size_t n, uint64_t off, uint64_t* out, size_t batch) try {
    TRY(check_len(n));
    if (n == 0 || batch == 0) return TF_OK;
    if (!v || !out) return TF_ERR_NULL_POINTER;
    if (off == 0) return TF_ERR_INVERSE_OF_ZERO;
    return host_roundtrip(v, 3 * n * batch, nullptr, 0, out, 3 * n * batch,
                          [&](u64* a, u64*, u64* o, hipStream_t s) { return coset_interp_dev(a, n, off, o, batch, 3, s); });
} TF_ABI_CATCH
int tf_poly_mul_bfe(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t batch) try {
    if (batch == 0 || na == 0 || nb == 0) return TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, na * batch, b, nb * batch, out, (na + nb - 1) * batch,
                          [&](u64* x, u64* y, u64* o, hipStream_t s) { return poly_mul_dev(x, na, y, nb, o, batch, 1, s); });
} TF_ABI_CATCH
int tf_poly_mul_xfe(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t batch) try {
    if (batch == 0 || na == 0 || nb == 0) return TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, 3 * na * batch, b, 3 * nb * batch, out, 3 * (na + nb - 1) * batch,
                          [&](u64* x, u64* y, u64* o, hipStream_t s) { return poly_mul_dev(x, na, y, nb, o, batch, 3, s); });
} TF_ABI_CATCH
int tf_poly_square_bfe(const uint64_t* a, size_t na, uint64_t* out, size_t batch) try {
    if (batch == 0 || na == 0) return TF_OK;
    if (!a || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, na * batch, nullptr, 0, out, (2 * na - 1) * batch,
                          [&](u64* x, u64*, u64* o, hipStream_t s) { return poly_square_dev(x, na, o, batch, 1, s); });
} TF_ABI_CATCH
int tf_poly_square_xfe(const uint64_t* a, size_t na, uint64_t* out, size_t batch) try {
    if (batch == 0 || na == 0) return TF_OK;
    if (!a || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, 3 * na * batch, nullptr, 0, out, 3 * (2 * na - 1) * batch,
                          [&](u64* x, u64*, u64* o, hipStream_t s) { return poly_square_dev(x, na, o, batch, 3, s); });
} TF_ABI_CATCH
int tf_poly_batch_evaluate_bfe(const uint64_t* c, size_t nc, const uint64_t* pts, size_t np, uint64_t* out) try {
    if (np == 0) return TF_OK;
    if (!pts || !out || (nc && !c)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(c, nc, pts, np, out, np,
                          [&](u64* dc, u64* dp, u64* o, hipStream_t s) { return batch_evaluate_dev(dc, nc, nc, 1, dp, np, o, 1, s); });
} TF_ABI_CATCH
int tf_poly_batch_evaluate_xfe(const uint64_t* c, size_t nc, const uint64_t* pts, size_t np, uint64_t* out) try {
    if (np == 0) return TF_OK;
    if (!pts || !out || (nc && !c)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(c, 3 * nc, pts, 3 * np, out, 3 * np,
                          [&](u64* dc, u64* dp, u64* o, hipStream_t s) { return batch_evaluate_dev(dc, nc, 3 * nc, 1, dp, np, o, 3, s); });
} TF_ABI_CATCH
int tf_poly_zerofier_bfe_dev(const uint64_t* r, size_t n, uint64_t* out, void* stream) try { return zerofier_dev(r, n, out, 1, stream); } TF_ABI_CATCH
int tf_poly_zerofier_xfe_dev(const uint64_t* r, size_t n, uint64_t* out, void* stream) try { return zerofier_dev(r, n, out, 3, stream); } TF_ABI_CATCH
static int zerofier_host(const uint64_t* r, size_t n, uint64_t* out, int L) {
    if (!out || (n && !r)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(r, n * L, nullptr, 0, out, (n + 1) * L, [&](u64* dr, u64*, u64* o, hipStream_t s) { return zerofier_dev(dr, n, o, L, s); });
}
int tf_poly_zerofier_bfe(const uint64_t* r, size_t n, uint64_t* out) try { return zerofier_host(r, n, out, 1); } TF_ABI_CATCH
int tf_poly_zerofier_xfe(const uint64_t* r, size_t n, uint64_t* out) try { return zerofier_host(r, n, out, 3); } TF_ABI_CATCH
int tf_poly_interpolate_bfe_dev(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, void* stream) try {
    return interpolate_dev(d, v, n, rows, out, 1, stream);
} TF_ABI_CATCH
int tf_poly_interpolate_xfe_dev(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, void* stream) try {
    return interpolate_dev(d, v, n, rows, out, 3, stream);
} TF_ABI_CATCH
static int interpolate_host(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, int L) {
    if (n == 0) return TF_ERR_EMPTY_DOMAIN;
    if (rows == 0) return TF_OK;
    if (!d || !v || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(d, n * L, v, rows * n * L, out, rows * n * L,
                          [&](u64* dd, u64* dv, u64* o, hipStream_t s) { return interpolate_dev(dd, dv, n, rows, o, L, s); });
}
int tf_poly_interpolate_bfe(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out) try { return interpolate_host(d, v, n, rows, out, 1); } TF_ABI_CATCH
int tf_poly_interpolate_xfe(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out) try { return interpolate_host(d, v, n, rows, out, 3); } TF_ABI_CATCH
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
int tf_zerofier_tree_new_bfe(const uint64_t* domain, size_t n, tf_zerofier_tree** tree) try { return tree_new_any(domain, n, 1, false, nullptr, tree); } TF_ABI_CATCH
int tf_zerofier_tree_new_xfe(const uint64_t* domain, size_t n, tf_zerofier_tree** tree) try { return tree_new_any(domain, n, 3, false, nullptr, tree); } TF_ABI_CATCH
int tf_zerofier_tree_new_bfe_dev(const uint64_t* d_domain, size_t n, void* stream, tf_zerofier_tree** tree) try {
    return tree_new_any(d_domain, n, 1, true, stream, tree);
} TF_ABI_CATCH
int tf_zerofier_tree_new_xfe_dev(const uint64_t* d_domain, size_t n, void* stream, tf_zerofier_tree** tree) try {
    return tree_new_any(d_domain, n, 3, true, stream, tree);
} TF_ABI_CATCH
void tf_zerofier_tree_free(tf_zerofier_tree* tree) { tree_handle_free(tree_of(tree)); }
size_t tf_zerofier_tree_num_points(const tf_zerofier_tree* tree) { return tree ? tree_handle_num_points(tree_of(tree)) : 0; }
int tf_zerofier_tree_width(const tf_zerofier_tree* tree) { return tree ? tree_handle_width(tree_of(tree)) : 0; }
int tf_zerofier_tree_zerofier_dev(const tf_zerofier_tree* tree, uint64_t* d_out, void* stream) try { return tree_handle_zerofier(tree_of(tree), d_out, stream); } TF_ABI_CATCH
int tf_zerofier_tree_batch_evaluate_dev(const tf_zerofier_tree* tree, const uint64_t* d_coeffs, size_t n_coeffs, size_t batch, uint64_t* d_out,
                                        void* stream) try {
    return tree_handle_batch_evaluate(tree_of(tree), d_coeffs, n_coeffs, batch, d_out, stream);
} TF_ABI_CATCH
int tf_zerofier_tree_interpolate_dev(tf_zerofier_tree* tree, const uint64_t* d_values, size_t rows, uint64_t* d_out, void* stream) try {
    return tree_handle_interpolate(tree_of(tree), d_values, rows, d_out, stream);
} TF_ABI_CATCH
// ---- enqueue-and-return variants (tf_hip.h): panic cases go to *d_status on the device, nothing synchronises
int tf_poly_interpolate_bfe_dev_async(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, void* stream, int* d_status) try {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return interpolate_dev(d, v, n, rows, out, 1, stream, d_status);
} TF_ABI_CATCH
int tf_poly_interpolate_xfe_dev_async(const uint64_t* d, const uint64_t* v, size_t n, size_t rows, uint64_t* out, void* stream, int* d_status) try {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return interpolate_dev(d, v, n, rows, out, 3, stream, d_status);
} TF_ABI_CATCH
int tf_poly_clean_divide_bfe_dev_async(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, void* stream, int* d_status) try {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return clean_divide_dev(a, na, b, nb, out, stream, 1, d_status);
} TF_ABI_CATCH
int tf_poly_clean_divide_many_bfe_dev_async(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out, void* stream,
                                            int* d_status) try {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return clean_divide_dev(a, na, b, nb, out, stream, batch, d_status);
} TF_ABI_CATCH
int tf_zerofier_tree_new_bfe_dev_async(const uint64_t* d_domain, size_t n, void* stream, tf_zerofier_tree** tree) try {
    return tree_new_async(d_domain, n, 1, stream, tree);
} TF_ABI_CATCH
int tf_zerofier_tree_new_xfe_dev_async(const uint64_t* d_domain, size_t n, void* stream, tf_zerofier_tree** tree) try {
    return tree_new_async(d_domain, n, 3, stream, tree);
} TF_ABI_CATCH
int tf_zerofier_tree_interpolate_dev_async(tf_zerofier_tree* tree, const uint64_t* d_values, size_t rows, uint64_t* d_out, void* stream, int* d_status) try {
    if (!d_status) return TF_ERR_NULL_POINTER;
    return tree_handle_interpolate(tree_of(tree), d_values, rows, d_out, stream, d_status);
} TF_ABI_CATCH
int tf_zerofier_tree_zerofier(const tf_zerofier_tree* tree, uint64_t* out) try {
    if (!tree || !out) return TF_ERR_NULL_POINTER;
    TreeHandle* H = tree_of(tree);
    const size_t hn = tree_handle_num_points(H), hl = (size_t)tree_handle_width(H);
    return host_roundtrip(nullptr, 0, nullptr, 0, out, (hn + 1) * hl, [&](u64*, u64*, u64* o, hipStream_t s) { return tree_handle_zerofier(H, o, s); });
} TF_ABI_CATCH
int tf_zerofier_tree_batch_evaluate(const tf_zerofier_tree* tree, const uint64_t* coeffs, size_t n_coeffs, size_t batch, uint64_t* out) try {
    if (!tree) return TF_ERR_NULL_POINTER;
    TreeHandle* H = tree_of(tree);
    const size_t hn = tree_handle_num_points(H), hl = (size_t)tree_handle_width(H);
    if (hn == 0 || batch == 0) return TF_OK;
    if (!out || (n_coeffs && !coeffs)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(coeffs, batch * n_coeffs * hl, nullptr, 0, out, batch * hn * hl,
                          [&](u64* dc, u64*, u64* o, hipStream_t s) { return tree_handle_batch_evaluate(H, dc, n_coeffs, batch, o, s); });
} TF_ABI_CATCH
int tf_zerofier_tree_interpolate(tf_zerofier_tree* tree, const uint64_t* values, size_t rows, uint64_t* out) try {
    if (!tree) return TF_ERR_NULL_POINTER;
    TreeHandle* H = tree_of(tree);
    const size_t hn = tree_handle_num_points(H), hl = (size_t)tree_handle_width(H);
    if (hn == 0) return TF_ERR_EMPTY_DOMAIN;
    if (rows == 0) return TF_OK;
    if (!values || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(values, rows * hn * hl, nullptr, 0, out, rows * hn * hl,
                          [&](u64* dv, u64*, u64* o, hipStream_t s) { return tree_handle_interpolate(H, dv, rows, o, s); });
} TF_ABI_CATCH
int tf_coset_eval_xfe_xoffset_dev(const uint64_t* c, size_t nc, const uint64_t offset[3], uint64_t* out, size_t order, size_t batch, void* stream) try {
    return coset_eval_xoffset_dev(c, nc, offset, out, order, batch, stream);
} TF_ABI_CATCH
int tf_coset_interpolate_xfe_xoffset_dev(const uint64_t* v, size_t n, const uint64_t offset[3], uint64_t* out, size_t batch, void* stream) try {
    return coset_interp_xoffset_dev(v, n, offset, out, batch, stream);
} TF_ABI_CATCH
int tf_coset_eval_xfe_xoffset(const uint64_t* c, size_t nc, const uint64_t offset[3], uint64_t* out, size_t order, size_t batch) try {
    if (nc > order) return TF_ERR_ORDER_NOT_ABOVE_DEGREE;
    int rc = check_len(order);
    if (rc) return rc;
    if (order == 0 || batch == 0) return TF_OK;
    if (!out || !offset || (nc && !c)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(c, 3 * nc * batch, nullptr, 0, out, 3 * order * batch,
                          [&](u64* dc, u64*, u64* o, hipStream_t s) { return coset_eval_xoffset_dev(dc, nc, offset, o, order, batch, s); });
} TF_ABI_CATCH
int tf_coset_interpolate_xfe_xoffset(const uint64_t* v, size_t n, const uint64_t offset[3], uint64_t* out, size_t batch) try {
    int rc = check_len(n);
    if (rc) return rc;
    if (n == 0 || batch == 0) return TF_OK;
    if (!v || !out || !offset) return TF_ERR_NULL_POINTER;
    return host_roundtrip(v, 3 * n * batch, nullptr, 0, out, 3 * n * batch,
                          [&](u64* dv, u64*, u64* o, hipStream_t s) { return coset_interp_xoffset_dev(dv, n, offset, o, batch, s); });
} TF_ABI_CATCH
int tf_barycentric_evaluate_bfe_dev(const uint64_t* cw, size_t n, size_t batch, const uint64_t x[3], uint64_t* out, void* stream) try {
    return barycentric_dev(cw, n, batch, 1, x, out, stream);
} TF_ABI_CATCH
int tf_barycentric_evaluate_xfe_dev(const uint64_t* cw, size_t n, size_t batch, const uint64_t x[3], uint64_t* out, void* stream) try {
    return barycentric_dev(cw, n, batch, 3, x, out, stream);
} TF_ABI_CATCH
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
int tf_barycentric_evaluate_bfe(const uint64_t* cw, size_t n, size_t batch, const uint64_t x[3], uint64_t* out) try { return barycentric_host(cw, n, batch, 1, x, out); } TF_ABI_CATCH
int tf_barycentric_evaluate_xfe(const uint64_t* cw, size_t n, size_t batch, const uint64_t x[3], uint64_t* out) try { return barycentric_host(cw, n, batch, 3, x, out); } TF_ABI_CATCH
// Polynomial<BFieldElement>::evaluate::<XFieldElement, XFieldElement> (polynomial.rs:309-320) for `batch` polynomials at n_points
// extension-field points: out[(b * n_points + i) * 3] = f_b(points[i]).  Horner (the shape of the use: every column polynomial
// of a table at a few out-of-domain points).
int tf_poly_evaluate_bfe_at_xfe_dev(const uint64_t* c, size_t nc, size_t batch, const uint64_t* pts, size_t np, uint64_t* out, void* stream) try {
    if (np == 0 || batch == 0) return TF_OK;
    if (!pts || !out || (nc && !c)) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    return batch_evaluate_horner(c, nc, nc, batch, pts, np, out, 3, stream, 1);
} TF_ABI_CATCH
int tf_poly_evaluate_bfe_at_xfe(const uint64_t* c, size_t nc, size_t batch, const uint64_t* pts, size_t np, uint64_t* out) try {
    if (np == 0 || batch == 0) return TF_OK;
    if (!pts || !out || (nc && !c)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(c, batch * nc, pts, 3 * np, out, 3 * batch * np,
                          [&](u64* dc, u64* dp, u64* o, hipStream_t s) { return batch_evaluate_horner(dc, nc, nc, batch, dp, np, o, 3, s, 1); });
} TF_ABI_CATCH
int tf_poly_clean_divide_bfe_dev(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, void* stream) try {
    return clean_divide_dev(a, na, b, nb, out, stream);
} TF_ABI_CATCH
int tf_poly_mul_shared_bfe_dev(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out, void* stream) try {
    return poly_mul_shared_dev(a, na, batch, b, nb, out, 1, stream);
} TF_ABI_CATCH
int tf_poly_mul_shared_xfe_dev(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out, void* stream) try {
    return poly_mul_shared_dev(a, na, batch, b, nb, out, 3, stream);
} TF_ABI_CATCH
int tf_poly_clean_divide_many_bfe_dev(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out, void* stream) try {
    return clean_divide_dev(a, na, b, nb, out, stream, batch);
} TF_ABI_CATCH
int tf_poly_clean_divide_many_bfe(const uint64_t* a, size_t na, size_t batch, const uint64_t* b, size_t nb, uint64_t* out) try {
    if (nb == 0) return TF_ERR_DIVISION_BY_ZERO;
    if (na < nb) return na ? TF_ERR_DIVISION_NOT_CLEAN : TF_OK;
    if (batch == 0) return TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, batch * na, b, nb, out, batch * (na - nb + 1),
                          [&](u64* da, u64* db, u64* o, hipStream_t s) { return clean_divide_dev(da, na, db, nb, o, s, batch); });
} TF_ABI_CATCH
int tf_poly_clean_divide_bfe(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out) try {
    if (nb == 0) return TF_ERR_DIVISION_BY_ZERO;
    if (na < nb) return na ? TF_ERR_DIVISION_NOT_CLEAN : TF_OK;
    if (!a || !b || !out) return TF_ERR_NULL_POINTER;
    return host_roundtrip(a, na, b, nb, out, na - nb + 1, [&](u64* da, u64* db, u64* o, hipStream_t s) { return clean_divide_dev(da, na, db, nb, o, s); });
} TF_ABI_CATCH
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
int tf_coset_extrapolate_bfe(uint64_t offset, const uint64_t* cw, size_t n, size_t batch, const uint64_t* pts, size_t np, uint64_t* out) try {
    return coset_extrapolate_host(offset, cw, n, batch, pts, np, out, 1);
} TF_ABI_CATCH
int tf_coset_extrapolate_xfe(uint64_t offset, const uint64_t* cw, size_t n, size_t batch, const uint64_t* pts, size_t np, uint64_t* out) try {
    return coset_extrapolate_host(offset, cw, n, batch, pts, np, out, 3);
} TF_ABI_CATCH
int tf_tip5_hash_table_rows(const uint64_t* table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t* digests,
                            size_t batch) try {
    if (width != 1 && width != 3) return TF_ERR_NULL_POINTER;
    if (n_rows == 0 || batch == 0) return TF_OK;
    if (!digests || (n_cols && !table)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(table, batch * n_cols * col_stride, nullptr, 0, digests, batch * n_rows * 5,
                          [&](u64* dt, u64*, u64* o, hipStream_t s) { return hash_table_rows_dev(dt, n_rows, n_cols, width, col_stride, o, batch, s); });
} TF_ABI_CATCH
int tf_merkle_from_columns(const uint64_t* table, size_t n_rows, size_t n_cols, int width, size_t col_stride, uint64_t* nodes_out,
                           size_t batch) try {
    if (width != 1 && width != 3) return TF_ERR_NULL_POINTER;
    int rc = check_leaves(n_rows);
    if (rc) return rc;
    if (batch == 0) return TF_OK;
    if (!nodes_out || (n_cols && !table)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(table, batch * n_cols * col_stride, nullptr, 0, nodes_out, batch * n_rows * 10,
                          [&](u64* dt, u64*, u64* o, hipStream_t s) { return merkle_from_columns_dev(dt, n_rows, n_cols, width, col_stride, o, batch, s); });
} TF_ABI_CATCH
int tf_merkle_from_rows(const uint64_t* rows, size_t row_len, size_t n_rows, uint64_t* nodes_out, size_t batch) try {
    TRY(check_leaves(n_rows));
    if (batch == 0) return TF_OK;
    if (!nodes_out || (row_len && !rows)) return TF_ERR_NULL_POINTER;
    return host_roundtrip(rows, n_rows * row_len * batch, nullptr, 0, nodes_out, n_rows * batch * 10,
                          [&](u64* r, u64*, u64* o, hipStream_t s) { return merkle_from_rows_dev(r, row_len, n_rows, o, batch, s); });
} TF_ABI_CATCH

int tf_merkle_auth_structure_indices(size_t num_leafs, const uint64_t* leaf_indices, size_t k, uint64_t* out_indices,
                                     size_t capacity, size_t* out_count) try {
    if ((k && !leaf_indices) || !out_count) return TF_ERR_NULL_POINTER;
    std::vector<unsigned long long> idx;
    TRY(auth_structure_indices(num_leafs, leaf_indices, k, &idx));
    *out_count = idx.size();
    if (!out_indices || capacity == 0) return TF_OK;  // sizing call: only the count
    if (capacity < idx.size()) return TF_ERR_BUFFER_TOO_SMALL;  // nothing is written; *out_count says what is needed
    for (size_t i = 0; i < idx.size(); ++i) out_indices[i] = idx[i];
    return TF_OK;
} TF_ABI_CATCH

int tf_merkle_authentication_structure_dev(const uint64_t* d_nodes, size_t num_leafs, const uint64_t* leaf_indices, size_t k,
                                           uint64_t* out_digests, size_t capacity_digests, size_t* out_count, void* stream) try {
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
    TRY(gather_digests_dev(d_nodes, reinterpret_cast<const unsigned long long*>(didx.p), idx.size(), dout.p, s));
    TRY(d2h(out_digests, dout.p, idx.size() * 5, s));
    return sync(s);
} TF_ABI_CATCH

}  // extern "C"


