// tf_internal.h -- what the translation units of libtf_hip.so share on the host side (nothing here is part of the ABI: the
// library is built with -fvisibility=hidden and only the tf_* functions of include/tf_hip.h are exported).
//   tf_ntt.hip    device context, table caches, the pass planner and every launcher of ntt_kernels.h
//   tf_lat.hip    the latency-shaped transforms and the one-launch-per-level kernels of the tree walks (lat_kernels.h)
//   tf_tip5.hip   Tip5 / Merkle launchers (tip5_kernels.h), authentication structures
//   tf_poly.hip   the callers on either side of the path, SURVEY 8(f) (poly_kernels.h)
//   tf_abi.hip    host-pointer wrappers and the extern "C" entry points
//   tf_multi.hip  one host-resident batch over several GPUs (tf_*_multi), device selection
#pragma once
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

#pragma GCC visibility push(default)
#include "../../include/tf_hip.h"
#pragma GCC visibility pop
#include "gl64.h"
#include "ntt_args.h"
#include "tf_guard.h"

namespace tfi {

using gl::u32;
using gl::u64;

// ------------------------------------------------------------------------------------ errors
extern thread_local std::string t_last_error;
int hip_fail(hipError_t e, const char* what, const char* file, int line);
#define HIPCHK(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

#define TRY(x)            \
    do {                  \
        int rc_ = (x);    \
        if (rc_) return rc_; \
    } while (0)

// The product library has ONE plan: every A/B switch and diagnostic knob of the laboratory (the TF_* environment variables of
// DESIGN_HISTORY.md, the measured-loser kernels, the ablation modes) exists only in the TF_AB_BUILD library (make ab ->
// libtf_hip_ab.so).  In the product build ab_env() is a constant null pointer, so every switch folds away at compile time; the
// product reads exactly two environment variables, both deployment settings: TF_NTT_TILE_BYTES (scratch budget between the passes)
// and TF_NTT_PIPE (side streams for pipelined batch tiles).
#ifdef TF_AB_BUILD
inline const char* ab_env(const char* name) { return getenv(name); }
#else
constexpr const char* ab_env(const char*) { return nullptr; }
#endif

u64 root_of_unity_mont(int log_n);
int ilog2(size_t v);

// ------------------------------------------------------------------------------------ per-device context
struct DeviceCtx {
    std::mutex mu;
    std::map<u64, u64*> tables;           // twiddle tables, never freed while the process lives
    std::map<std::pair<u64, u64>, u64*> pow_tables;  // (offset_raw, n) -> offset^j table
    std::map<std::pair<u64, u64>, u64*> scaled_post;  // (offset_raw, log_m << 8 | a) -> T[k B + b] * offset^b (get_scaled_post_table)
    size_t cached_scaled_post_bytes = 0;
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
    std::map<const u64*, ScratchBlock> temp_pow;  // temporary power tables in flight: scratch blocks, given back by release_pow_table (guarded by mu)
};

constexpr int kMaxDevices = 64;
extern DeviceCtx g_ctx[kMaxDevices];

// ------------------------------------------------------------------------------------ tf_ntt.hip
int current_ctx(DeviceCtx** out);
int device_cus();  // compute units of the calling thread's CURRENT device (cached per device; 256 if the runtime will not say)
hipError_t pool_malloc_async(void** p, size_t bytes, hipStream_t stream);  // every stream-ordered temporary of the library
int scratch_acquire(DeviceCtx* ctx, size_t bytes, hipStream_t stream, DeviceCtx::ScratchBlock* out);
void scratch_release(DeviceCtx* ctx, DeviceCtx::ScratchBlock blk, hipStream_t stream);
int release_caches(DeviceCtx* ctx);
void read_env();
int ensure_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done_mask);
int get_pow_table(DeviceCtx* ctx, u64 offset_raw, size_t n, hipStream_t stream, const u64** out, bool* temp, size_t cosets = 1, int log_order = 0);
void release_pow_table(DeviceCtx* ctx, const u64* table, bool temp, hipStream_t stream);  // after the launches that read a table get_pow_table returned

// configuration (tf_set_* hooks of the ABI; TF_* environment variables read once by read_env)
constexpr int kMaxPipe = 4;
extern size_t g_tile_bytes;
extern std::atomic<int> g_pipe, g_nt, g_min_passes, g_lat_mode, g_pre2_mode, g_small_launch_mode;
#ifdef TF_AB_BUILD
extern std::atomic<int> g_chain;
extern unsigned long long* g_dbg_buf;
#endif

// keys of DeviceCtx::tables
enum : u64 { TAG_INNER = 1, TAG_POST = 2, TAG_TINY = 3, TAG_BLOCK1 = 4, TAG_BLOCK2 = 5, TAG_LAT = 6 };
inline u64 make_key(u64 tag, u64 a, u64 b, u64 c, u64 d) { return (tag << 56) | (a << 40) | (b << 24) | (c << 8) | d; }
int upload_table(const std::vector<u64>& host, u64** dev);
int get_post_table(DeviceCtx* ctx, int log_m, int a, bool inverse, hipStream_t stream, const u64** out, bool* temp);

int check_len(size_t n);
int pass_count(int log_n);
void choose_split(int log_n, int P, int L, int (&a)[4]);
void pre2_split(int log_n, int (&a)[4]);
void pre2_split(int log_n, int (&a)[4], bool c8);
bool pre2_plan_ok(int log_n, int L, size_t n, size_t cosets, bool has_in2, long long n_out, bool inverse, bool load_work, bool store_scale);
bool can_truncate(size_t n, int L);
int run_ntt(DeviceCtx* ctx, const u64* in, u64* out, long long in_bs, long long out_bs, size_t n, size_t batch, int L, bool inverse,
            const u64* pre_scale, long long n_coeffs, hipStream_t stream, const u64* post_scale = nullptr, size_t cosets = 1,
            const u64* in2 = nullptr, long long n_out = -1, const u64* coset_offset = nullptr);
int ntt_dev(u64* d_x, size_t n, size_t batch, int L, int inverse, void* stream);
int coset_eval_dev(const u64* d_coeffs, size_t n_coeffs, u64 offset_raw, u64* d_out, size_t order, size_t batch, int L, void* stream);
// ------------------------------------------------------------------------------------ tf_lat.hip
// latency-shaped transforms and the one-launch-per-level kernels of the zerofier-tree walks
bool small_launch_for(size_t n, size_t cosets, size_t batch, int L, long long n_out);
bool lat_wanted(int log_n, size_t batch, int L);
bool lat2_wanted(int log_n, size_t batch, int L);
int launch_lat2(DeviceCtx* ctx, const u64* in, u64* out, long long in_bs, long long out_bs, int log_n, size_t batch, int L, bool inverse,
                long long n_coeffs, const u64* in2, hipStream_t stream, const u64* pre_scale = nullptr, const u64* post_scale = nullptr);
int launch_lat(DeviceCtx* ctx, const u64* in, u64* out, long long in_bs, long long out_bs, int log_n, size_t batch, int L, bool inverse,
               long long n_coeffs, const u64* in2, hipStream_t stream, const tfk::NttLatArgs* mods = nullptr, const u64* pre_scale = nullptr,
               const u64* post_scale = nullptr);
bool tree_level_wanted(long long order, long long lines, int L = 1, bool up = false);
template <bool UP>
int launch_tree_level(DeviceCtx* ctx, int log_n, tfk::TreeLevelArgs a, hipStream_t s, int L = 1);
bool tree_build_level_wanted(long long order, long long parents);
int launch_tree_build_level(DeviceCtx* ctx, int log_n2, tfk::TreeBuildArgs a, hipStream_t s);

// ------------------------------------------------------------------------------------ tf_tip5.hip
int check_leaves(size_t n);
int tip5_permute_dev(u64* d_states, size_t count, void* stream);
int tip5_trace_dev(u64* d_states, u64* d_trace, size_t count, void* stream);
int tip5_hash_pairs_dev(const u64* d_in, u64* d_out, size_t count, void* stream);
int tip5_hash_varlen_rows_dev(const u64* d_rows, size_t row_len, size_t n_rows, u64* d_out, void* stream);
int hash_table_rows_dev(const u64* d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, u64* d_digests, size_t batch, void* stream);
int merkle_build_dev(const u64* d_leaves, size_t n, u64* d_nodes, size_t batch, void* stream);
int merkle_subtree_host(const u64* leaves_sub, size_t m, u64* nodes_tree, size_t n_sub, size_t sub, u64* root_out);  // tf_abi.hip
int merkle_root_dev(const u64* d_leaves, size_t n, u64* d_root, size_t batch, void* stream);
int merkle_from_rows_dev(const u64* d_rows, size_t row_len, size_t n_rows, u64* d_nodes, size_t batch, void* stream);
int merkle_from_columns_dev(const u64* d_table, size_t n_rows, size_t n_cols, int width, size_t col_stride, u64* d_nodes, size_t batch, void* stream);
int gather_digests_dev(const u64* d_nodes, const unsigned long long* d_idx, size_t count, u64* d_out, hipStream_t s);

// ------------------------------------------------------------------------------------ tf_poly.hip
extern std::atomic<int> g_batch_eval_route;
int coset_interp_dev(const u64* d_values, size_t n, u64 offset_raw, u64* d_out, size_t batch, int L, void* stream);
int hadamard_dev(const u64* a, const u64* b, u64* out, size_t count, int L, void* stream);
int poly_mul_dev(const u64* a, size_t na, const u64* b, size_t nb, u64* out, size_t batch, int L, void* stream, long long a_bs = 0, long long b_bs = 0);
int poly_mul_shared_dev(const u64* a, size_t na, size_t batch, const u64* b, size_t nb, u64* out, int L, void* stream);
int poly_square_dev(const u64* a, size_t na, u64* out, size_t batch, int L, void* stream);
int lde_dev(const u64* values, size_t n, u64 offset_in, u64* out, size_t m, u64 offset_out, size_t batch, int L, void* stream);
bool tree_route(size_t n_coeffs, size_t n_points, size_t batch, int L);
int batch_evaluate_horner(const u64* coeffs, size_t n_coeffs, size_t poly_stride, size_t batch, const u64* points, size_t n_points, u64* out, int L,
                          void* stream, int CL);
int batch_evaluate_dev(const u64* coeffs, size_t n_coeffs, size_t poly_stride, size_t batch, const u64* points, size_t n_points, u64* out, int L,
                       void* stream);
int coset_extrapolate_dev(u64 offset_raw, const u64* codewords, size_t n, size_t batch, const u64* points, size_t n_points, u64* out, int L, void* stream);
int zerofier_dev(const u64* roots, size_t n_roots, u64* out, int L, void* stream);
int interpolate_dev(const u64* domain, const u64* values, size_t n, size_t rows, u64* out, int L, void* stream, int* d_status = nullptr);
int clean_divide_dev(const u64* a, size_t na, const u64* b, size_t nb, u64* out, void* stream, size_t batch = 1, int* d_status = nullptr);
int coset_eval_xoffset_dev(const u64* d_coeffs, size_t n_coeffs, const u64 offset[3], u64* d_out, size_t order, size_t batch, void* stream);
int coset_interp_xoffset_dev(const u64* d_values, size_t n, const u64 offset[3], u64* d_out, size_t batch, void* stream);
int barycentric_dev(const u64* codewords, size_t n, size_t batch, int cw_width, const u64 x[3], u64* out, void* stream);
int auth_structure_indices(size_t num_leafs, const uint64_t* leaf_indices, size_t k, std::vector<unsigned long long>* out);
struct TreeHandle;  // a zerofier tree that outlives the call (math/zerofier_tree.rs), opaque outside tf_poly.hip
int tree_handle_new(const u64* d_domain, size_t n, int L, void* stream, TreeHandle** out, bool async = false);
void tree_handle_free(TreeHandle* H);
size_t tree_handle_num_points(const TreeHandle* H);
int tree_handle_width(const TreeHandle* H);
int tree_handle_zerofier(const TreeHandle* H, u64* d_out, void* stream);
int tree_handle_batch_evaluate(const TreeHandle* H, const u64* d_coeffs, size_t n_coeffs, size_t batch, u64* d_out, void* stream);
int tree_handle_interpolate(TreeHandle* H, const u64* d_values, size_t rows, u64* d_out, void* stream, int* d_status = nullptr);


}  // namespace tfi
