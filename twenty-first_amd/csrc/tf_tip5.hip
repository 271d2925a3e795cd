// tf_tip5.hip -- Tip5 / Merkle launchers of libtf_hip.so (tip5_kernels.h) and the authentication structures.
#include "tf_internal.h"
#include "tip5_kernels.h"

namespace tfi {

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


int ensure_tip5(DeviceCtx* ctx) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->tip5_ready) return TF_OK;
    tfk::Tip5Consts c;
    for (int i = 0; i < 80; ++i) c.rc[i] = gl::to_mont(kRoundConstants[i]);
    // (the lookup table, L(x) = ((x+1)^3 mod 257) - 1, tip5/mod.rs:1022-1026, is computed by the kernels themselves: stage_lut)
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(tfk::g_tip5), &c, sizeof(c)));
    tfk::Tip5MxConsts mx;
    tfk::fill_tip5_mx(mx, c.rc);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(tfk::g_tip5_mx), &mx, sizeof(mx)));
    HIPCHK(hipDeviceSynchronize());
    ctx->tip5_ready = true;
    return TF_OK;
}


// ------------------------------------------------------------------------------------ Tip5 / Merkle
// Two kernel families (tip5_kernels.h).  Launches of at most kCoopMaxCount permutation chains are latency-bound (fewer chains
// than the chip has SIMD slots): 16 lanes per permutation, ~2.5 us per permutation of a chain.  Everything larger runs in the
// matrix-pipe form (4 lanes per permutation, MDS on v_mfma_i32_16x16x64_i8 since round 6): crossover measured at 2^13 chains with the f64 form
// (profiles/r05_tip5_small_times.txt: hash_varlen of 33 words, 2^13 rows 24.4 vs 25.6 us, 2^14 rows 34.6 vs 26.3 us).
constexpr long long kCoopMaxCount = 1ll << 13;
static_assert(kCoopMaxCount >= 64 && (kCoopMaxCount & (kCoopMaxCount - 1)) == 0, "a power of two: the level at which a tree narrows is found by halving");

// Round 6: a launch of at most 8 permutation chains per compute unit leaves half the chip's 16-lane rows idle; it runs every chain on a row
// PAIR instead (tip5_permutation_coop2: the circulant's sixteen rotation terms split over the two rows; 2.01 -> 1.68 us per permutation,
// profiles/r06_microbench_coop2.txt).  A workgroup then holds 8 chains, so up to this count every workgroup still has a CU of its own.
inline bool coop_two_rows(long long chains) {
    static const bool off = ab_env("TF_TIP5_NO_COOP2") != nullptr;  // A/B switch
    return !off && chains <= 8ll * device_cus();
}
// ... and the subtree launches use a row pair per hash_pair at the levels that have the rows to spare, when every workgroup has a CU of its own
inline int subtree_two_rows(long long workgroups) {
    static const bool off = ab_env("TF_TIP5_NO_COOP2") != nullptr;  // A/B switch
    return (!off && workgroups <= (long long)device_cus()) ? 1 : 0;
}

// per_tree = 2^shift, or -1
inline int shift_of(long long per_tree) { return (per_tree > 0 && !(per_tree & (per_tree - 1))) ? __builtin_ctzll((unsigned long long)per_tree) : -1; }

// grid of a matrix-pipe launch: one workgroup (4 waves x 16 permutations) per 64 items, capped at kMxBlocksPerCu per CU -- beyond
// that the waves walk the items with a grid stride, so the tables are staged once per wave and not once per 16 items (8 workgroups
// are resident per CU at the kernels' VGPR count; 56 keeps the hardware's dynamic balancing: measured 7 / 14 / 28 / 56 / no cap on
// the 2^24-leaf tree: 4.87 / 5.00 / 5.06 / 5.07 / 5.06 G leaves/s, profiles/r05_tip5_grid_cap.txt)
constexpr long long kMxBlocksPerCu = 56;
inline unsigned mx_blocks(long long count) {
    const long long cap = (long long)device_cus() * kMxBlocksPerCu;
    const long long want = (count + 63) / 64;
    return (unsigned)(want < cap ? want : cap);
}

void launch_permute_mx(u64* d_states, u64* d_trace, long long count, hipStream_t s) {
    hipLaunchKernelGGL(tfk::tip5_permute_mx_kernel<1>, dim3(mx_blocks(count)), dim3(256), 0, s, d_states, d_trace, count);
}

int tip5_permute_dev(u64* d_states, size_t count, void* stream) {
    if (count == 0) return TF_OK;
    if (!d_states) return TF_ERR_NULL_POINTER;
    DeviceCtx* ctx = nullptr;
    int rc = current_ctx(&ctx);
    if (rc) return rc;
    rc = ensure_tip5(ctx);
    if (rc) return rc;
    if ((long long)count <= kCoopMaxCount) {
        if (coop_two_rows((long long)count))
            hipLaunchKernelGGL(tfk::tip5_permute_coop_kernel<2>, dim3((unsigned)((count + 7) / 8)), dim3(256), 0, static_cast<hipStream_t>(stream), d_states,
                               (long long)count);
        else
            hipLaunchKernelGGL(tfk::tip5_permute_coop_kernel<1>, dim3((unsigned)((count + 15) / 16)), dim3(256), 0, static_cast<hipStream_t>(stream), d_states,
                               (long long)count);
    } else {
        launch_permute_mx(d_states, nullptr, (long long)count, static_cast<hipStream_t>(stream));
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
    launch_permute_mx(d_states, d_trace, (long long)count, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return TF_OK;
}

int launch_hash_pairs(const u64* in, u64* out, u64* leaf_copy, long long count, long long per_tree, long long in_ts,
                      long long out_ts, long long copy_ts, hipStream_t s) {
    if (count == 0) return TF_OK;
    if (count <= kCoopMaxCount && !leaf_copy) {
        // fewer permutations than the GPU has lanes: latency, not throughput, is what this launch costs -> 16 lanes each
        if (coop_two_rows(count))
            hipLaunchKernelGGL(tfk::tip5_hash_pairs_coop_kernel<2>, dim3((unsigned)((count + 7) / 8)), dim3(256), 0, s, in, out, count, per_tree, shift_of(per_tree),
                               in_ts, out_ts);
        else
            hipLaunchKernelGGL(tfk::tip5_hash_pairs_coop_kernel<1>, dim3((unsigned)((count + 15) / 16)), dim3(256), 0, s, in, out, count, per_tree, shift_of(per_tree),
                               in_ts, out_ts);
        HIPCHK(hipGetLastError());
        return TF_OK;
    }
    hipLaunchKernelGGL(tfk::tip5_hash_pairs_mx_kernel<1>, dim3(mx_blocks(count)), dim3(256), 0, s, in, out, leaf_copy, count, per_tree,
                       shift_of(per_tree), in_ts, out_ts, copy_ts);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

// hash_varlen of n_rows rows: few rows (or one long input) are latency-bound -> 16 lanes per row
int launch_hash_varlen_rows(const u64* rows, long long row_len, long long n_rows, u64* out, long long per_tree, long long out_ts,
                            hipStream_t s) {
    if (n_rows == 0) return TF_OK;
    if (n_rows <= kCoopMaxCount) {
        if (coop_two_rows(n_rows))
            hipLaunchKernelGGL(tfk::tip5_hash_varlen_rows_coop_kernel<2>, dim3((unsigned)((n_rows + 7) / 8)), dim3(256), 0, s, rows, row_len, n_rows, out, per_tree,
                               shift_of(per_tree), out_ts);
        else
            hipLaunchKernelGGL(tfk::tip5_hash_varlen_rows_coop_kernel<1>, dim3((unsigned)((n_rows + 15) / 16)), dim3(256), 0, s, rows, row_len, n_rows, out, per_tree,
                               shift_of(per_tree), out_ts);
    } else {
        hipLaunchKernelGGL(tfk::tip5_hash_varlen_rows_mx_kernel<1>, dim3(mx_blocks(n_rows)), dim3(256), 0, s, rows, row_len, n_rows, out,
                           per_tree, shift_of(per_tree), out_ts);
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

// Near the root a level is one permutation latency whatever its width, and what it costs is the launch around it.  Once a level has
// at most kCoopMaxCount pairs (all trees of the call together) the remaining levels run as SUBTREES: a workgroup takes 2^k consecutive
// nodes and computes the k levels above them through LDS (merkle_subtree_kernel; the last launch is merkle_top_kernel, one workgroup
// per tree), k <= kSubtreeMaxLog and the levels split evenly over the launches.  Measured (profiles/r05_top_width_ab.txt,
// r05_subtree_ab.txt): a workgroup is slow at 64+ concurrent 16-lane permutations (so 256-node tops lost 8-10 us to 64-node ones), and
// a launch per level costs ~2 us per level more than a barrier per level.
#ifndef TF_SUBTREE_MAX_LOG
#define TF_SUBTREE_MAX_LOG 6
#endif
constexpr int kSubtreeMaxLog = TF_SUBTREE_MAX_LOG;
// What the subtree launches assume about the two constants: merkle_subtree_kernel keeps two levels of at most 2^8 digests in
// LDS (buf[2][256 * 5]); merkle_narrow_levels lays its two scratch halves out for launches that shrink a level by >= 16 x -- with
// several launches (>= kSubtreeMaxLog + 1 levels) every launch but the last takes ceil(levels / launches) >= 4 levels only if
// kSubtreeMaxLog >= 4 --, and it is entered with at most 2 kCoopMaxCount digests per call above kTopWidth.
static_assert(kSubtreeMaxLog >= 4 && kSubtreeMaxLog <= 8, "merkle_subtree_kernel's LDS buffer and merkle_narrow_levels' scratch layout");
constexpr long long kTopWidth = 1ll << kSubtreeMaxLog;  // a tree of at most this many leaves is one merkle_top_kernel launch

inline int ilog2ll(long long v) { return 63 - __builtin_clzll((unsigned long long)v); }
inline unsigned subtree_threads(int chunk_log) { return (unsigned)std::min(1024, std::max(256, 8 << chunk_log)); }  // 16 lanes per pair

// Every level above the level of w nodes per tree (`level`: its digests, in_ts words from tree to tree), w a power of two with
// w <= kTopWidth or (w / 2) * batch <= kCoopMaxCount.  With d_nodes: in place in the heap-ordered node arrays (copy_input: `level` is
// the leaf level, to be copied to nodes[w .. 2 w) on the way); without: root only, subtree roots ping-pong through scratch (>= w / 2
// digests per tree when more than one launch is needed) and the roots land in d_root.
int merkle_narrow_levels(const u64* level, long long in_ts, long long w, u64* d_nodes, long long nodes_ts, u64* d_root, u64* scratch,
                         size_t batch, bool copy_input, hipStream_t s) {
    int remaining = ilog2ll(w);
    int launches = std::max(1, (remaining + kSubtreeMaxLog - 1) / kSubtreeMaxLog);
    u64* sa = scratch;
    // more than one launch means >= 7 levels, so every launch but the last takes >= 4: the first leaves <= w / 16 digests per tree (in
    // sa), the second <= w / 256 (in sb, which starts w / 4 digests per tree into the scratch block)
    u64* sb = scratch ? scratch + size_t(batch) * 5 * size_t(w >> 2) : nullptr;
    for (; launches > 1; --launches) {
        const int k = (remaining + launches - 1) / launches, chunks_log = remaining - k;
        const long long nw = w >> k;
        u64* out = d_nodes ? nullptr : sa;
        hipLaunchKernelGGL(tfk::merkle_subtree_kernel, dim3((unsigned)(batch << chunks_log)), dim3(subtree_threads(k)), 0, s, level, in_ts, k,
                           chunks_log, d_nodes, nodes_ts, out, 5 * nw, copy_input ? 1 : 0, subtree_two_rows((long long)(batch << chunks_log)));
        HIPCHK(hipGetLastError());
        w = nw;
        remaining -= k;
        copy_input = false;
        if (d_nodes) {
            level = d_nodes + 5 * w;
            in_ts = nodes_ts;
        } else {
            level = sa;
            in_ts = 5 * w;
            std::swap(sa, sb);
        }
    }
    hipLaunchKernelGGL(tfk::merkle_top_kernel, dim3((unsigned)batch), dim3(subtree_threads(remaining)), 0, s, level, in_ts, (int)w, d_nodes,
                       nodes_ts, d_root, copy_input ? level : (const u64*)nullptr, in_ts, subtree_two_rows((long long)batch));
    HIPCHK(hipGetLastError());
    return TF_OK;
}

inline bool narrow_from(long long w, size_t batch) { return w <= kTopWidth || (w / 2) * (long long)batch <= kCoopMaxCount; }

// Levels above the leaf level for trees whose leaves are already at nodes[n..2n) (nodes[0] gets zeroed).
int merkle_levels_in_place(u64* d_nodes, long long N, size_t batch, hipStream_t s) {
    const long long nodes_ts = 10 * N;
    long long w = N;
    while (!narrow_from(w, batch)) {  // nodes[w/2 .. w) from nodes[w .. 2w)
        const long long nw = w / 2;
        int rc = launch_hash_pairs(d_nodes + 5 * w, d_nodes + 5 * nw, nullptr, nw * (long long)batch, nw, nodes_ts, nodes_ts, 0, s);
        if (rc) return rc;
        w = nw;
    }
    return merkle_narrow_levels(d_nodes + 5 * w, nodes_ts, w, d_nodes, nodes_ts, nullptr, nullptr, batch, false, s);
}

// hash_varlen of the rows of `batch` column-major tables (one codeword per column): digests to out + t * out_ts + 5 * i
int launch_hash_table_rows(const u64* table, long long n_rows, long long n_cols, int width, long long col_stride, long long table_stride,
                           long long batch, u64* out, long long out_ts, hipStream_t s) {
    const long long total = n_rows * batch;
    if (total == 0) return TF_OK;
    if (total <= kCoopMaxCount) {
        if (coop_two_rows(total))
            hipLaunchKernelGGL(tfk::tip5_hash_table_rows_coop_kernel<2>, dim3((unsigned)((total + 7) / 8)), dim3(256), 0, s, table, n_rows, shift_of(n_rows), n_cols,
                               width, col_stride, table_stride, total, out, out_ts);
        else
            hipLaunchKernelGGL(tfk::tip5_hash_table_rows_coop_kernel<1>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s, table, n_rows, shift_of(n_rows), n_cols,
                               width, col_stride, table_stride, total, out, out_ts);
    } else {
        hipLaunchKernelGGL(tfk::tip5_hash_table_rows_mx_kernel<1>, dim3(mx_blocks(total)), dim3(256), 0, s, table, n_rows, shift_of(n_rows),
                           n_cols, width, col_stride, table_stride, total, out, out_ts);
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
    if (narrow_from(N, batch))  // small trees: subtrees from the leaf level on (the leaves are copied into nodes[n..2n) on the way)
        return merkle_narrow_levels(d_leaves, leaves_ts, N, d_nodes, nodes_ts, nullptr, nullptr, batch, true, s);
    // first level: read leaves, write the leaf copy nodes[n..2n) and the parents nodes[n/2..n)
    long long w = N / 2;
    rc = launch_hash_pairs(d_leaves, d_nodes + 5 * w, d_nodes + 5 * N, w * (long long)batch, w, leaves_ts, nodes_ts,
                           nodes_ts, s);
    if (rc) return rc;
    while (!narrow_from(w, batch)) {  // nodes[w/2 .. w) from nodes[w .. 2w)
        const long long nw = w / 2;
        rc = launch_hash_pairs(d_nodes + 5 * w, d_nodes + 5 * nw, nullptr, nw * (long long)batch, nw, nodes_ts, nodes_ts, 0,
                               s);
        if (rc) return rc;
        w = nw;
    }
    return merkle_narrow_levels(d_nodes + 5 * w, nodes_ts, w, d_nodes, nodes_ts, nullptr, nullptr, batch, false, s);
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
    if (N <= kTopWidth) return merkle_narrow_levels(d_leaves, leaves_ts, N, nullptr, 0, d_root, nullptr, batch, false, s);
    // ping-pong level buffers: n/2 + n/4 digests per tree
    u64* buf = nullptr;
    const size_t words = size_t(batch) * size_t(5) * size_t(N / 2 + N / 4);
    hipError_t e = pool_malloc_async(reinterpret_cast<void**>(&buf), words * sizeof(u64), s);
    if (e != hipSuccess) {
        hip_fail(e, "pool_malloc_async(merkle levels)", __FILE__, __LINE__);
        return TF_ERR_TREE_TOO_HIGH;
    }
    if (narrow_from(N, batch)) {  // subtrees from the leaf level on
        rc = merkle_narrow_levels(d_leaves, leaves_ts, N, nullptr, 0, d_root, buf, batch, false, s);
    } else {
        u64* a = buf;
        u64* b = buf + size_t(batch) * 5 * size_t(N / 2);
        long long w = N / 2;
        rc = launch_hash_pairs(d_leaves, a, nullptr, w * (long long)batch, w, leaves_ts, 5 * w, 0, s);
        while (rc == TF_OK && !narrow_from(w, batch)) {
            const long long nw = w / 2;
            rc = launch_hash_pairs(a, b, nullptr, nw * (long long)batch, nw, 5 * w, 5 * nw, 0, s);
            std::swap(a, b);
            w = nw;
        }
        // `a` holds the level of w digests per tree; the other buffer (>= w / 2 digests per tree) is free for the subtree roots
        if (rc == TF_OK) rc = merkle_narrow_levels(a, 5 * w, w, nullptr, 0, d_root, b, batch, false, s);
    }
    e = hipFreeAsync(buf, s);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(e, "hipFreeAsync", __FILE__, __LINE__);
    return TF_OK;
}


// out[k] = nodes[idx[k]]: the digests of an authentication structure from a device-resident tree
int gather_digests_dev(const u64* d_nodes, const unsigned long long* d_idx, size_t count, u64* d_out, hipStream_t s) {
    if (count == 0) return TF_OK;
    const long long total = (long long)count * 5;
    hipLaunchKernelGGL(tfk::gather_digests_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_nodes, d_idx, (long long)count, d_out);
    HIPCHK(hipGetLastError());
    return TF_OK;
}

}  // namespace tfi
