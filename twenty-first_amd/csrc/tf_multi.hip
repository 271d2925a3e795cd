// tf_multi.hip -- the batch split over the GPUs of one node, at the C ABI (include/tf_hip.h: tf_*_multi, tf_set_device).
//
// The reference parallelises a batch from ONE process: rayon over the polynomials of a batch (math/ntt.rs:250-274 is called from
// par_iter loops of its users) and over the subtrees of a tree (util_types/merkle_tree.rs:165-212).  The drop-in for that shape is
// one call that takes a host-resident batch and a device list: the units (transforms, trees) are independent, so the batch is cut
// into contiguous slices -- slice g of G holds units [g B / G + min(g, B % G), ...), the rule of sharding.shard_range -- and one
// worker thread per slice (from a persistent pool, below) runs the ordinary single-device host-pointer entry point on its device:
// allocate, H2D, compute, D2H on a stream of its own.  The workers run concurrently, so a host-resident batch crosses all the node's PCIe links at once; nothing is
// exchanged between devices (no RCCL: every result lands in the caller's buffer at its unit's offset).  The same device may be
// listed more than once (two workers on one GPU: two streams, two copies in flight) -- that is also how a one-GPU box tests this.
#include "tf_internal.h"

#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <thread>

namespace tfi {

void shard_range(size_t total, int shards, int shard, size_t* lo, size_t* hi) {
    const size_t base = total / (size_t)shards, extra = total % (size_t)shards, g = (size_t)shard;
    *lo = g * base + (g < extra ? g : extra);
    *hi = *lo + base + (g < extra ? 1 : 0);
}

// Persistent workers.  Round 5 started one std::thread per slice per call; measured (tools/round_robin_latency.py,
// profiles/r06_round_robin_latency.txt) that costs ~0.55 ms per worker -- thread start plus the hipStreamCreate / hipStreamDestroy of the
// worker's private stream, which died with it: a 4 x 2^12-point call took 0.62 ms with one worker and 2.2 ms with four against 0.06 ms for
// the single-device call.  Now the library keeps a pool of detached worker threads, grown lazily to the largest device list seen (at most
// kMaxDevices); a call hands each slice to the pool and waits on a latch of its own, so concurrent callers share the workers, and a
// worker's per-device stream (host_stream() in tf_abi.hip is thread_local) lives as long as the process.  The pool is never destroyed:
// at process exit the workers are parked on a condition variable and simply end with the process (no HIP call from a static destructor).
class WorkerPool {
  public:
    // queue `task`; make sure at least `want` workers exist.  false only if NO worker exists and none could be started.
    bool submit(std::function<void()> task, int want) {
        std::unique_lock<std::mutex> lk(mu_);
        while (threads_ < want && threads_ < kMaxDevices) {
            try {
                std::thread([this] { run(); }).detach();
                ++threads_;
            } catch (...) {  // (std::system_error must not cross the C ABI: fewer workers serve the queue)
                break;
            }
        }
        if (threads_ == 0) return false;
        q_.push_back(std::move(task));
        lk.unlock();
        cv_.notify_one();
        return true;
    }

  private:
    void run() {
        for (;;) {
            std::function<void()> task;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return !q_.empty(); });
                task = std::move(q_.front());
                q_.pop_front();
            }
            task();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    int threads_ = 0;
};
WorkerPool& pool() {
    static WorkerPool* p = new WorkerPool;  // intentionally leaked (see above)
    return *p;
}

// body(lo, hi) runs on a worker whose current device is the slice's and returns a TF status
template <class F>
int run_on_devices(size_t batch, const int* devices, int n_devices, F&& body) {
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        t_last_error = "no usable HIP device";
        return TF_ERR_NO_DEVICE;
    }
    std::vector<int> devs;
    if (!devices) {  // every visible device, in order
        for (int d = 0; d < visible && d < kMaxDevices; ++d) devs.push_back(d);
    } else {
        if (n_devices <= 0) return TF_ERR_NO_DEVICE;
        devs.assign(devices, devices + n_devices);
    }
    for (int d : devs)
        if (d < 0 || d >= visible || d >= kMaxDevices) {
            t_last_error = "device index " + std::to_string(d) + " out of range (" + std::to_string(visible) + " visible)";
            return TF_ERR_NO_DEVICE;
        }
    if (batch == 0) return TF_OK;
    const int G = (int)devs.size();
    struct Call {  // shared with the workers: lives until the last of them has signalled
        std::mutex mu;
        std::condition_variable cv;
        int remaining = 0;
        std::vector<int> rc;
        std::vector<std::string> err;
    };
    auto call = std::make_shared<Call>();
    call->rc.assign((size_t)G, TF_OK);
    call->err.resize((size_t)G);
    int slices = 0;
    for (int g = 0; g < G; ++g) {
        size_t lo, hi;
        shard_range(batch, G, g, &lo, &hi);
        if (lo != hi) ++slices;  // (fewer units than workers: empty slices start nothing)
    }
    call->remaining = slices;
    bool no_worker = false;
    for (int g = 0; g < G; ++g) {
        size_t lo, hi;
        shard_range(batch, G, g, &lo, &hi);
        if (lo == hi) continue;
        const int dev = devs[(size_t)g];
        bool ok = false;
        try {  // (queueing allocates: a failure here must not unwind past the wait below, the slices already queued reference `body`)
          ok = pool().submit([call, g, lo, hi, dev, &body] {
            int rc;
            std::string msg;
            try {  // (a worker has no caller to unwind to: whatever is thrown here becomes the slice's status, and the latch below is always released)
                hipError_t e = hipSetDevice(dev);
                if (e != hipSuccess) rc = hip_fail(e, "hipSetDevice", __FILE__, __LINE__);
                else rc = body(lo, hi);
                if (rc != TF_OK) msg = t_last_error;
            } catch (const std::bad_alloc&) {
                rc = TF_ERR_OUT_OF_MEMORY;
            } catch (...) {
                rc = TF_ERR_INTERNAL;
            }
            std::lock_guard<std::mutex> lk(call->mu);
            call->rc[(size_t)g] = rc;
            call->err[(size_t)g].swap(msg);
            if (--call->remaining == 0) call->cv.notify_all();
          }, std::min(G, slices));
        } catch (...) {
            ok = false;
        }
        if (!ok) {  // not a single worker thread could be started, or the slice could not be queued: it will never run
            std::lock_guard<std::mutex> lk(call->mu);
            --call->remaining;
            no_worker = true;
        }
    }
    {
        std::unique_lock<std::mutex> lk(call->mu);  // (`body` is referenced by the queued tasks: wait for all of them, whatever happened)
        call->cv.wait(lk, [&] { return call->remaining == 0; });
    }
    if (no_worker) {
        t_last_error = "could not start a worker thread / queue a slice";
        return TF_ERR_HIP;
    }
    for (int g = 0; g < G; ++g)
        if (call->rc[(size_t)g] != TF_OK) {  // the first failing slice (in batch order) reports
            t_last_error = "device " + std::to_string(devs[(size_t)g]) + " (slice " + std::to_string(g) + " of " + std::to_string(G) + "): " + call->err[(size_t)g];
            return call->rc[(size_t)g];
        }
    return TF_OK;
}

}  // namespace tfi

using namespace tfi;

extern "C" {

int tf_set_device(int device) try {
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        return TF_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= visible || device >= kMaxDevices) return TF_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(device));
    return TF_OK;
} TF_ABI_CATCH

int tf_get_device(int* device) try {
    if (!device) return TF_ERR_NULL_POINTER;
    HIPCHK(hipGetDevice(device));
    return TF_OK;
} TF_ABI_CATCH

int tf_shard_range(size_t total_units, int n_shards, int shard, size_t* begin, size_t* end) try {
    if (!begin || !end) return TF_ERR_NULL_POINTER;
    if (n_shards <= 0 || shard < 0 || shard >= n_shards) return TF_ERR_INVALID_ARGUMENT;
    shard_range(total_units, n_shards, shard, begin, end);
    return TF_OK;
} TF_ABI_CATCH

static int ntt_multi(uint64_t* x, size_t n, size_t batch, int L, int inverse, const int* devices, int n_devices) {
    if (batch && n > 1 && !x) return TF_ERR_NULL_POINTER;
    // argument errors must not depend on the split: let one empty single-device call validate n
    int rc = L == 1 ? tf_ntt_bfe(x, n, 0, inverse) : tf_ntt_xfe(x, n, 0, inverse);
    if (rc) return rc;
    const size_t unit = n * (size_t)L;
    return run_on_devices(batch, devices, n_devices, [=](size_t lo, size_t hi) {
        return L == 1 ? tf_ntt_bfe(x + lo * unit, n, hi - lo, inverse) : tf_ntt_xfe(x + lo * unit, n, hi - lo, inverse);
    });
}
int tf_ntt_bfe_multi(uint64_t* x, size_t n, size_t batch, int inverse, const int* devices, int n_devices) try {
    return ntt_multi(x, n, batch, 1, inverse, devices, n_devices);
} TF_ABI_CATCH
int tf_ntt_xfe_multi(uint64_t* x, size_t n, size_t batch, int inverse, const int* devices, int n_devices) try {
    return ntt_multi(x, n, batch, 3, inverse, devices, n_devices);
} TF_ABI_CATCH

static int coset_eval_multi(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch, int L, const int* devices,
                            int n_devices) {
    int rc = L == 1 ? tf_coset_eval_bfe(c, nc, off, out, order, 0) : tf_coset_eval_xfe(c, nc, off, out, order, 0);
    if (rc) return rc;
    if (batch && order && (!out || (nc && !c))) return TF_ERR_NULL_POINTER;
    const size_t in_unit = nc * (size_t)L, out_unit = order * (size_t)L;
    return run_on_devices(batch, devices, n_devices, [=](size_t lo, size_t hi) {
        const uint64_t* ci = c ? c + lo * in_unit : nullptr;
        return L == 1 ? tf_coset_eval_bfe(ci, nc, off, out + lo * out_unit, order, hi - lo)
                      : tf_coset_eval_xfe(ci, nc, off, out + lo * out_unit, order, hi - lo);
    });
}
int tf_coset_eval_bfe_multi(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch, const int* devices,
                            int n_devices) try {
    return coset_eval_multi(c, nc, off, out, order, batch, 1, devices, n_devices);
} TF_ABI_CATCH
int tf_coset_eval_xfe_multi(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch, const int* devices,
                            int n_devices) try {
    return coset_eval_multi(c, nc, off, out, order, batch, 3, devices, n_devices);
} TF_ABI_CATCH

// Subtrees per tree when there are more listed devices than trees: the largest power of two S with batch * S <= devices, every subtree
// at least two leaves (the reference's own bound on its thread count, merkle_tree.rs:182: num_threads <= num_remaining_nodes / 2).
static size_t subtrees_per_tree(size_t n_leaves, size_t batch, int n_workers) {
    size_t S = 1;
    while (batch && 2 * S * batch <= (size_t)(n_workers > 0 ? n_workers : 0) && 2 * S <= n_leaves / 2) S *= 2;
    return S;
}
static int listed_workers(const int* devices, int n_devices) {
    if (devices) return n_devices;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return visible < kMaxDevices ? visible : kMaxDevices;
}
int tf_merkle_multi_subtrees(size_t n_leaves, size_t batch, int n_devices) { return (int)subtrees_per_tree(n_leaves, batch, n_devices); }
int tf_merkle_subtree_layer_range(size_t n_leaves, size_t n_subtrees, size_t subtree, unsigned layer, size_t* begin, size_t* end) try {
    if (!begin || !end) return TF_ERR_NULL_POINTER;
    if (int rc = check_leaves(n_leaves)) return rc;
    if (n_subtrees == 0 || (n_subtrees & (n_subtrees - 1)) || n_subtrees > n_leaves || subtree >= n_subtrees ||
        (n_leaves / n_subtrees) >> layer == 0)
        return TF_ERR_INVALID_ARGUMENT;
    *begin = (n_subtrees + subtree) << layer;
    *end = (n_subtrees + subtree + 1) << layer;
    return TF_OK;
} TF_ABI_CATCH

// One tree (or fewer trees than devices): every tree is cut into S subtrees exactly as MerkleTree::par_new cuts it over its threads
// (util_types/merkle_tree.rs:165-212, :247-275); unit u = tree u / S, subtree u % S; the units are dealt to the listed devices like any
// other batch, each worker writes its subtrees' layers into the caller's heap-ordered array, the S roots of a tree are gathered on the
// host (S x 40 bytes) and its top log2 S layers finish on the first listed device.  No collective library.
static int merkle_multi(const uint64_t* leaves, size_t n, uint64_t* nodes_out, uint64_t* root_out, size_t batch, const int* devices, int n_devices) {
    const size_t S = subtrees_per_tree(n, batch, listed_workers(devices, n_devices));
    if (S == 1) {
        if (nodes_out)
            return run_on_devices(batch, devices, n_devices,
                                  [=](size_t lo, size_t hi) { return tf_merkle_build(leaves + lo * n * 5, n, nodes_out + lo * n * 10, hi - lo); });
        return run_on_devices(batch, devices, n_devices,
                              [=](size_t lo, size_t hi) { return tf_merkle_root(leaves + lo * n * 5, n, root_out + lo * 5, hi - lo); });
    }
    const size_t m = n / S;
    std::vector<uint64_t> sub_roots(batch * S * 5);
    uint64_t* const sr = sub_roots.data();
    int rc = run_on_devices(batch * S, devices, n_devices, [=](size_t lo, size_t hi) {
        for (size_t u = lo; u < hi; ++u) {
            const size_t b = u / S, sub = u % S;
            int r = merkle_subtree_host(leaves + (b * n + sub * m) * 5, m, nodes_out ? nodes_out + b * n * 10 : nullptr, S, sub, sr + u * 5);
            if (r) return r;
        }
        return (int)TF_OK;
    });
    if (rc) return rc;
    // the top log2 S layers of every tree: the tree whose leaves are the S subtree roots, on the first listed device
    const int first = devices ? devices[0] : 0;
    std::vector<uint64_t> top(batch * 2 * S * 5);
    uint64_t* const tp = top.data();
    rc = run_on_devices(1, &first, 1, [=](size_t, size_t) { return tf_merkle_build(sr, S, tp, batch); });
    if (rc) return rc;
    for (size_t b = 0; b < batch; ++b) {
        const uint64_t* t = tp + b * 2 * S * 5;
        if (nodes_out) memcpy(nodes_out + b * n * 10, t, S * 5 * sizeof(uint64_t));  // nodes[0] (zero) and the S - 1 internal nodes above the subtrees
        else memcpy(root_out + b * 5, t + 5, 5 * sizeof(uint64_t));
    }
    return TF_OK;
}

int tf_merkle_build_multi(const uint64_t* leaves, size_t n, uint64_t* nodes_out, size_t batch, const int* devices, int n_devices) try {
    int rc = tf_merkle_build(leaves, n, nodes_out, 0);  // leaf-count errors first, whatever the split
    if (rc) return rc;
    if (batch && (!leaves || !nodes_out)) return TF_ERR_NULL_POINTER;
    return merkle_multi(leaves, n, nodes_out, nullptr, batch, devices, n_devices);
} TF_ABI_CATCH
int tf_merkle_root_multi(const uint64_t* leaves, size_t n, uint64_t* root_out, size_t batch, const int* devices, int n_devices) try {
    int rc = tf_merkle_root(leaves, n, root_out, 0);
    if (rc) return rc;
    if (batch && (!leaves || !root_out)) return TF_ERR_NULL_POINTER;
    return merkle_multi(leaves, n, nullptr, root_out, batch, devices, n_devices);
} TF_ABI_CATCH

}  // extern "C"
