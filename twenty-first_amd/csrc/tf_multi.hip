// tf_multi.hip -- the batch split over the GPUs of one node, at the C ABI (include/tf_hip.h: tf_*_multi, tf_set_device).
//
// The reference parallelises a batch from ONE process: rayon over the polynomials of a batch (math/ntt.rs:250-274 is called from
// par_iter loops of its users) and over the subtrees of a tree (util_types/merkle_tree.rs:165-212).  The drop-in for that shape is
// one call that takes a host-resident batch and a device list: the units (transforms, trees) are independent, so the batch is cut
// into contiguous slices -- slice g of G holds units [g B / G + min(g, B % G), ...), the rule of sharding.shard_range -- and one
// worker thread per slice runs the ordinary single-device host-pointer entry point on its device: allocate, H2D, compute, D2H on a
// stream of its own.  The workers run concurrently, so a host-resident batch crosses all the node's PCIe links at once; nothing is
// exchanged between devices (no RCCL: every result lands in the caller's buffer at its unit's offset).  The same device may be
// listed more than once (two workers on one GPU: two streams, two copies in flight) -- that is also how a one-GPU box tests this.
#include "tf_internal.h"

#include <thread>

namespace tfi {

void shard_range(size_t total, int shards, int shard, size_t* lo, size_t* hi) {
    const size_t base = total / (size_t)shards, extra = total % (size_t)shards, g = (size_t)shard;
    *lo = g * base + (g < extra ? g : extra);
    *hi = *lo + base + (g < extra ? 1 : 0);
}

// body(lo, hi) runs on the calling worker's current device and returns a TF status
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
    std::vector<int> rc((size_t)G, TF_OK);
    std::vector<std::string> err((size_t)G);
    std::vector<std::thread> workers;
    workers.reserve((size_t)G);
    for (int g = 0; g < G; ++g) {
        size_t lo, hi;
        shard_range(batch, G, g, &lo, &hi);
        if (lo == hi) continue;  // fewer units than workers
        workers.emplace_back([&, g, lo, hi] {
            hipError_t e = hipSetDevice(devs[(size_t)g]);
            if (e != hipSuccess) {
                rc[(size_t)g] = hip_fail(e, "hipSetDevice", __FILE__, __LINE__);
            } else {
                rc[(size_t)g] = body(lo, hi);
            }
            if (rc[(size_t)g] != TF_OK) err[(size_t)g] = t_last_error;
        });
    }
    for (auto& w : workers) w.join();
    for (int g = 0; g < G; ++g)
        if (rc[(size_t)g] != TF_OK) {  // the first failing slice (in batch order) reports
            t_last_error = "device " + std::to_string(devs[(size_t)g]) + " (slice " + std::to_string(g) + " of " + std::to_string(G) + "): " + err[(size_t)g];
            return rc[(size_t)g];
        }
    return TF_OK;
}

}  // namespace tfi

using namespace tfi;

extern "C" {

int tf_set_device(int device) {
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        return TF_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= visible || device >= kMaxDevices) return TF_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(device));
    return TF_OK;
}

int tf_get_device(int* device) {
    if (!device) return TF_ERR_NULL_POINTER;
    HIPCHK(hipGetDevice(device));
    return TF_OK;
}

int tf_shard_range(size_t total_units, int n_shards, int shard, size_t* begin, size_t* end) {
    if (!begin || !end) return TF_ERR_NULL_POINTER;
    if (n_shards <= 0 || shard < 0 || shard >= n_shards) return TF_ERR_NO_DEVICE;
    shard_range(total_units, n_shards, shard, begin, end);
    return TF_OK;
}

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
int tf_ntt_bfe_multi(uint64_t* x, size_t n, size_t batch, int inverse, const int* devices, int n_devices) {
    return ntt_multi(x, n, batch, 1, inverse, devices, n_devices);
}
int tf_ntt_xfe_multi(uint64_t* x, size_t n, size_t batch, int inverse, const int* devices, int n_devices) {
    return ntt_multi(x, n, batch, 3, inverse, devices, n_devices);
}

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
                            int n_devices) {
    return coset_eval_multi(c, nc, off, out, order, batch, 1, devices, n_devices);
}
int tf_coset_eval_xfe_multi(const uint64_t* c, size_t nc, uint64_t off, uint64_t* out, size_t order, size_t batch, const int* devices,
                            int n_devices) {
    return coset_eval_multi(c, nc, off, out, order, batch, 3, devices, n_devices);
}

int tf_merkle_build_multi(const uint64_t* leaves, size_t n, uint64_t* nodes_out, size_t batch, const int* devices, int n_devices) {
    int rc = tf_merkle_build(leaves, n, nodes_out, 0);  // leaf-count errors first, whatever the split
    if (rc) return rc;
    if (batch && (!leaves || !nodes_out)) return TF_ERR_NULL_POINTER;
    return run_on_devices(batch, devices, n_devices,
                          [=](size_t lo, size_t hi) { return tf_merkle_build(leaves + lo * n * 5, n, nodes_out + lo * n * 10, hi - lo); });
}
int tf_merkle_root_multi(const uint64_t* leaves, size_t n, uint64_t* root_out, size_t batch, const int* devices, int n_devices) {
    int rc = tf_merkle_root(leaves, n, root_out, 0);
    if (rc) return rc;
    if (batch && (!leaves || !root_out)) return TF_ERR_NULL_POINTER;
    return run_on_devices(batch, devices, n_devices,
                          [=](size_t lo, size_t hi) { return tf_merkle_root(leaves + lo * n * 5, n, root_out + lo * 5, hi - lo); });
}

}  // extern "C"
