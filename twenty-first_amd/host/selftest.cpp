// selftest.cpp -- the reference's own known-answer tests, written against the C++ host mirror
// (twenty_first.hpp), i.e. through the C ABI onto the GPU.  Each block names the Rust test it restates.
// Exit code 0 = all passed; 77 = no GPU (skipped); anything else = failure.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "twenty_first.hpp"

using namespace twenty_first;

static int failures = 0;
#define EXPECT(cond)                                                         \
    do {                                                                     \
        if (!(cond)) {                                                       \
            ++failures;                                                      \
            fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
        }                                                                    \
    } while (0)

static std::vector<BFieldElement> bfe_vec(std::initializer_list<uint64_t> v) {
    std::vector<BFieldElement> out;
    for (auto x : v) out.push_back(BFieldElement::new_(x));
    return out;
}

// Every device of the node driven through the C ABI from its own host thread (independent transforms and
// trees shard over the GPUs with no exchange step; the reference's callers are rayon workers, math/ntt.rs:250-274): thread i
// binds device devs[i] (hipSetDevice), uploads the same words, runs one tf_ntt_bfe_dev and one tf_merkle_build_dev on its own
// stream with that device's tables, and the results must equal the words device devs[0] produced alone beforehand.
// With one GPU the same harness runs two threads on device 0 (the per-device state is then shared, not replicated).
static void multi_device_check(const std::vector<int>& devs) {
    const size_t n = size_t(1) << 16, batch = 8, leaves = size_t(1) << 12;
    std::vector<uint64_t> x(n * batch), lv(5 * leaves);
    uint64_t st = 0x7F210005ull;
    auto next = [&]() { st += 0x9e3779b97f4a7c15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31; return z % 0xffffffff00000001ull; };
    for (auto& v : x) v = next();
    for (auto& v : lv) v = next();
    struct Result { std::vector<uint64_t> ntt, nodes; int rc = 0; };
    auto run = [&](int dev, Result* r) {
        r->ntt.resize(x.size());
        r->nodes.resize(10 * leaves);
        uint64_t *dx = nullptr, *dl = nullptr, *dn = nullptr;
        hipStream_t s = nullptr;
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreate(&s) != hipSuccess || hipMalloc(&dx, x.size() * 8) != hipSuccess ||
            hipMalloc(&dl, lv.size() * 8) != hipSuccess || hipMalloc(&dn, r->nodes.size() * 8) != hipSuccess) { r->rc = -1; return; }
        if (hipMemcpyAsync(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(dl, lv.data(), lv.size() * 8, hipMemcpyHostToDevice, s) != hipSuccess) { r->rc = -2; return; }
        for (int rep = 0; rep < 3 && r->rc == 0; ++rep) {  // forward, inverse, forward: the final words are one forward transform
            r->rc = tf_ntt_bfe_dev(dx, n, batch, rep == 1, s);
        }
        if (r->rc == 0) r->rc = tf_merkle_build_dev(dl, leaves, dn, 1, s);
        if (r->rc == 0 && (hipMemcpyAsync(r->ntt.data(), dx, x.size() * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
                           hipMemcpyAsync(r->nodes.data(), dn, r->nodes.size() * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
                           hipStreamSynchronize(s) != hipSuccess)) r->rc = -3;
        (void)hipFree(dx); (void)hipFree(dl); (void)hipFree(dn); (void)hipStreamDestroy(s);
    };
    Result ref;
    run(devs[0], &ref);
    EXPECT(ref.rc == 0);
    std::vector<Result> res(devs.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < devs.size(); ++i) th.emplace_back(run, devs[i], &res[i]);
    for (auto& t : th) t.join();
    for (size_t i = 0; i < devs.size(); ++i) {
        EXPECT(res[i].rc == 0);
        EXPECT(res[i].ntt == ref.ntt);
        EXPECT(res[i].nodes == ref.nodes);
        // one line of evidence per host thread / device, so that the first multi-GPU box leaves a record (README.md, "8 GPUs")
        const bool ok = res[i].rc == 0 && res[i].ntt == ref.ntt && res[i].nodes == ref.nodes;
        printf("  host thread %zu on device %d: %s (tf_ntt_bfe_dev 8 x 2^16 x3, tf_merkle_build_dev 2^12 leaves; rc %d, root word 0 %016llx)\n", i, devs[i],
               ok ? "PASS, same words as device 0 alone" : "FAIL", res[i].rc, (unsigned long long)(res[i].nodes.size() > 5 ? res[i].nodes[5] : 0));
    }
    (void)hipSetDevice(devs[0]);
}

// One host-resident batch over every device of the node in ONE C call (tf_ntt_bfe_multi / tf_merkle_root_multi / tf_coset_eval_bfe_multi,
// include/tf_hip.h): the words must be those of the single-device call on device devs[0].  13 units over the device list: a ragged
// split whatever the device count.
static void multi_call_check(const std::vector<int>& devs) {
    const size_t n = size_t(1) << 14, batch = 13, leaves = size_t(1) << 10, nc = 1000, order = size_t(1) << 11;
    std::vector<uint64_t> x(n * batch), lv(5 * leaves * batch), co(nc * batch);
    uint64_t st = 0x7F210006ull;
    auto next = [&]() { st += 0x9e3779b97f4a7c15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31; return z % 0xffffffff00000001ull; };
    for (auto& v : x) v = next();
    for (auto& v : lv) v = next();
    for (auto& v : co) v = next();
    (void)hipSetDevice(devs[0]);
    std::vector<uint64_t> x1 = x, xm = x, r1(5 * batch), rm(5 * batch), e1(order * batch), em(order * batch);
    const uint64_t off = BFieldElement::new_(7).raw;
    EXPECT(tf_ntt_bfe(x1.data(), n, batch, 0) == TF_OK);
    EXPECT(tf_merkle_root(lv.data(), leaves, r1.data(), batch) == TF_OK);
    EXPECT(tf_coset_eval_bfe(co.data(), nc, off, e1.data(), order, batch) == TF_OK);
    const int rc_n = tf_ntt_bfe_multi(xm.data(), n, batch, 0, devs.data(), (int)devs.size());
    const int rc_m = tf_merkle_root_multi(lv.data(), leaves, rm.data(), batch, devs.data(), (int)devs.size());
    const int rc_c = tf_coset_eval_bfe_multi(co.data(), nc, off, em.data(), order, batch, devs.data(), (int)devs.size());
    EXPECT(rc_n == TF_OK && xm == x1);
    EXPECT(rc_m == TF_OK && rm == r1);
    EXPECT(rc_c == TF_OK && em == e1);
    int cur = -1;
    EXPECT(tf_get_device(&cur) == TF_OK && cur == devs[0]);  // the caller's device is untouched
    for (size_t g = 0; g < devs.size(); ++g) {
        size_t lo = 0, hi = 0;
        EXPECT(tf_shard_range(batch, (int)devs.size(), (int)g, &lo, &hi) == TF_OK);
        const bool ok = rc_n == TF_OK && rc_m == TF_OK && rc_c == TF_OK &&
                        std::equal(xm.begin() + lo * n, xm.begin() + hi * n, x1.begin() + lo * n) &&
                        std::equal(rm.begin() + lo * 5, rm.begin() + hi * 5, r1.begin() + lo * 5) &&
                        std::equal(em.begin() + lo * order, em.begin() + hi * order, e1.begin() + lo * order);
        printf("  tf_*_multi slice %zu on device %d, units [%zu, %zu) of %zu: %s (tf_ntt_bfe_multi 2^14, tf_merkle_root_multi 2^10 leaves, tf_coset_eval_bfe_multi 1000 -> 2^11)\n",
               g, devs[g], lo, hi, batch, ok ? "PASS, same words as the single-device call" : "FAIL");
    }
}

// ONE tree over the device list in one C call (fewer trees than devices: tf_merkle_{build,root}_multi cut the tree into subtrees the way
// MerkleTree::par_new cuts it over its threads, util_types/merkle_tree.rs:165-212, :247-275): every node must be the single-device call's.
// On a one-GPU box the list is device 0 four times, so the split is exercised either way.
static void single_tree_multi_check(const std::vector<int>& devs_in) {
    std::vector<int> devs = devs_in;
    while (devs.size() < 4) devs.push_back(devs_in[devs.size() % devs_in.size()]);
    const size_t leaves = size_t(1) << 16;
    std::vector<uint64_t> lv(5 * leaves);
    uint64_t st = 0x7F210007ull;
    auto next = [&]() { st += 0x9e3779b97f4a7c15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31; return z % 0xffffffff00000001ull; };
    for (auto& v : lv) v = next();
    (void)hipSetDevice(devs[0]);
    std::vector<uint64_t> n1(10 * leaves), nm(10 * leaves, 0xABABABABABABABABull), r1(5), rm(5);
    EXPECT(tf_merkle_build(lv.data(), leaves, n1.data(), 1) == TF_OK);
    const int S = tf_merkle_multi_subtrees(leaves, 1, (int)devs.size());
    const int rc_b = tf_merkle_build_multi(lv.data(), leaves, nm.data(), 1, devs.data(), (int)devs.size());
    const int rc_r = tf_merkle_root_multi(lv.data(), leaves, rm.data(), 1, devs.data(), (int)devs.size());
    EXPECT(S >= 4 && rc_b == TF_OK && rc_r == TF_OK && nm == n1 && std::equal(rm.begin(), rm.end(), n1.begin() + 5));
    for (int sub = 0; sub < S; ++sub) {
        bool ok = rc_b == TF_OK;
        for (unsigned layer = 0; (leaves / (size_t)S) >> layer; ++layer) {
            size_t a = 0, b = 0;
            EXPECT(tf_merkle_subtree_layer_range(leaves, (size_t)S, (size_t)sub, layer, &a, &b) == TF_OK);
            ok = ok && std::equal(nm.begin() + a * 5, nm.begin() + b * 5, n1.begin() + a * 5);
        }
        size_t lo = 0, hi = 0;
        int worker = -1;
        for (size_t g = 0; g < devs.size(); ++g)
            if (tf_shard_range((size_t)S, (int)devs.size(), (int)g, &lo, &hi) == TF_OK && (size_t)sub >= lo && (size_t)sub < hi) worker = (int)g;
        printf("  one 2^16-leaf tree, subtree %d of %d on device %d (worker %d): %s\n", sub, S, worker >= 0 ? devs[(size_t)worker] : -1, worker,
               ok ? "PASS, every layer where MerkleTree::par_new puts it" : "FAIL");
    }
}

int main() {
    if (tf_device_count() <= 0) {
        fprintf(stderr, "no HIP device: the backend has no CPU fallback; skipping\n");
        try {
            auto x = bfe_vec({1, 4, 0, 0});
            ntt(x);
            fprintf(stderr, "ERROR: ntt succeeded without a device\n");
            return 1;
        } catch (const BackendError& e) {
            if (e.code != TF_ERR_NO_DEVICE) return 1;
        }
        return 77;
    }
    {  // bfield_basic_test_of_chu_ntt, math/ntt.rs:424-445
        auto x = bfe_vec({1, 4, 0, 0});
        auto orig = x;
        ntt(x);
        EXPECT(x == bfe_vec({5, 1125899906842625ULL, 18446744069414584318ULL, 18445618169507741698ULL}));
        intt(x);
        EXPECT(x == orig);
    }
    {  // bfield_max_value_test_of_chu_ntt, math/ntt.rs:448-469
        auto x = bfe_vec({BFieldElement::MAX, 0, 0, 0});
        ntt(x);
        EXPECT(x == bfe_vec({BFieldElement::MAX, BFieldElement::MAX, BFieldElement::MAX, BFieldElement::MAX}));
    }
    {  // xfield_basic_test_of_chu_ntt, math/ntt.rs:398-421
        std::vector<XFieldElement> x(4);
        x[0].coefficients[0] = BFieldElement::new_(1);
        auto orig = x;
        ntt(x);
        for (auto& e : x) EXPECT(e.coefficients[0] == BFieldElement::new_(1) && e.coefficients[1].raw == 0 && e.coefficients[2].raw == 0);
        intt(x);
        EXPECT(x == orig);
    }
    {  // ntt panics on a non-power-of-two length, math/ntt.rs:135-140
        auto x = bfe_vec({1, 2, 3});
        bool panicked = false;
        try { ntt(x); } catch (const NttPanic&) { panicked = true; }
        EXPECT(panicked);
        std::vector<BFieldElement> empty;
        ntt(empty);  // the empty slice is fine
    }
    {  // hash10_test_vectors_snapshot, tip5/mod.rs:1294-1306
        std::array<BFieldElement, 10> pre{};
        for (int i = 0; i < 6; ++i) {
            auto d = Tip5::hash_10(pre);
            for (int k = 0; k < 5; ++k) pre[i + k] = d[k];
        }
        EXPECT(Digest{Tip5::hash_10(pre)}.to_hex() == "109cc2fe453bd9962f754b96d8f5b919b60af030940a275f5540da195fef65ee651c1b6fa19b2c6a");
    }
    {  // hash_varlen_test_vectors, tip5/mod.rs:1309-1325 (sum of digests checked by the Python suite; here: padding sanity)
        std::vector<BFieldElement> none;
        Digest a = Tip5::hash_varlen(none);
        Digest b = Tip5::hash_varlen(bfe_vec({0}));
        EXPECT(!(a == b));
    }
    {  // snapshot, tip5/mod.rs:1328-1362 (raw Montgomery words)
        Tip5 t;
        const uint64_t st[16] = {0x0000000ffffffff0ULL, 0x00000000ffffffffULL, 0x00000000ffffffffULL, 0x00000028ffffffd7ULL,
                                 0x00000006fffffff9ULL, 0x00000002fffffffdULL, 0x00000000ffffffffULL, 0x00000030ffffffcfULL,
                                 0x00000397fffffc68ULL, 0x0000000ffffffff0ULL, 0x316bfb7236382123ULL, 0x216f521b66ef83f5ULL,
                                 0x5689d7b363f52df0ULL, 0xeb2f59e3aeae25fcULL, 0xb08299d277cbb4dcULL, 0xcbe3d9fdc5349140ULL};
        for (int i = 0; i < 16; ++i) t.state[i] = BFieldElement::from_raw_u64(st[i]);
        t.permutation();
        const uint64_t want[5] = {0x15d38ea929f6632aULL, 0xf988e509ff738bb4ULL, 0x48bcdfae88a2e9f3ULL, 0x87339e832daac02aULL, 0x511e41268150fdacULL};
        for (int i = 0; i < 5; ++i) EXPECT(t.state[i].raw == want[i]);
    }
    {  // MerkleTree structure + errors, util_types/merkle_tree.rs:393-429, :933-965, :1089-1116
        std::vector<Digest> leafs(8);
        for (size_t i = 0; i < leafs.size(); ++i) leafs[i] = Tip5::hash_varlen(bfe_vec({(uint64_t)i}));  // test_tree_of_height, :980-987
        MerkleTree tree = MerkleTree::par_new(leafs);
        EXPECT(tree.num_leafs() == 8 && tree.height() == 3);
        for (auto& v : tree.nodes[0].values) EXPECT(v.raw == 0);
        for (size_t i = 1; i < 8; ++i) EXPECT(tree.nodes[i] == Tip5::hash_pair(tree.nodes[2 * i], tree.nodes[2 * i + 1]));
        for (size_t i = 0; i < 8; ++i) EXPECT(*tree.leaf(i) == leafs[i]);
        EXPECT(MerkleTree::par_frugal_root(leafs) == tree.root());
        EXPECT(tree.root().to_hex() == MerkleTree::sequential_new(leafs).root().to_hex());
        bool e1 = false, e2 = false;
        try { MerkleTree::par_new({}); } catch (const MerkleTreeError& e) { e1 = e.variant == MerkleTreeError::TooFewLeafs; }
        leafs.pop_back();
        try { MerkleTree::par_new(leafs); } catch (const MerkleTreeError& e) { e2 = e.variant == MerkleTreeError::IncorrectNumberOfLeafs; }
        EXPECT(e1 && e2);
    }
    {  // fast_coset_evaluate == ntt of the scaled, zero-padded coefficients (polynomial.rs:1394-1396) and the order check
        auto c = bfe_vec({3, 1, 4, 1, 5});
        Polynomial<BFieldElement> p(c);
        auto ev = p.fast_coset_evaluate(BFieldElement::new_(1), 8);  // offset 1: plain ntt of the padded coefficients
        auto x = c;
        x.resize(8);
        ntt(x);
        EXPECT(ev == x);
        bool panicked = false;
        try { p.fast_coset_evaluate(BFieldElement::generator(), 4); } catch (const NttPanic&) { panicked = true; }
        EXPECT(panicked);
    }
    {  // fast_coset_interpolate inverts fast_coset_evaluate (polynomial.rs:3646-3662), fast_multiply against the schoolbook product
        auto c = bfe_vec({3, 1, 4, 1, 5, 9, 2, 6});
        Polynomial<BFieldElement> p(c);
        auto ev = p.fast_coset_evaluate(BFieldElement::generator(), 8);
        EXPECT(Polynomial<BFieldElement>::fast_coset_interpolate(BFieldElement::generator(), ev).coefficients == c);
        Polynomial<BFieldElement> a(bfe_vec({1, 2, 3})), b(bfe_vec({4, 5}));
        EXPECT(a.fast_multiply(b).coefficients == bfe_vec({4, 13, 22, 15}));
        EXPECT(a.fast_multiply(Polynomial<BFieldElement>({})).degree() == -1);
        // batch_evaluate: 1 + 2x + 3x^2 at 0, 1, 2, 10
        EXPECT(a.batch_evaluate(bfe_vec({0, 1, 2, 10})) == bfe_vec({1, 6, 17, 321}));
    }
    {  // zerofier doc example (polynomial.rs:1426-1434): roots 2, 4, 6 -> degree 3, zero exactly there; interpolate doc example
       // (:1490-1497): through (0,1) (1,3) (2,5) (3,7) -> 1 + 2x (degree 1, value 9 at 4)
        using Poly = Polynomial<BFieldElement>;
        auto roots = bfe_vec({2, 4, 6});
        Poly z = Poly::zerofier(roots);
        EXPECT(z.degree() == 3 && z.coefficients.back() == BFieldElement::new_(1));
        EXPECT(z.batch_evaluate(roots) == bfe_vec({0, 0, 0}));
        for (auto& v : z.batch_evaluate(bfe_vec({0, 1, 3, 5}))) EXPECT(!(v == BFieldElement{}));
        EXPECT(Poly::zerofier({}).coefficients == bfe_vec({1}));
        Poly f = Poly::interpolate(bfe_vec({0, 1, 2, 3}), bfe_vec({1, 3, 5, 7}));
        EXPECT(f.coefficients == bfe_vec({1, 2}));
        EXPECT(f.batch_evaluate(bfe_vec({4})) == bfe_vec({9}));
        EXPECT(Poly::interpolate(bfe_vec({5}), bfe_vec({42})).coefficients == bfe_vec({42}));  // one point: the constant (:3562-3570)
        auto both = Poly::batch_fast_interpolate(bfe_vec({0, 1, 2, 3}), {bfe_vec({1, 3, 5, 7}), bfe_vec({0, 1, 4, 9})});
        EXPECT(both.size() == 2 && both[0].coefficients == bfe_vec({1, 2}) && both[1].coefficients == bfe_vec({0, 0, 1}));
        bool p1 = false, p2 = false, p3 = false;
        try { Poly::interpolate({}, {}); } catch (const NttPanic& e) { p1 = e.code == TF_ERR_EMPTY_DOMAIN; }          // :3522-3526
        try { Poly::interpolate(bfe_vec({1, 2}), bfe_vec({1})); } catch (const NttPanic&) { p2 = true; }              // :3546-3552
        try { Poly::interpolate(bfe_vec({1, 1}), bfe_vec({1, 2})); } catch (const NttPanic& e) { p3 = e.code == TF_ERR_INVERSE_OF_ZERO; }  // :3554-3560
        EXPECT(p1 && p2 && p3);
    }
    {  // evaluate doc example (polynomial.rs:296-307): 2 + 5x + 12x^2 is 19 at 1 and xfe!(60) at 2, evaluated into the extension field
        Polynomial<BFieldElement> p(bfe_vec({2, 5, 12}));
        XFieldElement one{}, two{};
        one.coefficients[0] = BFieldElement::new_(1);
        two.coefficients[0] = BFieldElement::new_(2);
        auto v = p.evaluate_at({one, two});
        EXPECT(v[0].coefficients[0] == BFieldElement::new_(19) && v[1].coefficients[0] == BFieldElement::new_(60));
        EXPECT(v[1].coefficients[1] == BFieldElement{} && v[1].coefficients[2] == BFieldElement{});
    }
    {  // Tip5::trace (tip5/mod.rs:538-548, test :1557-1565): first row = the state, last row = the permutation's output
        Tip5 a = Tip5::init(), b = Tip5::init();
        for (int i = 0; i < 16; ++i) a.state[i] = b.state[i] = BFieldElement::new_(1000 + i);
        auto before = a.state;
        auto t = a.trace();
        b.permutation();
        EXPECT(t[0] == before && t[5] == b.state && a.state == b.state);
    }
    {  // ZerofierTree (zerofier_tree.rs): built once, used for evaluation and interpolation; the empty tree's zerofier is 1
        using Poly = Polynomial<BFieldElement>;
        auto tree = ZerofierTree<BFieldElement>::new_from_domain(bfe_vec({0, 1, 2, 3}));
        EXPECT(tree.zerofier().coefficients == Poly::zerofier(bfe_vec({0, 1, 2, 3})).coefficients);
        EXPECT(tree.batch_evaluate(Poly(bfe_vec({1, 2}))) == bfe_vec({1, 3, 5, 7}));
        EXPECT(tree.interpolate(bfe_vec({1, 3, 5, 7})).coefficients == bfe_vec({1, 2}));
        EXPECT(tree.interpolate(bfe_vec({0, 1, 4, 9})).coefficients == bfe_vec({0, 0, 1}));
        auto empty = ZerofierTree<BFieldElement>::new_from_domain({});
        EXPECT(empty.zerofier().coefficients == bfe_vec({1}) && empty.batch_evaluate(Poly(bfe_vec({1, 2}))).empty());
    }
    {  // clean_divide (polynomial.rs:2358-2411): (x + 1)(x + 2)(x + 3) / (x + 2), x^9 * 6 / (x^3 * 3), and the panics
        using Poly = Polynomial<BFieldElement>;
        Poly prod = Poly(bfe_vec({1, 1})).fast_multiply(Poly(bfe_vec({2, 1}))).fast_multiply(Poly(bfe_vec({3, 1})));
        EXPECT(prod.clean_divide(Poly(bfe_vec({2, 1}))).coefficients == bfe_vec({3, 4, 1}));
        EXPECT(Poly(bfe_vec({0, 0, 0, 0, 0, 0, 0, 0, 0, 6})).clean_divide(Poly(bfe_vec({0, 0, 0, 3}))).coefficients == bfe_vec({0, 0, 0, 0, 0, 0, 2}));
        EXPECT(Poly({}).clean_divide(Poly(bfe_vec({2, 1}))).degree() == -1);
        bool z = false, u = false;
        try { prod.clean_divide(Poly({})); } catch (const NttPanic& e) { z = e.code == TF_ERR_DIVISION_BY_ZERO; }
        try { Poly(bfe_vec({1, 0, 1})).clean_divide(Poly(bfe_vec({1, 1}))); } catch (const NttPanic& e) { u = e.code == TF_ERR_DIVISION_NOT_CLEAN; }
        EXPECT(z && u);
    }
    {  // batch_coset_extrapolate doc example, polynomial.rs:2183-2195: constant codewords extrapolate to the constant
        const size_t n = 32;
        std::vector<BFieldElement> codewords;
        for (size_t i = 0; i < n; ++i) codewords.push_back(BFieldElement::new_(3));
        for (size_t i = 0; i < n; ++i) codewords.push_back(BFieldElement::new_(2));
        auto got = Polynomial<BFieldElement>::batch_coset_extrapolate(BFieldElement::new_(7), n, codewords, bfe_vec({0, 1}));
        EXPECT(got == bfe_vec({3, 3, 2, 2}));
        bool panicked = false;
        try { Polynomial<BFieldElement>::batch_coset_extrapolate(BFieldElement::new_(7), 24, bfe_vec({1, 2, 3}), bfe_vec({0})); } catch (const NttPanic&) { panicked = true; }
        EXPECT(panicked);
    }
    {  // Sponge: hash_varlen == init, pad_and_absorb_all, first five of the state (tip5/mod.rs:617-623); squeeze returns the rate part
        auto input = bfe_vec({1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13});
        Tip5 sp = Tip5::init();
        sp.pad_and_absorb_all(input);
        Digest d = Tip5::hash_varlen(input);
        for (int i = 0; i < 5; ++i) EXPECT(sp.state[i] == d.values[i]);
        Tip5 before = sp;
        auto out = sp.squeeze();
        for (size_t i = 0; i < Tip5::RATE; ++i) EXPECT(out[i] == before.state[i]);
        before.permutation();
        EXPECT(before.state == sp.state);
    }
    {  // authentication structure of a height-3 tree for leaves {0, 5}: nodes 13, 5 ... (merkle_tree.rs:449-504), and from_rows
        std::vector<BFieldElement> rows;
        for (uint64_t i = 0; i < 8 * 3; ++i) rows.push_back(BFieldElement::new_(i));
        MerkleTree tree = MerkleTree::from_rows(rows, 3);
        for (size_t i = 0; i < 8; ++i) EXPECT(*tree.leaf(i) == Tip5::hash_varlen(std::vector<BFieldElement>(rows.begin() + 3 * i, rows.begin() + 3 * i + 3)));
        std::vector<BFieldElement> cols(rows.size());  // the same table column-major: column j = (rows[i * 3 + j])_i
        for (size_t i = 0; i < 8; ++i) for (size_t j = 0; j < 3; ++j) cols[j * 8 + i] = rows[i * 3 + j];
        EXPECT(MerkleTree::from_columns(cols, 8).nodes == tree.nodes);
        auto auth = tree.authentication_structure({0, 5});
        // needed: siblings 9, 12 and uncles 5, 7 minus computable ones -> {12, 9, 7, 5} descending (merkle_tree.rs:493-503)
        EXPECT(auth.size() == 4 && auth[0] == tree.nodes[12] && auth[1] == tree.nodes[9] && auth[2] == tree.nodes[7] && auth[3] == tree.nodes[5]);
        bool bad = false;
        // merkle_tree.rs:486-488: MerkleTreeError::LeafIndexInvalid -- the same type and variant as in the reference, and the
        // message of a non-HIP failure carries no stale HIP error text
        try { tree.authentication_structure({8}); } catch (const MerkleTreeError& e) {
            bad = e.code == TF_ERR_LEAF_INDEX_INVALID && e.variant == MerkleTreeError::LeafIndexInvalid && std::string(e.what()).find('(') == std::string::npos;
        }
        EXPECT(bad);
        // b_field_element.rs:264-268: the inverse of a zero offset panics in fast_coset_interpolate
        bool panicked = false;
        try { Polynomial<BFieldElement>::fast_coset_interpolate(BFieldElement::new_(0), bfe_vec({1, 2, 3, 4})); } catch (const NttPanic& e) { panicked = e.code == TF_ERR_INVERSE_OF_ZERO; }
        EXPECT(panicked);
    }
    {
        const int nd = tf_device_count();
        std::vector<int> devs;
        if (nd > 1) for (int d = 0; d < nd; ++d) devs.push_back(d);
        else devs = {0, 0};
        multi_device_check(devs);
        printf("C ABI from %zu host threads on %d device(s)%s: same words as one device alone\n", devs.size(), nd,
               nd > 1 ? "" : " (one GPU here: both threads on device 0; the per-device path first runs on a multi-GPU node)");
        multi_call_check(devs);
        printf("tf_*_multi over %zu worker(s) on %d device(s): same words as the single-device calls\n", devs.size(), nd);
        single_tree_multi_check(devs);
    }
    if (failures) {
        fprintf(stderr, "%d failure(s)\n", failures);
        return 1;
    }
    printf("twenty_first.hpp selftest: all reference KATs pass on the GPU\n");
    return 0;
}
