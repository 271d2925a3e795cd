// twenty_first.hpp -- C++ host-side mirror of the reference's Rust API for the hot path, on top of the
// C ABI of libtf_hip.so (include/tf_hip.h).  The reference is a Rust crate; no Rust toolchain exists in
// the build image, so the host layer a Rust maintainer would write as an `extern "C"` shim
//  is provided here in C++ with the same names, argument meaning and error behaviour:
//
//   twenty_first::ntt / intt                          math/ntt.rs:67-82, :109-125   (panics -> NttPanic)
//   twenty_first::Polynomial<FF>::fast_coset_evaluate math/polynomial.rs:1374-1399
//   twenty_first::Tip5::{hash_10, hash_pair, hash_varlen, permutation}   tip5/mod.rs:529-623
//   twenty_first::MerkleTree::{par_new, sequential_new, par_frugal_root, sequential_frugal_root}
//                                                     util_types/merkle_tree.rs:149-364
//   twenty_first::MerkleTreeError                     util_types/merkle_tree.rs:933-965
//
// Everything executes on the GPU through the C ABI; there is no CPU fallback in this header.
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tf_hip.h"

namespace twenty_first {

// #[repr(transparent)] over one u64 in Montgomery form (math/b_field_element.rs:84-86)
struct BFieldElement {
    uint64_t raw = 0;
    static constexpr uint64_t P = 0xffffffff00000001ULL;  // :225
    static constexpr uint64_t MAX = P - 1;

    static uint64_t montyred(unsigned __int128 x) {  // :357-370
        uint64_t xl = (uint64_t)x, xh = (uint64_t)(x >> 64);
        uint64_t a = xl + (xl << 32);
        uint64_t e = a < xl;
        uint64_t b = a - (a >> 32) - e;
        uint64_t r = xh - b;
        return xh < b ? r - 0xffffffffULL : r;
    }
    static BFieldElement new_(uint64_t value) {  // BFieldElement::new, :235-237
        return BFieldElement{montyred((unsigned __int128)value * 0xfffffffe00000001ULL)};
    }
    static BFieldElement from_raw_u64(uint64_t r) { return BFieldElement{r}; }
    uint64_t value() const { return montyred((unsigned __int128)raw); }  // :248-250
    uint64_t raw_u64() const { return raw; }
    static BFieldElement generator() { return new_(7); }  // :312-314
    bool operator==(const BFieldElement& o) const { return raw == o.raw; }
    bool operator!=(const BFieldElement& o) const { return raw != o.raw; }
};
static_assert(sizeof(BFieldElement) == 8, "layout contract of the C ABI");

// #[repr(transparent)] over [BFieldElement; 3] (math/x_field_element.rs:56-59)
struct XFieldElement {
    std::array<BFieldElement, 3> coefficients{};
    bool operator==(const XFieldElement& o) const { return coefficients == o.coefficients; }
};
static_assert(sizeof(XFieldElement) == 24, "layout contract of the C ABI");

struct Digest {  // tip5/digest.rs:29
    std::array<BFieldElement, 5> values{};
    static constexpr size_t LEN = 5;
    bool operator==(const Digest& o) const { return values == o.values; }
    std::string to_hex() const {  // canonical values, little-endian bytes (digest.rs:85-90, :144-153)
        static const char* hx = "0123456789abcdef";
        std::string s;
        for (auto& v : values) {
            uint64_t c = v.value();
            for (int b = 0; b < 8; ++b) {
                unsigned byte = (c >> (8 * b)) & 0xff;
                s.push_back(hx[byte >> 4]);
                s.push_back(hx[byte & 15]);
            }
        }
        return s;
    }
};
static_assert(sizeof(Digest) == 40, "layout contract of the C ABI");

// ---- errors -------------------------------------------------------------------------------------
struct BackendError : std::runtime_error {
    int code;
    BackendError(int c, const std::string& where)
        : std::runtime_error(where + ": " + tf_status_string(c) +
                             (((c >= TF_ERR_NO_DEVICE && c <= TF_ERR_OUT_OF_MEMORY) || c == TF_ERR_INTERNAL) ? std::string(" (") + tf_last_error() + ")" : "")),
          code(c) {}  // tf_last_error() belongs to the HIP failures (codes 8..10) and to an exception caught at the ABI (18) only
};
// where the reference panics (math/ntt.rs:135-140, math/polynomial.rs:1388-1392)
struct NttPanic : BackendError {
    using BackendError::BackendError;
};
// util_types/merkle_tree.rs:933-965
struct MerkleTreeError : BackendError {
    enum Variant { TooFewLeafs = 1, IncorrectNumberOfLeafs = 2, TreeTooHigh = 3, LeafIndexInvalid = 11 } variant;
    MerkleTreeError(int c, const std::string& where) : BackendError(c, where), variant((Variant)c) {}
};

inline void check(int rc, const char* where) {
    if (rc == TF_OK) return;
    if ((rc >= 1 && rc <= 3) || rc == TF_ERR_LEAF_INDEX_INVALID) throw MerkleTreeError(rc, where);  // merkle_tree.rs:933-965
    if ((rc >= 4 && rc <= 6) || rc == TF_ERR_INVERSE_OF_ZERO || (rc >= TF_ERR_EMPTY_DOMAIN && rc <= TF_ERR_DIVISION_NOT_CLEAN)) throw NttPanic(rc, where);  // the reference panics here
    throw BackendError(rc, where);
}

// ---- ntt / intt (math/ntt.rs:67-82, :109-125) ----------------------------------------------------
inline void ntt(std::vector<BFieldElement>& x) { check(tf_ntt_bfe(reinterpret_cast<uint64_t*>(x.data()), x.size(), 1, 0), "ntt"); }
inline void intt(std::vector<BFieldElement>& x) { check(tf_ntt_bfe(reinterpret_cast<uint64_t*>(x.data()), x.size(), 1, 1), "intt"); }
inline void ntt(std::vector<XFieldElement>& x) { check(tf_ntt_xfe(reinterpret_cast<uint64_t*>(x.data()), x.size(), 1, 0), "ntt"); }
inline void intt(std::vector<XFieldElement>& x) { check(tf_ntt_xfe(reinterpret_cast<uint64_t*>(x.data()), x.size(), 1, 1), "intt"); }
// many equal-length slices in one call (what a rayon caller of ntt() does, ntt.rs:250-274)
inline void ntt_batch(BFieldElement* x, size_t n, size_t batch, bool inverse = false) {
    check(tf_ntt_bfe(reinterpret_cast<uint64_t*>(x), n, batch, inverse), inverse ? "intt" : "ntt");
}
// ... and the same batch over several GPUs of the node: `devices` empty = every visible device (tf_ntt_bfe_multi: contiguous
// slices of the batch, one worker thread + stream per listed device, results in place at each slice's offset)
inline void ntt_batch_multi(BFieldElement* x, size_t n, size_t batch, const std::vector<int>& devices = {}, bool inverse = false) {
    check(tf_ntt_bfe_multi(reinterpret_cast<uint64_t*>(x), n, batch, inverse, devices.empty() ? nullptr : devices.data(), (int)devices.size()),
          inverse ? "intt" : "ntt");
}
inline void ntt_batch_multi(XFieldElement* x, size_t n, size_t batch, const std::vector<int>& devices = {}, bool inverse = false) {
    check(tf_ntt_xfe_multi(reinterpret_cast<uint64_t*>(x), n, batch, inverse, devices.empty() ? nullptr : devices.data(), (int)devices.size()),
          inverse ? "intt" : "ntt");
}

// ---- Polynomial (math/polynomial.rs:78-84): only the hot-path members ------------------------------
template <class FF>
struct Polynomial {
    std::vector<FF> coefficients;  // low -> high degree
    explicit Polynomial(std::vector<FF> c) : coefficients(std::move(c)) {
        while (!coefficients.empty() && coefficients.back() == FF{}) coefficients.pop_back();  // Polynomial::new normalises
    }
    long degree() const { return (long)coefficients.size() - 1; }
    // fast_coset_evaluate (polynomial.rs:1374-1399); offset is a BFieldElement (the documented fast case, :1366-1368)
    std::vector<FF> fast_coset_evaluate(BFieldElement offset, size_t order) const {
        if ((long)order <= degree()) throw NttPanic(TF_ERR_ORDER_NOT_ABOVE_DEGREE, "fast_coset_evaluate");  // :1388-1392
        std::vector<FF> out(order);
        const uint64_t* c = reinterpret_cast<const uint64_t*>(coefficients.data());
        uint64_t* o = reinterpret_cast<uint64_t*>(out.data());
        if constexpr (sizeof(FF) == 8)
            check(tf_coset_eval_bfe(c, coefficients.size(), offset.raw, o, order, 1), "fast_coset_evaluate");
        else
            check(tf_coset_eval_xfe(c, coefficients.size(), offset.raw, o, order, 1), "fast_coset_evaluate");
        return out;
    }
    // fast_coset_interpolate (polynomial.rs:1907-1918): values on {offset * w^i} -> the interpolant; panics unless the
    // number of values is a power of two
    static Polynomial fast_coset_interpolate(BFieldElement offset, const std::vector<FF>& values) {
        std::vector<FF> c(values.size());
        const uint64_t* v = reinterpret_cast<const uint64_t*>(values.data());
        uint64_t* o = reinterpret_cast<uint64_t*>(c.data());
        if constexpr (sizeof(FF) == 8)
            check(tf_coset_interpolate_bfe(v, values.size(), offset.raw, o, 1), "fast_coset_interpolate");
        else
            check(tf_coset_interpolate_xfe(v, values.size(), offset.raw, o, 1), "fast_coset_interpolate");
        return Polynomial(std::move(c));
    }
    // fast_multiply (polynomial.rs:900-932), same field on both sides; the zero polynomial annihilates
    Polynomial fast_multiply(const Polynomial& other) const {
        if (degree() < 0 || other.degree() < 0) return Polynomial({});
        std::vector<FF> out(coefficients.size() + other.coefficients.size() - 1);
        const uint64_t* a = reinterpret_cast<const uint64_t*>(coefficients.data());
        const uint64_t* b = reinterpret_cast<const uint64_t*>(other.coefficients.data());
        uint64_t* o = reinterpret_cast<uint64_t*>(out.data());
        if constexpr (sizeof(FF) == 8)
            check(tf_poly_mul_bfe(a, coefficients.size(), b, other.coefficients.size(), o, 1), "fast_multiply");
        else
            check(tf_poly_mul_xfe(a, coefficients.size(), b, other.coefficients.size(), o, 1), "fast_multiply");
        return Polynomial(std::move(out));
    }
    // batch_evaluate (polynomial.rs:1840-1852): f at every point of `domain`
    std::vector<FF> batch_evaluate(const std::vector<FF>& domain) const {
        std::vector<FF> out(domain.size());
        const uint64_t* c = reinterpret_cast<const uint64_t*>(coefficients.data());
        const uint64_t* d = reinterpret_cast<const uint64_t*>(domain.data());
        uint64_t* o = reinterpret_cast<uint64_t*>(out.data());
        if constexpr (sizeof(FF) == 8)
            check(tf_poly_batch_evaluate_bfe(c, coefficients.size(), d, domain.size(), o), "batch_evaluate");
        else
            check(tf_poly_batch_evaluate_xfe(c, coefficients.size(), d, domain.size(), o), "batch_evaluate");
        return out;
    }
    // clean_divide (polynomial.rs:2358-2411, BFieldElement only): self / divisor for a division known to be clean; panics on a
    // zero divisor and on an unclean division
    Polynomial clean_divide(const Polynomial& divisor) const {
        static_assert(sizeof(FF) == 8, "clean_divide is defined for Polynomial<BFieldElement> (polynomial.rs:2333)");
        const size_t na = coefficients.size(), nb = divisor.coefficients.size();
        std::vector<FF> out(na >= nb ? na - nb + 1 : 0);
        check(tf_poly_clean_divide_bfe(reinterpret_cast<const uint64_t*>(coefficients.data()), na,
                                       reinterpret_cast<const uint64_t*>(divisor.coefficients.data()), nb, reinterpret_cast<uint64_t*>(out.data())),
              "clean_divide");
        if (na < nb) out.clear();
        return Polynomial(std::move(out));
    }
    // zerofier (polynomial.rs:1435-1441, par_zerofier :1444-1459): the monic polynomial with exactly these roots
    static Polynomial zerofier(const std::vector<FF>& roots) {
        std::vector<FF> out(roots.size() + 1);
        const uint64_t* r = reinterpret_cast<const uint64_t*>(roots.data());
        uint64_t* o = reinterpret_cast<uint64_t*>(out.data());
        if constexpr (sizeof(FF) == 8) check(tf_poly_zerofier_bfe(r, roots.size(), o), "zerofier");
        else check(tf_poly_zerofier_xfe(r, roots.size(), o), "zerofier");
        return Polynomial(std::move(out));
    }
    // batch_fast_interpolate (polynomial.rs:1703-1731): one interpolant per value row over the same domain
    static std::vector<Polynomial> batch_fast_interpolate(const std::vector<FF>& domain, const std::vector<std::vector<FF>>& values_matrix) {
        if (domain.empty()) throw NttPanic(TF_ERR_EMPTY_DOMAIN, "interpolate");  // :1503-1506
        std::vector<FF> flat;
        flat.reserve(values_matrix.size() * domain.size());
        for (auto& row : values_matrix) {
            if (row.size() != domain.size()) throw NttPanic(TF_ERR_EMPTY_DOMAIN, "interpolate: the domain and values lists have to be of equal length");  // :1507-1511
            flat.insert(flat.end(), row.begin(), row.end());
        }
        std::vector<FF> out(flat.size());
        const uint64_t* d = reinterpret_cast<const uint64_t*>(domain.data());
        const uint64_t* v = reinterpret_cast<const uint64_t*>(flat.data());
        uint64_t* o = reinterpret_cast<uint64_t*>(out.data());
        if constexpr (sizeof(FF) == 8) check(tf_poly_interpolate_bfe(d, v, domain.size(), values_matrix.size(), o), "interpolate");
        else check(tf_poly_interpolate_xfe(d, v, domain.size(), values_matrix.size(), o), "interpolate");
        std::vector<Polynomial> polys;
        for (size_t r = 0; r < values_matrix.size(); ++r)
            polys.emplace_back(std::vector<FF>(out.begin() + r * domain.size(), out.begin() + (r + 1) * domain.size()));
        return polys;
    }
    // interpolate (polynomial.rs:1502-1520; par_interpolate :1525-1545): panics on an empty domain, unequal lengths, repeated points
    static Polynomial interpolate(const std::vector<FF>& domain, const std::vector<FF>& values) {
        return batch_fast_interpolate(domain, {values})[0];
    }
    // evaluate::<XFieldElement, XFieldElement> (polynomial.rs:309-320) of a base-field polynomial at extension-field points
    std::vector<XFieldElement> evaluate_at(const std::vector<XFieldElement>& points) const {
        static_assert(sizeof(FF) == 8, "the mixed-field evaluation is for Polynomial<BFieldElement>; use batch_evaluate otherwise");
        std::vector<XFieldElement> out(points.size());
        check(tf_poly_evaluate_bfe_at_xfe(reinterpret_cast<const uint64_t*>(coefficients.data()), coefficients.size(), 1,
                                          reinterpret_cast<const uint64_t*>(points.data()), points.size(), reinterpret_cast<uint64_t*>(out.data())),
              "evaluate");
        return out;
    }
    // batch_coset_extrapolate (polynomial.rs:2196-2208, par_ :2262): codeword-major values of every interpolant at
    // every point; panics unless codeword_length is a power of two
    static std::vector<FF> batch_coset_extrapolate(BFieldElement domain_offset, size_t codeword_length, const std::vector<FF>& codewords,
                                                   const std::vector<FF>& points) {
        const size_t batch = codeword_length ? codewords.size() / codeword_length : 0;
        std::vector<FF> out(batch * points.size());
        const uint64_t* c = reinterpret_cast<const uint64_t*>(codewords.data());
        const uint64_t* p = reinterpret_cast<const uint64_t*>(points.data());
        uint64_t* o = reinterpret_cast<uint64_t*>(out.data());
        if constexpr (sizeof(FF) == 8)
            check(tf_coset_extrapolate_bfe(domain_offset.raw, c, codeword_length, batch, p, points.size(), o), "batch_coset_extrapolate");
        else
            check(tf_coset_extrapolate_xfe(domain_offset.raw, c, codeword_length, batch, p, points.size(), o), "batch_coset_extrapolate");
        return out;
    }
    static std::vector<FF> coset_extrapolate(BFieldElement domain_offset, const std::vector<FF>& codeword, const std::vector<FF>& points) {  // :2117-2128
        return batch_coset_extrapolate(domain_offset, codeword.size(), codeword, points);
    }
};

// ---- barycentric_evaluate (math/polynomial.rs:2609-2637): every codeword of a batch at one indeterminate ------------------
template <class Coeff>
inline std::vector<XFieldElement> barycentric_evaluate(const std::vector<Coeff>& codewords, size_t codeword_length, XFieldElement indeterminate) {
    const size_t batch = codeword_length ? codewords.size() / codeword_length : 0;
    std::vector<XFieldElement> out(batch);
    const uint64_t* c = reinterpret_cast<const uint64_t*>(codewords.data());
    const uint64_t* x = reinterpret_cast<const uint64_t*>(&indeterminate);
    uint64_t* o = reinterpret_cast<uint64_t*>(out.data());
    if constexpr (sizeof(Coeff) == 8) check(tf_barycentric_evaluate_bfe(c, codeword_length, batch, x, o), "barycentric_evaluate");
    else check(tf_barycentric_evaluate_xfe(c, codeword_length, batch, x, o), "barycentric_evaluate");
    return out;
}

// ---- ZerofierTree (math/zerofier_tree.rs): the tree of a domain, built once and kept in HBM -----------------------------
template <class FF>
struct ZerofierTree {
    tf_zerofier_tree* handle = nullptr;
    size_t num_points = 0;
    ZerofierTree() = default;
    ZerofierTree(const ZerofierTree&) = delete;
    ZerofierTree& operator=(const ZerofierTree&) = delete;
    ZerofierTree(ZerofierTree&& o) noexcept : handle(o.handle), num_points(o.num_points) { o.handle = nullptr; }
    ~ZerofierTree() { tf_zerofier_tree_free(handle); }
    static ZerofierTree new_from_domain(const std::vector<FF>& domain) {  // :66-87
        ZerofierTree t;
        const uint64_t* d = reinterpret_cast<const uint64_t*>(domain.data());
        if constexpr (sizeof(FF) == 8) check(tf_zerofier_tree_new_bfe(d, domain.size(), &t.handle), "ZerofierTree::new_from_domain");
        else check(tf_zerofier_tree_new_xfe(d, domain.size(), &t.handle), "ZerofierTree::new_from_domain");
        t.num_points = domain.size();
        return t;
    }
    Polynomial<FF> zerofier() const {  // :93-99
        std::vector<FF> out(num_points + 1);
        check(tf_zerofier_tree_zerofier(handle, reinterpret_cast<uint64_t*>(out.data())), "ZerofierTree::zerofier");
        return Polynomial<FF>(std::move(out));
    }
    // polynomial.divide_and_conquer_batch_evaluate(&tree) (polynomial.rs:1882-1894)
    std::vector<FF> batch_evaluate(const Polynomial<FF>& f) const {
        std::vector<FF> out(num_points);
        check(tf_zerofier_tree_batch_evaluate(handle, reinterpret_cast<const uint64_t*>(f.coefficients.data()), f.coefficients.size(), 1,
                                              reinterpret_cast<uint64_t*>(out.data())),
              "divide_and_conquer_batch_evaluate");
        return out;
    }
    // the interpolant of `values` over the tree's domain (weights cached after the first call)
    Polynomial<FF> interpolate(const std::vector<FF>& values) {
        if (values.size() != num_points || num_points == 0) throw NttPanic(TF_ERR_EMPTY_DOMAIN, "interpolate");
        std::vector<FF> out(num_points);
        check(tf_zerofier_tree_interpolate(handle, reinterpret_cast<const uint64_t*>(values.data()), 1, reinterpret_cast<uint64_t*>(out.data())), "interpolate");
        return Polynomial<FF>(std::move(out));
    }
};

// ---- Tip5 (tip5/mod.rs) ---------------------------------------------------------------------------
struct Tip5 {
    std::array<BFieldElement, 16> state{};  // :159-165
    static constexpr size_t RATE = 10;
    void permutation() { check(tf_tip5_permute(reinterpret_cast<uint64_t*>(state.data()), 1), "Tip5::permutation"); }  // :529-533
    std::array<std::array<BFieldElement, 16>, 6> trace() {  // :538-548: the state before the permutation and after each round
        std::array<std::array<BFieldElement, 16>, 6> t;
        check(tf_tip5_trace(reinterpret_cast<uint64_t*>(state.data()), reinterpret_cast<uint64_t*>(t.data()), 1), "Tip5::trace");
        return t;
    }
    static std::array<BFieldElement, 5> hash_10(const std::array<BFieldElement, 10>& in) {  // :559-569
        std::array<BFieldElement, 5> out;
        check(tf_tip5_hash_pairs(reinterpret_cast<const uint64_t*>(in.data()), reinterpret_cast<uint64_t*>(out.data()), 1), "Tip5::hash_10");
        return out;
    }
    static Digest hash_pair(const Digest& l, const Digest& r) {  // :577-586
        std::array<BFieldElement, 10> in;
        for (int i = 0; i < 5; ++i) { in[i] = l.values[i]; in[5 + i] = r.values[i]; }
        return Digest{hash_10(in)};
    }
    static Digest hash_varlen(const std::vector<BFieldElement>& in) {  // :617-623
        Digest d;
        check(tf_tip5_hash_varlen_rows(reinterpret_cast<const uint64_t*>(in.data()), in.size(), 1, reinterpret_cast<uint64_t*>(d.values.data())), "Tip5::hash_varlen");
        return d;
    }
    // impl Sponge for Tip5 (:677-699) and Sponge::pad_and_absorb_all (util_types/sponge.rs:41-55)
    static Tip5 init() { return Tip5{}; }  // Domain::VariableLength: the all-zero state
    void absorb(const std::array<BFieldElement, RATE>& input) {
        for (size_t i = 0; i < RATE; ++i) state[i] = input[i];
        permutation();
    }
    std::array<BFieldElement, RATE> squeeze() {
        std::array<BFieldElement, RATE> produce;
        for (size_t i = 0; i < RATE; ++i) produce[i] = state[i];
        permutation();
        return produce;
    }
    void pad_and_absorb_all(const std::vector<BFieldElement>& input) {
        size_t i = 0;
        std::array<BFieldElement, RATE> chunk;
        for (; i + RATE <= input.size(); i += RATE) {
            for (size_t k = 0; k < RATE; ++k) chunk[k] = input[i + k];
            absorb(chunk);
        }
        chunk.fill(BFieldElement{});
        const size_t rem = input.size() - i;
        for (size_t k = 0; k < rem; ++k) chunk[k] = input[i + k];
        chunk[rem] = BFieldElement::from_raw_u64(0xffffffffULL);  // BFieldElement::ONE
        absorb(chunk);
    }
    // batched forms -- the reason to cross the boundary at all
    static std::vector<Digest> hash_pairs(const std::vector<Digest>& pairs) {  // pairs.size() even: (l0, r0, l1, r1, ...)
        std::vector<Digest> out(pairs.size() / 2);
        check(tf_tip5_hash_pairs(reinterpret_cast<const uint64_t*>(pairs.data()), reinterpret_cast<uint64_t*>(out.data()), out.size()), "Tip5::hash_pair");
        return out;
    }
};

// ---- MerkleTree (util_types/merkle_tree.rs:85-88) -----------------------------------------------------
struct MerkleTree {
    std::vector<Digest> nodes;  // nodes[0] dummy, nodes[1] root, leaves at nodes[n..2n)
    static MerkleTree par_new(const std::vector<Digest>& leafs) {  // :165-212
        MerkleTree t;
        t.nodes.resize(2 * leafs.size() + (leafs.empty() ? 1 : 0));
        check(tf_merkle_build(reinterpret_cast<const uint64_t*>(leafs.data()), leafs.size(), reinterpret_cast<uint64_t*>(t.nodes.data()), 1), "MerkleTree::par_new");
        return t;
    }
    static MerkleTree sequential_new(const std::vector<Digest>& leafs) { return par_new(leafs); }  // :149-153, same result
    // `batch` trees of n_leafs leaves each, split over the GPUs of the node (tf_merkle_root_multi); devices empty = all
    static std::vector<Digest> roots_multi(const std::vector<Digest>& leafs, size_t n_leafs, const std::vector<int>& devices = {}) {
        const size_t batch = n_leafs ? leafs.size() / n_leafs : 0;
        std::vector<Digest> roots(batch);
        check(tf_merkle_root_multi(reinterpret_cast<const uint64_t*>(leafs.data()), n_leafs, reinterpret_cast<uint64_t*>(roots.data()), batch,
                                   devices.empty() ? nullptr : devices.data(), (int)devices.size()), "MerkleTree::par_frugal_root");
        return roots;
    }
    static Digest sequential_frugal_root(const std::vector<Digest>& leafs) {  // :299-309
        Digest r;
        check(tf_merkle_root(reinterpret_cast<const uint64_t*>(leafs.data()), leafs.size(), reinterpret_cast<uint64_t*>(r.values.data()), 1), "MerkleTree::sequential_frugal_root");
        return r;
    }
    static Digest par_frugal_root(const std::vector<Digest>& leafs) {  // :332-364
        if (leafs.empty()) throw MerkleTreeError(TF_ERR_INCORRECT_NUMBER_OF_LEAFS, "MerkleTree::par_frugal_root");  // :333-335
        return sequential_frugal_root(leafs);
    }
    // hash_varlen of every row -> leaves -> tree in one device pipeline (the producer idiom of tip5/mod.rs:617-623)
    static MerkleTree from_rows(const std::vector<BFieldElement>& rows, size_t row_len) {
        const size_t n_rows = row_len ? rows.size() / row_len : 0;
        MerkleTree t;
        t.nodes.resize(2 * n_rows + (n_rows ? 0 : 1));
        check(tf_merkle_from_rows(reinterpret_cast<const uint64_t*>(rows.data()), row_len, n_rows, reinterpret_cast<uint64_t*>(t.nodes.data()), 1), "MerkleTree::par_new");
        return t;
    }
    // rows taken across a column-major table: `columns` = n_cols columns of n_rows elements back to back (one codeword each)
    template <class FF>
    static MerkleTree from_columns(const std::vector<FF>& columns, size_t n_rows) {
        constexpr int width = sizeof(FF) / 8;
        const size_t n_cols = n_rows ? columns.size() / n_rows : 0;
        MerkleTree t;
        t.nodes.resize(2 * n_rows + (n_rows ? 0 : 1));
        check(tf_merkle_from_columns(reinterpret_cast<const uint64_t*>(columns.data()), n_rows, n_cols, width, n_rows * width,
                                     reinterpret_cast<uint64_t*>(t.nodes.data()), 1), "MerkleTree::par_new");
        return t;
    }
    // authentication_structure (:614-622) over authentication_structure_node_indices (:449-504)
    std::vector<Digest> authentication_structure(const std::vector<size_t>& leaf_indices) const {
        std::vector<uint64_t> li(leaf_indices.begin(), leaf_indices.end()), idx(leaf_indices.size() * 64 + 1);
        size_t count = 0;
        check(tf_merkle_auth_structure_indices(num_leafs(), li.data(), li.size(), idx.data(), idx.size(), &count), "MerkleTree::authentication_structure");
        std::vector<Digest> out(count);
        for (size_t i = 0; i < count; ++i) out[i] = nodes[idx[i]];
        return out;
    }
    const Digest& root() const { return nodes[1]; }            // :624-626
    size_t num_leafs() const { return nodes.size() / 2; }       // :628-631
    unsigned height() const { unsigned h = 0; for (size_t n = num_leafs(); n > 1; n >>= 1) ++h; return h; }  // :633-636
    const Digest* node(size_t i) const { return i < nodes.size() ? &nodes[i] : nullptr; }  // :638-645
    const Digest* leaf(size_t i) const { return i < num_leafs() ? &nodes[num_leafs() + i] : nullptr; }  // :654-661
};

}  // namespace twenty_first
