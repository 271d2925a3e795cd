// ntt_args.h -- the argument records of the ntt_kernels.h launchers that other translation units fill in (the zerofier-tree
// walks of tf_poly.hip run their transforms through the latency-shaped kernels and the one-launch-per-level kernels of tf_ntt.hip).
// Plain data: no device code here.
#pragma once

#include "gl64.h"

namespace tfk {

using gl::u32;
using gl::u64;

struct NttLatArgs {
    const u64* in;
    u64* out;
    const u64* in2;        // or null: second operand laid out like `in`, multiplied in on load (L = 1 only)
    const u64* pre_scale;  // or null: input element idx of every slice and limb times pre_scale[idx] (Polynomial::scale fused into the
                           // load of fast_coset_evaluate, polynomial.rs:760-773, :1374-1399); load_mode 0 only
    const u64* post_scale; // or null: output element idx times post_scale[idx] (the offset^-idx of fast_coset_interpolate, :1907-1918);
                           // store_mode 0 only
    const u64* tw;         // [2][n]: w_n^(+-e), then n^-1 w_n^(+-e) (the inverse's last stage)
    long long n_coeffs;    // < 0: none; else elements >= n_coeffs read as zero
    long long in_bs, out_bs;  // words between consecutive slices
    long long total;       // limb transforms = batch * L
    u64 ninv;              // Montgomery n^-1 (inverse only)
    int L;
    // ---- the steps of a zerofier-tree walk that used to be kernels of their own, as modifiers of this kernel's load and store
    // (math/zerofier_tree.rs / polynomial.rs:1882-1894 remaindering; the tree code in tf_poly.hip says which step is which)
    int load_mode;         // 0: element idx of slice b is in[b * in_bs + idx * L]
                           // 1: REVERSED: in[(b >> src_shift) * in_bs + (rev_top - idx) * L] for idx < n_coeffs (poly_reverse /
                           //    remainder_rev_high fused into the forward transform that follows them)
                           // 2: (L = 1) the interpolation walk's parent N_l (Z_r + s) + N_r (Z_l + s), s = (-1)^idx, from the children's
                           //    transforms in[2 b], in[2 b + 1] and the level's cached tail transforms th[2 node], th[2 node + 1],
                           //    node = b % parents (interpolant_pointwise_kernel fused into the inverse transform that follows it)
    int src_shift;
    long long rev_top;
    const u64* th;
    long long parents;
    int store_mode;        // 0: all n outputs; 1: only outputs k < keep, stored as  sub_src[(b >> 1) * sub_bs + k * L] - value
                           //    (remainder_finish_kernel fused into the inverse transform in front of it: r = f_low - (q * tail)_low)
    const u64* sub_src;
    long long sub_bs, keep;
};

struct TreeLevelArgs {
    const u64* cur;   // down: remainders of the level above (lines / 2 polynomials of 2d coefficients); up: the children's interpolants
    u64* nxt;         // down: lines x d remainders; up: lines x 2d interpolants
    const u64* ghat;  // down only: the level's cached transforms of the reversed-zerofier inverses, [children][2d]
    const u64* that;  // the level's cached tail transforms, [children][2d]
    const u64* tw_f;  // ntt_lat_kernel's tables of order 2d, forward and inverse
    const u64* tw_i;
    u64 ninv;
    long long lines;  // down: units x children; up: rows x parents
    long long per;    // down: children; up: parents  (the cached transforms repeat with this period)
};

struct TreeBuildArgs {
    const u64* tails;   // level l: [children][d]
    const u64* inv;     // level l: [children][d]
    u64* that;          // level l: [children][2d]
    u64* ghat;          // level l: [children][2d]
    u64* ptails;        // level l + 1: [parents][2d]   (null: top level, phase 1 only)
    u64* pinv;          // level l + 1: [parents][2d]
    const u64 *tw_f2, *tw_i2, *tw_f4, *tw_i4;  // ntt_lat_kernel's tables of order 2d and 4d
    u64 ninv2, ninv4;
    long long parents;
};

}  // namespace tfk
