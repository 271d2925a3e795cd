// aux_kernels.h -- debug / bench helpers of the ABI (tf_debug_*): the synthetic-input generator of SURVEY.md 8(d) and the
// shader-clock probe.  Included by tf_abi.hip only.
#pragma once

#include "gl64.h"

namespace tfk {

using gl::u32;
using gl::u64;

// out[0] = shader cycles, out[1] = wall-clock ticks spent in a fixed spin (tf_debug_sclk_mhz)
__global__ void sclk_probe_kernel(unsigned long long* out) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned v = threadIdx.x;
    for (int i = 0; i < 200000; ++i) v = v * 1664525u + 1013904223u;
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0 + (v == 0xdeadbeefu);
        out[1] = w1 - w0;
    }
}

// Synthetic inputs for benches and tests: element i = BFieldElement::new(splitmix64(seed ^ i) mod p), raw
// Montgomery word -- counter-based, so any slice can be regenerated; the oracle's tfo_fill_random is the same sequence.
__global__ void __launch_bounds__(256) fill_random_kernel(u64* out, unsigned long long count, u64 seed, unsigned long long first) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        u64 z = (seed ^ (first + i)) + 0x9e3779b97f4a7c15ULL;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        z ^= z >> 31;
        if (z >= gl::P) z -= gl::P;          // z mod p (z < 2^64 < 2p)
        out[i] = gl::mont_mul(z, gl::R2);    // BFieldElement::new (b_field_element.rs:235-237)
    }
}

}  // namespace tfk
