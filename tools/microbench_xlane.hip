// microbench_xlane.hip -- the radix-32 network of the NTT pass kernels, two ways, arithmetic only (no memory, no LDS):
//   A  32 elements per thread, all five levels thread-local (what ntt_pass_kernel does), 4 waves per SIMD
//   B  16 elements per thread: levels 1-4 thread-local, level 5 ACROSS the lane pair (l, l + 32) with v_permlane32_swap
//      (north_star's wave-level shuffles), 8 waves per SIMD
// Both finish with one Montgomery product per element (the inner twiddle), as in the kernel.  Reported: elements per second for
// the whole chip and VALU instructions per element -- the measurement behind DESIGN.md's "+10 VALU per element and level".
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I twenty-first_amd/csrc -o tools/microbench_xlane tools/microbench_xlane.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ntt_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
using gl::u32;
using gl::u64;
using namespace tfk;

__device__ __forceinline__ u64 seed_val(u64 z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27;
    return z >= gl::P ? z - gl::P : z;
}

__global__ void __launch_bounds__(256, 4) net32_local(u64* out, int iters, u64 seed) {
    u64 x[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = seed_val(seed + (u64)(blockIdx.x * 256 + threadIdx.x) * 32 + q);
    const u64 w = seed_val(seed * 7 + 1);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        dit_half<false, 0, true>(x);
        dit_half<false, 16, true>(x);
        dit_level<false, 5, true>(x);
#pragma unroll
        for (int q = 0; q < 32; q += 4) mul4_inplace(x, q, w, w, w, w);
    }
    u64 acc = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) acc ^= x[q];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int J>
__device__ __forceinline__ u64 tw5(u64 v) {  // v * w_32^J as a true (canonical) value
    constexpr int E = TwExp<false, 5, J>::value;
    const u64 r = gl::Pow2Mul<E>::apply(v);
    return gl::Pow2Mul<E>::negate ? gl::neg(r) : r;
}
template <int J = 0>
__device__ __forceinline__ void tw5_all(u64 (&x)[32]) {
    x[J] = tw5<J>(x[J]);
    if constexpr (J + 1 < 16) tw5_all<J + 1>(x);
}
__device__ __forceinline__ void swap64(u64& a, u64& b) {  // lanes 32-63 of a  <->  lanes 0-31 of b
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    a = ((u64)r1[0] << 32) | r0[0];
    b = ((u64)r1[1] << 32) | r0[1];
}

__global__ void __launch_bounds__(256, 8) net32_xlane(u64* out, int iters, u64 seed) {
    u64 x[32];  // slots 0 .. 15 used
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = q < 16 ? seed_val(seed + (u64)(blockIdx.x * 256 + threadIdx.x) * 16 + q) : 0;
    const u64 w = seed_val(seed * 7 + 1);
    const bool upper = (threadIdx.x & 32) != 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        // levels 1-4 on the thread's 16 values (lower lanes: the a half of the 32-point network, upper lanes: the b half)
        DitRange<false, 1, 0, 8, true>::run(x);
        DitRange<false, 2, 0, 8, true>::run(x);
        DitRange<false, 3, 0, 8, true>::run(x);
        DitRange<false, 4, 0, 8, true>::run(x);
        // level 5: b_j * w_32^j on the upper lanes only (the lower lanes sit the instructions out), half the values change lanes,
        // eight butterflies per lane
        if (upper) tw5_all(x);
#pragma unroll
        for (int s = 0; s < 8; ++s) swap64(x[s], x[s + 8]);
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            u64 s0, d0, s1, d1;
            gl::add_sub_lazy2(x[s], x[s + 8], x[s + 1], x[s + 9], s0, d0, s1, d1);
            x[s] = s0, x[s + 8] = d0, x[s + 1] = s1, x[s + 9] = d1;
        }
#pragma unroll
        for (int q = 0; q < 16; q += 4) mul4_inplace(x, q, w, w, w, w);
    }
    u64 acc = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc ^= x[q];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    u64* d_out;
    CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));
    const int iters = 2000;
    for (int variant = 0; variant < 2; ++variant) {
        const int wps = variant ? 8 : 4, per_thread = variant ? 16 : 32;
        const int grid = cus * wps;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (variant) hipLaunchKernelGGL(net32_xlane, dim3(grid), dim3(256), 0, 0, d_out, iters, 12345ull);
            else hipLaunchKernelGGL(net32_local, dim3(grid), dim3(256), 0, 0, d_out, iters, 12345ull);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) {
                const double elems = (double)grid * 256 * per_thread * iters;
                printf("%s  %d waves/SIMD, %2d elements/thread: %8.3f ms  %8.1f G elements/s per radix-32 network + product (chip)\n",
                       variant ? "B cross-lane level 5 (v_permlane32_swap)" : "A thread-local                         ", wps, per_thread, ms,
                       elems / (ms * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
