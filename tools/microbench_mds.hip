// microbench_mds.hip -- the Tip5 MDS layer (tip5/mod.rs:210-253) alone, one state per lane, two formulations:
//   A  the kernel's: plain circulant product on the two 32-bit halves of every word, 2 x 256 v_mad_u64_u32 whose addend absorbs
//      every accumulation, then one 85-bit recombination and reduction per output word
//   B  the shape of the reference's generated_function taken one CRT level deep, on three 22-bit limbs (the pre-additions
//      a_i +- a_{i+8} must stay below 32 bits, which two 32-bit halves do not): x^16 - 1 = (x^8 - 1)(x^8 + 1), the cyclic half
//      computed once (64 mads) and used as the addend of both signed negacyclic chains (2 x 64 v_mad_i64_i32) -- no post-additions
//      at all, 3 x 192 = 576 mads + limb split + pre-additions
// Both are checked against 128-bit arithmetic first.  The question: does the Karatsuba / CRT shape beat 512
// multiply-adds on this ISA?  On gfx950 a 64-bit addition costs as much as a multiply-add (profiles/r03_instr_rates.txt), so
// every saved product that needs a post-addition is a wash; the variant that needs none needs a third limb.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I twenty-first_amd/csrc -o tools/microbench_mds tools/microbench_mds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gl64.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
using gl::u32;
using gl::u64;
typedef long long i64;

__host__ __device__ constexpr u32 mds_entry(int i) {
    constexpr u32 col[16] = {61402, 1108, 28750, 33823, 7454, 43244, 53865, 12034, 56951, 27521, 41351, 40901, 12021, 59689, 26798, 17845};
    return col[i & 15];
}

// value (85 bits) = alo + ahi * 2^32  ->  canonical word   (the tail of tip5_round without the round constant)
__device__ __forceinline__ u64 fold85(u64 alo, u64 ahi) {
    unsigned c0, c1;
    const u32 w1 = __builtin_addc((u32)(alo >> 32), (u32)ahi, 0u, &c0);
    const u32 w2 = __builtin_addc((u32)(ahi >> 32), 0u, c0, &c1);
    const u64 l64 = ((u64)w1 << 32) | (u32)alo;
    const u64 t = (u64)w2 * 0xffffffffu + l64;
    const bool ca = t < l64;
    const u64 u = t + gl::EPS;
    const bool cb = u < t;
    return (ca | cb) ? u : t;
}

__device__ __forceinline__ void mds_a(u64 (&s)[16]) {
    u32 lo[16], hi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) lo[i] = (u32)s[i], hi[i] = (u32)(s[i] >> 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        u64 alo = 0, ahi = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const u32 m = mds_entry(16 + r - c);
            alo += (u64)m * lo[c];
            ahi += (u64)m * hi[c];
        }
        s[r] = fold85(alo, ahi);
    }
}

// one 22-bit limb vector a[16] -> 2 * (circulant product)[16] as signed 64-bit sums (the halving is folded into the recombination)
__device__ __forceinline__ void crt_limb(const u32 (&a)[16], i64 (&out)[16]) {
    int ap[8], am[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ap[i] = (int)(a[i] + a[i + 8]), am[i] = (int)a[i] - (int)a[i + 8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        i64 p = 0;  // cyclic half: sum_c (M[k] + M[k+8]) a+[c], k = (r - c) mod 8
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = (r - c) & 7;
            p += (i64)(int)(mds_entry(k) + mds_entry(k + 8)) * ap[c];
        }
        i64 q0 = p, q1 = p;  // negacyclic half with both signs, accumulated straight onto the cyclic one
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = (r - c) & 7;
            const int mm = (int)mds_entry(k) - (int)mds_entry(k + 8);
            const int sg = (r - c) < 0 ? -mm : mm;
            q0 += (i64)sg * am[c];
            q1 -= (i64)sg * am[c];
        }
        out[r] = q0;
        out[r + 8] = q1;
    }
}

__device__ __forceinline__ void mds_b(u64 (&s)[16]) {
    u32 l0[16], l1[16], l2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const u32 lo = (u32)s[i], hi = (u32)(s[i] >> 32);
        l0[i] = lo & 0x3fffffu;
        l1[i] = ((lo >> 22) | (hi << 10)) & 0x3fffffu;
        l2[i] = hi >> 12;
    }
    i64 o0[16], o1[16], o2[16];
    crt_limb(l0, o0);
    crt_limb(l1, o1);
    crt_limb(l2, o2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // 2 y = o0 + 2^22 o1 + 2^44 o2  (all non-negative: they are twice the plain sums): y = (o0 >> 1) + 2^21 o1 + 2^43 o2, < 2^85
        const u64 a = (u64)o0[r] >> 1, b = (u64)o1[r], c = (u64)o2[r];  // o0 is even whenever o1, o2 are integers... recombine exactly below
        // exact: 2y is even; form the 86-bit value v = o0 + (o1 << 22) + (o2 << 44) and halve
        unsigned __int128 v = (unsigned __int128)(u64)o0[r] + ((unsigned __int128)b << 22) + ((unsigned __int128)c << 44);
        v >>= 1;
        (void)a;
        const u64 vlo = (u64)v, vhi = (u64)(v >> 64);  // vhi < 2^21
        const u64 t = vhi * 0xffffffffull + vlo;
        const bool ca = t < vlo;
        const u64 u = t + gl::EPS;
        const bool cb = u < t;
        s[r] = (ca | cb) ? u : t;
    }
}

template <int V>
__global__ void __launch_bounds__(256) bench(u64* out, int iters, u64 seed) {
    u64 s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        u64 z = seed + (u64)(blockIdx.x * 256 + threadIdx.x) * 16 + i;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        s[i] = z ^ (z >> 27);
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (V == 0) mds_a(s);
        else mds_b(s);
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= s[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void check(const u64* in, int* bad) {
    u64 a[16], b[16], x[16];
    for (int i = 0; i < 16; ++i) x[i] = a[i] = b[i] = in[(blockIdx.x * blockDim.x + threadIdx.x) * 16 + i];
    mds_a(a);
    mds_b(b);
    for (int r = 0; r < 16; ++r) {
        unsigned __int128 acc = 0;
        for (int c = 0; c < 16; ++c) acc += (unsigned __int128)mds_entry(16 + r - c) * x[c];
        const u64 want = (u64)(acc % gl::P);
        if (a[r] != want) atomicOr(bad, 1);
        if (b[r] != want) atomicOr(bad, 2);
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int n = 1 << 14;
    std::vector<u64> h(n * 16);
    u64 st = 99;
    for (auto& v : h) { st += 0x9e3779b97f4a7c15ULL; u64 z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; v = z ^ (z >> 31); }
    for (int i = 0; i < 64; ++i) h[i] = (i & 1) ? 0xffffffffffffffffULL : 0xffffffff00000000ULL;  // extreme words (S-box outputs may exceed p)
    u64* d_in;
    int* d_bad;
    CK(hipMalloc(&d_in, h.size() * 8));
    CK(hipMalloc(&d_bad, 4));
    CK(hipMemset(d_bad, 0, 4));
    CK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad);
    int bad = 0;
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("MDS of %d random / extreme states against 128-bit arithmetic: %s (mask %d: 1 = halves, 2 = three-limb CRT)\n", n, bad ? "MISMATCH" : "both bit-exact", bad);
    u64* d_out;
    CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));
    const int iters = 400, grid = cus * 8;
    for (int v = 0; v < 2; ++v) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(bench<0>, dim3(grid), dim3(256), 0, 0, d_out, iters, 7ull);
            else hipLaunchKernelGGL(bench<1>, dim3(grid), dim3(256), 0, 0, d_out, iters, 7ull);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%s: %8.3f ms for %d x %d MDS layers = %7.2f G MDS/s\n", v ? "B three 22-bit limbs, one CRT level (576 mads)" : "A two 32-bit halves, plain circulant (512 mads)",
               ms, grid * 256, iters, (double)grid * 256 * iters / (ms * 1e-3) / 1e9);
    }
    return bad ? 1 : 0;
}
