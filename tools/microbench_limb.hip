// microbench_limb.hip -- round-4 verdict item 1: the radix-32 network of the NTT pass kernels on CARRY-FREE REDUNDANT LIMBS,
// arithmetic only (no memory, no LDS), beside variant A of microbench_xlane.hip (what ntt_pass_kernel does today).
//
// An element is 4 signed limbs of 24 bits in 32-bit registers, value = l0 + l1 2^24 + l2 2^48 + l3 2^72 in Z / (2^96 + 1)
// (p = 2^64 - 2^32 + 1 divides 2^96 + 1, so the ring maps onto the field; the map is applied once per network, before the
// general product).  In this form
//   a +- b          = 4 v_add_u32 / v_sub_u32 (the 2-cycle class of profiles/r03_instr_rates.txt), no carries: five levels grow a limb
//                     by five bits, 24 + 5 < 31;
//   b * 2^(24 m)    = a rotation of the limbs with signs (2^96 = -1) that the next butterfly's add / sub absorbs: FREE
//                     (w_8 = 2^120, w_4 = 2^48: 29 of the 49 twiddles of a radix-32 network);
//   b * 2^(24 m + s), 0 < s < 24: split every limb at bit 24 - s (v_and, v_ashrrev: 2-cycle), (lo << s) + hi of the limb below
//                     (one v_lshl_add_u32 / v_mad_i32_i24: 4-cycle): 3 instructions a limb, 20 twiddles of the 49;
//   u64 -> limbs    = 4 instructions; limbs -> u64 (any 64-bit representative, which is what the Montgomery product takes) =
//                     a bias that makes the limbs positive, three v_mad_u64_u32 and two carry folds: 16 instructions (to_u64_x4).
// Variants timed (all end with one Montgomery product per element, the inner twiddle, as in the kernel):
//   A    32 elements / thread, u64 words, lazy paired butterflies + shl_fold / shl_monty twiddles      4 waves / SIMD  (today)
//   L32  32 elements / thread on limbs: 128 data VGPRs                                                   2 waves / SIMD
//   L16  16 elements / thread on limbs (radix-16 network: 64 data VGPRs, the occupancy of A)              4 waves / SIMD
//   A16  16 elements / thread, u64 words, radix-16 network (the like-for-like partner of L16)             4 and 8 waves / SIMD
// Every variant is first checked word for word against A's arithmetic on the same inputs (one iteration, all outputs).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I twenty-first_amd/csrc -o tools/microbench_limb tools/microbench_limb.hip
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

// ---------------------------------------------------------------------------------------------------------------- limb form
struct L4 {
    int l[4];
};
constexpr int kM24 = 0xffffff;

__device__ __forceinline__ L4 from_u64(u64 x) {  // canonical or not: any 64-bit word
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    L4 v;
    v.l[0] = (int)(lo & kM24);
    v.l[1] = (int)(__builtin_amdgcn_alignbit(hi, lo, 24) & kM24);
    v.l[2] = (int)(hi >> 16);
    v.l[3] = 0;
    return v;
}

// b * 2^S, 0 < S < 24: limb i = hi_i 2^(24 - S) + lo_i exactly (arithmetic shift), so limb i of the product is
// (lo_i << S) + hi_(i-1), and hi_3 wraps around to limb 0 with the sign of 2^96 = -1.
template <int S>
__device__ __forceinline__ L4 shl_sub(const L4& b) {
    if constexpr (S == 0) {
        return b;
    } else {
        int hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hi[i] = b.l[i] >> (24 - S), lo[i] = b.l[i] & ((1 << (24 - S)) - 1);
        L4 r;
        r.l[0] = (lo[0] << S) - hi[3];
#pragma unroll
        for (int i = 1; i < 4; ++i) r.l[i] = (lo[i] << S) + hi[i - 1];
        return r;
    }
}

// (a, b) -> (a + b 2^E, a - b 2^E); the rotation by E / 24 limbs and its signs are folded into the adds
template <int E>
__device__ __forceinline__ void bfly_limb(L4& a, L4& b) {
    constexpr int e = ((E % 192) + 192) % 192, m = e / 24, s = e % 24;
    const L4 t = shl_sub<s>(b);
    L4 sa, sd;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pos = (i + m) & 3;
        const bool neg = ((i + m) >> 2) & 1;
        sa.l[pos] = neg ? a.l[pos] - t.l[i] : a.l[pos] + t.l[i];
        sd.l[pos] = neg ? a.l[pos] + t.l[i] : a.l[pos] - t.l[i];
    }
    a = sa, b = sd;
}

template <int LVL, int I, int END>
struct LimbRange {
    static __device__ __forceinline__ void run(L4 (&v)[32]) {
        using B = Bf<false, LVL, I>;
        bfly_limb<B::E>(v[B::ia], v[B::ib]);
        if constexpr (I + 1 < END) LimbRange<LVL, I + 1, END>::run(v);
    }
};

// limbs -> one 64-bit word congruent to the element mod p (any representative: the Montgomery product takes it).  |l_i| < 2^30 - 2^7.
//   bias: l0 += 2^30 + 2^6, l1..l3 += 2^30 - 2^6 adds 2^30 (1 + 2^24 + 2^48 + 2^72) + 2^6 (1 - 2^24 - 2^48 - 2^72) = 0 (2^102 = -2^6):
//         every limb positive, the element unchanged;
//   B = l2 + l3 2^24 (one v_mad_u64_u32);  B 2^48 = (B << 16 mod 2^64) 2^32 + (B >> 48) 2^96 = (c1:c0) 2^32 - c2,  and
//   c1 2^64 = c1 EPS:   X = [l1 2^24 + (c0 : l0 - c2)] + c1 EPS   -- two v_mad_u64_u32 whose carry-outs (2^64 = EPS) are folded
//   back with two instructions each (lo -= k, hi += k & ~borrow: the sum is below 2^64 + 2^56, so it cannot carry again).
// Four elements per block, the carry chains issued round-robin (no wait-state nops), as gl::mont_mul4 does.
// 7 two-cycle + 9 four-cycle instructions per element.
__device__ __forceinline__ void to_u64_x4(const L4 (&v)[32], int q0, u64 (&x)[32]) {
    u64 A[4], ka[4];
    u32 c1[4];
    const u32 k24 = 1u << 24, eps = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const L4& e = v[q0 + i];
        const u32 l0 = (u32)(e.l[0] + ((1 << 30) + 64)), l1 = (u32)(e.l[1] + ((1 << 30) - 64));
        const u32 l2 = (u32)(e.l[2] + ((1 << 30) - 64)), l3 = (u32)(e.l[3] + ((1 << 30) - 64));
        const u64 B = (u64)l3 * k24 + l2;
        const u32 b0 = (u32)B, b1 = (u32)(B >> 32);
        const u32 c0 = b0 << 16;
        c1[i] = __builtin_amdgcn_alignbit(b1, b0, 16);
        const u32 l0p = l0 - (b1 >> 16);
        const u64 addend = ((u64)c0 << 32) | l0p;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(A[i]), "=s"(ka[i]) : "v"(l1), "s"(k24), "v"(addend));
    }
    u32 lo[4], hi[4];
    u64 n[4];
#define TF_L4_FIX(LO, HI, K)                                                                                                            \
    asm("v_subbrev_co_u32_e64 %[l0], %[n0], 0, %[a0], %[k0]\n\t"                                                                         \
        "v_subbrev_co_u32_e64 %[l1], %[n1], 0, %[a1], %[k1]\n\t"                                                                         \
        "v_subbrev_co_u32_e64 %[l2], %[n2], 0, %[a2], %[k2]\n\t"                                                                         \
        "v_subbrev_co_u32_e64 %[l3], %[n3], 0, %[a3], %[k3]\n\t"                                                                         \
        "s_andn2_b64 %[k0], %[k0], %[n0]\n\t"                                                                                            \
        "s_andn2_b64 %[k1], %[k1], %[n1]\n\t"                                                                                            \
        "s_andn2_b64 %[k2], %[k2], %[n2]\n\t"                                                                                            \
        "s_andn2_b64 %[k3], %[k3], %[n3]\n\t"                                                                                            \
        "v_addc_co_u32_e64 %[h0], %[n0], 0, %[b0], %[k0]\n\t"                                                                            \
        "v_addc_co_u32_e64 %[h1], %[n1], 0, %[b1], %[k1]\n\t"                                                                            \
        "v_addc_co_u32_e64 %[h2], %[n2], 0, %[b2], %[k2]\n\t"                                                                            \
        "v_addc_co_u32_e64 %[h3], %[n3], 0, %[b3], %[k3]"                                                                                 \
        : [l0] "=&v"(LO[0]), [l1] "=&v"(LO[1]), [l2] "=&v"(LO[2]), [l3] "=&v"(LO[3]), [h0] "=&v"(HI[0]), [h1] "=&v"(HI[1]),               \
          [h2] "=&v"(HI[2]), [h3] "=&v"(HI[3]), [n0] "=&s"(n[0]), [n1] "=&s"(n[1]), [n2] "=&s"(n[2]), [n3] "=&s"(n[3]), [k0] "+s"(K[0]),   \
          [k1] "+s"(K[1]), [k2] "+s"(K[2]), [k3] "+s"(K[3])                                                                               \
        : [a0] "v"((u32)A[0]), [a1] "v"((u32)A[1]), [a2] "v"((u32)A[2]), [a3] "v"((u32)A[3]), [b0] "v"((u32)(A[0] >> 32)),                \
          [b1] "v"((u32)(A[1] >> 32)), [b2] "v"((u32)(A[2] >> 32)), [b3] "v"((u32)(A[3] >> 32))                                          \
        : "scc")
    TF_L4_FIX(lo, hi, ka);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u64 a2 = ((u64)hi[i] << 32) | lo[i];
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(A[i]), "=s"(ka[i]) : "v"(c1[i]), "s"(eps), "v"(a2));
    }
    TF_L4_FIX(lo, hi, ka);
#undef TF_L4_FIX
#pragma unroll
    for (int i = 0; i < 4; ++i) x[q0 + i] = ((u64)hi[i] << 32) | lo[i];
}

template <int NEL>
__device__ __forceinline__ void limb_network(u64 (&x)[32]) {
    L4 v[32];
#pragma unroll
    for (int q = 0; q < NEL; ++q) v[q] = from_u64(x[q]);
    LimbRange<1, 0, NEL / 2>::run(v);
    LimbRange<2, 0, NEL / 2>::run(v);
    LimbRange<3, 0, NEL / 2>::run(v);
    LimbRange<4, 0, NEL / 2>::run(v);
    if constexpr (NEL == 32) LimbRange<5, 0, 16>::run(v);
#pragma unroll
    for (int q = 0; q < NEL; q += 4) to_u64_x4(v, q, x);
}

template <int NEL>
__device__ __forceinline__ void word_network(u64 (&x)[32]) {
    if constexpr (NEL == 32) {
        dit_half<false, 0, true>(x);
        dit_half<false, 16, true>(x);
        dit_level<false, 5, true>(x);
    } else {
        dit_half<false, 0, true>(x);
    }
}

// VARIANT 0: words, 1: limbs
template <int VARIANT, int NEL, int WPS>
__global__ void __launch_bounds__(256, WPS) net_kernel(u64* out, int iters, u64 seed, u64* full) {
    u64 x[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) x[q] = q < NEL ? seed_val(seed + (u64)(blockIdx.x * 256 + threadIdx.x) * NEL + q) : 0;
    const u64 w = seed_val(seed * 7 + 1);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == 0) word_network<NEL>(x);
        else limb_network<NEL>(x);
#pragma unroll
        for (int q = 0; q < NEL; q += 4) mul4_inplace(x, q, w, w, w, w);
    }
    if (full) {
#pragma unroll
        for (int q = 0; q < NEL; ++q) full[(size_t)(blockIdx.x * 256 + threadIdx.x) * NEL + q] = x[q];
    }
    u64 acc = 0;
#pragma unroll
    for (int q = 0; q < NEL; ++q) acc ^= x[q];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int VARIANT, int NEL, int WPS>
static double run(const char* name, int cus, u64* d_out, int iters) {
    const int grid = cus * WPS;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((net_kernel<VARIANT, NEL, WPS>), dim3(grid), dim3(256), 0, 0, d_out, iters, 12345ull, (u64*)nullptr);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double elems = (double)grid * 256 * NEL * iters;
    const double rate = elems / (ms * 1e-3) / 1e9;
    printf("%-52s %d waves/SIMD, %2d elements/thread: %8.3f ms  %8.1f G elements/s per network + product (chip)\n", name, WPS, NEL, ms, rate);
    return rate;
}

template <int NEL, int LWPS>
static bool check(int cus) {
    const int grid = 8;
    const size_t n = (size_t)grid * 256 * NEL;
    u64 *d_a, *d_b, *d_out;
    CK(hipMalloc(&d_a, n * 8));
    CK(hipMalloc(&d_b, n * 8));
    CK(hipMalloc(&d_out, (size_t)grid * 256 * 8));
    bool ok = true;
    for (int iters = 1; iters <= 3; ++iters) {  // several rounds: the second and third start from arbitrary canonical words
        hipLaunchKernelGGL((net_kernel<0, NEL, 4>), dim3(grid), dim3(256), 0, 0, d_out, iters, 777ull, d_a);
        hipLaunchKernelGGL((net_kernel<1, NEL, LWPS>), dim3(grid), dim3(256), 0, 0, d_out, iters, 777ull, d_b);
        CK(hipDeviceSynchronize());
        std::vector<u64> a(n), b(n);
        CK(hipMemcpy(a.data(), d_a, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), d_b, n * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += a[i] != b[i];
        printf("check radix-%d, %d iteration(s): %zu of %zu words differ between the word and the limb network%s\n", NEL, iters, bad, n,
               bad ? "  <-- MISMATCH" : "");
        ok = ok && bad == 0;
    }
    CK(hipFree(d_a));
    CK(hipFree(d_b));
    CK(hipFree(d_out));
    (void)cus;
    return ok;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    u64* d_out;
    CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));
    const bool ok = check<32, 2>(cus) & check<16, 4>(cus);
    const int iters = 2000;
    const double a = run<0, 32, 4>("A    words, radix 32 (ntt_pass_kernel today)", cus, d_out, iters);
    const double l32 = run<1, 32, 2>("L32  limbs 4 x 24 bit, radix 32", cus, d_out, iters);
    const double a16_4 = run<0, 16, 4>("A16  words, radix 16", cus, d_out, iters);
    const double a16_8 = run<0, 16, 8>("A16  words, radix 16", cus, d_out, iters);
    const double l16 = run<1, 16, 4>("L16  limbs 4 x 24 bit, radix 16", cus, d_out, iters);
    // the same limb kernels with the register allocation squeezed for one more wave or two (spills show up as time)
    const double l32_3 = run<1, 32, 3>("L32  limbs 4 x 24 bit, radix 32", cus, d_out, iters);
    const double l16_5 = run<1, 16, 5>("L16  limbs 4 x 24 bit, radix 16", cus, d_out, iters);
    const double l16_6 = run<1, 16, 6>("L16  limbs 4 x 24 bit, radix 16", cus, d_out, iters);
    printf("L32(3 waves) / A = %.3f   L16(5 waves) / A16(8 waves) = %.3f   L16(6 waves) / A16(8 waves) = %.3f\n", l32_3 / a, l16_5 / a16_8, l16_6 / a16_8);
    printf("L32 / A = %.3f   L16 / A16(4 waves) = %.3f   L16 / A16(8 waves) = %.3f\n", l32 / a, l16 / a16_4, l16 / a16_8);
    printf("per 2^20-point transform: radix 32 = 4 networks + 3 products, radix 16 = 5 networks + 4 products\n");
    return ok ? 0 : 2;
}
