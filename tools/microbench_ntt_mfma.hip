// microbench_ntt_mfma.hip -- a radix-16 shift-only NTT stage on the i8 matrix pipe, arithmetic only (no memory, one LDS table), against
// the thread-local radix-32 network of the pass kernels (variant A of tools/microbench_xlane.hip).  The question (DESIGN.md section 8
// item 2, asked after the Tip5 MDS moved to v_mfma_i32_16x16x64_i8 in round 6): the NTT pass kernels are bound by vector-ALU issue --
// would the matrix pipe take the shift-only networks off it?
//
// A radix-16 stage is a constant matrix: X[r] = sum_c x[c] w^(r c), w = w_16 = 2^12 (2^96 = -1 mod p, so w^16 = 2^192 = 1).  Entry
// W[r][c] = 2^(12 (r c mod 16)) = +-2^s with s in {0, 12, ..., 84} (s >= 96: 2^s = -2^(s - 96)), s = 8 a + t, t in {0, 4}: the entry is +-1 or
// +-16 at byte offset a in {0, 1, 3, 4, 6, 7, 9, 10}.  With the input words cut into bytes d_b (b = 0..7), output byte plane p = a + b
// collects +-(1 << t) d_b[c]; planes 12..17 wrap onto 0..5 with the opposite sign (2^96 = -1), so twelve planes come back:
//     P_p[r][j] = sum_c sum_b [ (a(r,c) + b) mod 12 == p ] * (+-)(1 << t(r,c)) * d_b[c][j],      X[r] = sum_p 256^p P_p  (mod 2^96 + 1, hence mod p).
// On v_mfma_i32_16x16x64_i8 (same conventions as tip5_kernels.h: lane & 15 = row of A / column of B and D, K indexed by the same function of
// (lane >> 4, byte) on both sides, D row = 4 (lane >> 4) + register):
//   lane (j, q) holds elements 4 q .. 4 q + 3 of column j (16 independent 16-point DFTs per wave: in a pass kernel, 16 transforms of the batch
//   at one position);  B_lo = the four low dwords XOR 0x80808080 (bytes 0..3, biased to i8), B_hi = the high dwords: no byte shuffling;
//   A_p (p = 0..11) for B_lo;  for B_hi (bytes 4..7) plane p takes A_{p-4} (p >= 4) and the NEGATED A_{p+8} (p < 4: the wrap) -- sixteen
//   constant operands, 64 VGPRs;  24 MFMA per 256 elements and stage;
//   C_p (from LDS) = a bias that keeps every plane positive + the bytes of the constant that undoes bias and XOR (only row 0 sees the XOR:
//   sum_c W[r][c] = 16 [r == 0]);
//   recombination per element: L0, L1, L2 = planes 0..3, 4..7, 8..11 (one v_lshl_add_u32 + two v_mad_u64_u32 each),
//   X = L0 + 2^32 L1 + 2^64 L2 = (L0 - L2) + 2^32 (L1 + L2)  (2^64 = 2^32 - 1): one 64-bit add, one 64-bit subtract, one v_mad_u64_u32 for
//   the part of 2^32 (L1 + L2) above 2^64, then the "add to the high word, fold the carry" tail of the Tip5 round (tfk::mx_fold4_tail).
// Every output is checked against 128-bit host arithmetic first.  Then: stage + one Montgomery product per element (the general twiddle
// between stages), iterated in registers, G elements/s for the chip -- beside the radix-32 network + product (five levels per product where
// this has four: a 2^20-point plan is 4 x radix 32 or 5 x radix 16).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I twenty-first_amd/csrc -o tools/microbench_ntt_mfma tools/microbench_ntt_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ntt_kernels.h"
#include "tip5_kernels.h"  // tfk::mx_fold4_tail, v4i
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
using gl::u32;
using gl::u64;
using namespace tfk;
typedef unsigned __int128 u128;

struct Dft16Consts {
    int a[16][64][4];  // A operands: 0..11 = plane p for B_lo; 12..15 = the negated planes 8..11 (B_hi's planes 0..3)
    int c[12][4][4];   // accumulator starts [plane][quarter q][register t] for output row 4 q + t
};
__constant__ Dft16Consts g_dft16;

constexpr int kBiasLow = 1 << 18, kBiasHigh = 1 << 16;  // planes 0..3 / 4..11: every plane positive, and L0 >= L2 whatever the data

static void entry(int r, int c, int& a, int& val) {  // W[r][c] = val * 256^a, val in {+-1, +-16}
    int s = 12 * ((r * c) & 15);
    int sign = 1;
    if (s >= 96) s -= 96, sign = -1;
    a = s >> 3;
    val = sign * (1 << (s & 7));
}

static void fill_consts(Dft16Consts& t) {
    memset(&t, 0, sizeof t);
    // lo operand of plane p: K slot (q, i): element c = 4 q + (i >> 2), byte b = i & 3
    for (int p = 0; p < 12; ++p)
        for (int l = 0; l < 64; ++l) {
            const int r = l & 15, q = l >> 4;
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * q + e;
                int a, val;
                entry(r, c, a, val);
                u32 w = 0;
                for (int b = 0; b < 4; ++b) {
                    const int pl = a + b;  // < 12 + 4
                    int v = 0;
                    if (pl == p) v = val;
                    else if (pl - 12 == p) v = -val;
                    w |= ((u32)v & 0xff) << (8 * b);
                }
                t.a[p][l][e] = (int)w;
            }
        }
    // bytes 4..7 against plane p: the lo operand of plane p - 4; for p < 4 that is plane p + 8 one wrap further on: negated
    for (int p = 0; p < 4; ++p)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const u32 w = (u32)t.a[p + 8][l][e];
                u32 n = 0;
                for (int b = 0; b < 4; ++b) n |= ((u32)(-(int)(signed char)((w >> (8 * b)) & 0xff)) & 0xff) << (8 * b);
                t.a[12 + p][l][e] = (int)n;
            }
    // starts: sum_p 256^p C_p = K_bias + x_r with x_r = (128 * ones8 * 16 [r == 0] - K_bias) mod p  ->  recombined value = X[r] exactly (mod p)
    u128 kb = 0;
    for (int p = 11; p >= 0; --p) kb = (kb * 256 + (u128)(p < 4 ? kBiasLow : kBiasHigh)) % gl::P;
    const u128 ones8 = 0x0101010101010101ULL;
    for (int q = 0; q < 4; ++q)
        for (int v = 0; v < 4; ++v) {
            const int r = 4 * q + v;
            u128 x = (r == 0 ? (u128)(128 * 16) * (ones8 % gl::P) % gl::P : 0) + gl::P - kb;
            const u64 xr = (u64)(x % gl::P);
            for (int p = 0; p < 12; ++p) t.c[p][q][v] = (p < 4 ? kBiasLow : kBiasHigh) + (p < 8 ? (int)((xr >> (8 * p)) & 0xff) : 0);
        }
}

struct Dft16A {
    v4i p[16];
};
__device__ __forceinline__ void load_a(Dft16A& a) {
#pragma unroll
    for (int p = 0; p < 16; ++p) a.p[p] = *reinterpret_cast<const v4i*>(&g_dft16.a[p][threadIdx.x & 63][0]);
}

// one radix-16 stage on the wave's 16 columns: x[0..3] = elements 4 q .. 4 q + 3 of this lane's column, any 64-bit representatives in,
// element 0 canonical / 1..3 any representative out (mx_fold4_tail<false>)
__device__ __forceinline__ void dft16_stage(u64 (&x)[4], const Dft16A& a, const int (*lc)[4][4], int q) {
    v4i blo, bhi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        blo[i] = (int)((u32)x[i] ^ 0x80808080u);
        bhi[i] = (int)((u32)(x[i] >> 32) ^ 0x80808080u);
    }
    v4i d[12];
#pragma unroll
    for (int p = 0; p < 12; ++p) {
        const v4i c = *reinterpret_cast<const v4i*>(&lc[p][q][0]);
        d[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.p[p], blo, c, 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < 12; ++p) d[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.p[p >= 4 ? p - 4 : 12 + p], bhi, d[p], 0, 0, 0);
    u32 k16 = 1u << 16, k24 = 1u << 24, kff = 0xffffffffu;
    asm volatile("" : "+s"(k16), "+s"(k24), "+s"(kff));  // (opaque multipliers: the products stay single v_mad_u64_u32, as in tip5_round_mx)
    const auto mad = [](u32 xx, u32 y, u64 z) { return (u64)xx * y + z; };
    u32 tl[4], th[4], h0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const auto Q = [&](int p) { return (u32)d[p][t]; };
        u64 L0 = mad(Q(2), k16, (u64)((Q(1) << 8) + Q(0)));
        L0 = mad(Q(3), k24, L0);
        u64 L1 = mad(Q(6), k16, (u64)((Q(5) << 8) + Q(4)));
        L1 = mad(Q(7), k24, L1);
        u64 L2 = mad(Q(10), k16, (u64)((Q(9) << 8) + Q(8)));
        L2 = mad(Q(11), k24, L2);
        const u64 S = L1 + L2;               // < 2^44
        const u64 D = L0 - L2;               // >= 0 by the biases, < 2^46
        const u64 u = mad((u32)(S >> 32), kff, D);
        tl[t] = (u32)u, th[t] = (u32)(u >> 32), h0[t] = (u32)S;
    }
    mx_fold4_tail<false>(tl, th, h0, x);
}

__global__ void __launch_bounds__(256) dft16_check_kernel(const u64* in, u64* out, int waves) {
    __shared__ int lc[12][4][4];
    for (int i = threadIdx.x; i < 12 * 16; i += blockDim.x) (&lc[0][0][0])[i] = (&g_dft16.c[0][0][0])[i];
    __syncthreads();
    Dft16A a;
    load_a(a);
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= waves) return;
    u64 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = in[((size_t)wave * 16 + j) * 16 + 4 * q + i];
    dft16_stage(x, a, lc, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[((size_t)wave * 16 + j) * 16 + 4 * q + i] = x[i];
}

__device__ __forceinline__ u64 seed_val(u64 z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27;
    return z >= gl::P ? z - gl::P : z;
}

template <int WPS, bool PRODUCT>
__global__ void __launch_bounds__(256, WPS) dft16_loop_kernel(u64* out, int iters, u64 seed) {
    __shared__ int lc[12][4][4];
    for (int i = threadIdx.x; i < 12 * 16; i += blockDim.x) (&lc[0][0][0])[i] = (&g_dft16.c[0][0][0])[i];
    __syncthreads();
    Dft16A a;
    load_a(a);
    const int q = (threadIdx.x & 63) >> 4;
    u64 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = seed_val(seed + (u64)(blockIdx.x * 256 + threadIdx.x) * 4 + i);
    const u64 w[4] = {seed_val(seed * 7 + 1), seed_val(seed * 7 + 2), seed_val(seed * 7 + 3), seed_val(seed * 7 + 4)};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        dft16_stage(x, a, lc, q);
        if constexpr (PRODUCT) gl::mont_mul4(x, w, x);
    }
    out[blockIdx.x * 256 + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3];
}

// the yardstick: variant A of tools/microbench_xlane.hip (the radix-32 network of the pass kernels, thread-local, + one product per element)
template <bool PRODUCT>
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
        if constexpr (PRODUCT) {
#pragma unroll
            for (int q = 0; q < 32; q += 4) mul4_inplace(x, q, w, w, w, w);
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) acc ^= x[q];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static u64 host_dft16(const u64* x, int r) {
    u128 acc = 0;
    for (int c = 0; c < 16; ++c) {
        const int s = 12 * ((r * c) & 15);
        u128 pw = 1;
        for (int i = 0; i < s; ++i) pw = pw * 2 % gl::P;
        acc = (acc + (u128)(x[c] % gl::P) * pw) % gl::P;
    }
    return (u64)acc;
}

template <class K>
static double time_kernel(K launch) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    static Dft16Consts t;
    fill_consts(t);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_dft16), &t, sizeof t));

    // ---- parity: 64 waves x 16 columns x 16 elements; random words, words >= p, all-ones bytes, zeros
    const int waves = 64;
    const size_t n = (size_t)waves * 256;
    std::vector<u64> h(n), got(n);
    u64 z = 0x9e3779b97f4a7c15ULL;
    for (size_t i = 0; i < n; ++i) {
        z += 0x9e3779b97f4a7c15ULL;
        u64 v = z;
        v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ULL;
        v = (v ^ (v >> 27)) * 0x94d049bb133111ebULL;
        v ^= v >> 31;
        const size_t col = i / 16;
        if (col % 7 == 3) v = 0xffffffffffffffffULL - (v & 0xffff);       // representatives >= p, bytes 0xff
        else if (col % 7 == 4) v = (i % 16 == 5) ? gl::P - 1 : 0;          // a single non-zero element
        else if (col % 7 == 5) v = 0x8080808080808080ULL ^ (v & 0x0101010101010101ULL);  // bytes around the bias
        h[i] = v;
    }
    u64 *d_in, *d_out;
    CK(hipMalloc(&d_in, n * 8));
    CK(hipMalloc(&d_out, n * 8));
    CK(hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dft16_check_kernel, dim3(waves / 4), dim3(256), 0, 0, d_in, d_out, waves);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), d_out, n * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, noncanon = 0;
    for (size_t col = 0; col < n / 16; ++col)
        for (int r = 0; r < 16; ++r) {
            const u64 want = host_dft16(&h[col * 16], r), g = got[col * 16 + r];
            if (g % gl::P != want) {
                if (bad < 5) printf("MISMATCH col %zu row %d: got %016llx want %016llx\n", col, r, (unsigned long long)g, (unsigned long long)want);
                ++bad;
            }
            if (g >= gl::P) ++noncanon;
        }
    printf("radix-16 stage on v_mfma_i32_16x16x64_i8: %zu of %zu outputs differ from 128-bit arithmetic (%zu non-canonical representatives, allowed)\n", bad, n, noncanon);
    if (bad) return 1;

    // ---- throughput
    u64* d_acc;
    CK(hipMalloc(&d_acc, (size_t)cus * 8 * 256 * 8));
    const int iters = 4000;
    {
        const int grid = cus * 4;
        double ms = time_kernel([&] { hipLaunchKernelGGL(net32_local<true>, dim3(grid), dim3(256), 0, 0, d_acc, iters / 4, 12345ull); });
        const double e = (double)grid * 256 * 32 * (iters / 4) / (ms * 1e-3) / 1e9;
        printf("vector ALU, radix-32 network + product (5 levels), 4 waves/SIMD : %8.3f ms  %8.1f G elements/s  = %8.1f G element-levels/s\n", ms, e, 5 * e);
        ms = time_kernel([&] { hipLaunchKernelGGL(net32_local<false>, dim3(grid), dim3(256), 0, 0, d_acc, iters / 4, 12345ull); });
        const double e2 = (double)grid * 256 * 32 * (iters / 4) / (ms * 1e-3) / 1e9;
        printf("vector ALU, radix-32 network alone                               : %8.3f ms  %8.1f G elements/s  = %8.1f G element-levels/s\n", ms, e2, 5 * e2);
    }
    const auto run16 = [&](int wps, bool product) {
        const int grid = cus * wps;
        double ms;
        if (product) {
            ms = wps == 1   ? time_kernel([&] { hipLaunchKernelGGL((dft16_loop_kernel<1, true>), dim3(grid), dim3(256), 0, 0, d_acc, iters, 12345ull); })
                 : wps == 2 ? time_kernel([&] { hipLaunchKernelGGL((dft16_loop_kernel<2, true>), dim3(grid), dim3(256), 0, 0, d_acc, iters, 12345ull); })
                 : wps == 3 ? time_kernel([&] { hipLaunchKernelGGL((dft16_loop_kernel<3, true>), dim3(grid), dim3(256), 0, 0, d_acc, iters, 12345ull); })
                            : time_kernel([&] { hipLaunchKernelGGL((dft16_loop_kernel<4, true>), dim3(grid), dim3(256), 0, 0, d_acc, iters, 12345ull); });
        } else {
            ms = wps == 1   ? time_kernel([&] { hipLaunchKernelGGL((dft16_loop_kernel<1, false>), dim3(grid), dim3(256), 0, 0, d_acc, iters, 12345ull); })
                 : wps == 2 ? time_kernel([&] { hipLaunchKernelGGL((dft16_loop_kernel<2, false>), dim3(grid), dim3(256), 0, 0, d_acc, iters, 12345ull); })
                 : wps == 3 ? time_kernel([&] { hipLaunchKernelGGL((dft16_loop_kernel<3, false>), dim3(grid), dim3(256), 0, 0, d_acc, iters, 12345ull); })
                            : time_kernel([&] { hipLaunchKernelGGL((dft16_loop_kernel<4, false>), dim3(grid), dim3(256), 0, 0, d_acc, iters, 12345ull); });
        }
        const double e = (double)grid * 256 * 4 * iters / (ms * 1e-3) / 1e9;
        printf("matrix pipe, radix-16 stage %s (4 levels), %d waves/SIMD      : %8.3f ms  %8.1f G elements/s  = %8.1f G element-levels/s\n",
               product ? "+ product" : "alone    ", wps, ms, e, 4 * e);
    };
    for (int wps = 1; wps <= 4; ++wps) run16(wps, true);
    for (int wps = 1; wps <= 4; ++wps) run16(wps, false);
    printf("a 2^20-point transform is 4 x (radix-32 network + product) or 5 x (radix-16 stage + product): compare  4 / (G elements/s)  with  5 / (G elements/s)\n");
    return 0;
}
