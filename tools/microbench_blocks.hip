// microbench_blocks.hip -- throughput of the ACTUAL arithmetic building blocks of the NTT kernels (gl64.h), at 1/2/4/8 waves per
// SIMD, in shader cycles (s_memtime) AND wall time, plus a bit-exactness check of every block against 128-bit integer arithmetic.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I twenty-first_amd/csrc -o tools/microbench_blocks tools/microbench_blocks.hip
// Reconciles profiles/microbench_r01_a.txt (wall time at an assumed clock) with microbench_r01_b.txt (s_memtime cycles): both
// figures come from the same launch here, together with the clock they imply.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "gl64.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
using gl::u64;
using gl::u32;

constexpr int NV = 16;  // independent values per thread (ILP as in a radix-32 level: 16 butterflies)

template <int OP>
__device__ __forceinline__ void step(u64 (&x)[NV]) {
    if constexpr (OP == 0) {  // add_sub: round-1 canonical butterfly, one per block, 3 s_nop
#pragma unroll
        for (int i = 0; i < NV; i += 2) gl::add_sub(x[i], x[i + 1], x[i], x[i + 1]);
    } else if constexpr (OP == 1) {  // add_sub2: canonical, two butterflies per block, no s_nop
#pragma unroll
        for (int i = 0; i < NV; i += 4) gl::add_sub2(x[i], x[i + 1], x[i + 2], x[i + 3], x[i], x[i + 1], x[i + 2], x[i + 3]);
    } else if constexpr (OP == 2) {  // add_sub_lazy2: 8 VALU per butterfly, no s_nop (operands made <= p by the caller's data)
#pragma unroll
        for (int i = 0; i < NV; i += 4) gl::add_sub_lazy2(x[i], x[i + 1], x[i + 2], x[i + 3], x[i], x[i + 1], x[i + 2], x[i + 3]);
    } else if constexpr (OP == 3) {  // mont_mul2
#pragma unroll
        for (int i = 0; i < NV; i += 4) gl::mont_mul2(x[i], x[i + 1], x[i + 2], x[i + 3], x[i], x[i + 2]);
    } else if constexpr (OP == 10) {  // mont_mul4: four products per block, no s_nop
#pragma unroll
        for (int i = 0; i < NV; i += 8) {
            const u64 a4[4] = {x[i], x[i + 2], x[i + 4], x[i + 6]}, b4[4] = {x[i + 1], x[i + 3], x[i + 5], x[i + 7]};
            u64 r4[4];
            gl::mont_mul4(a4, b4, r4);
            x[i] = r4[0], x[i + 2] = r4[1], x[i + 4] = r4[2], x[i + 6] = r4[3];
        }
    } else if constexpr (OP == 4) {  // shl_fold<7> (compiler-scheduled)
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] = gl::shl_fold<7>(x[i]);
    } else if constexpr (OP == 5) {  // shl_monty<14>
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] = gl::shl_monty<14>(x[i]);
    } else if constexpr (OP == 6) {  // compiler add + sub (12 VALU per butterfly)
#pragma unroll
        for (int i = 0; i < NV; i += 2) { u64 s = gl::add(x[i], x[i + 1]); x[i + 1] = gl::sub(x[i], x[i + 1]); x[i] = s; }
    } else if constexpr (OP == 7) {  // mont_mul (compiler, 18)
#pragma unroll
        for (int i = 0; i < NV; i += 2) x[i] = gl::mont_mul(x[i], x[i + 1]);
    } else if constexpr (OP == 8) {  // plain v_add_u32 chain for reference (fast class)
#pragma unroll
        for (int i = 0; i < NV; ++i) { u32 lo = (u32)x[i], hi = (u32)(x[i] >> 32); asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %0" : "+v"(lo), "+v"(hi)); x[i] = ((u64)hi << 32) | lo; }
    } else if constexpr (OP == 9) {  // v_mad_u64_u32 chain
#pragma unroll
        for (int i = 0; i < NV; ++i) { u64 r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"((u32)x[i]), "v"((u32)(x[i] >> 32)), "v"(x[i]) : "vcc"); x[i] = r; }
    }
}

template <int OP>
__global__ void __launch_bounds__(256) bench(u64* out, unsigned long long* cyc, int iters, u64 seed) {
    u64 x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        u64 z = seed + (u64)(blockIdx.x * 256 + threadIdx.x) * NV + i;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z ^= z >> 27;
        x[i] = z >= gl::P ? z - gl::P : z;
    }
    const unsigned long long c0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) step<OP>(x);
    const unsigned long long c1 = __builtin_readcyclecounter();
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) acc ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

// ---- correctness: every block against exact arithmetic, on edge values and random ones
__device__ u64 ref_mod(unsigned __int128 v) { return (u64)(v % gl::P); }
__global__ void check(const u64* a, const u64* b, int n, int* bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 A = a[i], B = b[i];
    const u64 Ac = A >= gl::P ? A - gl::P : A, Bc = B >= gl::P ? B - gl::P : B;  // canonical versions
    const u64 Bp = B > gl::P ? B - gl::P : B;                                     // <= p
    int e = 0;
    u64 s0, d0, s1, d1;
    gl::add_sub2(Ac, Bc, Bc, Ac, s0, d0, s1, d1);
    if (s0 != ref_mod((unsigned __int128)Ac + Bc) || d0 != ref_mod((unsigned __int128)Ac + gl::P - Bc) || s1 != s0 ||
        d1 != ref_mod((unsigned __int128)Bc + gl::P - Ac))
        e |= 1;
    gl::add_sub_lazy2(A, Bp, B, Ac, s0, d0, s1, d1);  // first operand arbitrary, second <= p
    if (s0 % gl::P != ref_mod((unsigned __int128)A + Bp) || d0 % gl::P != ref_mod((unsigned __int128)A + 2 * (unsigned __int128)gl::P - Bp) ||
        s1 % gl::P != ref_mod((unsigned __int128)B + Ac) || d1 % gl::P != ref_mod((unsigned __int128)B + gl::P - Ac))
        e |= 2;
    gl::add_sub(Ac, Bc, s0, d0);
    if (s0 != ref_mod((unsigned __int128)Ac + Bc) || d0 != ref_mod((unsigned __int128)Ac + gl::P - Bc)) e |= 4;
    if (gl::shl_fold<7>(A) != ref_mod((unsigned __int128)A << 7) || gl::shl_fold<31>(A) != ref_mod((unsigned __int128)A << 31)) e |= 8;
    // mont_mul with a LAZY first operand: a * b * 2^-64, b canonical
    u64 m0, m1;
    gl::mont_mul2(A, Bc, B, Ac, m0, m1);
    const unsigned __int128 R = ((unsigned __int128)1 << 64) % gl::P;
    if (ref_mod((unsigned __int128)m0 * R) != ref_mod((unsigned __int128)(A % gl::P) * Bc) || m0 >= gl::P) e |= 16;
    if (ref_mod((unsigned __int128)m1 * R) != ref_mod((unsigned __int128)(B % gl::P) * Ac) || m1 >= gl::P) e |= 16;
    {
        const u64 a4[4] = {A, B, Ac, Bp}, b4[4] = {Bc, Ac, Bc, Ac};
        u64 r4[4];
        gl::mont_mul4(a4, b4, r4);
        for (int i = 0; i < 4; ++i)
            if (ref_mod((unsigned __int128)r4[i] * R) != ref_mod((unsigned __int128)(a4[i] % gl::P) * b4[i]) || r4[i] >= gl::P) e |= 32;
    }
    if (e) atomicOr(bad, e);
}

template <int OP>
void run(const char* name, int valu_per_block, int blocks_per_step, u64* d_out, unsigned long long* d_cyc) {
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    for (int w : {1, 2, 4, 8}) {
        const int grid = cus * w;  // blocks of 256 threads = one wave per SIMD each
        const int iters = 4000;
        hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, 200, 1);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, iters, 2);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc(grid * 4);
        CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        double avg = 0;
        for (auto c : cyc) avg += (double)c;
        avg /= cyc.size();
        const double blocks = (double)iters * blocks_per_step;              // per thread (= per wave)
        const double wave_instr = blocks * valu_per_block;                  // VALU instructions per wave
        const double cyc_per_instr_wave = avg / wave_instr;                 // s_memtime cycles per instruction of ONE wave
        const double cyc_per_instr_simd = cyc_per_instr_wave / w;           // ... per instruction issued by the SIMD
        const double gwips = wave_instr * grid * 4 / (ms * 1e-3) / 1e9;     // wall-clock wave-instructions/s, whole chip
        const double clk = avg / (ms * 1e-3) / 1e9;                         // implied clock of the s_memtime counter (GHz)
        printf("%-16s w/SIMD=%d  %7.3f ms  cyc/VALU(wave) %6.2f  cyc/VALU(SIMD) %5.2f  chip %7.1f G wave-instr/s  (%5.3f per SIMD)  counter %.2f GHz\n",
               name, w, ms, cyc_per_instr_wave, cyc_per_instr_simd, gwips, gwips / (cus * 4), clk);
    }
}

int main() {
    // correctness first
    const int n = 1 << 16;
    std::vector<u64> a(n), b(n);
    const u64 edge[] = {0, 1, 2, 0xfffffffeULL, 0xffffffffULL, 0x100000000ULL, 0x100000001ULL, gl::P - 2, gl::P - 1, gl::P, gl::P + 1,
                        0xfffffffffffffffeULL, 0xffffffffffffffffULL, 0x8000000000000000ULL, 0x7fffffffffffffffULL, 0xffffffff00000000ULL,
                        0xfffffffeffffffffULL, 0x1ffffffffULL, 0xfffffffe00000001ULL, 0xfffffffe00000002ULL};
    const int ne = sizeof(edge) / sizeof(edge[0]);
    u64 st = 12345;
    auto rnd = [&]() { st += 0x9e3779b97f4a7c15ULL; u64 z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); };
    for (int i = 0; i < n; ++i) {
        if (i < ne * ne) { a[i] = edge[i / ne]; b[i] = edge[i % ne]; }
        else { a[i] = (rnd() & 3) ? rnd() : edge[rnd() % ne]; b[i] = (rnd() & 3) ? rnd() : edge[rnd() % ne]; }
    }
    u64 *da, *db, *d_out;
    unsigned long long* d_cyc;
    int* d_bad;
    CK(hipMalloc(&da, n * 8));
    CK(hipMalloc(&db, n * 8));
    CK(hipMalloc(&d_bad, 4));
    CK(hipMemset(d_bad, 0, 4));
    CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, da, db, n, d_bad);
    int bad = 0;
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("block check on %d operand pairs (edge x edge + random): %s (mask %d: 1 add_sub2, 2 add_sub_lazy2, 4 add_sub, 8 shl_fold, 16 mont_mul2 lazy operand, 32 mont_mul4)\n",
           n, bad ? "MISMATCH" : "all bit-exact", bad);
    CK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * 8 * 2));
    CK(hipMalloc(&d_cyc, (size_t)256 * 8 * 4 * 8 * 2));
    run<0>("add_sub", 10, NV / 2, d_out, d_cyc);
    run<1>("add_sub2", 10, NV / 2, d_out, d_cyc);
    run<2>("add_sub_lazy2", 8, NV / 2, d_out, d_cyc);
    run<6>("add+sub (cc)", 12, NV / 2, d_out, d_cyc);
    run<3>("mont_mul2", 15, NV / 2, d_out, d_cyc);
    run<10>("mont_mul4", 15, NV / 2, d_out, d_cyc);
    run<7>("mont_mul (cc)", 18, NV / 2, d_out, d_cyc);
    run<4>("shl_fold<7>", 8, NV, d_out, d_cyc);
    run<5>("shl_monty<14>", 10, NV, d_out, d_cyc);
    run<8>("v_add_u32", 2, NV, d_out, d_cyc);
    run<9>("v_mad_u64_u32", 1, NV, d_out, d_cyc);
    return bad ? 1 : 0;
}
