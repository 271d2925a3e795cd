// microbench.hip -- gfx950 instruction-rate and memory-hierarchy probes that the kernel
// design of this repo is priced against (DESIGN.md "measured constants").
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o tools/microbench
//   run  : tools/microbench            (prints one line per probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef uint64_t u64;
typedef uint32_t u32;

// ---- VALU issue-rate probes: N_ITER iterations of 8 independent chains of one instruction.
#define ITERS 4096
#define REP8(S) S S S S S S S S

#define DEF_PROBE(NAME, ASM8)                                                            \
  __global__ void __launch_bounds__(256) NAME(u32* out) {                               \
    u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    u32 b0 = a0 * 3 + 1, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7; \
    u64 c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a4, c5 = a5, c6 = a6, c7 = a7;          \
    for (int i = 0; i < ITERS; ++i) {                                                    \
      asm volatile(ASM8                                                                  \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
          "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)  \
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7) : "vcc"); \
    }                                                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (u32)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7); \
  }

// operand numbering: %0-%7 = a (u32), %8-%15 = c (u64), %16-%23 = b (u32)
DEF_PROBE(p_add_u32,
  "v_add_u32 %0, %0, %16\n v_add_u32 %1, %1, %17\n v_add_u32 %2, %2, %18\n v_add_u32 %3, %3, %19\n"
  "v_add_u32 %4, %4, %20\n v_add_u32 %5, %5, %21\n v_add_u32 %6, %6, %22\n v_add_u32 %7, %7, %23\n")
DEF_PROBE(p_mul_lo_u32,
  "v_mul_lo_u32 %0, %0, %16\n v_mul_lo_u32 %1, %1, %17\n v_mul_lo_u32 %2, %2, %18\n v_mul_lo_u32 %3, %3, %19\n"
  "v_mul_lo_u32 %4, %4, %20\n v_mul_lo_u32 %5, %5, %21\n v_mul_lo_u32 %6, %6, %22\n v_mul_lo_u32 %7, %7, %23\n")
DEF_PROBE(p_mul_hi_u32,
  "v_mul_hi_u32 %0, %0, %16\n v_mul_hi_u32 %1, %1, %17\n v_mul_hi_u32 %2, %2, %18\n v_mul_hi_u32 %3, %3, %19\n"
  "v_mul_hi_u32 %4, %4, %20\n v_mul_hi_u32 %5, %5, %21\n v_mul_hi_u32 %6, %6, %22\n v_mul_hi_u32 %7, %7, %23\n")
DEF_PROBE(p_mad_u64_u32,
  "v_mad_u64_u32 %8, vcc, %0, %16, %8\n v_mad_u64_u32 %9, vcc, %1, %17, %9\n v_mad_u64_u32 %10, vcc, %2, %18, %10\n v_mad_u64_u32 %11, vcc, %3, %19, %11\n"
  "v_mad_u64_u32 %12, vcc, %4, %20, %12\n v_mad_u64_u32 %13, vcc, %5, %21, %13\n v_mad_u64_u32 %14, vcc, %6, %22, %14\n v_mad_u64_u32 %15, vcc, %7, %23, %15\n")
DEF_PROBE(p_mad_u32_u24,
  "v_mad_u32_u24 %0, %0, %16, %0\n v_mad_u32_u24 %1, %1, %17, %1\n v_mad_u32_u24 %2, %2, %18, %2\n v_mad_u32_u24 %3, %3, %19, %3\n"
  "v_mad_u32_u24 %4, %4, %20, %4\n v_mad_u32_u24 %5, %5, %21, %5\n v_mad_u32_u24 %6, %6, %22, %6\n v_mad_u32_u24 %7, %7, %23, %7\n")
DEF_PROBE(p_lshl_add_u64,
  "v_lshl_add_u64 %8, %8, 0, %9\n v_lshl_add_u64 %9, %9, 0, %10\n v_lshl_add_u64 %10, %10, 0, %11\n v_lshl_add_u64 %11, %11, 0, %12\n"
  "v_lshl_add_u64 %12, %12, 0, %13\n v_lshl_add_u64 %13, %13, 0, %14\n v_lshl_add_u64 %14, %14, 0, %15\n v_lshl_add_u64 %15, %15, 0, %8\n")
DEF_PROBE(p_add_co_addc,
  "v_add_co_u32 %0, vcc, %0, %16\n v_addc_co_u32 %1, vcc, %1, %17, vcc\n v_add_co_u32 %2, vcc, %2, %18\n v_addc_co_u32 %3, vcc, %3, %19, vcc\n"
  "v_add_co_u32 %4, vcc, %4, %20\n v_addc_co_u32 %5, vcc, %5, %21, vcc\n v_add_co_u32 %6, vcc, %6, %22\n v_addc_co_u32 %7, vcc, %7, %23, vcc\n")
DEF_PROBE(p_cmp_u64_cndmask,
  "v_cmp_lt_u64 vcc, %8, %9\n v_cndmask_b32 %0, %0, %16, vcc\n v_cmp_lt_u64 vcc, %10, %11\n v_cndmask_b32 %1, %1, %17, vcc\n"
  "v_cmp_lt_u64 vcc, %12, %13\n v_cndmask_b32 %2, %2, %18, vcc\n v_cmp_lt_u64 vcc, %14, %15\n v_cndmask_b32 %3, %3, %19, vcc\n")
DEF_PROBE(p_lshlrev_b64,
  "v_lshlrev_b64 %8, 3, %8\n v_lshlrev_b64 %9, 3, %9\n v_lshlrev_b64 %10, 3, %10\n v_lshlrev_b64 %11, 3, %11\n"
  "v_lshlrev_b64 %12, 3, %12\n v_lshlrev_b64 %13, 3, %13\n v_lshlrev_b64 %14, 3, %14\n v_lshlrev_b64 %15, 3, %15\n")
DEF_PROBE(p_alignbit,
  "v_alignbit_b32 %0, %0, %16, 7\n v_alignbit_b32 %1, %1, %17, 7\n v_alignbit_b32 %2, %2, %18, 7\n v_alignbit_b32 %3, %3, %19, 7\n"
  "v_alignbit_b32 %4, %4, %20, 7\n v_alignbit_b32 %5, %5, %21, 7\n v_alignbit_b32 %6, %6, %22, 7\n v_alignbit_b32 %7, %7, %23, 7\n")
DEF_PROBE(p_add3_u32,
  "v_add3_u32 %0, %0, %16, %17\n v_add3_u32 %1, %1, %17, %18\n v_add3_u32 %2, %2, %18, %19\n v_add3_u32 %3, %3, %19, %20\n"
  "v_add3_u32 %4, %4, %20, %21\n v_add3_u32 %5, %5, %21, %22\n v_add3_u32 %6, %6, %22, %23\n v_add3_u32 %7, %7, %23, %16\n")
DEF_PROBE(p_dot4_u32_u8,
  "v_dot4_u32_u8 %0, %0, %16, %0\n v_dot4_u32_u8 %1, %1, %17, %1\n v_dot4_u32_u8 %2, %2, %18, %2\n v_dot4_u32_u8 %3, %3, %19, %3\n"
  "v_dot4_u32_u8 %4, %4, %20, %4\n v_dot4_u32_u8 %5, %5, %21, %5\n v_dot4_u32_u8 %6, %6, %22, %6\n v_dot4_u32_u8 %7, %7, %23, %7\n")

typedef void (*probe_fn)(u32*);

static void run_probe(const char* name, probe_fn fn, int instr_per_iter, u32* d_out) {
  int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, d_out);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  double wave_instr = (double)blocks * (threads / 64) * ITERS * instr_per_iter;
  // cycles per wave-instruction per SIMD at 2.4 GHz nominal: 1024 SIMDs
  double cyc = best * 1e-3 * 2.4e9 * 1024.0 / wave_instr;
  printf("valu %-18s %8.3f ms  %7.2f G wave-instr/s  ~%5.2f cyc/wave-instr/SIMD @2.4GHz  (%.2f T lane-ops/s)\n", name, best,
         wave_instr / best * 1e-6, cyc, wave_instr * 64 / best * 1e-9);
}

// ---- memory probes
__global__ void __launch_bounds__(256) copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) copy8(const uint2* __restrict__ src, uint2* __restrict__ dst, size_t n8) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n8; i += stride) dst[i] = src[i];
}
// column-tile copy: each block copies a tile of 1024 rows x C u64 columns of a [1024][1024] matrix (8 MiB per matrix),
// mimicking the NTT pass-1 access pattern (C*8-byte segments at 8 KiB stride).
template <int C>
__global__ void __launch_bounds__(256) coltile_copy(const u64* __restrict__ src, u64* __restrict__ dst, int tiles_per_mat) {
  int tile = blockIdx.x % tiles_per_mat; size_t mat = blockIdx.x / tiles_per_mat;
  const u64* s = src + mat * (1u << 20) + (size_t)tile * C;
  u64* d = dst + mat * (1u << 20) + (size_t)tile * C;
  int c = threadIdx.x % C, g = threadIdx.x / C;  // g in [0, 256/C)
  constexpr int RPT = 1024 * C / 256;            // rows per thread
  u64 v[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) v[i] = s[(size_t)(g + (256 / C) * i) * 1024 + c];
#pragma unroll
  for (int i = 0; i < RPT; ++i) d[(size_t)(g + (256 / C) * i) * 1024 + c] = v[i];
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s  CUs %d  clock %d kHz  L2 %d  \n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.l2CacheSize);
  u32* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * 4 * 2));
  run_probe("v_add_u32", p_add_u32, 8, d_out);
  run_probe("v_add3_u32", p_add3_u32, 8, d_out);
  run_probe("v_mul_lo_u32", p_mul_lo_u32, 8, d_out);
  run_probe("v_mul_hi_u32", p_mul_hi_u32, 8, d_out);
  run_probe("v_mad_u64_u32", p_mad_u64_u32, 8, d_out);
  run_probe("v_mad_u32_u24", p_mad_u32_u24, 8, d_out);
  run_probe("v_lshl_add_u64", p_lshl_add_u64, 8, d_out);
  run_probe("add_co+addc", p_add_co_addc, 8, d_out);
  run_probe("cmp_lt_u64+cndmask", p_cmp_u64_cndmask, 8, d_out);
  run_probe("v_lshlrev_b64", p_lshlrev_b64, 8, d_out);
  run_probe("v_alignbit_b32", p_alignbit, 8, d_out);
  run_probe("v_dot4_u32_u8", p_dot4_u32_u8, 8, d_out);

  // ---- memory: 2 GiB working set like config C2
  size_t total = (size_t)2 << 30;
  char *a, *b, *scratch;
  CK(hipMalloc(&a, total)); CK(hipMalloc(&b, total)); CK(hipMalloc(&scratch, total));
  CK(hipMemset(a, 1, total)); CK(hipMemset(b, 2, total)); CK(hipMemset(scratch, 3, total));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int grid = 256 * 8;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, total / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = time_ms(e0, e1);
    printf("mem copy16 2GiB->2GiB          %7.3f ms  %6.2f TB/s (r+w)\n", ms, 2.0 * total / ms * 1e-9);
  }
  {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(copy8, dim3(grid), dim3(256), 0, 0, (const uint2*)a, (uint2*)b, total / 8);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = time_ms(e0, e1);
    printf("mem copy8  2GiB->2GiB          %7.3f ms  %6.2f TB/s (r+w)\n", ms, 2.0 * total / ms * 1e-9);
  }
  // two-pass through a scratch chunk: per chunk  a->scratch, scratch->b.  If the chunk stays in the
  // Infinity Cache the HBM traffic is ~half of the 4x total bytes moved.
  size_t chunks[] = {(size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20, (size_t)512 << 20, (size_t)2 << 30};
  for (size_t chunk : chunks) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (size_t off = 0; off < total; off += chunk) {
        hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)(a + off), (uint4*)scratch, chunk / 16);
        hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)scratch, (uint4*)(b + off), chunk / 16);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1);
      if (rep == 1)
        printf("mem 2-pass via %4zu MiB scratch  %7.3f ms  algorithmic(2x total) %6.2f TB/s  moved(4x) %6.2f TB/s  launches %zu\n",
               chunk >> 20, ms, 2.0 * total / ms * 1e-9, 4.0 * total / ms * 1e-9, 2 * (total / chunk));
    }
  }
  // same with in-place chunk (a->a would alias; use a->b chunk then b->a chunk: write then re-read same addresses)
  for (size_t chunk : chunks) {
    CK(hipEventRecord(e0));
    for (size_t off = 0; off < total; off += chunk) {
      hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)(a + off), (uint4*)(b + off), chunk / 16);
      hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)(b + off), (uint4*)(a + off), chunk / 16);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = time_ms(e0, e1);
    printf("mem 2-pass a->b->a chunk %4zu MiB %7.3f ms  algorithmic %6.2f TB/s\n", chunk >> 20, ms, 2.0 * total / ms * 1e-9);
  }
  // column-tile copies (NTT pass-1 pattern) over 256 matrices of 1024x1024 u64
  {
    int mats = 256;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(coltile_copy<8>, dim3(mats * 128), dim3(256), 0, 0, (const u64*)a, (u64*)b, 128);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1);
      if (rep) printf("mem coltile C=8  (64B segs)     %7.3f ms  %6.2f TB/s (r+w)\n", ms, 2.0 * total / ms * 1e-9);
    }
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(coltile_copy<16>, dim3(mats * 64), dim3(256), 0, 0, (const u64*)a, (u64*)b, 64);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1);
      if (rep) printf("mem coltile C=16 (128B segs)    %7.3f ms  %6.2f TB/s (r+w)\n", ms, 2.0 * total / ms * 1e-9);
    }
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(coltile_copy<4>, dim3(mats * 256), dim3(256), 0, 0, (const u64*)a, (u64*)b, 256);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1);
      if (rep) printf("mem coltile C=4  (32B segs)     %7.3f ms  %6.2f TB/s (r+w)\n", ms, 2.0 * total / ms * 1e-9);
    }
  }
  return 0;
}
