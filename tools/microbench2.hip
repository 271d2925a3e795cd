// microbench2.hip -- per-instruction issue cost in REAL shader cycles (s_memtime), by encoding class.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef uint64_t u64; typedef uint32_t u32;
#define ITERS 2048

#define DEF_PROBE(NAME, ASM8)                                                            \
  __global__ void __launch_bounds__(256) NAME(u32* out, u64* cyc) {                     \
    u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    u32 b0 = a0 * 3 + 1, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7; \
    u64 c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a4, c5 = a5, c6 = a6, c7 = a7;          \
    u64 t0 = __builtin_readcyclecounter();                                               \
    for (int i = 0; i < ITERS; ++i) {                                                    \
      asm volatile(ASM8                                                                  \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
          "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)  \
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7) : "vcc", "s4", "s5", "s6", "s7"); \
    }                                                                                    \
    u64 t1 = __builtin_readcyclecounter();                                               \
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (u32)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7); \
  }
#define R8(OP) OP " %0, %0, %16\n" OP " %1, %1, %17\n" OP " %2, %2, %18\n" OP " %3, %3, %19\n" OP " %4, %4, %20\n" OP " %5, %5, %21\n" OP " %6, %6, %22\n" OP " %7, %7, %23\n"
#define R8_3(OP) OP " %0, %0, %16, %17\n" OP " %1, %1, %17, %18\n" OP " %2, %2, %18, %19\n" OP " %3, %3, %19, %20\n" OP " %4, %4, %20, %21\n" OP " %5, %5, %21, %22\n" OP " %6, %6, %22, %23\n" OP " %7, %7, %23, %16\n"

DEF_PROBE(p_add_u32_e32, R8("v_add_u32_e32"))
DEF_PROBE(p_add_u32_e64, R8("v_add_u32_e64"))
DEF_PROBE(p_sub_u32_e32, R8("v_sub_u32_e32"))
DEF_PROBE(p_xor_e32, R8("v_xor_b32_e32"))
DEF_PROBE(p_and_e32, R8("v_and_b32_e32"))
DEF_PROBE(p_lshlrev_b32_e32, R8("v_lshlrev_b32_e32"))
DEF_PROBE(p_lshrrev_b32_e32, R8("v_lshrrev_b32_e32"))
DEF_PROBE(p_mul_u32_u24_e32, R8("v_mul_u32_u24_e32"))
DEF_PROBE(p_max_u32_e32, R8("v_max_u32_e32"))
DEF_PROBE(p_add_f32_e32, R8("v_add_f32_e32"))
DEF_PROBE(p_fma_f32, R8_3("v_fma_f32"))
DEF_PROBE(p_add3_u32, R8_3("v_add3_u32"))
DEF_PROBE(p_xad_u32, R8_3("v_xad_u32"))
DEF_PROBE(p_lshl_add_u32, R8_3("v_lshl_add_u32"))
DEF_PROBE(p_and_or_b32, R8_3("v_and_or_b32"))
DEF_PROBE(p_bfe_u32, R8_3("v_bfe_u32"))
DEF_PROBE(p_perm_b32, R8_3("v_perm_b32"))
DEF_PROBE(p_alignbit, R8_3("v_alignbit_b32"))
DEF_PROBE(p_mad_u32_u24, R8_3("v_mad_u32_u24"))
DEF_PROBE(p_mul_lo_u32, R8("v_mul_lo_u32"))
DEF_PROBE(p_mul_hi_u32, R8("v_mul_hi_u32"))
DEF_PROBE(p_pk_add_u16, R8("v_pk_add_u16"))
DEF_PROBE(p_pk_mul_lo_u16, R8("v_pk_mul_lo_u16"))
DEF_PROBE(p_cndmask_e32, "v_cndmask_b32_e32 %0, %0, %16, vcc\n v_cndmask_b32_e32 %1, %1, %17, vcc\n v_cndmask_b32_e32 %2, %2, %18, vcc\n v_cndmask_b32_e32 %3, %3, %19, vcc\n v_cndmask_b32_e32 %4, %4, %20, vcc\n v_cndmask_b32_e32 %5, %5, %21, vcc\n v_cndmask_b32_e32 %6, %6, %22, vcc\n v_cndmask_b32_e32 %7, %7, %23, vcc\n")
DEF_PROBE(p_cmp_u32_e32, "v_cmp_lt_u32_e32 vcc, %0, %16\n v_cmp_lt_u32_e32 vcc, %1, %17\n v_cmp_lt_u32_e32 vcc, %2, %18\n v_cmp_lt_u32_e32 vcc, %3, %19\n v_cmp_lt_u32_e32 vcc, %4, %20\n v_cmp_lt_u32_e32 vcc, %5, %21\n v_cmp_lt_u32_e32 vcc, %6, %22\n v_cmp_lt_u32_e32 vcc, %7, %23\n")
DEF_PROBE(p_cmp_u64_e32, "v_cmp_lt_u64_e32 vcc, %8, %9\n v_cmp_lt_u64_e32 vcc, %9, %10\n v_cmp_lt_u64_e32 vcc, %10, %11\n v_cmp_lt_u64_e32 vcc, %11, %12\n v_cmp_lt_u64_e32 vcc, %12, %13\n v_cmp_lt_u64_e32 vcc, %13, %14\n v_cmp_lt_u64_e32 vcc, %14, %15\n v_cmp_lt_u64_e32 vcc, %15, %8\n")
DEF_PROBE(p_add_co_e32, "v_add_co_u32_e32 %0, vcc, %0, %16\n v_add_co_u32_e32 %1, vcc, %1, %17\n v_add_co_u32_e32 %2, vcc, %2, %18\n v_add_co_u32_e32 %3, vcc, %3, %19\n v_add_co_u32_e32 %4, vcc, %4, %20\n v_add_co_u32_e32 %5, vcc, %5, %21\n v_add_co_u32_e32 %6, vcc, %6, %22\n v_add_co_u32_e32 %7, vcc, %7, %23\n")
DEF_PROBE(p_addc_co_e32, "v_addc_co_u32_e32 %0, vcc, %0, %16, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %17, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %18, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %19, vcc\n v_addc_co_u32_e32 %4, vcc, %4, %20, vcc\n v_addc_co_u32_e32 %5, vcc, %5, %21, vcc\n v_addc_co_u32_e32 %6, vcc, %6, %22, vcc\n v_addc_co_u32_e32 %7, vcc, %7, %23, vcc\n")
DEF_PROBE(p_add_co_e64_sgpr, "v_add_co_u32_e64 %0, s[4:5], %0, %16\n v_add_co_u32_e64 %1, s[6:7], %1, %17\n v_add_co_u32_e64 %2, s[4:5], %2, %18\n v_add_co_u32_e64 %3, s[6:7], %3, %19\n v_add_co_u32_e64 %4, s[4:5], %4, %20\n v_add_co_u32_e64 %5, s[6:7], %5, %21\n v_add_co_u32_e64 %6, s[4:5], %6, %22\n v_add_co_u32_e64 %7, s[6:7], %7, %23\n")
DEF_PROBE(p_mad_u64_u32, "v_mad_u64_u32 %8, vcc, %0, %16, %8\n v_mad_u64_u32 %9, vcc, %1, %17, %9\n v_mad_u64_u32 %10, vcc, %2, %18, %10\n v_mad_u64_u32 %11, vcc, %3, %19, %11\n v_mad_u64_u32 %12, vcc, %4, %20, %12\n v_mad_u64_u32 %13, vcc, %5, %21, %13\n v_mad_u64_u32 %14, vcc, %6, %22, %14\n v_mad_u64_u32 %15, vcc, %7, %23, %15\n")
DEF_PROBE(p_lshl_add_u64, "v_lshl_add_u64 %8, %8, 0, %9\n v_lshl_add_u64 %9, %9, 0, %10\n v_lshl_add_u64 %10, %10, 0, %11\n v_lshl_add_u64 %11, %11, 0, %12\n v_lshl_add_u64 %12, %12, 0, %13\n v_lshl_add_u64 %13, %13, 0, %14\n v_lshl_add_u64 %14, %14, 0, %15\n v_lshl_add_u64 %15, %15, 0, %8\n")
DEF_PROBE(p_mov_b32, "v_mov_b32_e32 %0, %16\n v_mov_b32_e32 %1, %17\n v_mov_b32_e32 %2, %18\n v_mov_b32_e32 %3, %19\n v_mov_b32_e32 %4, %20\n v_mov_b32_e32 %5, %21\n v_mov_b32_e32 %6, %22\n v_mov_b32_e32 %7, %23\n")
DEF_PROBE(p_pk_mov, "v_pk_mov_b32 %8, %9, %10\n v_pk_mov_b32 %9, %10, %11\n v_pk_mov_b32 %10, %11, %12\n v_pk_mov_b32 %11, %12, %13\n v_pk_mov_b32 %12, %13, %14\n v_pk_mov_b32 %13, %14, %15\n v_pk_mov_b32 %14, %15, %8\n v_pk_mov_b32 %15, %8, %9\n")

typedef void (*probe_fn)(u32*, u64*);
static void run_probe(const char* name, probe_fn fn, u32* d_out, u64* d_cyc, int waves_per_simd) {
  int threads = 256, blocks = 256 * waves_per_simd;   // 4 waves per block -> waves_per_simd per SIMD
  int nw = blocks * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  static u64 h[65536]; CK(hipMemcpy(h, d_cyc, sizeof(u64) * nw, hipMemcpyDeviceToHost));
  double avg = 0; for (int i = 0; i < nw; ++i) avg += (double)h[i]; avg /= nw;
  double per_wave_instr = avg / (ITERS * 8.0);               // cycles a wave takes per own instruction
  double per_simd = per_wave_instr / waves_per_simd;         // SIMD issue cost per wave-instruction
  double ghz = (double)ITERS * 8 * nw / 1024.0 * per_simd / (best * 1e-3) * 1e-9;  // implied clock if counter == shader clock
  printf("%-20s w/SIMD=%d  %7.3f ms  wave cyc/instr %6.2f  SIMD cyc/instr %5.2f  implied clk %.2f GHz\n", name, waves_per_simd, best, per_wave_instr, per_simd, ghz);
}
int main() {
  u32* d_out; u64* d_cyc; CK(hipMalloc(&d_out, 256 * 8 * 256 * 4)); CK(hipMalloc(&d_cyc, 65536 * 8));
#define RUN(N) run_probe(#N, N, d_out, d_cyc, 1); run_probe(#N, N, d_out, d_cyc, 4); run_probe(#N, N, d_out, d_cyc, 8);
  RUN(p_add_u32_e32) RUN(p_add_u32_e64) RUN(p_sub_u32_e32) RUN(p_xor_e32) RUN(p_and_e32) RUN(p_lshlrev_b32_e32) RUN(p_lshrrev_b32_e32)
  RUN(p_mul_u32_u24_e32) RUN(p_max_u32_e32) RUN(p_add_f32_e32) RUN(p_fma_f32) RUN(p_add3_u32) RUN(p_xad_u32) RUN(p_lshl_add_u32) RUN(p_and_or_b32)
  RUN(p_bfe_u32) RUN(p_perm_b32) RUN(p_alignbit) RUN(p_mad_u32_u24) RUN(p_mul_lo_u32) RUN(p_mul_hi_u32) RUN(p_pk_add_u16) RUN(p_pk_mul_lo_u16)
  RUN(p_cndmask_e32) RUN(p_cmp_u32_e32) RUN(p_cmp_u64_e32) RUN(p_add_co_e32) RUN(p_addc_co_e32) RUN(p_add_co_e64_sgpr) RUN(p_mad_u64_u32) RUN(p_lshl_add_u64)
  RUN(p_mov_b32) RUN(p_pk_mov)
  return 0;
}
