// instr_rates.hip -- issue throughput of single gfx950 VALU / DS instructions (independent chains, 4 and 8 waves per SIMD),
// relative to v_add_u32.  The table the arithmetic formulation of the NTT kernels is chosen from.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/instr_rates tools/instr_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef uint32_t u32;
typedef uint64_t u64;
constexpr int NV = 12;

// A: one 32-bit register per chain ("+v"(a)), B: a second read-only 32-bit source.  W: one 64-bit register per chain.
#define DEF32(NAME, ASM)                                                                                   \
    struct NAME { static __device__ __forceinline__ void op(u32& a, u32 b, u64& w, u32 c) { asm volatile(ASM : "+v"(a) : "v"(b), "v"(c) : "vcc"); } \
                  static constexpr const char* name = #NAME; };
#define DEF64(NAME, ASM)                                                                                   \
    struct NAME { static __device__ __forceinline__ void op(u32& a, u32 b, u64& w, u32 c) { asm volatile(ASM : "+v"(w) : "v"(b), "v"(c), "v"(a) : "vcc"); } \
                  static constexpr const char* name = #NAME; };
#define DEFS(NAME, ASM)  /* writes an SGPR pair */                                                         \
    struct NAME { static __device__ __forceinline__ void op(u32& a, u32 b, u64& w, u32 c) { u64 s; asm volatile(ASM : "+v"(a), "=s"(s) : "v"(b), "v"(c) : "vcc"); } \
                  static constexpr const char* name = #NAME; };

DEF32(v_add_u32, "v_add_u32 %0, %0, %1")
DEF32(v_sub_u32, "v_sub_u32 %0, %0, %1")
DEF32(v_add_co_u32_vcc, "v_add_co_u32_e32 %0, vcc, %0, %1")
DEFS(v_add_co_u32_sgpr, "v_add_co_u32_e64 %0, %1, %0, %2")
DEF32(v_addc_co_u32_vcc, "v_addc_co_u32_e32 %0, vcc, %0, %1, vcc")
DEFS(v_addc_co_u32_sgpr, "v_addc_co_u32_e64 %0, %1, %0, %2, vcc")
DEF32(v_add3_u32, "v_add3_u32 %0, %0, %1, %2")
DEF32(v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
DEF32(v_add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 3")
DEF32(v_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %1")
DEF32(v_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
DEF32(v_or3_b32, "v_or3_b32 %0, %0, %1, %2")
DEF32(v_xad_u32, "v_xad_u32 %0, %0, %1, %2")
DEF32(v_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
DEF32(v_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
DEF32(v_ashrrev_i32, "v_ashrrev_i32 %0, 3, %0")
DEF32(v_alignbit_b32, "v_alignbit_b32 %0, %0, %1, 7")
DEF32(v_bfe_u32, "v_bfe_u32 %0, %0, 3, 17")
DEF32(v_bfe_i32, "v_bfe_i32 %0, %0, 3, 17")
DEF32(v_bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
DEF32(v_perm_b32, "v_perm_b32 %0, %0, %1, %2")
DEF32(v_and_b32, "v_and_b32 %0, %0, %1")
DEF32(v_xor_b32, "v_xor_b32 %0, %0, %1")
DEF32(v_mov_b32, "v_mov_b32 %0, %1")
DEF32(v_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
DEF32(v_min_u32, "v_min_u32 %0, %0, %1")
DEF32(v_med3_u32, "v_med3_u32 %0, %0, %1, %2")
DEF32(v_sad_u32, "v_sad_u32 %0, %0, %1, %2")
DEF32(v_cmp_lt_u32_vcc, "v_cmp_lt_u32 vcc, %0, %1")
DEF32(v_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF32(v_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEF32(v_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
DEF32(v_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
DEF32(v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF32(v_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEF32(v_mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
DEF32(v_mad_u16, "v_mad_u16 %0, %0, %1, %2")
DEF32(v_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %2")
DEF32(v_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEF32(v_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
DEF32(v_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
DEF32(v_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
DEF32(v_dot2_u32_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
DEF32(v_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF32(v_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
DEF32(v_mov_dpp_ror, "v_mov_b32_dpp %0, %1 row_ror:3 row_mask:0xf bank_mask:0xf")
DEF32(v_add_dpp_ror, "v_add_u32_dpp %0, %1, %0 row_ror:3 row_mask:0xf bank_mask:0xf")
DEF32(v_mov_dpp_bcast, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF32(v_permlane32_swap, "v_permlane32_swap_b32 %0, %1")
DEF32(v_permlane16_swap, "v_permlane16_swap_b32 %0, %1")
DEF64(v_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
struct v_mad_u64_u32_sgprc { static __device__ __forceinline__ void op(u32& a, u32 b, u64& w, u32 c) { u64 s; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(w), "=s"(s) : "v"(b), "v"(c)); }
                  static constexpr const char* name = "v_mad_u64_u32_sgprc"; };
DEF64(v_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0")
DEF64(v_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
DEF64(v_lshrrev_b64, "v_lshrrev_b64 %0, 3, %0")
DEF64(v_mov_b64, "v_mov_b64 %0, %0")
DEF64(v_pk_mov_b32, "v_pk_mov_b32 %0, %0, %0")
DEF64(v_add_f64, "v_add_f64 %0, %0, %0")
DEF64(v_fma_f64, "v_fma_f64 %0, %0, %0, %0")
DEF64(v_pk_add_f32, "v_pk_add_f32 %0, %0, %0")
DEF64(v_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0")
DEF64(v_cmp_lt_u64_vcc, "v_cmp_lt_u64 vcc, %0, %0")

template <class I>
__global__ void __launch_bounds__(256) bench(u32* out, int iters, u32 seed) {
    u32 a[NV];
    u64 w[NV];
    const u32 b = seed * 2654435761u + threadIdx.x, c = seed ^ (threadIdx.x * 40503u);
#pragma unroll
    for (int i = 0; i < NV; ++i) { a[i] = b * (i + 3); w[i] = ((u64)a[i] << 20) ^ c; }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NV; ++i) I::op(a[i], b, w[i], c);
    }
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) acc ^= a[i] ^ (u32)w[i] ^ (u32)(w[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static double g_base[2] = {0, 0};
template <class I>
void run(u32* d_out, int cus) {
    printf("%-22s", I::name);
    int k = 0;
    for (int w : {4, 8}) {
        const int grid = cus * w, iters = 3000;
        hipLaunchKernelGGL(bench<I>, dim3(grid), dim3(256), 0, 0, d_out, 100, 1u);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<I>, dim3(grid), dim3(256), 0, 0, d_out, iters, 2u);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_instr = (double)iters * 4 * NV;
        const double rate = wave_instr * grid * 4 / (ms * 1e-3) / 1e9 / (cus * 4);  // G wave-instr/s per SIMD
        if (g_base[k] == 0) g_base[k] = rate;
        printf("  w/SIMD=%d %6.3f G/s/SIMD (x%.2f of v_add_u32)", w, rate, rate / g_base[k]);
        ++k;
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    u32* d_out;
    CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
    printf("per-instruction issue rate, %d CUs, %d independent chains per wave, wave-instructions per second per SIMD\n", cus, NV);
#define R(N) run<N>(d_out, cus);
    R(v_add_u32) R(v_sub_u32) R(v_add_co_u32_vcc) R(v_add_co_u32_sgpr) R(v_addc_co_u32_vcc) R(v_addc_co_u32_sgpr) R(v_add3_u32)
    R(v_lshl_add_u32) R(v_add_lshl_u32) R(v_lshl_or_b32) R(v_and_or_b32) R(v_or3_b32) R(v_xad_u32) R(v_lshlrev_b32) R(v_lshrrev_b32)
    R(v_ashrrev_i32) R(v_alignbit_b32) R(v_bfe_u32) R(v_bfe_i32) R(v_bfi_b32) R(v_perm_b32) R(v_and_b32) R(v_xor_b32) R(v_mov_b32)
    R(v_cndmask_b32) R(v_min_u32) R(v_med3_u32) R(v_sad_u32) R(v_cmp_lt_u32_vcc) R(v_mul_lo_u32) R(v_mul_hi_u32) R(v_mul_u32_u24)
    R(v_mul_hi_u32_u24) R(v_mad_u32_u24) R(v_mad_i32_i24) R(v_mul_lo_u16) R(v_mad_u16) R(v_mad_u32_u16) R(v_pk_add_u16)
    R(v_pk_mul_lo_u16) R(v_pk_mad_u16) R(v_dot4_u32_u8) R(v_dot2_u32_u16) R(v_fma_f32) R(v_cvt_f32_u32) R(v_mov_dpp_ror)
    R(v_add_dpp_ror) R(v_mov_dpp_bcast) R(v_permlane32_swap) R(v_permlane16_swap) R(v_mad_u64_u32) R(v_mad_u64_u32_sgprc)
    R(v_lshl_add_u64) R(v_lshlrev_b64) R(v_lshrrev_b64) R(v_mov_b64) R(v_pk_mov_b32) R(v_add_f64) R(v_fma_f64) R(v_pk_add_f32)
    R(v_pk_fma_f32) R(v_cmp_lt_u64_vcc)
    return 0;
}
