// microbench_mfma_valu_mix.hip -- do the matrix pipe and the vector ALU of one SIMD run concurrently on gfx950?
// Odd waves of every workgroup issue only MFMAs, even waves only VALU instructions; each half is timed alone and then together.
// "both = max" means the pipes overlap (what a Tip5 MDS on the matrix pipe needs), "both = sum" means they do not.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o tools/microbench_mfma_valu_mix tools/microbench_mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef uint64_t u64; typedef uint32_t u32;
// mode bit0: MFMA waves work, bit1: VALU waves work.  KIND 0 = f64 mfma, 1 = f32 16x16x4 mfma, 2 = bf16 32x32x16 mfma, 3 = i8 16x16x64 mfma
// VK: 0 v_mad_u64_u32, 1 v_addc_co_u32, 2 v_and_b32
template <int KIND, int VK>
__global__ void __launch_bounds__(256) mix(double* out, int it_m, int it_v, int mode) {
    const int wave = threadIdx.x >> 6;
    double res = 0;
    if (wave & 1) {
        if (mode & 1) {
            if (KIND == 0) {
                d4 acc[2] = {d4{0,0,0,0}, d4{0,0,0,0}};
                const double a = 1.0 + threadIdx.x, b = 0.5;
#pragma unroll 1
                for (int it = 0; it < it_m; ++it) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[1], 0, 0, 0); }
                }
                res = acc[0][0] + acc[1][1];
            } else if (KIND == 3) {
                i4v acc[2] = {i4v{0,0,0,0}, i4v{0,0,0,0}};
                const i4v a = {(int)threadIdx.x * 0x01010101, 0x01020304, 0x7f007f00, 0x11111111}, b = {0x01010101, 0x02020202, 0x03030303, 0x04040404};
#pragma unroll 1
                for (int it = 0; it < it_m; ++it) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[1], 0, 0, 0); }
                }
                res = (double)(acc[0][0] + acc[1][1]);
            } else if (KIND == 2) {
                f16v acc[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0;
                bf8 a, b;
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x + i), b[i] = (__bf16)0.5f;
#pragma unroll 1
                for (int it = 0; it < it_m; ++it) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0); }
                }
                res = acc[0][0] + acc[1][1];
            } else {
                f4 acc[2] = {f4{0,0,0,0}, f4{0,0,0,0}};
                const float a = 1.0f + threadIdx.x, b = 0.5f;
#pragma unroll 1
                for (int it = 0; it < it_m; ++it) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[1], 0, 0, 0); }
                }
                res = acc[0][0] + acc[1][1];
            }
        }
    } else {
        if (mode & 2) {
            u64 w[8]; u32 b = threadIdx.x * 2654435761u, c = threadIdx.x ^ 40503u;
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = (u64)b * (i + 3);
            u32 x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = b * (i + 5);
#pragma unroll 1
            for (int it = 0; it < it_v; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (VK == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
                        else if (VK == 1) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(x[i]) : "v"(b), "v"(c) : "vcc");
                        else asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(b), "v"(c) : "vcc");
                    }
            }
            u64 t = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) t ^= w[i] ^ x[i];
            res = (double)t;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = res;
}
template <int KIND, int VK>
void run(double* d, int cus, int blocks_per_cu, int it_m, int it_v) {
    float ms[4] = {0,0,0,0};
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((mix<KIND, VK>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, it_m, it_v, mode);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms[mode], e0, e1));
        }
    }
    const char* vn[3] = {"v_mad_u64_u32", "v_addc_co_u32", "v_and_b32"};
    const char* mn[4] = {"f64 16x16x4", "f32 16x16x4", "bf16 32x32x16", "i8 16x16x64"};
    printf("%-13s mfma + %-13s, %d blocks/CU (%d mfma waves + %d valu waves per SIMD): mfma only %7.3f ms, valu only %7.3f ms, both %7.3f ms  (sum %7.3f, max %7.3f)\n",
           mn[KIND], vn[VK], blocks_per_cu, blocks_per_cu / 2, blocks_per_cu / 2, ms[1], ms[2], ms[3], ms[1] + ms[2], ms[1] > ms[2] ? ms[1] : ms[2]);
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* d; CK(hipMalloc(&d, (size_t)cus * 8 * 256 * 8));
    // f64: 16 mfma x 64 cyc = 1024 cyc per iteration; valu: 32 mads x ~4.7 cyc = 150 cyc per iteration
    for (int b : {2, 8}) run<0, 0>(d, cus, b, 2000, 2000 * 7);
    for (int b : {2, 8}) run<0, 1>(d, cus, b, 2000, 2000 * 7);
    for (int b : {2, 8}) run<0, 2>(d, cus, b, 2000, 2000 * 14);
    for (int b : {2, 8}) run<1, 0>(d, cus, b, 4000, 2000 * 7);
    for (int b : {2, 8}) run<2, 0>(d, cus, b, 4000, 2000 * 7);
    for (int b : {2, 8}) run<2, 1>(d, cus, b, 4000, 2000 * 7);
    for (int b : {2, 8}) run<3, 0>(d, cus, b, 8000, 2000 * 7);
    for (int b : {2, 8}) run<3, 1>(d, cus, b, 8000, 2000 * 7);
    return 0;
}
