// microbench3.hip -- HBM streaming variants: what copy shape reaches the highest read+write rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef uint64_t u64;

template <int VEC, int UNROLL>  // VEC = 8 or 16 bytes per access
__global__ void __launch_bounds__(256) copy_direct(const char* __restrict__ src, char* __restrict__ dst) {
  // each block copies a contiguous chunk of 256*VEC*UNROLL bytes, UNROLL accesses per thread, all loads first
  size_t base = (size_t)blockIdx.x * 256 * VEC * UNROLL + (size_t)threadIdx.x * VEC;
  if constexpr (VEC == 16) {
    uint4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = *reinterpret_cast<const uint4*>(src + base + (size_t)i * 256 * VEC);
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) *reinterpret_cast<uint4*>(dst + base + (size_t)i * 256 * VEC) = v[i];
  } else {
    uint2 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = *reinterpret_cast<const uint2*>(src + base + (size_t)i * 256 * VEC);
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) *reinterpret_cast<uint2*>(dst + base + (size_t)i * 256 * VEC) = v[i];
  }
}
template <int UNROLL>
__global__ void __launch_bounds__(256) read_only(const char* __restrict__ src, u64* sink) {
  size_t base = (size_t)blockIdx.x * 256 * 16 * UNROLL + (size_t)threadIdx.x * 16;
  uint4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) { uint4 v = *reinterpret_cast<const uint4*>(src + base + (size_t)i * 256 * 16); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc.z;
}
template <int UNROLL>
__global__ void __launch_bounds__(256) write_only(char* __restrict__ dst, unsigned seed) {
  size_t base = (size_t)blockIdx.x * 256 * 16 * UNROLL + (size_t)threadIdx.x * 16;
  uint4 v = {seed, threadIdx.x, blockIdx.x, 7};
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) *reinterpret_cast<uint4*>(dst + base + (size_t)i * 256 * 16) = v;
}
// column tile pattern as in the NTT pass: 512 threads, 16 columns x 1024 rows of u64, 32 loads then 32 stores per thread
__global__ void __launch_bounds__(512) coltile16(const u64* __restrict__ src, u64* __restrict__ dst) {
  size_t mat = blockIdx.x / 64; int tile = blockIdx.x % 64;
  const u64* s = src + mat * (1u << 20) + tile * 16;
  u64* d = dst + mat * (1u << 20) + tile * 16;
  int c = threadIdx.x % 16, g = threadIdx.x / 16;
  u64 v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = s[(size_t)(g + 32 * i) * 1024 + c];
#pragma unroll
  for (int i = 0; i < 32; ++i) d[(size_t)(g + 32 * i) * 1024 + c] = v[i];
}
// row tile: each block copies 16 KiB contiguous x 8 rows?  (fully linear 128 KiB per block, 512 threads x 32 x 8B)
__global__ void __launch_bounds__(512) lintile(const u64* __restrict__ src, u64* __restrict__ dst) {
  size_t base = (size_t)blockIdx.x * 16384 + threadIdx.x;
  u64 v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = src[base + (size_t)i * 512];
#pragma unroll
  for (int i = 0; i < 32; ++i) dst[base + (size_t)i * 512] = v[i];
}
// 8-column tiles (64 B segments), 256 threads; PAIR=1: blocks b and b+8 (same XCD, back to back) take the two
// halves of the same 128-byte lines.
template <int PAIR>
__global__ void __launch_bounds__(256) coltile8(const u64* __restrict__ src, u64* __restrict__ dst, unsigned ntiles) {
  unsigned b = blockIdx.x, tileid;
  if (PAIR) { unsigned xcd = b & 7, slot = b >> 3; tileid = ((slot >> 1) * 8 + xcd) * 2 + (slot & 1); }
  else tileid = b;
  size_t mat = tileid / 128; int tile = tileid % 128;
  const u64* s = src + mat * (1u << 20) + tile * 8;
  u64* d = dst + mat * (1u << 20) + tile * 8;
  int c = threadIdx.x % 8, g = threadIdx.x / 8;
  u64 v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = s[(size_t)(g + 32 * i) * 1024 + c];
#pragma unroll
  for (int i = 0; i < 32; ++i) d[(size_t)(g + 32 * i) * 1024 + c] = v[i];
}
int main() {
  size_t total = (size_t)2 << 30;
  char *a, *b; u64* sink;
  CK(hipMalloc(&a, total)); CK(hipMalloc(&b, total)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, total)); CK(hipMemset(b, 2, total));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define TIME(NAME, BYTES, ...) { float best = 1e9; for (int r = 0; r < 4; ++r) { CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } CK(hipGetLastError()); printf("%-34s %7.3f ms  %6.2f TB/s\n", NAME, best, (double)(BYTES) / best * 1e-9); }
  TIME("copy 16B x1", 2 * total, hipLaunchKernelGGL((copy_direct<16, 1>), dim3(total / (256 * 16 * 1)), dim3(256), 0, 0, a, b));
  TIME("copy 16B x4", 2 * total, hipLaunchKernelGGL((copy_direct<16, 4>), dim3(total / (256 * 16 * 4)), dim3(256), 0, 0, a, b));
  TIME("copy 16B x8", 2 * total, hipLaunchKernelGGL((copy_direct<16, 8>), dim3(total / (256 * 16 * 8)), dim3(256), 0, 0, a, b));
  TIME("copy 16B x16", 2 * total, hipLaunchKernelGGL((copy_direct<16, 16>), dim3(total / (256 * 16 * 16)), dim3(256), 0, 0, a, b));
  TIME("copy 8B x8", 2 * total, hipLaunchKernelGGL((copy_direct<8, 8>), dim3(total / (256 * 8 * 8)), dim3(256), 0, 0, a, b));
  TIME("copy 8B x32", 2 * total, hipLaunchKernelGGL((copy_direct<8, 32>), dim3(total / (256 * 8 * 32)), dim3(256), 0, 0, a, b));
  TIME("read-only 16B x8", total, hipLaunchKernelGGL((read_only<8>), dim3(total / (256 * 16 * 8)), dim3(256), 0, 0, a, sink));
  TIME("read-only 16B x16", total, hipLaunchKernelGGL((read_only<16>), dim3(total / (256 * 16 * 16)), dim3(256), 0, 0, a, sink));
  TIME("write-only 16B x8", total, hipLaunchKernelGGL((write_only<8>), dim3(total / (256 * 16 * 8)), dim3(256), 0, 0, b, 3u));
  TIME("write-only 16B x16", total, hipLaunchKernelGGL((write_only<16>), dim3(total / (256 * 16 * 16)), dim3(256), 0, 0, b, 3u));
  TIME("coltile16 512thr (NTT pattern)", 2 * total, hipLaunchKernelGGL(coltile16, dim3(256 * 64), dim3(512), 0, 0, (const u64*)a, (u64*)b));
  TIME("lintile 512thr 32x8B", 2 * total, hipLaunchKernelGGL(lintile, dim3(total / (16384 * 8)), dim3(512), 0, 0, (const u64*)a, (u64*)b));
  TIME("in-place coltile16", 2 * total, hipLaunchKernelGGL(coltile16, dim3(256 * 64), dim3(512), 0, 0, (const u64*)a, (u64*)a));
  TIME("in-place copy 16B x8", 2 * total, hipLaunchKernelGGL((copy_direct<16, 8>), dim3(total / (256 * 16 * 8)), dim3(256), 0, 0, a, a));
  TIME("coltile8 256thr natural order", 2 * total, hipLaunchKernelGGL((coltile8<0>), dim3(256 * 128), dim3(256), 0, 0, (const u64*)a, (u64*)b, 256u * 128u));
  TIME("coltile8 256thr XCD-paired", 2 * total, hipLaunchKernelGGL((coltile8<1>), dim3(256 * 128), dim3(256), 0, 0, (const u64*)a, (u64*)b, 256u * 128u));
  return 0;
}
