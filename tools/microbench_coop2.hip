// microbench_coop2.hip -- the latency of ONE Tip5 permutation (what a tree level near the root, or a chunk of one long sponge, waits for):
//   A  the product's 16-lane form (tip5_permutation_coop, tip5_kernels.h): lane j of a DPP row owns state word j, the circulant is 16 row
//      rotations + 32 v_mad_u64_u32 in four chains
//   B  the same permutation on TWO rows (32 lanes): both rows hold the state and run the S-box layer side by side (one instruction stream:
//      no extra time), row h takes the eight rotation terms k = 8 h .. 8 h + 7 of the circulant -- the state rotated by 8 first in row 1, the
//      matrix entries M[k + 8 h] as per-lane registers -- and gfx950's v_permlane16_swap_b32 joins the two partial sums (four swaps, two
//      64-bit additions): 16 rotations + 16 products per lane instead of 32 + 32.
// The question: priced at about -170 of ~920 cycles a round.  Both forms run a chain
// of N dependent permutations in one wave (nothing else on the chip), checked against each other and against the lane-per-permutation
// round (tfk::tip5_permutation) word for word.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I twenty-first_amd/csrc -o tools/microbench_coop2 tools/microbench_coop2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gl64.h"
#include "tip5_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
using gl::u32;
using gl::u64;
using namespace tfk;

// (form B is tfk::tip5_permutation_coop2 of tip5_kernels.h since it was adopted; this file is the measurement behind the adoption)

// one wave; rows 0 (A) / rows 0-1 (B) run `chain` dependent permutations of the state at `state` (16 words), the result goes back
template <int FORM>
__global__ void __launch_bounds__(256) chain_kernel(u64* state, int chain) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    const int lane = threadIdx.x & 63, j = lane & 15, half = (lane >> 4) & 1;
    u64 s = state[j];
    u64 rcs[5];
    coop_round_constants(j, rcs);
    CoopHalfMatrix hm;
    coop_half_matrix(half, hm);
    stage_lut(lut);
    if (threadIdx.x >= 64) return;
    if (FORM == 0) {
        if (lane >= 16) return;
#pragma unroll 1
        for (int i = 0; i < chain; ++i) tip5_permutation_coop(s, j, lut, rcs);
    } else {
        if (lane >= 32) return;
#pragma unroll 1
        for (int i = 0; i < chain; ++i) tip5_permutation_coop2(s, j, half, lut, rcs, hm);
    }
    if (lane < 16) state[j] = s;
}

// reference: the lane-per-permutation round of the product
__global__ void __launch_bounds__(256) ref_kernel(u64* state, int chain) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256];
    stage_lut(lut);
    if (threadIdx.x) return;
    u64 s[16];
    for (int i = 0; i < 16; ++i) s[i] = state[i];
    for (int i = 0; i < chain; ++i) tip5_permutation(s, lut);
    for (int i = 0; i < 16; ++i) state[i] = s[i];
}

int main() {
    // the three forms read the same constant block; the comparison does not depend on its values (the product's literals live in tf_tip5.hip):
    // eighty pseudo-random canonical words
    static Tip5Consts hc;
    u64 z = 0x7F210003ULL;
    for (int i = 0; i < 80; ++i) {
        z += 0x9e3779b97f4a7c15ULL;
        u64 v = z;
        v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ULL;
        v = (v ^ (v >> 27)) * 0x94d049bb133111ebULL;
        v ^= v >> 31;
        hc.rc[i] = v % gl::P;
    }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tip5), &hc, sizeof hc));
    u64 h0[16];
    for (int i = 0; i < 16; ++i) h0[i] = gl::to_mont(0x1234567ull * (i + 1) + (u64)i * 0x9e3779b97f4a7c15ULL % gl::P);
    u64* d[3];
    for (int v = 0; v < 3; ++v) {
        CK(hipMalloc(&d[v], 128));
        CK(hipMemcpy(d[v], h0, 128, hipMemcpyHostToDevice));
    }
    const int check = 37;
    hipLaunchKernelGGL(ref_kernel, dim3(1), dim3(256), 0, 0, d[0], check);
    hipLaunchKernelGGL(chain_kernel<0>, dim3(1), dim3(256), 0, 0, d[1], check);
    hipLaunchKernelGGL(chain_kernel<1>, dim3(1), dim3(256), 0, 0, d[2], check);
    CK(hipDeviceSynchronize());
    u64 r[3][16];
    for (int v = 0; v < 3; ++v) CK(hipMemcpy(r[v], d[v], 128, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; ++i) bad += (r[0][i] != r[1][i]) + (r[0][i] != r[2][i]);
    printf("%d chained permutations: 16-lane form and two-row form against the lane-per-permutation round: %d of 32 words differ (word 0 = %016llx)\n", check, bad,
           (unsigned long long)r[0][0]);
    if (bad) return 1;
    const int chain = 20000;
    for (int form = 0; form < 2; ++form) {
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (form) hipLaunchKernelGGL(chain_kernel<1>, dim3(1), dim3(256), 0, 0, d[2], chain);
            else hipLaunchKernelGGL(chain_kernel<0>, dim3(1), dim3(256), 0, 0, d[1], chain);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%s: %d dependent permutations in %.3f ms = %.3f us per permutation\n", form ? "B two rows (32 lanes), v_permlane16_swap joins the halves" : "A one row (16 lanes), the product's form            ", chain,
               best, best * 1e3 / chain);
    }
    CK(hipMemcpy(r[1], d[1], 128, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r[2], d[2], 128, hipMemcpyDeviceToHost));
    bad = 0;
    for (int i = 0; i < 16; ++i) bad += r[1][i] != r[2][i];
    printf("after the timed chains (the same number of permutations on both): %d of 16 words differ\n", bad);
    return bad != 0;
}
