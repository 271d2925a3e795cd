// GPU unit check of the hand-scheduled field primitives in gl64.h against the plain C++ forms (edge values + random).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o asm_unit tools/asm_unit.hip && ./asm_unit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../twenty-first_amd/csrc/gl64.h"
typedef gl::u64 u64;
__global__ void k(const u64* a, const u64* v, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s, d, m1, m2;
    gl::add_sub(a[i], v[i], s, d);
    gl::mont_mul2(a[i], v[i], v[i], v[i], m1, m2);
    out[8 * i] = s; out[8 * i + 1] = d; out[8 * i + 2] = gl::add(a[i], v[i]); out[8 * i + 3] = gl::sub(a[i], v[i]);
    out[8 * i + 4] = m1; out[8 * i + 5] = m2; out[8 * i + 6] = gl::mont_mul(a[i], v[i]); out[8 * i + 7] = gl::mont_mul(v[i], v[i]);
}
int main() {
    std::vector<u64> edge = {0, 1, 2, 0xffffffffULL, 0x100000000ULL, 0xfffffffeULL, gl::P - 1, gl::P - 2, gl::P - 0xffffffffULL, gl::P - 0x100000000ULL, 0xffffffff00000000ULL, 0x8000000000000000ULL, 0x7fffffffffffffffULL, 0xfffffffeffffffffULL, 0xfffffffe00000001ULL, 0xfffffffe00000002ULL};
    std::vector<u64> a, v;
    for (u64 x : edge) for (u64 y : edge) { a.push_back(x % gl::P); v.push_back(y % gl::P); }
    u64 st = 88172645463325252ULL;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st % gl::P; };
    for (int i = 0; i < 1 << 22; i++) { a.push_back(rnd()); v.push_back(rnd()); }
    int n = a.size();
    u64 *da, *dv, *dout;
    if (hipMalloc(&da, n * 8) || hipMalloc(&dv, n * 8) || hipMalloc(&dout, (size_t)n * 64)) return 2;
    (void)hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(da, dv, dout, n);
    std::vector<u64> o(8 * (size_t)n); (void)hipMemcpy(o.data(), dout, (size_t)n * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++)
        if (o[8 * i] != o[8 * i + 2] || o[8 * i + 1] != o[8 * i + 3] || o[8 * i + 4] != o[8 * i + 6] || o[8 * i + 5] != o[8 * i + 7]) {
            if (bad++ < 10) printf("a=%lx v=%lx  s=%lx (%lx) d=%lx (%lx) ab=%lx (%lx) vv=%lx (%lx)\n", a[i], v[i], o[8*i], o[8*i+2], o[8*i+1], o[8*i+3], o[8*i+4], o[8*i+6], o[8*i+5], o[8*i+7]);
        }
    printf("n=%d bad=%d\n", n, bad);
    return bad != 0;
}
