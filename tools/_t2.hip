#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../twenty-first_amd/csrc/gl64.h"
typedef gl::u64 u64;
__global__ void k(const u64* a, const u64* v, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s, d;
    gl::add_sub(a[i], v[i], s, d);
    out[4 * i] = s; out[4 * i + 1] = d; out[4 * i + 2] = gl::add(a[i], v[i]); out[4 * i + 3] = gl::sub(a[i], v[i]);
}
int main() {
    std::vector<u64> edge = {0, 1, 2, 0xffffffffULL, 0x100000000ULL, 0xfffffffeULL, gl::P - 1, gl::P - 2, gl::P - 0xffffffffULL, gl::P - 0x100000000ULL, 0xffffffff00000000ULL, 0x8000000000000000ULL, 0x7fffffffffffffffULL, 0xfffffffeffffffffULL};
    std::vector<u64> a, v;
    for (u64 x : edge) for (u64 y : edge) { a.push_back(x); v.push_back(y); }
    u64 st = 88172645463325252ULL;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st % gl::P; };
    for (int i = 0; i < 1 << 20; i++) { a.push_back(rnd()); v.push_back(rnd()); }
    int n = a.size();
    u64 *da, *dv, *dout; hipMalloc(&da, n * 8); hipMalloc(&dv, n * 8); hipMalloc(&dout, n * 32);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(da, dv, dout, n);
    std::vector<u64> o(4 * (size_t)n); hipMemcpy(o.data(), dout, n * 32, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) if (o[4 * i] != o[4 * i + 2] || o[4 * i + 1] != o[4 * i + 3]) { if (bad++ < 10) printf("a=%lx v=%lx  s=%lx (%lx) d=%lx (%lx)\n", a[i], v[i], o[4*i], o[4*i+2], o[4*i+1], o[4*i+3]); }
    printf("n=%d bad=%d\n", n, bad);
}
