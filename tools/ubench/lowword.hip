// frac(scale v + tq) as the low word of fma(scale, v, tq + 1.5 * 2^20): is it uniform on the device?  (round 6, the int16 boundary guard)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
__global__ void k(const double* v, int n, double scale, double tqm, uint32_t near_lo, unsigned long long* cnt, uint32_t* sample) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double w = fma(scale, v[i], tqm);
    const uint32_t lo = (uint32_t)__double2loint(w);
    if (lo <= near_lo) atomicAdd(cnt, 1ull);
    if (i < 8) { sample[2 * i] = lo; sample[2 * i + 1] = (uint32_t)__double2hiint(w); }
}
int main() {
    const int n = 1 << 26;
    double* hv = (double*)malloc(n * 8);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hv[i] = ((double)(s >> 11) * 0x1p-53 - 0.5) * 0.06; }
    double* dv; unsigned long long* dc; uint32_t* ds;
    hipMalloc(&dv, n * 8); hipMalloc(&dc, 8); hipMalloc(&ds, 64);
    hipMemcpy(dv, hv, n * 8, hipMemcpyHostToDevice); hipMemset(dc, 0, 8);
    const double tq = 1e-7, tqm = tq + 0x1.8p20;
    const uint32_t near_lo = (uint32_t)((tq + tq) * 0x1p32) + 2u;
    k<<<n / 256, 256>>>(dv, n, 32767.0, tqm, near_lo, dc, ds);
    unsigned long long c; uint32_t hs[16];
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, 64, hipMemcpyDeviceToHost);
    unsigned long long hc = 0;
    for (int i = 0; i < n; ++i) { double w = __builtin_fma(32767.0, hv[i], tqm); uint64_t b; memcpy(&b, &w, 8); if ((uint32_t)b <= near_lo) ++hc; }
    printf("near_lo %u: device %llu, host %llu of %d (expected %.1f)\n", near_lo, c, hc, n, (double)n * near_lo / 4294967296.0);
    for (int i = 0; i < 4; ++i) printf("  w[%d] = %08x %08x\n", i, hs[2 * i + 1], hs[2 * i]);
    return 0;
}
