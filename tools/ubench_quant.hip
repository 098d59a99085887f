// Microbenchmark: access patterns for the float64 -> int16 quantiser (8 B read, 2 B written per sample; 150 M samples).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_quant.hip -o tools/ubench_quant.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ short q(double v, double scale, bool& bad) {
    double t = trunc(scale * v);
    if (!(t >= -32768.0 && t <= 32767.0)) { bad = true; t = 0.0; }
    return (short)(int)t;
}

// A: one sample per thread
__global__ __launch_bounds__(256) void kA(const double* __restrict__ in, size_t n, double scale, short* __restrict__ out, int* flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool bad = false;
    out[i] = q(in[i], scale, bad);
    if (bad) *flag = 1;
}
// B: 8 consecutive samples per thread (four 16-byte loads 64 bytes apart between lanes), one 16-byte store
template <bool NT>
__global__ __launch_bounds__(256) void kB(const d2* __restrict__ in, size_t n, double scale, s8* __restrict__ out, int* flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n / 8) return;
    d2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = NT ? __builtin_nontemporal_load(in + 4 * i + j) : in[4 * i + j];
    s8 r; bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = q(v[j >> 1][j & 1], scale, bad);
    if (bad) *flag = 1;
    out[i] = r;
}
// C: U coalesced 16-byte loads per thread (a workgroup covers U*512 samples), 4-byte stores
template <int U, bool NT>
__global__ __launch_bounds__(256) void kC(const d2* __restrict__ in, size_t n, double scale, s2* __restrict__ out, int* flag) {
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    d2 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) if (base + j * 256 < n / 2) v[j] = NT ? __builtin_nontemporal_load(in + base + j * 256) : in[base + j * 256];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < U; ++j) if (base + j * 256 < n / 2) { s2 r; r[0] = q(v[j][0], scale, bad); r[1] = q(v[j][1], scale, bad); out[base + j * 256] = r; }
    if (bad) *flag = 1;
}
// D: 4 consecutive samples per thread (two 16-byte loads), one 8-byte store
__global__ __launch_bounds__(256) void kD(const d2* __restrict__ in, size_t n, double scale, s4* __restrict__ out, int* flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n / 4) return;
    const d2 a = in[2 * i], b = in[2 * i + 1];
    s4 r; bool bad = false;
    r[0] = q(a[0], scale, bad); r[1] = q(a[1], scale, bad); r[2] = q(b[0], scale, bad); r[3] = q(b[1], scale, bad);
    if (bad) *flag = 1;
    out[i] = r;
}
// F: coalesced 16-byte loads, transposition through LDS, 16-byte stores (a workgroup covers 2048 samples)
__global__ __launch_bounds__(256) void kF(const d2* __restrict__ in, size_t n, double scale, s8* __restrict__ out, int* flag) {
    __shared__ s2 sh[1024];
    const size_t base = (size_t)blockIdx.x * 1024;
    bool bad = false;
    d2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(in + base + j * 256 + threadIdx.x);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s2 r; r[0] = q(v[j][0], scale, bad); r[1] = q(v[j][1], scale, bad); sh[j * 256 + threadIdx.x] = r; }
    if (bad) *flag = 1;
    __syncthreads();
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = ((const s8*)sh)[threadIdx.x];
}
// G: grid-stride loop of C-shaped accesses, 2048 workgroups
template <int U>
__global__ __launch_bounds__(256) void kG(const d2* __restrict__ in, size_t n, double scale, s2* __restrict__ out, int* flag) {
    const size_t step = (size_t)gridDim.x * 256;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * step < n / 2; i += U * step) {
        d2 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = __builtin_nontemporal_load(in + i + j * step);
#pragma unroll
        for (int j = 0; j < U; ++j) { s2 r; r[0] = q(v[j][0], scale, bad); r[1] = q(v[j][1], scale, bad); out[i + j * step] = r; }
    }
    if (bad) *flag = 1;
}

template <typename F>
static double time_ms(F&& launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 200; ++w) launch();        // clocks up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int R = 50;
    for (int r = 0; r < R; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / R;
}

int main() {
    const size_t n = 150000000 / 2048 * 2048;
    double* in; short* out; int* flag;
    hipMalloc(&in, n * 8); hipMalloc(&out, n * 2); hipMalloc(&flag, 4);
    hipMemset(in, 0, n * 8); hipMemset(flag, 0, 4);
    const double sc = 32767.0;
    const double bytes = n * 10.0;
    auto rep = [&](const char* name, double ms) { printf("%-44s %.4f ms  %.2f TB/s\n", name, ms, bytes / ms * 1e-9); };
    rep("A scalar", time_ms([&] { hipLaunchKernelGGL(kA, dim3((n + 255) / 256), dim3(256), 0, 0, in, n, sc, out, flag); }));
    rep("B lane-contiguous 4x16B nt -> 16B", time_ms([&] { hipLaunchKernelGGL(kB<true>, dim3((n / 8 + 255) / 256), dim3(256), 0, 0, (const d2*)in, n, sc, (s8*)out, flag); }));
    rep("B lane-contiguous 4x16B -> 16B", time_ms([&] { hipLaunchKernelGGL(kB<false>, dim3((n / 8 + 255) / 256), dim3(256), 0, 0, (const d2*)in, n, sc, (s8*)out, flag); }));
    rep("C coalesced 4x16B nt -> 4B", time_ms([&] { hipLaunchKernelGGL((kC<4, true>), dim3(n / 2048), dim3(256), 0, 0, (const d2*)in, n, sc, (s2*)out, flag); }));
    rep("C coalesced 4x16B -> 4B", time_ms([&] { hipLaunchKernelGGL((kC<4, false>), dim3(n / 2048), dim3(256), 0, 0, (const d2*)in, n, sc, (s2*)out, flag); }));
    rep("C coalesced 2x16B nt -> 4B", time_ms([&] { hipLaunchKernelGGL((kC<2, true>), dim3(n / 1024), dim3(256), 0, 0, (const d2*)in, n, sc, (s2*)out, flag); }));
    rep("C coalesced 8x16B nt -> 4B", time_ms([&] { hipLaunchKernelGGL((kC<8, true>), dim3(n / 4096), dim3(256), 0, 0, (const d2*)in, n, sc, (s2*)out, flag); }));
    rep("D lane-contiguous 2x16B -> 8B", time_ms([&] { hipLaunchKernelGGL(kD, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, (const d2*)in, n, sc, (s4*)out, flag); }));
    rep("F coalesced 4x16B nt, LDS -> 16B", time_ms([&] { hipLaunchKernelGGL(kF, dim3(n / 2048), dim3(256), 0, 0, (const d2*)in, n, sc, (s8*)out, flag); }));
    rep("G grid-stride 4x16B nt -> 4B, 2048 wg", time_ms([&] { hipLaunchKernelGGL(kG<4>, dim3(2048), dim3(256), 0, 0, (const d2*)in, n, sc, (s2*)out, flag); }));
    rep("G grid-stride 8x16B nt -> 4B, 4096 wg", time_ms([&] { hipLaunchKernelGGL(kG<8>, dim3(4096), dim3(256), 0, 0, (const d2*)in, n, sc, (s2*)out, flag); }));
    return 0;
}
