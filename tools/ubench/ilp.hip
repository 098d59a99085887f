// Microbenchmark: how many INDEPENDENT float64 FMA chains does one wavefront need to keep a gfx950 SIMD's FMA pipe busy, as a
// function of the wavefronts that share the SIMD?  (Round 4: the headline kernel interleaves two Horner chains per wavefront and
// runs three wavefronts per SIMD; the oldest wavefront is served first, so the launch ends with SIMDs that hold one or two
// wavefronts -- does a lone wavefront with two chains stall?)
//   grid = 256 * W workgroups of 256 threads (one wavefront per SIMD each), dynamic LDS sized so that at most W workgroups fit a CU;
//   every wavefront runs ITERS iterations of CHAINS chains x = fma(x, a, b) and reports its s_memtime cycles.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/ilp.hip -o tools/ubench/ilp.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

constexpr int ITERS = 4096;

template <int CHAINS>
__global__ __launch_bounds__(256) void k(double a, double b, double* out, unsigned long long* cyc) {
    extern __shared__ double pad[];
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = (double)(threadIdx.x + c) * 1e-3;
    if (a == 12345.0) pad[threadIdx.x] = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) x[c] = fma(x[c], a, b);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    if (s == 0.123) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS>
void run(int W, double* d_out, unsigned long long* d_cyc) {
    const int wgs = 256 * W;
    const size_t lds = W == 1 ? 160 * 1024 - 256 : (160 * 1024 / W) - 1024;       // at most W workgroups per CU
    hipFuncSetAttribute((const void*)k<CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CHAINS>, dim3(wgs), dim3(256), lds, 0, 0.999, 1e-3, d_out, d_cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> c(wgs * 4);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2], mx = (double)c.back();
    // s_memtime counts at 100 MHz on gfx950?  report both the raw count per instruction and the event time
    const double instr = (double)ITERS * CHAINS;
    printf("W=%d waves/SIMD  chains=%d : %8.1f us  | memtime ticks per wave: median %.0f max %.0f | ns per instr per wave (event) %.3f | "
           "SIMD-ns per instruction (event / (W x instr)) %.3f\n",
           W, CHAINS, best * 1e3, med, mx, best * 1e6 / instr, best * 1e6 / (instr * W));
}

int main() {
    double* d_out;
    unsigned long long* d_cyc;
    hipMalloc(&d_out, 4096);
    hipMalloc(&d_cyc, 8 * 4096);
    for (int W = 1; W <= 4; ++W) {
        run<1>(W, d_out, d_cyc);
        run<2>(W, d_out, d_cyc);
        run<3>(W, d_out, d_cyc);
        run<4>(W, d_out, d_cyc);
        run<6>(W, d_out, d_cyc);
        run<8>(W, d_out, d_cyc);
    }
    return 0;
}
