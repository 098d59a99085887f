// Microbenchmark: float64 / float32 VALU instruction throughput on gfx950 (peak for the oscillator kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    float fa = (float)a, fb = (float)b;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent f64 fma chains
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
            x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
        } else if (MODE == 1) {   // 1 dependent f64 fma chain (8 per iter)
            x0 = fma(x0, a, b); x0 = fma(x0, a, b); x0 = fma(x0, a, b); x0 = fma(x0, a, b);
            x0 = fma(x0, a, b); x0 = fma(x0, a, b); x0 = fma(x0, a, b); x0 = fma(x0, a, b);
        } else if (MODE == 2) {   // f64 mul
            x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
        } else if (MODE == 3) {   // f64 add
            x0 = x0 + a; x1 = x1 + a; x2 = x2 + a; x3 = x3 + a; x4 = x4 + a; x5 = x5 + a; x6 = x6 + a; x7 = x7 + a;
        } else if (MODE == 4) {   // f32 fma
            f0 = fmaf(f0, fa, fb); f1 = fmaf(f1, fa, fb); f2 = fmaf(f2, fa, fb); f3 = fmaf(f3, fa, fb);
            f4 = fmaf(f4, fa, fb); f5 = fmaf(f5, fa, fb); f6 = fmaf(f6, fa, fb); f7 = fmaf(f7, fa, fb);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int MODE>
void run(const char* name, int blocks, int iters) {
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999999, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * 8;      // wave-instructions
    double per_simd_cycle = insts / 1024.0 / (ms * 1e-3 * 2.4e9);
    printf("%-28s blocks %5d: %.3f ms, %.2f T lane-ops/s, %.3f wave-instr/cycle/SIMD @2.4GHz (=> %.2f cyc/instr)\n", name, blocks, ms,
           insts * 64 / (ms * 1e-3) / 1e12, per_simd_cycle, 1.0 / per_simd_cycle);
    hipFree(out);
}

int main() {
    for (int blocks : {1024, 2048, 8192}) {
        run<0>("f64 fma x8 independent", blocks, 20000);
        run<1>("f64 fma dependent chain", blocks, 20000);
        run<2>("f64 mul", blocks, 20000);
        run<3>("f64 add", blocks, 20000);
        run<4>("f32 fma", blocks, 20000);
    }
    return 0;
}
