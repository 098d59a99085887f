// Microbenchmark: what the instructions of the integer resample kernel's exact division cost on gfx950 -- float64 <-> uint32
// conversions, the float64 fma between them, and their float32 counterparts -- as wave-instructions per cycle per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_cvt.hip -o tools/ubench_cvt.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, double a, double b, unsigned m) {
    unsigned u0 = threadIdx.x, u1 = u0 + 11, u2 = u0 + 22, u3 = u0 + 33, u4 = u0 + 44, u5 = u0 + 55, u6 = u0 + 66, u7 = u0 + 77;
    const float fa = (float)a, fb = (float)b;
    for (int i = 0; i < iters; ++i) {
#define EACH(X) X(u0) X(u1) X(u2) X(u3) X(u4) X(u5) X(u6) X(u7)
        if (MODE == 0) {        // the kernel's division: cvt_f64_u32, fma_f64, cvt_u32_f64   (3 instructions)
#define OP(v) v = (unsigned)fma((double)v, a, b);
            EACH(OP)
#undef OP
        } else if (MODE == 1) { // the float32 form: cvt_f32_u32, fma_f32, cvt_u32_f32         (3 instructions)
#define OP(v) v = (unsigned)fmaf((float)v, fa, fb);
            EACH(OP)
#undef OP
        } else if (MODE == 2) { // float32 with an integer correction step                       (7 instructions)
#define OP(v) { unsigned q = (unsigned)((float)v * fa); int r = (int)v - (int)__umul24(q, m); q += (r >= (int)m) ? 1u : 0u; v = q + (v & 0xFFFFu); }
            EACH(OP)
#undef OP
        } else if (MODE == 3) { // 24-bit multiply + add (reference: a plain full-rate pair)      (2 instructions)
#define OP(v) v = __umul24(v, m) + 12345u;
            EACH(OP)
#undef OP
        } else if (MODE == 4) { // cvt_f64_u32 + cvt_u32_f64 only                                  (2 instructions)
#define OP(v) v = (unsigned)((double)v);
            EACH(OP)
#undef OP
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7;
}

template <int MODE>
void run(const char* name, int per_value) {
    const int blocks = 8192, iters = 4000;
    unsigned* out;
    hipMalloc(&out, sizeof(unsigned) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.00625, 0.003125, 160u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double values = (double)blocks * 4 * iters * 8;          // wave-level operations
    const double cyc = ms * 1e-3 * 2.4e9 * 1024.0 / values;          // SIMD cycles per wave-operation
    printf("%-44s %.3f ms: %.2f cycles per value (%d instructions -> %.2f cycles each)\n", name, ms, cyc, per_value, cyc / per_value);
    hipFree(out);
}

int main() {
    run<3>("u24 mul + add", 2);
    run<4>("cvt f64<-u32, cvt u32<-f64", 2);
    run<0>("cvt f64<-u32, fma f64, cvt u32<-f64", 3);
    run<1>("cvt f32<-u32, fma f32, cvt u32<-f32", 3);
    run<2>("f32 quotient + integer correction", 7);
    return 0;
}
