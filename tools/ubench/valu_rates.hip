// Microbenchmark: issue cost of the VALU instructions the 16-bit resample kernel is made of (gfx950), in SIMD cycles per
// wave-instruction: conversions to and from float64 / float32, 32-bit and 24-bit integer multiplies, a float64 FMA for scale.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t c) {
    uint32_t u[8];
    double d[8];
    float f[8];
    for (int j = 0; j < 8; ++j) { u[j] = threadIdx.x * 8 + j + c; d[j] = (double)u[j]; f[j] = (float)u[j]; }
    for (int i = 0; i < iters; ++i) {
#define CVT_F64_U32(j) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[j]) : "v"(u[j]));
#define CVT_U32_F64(j) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u[j]) : "v"(d[j]));
#define CVT_F32_U32(j) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[j]) : "v"(u[j]));
#define CVT_U32_F32(j) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[j]) : "v"(f[j]));
#define MUL_HI(j) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define MUL_LO(j) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define MUL_U24(j) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define MAD_U24(j) asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define FMA_F64(j) asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(d[j]) : "v"(d[j]));
#define FMA_F32(j) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(f[j]) : "v"(f[j]));
#define ALIGNBIT(j) asm volatile("v_alignbit_b32 %0, %1, %2, %2" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define ADD_U32(j) asm volatile("v_add_u32 %0, %1, %2" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define XOR3(j) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u[j]) : "v"(u[j]), "v"(c));
// (round 5: what the int16 materialisation's epilogue and the pan chain are made of)
#define FLOOR_F64(j) asm volatile("v_floor_f64 %0, %1" : "=v"(d[j]) : "v"(d[j]));
#define TRUNC_F64(j) asm volatile("v_trunc_f64 %0, %1" : "=v"(d[j]) : "v"(d[j]));
#define CVT_I32_F64(j) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[j]) : "v"(d[j]));
#define CVT_F64_I32(j) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[j]) : "v"(u[j]));
#define MUL_F64(j) asm volatile("v_mul_f64 %0, %1, %1" : "=v"(d[j]) : "v"(d[j]));
#define ADD_F64(j) asm volatile("v_add_f64 %0, %1, %1" : "=v"(d[j]) : "v"(d[j]));
#define MAX_F64(j) asm volatile("v_max_f64 %0, %1, %1" : "=v"(d[j]) : "v"(d[j]));
#define CVT_PK(j) asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define PK_ADD(j) asm volatile("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define PERM(j) asm volatile("v_perm_b32 %0, %1, %2, %2" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define DPP(j) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(u[j]) : "v"(u[j]));
#define MAX3(j) asm volatile("v_max3_i32 %0, %1, %2, %2" : "=v"(u[j]) : "v"(u[j]), "v"(c));
#define BFE(j) asm volatile("v_bfe_i32 %0, %1, 0, 16" : "=v"(u[j]) : "v"(u[j]));
#define CMP_CND(j) asm volatile("v_cmp_lt_f64 vcc, %1, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(u[j]) : "v"(d[j]), "v"(c) : "vcc");
#define DOT2(j) asm volatile("v_dot2_i32_i16 %0, %1, %2, %1" : "=v"(u[j]) : "v"(u[j]), "v"(c));
        if (MODE == 0) { REP8(CVT_F64_U32) }
        else if (MODE == 1) { REP8(CVT_U32_F64) }
        else if (MODE == 2) { REP8(CVT_F32_U32) }
        else if (MODE == 3) { REP8(CVT_U32_F32) }
        else if (MODE == 4) { REP8(MUL_HI) }
        else if (MODE == 5) { REP8(MUL_LO) }
        else if (MODE == 6) { REP8(MUL_U24) }
        else if (MODE == 7) { REP8(MAD_U24) }
        else if (MODE == 8) { REP8(FMA_F64) }
        else if (MODE == 9) { REP8(FMA_F32) }
        else if (MODE == 10) { REP8(ALIGNBIT) }
        else if (MODE == 11) { REP8(ADD_U32) }
        else if (MODE == 12) { REP8(XOR3) }
        else if (MODE == 13) { REP8(FLOOR_F64) }
        else if (MODE == 14) { REP8(TRUNC_F64) }
        else if (MODE == 15) { REP8(CVT_I32_F64) }
        else if (MODE == 16) { REP8(CVT_F64_I32) }
        else if (MODE == 17) { REP8(MUL_F64) }
        else if (MODE == 18) { REP8(ADD_F64) }
        else if (MODE == 19) { REP8(MAX_F64) }
        else if (MODE == 20) { REP8(CVT_PK) }
        else if (MODE == 21) { REP8(PK_ADD) }
        else if (MODE == 22) { REP8(PERM) }
        else if (MODE == 23) { REP8(DPP) }
        else if (MODE == 24) { REP8(MAX3) }
        else if (MODE == 25) { REP8(BFE) }
        else if (MODE == 26) { REP8(CMP_CND) }
        else if (MODE == 27) { REP8(DOT2) }
    }
    uint32_t s = 0;
    for (int j = 0; j < 8; ++j) s += u[j] + (uint32_t)d[j] + (uint32_t)f[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters) {
    uint32_t* out;
    (void)hipMalloc(&out, sizeof(uint32_t) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    double insts = (double)blocks * 4 * iters * 8;      // wave-instructions
    double per_simd_cycle = insts / 1024.0 / (ms * 1e-3 * 2.4e9);
    printf("%-16s blocks %5d: %.3f ms, %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, blocks, ms, 1.0 / per_simd_cycle);
    (void)hipFree(out);
}

int main() {
    const int blocks = 8192, it = 4000;
    run<8>("v_fma_f64", blocks, it);
    run<9>("v_fma_f32", blocks, it);
    run<0>("v_cvt_f64_u32", blocks, it);
    run<1>("v_cvt_u32_f64", blocks, it);
    run<2>("v_cvt_f32_u32", blocks, it);
    run<3>("v_cvt_u32_f32", blocks, it);
    run<4>("v_mul_hi_u32", blocks, it);
    run<5>("v_mul_lo_u32", blocks, it);
    run<6>("v_mul_u32_u24", blocks, it);
    run<7>("v_mad_u32_u24", blocks, it);
    run<10>("v_alignbit_b32", blocks, it);
    run<11>("v_add_u32", blocks, it);
    run<12>("v_xor_b32", blocks, it);
    run<13>("v_floor_f64", blocks, it);
    run<14>("v_trunc_f64", blocks, it);
    run<15>("v_cvt_i32_f64", blocks, it);
    run<16>("v_cvt_f64_i32", blocks, it);
    run<17>("v_mul_f64", blocks, it);
    run<18>("v_add_f64", blocks, it);
    run<19>("v_max_f64", blocks, it);
    run<20>("v_cvt_pk_i16_i32", blocks, it);
    run<21>("v_pk_add_i16", blocks, it);
    run<22>("v_perm_b32", blocks, it);
    run<23>("v_mov_b32_dpp", blocks, it);
    run<24>("v_max3_i32", blocks, it);
    run<25>("v_bfe_i32", blocks, it);
    run<26>("v_cmp_f64+cndmask", blocks, it);
    run<27>("v_dot2_i32_i16", blocks, it);
    return 0;
}
