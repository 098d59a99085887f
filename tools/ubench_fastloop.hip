// Microbenchmark: the inner loop of k_bank_render reduced to its common case (polynomial Harmonics voice, folded
// gains, tile on one table piece), in isolation -- how fast can one (wave, voice) iteration go on gfx950 when the
// general kernel's flag tests, cold paths and register pressure are taken away?  Variants:
//   MODE 0: record (184 B) through scalar loads, like the production kernel
//   MODE 1: same, next record's loads issued before this voice's arithmetic (software prefetch into SGPRs)
//   MODE 2: no record loads at all (coefficients loop-invariant): the pure arithmetic floor
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I synthesizer_amd/csrc tools/ubench_fastloop.hip -o tools/ubench_fastloop.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "devmath.hpp"

struct alignas(64) FastRec {
    double t_base, dt, gl, gr, rot_c, rot_s;
    double poly[16];
    double pad[2];
};
static_assert(sizeof(FastRec) == 192, "FastRec");

#define AS4 __attribute__((address_space(4)))

template <int FPL, int MODE, int WAVES, int MINW, int EXTRA = 0>
__global__ __launch_bounds__(WAVES * 64, MINW) void k(const FastRec* __restrict__ recs, const shm::sc_pair* __restrict__ trig_g,
                                                      uint32_t nvoices, uint32_t vpg, uint32_t nframes, double2* __restrict__ parts) {
    __shared__ double red[WAVES][2][64 * FPL];
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    __shared__ double extra[EXTRA ? EXTRA : 1];           // occupancy limiter
    if (EXTRA && nframes == 0xFFFFFFFFu) extra[threadIdx.x] = 1.0;
    for (uint32_t k2 = threadIdx.x; k2 < shm::TRIG_N; k2 += WAVES * 64) trig[k2] = trig_g[k2];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile0 = blockIdx.x * (64 * FPL);
    const uint32_t v0 = blockIdx.y * vpg;
    uint32_t v1 = v0 + vpg;
    if (v1 > nvoices) v1 = nvoices;
    double di0 = (double)(tile0 + lane);
    double accl[FPL], accr[FPL];
#pragma unroll
    for (int j = 0; j < FPL; ++j) { accl[j] = 0.0; accr[j] = 0.0; }
    const FastRec AS4* rp = (const FastRec AS4*)(recs + v0 + wave);
    for (uint32_t vi = v0 + wave; vi < v1; vi += WAVES, rp += WAVES) {
        const FastRec AS4* q = MODE == 2 ? (const FastRec AS4*)recs : rp;
        const double t_base = q->t_base, dt = q->dt, gl = q->gl, gr = q->gr, rc = q->rot_c, rs = q->rot_s;
        double poly[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) poly[u] = q->poly[u];
        if constexpr (MODE == 4) {
            // mixed precision: float64 phase, lookup, rotation and recurrence; the harmonic series by
            // CLENSHAW IN FLOAT32, two frames per packed instruction (v_pk_fma_f32); accumulation in float64.
            // q->poly[k] here = a_k (series amplitudes), used as float.
            typedef float v2f __attribute__((ext_vector_type(2)));
            const double k2 = q->pad[0];
            float af[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) af[u] = (float)poly[u];
            double s0, c0, s1, c1;
            shm::sincos_tab(fma(di0, dt, t_base), trig, s0, c0);
            s1 = fma(s0, rc, c0 * rs);
            c1 = fma(c0, rc, -(s0 * rs));
#pragma unroll
            for (int h = 0; h < FPL; h += 2) {
                const v2f c2 = {(float)(c0 + c0), (float)(c1 + c1)};
                const v2f sv = {(float)s0, (float)s1};
                v2f b1 = {0.f, 0.f}, b2 = {0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const v2f ak = {af[u], af[u]};
                    const v2f bn = __builtin_elementwise_fma(c2, b1, ak - b2);
                    b2 = b1;
                    b1 = bn;
                }
                const v2f x = b1 * sv;
                accl[h] = fma(gl, (double)x[0], accl[h]); accr[h] = fma(gr, (double)x[0], accr[h]);
                accl[h + 1] = fma(gl, (double)x[1], accl[h + 1]); accr[h + 1] = fma(gr, (double)x[1], accr[h + 1]);
                if (h + 2 < FPL) {
                    const double s2 = fma(k2, s1, -s0), c2d = fma(k2, c1, -c0);
                    const double s3 = fma(k2, s2, -s1), c3 = fma(k2, c2d, -c1);
                    s0 = s2; c0 = c2d; s1 = s3; c1 = c3;
                }
            }
            continue;
        }
        if constexpr (MODE == 5) {
            // the same with float32 accumulators (packed), summed to float64 once per voice loop
            typedef float v2f __attribute__((ext_vector_type(2)));
            const double k2 = q->pad[0];
            float af[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) af[u] = (float)poly[u];
            const float glf = (float)gl, grf = (float)gr;
            double s0, c0, s1, c1;
            shm::sincos_tab(fma(di0, dt, t_base), trig, s0, c0);
            s1 = fma(s0, rc, c0 * rs);
            c1 = fma(c0, rc, -(s0 * rs));
#pragma unroll
            for (int h = 0; h < FPL; h += 2) {
                const v2f c2 = {(float)(c0 + c0), (float)(c1 + c1)};
                const v2f sv = {(float)s0, (float)s1};
                v2f b1 = {0.f, 0.f}, b2 = {0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const v2f ak = {af[u], af[u]};
                    const v2f bn = __builtin_elementwise_fma(c2, b1, ak - b2);
                    b2 = b1;
                    b1 = bn;
                }
                const v2f x = b1 * sv;
                v2f al = {(float)accl[h], (float)accl[h + 1]}, ar = {(float)accr[h], (float)accr[h + 1]};
                al = __builtin_elementwise_fma((v2f){glf, glf}, x, al);
                ar = __builtin_elementwise_fma((v2f){grf, grf}, x, ar);
                accl[h] = al[0]; accl[h + 1] = al[1]; accr[h] = ar[0]; accr[h + 1] = ar[1];
                if (h + 2 < FPL) {
                    const double s2 = fma(k2, s1, -s0), c2d = fma(k2, c1, -c0);
                    const double s3 = fma(k2, s2, -s1), c3 = fma(k2, c2d, -c1);
                    s0 = s2; c0 = c2d; s1 = s3; c1 = c3;
                }
            }
            continue;
        }
        if constexpr (MODE == 3) {
            // frames j >= 2 by the three-term recurrence x[j] = 2cos(64dt) x[j-1] - x[j-2] (one FMA per value instead of a
            // two-FMA + two-MUL rotation); pad[0] holds 2cos(64dt)
            const double k2 = q->pad[0];
            double s0, c0, s1, c1;
            shm::sincos_tab(fma(di0, dt, t_base), trig, s0, c0);
            s1 = fma(s0, rc, c0 * rs);
            c1 = fma(c0, rc, -(s0 * rs));
#pragma unroll
            for (int h = 0; h < FPL; h += 2) {
                double p0 = fma(poly[0], c0, poly[1]), p1 = fma(poly[0], c1, poly[1]);
#pragma unroll
                for (int u = 2; u < 16; ++u) { p0 = fma(p0, c0, poly[u]); p1 = fma(p1, c1, poly[u]); }
                const double x0 = p0 * s0, x1 = p1 * s1;
                accl[h] = fma(gl, x0, accl[h]); accr[h] = fma(gr, x0, accr[h]);
                accl[h + 1] = fma(gl, x1, accl[h + 1]); accr[h + 1] = fma(gr, x1, accr[h + 1]);
                if (h + 2 < FPL) {
                    const double s2 = fma(k2, s1, -s0), c2 = fma(k2, c1, -c0);
                    const double s3 = fma(k2, s2, -s1), c3 = fma(k2, c2, -c1);
                    s0 = s2; c0 = c2; s1 = s3; c1 = c3;
                }
            }
            continue;
        }
        double sn[FPL], cs[FPL], pv[FPL];
        shm::sincos_tab(fma(di0, dt, t_base), trig, sn[0], cs[0]);
#pragma unroll
        for (int j = 1; j < FPL; ++j) {
            sn[j] = fma(sn[j - 1], rc, cs[j - 1] * rs);
            cs[j] = fma(cs[j - 1], rc, -(sn[j - 1] * rs));
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) pv[j] = fma(poly[0], cs[j], poly[1]);
#pragma unroll
        for (int u = 2; u < 16; ++u) {
#pragma unroll
            for (int j = 0; j < FPL; ++j) pv[j] = fma(pv[j], cs[j], poly[u]);
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const double x = pv[j] * sn[j];
            accl[j] = fma(gl, x, accl[j]);
            accr[j] = fma(gr, x, accr[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        red[wave][0][j * 64 + lane] = accl[j];
        red[wave][1][j * 64 + lane] = accr[j];
    }
    __syncthreads();
    if (wave < FPL) {
        const uint32_t f = wave * 64 + lane;
        const uint32_t raw = tile0 + f;
        if (raw < nframes) {
            double l = red[0][0][f], r = red[0][1][f];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) { l += red[w][0][f]; r += red[w][1][f]; }
            parts[(size_t)blockIdx.y * nframes + raw] = make_double2(l, r);
        }
    }
}

template <int FPL, int MODE, int WAVES, int MINW, int EXTRA = 0>
void run(const char* name, const FastRec* d_recs, const shm::sc_pair* d_trig, double2* d_parts, uint32_t nvoices, uint32_t nframes, uint32_t groups) {
    const uint32_t tiles = (nframes + 64 * FPL - 1) / (64 * FPL);
    const uint32_t vpg = (nvoices + groups - 1) / groups;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it)
            hipLaunchKernelGGL((k<FPL, MODE, WAVES, MINW, EXTRA>), dim3(tiles, groups), dim3(WAVES * 64), 0, 0, d_recs, d_trig, nvoices, vpg, nframes, d_parts);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    const double per_vs = (17.0 + 4.0 * (FPL - 1)) / FPL + 1.0 / FPL + 16.0 + 3.0;      // float64 ops per voice-sample
    const double fl = (double)nvoices * nframes * per_vs / (best * 1e-3) / 1e12;
    printf("%-34s groups %2u: %.1f us, %.0f G voice-samples/s, %.1f T f64 lane-ops/s (%.0f%% of 39.3)\n", name, groups, best * 1e3,
           (double)nvoices * nframes / (best * 1e-3) / 1e9, fl, fl / 39.3 * 100);
}

int main() {
    const uint32_t nvoices = 1024, nframes = 48000;
    std::vector<FastRec> recs(nvoices);
    srand(1);
    for (auto& r : recs) {
        r.t_base = (rand() % 1000) * 0.01;
        r.dt = 0.01 + (rand() % 1000) * 1e-4;
        r.gl = 0.01; r.gr = 0.02;
        r.rot_c = cos(64 * r.dt); r.rot_s = sin(64 * r.dt);
        for (int u = 0; u < 16; ++u) r.poly[u] = (rand() % 2000 - 1000) * 1e-3;
        r.pad[0] = 2.0 * cos(64 * r.dt);
    }
    std::vector<shm::sc_pair> trig(shm::TRIG_N);
    for (int k2 = 0; k2 < shm::TRIG_N; ++k2) { trig[k2].s = sin(2 * M_PI * k2 / shm::TRIG_N); trig[k2].c = cos(2 * M_PI * k2 / shm::TRIG_N); }
    FastRec* d_recs; shm::sc_pair* d_trig; double2* d_parts;
    hipMalloc(&d_recs, sizeof(FastRec) * nvoices);
    hipMalloc(&d_trig, sizeof(shm::sc_pair) * shm::TRIG_N);
    hipMalloc(&d_parts, sizeof(double2) * nframes * 64 * 4);
    hipMemcpy(d_recs, recs.data(), sizeof(FastRec) * nvoices, hipMemcpyHostToDevice);
    hipMemcpy(d_trig, trig.data(), sizeof(shm::sc_pair) * shm::TRIG_N, hipMemcpyHostToDevice);
    // sustained load: does the clock hold?  2000 launches back to back
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const uint32_t tiles = (nframes + 255) / 256, vpg = 128;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 3000; ++it)
                hipLaunchKernelGGL((k<4, 0, 8, 4, 5120>), dim3(tiles, 8), dim3(512), 0, 0, d_recs, d_trig, nvoices, vpg, nframes, d_parts);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("sustained 2000 launches: %.1f us per launch\n", ms / 2000 * 1e3);
        }
    }
    // the same launches alternating between two streams: how much of a launch is tail + launch gap that a second,
    // independent launch could fill?  (production launches depend on their predecessor: records, partial buses)
    {
        hipStream_t s2[4]; for (int q = 0; q < 4; ++q) hipStreamCreate(&s2[q]);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const uint32_t tiles = (nframes + 255) / 256, vpg = 128;
        for (int nstreams = 1; nstreams <= 4; ++nstreams) {
            for (int rep = 0; rep < 3; ++rep) {
                hipDeviceSynchronize();
                hipEventRecord(e0, s2[0]);
                for (int it = 0; it < 2000; ++it)
                    hipLaunchKernelGGL((k<4, 0, 4, 4>), dim3(tiles, 8), dim3(256), 0, s2[it % nstreams], d_recs, d_trig, nvoices, vpg, nframes,
                                       d_parts + (size_t)(it % nstreams) * nframes * 8);
                hipDeviceSynchronize();
                hipEventRecord(e1, s2[0]); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("production shape, %d stream(s), 2000 launches: %.1f us per launch\n", nstreams, ms / 2000 * 1e3);
            }
        }
    }
    for (uint32_t groups : {8u}) {
        run<8, 4, 4, 4>("FPL8 f32 pk Clenshaw, f64 acc, g8", d_recs, d_trig, d_parts, nvoices, nframes, 8);
        run<8, 4, 4, 5>("FPL8 f32 pk Clenshaw, f64 acc, min5", d_recs, d_trig, d_parts, nvoices, nframes, 8);
        run<16, 4, 4, 4>("FPL16 f32 pk Clenshaw, f64 acc, g16", d_recs, d_trig, d_parts, nvoices, nframes, 16);
        run<8, 5, 4, 4>("FPL8 f32 pk Clenshaw, f32 acc (cvt), g8", d_recs, d_trig, d_parts, nvoices, nframes, 8);
        run<4, 3, 4, 4>("FPL4 recurrence 4w min4", d_recs, d_trig, d_parts, nvoices, nframes, 8);
        run<8, 3, 4, 4>("FPL8 recurrence 4w min4 g16", d_recs, d_trig, d_parts, nvoices, nframes, 16);
        run<8, 3, 4, 5>("FPL8 recurrence 4w min5 g16", d_recs, d_trig, d_parts, nvoices, nframes, 16);
        run<8, 3, 8, 4>("FPL8 recurrence 8w min4 g8", d_recs, d_trig, d_parts, nvoices, nframes, 8);
        run<8, 3, 4, 4>("FPL8 recurrence 4w min4 g8", d_recs, d_trig, d_parts, nvoices, nframes, 8);
        run<16, 3, 4, 4>("FPL16 recurrence 4w min4 g16", d_recs, d_trig, d_parts, nvoices, nframes, 16);
        run<16, 3, 4, 3>("FPL16 recurrence 4w min3 g32", d_recs, d_trig, d_parts, nvoices, nframes, 32);
        run<4, 0, 8, 4>("FPL4 s_load rec, 8w min4", d_recs, d_trig, d_parts, nvoices, nframes, groups);
        run<4, 0, 8, 6>("FPL4 s_load rec, 8w min6", d_recs, d_trig, d_parts, nvoices, nframes, groups);
        run<4, 0, 8, 8>("FPL4 s_load rec, 8w min8", d_recs, d_trig, d_parts, nvoices, nframes, groups);
        run<4, 2, 8, 6>("FPL4 no loads,   8w min6", d_recs, d_trig, d_parts, nvoices, nframes, groups);
        run<2, 0, 8, 6>("FPL2 s_load rec, 8w min6", d_recs, d_trig, d_parts, nvoices, nframes, groups);
        run<2, 2, 8, 6>("FPL2 no loads,   8w min6", d_recs, d_trig, d_parts, nvoices, nframes, groups);
        run<8, 0, 8, 4>("FPL8 s_load rec, 8w min4", d_recs, d_trig, d_parts, nvoices, nframes, groups);
        run<4, 0, 4, 6>("FPL4 s_load rec, 4w min6", d_recs, d_trig, d_parts, nvoices, nframes, groups);
        run<4, 0, 8, 4, 5120>("FPL4 8w, LDS 80K: 2 blocks/CU", d_recs, d_trig, d_parts, nvoices, nframes, groups);
        run<4, 0, 4, 4>("FPL4 4w min4 (production shape)", d_recs, d_trig, d_parts, nvoices, nframes, 8);
        run<8, 0, 4, 4>("FPL8 4w min4 groups 16", d_recs, d_trig, d_parts, nvoices, nframes, 16);
        run<8, 0, 4, 2>("FPL8 4w min2 groups 16", d_recs, d_trig, d_parts, nvoices, nframes, 16);
        run<8, 0, 4, 4>("FPL8 4w min4 groups 8", d_recs, d_trig, d_parts, nvoices, nframes, 8);
        run<6, 0, 4, 4>("FPL6 4w min4 groups 8", d_recs, d_trig, d_parts, nvoices, nframes, 8);
        run<6, 0, 4, 4>("FPL6 4w min4 groups 16", d_recs, d_trig, d_parts, nvoices, nframes, 16);
        run<4, 0, 8, 4, 1536>("FPL4 8w, LDS 52K: 3 blocks/CU", d_recs, d_trig, d_parts, nvoices, nframes, groups);
    }
    return 0;
}
