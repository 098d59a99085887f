// osc_mixbus.hip -- the mixer sum bus over materialised float32 voices (HBM-bound: 4N+8 bytes per frame).
#include "osc_host.hpp"

namespace {

// ---- mixer over materialised float32 voices ------------------------------------------
// block = W waves; a wave owns 256 consecutive frames (float4 per lane) and a strided subset of
// the voice rows of its group; partial (L,R) x4 per lane are summed across the block's waves in
// LDS.  grid = (frame tiles, voice groups); groups > 1 write partial buses that k_bus_sum folds.
template <bool NT>
__device__ __forceinline__ float4 ldf4(const float* p) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v t = sh::load_vec<NT, f4v>(p);
    return make_float4(t[0], t[1], t[2], t[3]);
}

template <int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_mix_bus_f32(const float* __restrict__ voices, uint32_t nvoices,
                                                            size_t stride, uint32_t nframes,
                                                            const float2* __restrict__ gains,
                                                            uint32_t voices_per_group,
                                                            float2* __restrict__ out, size_t out_group_stride) {
    __shared__ float red[WAVES][8][64];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t f0 = blockIdx.x * 256 + lane * 4;
    const uint32_t g = blockIdx.y;
    const uint32_t v_begin = g * voices_per_group;
    uint32_t v_end = v_begin + voices_per_group;
    if (v_end > nvoices) v_end = nvoices;
    float l0 = 0, l1 = 0, l2 = 0, l3 = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    const bool full = (f0 + 3 < nframes) && ((stride & 3) == 0);
    if (full) {
        uint32_t v = v_begin + wave;
#define SH_ACC(X_, G_)                                                     \
    l0 = fmaf(G_.x, X_.x, l0); l1 = fmaf(G_.x, X_.y, l1); l2 = fmaf(G_.x, X_.z, l2); l3 = fmaf(G_.x, X_.w, l3); \
    r0 = fmaf(G_.y, X_.x, r0); r1 = fmaf(G_.y, X_.y, r1); r2 = fmaf(G_.y, X_.z, r2); r3 = fmaf(G_.y, X_.w, r3);
        // 8 rows (8 KB per wave) in flight, then 4, then 1
        for (; v + 7 * WAVES < v_end; v += 8 * WAVES) {
            float4 x[8];
            float2 gg[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = ldf4<NT>(voices + (size_t)(v + k * WAVES) * stride + f0);
#pragma unroll
            for (int k = 0; k < 8; ++k) gg[k] = gains[v + k * WAVES];
#pragma unroll
            for (int k = 0; k < 8; ++k) { SH_ACC(x[k], gg[k]) }
        }
        for (; v + 3 * WAVES < v_end; v += 4 * WAVES) {
            float4 x0 = ldf4<NT>(voices + (size_t)v * stride + f0);
            float4 x1 = ldf4<NT>(voices + (size_t)(v + WAVES) * stride + f0);
            float4 x2 = ldf4<NT>(voices + (size_t)(v + 2 * WAVES) * stride + f0);
            float4 x3 = ldf4<NT>(voices + (size_t)(v + 3 * WAVES) * stride + f0);
            float2 g0 = gains[v], g1 = gains[v + WAVES], g2 = gains[v + 2 * WAVES], g3 = gains[v + 3 * WAVES];
            SH_ACC(x0, g0) SH_ACC(x1, g1) SH_ACC(x2, g2) SH_ACC(x3, g3)
        }
        for (; v < v_end; v += WAVES) {
            float4 x0 = ldf4<NT>(voices + (size_t)v * stride + f0);
            float2 g0 = gains[v];
            SH_ACC(x0, g0)
        }
    } else if (f0 < nframes) {
        for (uint32_t v = v_begin + wave; v < v_end; v += WAVES) {
            const float* row = voices + (size_t)v * stride;
            float2 gg = gains[v];
            float4 x;
            x.x = row[f0];
            x.y = (f0 + 1 < nframes) ? row[f0 + 1] : 0.f;
            x.z = (f0 + 2 < nframes) ? row[f0 + 2] : 0.f;
            x.w = (f0 + 3 < nframes) ? row[f0 + 3] : 0.f;
            SH_ACC(x, gg)
        }
    }
#undef SH_ACC
    red[wave][0][lane] = l0; red[wave][1][lane] = r0; red[wave][2][lane] = l1; red[wave][3][lane] = r1;
    red[wave][4][lane] = l2; red[wave][5][lane] = r2; red[wave][6][lane] = l3; red[wave][7][lane] = r3;
    __syncthreads();
    if (wave == 0 && f0 < nframes) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = red[0][j][lane];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) s += red[w][j][lane];
            acc[j] = s;
        }
        float2* o = out + (size_t)g * out_group_stride + f0;
        if (f0 + 3 < nframes) {
            // 4 frames x (L,R) = 32 contiguous bytes
            reinterpret_cast<float4*>(o)[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            reinterpret_cast<float4*>(o)[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        } else {
            for (uint32_t j = 0; j < 4 && f0 + j < nframes; ++j) o[j] = make_float2(acc[2 * j], acc[2 * j + 1]);
        }
    }
}

// Long buffers (the frame range alone fills the chip): no voice split -- a lane walks all the rows for its 4 frames,
// a workgroup covers WAVES KB of every row it visits, no LDS.  (Same lesson as the integer fold, DESIGN.md section 4
// item 15: eight waves fetching one 1 KB column of eight distant rows cost 10 % of the bandwidth.)
template <int WAVES, int INFLIGHT, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_mix_bus_direct(const float* __restrict__ voices, uint32_t nvoices, size_t stride,
                                                               uint32_t nframes, const float2* __restrict__ gains,
                                                               float2* __restrict__ out) {
    const uint32_t f0 = (blockIdx.x * (WAVES * 64) + threadIdx.x) * 4;
    if (f0 + 3 >= nframes) {
        for (uint32_t j = 0; f0 + j < nframes; ++j) {
            float l = 0.f, r = 0.f;
            for (uint32_t v = 0; v < nvoices; ++v) {
                const float x = voices[(size_t)v * stride + f0 + j];
                const float2 g = gains[v];
                l = fmaf(g.x, x, l);
                r = fmaf(g.y, x, r);
            }
            out[f0 + j] = make_float2(l, r);
        }
        return;
    }
    float l0 = 0, l1 = 0, l2 = 0, l3 = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    const float* col = voices + f0;
    uint32_t v = 0;
    for (; v + INFLIGHT <= nvoices; v += INFLIGHT) {
        float4 x[INFLIGHT];
        float2 gg[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) x[k] = ldf4<NT>(col + (size_t)(v + k) * stride);       // NT: 0.81 -> 0.90 of the HBM peak on 2 GB
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) gg[k] = gains[v + k];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            l0 = fmaf(gg[k].x, x[k].x, l0); l1 = fmaf(gg[k].x, x[k].y, l1); l2 = fmaf(gg[k].x, x[k].z, l2); l3 = fmaf(gg[k].x, x[k].w, l3);
            r0 = fmaf(gg[k].y, x[k].x, r0); r1 = fmaf(gg[k].y, x[k].y, r1); r2 = fmaf(gg[k].y, x[k].z, r2); r3 = fmaf(gg[k].y, x[k].w, r3);
        }
    }
    for (; v < nvoices; ++v) {
        const float4 x = *reinterpret_cast<const float4*>(col + (size_t)v * stride);
        const float2 g = gains[v];
        l0 = fmaf(g.x, x.x, l0); l1 = fmaf(g.x, x.y, l1); l2 = fmaf(g.x, x.z, l2); l3 = fmaf(g.x, x.w, l3);
        r0 = fmaf(g.y, x.x, r0); r1 = fmaf(g.y, x.y, r1); r2 = fmaf(g.y, x.z, r2); r3 = fmaf(g.y, x.w, r3);
    }
    float2* o = out + f0;
    reinterpret_cast<float4*>(o)[0] = make_float4(l0, r0, l1, r1);
    reinterpret_cast<float4*>(o)[1] = make_float4(l2, r2, l3, r3);
}

__global__ void k_bus_sum(const float2* __restrict__ parts, uint32_t ngroups, size_t group_stride,
                          uint32_t nframes, float2* __restrict__ out) {
    const size_t i = sh::block_id() * blockDim.x + threadIdx.x;
    if (i >= nframes) return;
    float2 s = parts[i];
    for (uint32_t g = 1; g < ngroups; ++g) {
        float2 p = parts[(size_t)g * group_stride + i];
        s.x += p.x;
        s.y += p.y;
    }
    out[i] = s;
}

__global__ void k_bus_finalize(const double* __restrict__ in, size_t n, float* __restrict__ out) {
    size_t i = sh::block_id() * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

}  // namespace

namespace sh {
int bus_finalize_on(hipStream_t st, const double* in, size_t nvalues, float* out) {
    if (!nvalues) return SH_OK;
    hipLaunchKernelGGL(k_bus_finalize, sh::grid1d(nvalues, 256), dim3(256), 0, st, in, nvalues, out);
    SH_CHECK_LAUNCH("k_bus_finalize");
    return SH_OK;
}
}  // namespace sh

extern "C" {

int sh_mix_bus_f32(const sh_buf* voices, uint32_t nvoices, size_t stride, uint32_t nframes,
                   const sh_buf* gains_lr, sh_buf* bus_f32) {
    SH_REQUIRE_INIT();
    if (!voices || !gains_lr || !bus_f32 || nvoices == 0) return sh::set_error(SH_ERR_INVALID, "sh_mix_bus_f32: NULL argument");
    if (nframes == 0) return SH_OK;
    if (stride < nframes || voices->bytes / 4 < (size_t)(nvoices - 1) * stride + nframes)
        return sh::set_error(SH_ERR_INVALID, "sh_mix_bus_f32: voice buffer too small for %u x %u (stride %zu)", nvoices, nframes, stride);
    if (bus_f32->bytes < (size_t)nframes * 8) return sh::set_error(SH_ERR_INVALID, "sh_mix_bus_f32: bus too small");
    constexpr uint32_t MIX_MAX_FRAMES = 1u << 24;            // per launch: tiles x 512 threads must stay below 2^32 work-items
    if (nframes > MIX_MAX_FRAMES) {
        for (uint64_t off = 0; off < nframes; off += MIX_MAX_FRAMES) {
            const uint32_t n = nframes - off < MIX_MAX_FRAMES ? (uint32_t)(nframes - off) : MIX_MAX_FRAMES;
            sh_buf v{(char*)voices->ptr + off * 4, voices->bytes - off * 4, false, 0};
            sh_buf o{(char*)bus_f32->ptr + off * 8, bus_f32->bytes - off * 8, false, 0};
            const int rc = sh_mix_bus_f32(&v, nvoices, stride, n, gains_lr, &o);
            if (rc) return rc;
        }
        return SH_OK;
    }
    constexpr int W = 8;
    const uint32_t tiles = sh::div_up(nframes, 256);
    // enough workgroups to cover 256 CUs several times over: split the voices into groups when the
    // frame range alone gives too few tiles
    uint32_t groups = 1;
    while (tiles * groups < 1024 && nvoices / (groups * 2) >= 4 * W) groups *= 2;
    const uint32_t vpg = (nvoices + groups - 1) / groups;
    if (gains_lr->bytes < (size_t)nvoices * 8) return sh::set_error(SH_ERR_INVALID, "sh_mix_bus_f32: gains buffer too small");
    size_t part_bytes = groups > 1 ? (size_t)groups * nframes * 8 : 0;
    int rc = sh::ensure_scratch(part_bytes);
    if (rc) return rc;
    hipStream_t st = sh::state().stream;
    const bool stream = (size_t)nvoices * nframes * 4 > sh::STREAM_BYTES;              // rows beyond the Infinity Cache: streaming loads
    if (tiles >= 1536 && (stride & 3) == 0 && ((uintptr_t)voices->ptr & 15) == 0 && ((uintptr_t)bus_f32->ptr & 15) == 0) {
        if (stream) hipLaunchKernelGGL((k_mix_bus_direct<8, 4, true>), sh::grid1d(nframes, 256 * 8), dim3(8 * 64), 0, st,
                                       (const float*)voices->ptr, nvoices, stride, nframes, (const float2*)gains_lr->ptr, (float2*)bus_f32->ptr);
        else hipLaunchKernelGGL((k_mix_bus_direct<8, 4, false>), sh::grid1d(nframes, 256 * 8), dim3(8 * 64), 0, st,
                                (const float*)voices->ptr, nvoices, stride, nframes, (const float2*)gains_lr->ptr, (float2*)bus_f32->ptr);
        SH_CHECK_LAUNCH("k_mix_bus_direct");
        return SH_OK;
    }
    float2* parts = (float2*)sh::state().scratch;
    float2* dst = groups > 1 ? parts : (float2*)bus_f32->ptr;
    hipLaunchKernelGGL((k_mix_bus_f32<W, false>), dim3(tiles, groups), dim3(W * 64), 0, st,       // (plain loads: streaming ones gain nothing here)
                       (const float*)voices->ptr, nvoices, stride, nframes, (const float2*)gains_lr->ptr, vpg,
                       dst, (size_t)nframes);
    SH_CHECK_LAUNCH("k_mix_bus_f32");
    if (groups > 1) {
        hipLaunchKernelGGL(k_bus_sum, sh::grid1d(nframes, 256), dim3(256), 0, st,
                           (const float2*)parts, groups, (size_t)nframes, nframes, (float2*)bus_f32->ptr);
        SH_CHECK_LAUNCH("k_bus_sum");
    }
    return SH_OK;
}

int sh_bus_finalize(const sh_buf* bus_f64, size_t nvalues, sh_buf* bus_f32) {
    SH_REQUIRE_INIT();
    if (!bus_f64 || !bus_f32) return sh::set_error(SH_ERR_INVALID, "sh_bus_finalize: NULL argument");
    if (bus_f64->bytes < nvalues * 8 || bus_f32->bytes < nvalues * 4) return sh::set_error(SH_ERR_INVALID, "sh_bus_finalize: buffer too small");
    return sh::bus_finalize_on(sh::state().stream, (const double*)bus_f64->ptr, nvalues, (float*)bus_f32->ptr);
}

}  // extern "C"
