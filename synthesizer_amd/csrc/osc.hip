// osc.hip -- oscillator bank kernels for gfx950.
//
// One thread per output sample.  The reference's per-voice generator state
// (oscillators.py: running `t`, FM `phase_correction`, EnvelopeFilter `time`/`amp`) is
// replaced by closed forms of the sample index:
//   * non-FM phase: exact piecewise-linear table of the float64 accumulation (sh_segment)
//   * FM phase:     theta_n = phase0 + f*T_n + f*inc*L(n), T = accumulated time table,
//                   L = running sum of the LFO (closed form for a Sine LFO, or a scanned buffer)
//   * envelope:     integer sample boundaries + slopes prepared by the host
// Arithmetic is float64 up to the final store (one rounding to float32), built with
// -ffp-contract=off so that a*b+c is fused only where fma() is written.
//
// Kernels
//   k_locate        one thread per voice: find the table piece that holds `start`
//   k_generate      grid (frame tiles, voices): voice-major float32 PCM in HBM (4 B per voice-sample)
//   k_bank_render   fused generate-and-mix: block = 64 frames x W waves, each wave walks a strided
//                   subset of the voices with the voice record in SGPRs, float64 partial (L, R)
//                   per lane, LDS-staged sum across the W waves, one float2 store per frame
//   k_mix_bus_f32   HBM-bound mixer over materialised voices (4N+8 B per frame)
#include "common.hpp"
#include "devmath.hpp"
#include <new>
#include <vector>

namespace {

struct Located {           // where `start` falls in a voice's phase (or time) table
    double   t_base;       // table value at `start`
    double   dt;           // increment of that piece
    uint32_t remain;       // frames from `start` that stay on the piece (saturated)
    uint32_t seg;          // piece index inside the voice's table
};

struct BankPtrs {
    const sh_voice*   voices;
    const sh_segment* segs;
    const double*     coefs;
    const sh_partial* partials;
};

__device__ __forceinline__ double table_value(const sh_segment* __restrict__ tab, uint32_t cnt,
                                              const Located& L, uint64_t start, uint32_t i) {
    if (i < L.remain) return fma((double)i, L.dt, L.t_base);          // exact: stays on the piece
    uint64_t n = start + i;                                            // rare: the tile straddles a binade
    uint32_t s = L.seg;
    while (s + 1 < cnt && tab[s + 1].n0 <= n) ++s;
    return fma((double)(n - tab[s].n0), tab[s].dt, tab[s].t0);
}

__device__ __forceinline__ double env_gain(const sh_envelope& e, uint64_t n) {
    if (n < e.n_attack_end) return (double)n * e.attack_slope;
    if (n < e.n_decay_end) return fma((double)(n - e.n_attack_end), e.decay_slope, 1.0);
    if (n < e.n_sustain_end) return e.sustain_level;
    if (n < e.n_release_end) return fma((double)(n - e.n_sustain_end), e.release_slope, e.sustain_level);
    if (n == e.n_release_end && e.has_tail) return e.tail_amp;
    return 0.0;
}

// One sample of one voice, float64.  `v` is wave-uniform in every kernel below, so the
// branches on kind / fm_mode do not diverge.
__device__ __forceinline__ double voice_sample(const sh_voice& v, const BankPtrs& B, const Located& L,
                                               uint64_t start, uint32_t i,
                                               const double* __restrict__ fm_cumsum,
                                               const double* __restrict__ pwm) {
    double theta;
    if (v.fm_mode == SH_FM_NONE) {
        theta = table_value(B.segs + v.seg_offset, v.seg_count, L, start, i);
    } else {
        double T = table_value(B.segs + v.time_seg_offset, v.time_seg_count, L, start, i);
        double Ln;
        if (v.fm_mode == SH_FM_SINE) {
            double n = (double)(start + i);
            double arg = fma(n - 0.5, v.lfo_d, v.lfo_a);
            Ln = fma(v.lfo_K, v.lfo_C0 - shm::cos_f64(arg), v.lfo_bias * n);
        } else {
            Ln = fm_cumsum[i];
        }
        theta = fma(v.frequency * v.fm_inc, Ln, fma(v.frequency, T, v.fm_phase0));
    }
    double val;
    switch (v.kind) {
    case SH_SINE:
        val = shm::sin_f64(theta) * v.amplitude + v.bias;
        break;
    case SH_SAWTOOTH:
        val = shm::saw_value(theta, v.amplitude * 2.0, v.bias);
        break;
    case SH_SQUARE:
        val = shm::square_value(theta, v.amplitude, v.bias);
        break;
    case SH_PULSE:
        val = shm::pulse_value(theta, pwm ? pwm[i] : v.pulsewidth, v.amplitude, v.bias);
        break;
    default: {   // SH_HARMONICS
        double h;
        if (v.harm_dense) {
            // sum_k a_k sin(k*theta) by Clenshaw: b_k = a_k + 2cos(theta) b_{k+1} - b_{k+2}; sum = b_1 sin(theta)
            double s, c;
            shm::sincos_f64(theta, s, c);
            const double c2 = c + c;
            const double* __restrict__ a = B.coefs + v.harm_offset;
            double b1 = 0.0, b2 = 0.0;
            for (uint32_t k = 0; k < v.harm_count; k += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    double bn = fma(c2, b1, a[k + u] - b2);
                    b2 = b1;
                    b1 = bn;
                }
            }
            h = b1 * s;
        } else {
            const sh_partial* __restrict__ p = B.partials + v.harm_offset;
            h = 0.0;
            for (uint32_t k = 0; k < v.harm_count; ++k) h += shm::sin_f64(theta * p[k].k) * p[k].amp;
        }
        val = h * v.amplitude + v.bias;
    } break;
    }
    if (v.env.enabled) val = val * env_gain(v.env, start + i);
    return val;
}

__global__ void k_locate(BankPtrs B, uint32_t first, uint32_t nvoices, uint64_t start, Located* __restrict__ out) {
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvoices) return;
    const sh_voice& vo = B.voices[first + v];
    const bool fm = vo.fm_mode != SH_FM_NONE;
    const uint32_t off = fm ? vo.time_seg_offset : vo.seg_offset;
    const uint32_t cnt = fm ? vo.time_seg_count : vo.seg_count;
    const sh_segment* tab = B.segs + off;
    uint32_t lo = 0, hi = cnt - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (tab[mid].n0 <= start) lo = mid; else hi = mid - 1;
    }
    Located L;
    L.t_base = fma((double)(start - tab[lo].n0), tab[lo].dt, tab[lo].t0);
    L.dt = tab[lo].dt;
    uint64_t rem = (lo + 1 < cnt) ? (tab[lo + 1].n0 - start) : 0xFFFFFFFFull;
    L.remain = rem > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rem;
    L.seg = lo;
    out[v] = L;
}

// voice-major materialisation: out[v*stride + i]
__global__ __launch_bounds__(256) void k_generate(BankPtrs B, uint32_t first, const Located* __restrict__ loc,
                                                  uint64_t start, uint32_t n,
                                                  const double* __restrict__ fm_cumsum,
                                                  const double* __restrict__ pwm,
                                                  float* __restrict__ out32, double* __restrict__ out64,
                                                  size_t stride) {
    const uint32_t vi = blockIdx.y;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const sh_voice& v = B.voices[first + vi];
    const Located L = loc[vi];
    double x = voice_sample(v, B, L, start, i, fm_cumsum, pwm);
    if (out32) out32[(size_t)vi * stride + i] = (float)x;
    if (out64) out64[(size_t)vi * stride + i] = x;
}

// fused generate-and-mix
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_bank_render(BankPtrs B, uint32_t nvoices,
                                                            const Located* __restrict__ loc,
                                                            uint64_t start, uint32_t nframes,
                                                            float2* __restrict__ bus32,
                                                            double2* __restrict__ bus64) {
    __shared__ double red[WAVES][2][64];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t i_raw = blockIdx.x * 64 + lane;
    const uint32_t i = i_raw < nframes ? i_raw : nframes - 1;
    double accl = 0.0, accr = 0.0;
    for (uint32_t vi = wave; vi < nvoices; vi += WAVES) {
        const sh_voice& v = B.voices[vi];
        const Located L = loc[vi];
        double x = voice_sample(v, B, L, start, i, nullptr, nullptr);
        accl = fma((double)v.gain_l, x, accl);
        accr = fma((double)v.gain_r, x, accr);
    }
    red[wave][0][lane] = accl;
    red[wave][1][lane] = accr;
    __syncthreads();
    if (wave == 0 && i_raw < nframes) {
        double l = red[0][0][lane], r = red[0][1][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            l += red[w][0][lane];
            r += red[w][1][lane];
        }
        if (bus32) bus32[i_raw] = make_float2((float)l, (float)r);
        if (bus64) bus64[i_raw] = make_double2(l, r);
    }
}

// ---- mixer over materialised float32 voices ------------------------------------------
// block = W waves; a wave owns 256 consecutive frames (float4 per lane) and a strided subset of
// the voice rows of its group; partial (L,R) x4 per lane are summed across the block's waves in
// LDS.  grid = (frame tiles, voice groups); groups > 1 write partial buses that k_bus_sum folds.
template <int WAVES>
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
        // 4 rows in flight per wave
        for (; v + 3 * WAVES < v_end; v += 4 * WAVES) {
            float4 x0 = *reinterpret_cast<const float4*>(voices + (size_t)v * stride + f0);
            float4 x1 = *reinterpret_cast<const float4*>(voices + (size_t)(v + WAVES) * stride + f0);
            float4 x2 = *reinterpret_cast<const float4*>(voices + (size_t)(v + 2 * WAVES) * stride + f0);
            float4 x3 = *reinterpret_cast<const float4*>(voices + (size_t)(v + 3 * WAVES) * stride + f0);
            float2 g0 = gains[v], g1 = gains[v + WAVES], g2 = gains[v + 2 * WAVES], g3 = gains[v + 3 * WAVES];
#define SH_ACC(X_, G_)                                                     \
    l0 = fmaf(G_.x, X_.x, l0); l1 = fmaf(G_.x, X_.y, l1); l2 = fmaf(G_.x, X_.z, l2); l3 = fmaf(G_.x, X_.w, l3); \
    r0 = fmaf(G_.y, X_.x, r0); r1 = fmaf(G_.y, X_.y, r1); r2 = fmaf(G_.y, X_.z, r2); r3 = fmaf(G_.y, X_.w, r3);
            SH_ACC(x0, g0) SH_ACC(x1, g1) SH_ACC(x2, g2) SH_ACC(x3, g3)
        }
        for (; v < v_end; v += WAVES) {
            float4 x0 = *reinterpret_cast<const float4*>(voices + (size_t)v * stride + f0);
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

__global__ void k_bus_sum(const float2* __restrict__ parts, uint32_t ngroups, size_t group_stride,
                          uint32_t nframes, float2* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
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
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// ---- float64 exclusive scan (FM with an arbitrary modulator) ---------------------------
constexpr int SCAN_TILE = 2048;   // values per block (256 threads x 8)

__device__ __forceinline__ double block_exclusive_scan_256(double x, double* sh, double& total) {
    // sh: 256 doubles.  Hillis-Steele; plenty fast for the few MB a modulator block has.
    const int t = threadIdx.x;
    sh[t] = x;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        double add = (t >= o) ? sh[t - o] : 0.0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    total = sh[255];
    double incl = sh[t];
    __syncthreads();
    return incl - x;
}

__global__ __launch_bounds__(256) void k_scan_tile_sums(const double* __restrict__ x, uint32_t n, double* __restrict__ sums) {
    __shared__ double sh[256];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (base + j < n) s += x[base + j];
    double total;
    block_exclusive_scan_256(s, sh, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_scan_sums(double* __restrict__ sums, uint32_t ntiles, double carry_in) {
    // single block: exclusive scan of the tile sums in place; sums[ntiles] = grand total
    __shared__ double sh[256];
    __shared__ double carry;
    if (threadIdx.x == 0) carry = carry_in;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 256) {
        uint32_t i = base + threadIdx.x;
        double v = (i < ntiles) ? sums[i] : 0.0;
        double total;
        double ex = block_exclusive_scan_256(v, sh, total);
        double c = carry;
        if (i < ntiles) sums[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[ntiles] = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const double* __restrict__ x, uint32_t n,
                                                    const double* __restrict__ sums, double* __restrict__ out) {
    __shared__ double sh[256];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    double v[8];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = (base + j < n) ? x[base + j] : 0.0;
        s += v[j];
    }
    double total;
    double ex = block_exclusive_scan_256(s, sh, total) + sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------

struct sh_bank {
    uint32_t    nvoices = 0;
    sh_voice*   d_voices = nullptr;
    sh_segment* d_segs = nullptr;
    double*     d_coefs = nullptr;
    sh_partial* d_partials = nullptr;
    Located*    d_loc = nullptr;
    float2*     d_gains = nullptr;
    uint32_t    nsegs = 0, ncoefs = 0, npartials = 0;
    std::vector<sh_voice> h_voices;    // for validation of per-call arguments
};

static BankPtrs ptrs(const sh_bank* b) {
    BankPtrs p;
    p.voices = b->d_voices;
    p.segs = b->d_segs;
    p.coefs = b->d_coefs;
    p.partials = b->d_partials;
    return p;
}

template <typename T>
static int upload_array(T** dst, const T* src, size_t count, hipStream_t st) {
    *dst = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    SH_HIP(hipMalloc((void**)dst, bytes));
    if (count) SH_HIP(hipMemcpyAsync(*dst, src, count * sizeof(T), hipMemcpyHostToDevice, st));
    return SH_OK;
}

extern "C" {

int sh_bank_create(const sh_voice* voices, uint32_t nvoices, const sh_segment* segs, uint32_t nsegs,
                   const double* coefs, uint32_t ncoefs, const sh_partial* partials, uint32_t npartials,
                   sh_bank** out) {
    SH_REQUIRE_INIT();
    if (!out || !voices || nvoices == 0) return sh::set_error(SH_ERR_INVALID, "sh_bank_create: no voices");
    if (!segs || nsegs == 0) return sh::set_error(SH_ERR_INVALID, "sh_bank_create: no phase tables");
    for (uint32_t i = 0; i < nvoices; ++i) {
        const sh_voice& v = voices[i];
        if (v.kind < SH_SINE || v.kind > SH_HARMONICS)
            return sh::set_error(SH_ERR_INVALID, "voice %u: unknown kind %d", i, v.kind);
        if (v.fm_mode < SH_FM_NONE || v.fm_mode > SH_FM_BUFFER)
            return sh::set_error(SH_ERR_INVALID, "voice %u: unknown fm_mode %d", i, v.fm_mode);
        uint32_t off = v.fm_mode ? v.time_seg_offset : v.seg_offset;
        uint32_t cnt = v.fm_mode ? v.time_seg_count : v.seg_count;
        if (cnt == 0 || off > nsegs || cnt > nsegs - off)
            return sh::set_error(SH_ERR_INVALID, "voice %u: phase table [%u,+%u) outside %u pieces", i, off, cnt, nsegs);
        if (segs[off].n0 != 0) return sh::set_error(SH_ERR_INVALID, "voice %u: phase table does not start at sample 0", i);
        if (v.kind == SH_HARMONICS) {
            uint32_t lim = v.harm_dense ? ncoefs : npartials;
            if (v.harm_offset > lim || v.harm_count > lim - v.harm_offset)
                return sh::set_error(SH_ERR_INVALID, "voice %u: harmonics [%u,+%u) outside table of %u", i, v.harm_offset, v.harm_count, lim);
            if (v.harm_dense && (v.harm_count & 7))
                return sh::set_error(SH_ERR_INVALID, "voice %u: dense harmonic count %u is not a multiple of 8", i, v.harm_count);
        }
    }
    sh_bank* b = new (std::nothrow) sh_bank;
    if (!b) return sh::set_error(SH_ERR_NOMEM, "host allocation failed");
    hipStream_t st = sh::state().stream;
    b->nvoices = nvoices;
    b->nsegs = nsegs;
    b->ncoefs = ncoefs;
    b->npartials = npartials;
    b->h_voices.assign(voices, voices + nvoices);
    std::vector<float2> gains(nvoices);
    for (uint32_t i = 0; i < nvoices; ++i) gains[i] = make_float2(voices[i].gain_l, voices[i].gain_r);
    int rc = upload_array(&b->d_voices, voices, nvoices, st);
    if (!rc) rc = upload_array(&b->d_segs, segs, nsegs, st);
    if (!rc) rc = upload_array(&b->d_coefs, coefs, ncoefs, st);
    if (!rc) rc = upload_array(&b->d_partials, partials, npartials, st);
    if (!rc) rc = upload_array(&b->d_gains, gains.data(), nvoices, st);
    if (!rc) {
        hipError_t e = hipMalloc((void**)&b->d_loc, sizeof(Located) * nvoices);
        if (e != hipSuccess) rc = sh::hip_error(e, "hipMalloc(loc)");
    }
    if (!rc) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = sh::hip_error(e, "upload voice table");
    }
    if (rc) {
        sh_bank_destroy(b);
        return rc;
    }
    *out = b;
    return SH_OK;
}

int sh_bank_destroy(sh_bank* b) {
    if (!b) return SH_OK;
    if (sh::state().initialized) {
        hipStreamSynchronize(sh::state().stream);
        if (b->d_voices) hipFree(b->d_voices);
        if (b->d_segs) hipFree(b->d_segs);
        if (b->d_coefs) hipFree(b->d_coefs);
        if (b->d_partials) hipFree(b->d_partials);
        if (b->d_loc) hipFree(b->d_loc);
        if (b->d_gains) hipFree(b->d_gains);
    }
    delete b;
    return SH_OK;
}

uint32_t sh_bank_nvoices(const sh_bank* b) { return b ? b->nvoices : 0; }

static int locate(sh_bank* b, uint32_t first, uint32_t count, uint64_t start) {
    hipLaunchKernelGGL(k_locate, dim3(sh::div_up(count, 64)), dim3(64), 0, sh::state().stream,
                       ptrs(b), first, count, start, b->d_loc);
    SH_CHECK_LAUNCH("k_locate");
    return SH_OK;
}

int sh_osc_render(sh_bank* bank, uint32_t voice, const sh_buf* fm_cumsum, const sh_buf* pwm,
                  uint64_t start, uint32_t n, float* out_host, sh_buf* out_f32, size_t out_off, sh_buf* out_f64) {
    SH_REQUIRE_INIT();
    if (!bank || voice >= bank->nvoices) return sh::set_error(SH_ERR_INVALID, "sh_osc_render: bad bank/voice");
    if (!out_host && !out_f32 && !out_f64) return sh::set_error(SH_ERR_INVALID, "sh_osc_render: no destination");
    if (n == 0) return SH_OK;
    const sh_voice& v = bank->h_voices[voice];
    if (v.fm_mode == SH_FM_BUFFER && (!fm_cumsum || fm_cumsum->bytes < (size_t)n * 8))
        return sh::set_error(SH_ERR_INVALID, "sh_osc_render: SH_FM_BUFFER voice needs fm_cumsum with >= n doubles");
    if (pwm && pwm->bytes < (size_t)n * 8) return sh::set_error(SH_ERR_INVALID, "sh_osc_render: pwm buffer too small");
    if (out_f32 && (out_off > out_f32->bytes / 4 || n > out_f32->bytes / 4 - out_off))
        return sh::set_error(SH_ERR_INVALID, "sh_osc_render: out_f32 too small");
    if (out_f64 && out_f64->bytes < (size_t)n * 8) return sh::set_error(SH_ERR_INVALID, "sh_osc_render: out_f64 too small");
    float* d32 = out_f32 ? (float*)out_f32->ptr + out_off : nullptr;
    if (!d32 && out_host) {
        int rc = sh::ensure_scratch((size_t)n * 4);
        if (rc) return rc;
        d32 = (float*)sh::state().scratch;
    }
    int rc = locate(bank, voice, 1, start);
    if (rc) return rc;
    hipLaunchKernelGGL(k_generate, dim3(sh::div_up(n, 256), 1), dim3(256), 0, sh::state().stream,
                       ptrs(bank), voice, bank->d_loc, start, n,
                       fm_cumsum ? (const double*)fm_cumsum->ptr : nullptr,
                       pwm ? (const double*)pwm->ptr : nullptr,
                       d32, out_f64 ? (double*)out_f64->ptr : nullptr, (size_t)0);
    SH_CHECK_LAUNCH("k_generate");
    if (out_host) {
        SH_HIP(hipMemcpyAsync(out_host, d32, (size_t)n * 4, hipMemcpyDeviceToHost, sh::state().stream));
        SH_HIP(hipStreamSynchronize(sh::state().stream));
    }
    return SH_OK;
}

static int bank_check_plain(const sh_bank* b, const char* who) {
    for (uint32_t i = 0; i < b->nvoices; ++i)
        if (b->h_voices[i].fm_mode == SH_FM_BUFFER)
            return sh::set_error(SH_ERR_INVALID, "%s: voice %u needs a modulator buffer (SH_FM_BUFFER); render it with sh_osc_render", who, i);
    return SH_OK;
}

int sh_bank_generate(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* voices_out, size_t stride) {
    SH_REQUIRE_INIT();
    if (!b || !voices_out) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate: NULL argument");
    if (nframes == 0) return SH_OK;
    if (stride < nframes) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate: stride < nframes");
    if (voices_out->bytes / 4 < (size_t)(b->nvoices - 1) * stride + nframes)
        return sh::set_error(SH_ERR_INVALID, "sh_bank_generate: output buffer too small");
    int rc = bank_check_plain(b, "sh_bank_generate");
    if (rc) return rc;
    rc = locate(b, 0, b->nvoices, start);
    if (rc) return rc;
    // gridDim.y is limited to 65535 voices per launch
    for (uint32_t first = 0; first < b->nvoices; first += 65535) {
        uint32_t cnt = b->nvoices - first < 65535 ? b->nvoices - first : 65535;
        hipLaunchKernelGGL(k_generate, dim3(sh::div_up(nframes, 256), cnt), dim3(256), 0, sh::state().stream,
                           ptrs(b), first, b->d_loc + first, start, nframes, (const double*)nullptr, (const double*)nullptr,
                           (float*)voices_out->ptr + (size_t)first * stride, (double*)nullptr, stride);
        SH_CHECK_LAUNCH("k_generate");
    }
    return SH_OK;
}

int sh_bank_render(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* bus_f32, sh_buf* bus_f64) {
    SH_REQUIRE_INIT();
    if (!b || (!bus_f32 && !bus_f64)) return sh::set_error(SH_ERR_INVALID, "sh_bank_render: NULL argument");
    if (nframes == 0) return SH_OK;
    if (bus_f32 && bus_f32->bytes < (size_t)nframes * 8) return sh::set_error(SH_ERR_INVALID, "sh_bank_render: bus_f32 too small");
    if (bus_f64 && bus_f64->bytes < (size_t)nframes * 16) return sh::set_error(SH_ERR_INVALID, "sh_bank_render: bus_f64 too small");
    int rc = bank_check_plain(b, "sh_bank_render");
    if (rc) return rc;
    rc = locate(b, 0, b->nvoices, start);
    if (rc) return rc;
    float2* o32 = bus_f32 ? (float2*)bus_f32->ptr : nullptr;
    double2* o64 = bus_f64 ? (double2*)bus_f64->ptr : nullptr;
    dim3 grid(sh::div_up(nframes, 64));
    hipStream_t st = sh::state().stream;
    if (b->nvoices >= 128) {
        hipLaunchKernelGGL(k_bank_render<16>, grid, dim3(16 * 64), 0, st, ptrs(b), b->nvoices, b->d_loc, start, nframes, o32, o64);
    } else if (b->nvoices >= 16) {
        hipLaunchKernelGGL(k_bank_render<8>, grid, dim3(8 * 64), 0, st, ptrs(b), b->nvoices, b->d_loc, start, nframes, o32, o64);
    } else {
        hipLaunchKernelGGL(k_bank_render<2>, grid, dim3(2 * 64), 0, st, ptrs(b), b->nvoices, b->d_loc, start, nframes, o32, o64);
    }
    SH_CHECK_LAUNCH("k_bank_render");
    return SH_OK;
}

int sh_mix_bus_f32(const sh_buf* voices, uint32_t nvoices, size_t stride, uint32_t nframes,
                   const sh_buf* gains_lr, sh_buf* bus_f32) {
    SH_REQUIRE_INIT();
    if (!voices || !gains_lr || !bus_f32 || nvoices == 0) return sh::set_error(SH_ERR_INVALID, "sh_mix_bus_f32: NULL argument");
    if (nframes == 0) return SH_OK;
    if (stride < nframes || voices->bytes / 4 < (size_t)(nvoices - 1) * stride + nframes)
        return sh::set_error(SH_ERR_INVALID, "sh_mix_bus_f32: voice buffer too small for %u x %u (stride %zu)", nvoices, nframes, stride);
    if (bus_f32->bytes < (size_t)nframes * 8) return sh::set_error(SH_ERR_INVALID, "sh_mix_bus_f32: bus too small");
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
    float2* parts = (float2*)sh::state().scratch;
    float2* dst = groups > 1 ? parts : (float2*)bus_f32->ptr;
    hipLaunchKernelGGL(k_mix_bus_f32<W>, dim3(tiles, groups), dim3(W * 64), 0, st,
                       (const float*)voices->ptr, nvoices, stride, nframes, (const float2*)gains_lr->ptr, vpg,
                       dst, (size_t)nframes);
    SH_CHECK_LAUNCH("k_mix_bus_f32");
    if (groups > 1) {
        hipLaunchKernelGGL(k_bus_sum, dim3(sh::div_up(nframes, 256)), dim3(256), 0, st,
                           (const float2*)parts, groups, (size_t)nframes, nframes, (float2*)bus_f32->ptr);
        SH_CHECK_LAUNCH("k_bus_sum");
    }
    return SH_OK;
}

int sh_bus_finalize(const sh_buf* bus_f64, size_t nvalues, sh_buf* bus_f32) {
    SH_REQUIRE_INIT();
    if (!bus_f64 || !bus_f32) return sh::set_error(SH_ERR_INVALID, "sh_bus_finalize: NULL argument");
    if (bus_f64->bytes < nvalues * 8 || bus_f32->bytes < nvalues * 4) return sh::set_error(SH_ERR_INVALID, "sh_bus_finalize: buffer too small");
    if (!nvalues) return SH_OK;
    hipLaunchKernelGGL(k_bus_finalize, dim3(sh::div_up(nvalues, 256)), dim3(256), 0, sh::state().stream,
                       (const double*)bus_f64->ptr, nvalues, (float*)bus_f32->ptr);
    SH_CHECK_LAUNCH("k_bus_finalize");
    return SH_OK;
}

int sh_scan_f64(const sh_buf* x, uint32_t n, double carry_in, sh_buf* out, double* carry_out) {
    SH_REQUIRE_INIT();
    if (!x || !out) return sh::set_error(SH_ERR_INVALID, "sh_scan_f64: NULL argument");
    if (x->bytes < (size_t)n * 8 || out->bytes < (size_t)n * 8) return sh::set_error(SH_ERR_INVALID, "sh_scan_f64: buffer too small");
    if (n == 0) {
        if (carry_out) *carry_out = carry_in;
        return SH_OK;
    }
    uint32_t ntiles = sh::div_up(n, SCAN_TILE);
    int rc = sh::ensure_scratch((size_t)(ntiles + 1) * 8);
    if (rc) return rc;
    double* sums = (double*)sh::state().scratch;
    hipStream_t st = sh::state().stream;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(256), 0, st, (const double*)x->ptr, n, sums);
    SH_CHECK_LAUNCH("k_scan_tile_sums");
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, sums, ntiles, carry_in);
    SH_CHECK_LAUNCH("k_scan_sums");
    hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(256), 0, st, (const double*)x->ptr, n, (const double*)sums, (double*)out->ptr);
    SH_CHECK_LAUNCH("k_scan_apply");
    if (carry_out) {
        SH_HIP(hipMemcpyAsync(carry_out, sums + ntiles, 8, hipMemcpyDeviceToHost, st));
        SH_HIP(hipStreamSynchronize(st));
    }
    return SH_OK;
}

}  // extern "C"
