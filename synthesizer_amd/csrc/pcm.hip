// pcm.hip -- integer/float PCM kernels: quantise, saturating add (Sample.mix), the mixer's
// saturating chain, and linear-interpolation resampling (Sample.resample).
//
// These restate CPython 3.10 Modules/audioop.c (audioop_add_impl, audioop_ratecv_impl), the
// arithmetic synthplayer's sample.py delegates to.  All of them are HBM-bound byte/integer work:
// coalesced loads, one output element (or a small vector) per thread, no LDS except for the
// cross-wave combine of the mixer chain.  Built with -ffp-contract=off: audioop forms
// prev*d + cur*(outrate-d) with two roundings and a division, and so does k_resample.
#include "common.hpp"
#include <mutex>
#include <string.h>
#include <new>

#define SH_PCM_CONST __attribute__((address_space(4)))

namespace {

typedef short short2v __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef int   int4v   __attribute__((ext_vector_type(4)));
typedef char  char16v __attribute__((ext_vector_type(16)));

// ---- quantise: int(scale*v), truncation toward zero (sample.py Sample.from_osc_block) -------
// rnd (sh_set_option(SH_OPT_QUANTISE_ROUND)): the other reading of the reference's rule, round(scale*v) -- half to even, rint -- in
// place of int()'s truncation toward zero; never with clip (the saturating forms are this build's own: [SPEC]).
template <typename OutT, typename InT = float>
__global__ __launch_bounds__(256) void k_quantize(const InT* __restrict__ in, size_t n, double scale,
                                                  double lo, double hi, OutT* __restrict__ out,
                                                  int* __restrict__ flag, int clip, int rnd = 0) {
    size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    double v = scale * (double)in[i];          // float64 product, like the Python expression
    double t = rnd ? rint(v) : trunc(v);
    if (!(t >= lo && t <= hi)) {               // also catches NaN
        if (clip) {
            t = (t > hi) ? hi : lo;
            if (v != v) t = 0.0;
        } else {
            *flag = 1;
            t = 0.0;
        }
    }
    out[i] = (OutT)(long long)t;
}

// The vector forms: a workgroup covers 512 vectors of 16 bytes, each thread two of them 256 apart -- every load instruction
// of a wave reads one contiguous kilobyte (streaming: nothing is read twice) and every store instruction writes a contiguous
// 256 / 512 bytes.  round 3 microbenchmark (profiles/r03_summary.md): float64 6.3 TB/s this way, 5.8 with eight consecutive samples per thread and a
// 16-byte store, 4.4 with those loads marked non-temporal, 4.9 one sample per thread.
template <bool CLIP>
__device__ __forceinline__ short quantize16(double p, bool& bad, bool rnd = false) {
    double t = rnd ? rint(p) : trunc(p);
    if (!(t >= -32768.0 && t <= 32767.0)) {                  // also catches NaN
        if (CLIP) { t = (t > 32767.0) ? 32767.0 : -32768.0; if (p != p) t = 0.0; }
        else { bad = true; t = 0.0; }
    }
    return (short)(int)t;
}

// float32 -> int16, four samples per 16-byte vector
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_quantize_f32_i16_vec(const float4v* __restrict__ in, size_t nvec, double scale,
                                                              short4v* __restrict__ out, int* __restrict__ flag, int clip, int rnd = 0) {
    const size_t base = sh::block_id() * 512 + threadIdx.x;
    float4v v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) if (base + u * 256 < nvec) v[u] = __builtin_nontemporal_load(in + base + u * 256);
    bool bad = false;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (base + u * 256 >= nvec) break;
        short4v r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = clip ? quantize16<true>(scale * (double)v[u][j], bad) : quantize16<false>(scale * (double)v[u][j], bad, rnd != 0);
        __builtin_nontemporal_store(r, out + base + u * 256);       // streaming store: +3.5 %
    }
    if (bad) *flag = 1;
}

// float64 -> int16, two samples per 16-byte vector; the route WaveSynth.to_sample takes
typedef double double2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_quantize_f64_i16_vec(const double2v* __restrict__ in, size_t nvec, double scale,
                                                              short2v* __restrict__ out, int* __restrict__ flag, int rnd = 0) {
    const size_t base = sh::block_id() * 512 + threadIdx.x;
    double2v v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) if (base + u * 256 < nvec) v[u] = __builtin_nontemporal_load(in + base + u * 256);
    bool bad = false;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (base + u * 256 >= nvec) break;
        short2v r;
        r[0] = quantize16<false>(scale * v[u][0], bad, rnd != 0);
        r[1] = quantize16<false>(scale * v[u][1], bad, rnd != 0);
        __builtin_nontemporal_store(r, out + base + u * 256);       // streaming store: +3.5 %
    }
    if (bad) *flag = 1;
}

// ---- audioop.add (no __restrict__: Sample.mix_at adds in place) ---------------------------------
template <typename V, bool NT>
__global__ __launch_bounds__(256) void k_add_vec(const V* a, const V* b, V* o, size_t nvec) {
    size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= nvec) return;
    const V r = __builtin_elementwise_add_sat(sh::load_vec<NT, V>(a + i), sh::load_vec<NT, V>(b + i));
    if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r;        // streaming sizes: loads +3 %, store +4 %
}

template <typename T>
__global__ __launch_bounds__(256) void k_add_scalar(const T* a, const T* b,
                                                    T* o, size_t n) {
    size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    o[i] = __builtin_elementwise_add_sat(a[i], b[i]);
}


// ---- mixer chain: mixed = add(...add(add(c0, c1), c2)..., c_{N-1}), saturating at every step ----
// x -> clamp(x + s, lo, hi) composes into x -> clamp(x + a, L, U) (closed under composition), so
// each wave folds a contiguous range of voices into one (a, L, U) triple per sample and wave 0
// applies the W triples in voice order to x = 0.  Bit-exact with the sequential fold.
constexpr int CH_BIG = 1 << 28;

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Per-lane fold state for 8 samples: the sums in int32, the two clamp bounds as packed int16 pairs.  Once a range
// holds one voice the bounds are inside the int16 range for good (L = U = "no bound" only before), and
// bound' = clamp(bound + s, -32768, 32767) is exactly the packed saturating add: one instruction per two samples;
// the sum takes one dot-product instruction per sample ((s_lo, s_hi) . (1, 0) + a), no unpacking.
typedef short short2v __attribute__((ext_vector_type(2)));
struct ChainFold {
    int a[8];
    short2v L[4], U[4];
    bool any;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { L[q] = (short2v){-32768, -32768}; U[q] = (short2v){32767, 32767}; }
        any = false;
    }
    // first voice of the range: the sum starts, the bounds become the int16 range (set by init)
    __device__ __forceinline__ void first(const short8v x) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = x[j];
        any = true;
    }
    __device__ __forceinline__ void add(const short8v x) {
#define SH_PAIR(Q_)                                                                              \
        {                                                                                        \
            const short2v s2 = __builtin_shufflevector(x, x, 2 * Q_, 2 * Q_ + 1);                \
            L[Q_] = __builtin_elementwise_add_sat(L[Q_], s2);                                    \
            U[Q_] = __builtin_elementwise_add_sat(U[Q_], s2);                                    \
            a[2 * Q_] = __builtin_amdgcn_sdot2(s2, (short2v){1, 0}, a[2 * Q_], false);           \
            a[2 * Q_ + 1] = __builtin_amdgcn_sdot2(s2, (short2v){0, 1}, a[2 * Q_ + 1], false);   \
        }
        SH_PAIR(0) SH_PAIR(1) SH_PAIR(2) SH_PAIR(3)
#undef SH_PAIR
    }
    __device__ __forceinline__ int lo(int j) const { return any ? (int)L[j >> 1][j & 1] : -CH_BIG; }
    __device__ __forceinline__ int hi(int j) const { return any ? (int)U[j >> 1][j & 1] : CH_BIG; }
};

// PAN forms of the two chain kernels below (sh_mix_chain_pan_i16): the rows are MONO voices and every voice enters the fold as
// Sample.stereo(lf, rf) of itself -- audioop.tostereo: (fbound(v * lf), fbound(v * rf)) per frame, fbound = clamp to the sample
// range, then floor -- made in registers from the 8 bytes of four mono frames; the fold is the same chain over the eight
// stereo samples.  For every finite product fbound(p) == clamp(floor(p), -32768, 32767) (the "val < minval + 1 -> minval" branch
// selects values whose floor is minval anyway), so a frame costs: one int -> float64 conversion, two products, two floors, two
// conversions (saturating at the int32 range) and ONE v_cvt_pk_i16_i32, whose saturation is the clamp and whose packed result is
// the interleaved (L, R) pair.  The factors of a voice are wave-uniform: scalar loads.
typedef short short4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ short8v pan4(const short4v m, const double lf, const double rf) {
    union { short8v v; short2v p[4]; } r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double x = (double)m[j];
        r.p[j] = __builtin_amdgcn_cvt_pk_i16((int)floor(x * lf), (int)floor(x * rf));
    }
    return r.v;
}
template <bool NT>
__device__ __forceinline__ short8v load_row8(const short* __restrict__ chunks, size_t v, size_t stride, uint32_t s0, const double2* __restrict__ pan) {
    if (pan) {                                           // (uniform) s0 = the first of eight STEREO samples: four mono frames from s0 / 2
        const double2 f = pan[v];
        return pan4(sh::load_vec<NT, short4v>(chunks + v * stride + (s0 >> 1)), f.x, f.y);
    }
    return sh::load_vec<NT, short8v>(chunks + v * stride + s0);
}

// WAVES waves = (WAVES / COLS) voice ranges x COLS adjacent 1 KB columns: a workgroup visits COLS KB of a row at a time.
template <int WAVES, int COLS, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_mix_chain_i16(const short* __restrict__ chunks, uint32_t nvoices,
                                                              size_t stride, uint32_t nsamples,
                                                              short* __restrict__ out, const double2* __restrict__ pan = nullptr) {
    constexpr int S = 8;                                  // samples per lane: one 16-byte load per voice row
    constexpr int VG = WAVES / COLS;
    __shared__ int red[WAVES][3][S][64];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t col = wave % COLS, vg = wave / COLS;
    const uint32_t s0 = (((uint32_t)sh::block_id() * COLS + col) * 64 + lane) * S;       // (grid1d folds beyond 2^21 workgroups)
    const uint32_t per = (nvoices + VG - 1) / VG;
    const uint32_t v0 = vg * per;
    uint32_t v1 = v0 + per;
    if (v1 > nvoices) v1 = nvoices;
    ChainFold f;
    f.init();
    const bool vec = (s0 + S - 1 < nsamples) && ((stride & (pan ? 3 : S - 1)) == 0);
    if (s0 < nsamples && v0 < v1) {
        if (vec) {
            uint32_t v = v0;
            f.first(load_row8<NT>(chunks, v, stride, s0, pan));
            ++v;
            for (; v + 3 < v1; v += 4) {                  // four voice rows in flight
                const short8v x0 = load_row8<NT>(chunks, v, stride, s0, pan);
                const short8v x1 = load_row8<NT>(chunks, v + 1, stride, s0, pan);
                const short8v x2 = load_row8<NT>(chunks, v + 2, stride, s0, pan);
                const short8v x3 = load_row8<NT>(chunks, v + 3, stride, s0, pan);
                f.add(x0); f.add(x1); f.add(x2); f.add(x3);
            }
            for (; v < v1; ++v) f.add(load_row8<NT>(chunks, v, stride, s0, pan));
        } else {
            for (uint32_t v = v0; v < v1; ++v) {
                short8v x;
                if (pan) {
                    const short* row = chunks + (size_t)v * stride + (s0 >> 1);
                    short4v m;
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[j] = (s0 + 2 * j < nsamples) ? row[j] : (short)0;
                    const double2 fac = pan[v];
                    x = pan4(m, fac.x, fac.y);
                } else {
                    const short* row = chunks + (size_t)v * stride + s0;
#pragma unroll
                    for (int j = 0; j < S; ++j) x[j] = (s0 + j < nsamples) ? row[j] : (short)0;
                }
                if (v == v0) f.first(x); else f.add(x);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
        red[wave][0][j][lane] = f.a[j];
        red[wave][1][j][lane] = f.lo(j);
        red[wave][2][j][lane] = f.hi(j);
    }
    __syncthreads();
    if (vg == 0 && s0 < nsamples) {
        short8v r;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            int x = 0;
#pragma unroll
            for (int g = 0; g < VG; ++g) {
                const int w = g * COLS + col;
                x = clampi(x + red[w][0][j][lane], red[w][1][j][lane], red[w][2][j][lane]);
            }
            r[j] = (short)x;
        }
        if (s0 + S - 1 < nsamples && ((reinterpret_cast<uintptr_t>(out + s0) & 15) == 0)) {
            *reinterpret_cast<short8v*>(out + s0) = r;
        } else {
            for (uint32_t j = 0; j < S && s0 + j < nsamples; ++j) out[s0 + j] = r[j];
        }
    }
}

// Long buffers: enough columns to fill the chip without splitting the voices, so a lane simply runs the reference's
// loop -- mixed = add_sat(mixed, row) down all the rows, packed int16 -- and a workgroup walks WAVES KB of every row.
// No fold state, no LDS; INFLIGHT independent row loads per lane.
template <int WAVES, int INFLIGHT, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_mix_chain_direct(const short* __restrict__ chunks, uint32_t nvoices, size_t stride,
                                                                 uint32_t nsamples, short* __restrict__ out) {
    const uint32_t s0 = (uint32_t)(sh::block_id() * (WAVES * 64) + threadIdx.x) * 8;
    if (s0 >= nsamples) return;
    const short* col = chunks + s0;
    if (s0 + 8 <= nsamples) {
        short8v acc = *reinterpret_cast<const short8v*>(col);
        uint32_t v = 1;
        for (; v + INFLIGHT <= nvoices; v += INFLIGHT) {
            short8v x[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) x[k] = sh::load_vec<NT, short8v>(col + (size_t)(v + k) * stride);     // NT: 0.81 -> 0.90 of the HBM peak on 2 GB
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) acc = __builtin_elementwise_add_sat(acc, x[k]);
        }
        for (; v < nvoices; ++v) acc = __builtin_elementwise_add_sat(acc, *reinterpret_cast<const short8v*>(col + (size_t)v * stride));
        *reinterpret_cast<short8v*>(out + s0) = acc;
    } else {
        for (uint32_t j = 0; s0 + j < nsamples; ++j) {
            short acc = col[j];
            for (uint32_t v = 1; v < nvoices; ++v) acc = __builtin_elementwise_add_sat(acc, col[(size_t)v * stride + j]);
            out[s0 + j] = acc;
        }
    }
}

// The direct loops with S samples (F frames) per lane: for rows of MIDDLING length (a few hundred thousand samples: too few 1 KB
// columns to fill the chip with eight samples per lane, more than the split kernels like) four samples per lane -- 8-byte loads,
// twice the wavefronts, eight row loads in flight -- run the mono chain at 0.87 of the HBM peak where the split kernel reaches 0.74
// (1024 rows x 480 000 samples: 166 -> 142 us).  The PAN chain -- mono rows that enter as Sample.stereo(lf, rf) of themselves, see pan4
// -- is bound by its float64 arithmetic (ten operations per frame), not by HBM: four frames per lane at every length from ~300 000
// frames (480 000: 290 -> 202 us, 0.61 of HBM; 960 000: 392 -> 358 us, 0.69); eight frames per lane were slower at both.
template <int S> struct ShortVec;
template <> struct ShortVec<8> { typedef short8v type; };
template <> struct ShortVec<4> { typedef short4v type; };
template <> struct ShortVec<2> { typedef short2v type; };

template <int S, int WAVES, int INFLIGHT, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_mix_chain_direct_s(const short* __restrict__ chunks, uint32_t nvoices, size_t stride,
                                                                   uint32_t nsamples, short* __restrict__ out) {
    typedef typename ShortVec<S>::type vec;
    const uint32_t s0 = (uint32_t)(sh::block_id() * (WAVES * 64) + threadIdx.x) * S;
    if (s0 >= nsamples) return;
    const short* col = chunks + s0;
    if (s0 + S <= nsamples) {
        vec acc = *reinterpret_cast<const vec*>(col);
        uint32_t v = 1;
        for (; v + INFLIGHT <= nvoices; v += INFLIGHT) {
            vec x[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) x[k] = sh::load_vec<NT, vec>(col + (size_t)(v + k) * stride);
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) acc = __builtin_elementwise_add_sat(acc, x[k]);
        }
        for (; v < nvoices; ++v) acc = __builtin_elementwise_add_sat(acc, *reinterpret_cast<const vec*>(col + (size_t)v * stride));
        __builtin_nontemporal_store(acc, reinterpret_cast<vec*>(out + s0));
    } else {
        for (uint32_t j = 0; s0 + j < nsamples; ++j) {
            short acc = col[j];
            for (uint32_t v = 1; v < nvoices; ++v) acc = __builtin_elementwise_add_sat(acc, col[(size_t)v * stride + j]);
            out[s0 + j] = acc;
        }
    }
}

template <int F, int WAVES, int INFLIGHT, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_mix_chain_pan_direct_s(const short* __restrict__ chunks, uint32_t nvoices, size_t stride,
                                                                       uint32_t nframes, const double2* __restrict__ pan, short* __restrict__ out) {
    typedef typename ShortVec<F>::type vin;                       // F mono frames in, 2 F stereo samples out
    typedef typename ShortVec<2 * F>::type vout;
    const uint32_t f0 = (uint32_t)(sh::block_id() * (WAVES * 64) + threadIdx.x) * F;
    if (f0 >= nframes) return;
    const short* col = chunks + f0;
    const double SH_PCM_CONST* fac = (const double SH_PCM_CONST*)pan;
    auto stereo = [](const vin m, const double lf, const double rf) {
        union { vout v; short2v p[F]; } r;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const double x = (double)m[j];
            r.p[j] = __builtin_amdgcn_cvt_pk_i16((int)floor(x * lf), (int)floor(x * rf));
        }
        return r.v;
    };
    if (f0 + F <= nframes) {
        vout acc = stereo(*reinterpret_cast<const vin*>(col), fac[0], fac[1]);
        uint32_t v = 1;
        for (; v + INFLIGHT <= nvoices; v += INFLIGHT) {
            vin x[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) x[k] = sh::load_vec<NT, vin>(col + (size_t)(v + k) * stride);
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) acc = __builtin_elementwise_add_sat(acc, stereo(x[k], fac[2 * (v + k)], fac[2 * (v + k) + 1]));
        }
        for (; v < nvoices; ++v) acc = __builtin_elementwise_add_sat(acc, stereo(*reinterpret_cast<const vin*>(col + (size_t)v * stride), fac[2 * v], fac[2 * v + 1]));
        __builtin_nontemporal_store(acc, reinterpret_cast<vout*>(out + 2 * (size_t)f0));
    } else {
        for (uint32_t j = 0; f0 + j < nframes; ++j) {
            short2v acc = {0, 0};
            for (uint32_t v = 0; v < nvoices; ++v) {
                const double x = (double)col[(size_t)v * stride + j];
                const short2v p = __builtin_amdgcn_cvt_pk_i16((int)floor(x * fac[2 * v]), (int)floor(x * fac[2 * v + 1]));
                acc = v == 0 ? p : __builtin_elementwise_add_sat(acc, p);
            }
            out[2 * (size_t)(f0 + j)] = acc[0];
            out[2 * (size_t)(f0 + j) + 1] = acc[1];
        }
    }
}

// The same fold over chunks that live where their samples live: a table of (pointer, samples available) per
// source instead of one padded array -- the real-time mixer's loop without the staging copy (every active sample is
// read in place at its play position; past its end it counts as silence, which the fold skips: x + 0 saturates to x).
struct ChainSrc {
    const short* p;
    uint32_t n;
    uint32_t pad;
};

__device__ __forceinline__ short8v chain_load8(const short* p, uint32_t n, uint32_t s0) {
    short8v x = {0, 0, 0, 0, 0, 0, 0, 0};
    if (s0 + 8 <= n && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        x = *reinterpret_cast<const short8v*>(p + s0);
    } else if (s0 < n) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (s0 + j < n) x[j] = p[s0 + j];
    }
    return x;
}

template <int WAVES>
__device__ __forceinline__ void mix_chain_gather_body(const ChainSrc* __restrict__ tab, uint32_t nsrc, uint32_t nsamples,
                                                      short* __restrict__ out) {
    constexpr int S = 8;
    __shared__ int red[WAVES][3][S][64];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t s0 = (blockIdx.x * 64 + lane) * S;
    const uint32_t per = (nsrc + WAVES - 1) / WAVES;
    const uint32_t v0 = wave * per;
    uint32_t v1 = v0 + per;
    if (v1 > nsrc) v1 = nsrc;
    ChainFold f;
    f.init();
    if (s0 < nsamples && v0 < v1) {
        uint32_t v = v0;
        {
            const ChainSrc c0 = tab[v];
            f.first(chain_load8(c0.p, c0.n, s0));         // a source past its end contributes zeros: still a voice of the fold
            ++v;
        }
        for (; v + 3 < v1; v += 4) {                      // four sources in flight
            const ChainSrc c0 = tab[v], c1 = tab[v + 1], c2 = tab[v + 2], c3 = tab[v + 3];
            const short8v x0 = chain_load8(c0.p, c0.n, s0);
            const short8v x1 = chain_load8(c1.p, c1.n, s0);
            const short8v x2 = chain_load8(c2.p, c2.n, s0);
            const short8v x3 = chain_load8(c3.p, c3.n, s0);
            f.add(x0); f.add(x1); f.add(x2); f.add(x3);
        }
        for (; v < v1; ++v) {
            const ChainSrc c0 = tab[v];
            f.add(chain_load8(c0.p, c0.n, s0));
        }
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
        red[wave][0][j][lane] = f.a[j];
        red[wave][1][j][lane] = f.lo(j);
        red[wave][2][j][lane] = f.hi(j);
    }
    __syncthreads();
    if (wave == 0 && s0 < nsamples) {
        short8v r;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            int x = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) x = clampi(x + red[w][0][j][lane], red[w][1][j][lane], red[w][2][j][lane]);
            r[j] = (short)x;
        }
        if (s0 + S - 1 < nsamples && ((reinterpret_cast<uintptr_t>(out + s0) & 15) == 0)) {
            *reinterpret_cast<short8v*>(out + s0) = r;
        } else {
            for (uint32_t j = 0; j < S && s0 + j < nsamples; ++j) out[s0 + j] = r[j];
        }
    }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mix_chain_gather(const ChainSrc* __restrict__ tab, uint32_t nsrc, uint32_t nsamples,
                                                                 short* __restrict__ out) {
    mix_chain_gather_body<WAVES>(tab, nsrc, nsamples, out);
}

// The same with the source table IN the kernel arguments (up to 64 sources: one real-time mixer turn): no table upload in front of
// the launch -- the copy of a pageable kilobyte costs more than the kernel.
struct ChainTab { ChainSrc e[64]; };
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mix_chain_gather_args(const ChainTab tab, uint32_t nsrc, uint32_t nsamples,
                                                                      short* __restrict__ out) {
    mix_chain_gather_body<WAVES>(tab.e, nsrc, nsamples, out);
}

// the direct loop over the pointer table (long samples: mix_samples of whole tracks)
template <int WAVES, int INFLIGHT, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_mix_chain_gather_direct(const ChainSrc* __restrict__ tab, uint32_t nsrc, uint32_t nsamples,
                                                                        short* __restrict__ out) {
    const uint32_t s0 = (blockIdx.x * (WAVES * 64) + threadIdx.x) * 8;
    const uint32_t wave_s0 = __builtin_amdgcn_readfirstlane(s0 - (threadIdx.x & 63) * 8);
    if (s0 >= nsamples) return;
    short8v acc = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t v = 0;
    if (nsrc >= INFLIGHT) {
        ChainSrc c[INFLIGHT], nx[INFLIGHT];               // table entries one batch ahead: their scalar loads overlap the row loads
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) c[k] = tab[k];
        for (; v + INFLIGHT <= nsrc; v += INFLIGHT) {
            const bool more = v + 2 * INFLIGHT <= nsrc;
            if (more) {
#pragma unroll
                for (int k = 0; k < INFLIGHT; ++k) nx[k] = tab[v + INFLIGHT + k];
            }
            short8v x[INFLIGHT];
            bool whole = true;                            // wave-uniform: every source of the batch covers this wave's 1 KB, aligned
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) whole = whole && wave_s0 + 512 <= c[k].n && (reinterpret_cast<uintptr_t>(c[k].p) & 15) == 0;
            if (whole) {
#pragma unroll
                for (int k = 0; k < INFLIGHT; ++k) x[k] = sh::load_vec<NT, short8v>(c[k].p + s0);
            } else {
#pragma unroll
                for (int k = 0; k < INFLIGHT; ++k) x[k] = chain_load8(c[k].p, c[k].n, s0);
            }
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) acc = __builtin_elementwise_add_sat(acc, x[k]);
            if (more) {
#pragma unroll
                for (int k = 0; k < INFLIGHT; ++k) c[k] = nx[k];
            }
        }
    }
    for (; v < nsrc; ++v) {
        const ChainSrc c = tab[v];
        acc = __builtin_elementwise_add_sat(acc, chain_load8(c.p, c.n, s0));
    }
    if (s0 + 8 <= nsamples && ((reinterpret_cast<uintptr_t>(out + s0) & 15) == 0)) {
        *reinterpret_cast<short8v*>(out + s0) = acc;
    } else {
        for (uint32_t j = 0; j < 8 && s0 + j < nsamples; ++j) out[s0 + j] = acc[j];
    }
}

// ---- the mixer fold for the other sample widths audioop.add takes (8, 24 and 32 bits) -------------------------------------
// The reference's loop as it stands -- mixed = clamp(mixed + sample) down the sources, in their order, per output sample --
// with four consecutive samples per thread and the source table read by scalar loads (uniform).  8-bit and 24-bit samples are
// assembled from bytes (a 24-bit sample: GETINT24, sign-extended; stored back as its three low bytes), 32-bit sums are taken in
// 64 bits.  Sources past their end count as silence.  These widths are the mixer's side door -- WAV files that were not
// converted to 16 bits first -- so the kernel is the plain one; the 16-bit shapes above are the tuned ones.
struct ChainSrcB {
    const unsigned char* p;
    uint32_t n;               // samples available
    uint32_t pad;
};

template <int WIDTH>
__device__ __forceinline__ long long chain_get(const unsigned char* p, size_t i) {
    if (WIDTH == 1) return (long long)(signed char)p[i];
    if (WIDTH == 3) {
        const unsigned char* q = p + 3 * i;
        return (long long)((int)q[0] | ((int)q[1] << 8) | ((int)(signed char)q[2] << 16));
    }
    int v;
    __builtin_memcpy(&v, p + 4 * i, 4);
    return (long long)v;
}

template <int WIDTH>
__device__ __forceinline__ void chain_put(unsigned char* p, size_t i, long long x) {
    if (WIDTH == 1) { p[i] = (unsigned char)(signed char)x; return; }
    if (WIDTH == 3) {
        unsigned char* q = p + 3 * i;
        const int v = (int)x;
        q[0] = (unsigned char)(v & 0xFF); q[1] = (unsigned char)((v >> 8) & 0xFF); q[2] = (unsigned char)((v >> 16) & 0xFF);
        return;
    }
    const int v = (int)x;
    __builtin_memcpy(p + 4 * i, &v, 4);
}

template <int WIDTH>
__global__ __launch_bounds__(256) void k_mix_chain_gather_w(const ChainSrcB* __restrict__ tab, uint32_t nsrc, uint32_t nsamples,
                                                            unsigned char* __restrict__ out) {
    const size_t s0 = (sh::block_id() * 256 + threadIdx.x) * 4;
    if (s0 >= nsamples) return;
    constexpr long long HI = WIDTH == 1 ? 127LL : (WIDTH == 3 ? 8388607LL : 2147483647LL), LO = -HI - 1;
    long long acc[4] = {0, 0, 0, 0};
    for (uint32_t v = 0; v < nsrc; ++v) {
        const unsigned char* p = tab[v].p;
        const uint32_t n = tab[v].n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = s0 + j;
            if (i < n) {
                const long long t = acc[j] + chain_get<WIDTH>(p, i);
                acc[j] = t > HI ? HI : (t < LO ? LO : t);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (s0 + j < nsamples) chain_put<WIDTH>(out, s0 + j, acc[j]);
}

// ---- audioop.ratecv / float32 resample ----------------------------------------------------------
// One thread per output sample (frame m, channel c).  Output m interpolates input frames j-1 and j,
// j = ceil(m*inrate/outrate), d = j*outrate - m*inrate (rates gcd-reduced): identical index
// arithmetic to the reference's state machine, evaluated in closed form.
struct RatecvArgs {
    uint64_t n_out_samples;     // out_frames * nch
    uint64_t m_base;            // output frame index of the launch's first frame (range launches; else 0)
    uint32_t nch;
    uint32_t inr, outr;
    uint32_t step_q, step_r;    // inr / outr and inr % outr: output frame m+1 starts (step_q, step_r) after frame m
    double   inv_outr;
    int      shift;             // 32 - 8*width (integer PCM)
};

// how a kernel forms one output sample
enum { RS_INT_F64 = 0,          // integer PCM through the float64 expression (any width, any rate)
       RS_FLOAT = 1,            // float32 PCM through the float64 expression
       RS_INT_SMALL = 2 };      // 8/16-bit PCM with reduced outrate < 65536: exact 32-bit integer arithmetic

__device__ __forceinline__ void ratecv_index(const RatecvArgs& A, uint64_t m, uint64_t& j, uint32_t& d) {
    const uint64_t M = m * (uint64_t)A.inr;
    uint64_t q;
    int64_t r;
    if (M < (1ull << 52)) {
        q = (uint64_t)floor((double)M * A.inv_outr);
        r = (int64_t)(M - q * (uint64_t)A.outr);
        if (r < 0) { q -= 1; r += A.outr; }
        else if (r >= (int64_t)A.outr) { q += 1; r -= A.outr; }
    } else {
        q = M / A.outr;
        r = (int64_t)(M % A.outr);
    }
    j = q + (r != 0);
    d = r ? (uint32_t)(A.outr - (uint32_t)r) : 0u;
}

// (prev*d + cur*(outrate-d)) / outrate in float64, exactly as audioop forms it: two products, one sum, one
// correctly rounded division.  The division is Markstein's sequence q = a*y, r = fma(-q, b, a),
// q' = fma(r, y, q) with y = RN(1/b): it returns the correctly rounded quotient (checked against IEEE
// division on 3e8 operands in tests/ and by every bit-exact parity test), at 3 instructions instead of
// the ~12 of the generic lowering -- this kernel must stay HBM-bound.
__device__ __forceinline__ double ratecv_value(double prev, double cur, double dd, double od, double outr, double inv_outr) {
    const double a = prev * dd + cur * od;
    const double q = a * inv_outr;
    const double r = fma(-q, outr, a);
    return fma(r, inv_outr, q);
}

// 8/16-bit PCM, reduced outrate < 65536.  audioop computes trunc(fl(N / outr)) >> s with N = (prev*d + cur*(outr-d)) << s
// (s = 32 - bits): N is an exact float64 integer (< 2^48), a non-integer N/outr is at least 1/outr > 2^-16 away
// from an integer while its float64 rounding error is below 2^-21, so the truncation equals integer division, and
// trunc(.) >> s == floor(M / outr) with M = prev*d + cur*(outr-d) (|M| <= 2^(bits-1)*outr < 2^31; for M < 0 the
// inner truncation loses less than 2^-s < 1/outr, which the floor of the arithmetic shift restores).  floor(M/outr)
// is formed as an unsigned division of u = M + 2^(bits-1)*outr (0 <= u < 2^32): trunc(fma(u, 1/outr, 1/(2 outr))) in
// float64 -- (u + 1/2)/outr is at least 1/(2 outr) > 2^-17 away from every integer and the evaluation error is below
// 2^-20, so no correction step is needed.  Bit-exactness against audioop is what tests/test_gpu_pcm.py asserts on
// both paths.
template <typename T>
__device__ __forceinline__ T ratecv_small_int(T prev, T cur, uint32_t d, uint32_t outr, double inv_outr) {
    constexpr int HALF = 1 << (8 * (int)sizeof(T) - 1);
    const int M = (int)prev * (int)d + (int)cur * (int)(outr - d);
    const uint32_t u = (uint32_t)M + (uint32_t)HALF * outr;
    const uint32_t q = (uint32_t)fma((double)u, inv_outr, 0.5 * inv_outr);
    return (T)((int)q - HALF);
}

template <typename T, int MODE>
__device__ __forceinline__ T ratecv_sample(T prev, T cur, uint32_t d, const RatecvArgs& A) {
    if (MODE == RS_INT_SMALL) {
        if constexpr (sizeof(T) <= 2) return ratecv_small_int<T>(prev, cur, d, A.outr, A.inv_outr);
        else return (T)0;
    }
    const double dd = (double)d, od = (double)(A.outr - d), outr = (double)A.outr;
    if (MODE == RS_FLOAT) return (T)ratecv_value((double)prev, (double)cur, dd, od, outr, A.inv_outr);
    const int ci = (int)((unsigned)(int)cur << A.shift);                               // GETSAMPLE32
    const int pi = (int)((unsigned)(int)prev << A.shift);
    return (T)((int)ratecv_value((double)pi, (double)ci, dd, od, outr, A.inv_outr) >> A.shift);   // SETSAMPLE32
}

// One thread = one output frame x VEC channels, moved as one vector (VEC*sizeof(T) bytes).
template <typename T, int VEC, int MODE>
__global__ __launch_bounds__(256) void k_resample(const T* __restrict__ in, T* __restrict__ out, RatecvArgs A) {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= A.n_out_samples) return;                     // here: number of (frame, channel group) units
    const uint32_t groups = A.nch / VEC;
    uint64_t m;
    uint32_t cg;
    if (groups == 1) { m = u; cg = 0; }
    else if (u < 0xFFFFFFFFull) { uint32_t u32 = (uint32_t)u; uint32_t m32 = u32 / groups; m = m32; cg = u32 - m32 * groups; }
    else { m = u / groups; cg = (uint32_t)(u - m * groups); }
    m += A.m_base;
    uint64_t j;
    uint32_t d;
    ratecv_index(A, m, j, d);
    const size_t cur_at = (size_t)j * A.nch + (size_t)cg * VEC;
    vec_t cur, prev, res;
    if (VEC == 1) cur[0] = in[cur_at]; else cur = *reinterpret_cast<const vec_t*>(in + cur_at);
    if (j && d) {
        if (VEC == 1) prev[0] = in[cur_at - A.nch]; else prev = *reinterpret_cast<const vec_t*>(in + cur_at - A.nch);
    } else {
#pragma unroll
        for (int c = 0; c < VEC; ++c) prev[c] = (T)0;
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) res[c] = ratecv_sample<T, MODE>(prev[c], cur[c], d, A);
    const size_t out_at = (size_t)m * A.nch + (size_t)cg * VEC;
    // (streaming store: +4 % on the 8-channel rows; the frames-per-thread kernels below lose with it -- stereo float32 -14 %)
    if (VEC == 1) out[out_at] = res[0]; else __builtin_nontemporal_store(res, reinterpret_cast<vec_t*>(out + out_at));
}

// Few channels (nch == VEC): one thread = FR consecutive output frames x all channels, so that the store
// is one 8..16-byte vector even for mono 16-bit PCM (a 2-byte store per lane reaches ~1/4 of the bandwidth).
template <typename T, int VEC, int FR, int MODE>
__global__ __launch_bounds__(256) void k_resample_frames(const T* __restrict__ in, T* __restrict__ out, RatecvArgs A, uint64_t out_frames) {
    typedef T vec_t __attribute__((ext_vector_type(VEC * FR)));
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t m0 = A.m_base + u * FR;
    if (m0 >= out_frames) return;
    vec_t res;
    // position of frame m0 in input frames: q + r/outr (exact), then (step_q, step_r) per output frame
    uint64_t q;
    uint32_t r;
    {
        uint64_t j0;
        uint32_t d0;
        ratecv_index(A, m0, j0, d0);
        r = d0 ? A.outr - d0 : 0u;
        q = j0 - (r != 0);
    }
    const uint64_t last_q_frames = out_frames - 1 - m0;           // frames after m0 that exist
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const uint64_t j = q + (r != 0);
        const uint32_t d = r ? A.outr - r : 0u;
        if ((uint64_t)f < last_q_frames) {                        // advance, but never past the last output frame
            r += A.step_r;
            q += A.step_q;
            if (r >= A.outr) { r -= A.outr; q += 1; }
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const T cur = in[j * VEC + c];
            const T prev = (j && d) ? in[(j - 1) * VEC + c] : (T)0;
            res[f * VEC + c] = ratecv_sample<T, MODE>(prev, cur, d, A);
        }
    }
    if (m0 + FR <= out_frames) {
        *reinterpret_cast<vec_t*>(out + m0 * VEC) = res;
    } else {
        for (int f = 0; f < FR && m0 + f < out_frames; ++f)
            for (int c = 0; c < VEC; ++c) out[(m0 + f) * VEC + c] = res[f * VEC + c];
    }
}

// Mono / stereo: the input span of a workgroup is staged in LDS with aligned 16-byte loads (coalesced, every
// input byte fetched once), and each thread interpolates FR consecutive output frames from LDS -- instead of
// 2*FR narrow gathers per thread.  Used when the span fits the LDS budget (ratios up to ~10:1).
constexpr uint32_t RS_LDS_BYTES = 48 * 1024;

template <typename T, int VEC, int FR, int MODE>
__global__ __launch_bounds__(256) void k_resample_lds(const T* __restrict__ in, T* __restrict__ out, RatecvArgs A,
                                                      uint64_t in_frames, uint64_t out_frames) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* lds = reinterpret_cast<T*>(smem);
    typedef T vec_t __attribute__((ext_vector_type(VEC * FR)));
    typedef T ld_t __attribute__((ext_vector_type(16 / sizeof(T))));
    constexpr uint32_t EPV = 16 / sizeof(T);                      // elements per 16-byte vector
    const uint64_t m_first = A.m_base + (uint64_t)blockIdx.x * (256 * FR);
    if (m_first >= out_frames) return;
    uint64_t m_last = m_first + 256 * FR - 1;
    if (m_last > out_frames - 1) m_last = out_frames - 1;
    // input frames this workgroup reads: [j(m_first) - 1, j(m_last)]   (uniform)
    uint64_t jf, jl;
    uint32_t df, dl;
    ratecv_index(A, m_first, jf, df);
    ratecv_index(A, m_last, jl, dl);
    const uint64_t lo_frame = jf ? jf - 1 : 0;
    const uint64_t lo_elem = (lo_frame * VEC) & ~(uint64_t)(EPV - 1);            // 16-byte aligned start
    const uint64_t hi_elem = (jl + 1) * VEC;                                      // exclusive
    const uint64_t total_elems = in_frames * VEC;
    const uint32_t nvec = (uint32_t)((hi_elem - lo_elem + EPV - 1) / EPV);
    for (uint32_t v = threadIdx.x; v < nvec; v += 256) {
        const uint64_t e = lo_elem + (uint64_t)v * EPV;
        if (e + EPV <= total_elems) {
            reinterpret_cast<ld_t*>(lds)[v] = *reinterpret_cast<const ld_t*>(in + e);
        } else {
            for (uint32_t k = 0; k < EPV; ++k) lds[v * EPV + k] = (e + k < total_elems) ? in[e + k] : (T)0;
        }
    }
    __syncthreads();
    const uint64_t m0 = m_first + (uint64_t)threadIdx.x * FR;
    if (m0 >= out_frames) return;
    uint64_t q;
    uint32_t r;
    {
        uint64_t j0;
        uint32_t d0;
        ratecv_index(A, m0, j0, d0);
        r = d0 ? A.outr - d0 : 0u;
        q = j0 - (r != 0);
    }
    const uint64_t frames_after = out_frames - 1 - m0;
    vec_t res;
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const uint64_t j = q + (r != 0);
        const uint32_t d = r ? A.outr - r : 0u;
        if ((uint64_t)f < frames_after) {
            r += A.step_r;
            q += A.step_q;
            if (r >= A.outr) { r -= A.outr; q += 1; }
        }
        const uint32_t at = (uint32_t)(j * VEC - lo_elem);                        // LDS element index of frame j
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const T cur = lds[at + c];
            const T prev = (j && d) ? lds[at - VEC + c] : (T)0;
            res[f * VEC + c] = ratecv_sample<T, MODE>(prev, cur, d, A);
        }
    }
    if (m0 + FR <= out_frames) {
        *reinterpret_cast<vec_t*>(out + m0 * VEC) = res;
    } else {
        for (int f = 0; f < FR && m0 + f < out_frames; ++f)
            for (int c = 0; c < VEC; ++c) out[(m0 + f) * VEC + c] = res[f * VEC + c];
    }
}

// One thread's FR output frames of k_resample_small, from the staged span (LDS element 0 = input element lo_elem).  GROUPS == 2:
// the thread's frames are two runs of FR/2, 256*FR/2 frames apart, so that each of its stores is one 16-byte vector NEXT to its
// neighbour lanes' (a wave's store instruction then writes 1 KB of consecutive bytes; with one run of 16 16-bit frames a lane's two
// 16-byte stores interleave with its neighbours' at a 32-byte stride).
template <typename T, int VEC, int FR, int GROUPS = 1>
__device__ __forceinline__ void resample_small_frames(const unsigned char* smem, T* __restrict__ out, const RatecvArgs& A, uint64_t m_first,
                                                      uint64_t q0, uint32_t r0, uint64_t lo_elem, uint64_t out_frames) {
    const T* lds = reinterpret_cast<const T*>(smem);
    constexpr int FRG = FR / GROUPS;
    typedef T vec_t __attribute__((ext_vector_type(VEC * FRG)));
    constexpr int HALF = 1 << (8 * (int)sizeof(T) - 1);
    uint64_t m0 = m_first + (uint64_t)threadIdx.x * FRG;
    if (m0 >= out_frames) return;
    // this thread's first frame: (q0, r0) advanced by threadIdx.x*FR output frames (host guarantees < 2^31)
    uint32_t r, qe_elem;
    {
        const uint32_t tot = r0 + __umul24(threadIdx.x * FRG, A.inr);
        const uint32_t dq = (uint32_t)fma((double)tot, A.inv_outr, 0.5 * A.inv_outr);   // floor(tot/outr), exact as below
        r = tot - dq * A.outr;
        qe_elem = (uint32_t)(q0 * VEC - lo_elem) + dq * VEC;                     // LDS element index of frame q
    }
    const uint32_t step_elem = A.step_q * VEC;
    const double half_inv = 0.5 * A.inv_outr;
    constexpr bool PAIR_IN_DWORD = 2 * VEC * sizeof(T) <= 4;
    constexpr uint32_t MASK = (1u << (8 * sizeof(T))) - 1u;
    constexpr uint32_t FLIP = sizeof(T) == 2 ? 0x80008000u : 0x80808080u;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
    vec_t res;
#pragma unroll
    for (int f = 0; f < FRG; ++f) {
        uint32_t pair = 0;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            // ua = a + HALF, ub = b + HALF as unsigned bit patterns (x + HALF == x ^ HALF on the sample width)
            uint32_t ua, ub;
            if (PAIR_IN_DWORD) {
                // frames q and q+1 together are <= 4 bytes at a 1- or 2-byte aligned address: two aligned dwords and a
                // funnel shift (a misaligned ds_read_b32 is several times slower than the extra three instructions)
                if (c == 0) {
                    const uint32_t byte_off = qe_elem * (uint32_t)sizeof(T);
                    const uint32_t* l32 = reinterpret_cast<const uint32_t*>(smem) + (byte_off >> 2);
                    pair = __builtin_amdgcn_alignbit(l32[1], l32[0], (byte_off & 3u) * 8u) ^ FLIP;
                }
                ua = (pair >> (8 * sizeof(T) * c)) & MASK;
                ub = (pair >> (8 * sizeof(T) * (VEC + c))) & MASK;
            } else {
                ua = ((uint32_t)lds[qe_elem + c] & MASK) ^ (uint32_t)HALF;
                ub = ((uint32_t)lds[qe_elem + VEC + c] & MASK) ^ (uint32_t)HALF;
            }
            const uint32_t u = (uint32_t)__mul24((int)ub - (int)ua, (int)r) + __umul24(ua, A.outr);
            // (u + 1/2)/outr is at least 1/(2 outr) > 2^-17 away from every integer and the float64 evaluation is off by
            // less than 2^-20, so the truncation is floor(u/outr) exactly: 3 instructions, no correction step
            const uint32_t q = (uint32_t)fma((double)u, A.inv_outr, half_inv);
            res[f * VEC + c] = (T)(q ^ (uint32_t)HALF);
        }
        r += A.step_r;
        const bool wrap = r >= A.outr;
        r -= wrap ? A.outr : 0u;
        qe_elem += step_elem + (wrap ? (uint32_t)VEC : 0u);
    }
    if (m0 + FRG <= out_frames) {
        __builtin_nontemporal_store(res, reinterpret_cast<vec_t*>(out + m0 * VEC));
    } else {
        for (int f = 0; f < FRG && m0 + f < out_frames; ++f)
            for (int c = 0; c < VEC; ++c) out[(m0 + f) * VEC + c] = res[f * VEC + c];
    }
    if (g + 1 < GROUPS) {
        // on to the thread's next run: 255 * FRG frames further (the loop above has moved FRG already)
        m0 += 256 * FRG;
        if (m0 >= out_frames) return;
        const uint32_t tot = r + (uint32_t)(255 * FRG) * A.inr;
        const uint32_t dq = (uint32_t)fma((double)tot, A.inv_outr, half_inv);
        r = tot - dq * A.outr;
        qe_elem += dq * VEC;
    }
    }
}

// 8/16-bit PCM, few channels, reduced rates below 65536 (the common Sample.resample case: 16-bit mono/stereo
// between 44.1k/48k/96k).  The generic kernels above are VALU-issue-bound there (~55 instructions per output
// sample at 2-4 bytes of traffic each), so this one strips the arithmetic to ~20 full-rate instructions:
//  * the workgroup's input span goes through LDS (aligned 16-byte loads), positions are 32-bit LDS-relative;
//  * output m sits at input position q + r/outr; with a = x[q], b = x[q+1] the reference's expression is
//    M = a*(outr-r) + b*r for every r (r == 0 gives cur = x[q], weight outr), so there is no prev/cur select;
//  * u = M + HALF*outr = (b-a)*r + (a+HALF)*outr in 24-bit multiplies (mod 2^32; 0 <= u < 2^32);
//  * floor(u/outr) = trunc(fma(u, 1/outr, 1/(2 outr))) in float64, exact without a correction step.  See
//    ratecv_small_int for why the floor equals audioop's float64 expression.
// (Measured and dropped, bit-identical both: the interpolation as ONE v_dot2_u32_u16 on packed weights -- 0.395 vs 0.391 ms on 900 MB;
// the output frames dealt to the lanes, no LDS bank conflicts and 16 instead of 20 instructions per sample -- not faster either:
// CHANGELOG items 39 and 22; profiles/r03_summary.md.)
template <typename T, int VEC, int FR, int GROUPS = 1>
__global__ __launch_bounds__(256) void k_resample_small(const T* __restrict__ in, T* __restrict__ out, RatecvArgs A,
                                                        uint64_t in_frames, uint64_t out_frames, uint32_t span_vecs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* lds = reinterpret_cast<T*>(smem);
    typedef T ld_t __attribute__((ext_vector_type(16 / sizeof(T))));
    constexpr uint32_t EPV = 16 / sizeof(T);
    const uint64_t m_first = A.m_base + (uint64_t)blockIdx.x * (256 * FR);
    if (m_first >= out_frames) return;
    // position of the workgroup's first output frame: q0 + r0/outr   (uniform)
    uint64_t q0;
    uint32_t r0;
    {
        uint64_t jf;
        uint32_t df;
        ratecv_index(A, m_first, jf, df);
        r0 = df ? A.outr - df : 0u;
        q0 = jf - (r0 != 0);
    }
    const uint64_t lo_elem = (q0 * VEC) & ~(uint64_t)(EPV - 1);                  // 16-byte aligned start of the span
    const uint64_t total_elems = in_frames * VEC;
    for (uint32_t v = threadIdx.x; v < span_vecs; v += 256) {
        const uint64_t e = lo_elem + (uint64_t)v * EPV;
        if (e + EPV <= total_elems) {
            // streaming on both sides (input read once per workgroup, output never re-read): +1..3 % on the 16-bit rows
            reinterpret_cast<ld_t*>(lds)[v] = __builtin_nontemporal_load(reinterpret_cast<const ld_t*>(in + e));
        } else {
            for (uint32_t k = 0; k < EPV; ++k) lds[v * EPV + k] = (e + k < total_elems) ? in[e + k] : (T)0;
        }
    }
    __syncthreads();
    resample_small_frames<T, VEC, FR, GROUPS>(smem, out, A, m_first, q0, r0, lo_elem, out_frames);
}

// ---- 16-bit mono between rates with a SHORT period (44.1k <-> 48k <-> 96k ...: reduced outrate <= 2048) --------------------------------
// k_resample_small needs ~23 VALU instructions per output sample (the position's remainder stepped and wrapped, the frame pair cut out of
// two dwords, the weights), and with its 78 % of the VALU slots taken it sits between its two roofs: 0.68 of HBM whatever one of those
// instructions is replaced by (five variants: profiles/r06_resample_ab.txt).  But output frame m and m + outr lie at the same fraction
// r / outr, inr input frames apart: with CHUNKS of K whole periods (L = K outr output frames, K inr input frames, starting at remainder 0)
// thread t's sixteen frames of EVERY chunk have the same weights (outr - r, r) and the same offsets into the chunk's input span.  The
// workgroups stay (chunk C, C + grid, ...), a thread works its sixteen (weights, offset) pairs out ONCE, and a sample is
//      an address (offset + where the span starts in its first 16-byte vector), two sign-extending 16-bit LDS reads,
//      u = a (outr - r) + b r + 65536 outr in two 24-bit multiply-adds, floor(u / outr) by ratecv_small_int's float64 step (3 instructions) --
// whose low sixteen bits ARE the sample (floor(M / outr) + 65536: no bias to take off again) -- 6.5 instructions instead of 23.
// The same integers as k_resample_small's, i.e. audioop.ratecv's (tests/test_gpu_pcm.py against the live module).
struct PeriodArgs {
    uint64_t c0, c1;             // chunks [c0, c1), absolute: chunk C = output frames [C L, (C + 1) L) = input frames from C kinr on.  The host
                                 // passes INTERIOR chunks only -- span wholly inside the held input, frames wholly inside the launch's range --
                                 // so the kernel tests nothing; what lies in front of and behind them goes through k_resample_small
    uint32_t L, kinr;            // output / input frames per chunk (K periods); L is a multiple of 8: 16-byte stores
    uint32_t inr, outr;
    uint32_t span_vecs;          // 16-byte vectors staged per chunk
    uint32_t per_wg;             // consecutive chunks per workgroup
    double   inv_outr;
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// NV: 16-byte vectors of the span per thread (span_vecs <= 256 NV).  A workgroup works through per_wg CONSECUTIVE chunks and ends; the
// span of the next chunk is loaded into registers while this one is worked on.  What the shape of the loop is for
// (profiles/r06_resample_period.txt):
//  * this chip counts loads and stores in ONE in-order counter (vmcnt): "wait for my loads" also waits for every store issued before
//    them.  The next span's loads are therefore issued BEFORE the chunk's stores and waited for at the END of the turn, where the wait
//    the compiler inserts is vmcnt(stores of this turn): the stores stay in flight across the barrier.  That needs a turn without a
//    branch around a memory instruction (a path with fewer stores, and the count drops to zero): lanes beyond the span load its last
//    vector again, lanes beyond the chunk's last run do that run again (same frames, same values, same address), and L is a multiple
//    of 8, so a run is whole or absent;
//  * workgroups that STAY for the whole call (chunks C, C + grid, ...) march in step -- all load, all compute, all store -- and reach 0.63-0.70
//    of HBM on the 44.1 -> 48 kHz row where workgroups of ONE chunk each (dispatched as others end, their phases mixed) reach 0.72;
//    but one chunk per workgroup pays the sixteen (weights, offset) set-ups for sixteen samples (upsampling 44.1 -> 96 kHz: 0.60
//    against 0.70).  A few chunks per workgroup keep both.
// VEC: channels (1: mono, a run = 8 frames; 2: stereo, a run = 4 frames -- 16 bytes either way).  All positions below are in FRAMES; a frame
// is VEC shorts.
template <int VEC, int NV>
__global__ __launch_bounds__(256) void k_resample_period_i16(const short* __restrict__ in, short* __restrict__ out, PeriodArgs P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FPR = 8 / VEC;                      // frames per run
    constexpr uint32_t RUN2 = 256u * FPR;             // the second run lies this many frames behind the first: a wave's store instruction writes 1 KB of consecutive bytes
    const uint32_t t = threadIdx.x;
    const double half_inv = 0.5 * P.inv_outr;
    // the thread's frames of a chunk: two runs (the last run of each half that lies inside the chunk: L >= RUN2 + FPR, a multiple of FPR)
    uint32_t w0[2 * FPR], w1[2 * FPR], ob[2 * FPR];
    const uint32_t last0 = P.L / FPR - 1u, last1 = (P.L - RUN2) / FPR - 1u;
    const uint32_t run0[2] = {(uint32_t)FPR * (t < last0 ? t : last0), RUN2 + (uint32_t)FPR * (t < last1 ? t : last1)};
#pragma unroll
    for (int k = 0; k < 2 * FPR; ++k) {
        const uint32_t e = __umul24(run0[k / FPR] + (uint32_t)(k % FPR), P.inr);  // < 2^12 * 2^16
        const uint32_t dq = (uint32_t)fma((double)e, P.inv_outr, half_inv);        // floor(e / outr), exact (ratecv_small_int)
        const uint32_t r = e - dq * P.outr;
        w0[k] = P.outr - r;
        w1[k] = r;
        ob[k] = dq * (uint32_t)(2 * VEC);              // bytes
    }
    const int acc = (int)(65536u * P.outr);
    // (the LDS address of the staged span for the hand-written reads below: 0 in this kernel -- it has no other shared memory -- but asked for, not assumed)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint64_t C = P.c0 + (uint64_t)blockIdx.x * P.per_wg;
    if (C >= P.c1) return;
    const uint64_t c_end = C + P.per_wg < P.c1 ? C + P.per_wg : P.c1;
    short8v pre[NV];
    uint32_t vi[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) vi[i] = t + 256u * i < P.span_vecs ? t + 256u * i : P.span_vecs - 1u;
    {
        const short8v* __restrict__ src = reinterpret_cast<const short8v*>(in + ((C * (uint64_t)P.kinr * VEC) & ~(uint64_t)7));
#pragma unroll
        for (int i = 0; i < NV; ++i) pre[i] = __builtin_nontemporal_load(src + vi[i]);
#pragma unroll
        for (int i = 0; i < NV; ++i) reinterpret_cast<short8v*>(smem)[vi[i]] = pre[i];
    }
    for (;;) {
        __syncthreads();                                   // the chunk's span is in LDS
        const bool more = C + 1 < c_end;                   // (uniform)
        if (more) {
            const short8v* __restrict__ src = reinterpret_cast<const short8v*>(in + (((C + 1) * (uint64_t)P.kinr * VEC) & ~(uint64_t)7));
            static_for<0, NV>([&](auto i_) { constexpr int i = decltype(i_)::value; pre[i] = __builtin_nontemporal_load(src + vi[i]); });
        }
        const uint32_t rel0b = (uint32_t)((C * (uint64_t)P.kinr * VEC) & 7) * 2u;
        short* __restrict__ outC = out + C * (uint64_t)P.L * VEC;
        auto run = [&](auto g_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            // The run's frame pairs (a, b) out of LDS, by hand.  Mono: two sign-extending 16-bit reads per frame -- written as plain loads the
            // compiler fuses each pair into ONE ds_read_b32 at a 2-byte-aligned address, which the LDS serves at a fraction of the rate
            // (0.87 ms for the 900 MB row against 0.44 for k_resample_small); `volatile` loads become FLAT loads; and the D16 forms that
            // would fill the halves of one register for a v_dot2 clear the other half on this chip (SRAM ECC).  Stereo: frames are dwords,
            // one ds_read2_b32 per pair.  The reads are waited for inside the statement (the compiler's counters do not see them).
            uint32_t at[FPR];
#pragma unroll
            for (int f = 0; f < FPR; ++f) at[f] = ob[FPR * g + f] + rel0b + lds0;
            short8v res;
            if constexpr (VEC == 1) {
                int a[8], b[8];
                asm volatile(
                    "ds_read_i16 %0, %16\n\tds_read_i16 %8, %16 offset:2\n\t"
                    "ds_read_i16 %1, %17\n\tds_read_i16 %9, %17 offset:2\n\t"
                    "ds_read_i16 %2, %18\n\tds_read_i16 %10, %18 offset:2\n\t"
                    "ds_read_i16 %3, %19\n\tds_read_i16 %11, %19 offset:2\n\t"
                    "ds_read_i16 %4, %20\n\tds_read_i16 %12, %20 offset:2\n\t"
                    "ds_read_i16 %5, %21\n\tds_read_i16 %13, %21 offset:2\n\t"
                    "ds_read_i16 %6, %22\n\tds_read_i16 %14, %22 offset:2\n\t"
                    "ds_read_i16 %7, %23\n\tds_read_i16 %15, %23 offset:2\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]),
                      "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7])
                    : "v"(at[0]), "v"(at[1]), "v"(at[2]), "v"(at[3]), "v"(at[4]), "v"(at[5]), "v"(at[6]), "v"(at[7])
                    : "memory");
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    const uint32_t u = (uint32_t)__mul24(a[f], (int)w0[8 * g + f]) + (uint32_t)(__mul24(b[f], (int)w1[8 * g + f]) + acc);
                    res[f] = (short)(uint32_t)fma((double)u, P.inv_outr, half_inv);
                }
            } else {
                uint64_t ab[4];                            // low dword: frame q (L | R << 16), high dword: frame q + 1
                asm volatile(
                    "ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %5 offset1:1\n\t"
                    "ds_read2_b32 %2, %6 offset1:1\n\tds_read2_b32 %3, %7 offset1:1\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(ab[0]), "=&v"(ab[1]), "=&v"(ab[2]), "=&v"(ab[3])
                    : "v"(at[0]), "v"(at[1]), "v"(at[2]), "v"(at[3])
                    : "memory");
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int fa = (int)(uint32_t)ab[f], fb = (int)(uint32_t)(ab[f] >> 32);
                    const int la = (int)(short)fa, ra = fa >> 16, lb = (int)(short)fb, rb = fb >> 16;
                    const int k0 = (int)w0[4 * g + f], k1 = (int)w1[4 * g + f];
                    const uint32_t ul = (uint32_t)__mul24(la, k0) + (uint32_t)(__mul24(lb, k1) + acc);
                    const uint32_t ur = (uint32_t)__mul24(ra, k0) + (uint32_t)(__mul24(rb, k1) + acc);
                    res[2 * f] = (short)(uint32_t)fma((double)ul, P.inv_outr, half_inv);
                    res[2 * f + 1] = (short)(uint32_t)fma((double)ur, P.inv_outr, half_inv);
                }
            }
            __builtin_nontemporal_store(res, reinterpret_cast<short8v*>(outC + (size_t)run0[g] * VEC));
        };
        run(std::integral_constant<int, 0>{});
        run(std::integral_constant<int, 1>{});
        __syncthreads();                                   // everybody has read the span
        if (!more) break;
        // (waits for the loads; this turn's stores stay in flight)
        static_for<0, NV>([&](auto i_) { constexpr int i = decltype(i_)::value; reinterpret_cast<short8v*>(smem)[vi[i]] = pre[i]; });
        C += 1;
    }
}

uint64_t gcd_u64(uint64_t a, uint64_t b) {
    while (b) {
        uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

int fetch_flag(int* result) {
    sh::State& s = sh::state();
    SH_HIP(hipMemcpyAsync(s.flag_host, s.flag, sizeof(int), hipMemcpyDeviceToHost, s.stream));
    SH_HIP(hipStreamSynchronize(s.stream));
    *result = s.flag_host[0];
    if (*result) {
        SH_HIP(hipMemsetAsync(s.flag, 0, sizeof(int), s.stream));
    }
    return SH_OK;
}

}  // namespace

extern "C" {

int sh_quantize_f32(const sh_buf* in_f32, size_t in_off, size_t n, double scale, int width,
                    sh_buf* out_pcm, size_t out_off) {
    SH_REQUIRE_INIT();
    if (!in_f32 || !out_pcm) return sh::set_error(SH_ERR_INVALID, "sh_quantize_f32: NULL argument");
    if (width != 1 && width != 2 && width != 4) return sh::set_error(SH_ERR_INVALID, "sh_quantize_f32: width %d not in {1,2,4}", width);
    if (in_off > in_f32->bytes / 4 || n > in_f32->bytes / 4 - in_off) return sh::set_error(SH_ERR_INVALID, "sh_quantize_f32: input range outside buffer");
    if (out_off > out_pcm->bytes / width || n > out_pcm->bytes / width - out_off) return sh::set_error(SH_ERR_INVALID, "sh_quantize_f32: output range outside buffer");
    if (!n) return SH_OK;
    const double lo = -ldexp(1.0, 8 * width - 1), hi = ldexp(1.0, 8 * width - 1) - 1.0;
    const float* in = (const float*)in_f32->ptr + in_off;
    const dim3 grid = sh::grid1d(n, 256);
    hipStream_t st = sh::state().stream;
    int* flag = sh::state().flag;
    const int rnd = sh::state().quantise_round;
    if (width == 2) {
        short* o = (short*)out_pcm->ptr + out_off;
        const bool aligned = ((uintptr_t)in & 15) == 0 && ((uintptr_t)o & 7) == 0;
        const size_t nvec = aligned ? n / 4 : 0, done = nvec * 4;
        if (nvec) hipLaunchKernelGGL(k_quantize_f32_i16_vec, sh::grid1d(nvec, 512), dim3(256), 0, st, (const float4v*)in, nvec, scale, (short4v*)o, flag, 0, rnd);
        if (n > done) hipLaunchKernelGGL(k_quantize<short>, sh::grid1d(n - done, 256), dim3(256), 0, st, in + done, n - done, scale, lo, hi, o + done, flag, 0, rnd);
    }
    else if (width == 1) hipLaunchKernelGGL(k_quantize<signed char>, grid, dim3(256), 0, st, in, n, scale, lo, hi, (signed char*)out_pcm->ptr + out_off, flag, 0, rnd);
    else hipLaunchKernelGGL(k_quantize<int>, grid, dim3(256), 0, st, in, n, scale, lo, hi, (int*)out_pcm->ptr + out_off, flag, 0, rnd);
    SH_CHECK_LAUNCH("k_quantize");
    int overflow = 0;
    int rc = fetch_flag(&overflow);
    if (rc) return rc;
    if (overflow) return sh::set_error(SH_ERR_OVERFLOW, "signed integer out of range for sample width %d", width);
    return SH_OK;
}

int sh_quantize_f64(const sh_buf* in_f64, size_t in_off, size_t n, double scale, int width,
                    sh_buf* out_pcm, size_t out_off) {
    SH_REQUIRE_INIT();
    if (!in_f64 || !out_pcm) return sh::set_error(SH_ERR_INVALID, "sh_quantize_f64: NULL argument");
    if (width != 1 && width != 2 && width != 4) return sh::set_error(SH_ERR_INVALID, "sh_quantize_f64: width %d not in {1,2,4}", width);
    if (in_off > in_f64->bytes / 8 || n > in_f64->bytes / 8 - in_off) return sh::set_error(SH_ERR_INVALID, "sh_quantize_f64: input range outside buffer");
    if (out_off > out_pcm->bytes / width || n > out_pcm->bytes / width - out_off) return sh::set_error(SH_ERR_INVALID, "sh_quantize_f64: output range outside buffer");
    if (!n) return SH_OK;
    const double lo = -ldexp(1.0, 8 * width - 1), hi = ldexp(1.0, 8 * width - 1) - 1.0;
    const double* in = (const double*)in_f64->ptr + in_off;
    const dim3 grid = sh::grid1d(n, 256);
    hipStream_t st = sh::state().stream;
    int* flag = sh::state().flag;
    const int rnd = sh::state().quantise_round;
    if (width == 2) {
        short* out = (short*)out_pcm->ptr + out_off;
        const bool aligned = ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 3) == 0;
        const size_t nvec = aligned ? n / 2 : 0, done = nvec * 2;
        if (nvec) hipLaunchKernelGGL(k_quantize_f64_i16_vec, sh::grid1d(nvec, 512), dim3(256), 0, st, (const double2v*)in, nvec, scale, (short2v*)out, flag, rnd);
        if (n > done) hipLaunchKernelGGL((k_quantize<short, double>), sh::grid1d(n - done, 256), dim3(256), 0, st, in + done, n - done, scale, lo, hi, out + done, flag, 0, rnd);
    } else if (width == 1) hipLaunchKernelGGL((k_quantize<signed char, double>), grid, dim3(256), 0, st, in, n, scale, lo, hi, (signed char*)out_pcm->ptr + out_off, flag, 0, rnd);
    else hipLaunchKernelGGL((k_quantize<int, double>), grid, dim3(256), 0, st, in, n, scale, lo, hi, (int*)out_pcm->ptr + out_off, flag, 0, rnd);
    SH_CHECK_LAUNCH("k_quantize");
    int overflow = 0;
    int rc = fetch_flag(&overflow);
    if (rc) return rc;
    if (overflow) return sh::set_error(SH_ERR_OVERFLOW, "signed integer out of range for sample width %d", width);
    return SH_OK;
}

int sh_quantize_clip_f32(const sh_buf* in_f32, size_t n, double scale, sh_buf* out_i16) {
    SH_REQUIRE_INIT();
    if (!in_f32 || !out_i16) return sh::set_error(SH_ERR_INVALID, "sh_quantize_clip_f32: NULL argument");
    if (in_f32->bytes / 4 < n || out_i16->bytes / 2 < n) return sh::set_error(SH_ERR_INVALID, "sh_quantize_clip_f32: buffer too small");
    if (!n) return SH_OK;
    {
        hipStream_t st = sh::state().stream;
        const size_t nvec = n / 4, done = nvec * 4;
        if (nvec) hipLaunchKernelGGL(k_quantize_f32_i16_vec, sh::grid1d(nvec, 512), dim3(256), 0, st, (const float4v*)in_f32->ptr, nvec, scale, (short4v*)out_i16->ptr, sh::state().flag, 1);
        if (n > done) hipLaunchKernelGGL(k_quantize<short>, sh::grid1d(n - done, 256), dim3(256), 0, st,
                                         (const float*)in_f32->ptr + done, n - done, scale, -32768.0, 32767.0, (short*)out_i16->ptr + done, sh::state().flag, 1);
    }
    SH_CHECK_LAUNCH("k_quantize(clip)");
    return SH_OK;
}

static int pcm_add_dev(const char* a, const char* b, char* o, size_t nbytes, int width) {
    hipStream_t st = sh::state().stream;
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)o) & 15) == 0;
    size_t nvec = aligned ? nbytes / 16 : 0;
    size_t done = nvec * 16;
    if (nvec) {
        const dim3 grid = sh::grid1d(nvec, 256);
        const bool stream = 2 * nbytes > sh::STREAM_BYTES;
#define SH_ADD(V_) do { if (stream) hipLaunchKernelGGL((k_add_vec<V_, true>), grid, dim3(256), 0, st, (const V_*)a, (const V_*)b, (V_*)o, nvec); \
                        else hipLaunchKernelGGL((k_add_vec<V_, false>), grid, dim3(256), 0, st, (const V_*)a, (const V_*)b, (V_*)o, nvec); } while (0)
        if (width == 2) SH_ADD(short8v);
        else if (width == 4) SH_ADD(int4v);
        else SH_ADD(char16v);
#undef SH_ADD
        SH_CHECK_LAUNCH("k_add_vec");
    }
    size_t rest = (nbytes - done) / width;
    if (rest) {
        const dim3 grid = sh::grid1d(rest, 256);
        if (width == 2) hipLaunchKernelGGL(k_add_scalar<short>, grid, dim3(256), 0, st, (const short*)(a + done), (const short*)(b + done), (short*)(o + done), rest);
        else if (width == 4) hipLaunchKernelGGL(k_add_scalar<int>, grid, dim3(256), 0, st, (const int*)(a + done), (const int*)(b + done), (int*)(o + done), rest);
        else hipLaunchKernelGGL(k_add_scalar<signed char>, grid, dim3(256), 0, st, (const signed char*)(a + done), (const signed char*)(b + done), (signed char*)(o + done), rest);
        SH_CHECK_LAUNCH("k_add_scalar");
    }
    return SH_OK;
}

int sh_pcm_add(const sh_buf* a, size_t a_off, const sh_buf* b, size_t b_off, size_t nbytes, int width,
               sh_buf* out, size_t out_off) {
    SH_REQUIRE_INIT();
    if (!a || !b || !out) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add: NULL argument");
    if (width == 3) {
        // audioop.add at width 3: int sum clamped to [-2^23, 2^23 - 1] == the 32-bit saturating add of the samples << 8, >> 8
        if (nbytes % 3 || (a_off | b_off | out_off) % 3) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add: not a whole number of frames");
        if (a_off > a->bytes || nbytes > a->bytes - a_off || b_off > b->bytes || nbytes > b->bytes - b_off ||
            out_off > out->bytes || nbytes > out->bytes - out_off)
            return sh::set_error(SH_ERR_LENGTH, "sh_pcm_add: range outside buffer (Lengths should be the same)");
        const size_t n = nbytes / 3;
        if (!n) return SH_OK;
        sh::Temp ta, tb;
        int rc = ta.alloc(n * 4);
        if (!rc) rc = tb.alloc(n * 4);
        if (!rc) rc = sh::unpack24((const char*)a->ptr + a_off, n, 8, (int32_t*)ta.buf.ptr);
        if (!rc) rc = sh::unpack24((const char*)b->ptr + b_off, n, 8, (int32_t*)tb.buf.ptr);
        if (!rc) rc = pcm_add_dev((const char*)ta.buf.ptr, (const char*)tb.buf.ptr, (char*)ta.buf.ptr, n * 4, 4);
        if (!rc) rc = sh::pack24((const int32_t*)ta.buf.ptr, n, 8, (char*)out->ptr + out_off);
        return rc;
    }
    if (width != 1 && width != 2 && width != 4) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add: width %d not in {1,2,4}", width);
    if (nbytes % width) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add: not a whole number of frames");
    if (a_off > a->bytes || nbytes > a->bytes - a_off || b_off > b->bytes || nbytes > b->bytes - b_off ||
        out_off > out->bytes || nbytes > out->bytes - out_off)
        return sh::set_error(SH_ERR_LENGTH, "sh_pcm_add: range outside buffer (Lengths should be the same)");
    if ((a_off | b_off | out_off) % width) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add: offsets not sample-aligned");
    if (!nbytes) return SH_OK;
    return pcm_add_dev((const char*)a->ptr + a_off, (const char*)b->ptr + b_off, (char*)out->ptr + out_off, nbytes, width);
}

int sh_pcm_add_host(const void* a, const void* b, size_t nbytes, int width, void* out) {
    SH_REQUIRE_INIT();
    if (width == 3) {                                       // staged in device temporaries, then the device form (24-bit via 32-bit)
        if (nbytes % 3) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add_host: not a whole number of frames");
        if (!nbytes) return SH_OK;
        if (!a || !b || !out) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add_host: NULL argument");
        sh::Temp da, db;
        int rc3 = da.alloc(nbytes);
        if (!rc3) rc3 = db.alloc(nbytes);
        if (rc3) return rc3;
        hipStream_t st3 = sh::state().stream;
        SH_HIP(hipMemcpyAsync(da.buf.ptr, a, nbytes, hipMemcpyHostToDevice, st3));
        SH_HIP(hipMemcpyAsync(db.buf.ptr, b, nbytes, hipMemcpyHostToDevice, st3));
        rc3 = sh_pcm_add(&da.buf, 0, &db.buf, 0, nbytes, 3, &da.buf, 0);
        if (rc3) return rc3;
        SH_HIP(hipMemcpyAsync(out, da.buf.ptr, nbytes, hipMemcpyDeviceToHost, st3));
        SH_HIP(hipStreamSynchronize(st3));
        return SH_OK;
    }
    if (width != 1 && width != 2 && width != 4) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add_host: width %d not in {1,2,4}", width);
    if (nbytes % width) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add_host: not a whole number of frames");
    if (!nbytes) return SH_OK;
    if (!a || !b || !out) return sh::set_error(SH_ERR_INVALID, "sh_pcm_add_host: NULL argument");
    size_t pad = (nbytes + 255) & ~size_t(255);
    int rc = sh::ensure_scratch(3 * pad);
    if (rc) return rc;
    char* s = (char*)sh::state().scratch;
    hipStream_t st = sh::state().stream;
    SH_HIP(hipMemcpyAsync(s, a, nbytes, hipMemcpyHostToDevice, st));
    SH_HIP(hipMemcpyAsync(s + pad, b, nbytes, hipMemcpyHostToDevice, st));
    rc = pcm_add_dev(s, s + pad, s + 2 * pad, nbytes, width);
    if (rc) return rc;
    SH_HIP(hipMemcpyAsync(out, s + 2 * pad, nbytes, hipMemcpyDeviceToHost, st));
    SH_HIP(hipStreamSynchronize(st));
    return SH_OK;
}

int sh_mix_chain_i16(const sh_buf* chunks, uint32_t nvoices, size_t stride, uint32_t nsamples, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (!chunks || !out || nvoices == 0) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_i16: NULL argument");
    if (nvoices > 32768) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_i16: at most 32768 voices");
    if (nsamples > 0xFFFF0000u) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_i16: at most 2^32 - 65536 samples per call");
    if (!nsamples) return SH_OK;
    if (stride < nsamples || chunks->bytes / 2 < (size_t)(nvoices - 1) * stride + nsamples)
        return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_i16: chunk buffer too small");
    if (out->bytes / 2 < nsamples) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_i16: output too small");
    hipStream_t st = sh::state().stream;
    // shape by the number of 1 KB columns (= waves when the voices are not split): plenty -> the direct loop
    // (6.4-6.5 TB/s at 1875 columns, where the split kernel's waves of one workgroup fetch the same column of eight
    // distant rows: 5.8); fewer -> split the voices over the waves of a workgroup for parallelism
    const uint32_t columns = (uint32_t)sh::div_up(nsamples, 512);
    const bool aligned = (stride & 7) == 0 && ((uintptr_t)chunks->ptr & 15) == 0 && ((uintptr_t)out->ptr & 15) == 0;
    const bool stream = (size_t)nvoices * nsamples * 2 > sh::STREAM_BYTES;           // rows beyond the Infinity Cache: streaming loads
    // (the split kernel keeps plain loads: 1024 x 96 000 samples = 197 MB ran 13 % slower with streaming ones)
#define SH_CHAIN(W_, C_) hipLaunchKernelGGL((k_mix_chain_i16<W_, C_, false>), sh::grid1d(nsamples, 512 * C_), dim3(W_ * 64), 0, st, \
                                            (const short*)chunks->ptr, nvoices, stride, nsamples, (short*)out->ptr)
#define SH_DIRECT_S(S_, INF_) do { \
        if (stream) hipLaunchKernelGGL((k_mix_chain_direct_s<S_, 4, INF_, true>), sh::grid1d(nsamples, 256 * S_), dim3(256), 0, st, (const short*)chunks->ptr, nvoices, stride, nsamples, (short*)out->ptr); \
        else hipLaunchKernelGGL((k_mix_chain_direct_s<S_, 4, INF_, false>), sh::grid1d(nsamples, 256 * S_), dim3(256), 0, st, (const short*)chunks->ptr, nvoices, stride, nsamples, (short*)out->ptr); } while (0)
    if (columns >= 640 && columns < 1536 && aligned) SH_DIRECT_S(4, 8);      // (measured at 937 columns; 187 columns: the split kernel, 31 against 86 us)
    else if (columns >= 1536 && aligned) {
        if (stream) hipLaunchKernelGGL((k_mix_chain_direct<8, 4, true>), sh::grid1d(nsamples, 512 * 8), dim3(8 * 64), 0, st,
                                       (const short*)chunks->ptr, nvoices, stride, nsamples, (short*)out->ptr);
        else hipLaunchKernelGGL((k_mix_chain_direct<8, 4, false>), sh::grid1d(nsamples, 512 * 8), dim3(8 * 64), 0, st,
                                (const short*)chunks->ptr, nvoices, stride, nsamples, (short*)out->ptr);
    }
#undef SH_DIRECT_S
    else if (nvoices < 64) SH_CHAIN(2, 1);
    else if (columns >= 512) SH_CHAIN(8, 2);
    else SH_CHAIN(8, 1);
#undef SH_CHAIN
    SH_CHECK_LAUNCH("k_mix_chain_i16");
    return SH_OK;
}

int sh_mix_chain_pan_i16(const sh_buf* chunks, uint32_t nvoices, size_t stride, uint32_t nframes, const sh_buf* factors_lr, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (!chunks || !out || !factors_lr || nvoices == 0) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_pan_i16: NULL argument");
    if (nvoices > 32768) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_pan_i16: at most 32768 voices");
    if (nframes > 0x7FFF0000u) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_pan_i16: at most 2^31 - 65536 frames per call");
    if (!nframes) return SH_OK;
    if (stride < nframes || chunks->bytes / 2 < (size_t)(nvoices - 1) * stride + nframes)
        return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_pan_i16: chunk buffer too small");
    if (factors_lr->bytes / 16 < nvoices) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_pan_i16: one (left, right) pair of doubles per voice");
    if (out->bytes / 4 < nframes) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_pan_i16: output too small");
    hipStream_t st = sh::state().stream;
    const short* in = (const short*)chunks->ptr;
    const double2* fac = (const double2*)factors_lr->ptr;
    const uint32_t nsamples = 2 * nframes;
    const uint32_t columns = (uint32_t)sh::div_up(nframes, 512);                      // 1 KB of a mono row
    const bool aligned = (stride & 7) == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out->ptr & 15) == 0 && ((uintptr_t)fac & 15) == 0;
    const bool stream = (size_t)nvoices * nframes * 2 > sh::STREAM_BYTES;
    if (columns >= 640 && aligned) {
        if (stream) hipLaunchKernelGGL((k_mix_chain_pan_direct_s<4, 4, 8, true>), sh::grid1d(nframes, 256 * 4), dim3(256), 0, st, in, nvoices, stride, nframes, fac, (short*)out->ptr);
        else hipLaunchKernelGGL((k_mix_chain_pan_direct_s<4, 4, 8, false>), sh::grid1d(nframes, 256 * 4), dim3(256), 0, st, in, nvoices, stride, nframes, fac, (short*)out->ptr);
    } else if (nvoices < 64) {
        hipLaunchKernelGGL((k_mix_chain_i16<2, 1, false>), sh::grid1d(nsamples, 512), dim3(128), 0, st, in, nvoices, stride, nsamples, (short*)out->ptr, fac);
    } else {
        hipLaunchKernelGGL((k_mix_chain_i16<8, 1, false>), sh::grid1d(nsamples, 512), dim3(512), 0, st, in, nvoices, stride, nsamples, (short*)out->ptr, fac);
    }
    SH_CHECK_LAUNCH("k_mix_chain_pan");
    return SH_OK;
}

}  // extern "C"

namespace {
// Where a gather fold runs: the library's stream with its grow-only scratch (the table of a turn with more than 64 sources), or a
// real-time lane's stream with the lane's own table buffer (sh_rt: below).
struct GatherLane {
    hipStream_t st;
    void*       tab_dev;          // NULL: the library's scratch (sh::ensure_scratch)
    size_t      tab_cap;
};
int lane_table(const GatherLane& L, const void* host, size_t bytes, const void** dev) {
    if (!L.tab_dev) {
        int rc = sh::ensure_scratch(bytes);
        if (rc) return rc;
        *dev = sh::state().scratch;
    } else {
        if (bytes > L.tab_cap) return sh::set_error(SH_ERR_INVALID, "real-time lane: %zu sources, the lane was created for %zu", bytes / 16, L.tab_cap / 16);
        *dev = L.tab_dev;
    }
    // the table goes through the stream (pageable source: staged before the call returns, ordered after earlier kernels)
    SH_HIP(hipMemcpyAsync(const_cast<void*>(*dev), host, bytes, hipMemcpyHostToDevice, L.st));
    return SH_OK;
}

int gather_i16_on(const GatherLane& L, const sh_buf* const* srcs, const size_t* sample_offsets, const uint32_t* nsamples_each, uint32_t nsrc,
                  uint32_t nsamples, short* op) {
    std::vector<ChainSrc> tab;
    tab.reserve(nsrc);
    for (uint32_t v = 0; v < nsrc; ++v) {
        if (!nsamples_each[v]) continue;                  // silence: the fold's identity
        if (!srcs[v] || sample_offsets[v] > srcs[v]->bytes / 2 || nsamples_each[v] > srcs[v]->bytes / 2 - sample_offsets[v])
            return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather_i16: source %u range outside its buffer", v);
        ChainSrc c;
        c.p = (const short*)srcs[v]->ptr + sample_offsets[v];
        c.n = nsamples_each[v] < nsamples ? nsamples_each[v] : nsamples;
        c.pad = 0;
        tab.push_back(c);
    }
    hipStream_t st = L.st;
    if (tab.empty()) {
        SH_HIP(hipMemsetAsync(op, 0, (size_t)nsamples * 2, st));
        return SH_OK;
    }
    if (tab.size() <= 64 && sh::div_up(nsamples, 512) < 1536) {          // a mixer turn: the table travels in the kernel arguments
        ChainTab args;
        for (size_t k = 0; k < 64; ++k) args.e[k] = k < tab.size() ? tab[k] : ChainSrc{nullptr, 0, 0};
        const uint32_t n = (uint32_t)tab.size();
        dim3 grid(sh::div_up(nsamples, 512));
        if (n >= 64) hipLaunchKernelGGL(k_mix_chain_gather_args<8>, grid, dim3(8 * 64), 0, st, args, n, nsamples, op);
        else hipLaunchKernelGGL(k_mix_chain_gather_args<2>, grid, dim3(2 * 64), 0, st, args, n, nsamples, op);
        SH_CHECK_LAUNCH("k_mix_chain_gather_args");
        return SH_OK;
    }
    const void* dtab = nullptr;
    int rc = lane_table(L, tab.data(), tab.size() * sizeof(ChainSrc), &dtab);
    if (rc) return rc;
    const uint32_t n = (uint32_t)tab.size();
    dim3 grid(sh::div_up(nsamples, 512));
    size_t read_bytes = 0;
    for (const ChainSrc& c : tab) read_bytes += (size_t)c.n * 2;
    if (sh::div_up(nsamples, 512) >= 1536 && read_bytes > sh::STREAM_BYTES)
        hipLaunchKernelGGL((k_mix_chain_gather_direct<8, 4, true>), sh::grid1d(nsamples, 512 * 8), dim3(8 * 64), 0, st, (const ChainSrc*)dtab, n, nsamples, op);
    else if (sh::div_up(nsamples, 512) >= 1536)
        hipLaunchKernelGGL((k_mix_chain_gather_direct<8, 4, false>), sh::grid1d(nsamples, 512 * 8), dim3(8 * 64), 0, st, (const ChainSrc*)dtab, n, nsamples, op);
    else if (n >= 64) hipLaunchKernelGGL(k_mix_chain_gather<8>, grid, dim3(8 * 64), 0, st, (const ChainSrc*)dtab, n, nsamples, op);
    else hipLaunchKernelGGL(k_mix_chain_gather<2>, grid, dim3(2 * 64), 0, st, (const ChainSrc*)dtab, n, nsamples, op);
    SH_CHECK_LAUNCH("k_mix_chain_gather");
    return SH_OK;
}

// widths 1, 3, 4
int gather_w_on(const GatherLane& L, const sh_buf* const* srcs, const size_t* sample_offsets, const uint32_t* nsamples_each, uint32_t nsrc,
                uint32_t nsamples, int width, unsigned char* op) {
    const size_t w = (size_t)width;
    std::vector<ChainSrcB> tab;
    tab.reserve(nsrc);
    for (uint32_t v = 0; v < nsrc; ++v) {
        if (!nsamples_each[v]) continue;                  // silence: the fold's identity
        if (!srcs[v] || sample_offsets[v] > srcs[v]->bytes / w || nsamples_each[v] > srcs[v]->bytes / w - sample_offsets[v])
            return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather: source %u range outside its buffer", v);
        ChainSrcB c;
        c.p = (const unsigned char*)srcs[v]->ptr + sample_offsets[v] * w;
        c.n = nsamples_each[v] < nsamples ? nsamples_each[v] : nsamples;
        c.pad = 0;
        tab.push_back(c);
    }
    hipStream_t st = L.st;
    if (tab.empty()) {
        SH_HIP(hipMemsetAsync(op, 0, (size_t)nsamples * w, st));
        return SH_OK;
    }
    const void* dtab = nullptr;
    int rc = lane_table(L, tab.data(), tab.size() * sizeof(ChainSrcB), &dtab);
    if (rc) return rc;
    const uint32_t n = (uint32_t)tab.size();
    const dim3 grid = sh::grid1d(nsamples, 1024);
    const ChainSrcB* dt = (const ChainSrcB*)dtab;
    if (width == 1) hipLaunchKernelGGL(k_mix_chain_gather_w<1>, grid, dim3(256), 0, st, dt, n, nsamples, op);
    else if (width == 3) hipLaunchKernelGGL(k_mix_chain_gather_w<3>, grid, dim3(256), 0, st, dt, n, nsamples, op);
    else hipLaunchKernelGGL(k_mix_chain_gather_w<4>, grid, dim3(256), 0, st, dt, n, nsamples, op);
    SH_CHECK_LAUNCH("k_mix_chain_gather_w");
    return SH_OK;
}
}  // namespace

// ---- the real-time lane (include/synthhip.h: sh_rt_*) -----------------------------------------------------------------------------
// The reference drives its mixer from a thread of its own (playback.py: the output thread pulls chunks) while other threads make
// sound.  Through the library's one lock and one stream pair that thread's turn queues behind whatever the others have enqueued and
// its download holds the lock while the stream drains.  A lane is a stream (high priority), a lock, a table buffer and a chunk
// buffer of its own: a turn -- the gather fold + the chunk back to the host -- takes the LANE's lock only and waits for the lane's
// stream only.  Order against the library's streams is established once per source, by sh_rt_acquire.
struct sh_rt {
    hipStream_t stream = nullptr;
    hipEvent_t  ev1 = nullptr, ev2 = nullptr;
    std::mutex  mu;
    void*       tab_dev = nullptr;
    size_t      tab_cap = 0;
    void*       out_dev = nullptr;
    void*       out_pinned = nullptr;
    size_t      out_cap = 0;
};

extern "C" {

int sh_rt_create(size_t max_chunk_bytes, uint32_t max_sources, sh_rt** out) {
    SH_REQUIRE_INIT();
    if (!out || max_chunk_bytes == 0) return sh::set_error(SH_ERR_INVALID, "sh_rt_create: NULL / empty argument");
    sh_rt* r = new (std::nothrow) sh_rt;
    if (!r) return sh::set_error(SH_ERR_NOMEM, "host allocation failed");
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);                      // (hi: the numerically lowest = most urgent)
    hipError_t e;
    const int rt_cus = sh::knobs().rt_cus;
    if (rt_cus > 0) {                                                     // the compute units the library's streams leave out (SYNTHHIP_RT_CUS)
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, sh::state().device);
        const int ncu = e == hipSuccess ? prop.multiProcessorCount : 0;
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int c = ncu - rt_cus; c >= 0 && c < ncu; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
        if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&r->stream, (uint32_t)mask.size(), mask.data());
    } else {
        e = hipStreamCreateWithPriority(&r->stream, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev1, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev2, hipEventDisableTiming);
    r->tab_cap = (size_t)(max_sources < 64 ? 64 : max_sources) * 16;
    r->out_cap = (max_chunk_bytes + 255) & ~(size_t)255;
    if (e == hipSuccess) e = hipMalloc(&r->tab_dev, r->tab_cap);
    if (e == hipSuccess) e = hipMalloc(&r->out_dev, r->out_cap);
    if (e == hipSuccess) e = hipHostMalloc(&r->out_pinned, r->out_cap, hipHostMallocDefault);
    if (e != hipSuccess) {
        sh_rt_destroy(r);
        return sh::hip_error(e, "sh_rt_create");
    }
    *out = r;
    return SH_OK;
}

int sh_rt_destroy(sh_rt* r) {
    if (!r) return SH_OK;
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    if (r->out_pinned) (void)hipHostFree(r->out_pinned);
    if (r->out_dev) (void)hipFree(r->out_dev);
    if (r->tab_dev) (void)hipFree(r->tab_dev);
    if (r->ev1) (void)hipEventDestroy(r->ev1);
    if (r->ev2) (void)hipEventDestroy(r->ev2);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
    return SH_OK;
}

int sh_rt_acquire(sh_rt* r, const sh_buf* src) {
    SH_API_LOCK();                                            // (the library's lock: this call looks at the library's streams)
    if (!sh::state().initialized) return sh::set_error(SH_ERR_NOTINIT, "sh_init() has not been called");
    if (!r || !src) return sh::set_error(SH_ERR_INVALID, "sh_rt_acquire: NULL argument");
    int rc = sh::bind_thread_to_device();
    if (rc) return rc;
    if (sh::fold_owed_into(src->ptr, src->bytes)) {           // a render still owes a fold into this memory: done now
        rc = sh::flush_pending();
        if (rc) return rc;
    }
    // whatever either of the library's streams has been given so far precedes the lane's next turn (events: nobody waits on the host)
    sh::State& S = sh::state();
    SH_HIP(hipEventRecord(r->ev1, S.stream));
    SH_HIP(hipStreamWaitEvent(r->stream, r->ev1, 0));
    if (S.stream2) {
        SH_HIP(hipEventRecord(r->ev2, S.stream2));
        SH_HIP(hipStreamWaitEvent(r->stream, r->ev2, 0));
    }
    return SH_OK;
}

int sh_rt_mix_turn(sh_rt* r, const sh_buf* const* srcs, const size_t* sample_offsets, const uint32_t* nsamples_each, uint32_t nsrc,
                   uint32_t nsamples, int width, void* out_host) {
    if (!r || !out_host || (nsrc && (!srcs || !sample_offsets || !nsamples_each))) return sh::set_error(SH_ERR_INVALID, "sh_rt_mix_turn: NULL argument");
    if (width < 1 || width > 4) return sh::set_error(SH_ERR_INVALID, "sh_rt_mix_turn: width %d not in {1, 2, 3, 4}", width);
    if (nsrc > 32768) return sh::set_error(SH_ERR_INVALID, "sh_rt_mix_turn: at most 32768 sources");
    const size_t nbytes = (size_t)nsamples * (size_t)width;
    if (nbytes > r->out_cap) return sh::set_error(SH_ERR_INVALID, "sh_rt_mix_turn: %zu bytes, the lane was created for chunks of %zu", nbytes, r->out_cap);
    if (!nsamples) return SH_OK;
    std::lock_guard<std::mutex> lane_lock(r->mu);             // the LANE's lock: the library's is not taken
    if (!sh::state().initialized) return sh::set_error(SH_ERR_NOTINIT, "sh_init() has not been called");
    int rc = sh::bind_thread_to_device();
    if (rc) return rc;
    const GatherLane L{r->stream, r->tab_dev, r->tab_cap};
    rc = width == 2 ? gather_i16_on(L, srcs, sample_offsets, nsamples_each, nsrc, nsamples, (short*)r->out_dev)
                    : gather_w_on(L, srcs, sample_offsets, nsamples_each, nsrc, nsamples, width, (unsigned char*)r->out_dev);
    if (rc) return rc;
    SH_HIP(hipMemcpyAsync(r->out_pinned, r->out_dev, nbytes, hipMemcpyDeviceToHost, r->stream));
    SH_HIP(hipStreamSynchronize(r->stream));
    memcpy(out_host, r->out_pinned, nbytes);
    return SH_OK;
}

int sh_mix_chain_gather_i16(const sh_buf* const* srcs, const size_t* sample_offsets, const uint32_t* nsamples_each, uint32_t nsrc,
                            uint32_t nsamples, sh_buf* out, size_t out_sample_off) {
    SH_REQUIRE_INIT();
    if (!out || (nsrc && (!srcs || !sample_offsets || !nsamples_each))) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather_i16: NULL argument");
    if (nsrc > 32768) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather_i16: at most 32768 sources");
    if (nsamples > 0xFFFF0000u) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather_i16: at most 2^32 - 65536 samples per call");
    if (out_sample_off > out->bytes / 2 || nsamples > out->bytes / 2 - out_sample_off)
        return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather_i16: output range outside buffer");
    if (!nsamples) return SH_OK;
    return gather_i16_on(GatherLane{sh::state().stream, nullptr, 0}, srcs, sample_offsets, nsamples_each, nsrc, nsamples, (short*)out->ptr + out_sample_off);
}

int sh_mix_chain_gather(const sh_buf* const* srcs, const size_t* sample_offsets, const uint32_t* nsamples_each, uint32_t nsrc,
                        uint32_t nsamples, int width, sh_buf* out, size_t out_sample_off) {
    if (width == 2) return sh_mix_chain_gather_i16(srcs, sample_offsets, nsamples_each, nsrc, nsamples, out, out_sample_off);
    SH_REQUIRE_INIT();
    if (width != 1 && width != 3 && width != 4) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather: width %d not in {1, 2, 3, 4}", width);
    if (!out || (nsrc && (!srcs || !sample_offsets || !nsamples_each))) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather: NULL argument");
    if (nsrc > 32768) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather: at most 32768 sources");
    if (nsamples > 0xFFFF0000u) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather: at most 2^32 - 65536 samples per call");
    const size_t w = (size_t)width;
    if (out_sample_off > out->bytes / w || nsamples > out->bytes / w - out_sample_off)
        return sh::set_error(SH_ERR_INVALID, "sh_mix_chain_gather: output range outside buffer");
    if (!nsamples) return SH_OK;
    return gather_w_on(GatherLane{sh::state().stream, nullptr, 0}, srcs, sample_offsets, nsamples_each, nsrc, nsamples, width,
                       (unsigned char*)out->ptr + out_sample_off * w);
}

int sh_mix_chain(const sh_buf* chunks, uint32_t nvoices, size_t stride, uint32_t nsamples, int width, sh_buf* out) {
    if (width == 2) return sh_mix_chain_i16(chunks, nvoices, stride, nsamples, out);
    if (!chunks || !out || nvoices == 0) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain: NULL argument");
    if (nvoices > 32768) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain: at most 32768 voices");
    if (width != 1 && width != 3 && width != 4) return sh::set_error(SH_ERR_INVALID, "sh_mix_chain: width %d not in {1, 2, 3, 4}", width);
    if (stride < nsamples || chunks->bytes / (size_t)width < (size_t)(nvoices - 1) * stride + nsamples)
        return sh::set_error(SH_ERR_INVALID, "sh_mix_chain: chunk buffer too small");
    std::vector<const sh_buf*> srcs(nvoices, chunks);
    std::vector<size_t> offs(nvoices);
    std::vector<uint32_t> lens(nvoices, nsamples);
    for (uint32_t v = 0; v < nvoices; ++v) offs[v] = (size_t)v * stride;
    return sh_mix_chain_gather(srcs.data(), offs.data(), lens.data(), nvoices, nsamples, width, out, 0);
}

size_t sh_resample_out_frames(size_t in_frames, int inrate, int outrate) {
    if (in_frames == 0 || inrate <= 0 || outrate <= 0) return 0;
    uint64_t g = gcd_u64((uint64_t)inrate, (uint64_t)outrate);
    uint64_t inr = (uint64_t)inrate / g, outr = (uint64_t)outrate / g;
    unsigned __int128 t = (unsigned __int128)(in_frames - 1) * outr;
    return (size_t)(t / inr) + 1;
}

// in / out are the addresses input frame 0 / output frame 0 would have (range launches pass pointers shifted back by
// the frames they do not hold: never dereferenced outside [held input), [m_base, m_end)).  in_frames = end of the
// held input, m_base / m_end = output frame range.
static int resample_launch(const void* in, size_t in_frames, int nch, int width, int is_float, int inrate, int outrate,
                           void* out, size_t m_base, size_t m_end, size_t in_lo, bool allow_period = true);

// Output ranges of any length: one launch per 2^30 output samples at most (a dispatch holds fewer than 2^32 work-items per grid
// dimension; the kernels work from absolute output positions, so a range cut at multiples of 4096 frames is the same range)
// (in_lo: the first input frame that is held -- range launches; the kernels that stage whole chunks must not read below it)
static int resample_dev(const void* in, size_t in_frames, int nch, int width, int is_float, int inrate, int outrate,
                        void* out, size_t m_base, size_t m_end, size_t in_lo = 0) {
    size_t chunk = (((size_t)1 << 30) / (size_t)nch) & ~(size_t)4095;
    if (chunk < 4096) chunk = 4096;
    for (size_t m = m_base; m < m_end; m += chunk) {
        const int rc = resample_launch(in, in_frames, nch, width, is_float, inrate, outrate, out, m, m_end - m < chunk ? m_end : m + chunk, in_lo);
        if (rc) return rc;
    }
    return SH_OK;
}

static int resample_launch(const void* in, size_t in_frames, int nch, int width, int is_float, int inrate, int outrate,
                           void* out, size_t m_base, size_t m_end, size_t in_lo, bool allow_period) {
    const size_t out_frames = m_end - m_base;        // frames this launch writes
    uint64_t g = gcd_u64((uint64_t)inrate, (uint64_t)outrate);
    RatecvArgs A;
    A.nch = (uint32_t)nch;
    A.inr = (uint32_t)((uint64_t)inrate / g);
    A.outr = (uint32_t)((uint64_t)outrate / g);
    A.inv_outr = 1.0 / (double)A.outr;
    A.step_q = A.inr / A.outr;
    A.step_r = A.inr % A.outr;
    A.shift = 32 - 8 * width;
    A.m_base = (uint64_t)m_base;
    const bool small = !is_float && width <= 2 && A.outr < 65536u;
#define SH_I(M, ...) do { if (small) M(__VA_ARGS__, RS_INT_SMALL); else M(__VA_ARGS__, RS_INT_F64); } while (0)
    if (!out_frames) return SH_OK;
    hipStream_t st = sh::state().stream;
    // widest channel vector that divides nch, stays <= 16 bytes and keeps every access aligned
    const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    int vec = 1;
    if (aligned) {
        const int maxvec = 16 / width;
        for (int v = maxvec; v > 1; v >>= 1)
            if (nch % v == 0) { vec = v; break; }
    }
    // mono / stereo (and other narrow layouts that fit one vector): several frames per thread
    if (aligned && nch * width <= 8 && (nch == 1 || nch == 2 || nch == 4)) {
        int fr = 16 / (nch * width) > 8 ? 8 : 16 / (nch * width);        // 16-byte stores, at most 8 frames per thread
        // 16-bit mono through the LDS kernel: 16 frames (two 16-byte stores) per thread -- the per-thread set-up (position of the
        // first frame, staging loop) is a fifth of the instructions at 8 frames; +4 % (stereo, already at 16 bytes per 4 frames: -5 %)
        // (only when the doubled input span still fits the LDS budget of that kernel: the other kernels keep 8 frames)
        // 16-bit mono / stereo between rates with a short period: chunks of whole periods, the weights loop-invariant per thread (k_resample_period_i16)
        if (allow_period && small && width == 2 && (nch == 1 || nch == 2) && A.inr < 65536u && A.outr <= 2048u && !sh::knobs().no_period) {
            const uint32_t V = (uint32_t)nch, fpr = 8u / V, lmax = 512u * fpr;         // a run = 16 bytes; two runs per thread
            const uint32_t s_out = fpr / (uint32_t)gcd_u64(A.outr, fpr);                // L = K outr must be a multiple of a run (16-byte stores)
            uint32_t K = lmax / A.outr;
            const uint32_t k_span = (14336u / V) / A.inr;                               // ... and the chunk's input span at most 28 KB
            if (K > k_span) K = k_span;
            K -= K % s_out;
            const uint32_t L = K * A.outr;
            if (L >= lmax / 4u * 3u) {                                                  // (three quarters of the threads' frames in use, at least)
                PeriodArgs P;
                P.L = L; P.kinr = K * A.inr; P.inr = A.inr; P.outr = A.outr; P.inv_outr = A.inv_outr;
                const uint32_t off_max = (uint32_t)(((uint64_t)(L - 1) * A.inr) / A.outr);
                P.span_vecs = (7u + (off_max + 2u) * V + 7u) / 8u;
                const uint32_t lds_bytes = P.span_vecs * 16u;
                // the interior chunks: frames wholly inside [m_base, m_end), span wholly inside the held input [in_lo, in_frames)
                uint64_t cA = ((uint64_t)m_base + L - 1) / L, cB = (uint64_t)m_end / L;
                while (cA < cB && ((cA * P.kinr * V) & ~(uint64_t)7) < (uint64_t)in_lo * V) ++cA;
                while (cB > cA && (((cB - 1) * P.kinr * V) & ~(uint64_t)7) + 8ull * P.span_vecs > (uint64_t)in_frames * V) --cB;
                if (cB > cA + 1) {
                    P.c0 = cA; P.c1 = cB;
                    // consecutive chunks per workgroup (profiles/r06_resample_period.txt; SYNTHHIP_PERIOD_CHUNKS overrides).  Mono: two -- one where
                    // the input is the larger side (nothing to amortise the set-up against but reads), four where the output is (upsampling by
                    // two or more).  Stereo: one (a thread's set-up is eight entries, and the longer a workgroup stays the more of them march in step).
                    P.per_wg = sh::knobs().period_chunks > 0 ? (uint32_t)sh::knobs().period_chunks
                             : nch == 2 || A.inr >= 2 * A.outr ? 1u : A.outr >= 2 * A.inr ? 4u : 2u;
                    const int nvk = P.span_vecs <= 512u ? 2 : P.span_vecs <= 1024u ? 4 : 8;
                    const dim3 gp((uint32_t)((cB - cA + P.per_wg - 1) / P.per_wg));
#define SH_RP(V_, N_) hipLaunchKernelGGL((k_resample_period_i16<V_, N_>), gp, dim3(256), lds_bytes, st, (const short*)in, (short*)out, P)
                    if (nch == 1) { if (nvk == 2) SH_RP(1, 2); else if (nvk == 4) SH_RP(1, 4); else SH_RP(1, 8); }
                    else { if (nvk == 2) SH_RP(2, 2); else if (nvk == 4) SH_RP(2, 4); else SH_RP(2, 8); }
#undef SH_RP
                    SH_CHECK_LAUNCH("k_resample_period_i16");
                    // ... and what lies in front of the first and behind the last interior chunk: k_resample_small (allow_period = false)
                    int rc = SH_OK;
                    if ((uint64_t)m_base < cA * L) rc = resample_launch(in, in_frames, nch, width, is_float, inrate, outrate, out, m_base, (size_t)(cA * L), in_lo, false);
                    if (!rc && cB * L < (uint64_t)m_end) rc = resample_launch(in, in_frames, nch, width, is_float, inrate, outrate, out, (size_t)(cB * L), m_end, in_lo, false);
                    return rc;
                }
            }
        }
        bool wide = false;
        if (small && width == 2 && nch == 1 && A.inr < 65536u) {
            const uint64_t sf2 = ((uint64_t)256 * 2 * fr * A.inr + A.outr - 1) / A.outr + 3;
            wide = ((sf2 * nch + 8 + 7) / 8 + 1) * 16 <= RS_LDS_BYTES;
        }
        if (wide) fr *= 2;
        A.n_out_samples = (uint64_t)out_frames;
        dim3 g2(sh::div_up(sh::div_up(out_frames, fr), 256));
        // input bytes one workgroup (256*fr output frames) touches; stage them in LDS when they fit
        const uint64_t span_frames = ((uint64_t)256 * fr * A.inr + A.outr - 1) / A.outr + 3;
        const uint64_t span_bytes = span_frames * nch * width + 32;
        if (small && A.inr < 65536u) {
            // span: frames q0 .. q0 + floor((r0 + (256*fr-1)*inr)/outr) + 1, the alignment slack of the first vector,
            // and one more vector for the dword-pair reads
            const uint32_t epv = 16 / width;
            const uint64_t svecs = (span_frames * nch + epv + epv - 1) / epv + 1;
            if (svecs * 16 <= RS_LDS_BYTES) {
                const uint32_t span_vecs = (uint32_t)svecs, lds_bytes = span_vecs * 16;
#define SH_RM(T, V, F) hipLaunchKernelGGL((k_resample_small<T, V, F>), g2, dim3(256), lds_bytes, st, (const T*)in, (T*)out, A, (uint64_t)in_frames, (uint64_t)m_end, span_vecs)
                // 16-bit mono: 16 frames per thread as two runs of 8 (GROUPS = 2) -- a wave's store instruction then writes 1 KB of
                // consecutive bytes: +0.4 / +2.2 / +4 % on 44.1 -> 48, 96 -> 44.1, 48 -> 44.1 kHz against one run of 16
                if (wide) hipLaunchKernelGGL((k_resample_small<short, 1, 16, 2>), g2, dim3(256), lds_bytes, st, (const short*)in, (short*)out, A,
                                             (uint64_t)in_frames, (uint64_t)m_end, span_vecs);
                else if (width == 2) { if (nch == 1) SH_RM(short, 1, 8); else if (nch == 2) SH_RM(short, 2, 4); else SH_RM(short, 4, 2); }
                else { if (nch == 1) SH_RM(signed char, 1, 8); else if (nch == 2) SH_RM(signed char, 2, 8); else SH_RM(signed char, 4, 4); }
#undef SH_RM
                SH_CHECK_LAUNCH("k_resample_small");
                return SH_OK;
            }
        }
        if (nch <= 2 && span_bytes <= RS_LDS_BYTES) {
            const uint32_t lds_bytes = (uint32_t)((span_bytes + 15) & ~15ull);
#define SH_RL(T, V, F, FL) hipLaunchKernelGGL((k_resample_lds<T, V, F, FL>), g2, dim3(256), lds_bytes, st, (const T*)in, (T*)out, A, (uint64_t)in_frames, (uint64_t)m_end)
            if (is_float) { if (nch == 1) SH_RL(float, 1, 4, RS_FLOAT); else SH_RL(float, 2, 2, RS_FLOAT); }
            else if (width == 2) { if (nch == 1) SH_I(SH_RL, short, 1, 8); else SH_I(SH_RL, short, 2, 4); }
            else if (width == 4) { if (nch == 1) SH_RL(int, 1, 4, RS_INT_F64); else SH_RL(int, 2, 2, RS_INT_F64); }
            else { if (nch == 1) SH_I(SH_RL, signed char, 1, 8); else SH_I(SH_RL, signed char, 2, 8); }
#undef SH_RL
            SH_CHECK_LAUNCH("k_resample_lds");
            return SH_OK;
        }
#define SH_RF(T, V, F, FL) hipLaunchKernelGGL((k_resample_frames<T, V, F, FL>), g2, dim3(256), 0, st, (const T*)in, (T*)out, A, (uint64_t)m_end)
        bool launched = true;
        if (is_float) {
            if (nch == 1) SH_RF(float, 1, 4, RS_FLOAT); else if (nch == 2) SH_RF(float, 2, 2, RS_FLOAT); else launched = false;
        } else if (width == 2) {
            if (nch == 1) SH_I(SH_RF, short, 1, 8); else if (nch == 2) SH_I(SH_RF, short, 2, 4); else SH_I(SH_RF, short, 4, 2);
        } else if (width == 4) {
            if (nch == 1) SH_RF(int, 1, 4, RS_INT_F64); else if (nch == 2) SH_RF(int, 2, 2, RS_INT_F64); else launched = false;
        } else {
            if (nch == 1) SH_I(SH_RF, signed char, 1, 8); else if (nch == 2) SH_I(SH_RF, signed char, 2, 8); else SH_I(SH_RF, signed char, 4, 4);
        }
#undef SH_RF
        if (launched) {
            SH_CHECK_LAUNCH("k_resample_frames");
            return SH_OK;
        }
    }
    A.n_out_samples = (uint64_t)out_frames * (nch / vec);
    dim3 grid(sh::div_up(A.n_out_samples, 256));
#define SH_RS(T, V, F) hipLaunchKernelGGL((k_resample<T, V, F>), grid, dim3(256), 0, st, (const T*)in, (T*)out, A)
    if (is_float) {
        if (vec == 4) SH_RS(float, 4, RS_FLOAT); else if (vec == 2) SH_RS(float, 2, RS_FLOAT); else SH_RS(float, 1, RS_FLOAT);
    } else if (width == 2) {
        if (vec == 8) SH_I(SH_RS, short, 8); else if (vec == 4) SH_I(SH_RS, short, 4);
        else if (vec == 2) SH_I(SH_RS, short, 2); else SH_I(SH_RS, short, 1);
    } else if (width == 4) {
        if (vec == 4) SH_RS(int, 4, RS_INT_F64); else if (vec == 2) SH_RS(int, 2, RS_INT_F64); else SH_RS(int, 1, RS_INT_F64);
    } else {
        if (vec == 16) SH_I(SH_RS, signed char, 16); else if (vec == 8) SH_I(SH_RS, signed char, 8);
        else if (vec == 4) SH_I(SH_RS, signed char, 4); else if (vec == 2) SH_I(SH_RS, signed char, 2);
        else SH_I(SH_RS, signed char, 1);
    }
#undef SH_RS
#undef SH_I
    SH_CHECK_LAUNCH("k_resample");
    return SH_OK;
}

// audioop.ratecv at width 3 works on GETSAMPLE32 = value << 8 and stores SETSAMPLE32 = result >> 8: the 32-bit path on
// unpacked samples, packed again.
static int resample24(const void* in, size_t in_frames_held, size_t in_first, int nch, int inrate, int outrate,
                      void* out, size_t out_first, size_t out_n) {
    sh::Temp tin, tout;
    int rc = tin.alloc(in_frames_held * nch * 4);
    if (!rc) rc = tout.alloc(out_n * nch * 4);
    if (!rc) rc = sh::unpack24(in, in_frames_held * nch, 8, (int32_t*)tin.buf.ptr);
    if (rc) return rc;
    const size_t fb = (size_t)4 * nch;
    const char* in0 = (const char*)tin.buf.ptr - in_first * fb;
    char* out0 = (char*)tout.buf.ptr - out_first * fb;
    rc = resample_dev(in0, in_first + in_frames_held, nch, 4, 0, inrate, outrate, out0, out_first, out_first + out_n);
    if (!rc) rc = sh::pack24((const int32_t*)tout.buf.ptr, out_n * nch, 8, out);
    return rc;
}

static int resample_check(int nch, int width, int is_float, int inrate, int outrate) {
    if (nch < 1) return sh::set_error(SH_ERR_INVALID, "resample: # of channels should be >= 1");
    if (width != 1 && width != 2 && width != 3 && width != 4) return sh::set_error(SH_ERR_INVALID, "resample: width %d not in {1,2,3,4}", width);
    if (is_float && width != 4) return sh::set_error(SH_ERR_INVALID, "resample: float PCM must have width 4");
    if (inrate <= 0 || outrate <= 0) return sh::set_error(SH_ERR_INVALID, "resample: sampling rate not > 0");
    return SH_OK;
}

int sh_resample(const sh_buf* in, size_t in_frames, int nchannels, int width, int is_float,
                int inrate, int outrate, sh_buf* out, size_t* out_frames) {
    SH_REQUIRE_INIT();
    if (!in || !out) return sh::set_error(SH_ERR_INVALID, "sh_resample: NULL argument");
    int rc = resample_check(nchannels, width, is_float, inrate, outrate);
    if (rc) return rc;
    size_t nout = sh_resample_out_frames(in_frames, inrate, outrate);
    if (in->bytes / ((size_t)width * nchannels) < in_frames) return sh::set_error(SH_ERR_INVALID, "sh_resample: input buffer smaller than in_frames");
    if (out->bytes / ((size_t)width * nchannels) < nout) return sh::set_error(SH_ERR_INVALID, "sh_resample: output buffer too small (%zu frames needed)", nout);
    if (out_frames) *out_frames = nout;
    if (width == 3) return nout ? resample24(in->ptr, in_frames, 0, nchannels, inrate, outrate, out->ptr, 0, nout) : (int)SH_OK;
    return resample_dev(in->ptr, in_frames, nchannels, width, is_float, inrate, outrate, out->ptr, 0, nout);
}

int sh_resample_span(size_t in_total_frames, int inrate, int outrate, size_t out_first, size_t out_n,
                     size_t* in_first, size_t* in_count) {
    if (!in_first || !in_count) return sh::set_error(SH_ERR_INVALID, "sh_resample_span: NULL argument");
    if (inrate <= 0 || outrate <= 0) return sh::set_error(SH_ERR_INVALID, "resample: sampling rate not > 0");
    const size_t nout = sh_resample_out_frames(in_total_frames, inrate, outrate);
    if (out_first > nout || out_n > nout - out_first) return sh::set_error(SH_ERR_INVALID, "sh_resample_span: output range outside the %zu output frames", nout);
    *in_first = 0;
    *in_count = 0;
    if (!out_n) return SH_OK;
    const uint64_t g = gcd_u64((uint64_t)inrate, (uint64_t)outrate);
    const unsigned __int128 inr = (uint64_t)inrate / g, outr = (uint64_t)outrate / g;
    // output frame m interpolates input frames j-1 and j, j = ceil(m*inr/outr)
    const uint64_t j_lo = (uint64_t)(((unsigned __int128)out_first * inr + outr - 1) / outr);
    const uint64_t j_hi = (uint64_t)(((unsigned __int128)(out_first + out_n - 1) * inr + outr - 1) / outr);
    uint64_t first = j_lo ? j_lo - 1 : 0;
    first &= ~(uint64_t)15;                          // 16 frames of any layout are a multiple of 16 bytes: vector loads stay aligned
    *in_first = (size_t)first;
    *in_count = (size_t)(j_hi - first + 1);
    return SH_OK;
}

int sh_resample_range(const sh_buf* in, size_t in_first, size_t in_held, int nchannels, int width, int is_float,
                      int inrate, int outrate, size_t out_first, size_t out_n, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (!in || !out) return sh::set_error(SH_ERR_INVALID, "sh_resample_range: NULL argument");
    int rc = resample_check(nchannels, width, is_float, inrate, outrate);
    if (rc) return rc;
    if ((out_first | in_first) & 15) return sh::set_error(SH_ERR_INVALID, "sh_resample_range: ranges must start at a multiple of 16 frames");
    const size_t fb = (size_t)width * nchannels;
    if (in->bytes / fb < in_held) return sh::set_error(SH_ERR_INVALID, "sh_resample_range: input buffer smaller than in_held frames");
    if (out->bytes / fb < out_n) return sh::set_error(SH_ERR_INVALID, "sh_resample_range: output buffer too small (%zu frames needed)", out_n);
    if (!out_n) return SH_OK;
    // the input frames the range reads (exact rational index arithmetic, as in sh_resample_span)
    const uint64_t g = gcd_u64((uint64_t)inrate, (uint64_t)outrate);
    const unsigned __int128 inr = (uint64_t)inrate / g, outr = (uint64_t)outrate / g;
    const uint64_t j_lo = (uint64_t)(((unsigned __int128)out_first * inr + outr - 1) / outr);
    const uint64_t j_hi = (uint64_t)(((unsigned __int128)(out_first + out_n - 1) * inr + outr - 1) / outr);
    const uint64_t need_lo = j_lo ? j_lo - 1 : 0;
    if (in_first > need_lo || j_hi >= (uint64_t)in_first + in_held)
        return sh::set_error(SH_ERR_INVALID, "sh_resample_range: output frames [%zu,+%zu) read input frames [%llu,%llu], buffer holds [%zu,+%zu)",
                             out_first, out_n, (unsigned long long)need_lo, (unsigned long long)j_hi, in_first, in_held);
    if (width == 3) return resample24(in->ptr, in_held, in_first, nchannels, inrate, outrate, out->ptr, out_first, out_n);
    const char* in0 = (const char*)in->ptr - in_first * fb;        // where input frame 0 would be
    char* out0 = (char*)out->ptr - out_first * fb;                  // where output frame 0 would be
    return resample_dev(in0, in_first + in_held, nchannels, width, is_float, inrate, outrate, out0, out_first, out_first + out_n, in_first);
}

int sh_resample_host(const void* in, size_t in_frames, int nchannels, int width, int is_float,
                     int inrate, int outrate, void* out, size_t* out_frames) {
    SH_REQUIRE_INIT();
    int rc = resample_check(nchannels, width, is_float, inrate, outrate);
    if (rc) return rc;
    size_t nout = sh_resample_out_frames(in_frames, inrate, outrate);
    if (out_frames) *out_frames = nout;
    if (!nout) return SH_OK;
    if (!in || !out) return sh::set_error(SH_ERR_INVALID, "sh_resample_host: NULL argument");
    size_t fb = (size_t)width * nchannels;
    size_t in_bytes = in_frames * fb, out_bytes = nout * fb;
    size_t in_pad = (in_bytes + 255) & ~size_t(255);
    rc = sh::ensure_scratch(in_pad + out_bytes);
    if (rc) return rc;
    char* s = (char*)sh::state().scratch;
    hipStream_t st = sh::state().stream;
    SH_HIP(hipMemcpyAsync(s, in, in_bytes, hipMemcpyHostToDevice, st));
    rc = width == 3 ? resample24(s, in_frames, 0, nchannels, inrate, outrate, s + in_pad, 0, nout)
                    : resample_dev(s, in_frames, nchannels, width, is_float, inrate, outrate, s + in_pad, 0, nout);
    if (rc) return rc;
    SH_HIP(hipMemcpyAsync(out, s + in_pad, out_bytes, hipMemcpyDeviceToHost, st));
    SH_HIP(hipStreamSynchronize(st));
    return SH_OK;
}

}  // extern "C"
