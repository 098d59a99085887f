// pcm_ops.hip -- the remaining elementwise / reduction operations that synthplayer's Sample delegates to
// CPython's audioop (Modules/audioop.c, 3.10): mul (amplify, invert), bias, reverse, tomono, tostereo,
// lin2lin, max, rms -- plus the per-sample fade ramps of Sample.fadein / fadeout.  SURVEY.md section 8(f)
// item 2.  HBM-bound byte/integer work: 16-byte vectors per thread where alignment allows, wavefront
// shuffle (DPP) reductions + one integer atomic per workgroup for max / sum of squares.
// Built with -ffp-contract=off: audioop forms val1*lfactor + val2*rfactor with separate roundings.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace {

// audioop's fbound(): clamp, then round toward minus infinity
__device__ __forceinline__ int fbound(double val, double minval, double maxval) {
    if (val > maxval) val = maxval;
    else if (val < minval + 1.0) val = minval;
    return (int)floor(val);
}

template <typename T> struct Lim;
template <> struct Lim<signed char> { static constexpr double lo = -128.0, hi = 127.0; };
template <> struct Lim<short> { static constexpr double lo = -32768.0, hi = 32767.0; };
template <> struct Lim<int> { static constexpr double lo = -2147483648.0, hi = 2147483647.0; };

// out[i] = fbound(in[i] * factor)            (audioop.mul)
template <typename T, int VEC, bool NT = false>
__global__ __launch_bounds__(256) void k_mul(const T* in, T* out, size_t nvec, double factor) {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= nvec) return;
    vec_t v = sh::load_vec<NT, vec_t>(reinterpret_cast<const vec_t*>(in) + i), r;       // NT (streaming sizes): +1 % (tomono: -1 %, left as it was)
#pragma unroll
    for (int c = 0; c < VEC; ++c) r[c] = (T)fbound((double)v[c] * factor, Lim<T>::lo, Lim<T>::hi);
    reinterpret_cast<vec_t*>(out)[i] = r;
}

// per-sample linear ramp: out[i] = int(in[i] * (i*slope/numsamples + offset)), truncation toward zero
// (Sample.fadeout: offset 1, slope -decrease; Sample.fadein: offset start_volume, slope increase)
template <typename T>
__global__ __launch_bounds__(256) void k_fade(const T* in, T* out, size_t n, double slope, double numsamples, double offset, int fadeout) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    const double ramp = (double)i * slope / numsamples;
    const double f = fadeout ? (1.0 - ramp) : (ramp + offset);
    out[i] = (T)(long long)trunc((double)in[i] * f);
}

// out[i] = int(in[i] * mod[i mod nmod])          (Sample.modulate_amp: Python float product, int() truncation;
// a product outside the sample range raises upstream -> flag)
template <typename T>
__global__ __launch_bounds__(256) void k_modulate(const T* __restrict__ in, T* __restrict__ out, size_t n,
                                                  const double* __restrict__ mod, size_t nmod, int* flag) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t k = i < nmod ? i : (i < 0xFFFFFFFFull && nmod < 0xFFFFFFFFull ? (size_t)((uint32_t)i % (uint32_t)nmod) : i % nmod);
    constexpr double HI = (double)((1ll << (8 * sizeof(T) - 1)) - 1), LO = -(double)(1ll << (8 * sizeof(T) - 1));
    double t = trunc((double)in[i] * mod[k]);
    if (!(t >= LO && t <= HI)) {
        *flag = 1;
        t = t > HI ? HI : LO;
    }
    out[i] = (T)(long long)t;
}

// Sample.pan(lfo=...): frame i becomes (int(l * (1 - p) / 2), int(r * (1 + p) / 2)) with p = pan[i]; a mono source
// feeds both sides.  float64 like the Python expression (the halving is exact); a value outside the sample range
// raises upstream (array assignment) -> flag.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void k_pan_lfo(const T* __restrict__ in, T* __restrict__ out, size_t nframes,
                                                 const double* __restrict__ pan, int* flag) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= nframes) return;
    constexpr double HI = (double)((1ll << (8 * sizeof(T) - 1)) - 1), LO = -(double)(1ll << (8 * sizeof(T) - 1));
    const double p = pan[i];
    const double l = (double)in[i * NCH], r = (double)in[i * NCH + NCH - 1];
    double tl = trunc(l * (1.0 - p) / 2.0), tr = trunc(r * (1.0 + p) / 2.0);
    if (!(tl >= LO && tl <= HI)) { *flag = 1; tl = tl > HI ? HI : LO; }
    if (!(tr >= LO && tr <= HI)) { *flag = 1; tr = tr > HI ? HI : LO; }
    typedef T pair_t __attribute__((ext_vector_type(2)));
    pair_t o = {(T)(long long)tl, (T)(long long)tr};
    reinterpret_cast<pair_t*>(out)[i] = o;
}

// out[i] = in[i] / divisor in float64              (Sample.get_frames_as_floats; waveform modulators)
template <typename T>
__global__ __launch_bounds__(256) void k_to_f64(const T* __restrict__ in, double* __restrict__ out, size_t n, double divisor) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = (double)in[i] / divisor;
}

// out[i] = in[i] + bias, wrapping                (audioop.bias)
template <typename T>
__global__ __launch_bounds__(256) void k_bias(const T* in, T* out, size_t n, int bias) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = (T)((unsigned)(int)in[i] + (unsigned)bias);
}

// out[i] = in[n-1-i]                             (audioop.reverse: samples, not frames)
template <typename T>
__global__ __launch_bounds__(256) void k_reverse(const T* __restrict__ in, T* __restrict__ out, size_t n) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = in[n - 1 - i];
}

// stereo -> mono: fbound(l*lfactor + r*rfactor)  (audioop.tomono); F frames per thread (16-byte loads)
template <typename T, int F>
__global__ __launch_bounds__(256) void k_tomono(const T* __restrict__ in, T* __restrict__ out, size_t nunits, double lf, double rf) {
    typedef T vin __attribute__((ext_vector_type(2 * F)));
    typedef T vout __attribute__((ext_vector_type(F)));
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= nunits) return;
    const vin v = reinterpret_cast<const vin*>(in)[i];
    vout r;
#pragma unroll
    for (int f = 0; f < F; ++f) r[f] = (T)fbound((double)v[2 * f] * lf + (double)v[2 * f + 1] * rf, Lim<T>::lo, Lim<T>::hi);
    if (F == 1) out[i] = r[0]; else reinterpret_cast<vout*>(out)[i] = r;
}

// mono -> stereo: (fbound(v*lfactor), fbound(v*rfactor))   (audioop.tostereo)
template <typename T, int F>
__global__ __launch_bounds__(256) void k_tostereo(const T* __restrict__ in, T* __restrict__ out, size_t nunits, double lf, double rf) {
    typedef T vin __attribute__((ext_vector_type(F)));
    typedef T vout __attribute__((ext_vector_type(2 * F)));
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= nunits) return;
    vin v;
    if (F == 1) v[0] = in[i]; else v = reinterpret_cast<const vin*>(in)[i];
    vout r;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const double x = (double)v[f];
        r[2 * f] = (T)fbound(x * lf, Lim<T>::lo, Lim<T>::hi);
        r[2 * f + 1] = (T)fbound(x * rf, Lim<T>::lo, Lim<T>::hi);
    }
    reinterpret_cast<vout*>(out)[i] = r;
}

// width conversion through the 32-bit form (GETSAMPLE32 / SETSAMPLE32)   (audioop.lin2lin)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_lin2lin(const TI* __restrict__ in, TO* __restrict__ out, size_t n) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    const int v32 = (int)((unsigned)(int)in[i] << (32 - 8 * (int)sizeof(TI)));
    out[i] = (TO)(v32 >> (32 - 8 * (int)sizeof(TO)));
}

// reductions: acc[0] = max |v| (unsigned), acc[1] = sum v*v (u64, exact for widths 1 and 2)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned w = __shfl_down(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

template <typename T, int INFLIGHT>
__global__ __launch_bounds__(256) void k_absmax_sumsq(const T* __restrict__ in, size_t n, unsigned long long* __restrict__ acc) {
    __shared__ unsigned long long s_sum[4];
    __shared__ unsigned s_max[4];
    unsigned mx = 0;
    unsigned long long sq = 0;
    constexpr int V = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const size_t nvec = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) ? n / V : 0;
    if constexpr (sizeof(T) == 2) {
        // 16-bit: packed arithmetic, two samples per instruction.  |x| = max(x, 0 - x) as int16 pairs, read as
        // uint16 (so |-32768| = 32768 comes out right); running maximum as uint16 pairs; a pair's squares summed
        // by the dot-product instruction (<= 2^31, fits uint32) and added to the 64-bit total.
        typedef short s2 __attribute__((ext_vector_type(2)));
        typedef unsigned short u2 __attribute__((ext_vector_type(2)));
        u2 mx2 = {0, 0};
#define SH_PAIR(X_, A_, B_)                                                            \
            {                                                                            \
                const s2 v = __builtin_shufflevector(X_, X_, A_, B_);                    \
                const s2 neg = (s2){0, 0} - v;                                           \
                const u2 au = __builtin_bit_cast(u2, __builtin_elementwise_max(v, neg)); \
                mx2 = __builtin_elementwise_max(mx2, au);                                \
                sq += (unsigned long long)__builtin_amdgcn_udot2(au, au, 0u, false);     \
            }
        // a workgroup reads INFLIGHT * 4 KB contiguous per turn (the loads in flight are neighbours, not a grid apart:
        // the whole chip then sweeps one window of memory at a time, which the DRAM pages like)
        const size_t step = (size_t)gridDim.x * 256 * INFLIGHT;
        size_t i = (size_t)blockIdx.x * 256 * INFLIGHT + threadIdx.x;
        for (; i + (INFLIGHT - 1) * 256 < nvec; i += step) {
            vec_t x[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) x[k] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(in) + i + k * 256);
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) { SH_PAIR(x[k], 0, 1) SH_PAIR(x[k], 2, 3) SH_PAIR(x[k], 4, 5) SH_PAIR(x[k], 6, 7) }
        }
        if (i < nvec) {                                            // the last, partial turn of this workgroup
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) {
                if (i + k * 256 < nvec) {
                    const vec_t x = reinterpret_cast<const vec_t*>(in)[i + k * 256];
                    SH_PAIR(x, 0, 1) SH_PAIR(x, 2, 3) SH_PAIR(x, 4, 5) SH_PAIR(x, 6, 7)
                }
            }
        }
#undef SH_PAIR
        mx = mx2[0] > mx2[1] ? mx2[0] : mx2[1];
    } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const vec_t x = reinterpret_cast<const vec_t*>(in)[i];
#pragma unroll
        for (int c = 0; c < V; ++c) {
            const long long v = (long long)x[c];
            const unsigned a = (unsigned)(v < 0 ? -v : v);
            mx = a > mx ? a : mx;
            sq += (unsigned long long)(v * v);
        }
    }
    }
    for (size_t i = nvec * V + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const long long v = (long long)in[i];
        const unsigned a = (unsigned)(v < 0 ? -v : v);
        mx = a > mx ? a : mx;
        sq += (unsigned long long)(v * v);
    }
    mx = wave_max_u32(mx);
    sq = wave_sum_u64(sq);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_sum[wave] = sq; s_max[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        unsigned m = s_max[0];
        for (int w = 1; w < 4; ++w) m = s_max[w] > m ? s_max[w] : m;
        // per-workgroup results, folded by k_stats_fold: thousands of atomics on two addresses serialise (10 ns each,
        // 40 % of this kernel's time at 4096 workgroups)
        acc[2 * blockIdx.x] = (unsigned long long)m;
        acc[2 * blockIdx.x + 1] = t;
    }
}

// one workgroup: out[0] = max, out[1] = sum over the per-workgroup pairs (integer: exact, order-independent)
__global__ __launch_bounds__(256) void k_stats_fold(const unsigned long long* __restrict__ part, unsigned nblocks,
                                                    unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s_sum[4];
    __shared__ unsigned s_max[4];
    unsigned mx = 0;
    unsigned long long sq = 0;
    for (unsigned b = threadIdx.x; b < nblocks; b += 256) {
        const unsigned m = (unsigned)part[2 * b];
        mx = m > mx ? m : mx;
        sq += part[2 * b + 1];
    }
    mx = wave_max_u32(mx);
    sq = wave_sum_u64(sq);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_sum[wave] = sq; s_max[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = s_max[0];
        for (int w = 1; w < 4; ++w) m = s_max[w] > m ? s_max[w] : m;
        out[0] = (unsigned long long)m;
        out[1] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    }
}

// Interleaved stereo, both channels in one pass (Sample.level_db_peak / level_db_rms: upstream takes
// audioop.tomono(frames, w, 1, 0) and (.., 0, 1), two copies, and runs audioop.max / rms over each -- four passes and
// two temporaries; here: one read).  A 16-byte vector holds whole frames, so even elements are left, odd right.
// part[4b..4b+3] = max|L|, max|R|, sum L^2, sum R^2; for width 4 the sums are float64 bit patterns.
template <typename T>
__global__ __launch_bounds__(256) void k_stats_stereo(const T* __restrict__ in, size_t nframes, unsigned long long* __restrict__ part) {
    __shared__ unsigned long long s_sum[4][2];
    __shared__ unsigned s_max[4][2];
    typedef typename std::conditional<sizeof(T) == 4, double, unsigned long long>::type acc_t;
    unsigned mxl = 0, mxr = 0;
    acc_t sql = 0, sqr = 0;
    constexpr int V = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const size_t n = nframes * 2;
    const size_t nvec = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) ? n / V : 0;
    const size_t step = (size_t)gridDim.x * 256;
    if constexpr (sizeof(T) == 2) {
        typedef short s2 __attribute__((ext_vector_type(2)));
        typedef unsigned short u2 __attribute__((ext_vector_type(2)));
        u2 mx2 = {0, 0};
#define SH_FRAME(X_, A_, B_)                                                             \
            {                                                                            \
                const s2 v = __builtin_shufflevector(X_, X_, A_, B_);                    \
                const s2 neg = (s2){0, 0} - v;                                           \
                const u2 au = __builtin_bit_cast(u2, __builtin_elementwise_max(v, neg)); \
                mx2 = __builtin_elementwise_max(mx2, au);                                \
                sql += (unsigned long long)__umul24(au[0], au[0]);                       \
                sqr += (unsigned long long)__umul24(au[1], au[1]);                       \
            }
        constexpr int INFLIGHT = 4;                                 // neighbouring loads, as in k_absmax_sumsq
        const size_t bstep = step * INFLIGHT;
        size_t i = (size_t)blockIdx.x * 256 * INFLIGHT + threadIdx.x;
        for (; i + (INFLIGHT - 1) * 256 < nvec; i += bstep) {
            vec_t x[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) x[k] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(in) + i + k * 256);
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) { SH_FRAME(x[k], 0, 1) SH_FRAME(x[k], 2, 3) SH_FRAME(x[k], 4, 5) SH_FRAME(x[k], 6, 7) }
        }
        if (i < nvec) {
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) {
                if (i + k * 256 < nvec) {
                    const vec_t x = reinterpret_cast<const vec_t*>(in)[i + k * 256];
                    SH_FRAME(x, 0, 1) SH_FRAME(x, 2, 3) SH_FRAME(x, 4, 5) SH_FRAME(x, 6, 7)
                }
            }
        }
#undef SH_FRAME
        mxl = mx2[0];
        mxr = mx2[1];
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += step) {
            const vec_t x = reinterpret_cast<const vec_t*>(in)[i];
#pragma unroll
            for (int c = 0; c < V; c += 2) {
                const long long l = (long long)x[c], r = (long long)x[c + 1];
                const unsigned al = (unsigned)(l < 0 ? -l : l), ar = (unsigned)(r < 0 ? -r : r);
                mxl = al > mxl ? al : mxl;
                mxr = ar > mxr ? ar : mxr;
                if constexpr (sizeof(T) == 4) { sql += (double)l * (double)l; sqr += (double)r * (double)r; }
                else { sql += (unsigned long long)(l * l); sqr += (unsigned long long)(r * r); }
            }
        }
    }
    for (size_t f = nvec * V / 2 + (size_t)blockIdx.x * 256 + threadIdx.x; f < nframes; f += step) {
        const long long l = (long long)in[2 * f], r = (long long)in[2 * f + 1];
        const unsigned al = (unsigned)(l < 0 ? -l : l), ar = (unsigned)(r < 0 ? -r : r);
        mxl = al > mxl ? al : mxl;
        mxr = ar > mxr ? ar : mxr;
        if constexpr (sizeof(T) == 4) { sql += (double)l * (double)l; sqr += (double)r * (double)r; }
        else { sql += (unsigned long long)(l * l); sqr += (unsigned long long)(r * r); }
    }
    mxl = wave_max_u32(mxl);
    mxr = wave_max_u32(mxr);
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sql += __shfl_down(sql, o, 64); sqr += __shfl_down(sqr, o, 64); }
    } else {
        sql = wave_sum_u64(sql);
        sqr = wave_sum_u64(sqr);
    }
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_sum[wave][0] = __builtin_bit_cast(unsigned long long, sql);
        s_sum[wave][1] = __builtin_bit_cast(unsigned long long, sqr);
        s_max[wave][0] = mxl;
        s_max[wave][1] = mxr;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned c = threadIdx.x;
        unsigned m = s_max[0][c];
        for (int w = 1; w < 4; ++w) m = s_max[w][c] > m ? s_max[w][c] : m;
        acc_t t = __builtin_bit_cast(acc_t, s_sum[0][c]);
        for (int w = 1; w < 4; ++w) t += __builtin_bit_cast(acc_t, s_sum[w][c]);
        part[4 * (size_t)blockIdx.x + c] = (unsigned long long)m;
        part[4 * (size_t)blockIdx.x + 2 + c] = __builtin_bit_cast(unsigned long long, t);
    }
}

// one workgroup: out[0..3] = max L, max R, sum L^2, sum R^2 over the per-workgroup quadruples (F64: sums are float64,
// added in a fixed order: thread t takes workgroups t, t+256, ...; then the 64-lane tree; then the four waves)
template <bool F64>
__global__ __launch_bounds__(256) void k_stats_fold_stereo(const unsigned long long* __restrict__ part, unsigned nblocks,
                                                           unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s_sum[4][2];
    __shared__ unsigned s_max[4][2];
    typedef typename std::conditional<F64, double, unsigned long long>::type acc_t;
    unsigned mxl = 0, mxr = 0;
    acc_t sql = 0, sqr = 0;
    for (unsigned b = threadIdx.x; b < nblocks; b += 256) {
        const unsigned ml = (unsigned)part[4 * b], mr = (unsigned)part[4 * b + 1];
        mxl = ml > mxl ? ml : mxl;
        mxr = mr > mxr ? mr : mxr;
        sql += __builtin_bit_cast(acc_t, part[4 * b + 2]);
        sqr += __builtin_bit_cast(acc_t, part[4 * b + 3]);
    }
    mxl = wave_max_u32(mxl);
    mxr = wave_max_u32(mxr);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sql += __shfl_down(sql, o, 64); sqr += __shfl_down(sqr, o, 64); }
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_sum[wave][0] = __builtin_bit_cast(unsigned long long, sql);
        s_sum[wave][1] = __builtin_bit_cast(unsigned long long, sqr);
        s_max[wave][0] = mxl;
        s_max[wave][1] = mxr;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned c = threadIdx.x;
        unsigned m = s_max[0][c];
        for (int w = 1; w < 4; ++w) m = s_max[w][c] > m ? s_max[w][c] : m;
        acc_t t = __builtin_bit_cast(acc_t, s_sum[0][c]);
        for (int w = 1; w < 4; ++w) t += __builtin_bit_cast(acc_t, s_sum[w][c]);
        out[c] = (unsigned long long)m;
        out[2 + c] = __builtin_bit_cast(unsigned long long, t);
    }
}

// width 4: squares do not fit u64 sums exactly; accumulate the squares in float64 per thread, then a fixed
// tree -- close to, but not bit-identical with, audioop's sequential float64 sum (documented).
__global__ __launch_bounds__(256) void k_sumsq_f64(const int* __restrict__ in, size_t n, double* __restrict__ part) {
    __shared__ double s[256];
    double sq = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v = (double)in[i];
        sq += v * v;
    }
    s[threadIdx.x] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

// 24-bit <-> 32-bit: a thread converts four samples (12 bytes <-> 16 bytes); the 12 bytes are read / written as three
// dwords when the 3-byte stream starts on a dword boundary, byte by byte otherwise (the tail always is).
__global__ __launch_bounds__(256) void k_unpack24(const unsigned char* __restrict__ in, size_t n, int shift, int* __restrict__ out, int dwords) {
    const size_t q = sh::block_id() * 256 + threadIdx.x;              // group of four samples
    if (q * 4 >= n) return;
    if (dwords && q * 4 + 4 <= n) {
        const unsigned* w = reinterpret_cast<const unsigned*>(in) + q * 3;
        const unsigned a = w[0], b = w[1], c = w[2];
        int4 r;
        r.x = ((int)(a << 8)) >> 8;
        r.y = ((int)(((a >> 24) | (b << 8)) << 8)) >> 8;
        r.z = ((int)(((b >> 16) | (c << 16)) << 8)) >> 8;
        r.w = ((int)c) >> 8;
        r.x <<= shift; r.y <<= shift; r.z <<= shift; r.w <<= shift;
        if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) reinterpret_cast<int4*>(out)[q] = r;
        else { out[q * 4] = r.x; out[q * 4 + 1] = r.y; out[q * 4 + 2] = r.z; out[q * 4 + 3] = r.w; }
        return;
    }
    for (size_t i = q * 4; i < q * 4 + 4 && i < n; ++i) {
        const unsigned v = (unsigned)in[3 * i] | ((unsigned)in[3 * i + 1] << 8) | ((unsigned)in[3 * i + 2] << 16);
        out[i] = (((int)(v << 8)) >> 8) << shift;
    }
}

__global__ __launch_bounds__(256) void k_pack24(const int* __restrict__ in, size_t n, int shift, unsigned char* __restrict__ out, int dwords) {
    const size_t q = sh::block_id() * 256 + threadIdx.x;
    if (q * 4 >= n) return;
    if (dwords && q * 4 + 4 <= n) {
        const unsigned a = (unsigned)(in[q * 4] >> shift) & 0xFFFFFFu, b = (unsigned)(in[q * 4 + 1] >> shift) & 0xFFFFFFu;
        const unsigned c = (unsigned)(in[q * 4 + 2] >> shift) & 0xFFFFFFu, d = (unsigned)(in[q * 4 + 3] >> shift) & 0xFFFFFFu;
        unsigned* w = reinterpret_cast<unsigned*>(out) + q * 3;
        w[0] = a | (b << 24);
        w[1] = (b >> 8) | (c << 16);
        w[2] = (c >> 16) | (d << 8);
        return;
    }
    for (size_t i = q * 4; i < q * 4 + 4 && i < n; ++i) {
        const unsigned v = (unsigned)(in[i] >> shift);
        out[3 * i] = (unsigned char)v;
        out[3 * i + 1] = (unsigned char)(v >> 8);
        out[3 * i + 2] = (unsigned char)(v >> 16);
    }
}

template <typename F>
int dispatch_width(int width, F&& f) {
    if (width == 1) return f((signed char)0);
    if (width == 2) return f((short)0);
    if (width == 4) return f((int)0);
    return sh::set_error(SH_ERR_INVALID, "sample width %d not in {1,2,4}", width);
}

inline bool valid_width(int width) { return width == 1 || width == 2 || width == 4; }      // (width 3 is diverted to the 32-bit kernels before this test)
int bad_width(const char* who, int width) { return sh::set_error(SH_ERR_INVALID, "%s: sample width %d not in {1,2,4}", who, width); }

int check_io(const sh_buf* in, size_t in_off, size_t in_bytes, const sh_buf* out, size_t out_off, size_t out_bytes, const char* who) {
    if (!in || !out) return sh::set_error(SH_ERR_INVALID, "%s: NULL buffer", who);
    if (in_off > in->bytes || in_bytes > in->bytes - in_off) return sh::set_error(SH_ERR_INVALID, "%s: input range outside buffer", who);
    if (out_off > out->bytes || out_bytes > out->bytes - out_off) return sh::set_error(SH_ERR_INVALID, "%s: output range outside buffer", who);
    return SH_OK;
}

}  // namespace

namespace sh {
int unpack24(const void* in, size_t nsamples, int shift, int32_t* out) {
    if (!nsamples) return SH_OK;
    hipLaunchKernelGGL(k_unpack24, sh::grid1d((nsamples + 3) / 4, 256), dim3(256), 0, state().stream, (const unsigned char*)in, nsamples, shift,
                       (int*)out, ((uintptr_t)in & 3) == 0 ? 1 : 0);
    SH_CHECK_LAUNCH("k_unpack24");
    return SH_OK;
}
int pack24(const int32_t* in, size_t nsamples, int shift, void* out) {
    if (!nsamples) return SH_OK;
    hipLaunchKernelGGL(k_pack24, sh::grid1d((nsamples + 3) / 4, 256), dim3(256), 0, state().stream, (const int*)in, nsamples, shift,
                       (unsigned char*)out, ((uintptr_t)out & 3) == 0 ? 1 : 0);
    SH_CHECK_LAUNCH("k_pack24");
    return SH_OK;
}
}  // namespace sh

namespace {
// width 3 through the 32-bit kernels: `nin` samples unpacked (<< shift_in), op32(temporary in, temporary out), `nout` samples
// packed (>> shift_out) to out + out_off.  inplace: the 32-bit operation may write where it reads.
template <typename F>
int via32(const sh_buf* in, size_t in_off, size_t nin, int shift_in, sh_buf* out, size_t out_off, size_t nout, int shift_out,
          bool inplace, const char* who, F&& op32) {
    if (!in || !out) return sh::set_error(SH_ERR_INVALID, "%s: NULL buffer", who);
    if (in_off > in->bytes || nin * 3 > in->bytes - in_off) return sh::set_error(SH_ERR_INVALID, "%s: input range outside buffer", who);
    if (out_off > out->bytes || nout * 3 > out->bytes - out_off) return sh::set_error(SH_ERR_INVALID, "%s: output range outside buffer", who);
    sh::Temp tin, tout;
    int rc = tin.alloc(nin * 4);
    if (rc) return rc;
    if (!inplace) { rc = tout.alloc(nout * 4); if (rc) return rc; }
    rc = sh::unpack24((const char*)in->ptr + in_off, nin, shift_in, (int32_t*)tin.buf.ptr);
    if (rc) return rc;
    sh_buf* o32 = inplace ? &tin.buf : &tout.buf;
    rc = op32(&tin.buf, o32);
    if (rc) return rc;
    return sh::pack24((const int32_t*)o32->ptr, nout, shift_out, (char*)out->ptr + out_off);
}
}  // namespace

extern "C" {

int sh_pcm_mul(const sh_buf* in, size_t in_off, size_t nbytes, int width, double factor, sh_buf* out, size_t out_off) {
    SH_REQUIRE_INIT();
    if (width == 3) {
        if (nbytes % 3 || (in_off | out_off) % 3) return sh::set_error(SH_ERR_INVALID, "sh_pcm_mul: not a whole number of frames");
        const size_t n = nbytes / 3;
        return via32(in, in_off, n, 8, out, out_off, n, 8, true, "sh_pcm_mul",
                     [&](sh_buf* a, sh_buf* o) { return sh_pcm_mul(a, 0, n * 4, 4, factor, o, 0); });
    }
    if (!valid_width(width)) return bad_width("sh_pcm_mul", width);
    int rc = check_io(in, in_off, nbytes, out, out_off, nbytes, "sh_pcm_mul");
    if (rc) return rc;
    if (nbytes % width || (in_off | out_off) % width) return sh::set_error(SH_ERR_INVALID, "sh_pcm_mul: not a whole number of frames");
    if (!nbytes) return SH_OK;
    const char* ip = (const char*)in->ptr + in_off;
    char* op = (char*)out->ptr + out_off;
    hipStream_t st = sh::state().stream;
    return dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        constexpr int V = 16 / sizeof(T);
        const bool aligned = (((uintptr_t)ip | (uintptr_t)op) & 15) == 0;
        size_t nvec = aligned ? nbytes / 16 : 0, done = nvec * 16, rest = (nbytes - done) / sizeof(T);
        if (nvec && nbytes > sh::STREAM_BYTES) hipLaunchKernelGGL((k_mul<T, V, true>), sh::grid1d(nvec, 256), dim3(256), 0, st, (const T*)ip, (T*)op, nvec, factor);
        else if (nvec) hipLaunchKernelGGL((k_mul<T, V, false>), sh::grid1d(nvec, 256), dim3(256), 0, st, (const T*)ip, (T*)op, nvec, factor);
        if (rest) hipLaunchKernelGGL((k_mul<T, 1>), sh::grid1d(rest, 256), dim3(256), 0, st, (const T*)(ip + done), (T*)(op + done), rest, factor);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_mul");
    });
}

int sh_pcm_fade(const sh_buf* in, size_t in_off, size_t nbytes, int width, int fadeout, double slope, double offset,
                sh_buf* out, size_t out_off) {
    SH_REQUIRE_INIT();
    if (!valid_width(width)) return bad_width("sh_pcm_fade", width);
    int rc = check_io(in, in_off, nbytes, out, out_off, nbytes, "sh_pcm_fade");
    if (rc) return rc;
    if (nbytes % width) return sh::set_error(SH_ERR_INVALID, "sh_pcm_fade: not a whole number of samples");
    if (!nbytes) return SH_OK;
    const size_t n = nbytes / width;
    hipStream_t st = sh::state().stream;
    return dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        hipLaunchKernelGGL(k_fade<T>, sh::grid1d(n, 256), dim3(256), 0, st, (const T*)((const char*)in->ptr + in_off),
                           (T*)((char*)out->ptr + out_off), n, slope, (double)nbytes / (double)width, offset, fadeout);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_fade");
    });
}

int sh_pcm_modulate(const sh_buf* in, size_t nbytes, int width, const sh_buf* mod_f64, size_t nmod, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (!valid_width(width)) return bad_width("sh_pcm_modulate", width);
    int rc = check_io(in, 0, nbytes, out, 0, nbytes, "sh_pcm_modulate");
    if (rc) return rc;
    if (nbytes % width) return sh::set_error(SH_ERR_INVALID, "sh_pcm_modulate: not a whole number of samples");
    if (!nbytes) return SH_OK;
    if (!mod_f64 || !nmod || mod_f64->bytes / 8 < nmod) return sh::set_error(SH_ERR_INVALID, "sh_pcm_modulate: modulator buffer empty or smaller than nmod");
    const size_t n = nbytes / width;
    sh::State& S = sh::state();
    rc = dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        hipLaunchKernelGGL(k_modulate<T>, sh::grid1d(n, 256), dim3(256), 0, S.stream, (const T*)in->ptr, (T*)out->ptr, n,
                           (const double*)mod_f64->ptr, nmod, S.flag);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_modulate");
    });
    if (rc) return rc;
    SH_HIP(hipMemcpyAsync(S.flag_host, S.flag, sizeof(int), hipMemcpyDeviceToHost, S.stream));
    SH_HIP(hipStreamSynchronize(S.stream));
    if (S.flag_host[0]) {
        SH_HIP(hipMemsetAsync(S.flag, 0, sizeof(int), S.stream));
        return sh::set_error(SH_ERR_OVERFLOW, "signed integer out of range for sample width %d", width);
    }
    return SH_OK;
}

int sh_pcm_pan_lfo(const sh_buf* in, size_t nframes, int width, int nchannels, const sh_buf* pan_f64, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (width != 1 && width != 2 && width != 4) return sh::set_error(SH_ERR_INVALID, "sh_pcm_pan_lfo: width %d not in {1,2,4}", width);
    if (nchannels != 1 && nchannels != 2) return sh::set_error(SH_ERR_INVALID, "sh_pcm_pan_lfo: %d channels (1 or 2)", nchannels);
    int rc = check_io(in, 0, nframes * width * nchannels, out, 0, nframes * width * 2, "sh_pcm_pan_lfo");
    if (rc) return rc;
    if (!nframes) return SH_OK;
    if (!pan_f64 || pan_f64->bytes / 8 < nframes) return sh::set_error(SH_ERR_INVALID, "sh_pcm_pan_lfo: fewer pan values than frames");
    sh::State& S = sh::state();
    rc = dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        if (nchannels == 1)
            hipLaunchKernelGGL((k_pan_lfo<T, 1>), sh::grid1d(nframes, 256), dim3(256), 0, S.stream, (const T*)in->ptr, (T*)out->ptr, nframes,
                               (const double*)pan_f64->ptr, S.flag);
        else
            hipLaunchKernelGGL((k_pan_lfo<T, 2>), sh::grid1d(nframes, 256), dim3(256), 0, S.stream, (const T*)in->ptr, (T*)out->ptr, nframes,
                               (const double*)pan_f64->ptr, S.flag);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_pan_lfo");
    });
    if (rc) return rc;
    SH_HIP(hipMemcpyAsync(S.flag_host, S.flag, sizeof(int), hipMemcpyDeviceToHost, S.stream));
    SH_HIP(hipStreamSynchronize(S.stream));
    if (S.flag_host[0]) {
        SH_HIP(hipMemsetAsync(S.flag, 0, sizeof(int), S.stream));
        return sh::set_error(SH_ERR_OVERFLOW, "signed integer out of range for sample width %d", width);
    }
    return SH_OK;
}

int sh_pcm_to_f64(const sh_buf* in, size_t nsamples, int width, double divisor, sh_buf* out_f64) {
    SH_REQUIRE_INIT();
    if (width != 1 && width != 2 && width != 4) return sh::set_error(SH_ERR_INVALID, "sh_pcm_to_f64: width %d not in {1,2,4}", width);
    int rc = check_io(in, 0, nsamples * width, out_f64, 0, nsamples * 8, "sh_pcm_to_f64");
    if (rc) return rc;
    if (!(divisor != 0.0)) return sh::set_error(SH_ERR_INVALID, "sh_pcm_to_f64: divisor is zero");
    if (!nsamples) return SH_OK;
    hipStream_t st = sh::state().stream;
    return dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        hipLaunchKernelGGL(k_to_f64<T>, sh::grid1d(nsamples, 256), dim3(256), 0, st, (const T*)in->ptr, (double*)out_f64->ptr, nsamples, divisor);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_to_f64");
    });
}

int sh_pcm_bias(const sh_buf* in, size_t nbytes, int width, int bias, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (width == 3) {
        if (nbytes % 3) return sh::set_error(SH_ERR_INVALID, "sh_pcm_bias: not a whole number of frames");
        const size_t n = nbytes / 3;
        const int bias32 = (int)(((unsigned)bias & 0xFFFFFFu) << 8);           // (v + bias) mod 2^24 == ((v << 8) + (bias << 8) mod 2^32) >> 8
        return via32(in, 0, n, 8, out, 0, n, 8, true, "sh_pcm_bias", [&](sh_buf* a, sh_buf* o) { return sh_pcm_bias(a, n * 4, 4, bias32, o); });
    }
    if (!valid_width(width)) return bad_width("sh_pcm_bias", width);
    int rc = check_io(in, 0, nbytes, out, 0, nbytes, "sh_pcm_bias");
    if (rc) return rc;
    if (nbytes % width) return sh::set_error(SH_ERR_INVALID, "sh_pcm_bias: not a whole number of frames");
    if (!nbytes) return SH_OK;
    const size_t n = nbytes / width;
    hipStream_t st = sh::state().stream;
    return dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        hipLaunchKernelGGL(k_bias<T>, sh::grid1d(n, 256), dim3(256), 0, st, (const T*)in->ptr, (T*)out->ptr, n, bias);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_bias");
    });
}

int sh_pcm_reverse(const sh_buf* in, size_t nbytes, int width, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (width == 3) {
        if (nbytes % 3) return sh::set_error(SH_ERR_INVALID, "sh_pcm_reverse: not a whole number of frames");
        const size_t n = nbytes / 3;
        return via32(in, 0, n, 0, out, 0, n, 0, false, "sh_pcm_reverse", [&](sh_buf* a, sh_buf* o) { return sh_pcm_reverse(a, n * 4, 4, o); });
    }
    if (!valid_width(width)) return bad_width("sh_pcm_reverse", width);
    int rc = check_io(in, 0, nbytes, out, 0, nbytes, "sh_pcm_reverse");
    if (rc) return rc;
    if (nbytes % width) return sh::set_error(SH_ERR_INVALID, "sh_pcm_reverse: not a whole number of frames");
    if (!nbytes) return SH_OK;
    if (in == out || in->ptr == out->ptr) return sh::set_error(SH_ERR_INVALID, "sh_pcm_reverse: cannot reverse in place");
    const size_t n = nbytes / width;
    hipStream_t st = sh::state().stream;
    return dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        hipLaunchKernelGGL(k_reverse<T>, sh::grid1d(n, 256), dim3(256), 0, st, (const T*)in->ptr, (T*)out->ptr, n);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_reverse");
    });
}

int sh_pcm_tomono(const sh_buf* in, size_t nframes, int width, double lfactor, double rfactor, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (width == 3)
        return via32(in, 0, nframes * 2, 8, out, 0, nframes, 8, false, "sh_pcm_tomono",
                     [&](sh_buf* a, sh_buf* o) { return sh_pcm_tomono(a, nframes, 4, lfactor, rfactor, o); });
    if (!valid_width(width)) return bad_width("sh_pcm_tomono", width);
    int rc = check_io(in, 0, nframes * 2 * width, out, 0, nframes * width, "sh_pcm_tomono");
    if (rc) return rc;
    if (!nframes) return SH_OK;
    hipStream_t st = sh::state().stream;
    return dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        constexpr int F = 8 / sizeof(T);                       // frames per 16-byte load
        const size_t nvec = nframes / F, done = nvec * F;
        if (nvec) hipLaunchKernelGGL((k_tomono<T, F>), sh::grid1d(nvec, 256), dim3(256), 0, st, (const T*)in->ptr, (T*)out->ptr, nvec, lfactor, rfactor);
        if (nframes > done) hipLaunchKernelGGL((k_tomono<T, 1>), sh::grid1d(nframes - done, 256), dim3(256), 0, st,
                                               (const T*)in->ptr + 2 * done, (T*)out->ptr + done, nframes - done, lfactor, rfactor);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_tomono");
    });
}

int sh_pcm_tostereo(const sh_buf* in, size_t nframes, int width, double lfactor, double rfactor, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (width == 3)
        return via32(in, 0, nframes, 8, out, 0, nframes * 2, 8, false, "sh_pcm_tostereo",
                     [&](sh_buf* a, sh_buf* o) { return sh_pcm_tostereo(a, nframes, 4, lfactor, rfactor, o); });
    if (!valid_width(width)) return bad_width("sh_pcm_tostereo", width);
    int rc = check_io(in, 0, nframes * width, out, 0, nframes * 2 * width, "sh_pcm_tostereo");
    if (rc) return rc;
    if (!nframes) return SH_OK;
    hipStream_t st = sh::state().stream;
    return dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        constexpr int F = 8 / sizeof(T);                       // frames per 16-byte store
        const size_t nvec = nframes / F, done = nvec * F;
        if (nvec) hipLaunchKernelGGL((k_tostereo<T, F>), sh::grid1d(nvec, 256), dim3(256), 0, st, (const T*)in->ptr, (T*)out->ptr, nvec, lfactor, rfactor);
        if (nframes > done) hipLaunchKernelGGL((k_tostereo<T, 1>), sh::grid1d(nframes - done, 256), dim3(256), 0, st,
                                               (const T*)in->ptr + done, (T*)out->ptr + 2 * done, nframes - done, lfactor, rfactor);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_tostereo");
    });
}

int sh_pcm_lin2lin(const sh_buf* in, size_t nsamples, int width, int new_width, sh_buf* out) {
    SH_REQUIRE_INIT();
    if (width == 3 || new_width == 3) {
        if (!in || !out) return sh::set_error(SH_ERR_INVALID, "sh_pcm_lin2lin: NULL buffer");
        if ((width != 1 && width != 2 && width != 3 && width != 4) || (new_width != 1 && new_width != 2 && new_width != 3 && new_width != 4))
            return sh::set_error(SH_ERR_INVALID, "sh_pcm_lin2lin: widths %d -> %d not in {1,2,3,4}", width, new_width);
        if (nsamples * (size_t)width > in->bytes || nsamples * (size_t)new_width > out->bytes)
            return sh::set_error(SH_ERR_INVALID, "sh_pcm_lin2lin: range outside buffer");
        if (!nsamples) return SH_OK;
        // GETSAMPLE32 of a 24-bit sample is value << 8; SETSAMPLE32 to 24 bits keeps value >> 8
        sh::Temp t32;
        int rc3 = t32.alloc(nsamples * 4);
        if (rc3) return rc3;
        rc3 = width == 3 ? sh::unpack24(in->ptr, nsamples, 8, (int32_t*)t32.buf.ptr) : sh_pcm_lin2lin(in, nsamples, width, 4, &t32.buf);
        if (rc3) return rc3;
        return new_width == 3 ? sh::pack24((const int32_t*)t32.buf.ptr, nsamples, 8, out->ptr) : sh_pcm_lin2lin(&t32.buf, nsamples, 4, new_width, out);
    }
    int rc = check_io(in, 0, nsamples * width, out, 0, nsamples * new_width, "sh_pcm_lin2lin");
    if (rc) return rc;
    if (!nsamples) return SH_OK;
    hipStream_t st = sh::state().stream;
    const dim3 grid = sh::grid1d(nsamples, 256);
#define SH_L2L(TI, TO) hipLaunchKernelGGL((k_lin2lin<TI, TO>), grid, dim3(256), 0, st, (const TI*)in->ptr, (TO*)out->ptr, nsamples)
    if (width == 1 && new_width == 1) SH_L2L(signed char, signed char);
    else if (width == 1 && new_width == 2) SH_L2L(signed char, short);
    else if (width == 1 && new_width == 4) SH_L2L(signed char, int);
    else if (width == 2 && new_width == 1) SH_L2L(short, signed char);
    else if (width == 2 && new_width == 2) SH_L2L(short, short);
    else if (width == 2 && new_width == 4) SH_L2L(short, int);
    else if (width == 4 && new_width == 1) SH_L2L(int, signed char);
    else if (width == 4 && new_width == 2) SH_L2L(int, short);
    else if (width == 4 && new_width == 4) SH_L2L(int, int);
    else return sh::set_error(SH_ERR_INVALID, "sh_pcm_lin2lin: widths %d -> %d not in {1,2,4}", width, new_width);
#undef SH_L2L
    SH_CHECK_LAUNCH("k_lin2lin");
    return SH_OK;
}

int sh_pcm_stats(const sh_buf* in, size_t nbytes, int width, uint32_t* max_abs, double* sum_squares) {
    SH_REQUIRE_INIT();
    if (width == 3) {                           // raw 24-bit values (audioop.max / rms read GETRAWSAMPLE)
        if (!in || nbytes > in->bytes || nbytes % 3) return sh::set_error(SH_ERR_INVALID, "sh_pcm_stats: range outside buffer / not whole samples");
        const size_t n = nbytes / 3;
        sh::Temp t32;
        int rc3 = t32.alloc(n * 4);
        if (rc3) return rc3;
        rc3 = sh::unpack24(in->ptr, n, 0, (int32_t*)t32.buf.ptr);
        if (rc3) return rc3;
        return sh_pcm_stats(&t32.buf, n * 4, 4, max_abs, sum_squares);
    }
    if (!valid_width(width)) return bad_width("sh_pcm_stats", width);
    if (!in || nbytes > in->bytes) return sh::set_error(SH_ERR_INVALID, "sh_pcm_stats: range outside buffer");
    if (nbytes % width) return sh::set_error(SH_ERR_INVALID, "sh_pcm_stats: not a whole number of frames");
    if (max_abs) *max_abs = 0;
    if (sum_squares) *sum_squares = 0.0;
    if (!nbytes) return SH_OK;
    const size_t n = nbytes / width;
    // 512 workgroups x 4 neighbouring 16-byte loads in flight per lane: 6.0 TB/s on 900 MB; 4096 workgroups with the four
    // loads a grid apart read the same data at 4.2-4.8
    const unsigned blocks = n / 8192 < 512 ? (unsigned)(n / 8192 + 1) : 512u;
    int rc = sh::ensure_scratch(16 + (size_t)blocks * 24);
    if (rc) return rc;
    hipStream_t st = sh::state().stream;
    unsigned long long* acc = (unsigned long long*)sh::state().scratch;          // [0..1]: result; then per-workgroup pairs
    unsigned long long* pairs = acc + 2;
    double* part = (double*)(pairs + 2 * (size_t)blocks);
    rc = dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        hipLaunchKernelGGL((k_absmax_sumsq<T, 4>), dim3(blocks), dim3(256), 0, st, (const T*)in->ptr, n, pairs);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_absmax_sumsq");
    });
    if (rc) return rc;
    hipLaunchKernelGGL(k_stats_fold, dim3(1), dim3(256), 0, st, (const unsigned long long*)pairs, blocks, acc);
    SH_CHECK_LAUNCH("k_stats_fold");
    if (width == 4) {
        hipLaunchKernelGGL(k_sumsq_f64, dim3(blocks), dim3(256), 0, st, (const int*)in->ptr, n, part);
        SH_CHECK_LAUNCH("k_sumsq_f64");
    }
    unsigned long long host_acc[2];
    SH_HIP(hipMemcpyAsync(host_acc, acc, 16, hipMemcpyDeviceToHost, st));
    std::vector<double> host_part;
    if (width == 4) {
        host_part.resize(blocks);
        SH_HIP(hipMemcpyAsync(host_part.data(), part, (size_t)blocks * 8, hipMemcpyDeviceToHost, st));
    }
    SH_HIP(hipStreamSynchronize(st));
    if (max_abs) *max_abs = (uint32_t)host_acc[0];
    if (sum_squares) {
        if (width == 4) {
            double t = 0.0;
            for (double p : host_part) t += p;
            *sum_squares = t;
        } else {
            *sum_squares = (double)host_acc[1];
        }
    }
    return SH_OK;
}

int sh_pcm_stats_stereo(const sh_buf* in, size_t nframes, int width, uint32_t max_abs[2], double sum_squares[2]) {
    SH_REQUIRE_INIT();
    if (width == 3) {
        if (!in || nframes > in->bytes / 6) return sh::set_error(SH_ERR_INVALID, "sh_pcm_stats_stereo: range outside buffer");
        sh::Temp t32;
        int rc3 = t32.alloc(nframes * 8);
        if (rc3) return rc3;
        rc3 = sh::unpack24(in->ptr, nframes * 2, 0, (int32_t*)t32.buf.ptr);
        if (rc3) return rc3;
        return sh_pcm_stats_stereo(&t32.buf, nframes, 4, max_abs, sum_squares);
    }
    if (width != 1 && width != 2 && width != 4) return sh::set_error(SH_ERR_INVALID, "sh_pcm_stats_stereo: width %d not in {1,2,4}", width);
    if (!in || nframes > in->bytes / (2 * (size_t)width)) return sh::set_error(SH_ERR_INVALID, "sh_pcm_stats_stereo: range outside buffer");
    if (max_abs) max_abs[0] = max_abs[1] = 0;
    if (sum_squares) sum_squares[0] = sum_squares[1] = 0.0;
    if (!nframes) return SH_OK;
    const unsigned blocks = nframes / 4096 < 512 ? (unsigned)(nframes / 4096 + 1) : 512u;
    int rc = sh::ensure_scratch(32 + (size_t)blocks * 32);
    if (rc) return rc;
    hipStream_t st = sh::state().stream;
    unsigned long long* acc = (unsigned long long*)sh::state().scratch;          // [0..3]: result; then per-workgroup quadruples
    unsigned long long* quads = acc + 4;
    rc = dispatch_width(width, [&](auto tag) {
        typedef decltype(tag) T;
        hipLaunchKernelGGL(k_stats_stereo<T>, dim3(blocks), dim3(256), 0, st, (const T*)in->ptr, nframes, quads);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? (int)SH_OK : sh::hip_error(e, "k_stats_stereo");
    });
    if (rc) return rc;
    if (width == 4) hipLaunchKernelGGL(k_stats_fold_stereo<true>, dim3(1), dim3(256), 0, st, (const unsigned long long*)quads, blocks, acc);
    else hipLaunchKernelGGL(k_stats_fold_stereo<false>, dim3(1), dim3(256), 0, st, (const unsigned long long*)quads, blocks, acc);
    SH_CHECK_LAUNCH("k_stats_fold_stereo");
    unsigned long long host_acc[4];
    SH_HIP(hipMemcpyAsync(host_acc, acc, 32, hipMemcpyDeviceToHost, st));
    SH_HIP(hipStreamSynchronize(st));
    for (int c = 0; c < 2; ++c) {
        if (max_abs) max_abs[c] = (uint32_t)host_acc[c];
        if (sum_squares) sum_squares[c] = width == 4 ? __builtin_bit_cast(double, host_acc[2 + c]) : (double)host_acc[2 + c];
    }
    return SH_OK;
}

}  // extern "C"
