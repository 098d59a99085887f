// devmath.hpp -- float64 trigonometry and waveform formulas for the oscillator kernels.
//
// Everything here is written so that the translation unit can be built with
// -ffp-contract=off: a fused multiply-add appears only where fma() is written.
// The functions are __host__ __device__ so tests/ can compile this header with g++
// and compare it with libm on the CPU (tests/test_devmath.py) -- the product only
// ever runs them on the GPU.
//
// Accuracy target: |error| <= ~4e-16 absolute for |t| < 3e9 rad, i.e. the float32
// value written to HBM is the correctly rounded float32 of the reference's float64
// sample in all but ~1e-8 of cases.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define SH_HD __host__ __device__ __forceinline__
#else
#define SH_HD static inline
#endif

namespace shm {

// pi split in three doubles (Cody-Waite); with fma every partial product is exact enough
// that r = t - n*pi carries < 2e-16 absolute error for |n| < 2^31.
constexpr double PI_1 = 0x1.921fb54442d18p+1;
constexpr double PI_2 = 0x1.1a62633145c07p-53;
constexpr double PI_3 = -0x1.f1976b7ed8fbcp-109;
constexpr double INV_PI = 0.3183098861837907;
constexpr double PIO2_1 = 0x1.921fb54442d18p+0;
constexpr double PIO2_2 = 0x1.1a62633145c07p-54;
constexpr double PIO2_3 = -0x1.f1976b7ed8fbcp-110;
constexpr double TWO_OVER_PI = 0.6366197723675814;

SH_HD double flip_sign(double x, int odd) {
    // x * (-1)^odd without a multiply
    union { double d; uint64_t u; } v;
    v.d = x;
    v.u ^= ((uint64_t)(odd & 1)) << 63;
    return v.d;
}

// r in [-pi/2, pi/2], returns n (t = n*pi + r)
SH_HD double reduce_pi(double t, int& n) {
    double fn = rint(t * INV_PI);
    double r = fma(-fn, PI_1, t);
    r = fma(-fn, PI_2, r);
    r = fma(-fn, PI_3, r);
    n = (int)fn;
    return r;
}

// odd Taylor polynomial of sin on [-pi/2, pi/2], degree 21 (truncation < 2e-18)
SH_HD double sin_poly_halfpi(double r) {
    double z = r * r;
    double p = 1.9572941063391263e-20;
    p = fma(p, z, -8.22063524662433e-18);
    p = fma(p, z, 2.8114572543455206e-15);
    p = fma(p, z, -7.647163731819816e-13);
    p = fma(p, z, 1.6059043836821613e-10);
    p = fma(p, z, -2.505210838544172e-08);
    p = fma(p, z, 2.7557319223985893e-06);
    p = fma(p, z, -0.0001984126984126984);
    p = fma(p, z, 0.008333333333333333);
    p = fma(p, z, -0.16666666666666666);
    return fma(r * z, p, r);
}

// even Taylor polynomial of cos on [-pi/2, pi/2], degree 22
SH_HD double cos_poly_halfpi(double r) {
    double z = r * r;
    double p = -8.896791392450574e-22;      // -1/22!
    p = fma(p, z, 4.110317623312165e-19);
    p = fma(p, z, -1.5619206968586225e-16);
    p = fma(p, z, 4.779477332387385e-14);
    p = fma(p, z, -1.1470745597729725e-11);
    p = fma(p, z, 2.08767569878681e-09);
    p = fma(p, z, -2.755731922398589e-07);
    p = fma(p, z, 2.48015873015873e-05);
    p = fma(p, z, -0.001388888888888889);
    p = fma(p, z, 0.041666666666666664);
    p = fma(p, z, -0.5);
    return fma(z, p, 1.0);
}

SH_HD double sin_f64(double t) {
    int n;
    double r = reduce_pi(t, n);
    return flip_sign(sin_poly_halfpi(r), n);
}

SH_HD double cos_f64(double t) {
    int n;
    double r = reduce_pi(t, n);
    return flip_sign(cos_poly_halfpi(r), n);
}

// sin and cos together: quadrant reduction to [-pi/4, pi/4], degree 15/16 Taylor
SH_HD void sincos_f64(double t, double& s, double& c) {
    double fn = rint(t * TWO_OVER_PI);
    double r = fma(-fn, PIO2_1, t);
    r = fma(-fn, PIO2_2, r);
    r = fma(-fn, PIO2_3, r);
    int q = (int)fn;
    double z = r * r;
    double ps = -7.647163731819816e-13;
    ps = fma(ps, z, 1.6059043836821613e-10);
    ps = fma(ps, z, -2.505210838544172e-08);
    ps = fma(ps, z, 2.7557319223985893e-06);
    ps = fma(ps, z, -0.0001984126984126984);
    ps = fma(ps, z, 0.008333333333333333);
    ps = fma(ps, z, -0.16666666666666666);
    double sr = fma(r * z, ps, r);
    double pc = 4.779477332387385e-14;
    pc = fma(pc, z, -1.1470745597729725e-11);
    pc = fma(pc, z, 2.08767569878681e-09);
    pc = fma(pc, z, -2.755731922398589e-07);
    pc = fma(pc, z, 2.48015873015873e-05);
    pc = fma(pc, z, -0.001388888888888889);
    pc = fma(pc, z, 0.041666666666666664);
    pc = fma(pc, z, -0.5);
    double cr = fma(z, pc, 1.0);
    // q mod 4: 0 -> (s, c), 1 -> (c, -s), 2 -> (-s, -c), 3 -> (-c, s)
    double s0 = (q & 1) ? cr : sr;
    double c0 = (q & 1) ? sr : cr;
    s = flip_sign(s0, (q >> 1) & 1);
    c = flip_sign(c0, ((q + 1) >> 1) & 1);
}

// ---- table-driven sincos ------------------------------------------------------------------
// t = k*(2pi/N) + r with k = rint(t*N/2pi), |r| <= pi/N; (sin, cos)(k*2pi/N) comes from an N-entry
// table (LDS on the GPU), the small angle from degree-5/4 Taylor polynomials, and one rotation
// combines them.  6 constants instead of ~30: on gfx950 every float64 constant costs scalar-ALU
// instructions to materialise, and the scalar unit (one per CU) was the bottleneck of the long
// polynomials above.  Error: table entries are correctly rounded doubles, the truncation of the
// polynomials is < 1e-16 for N = 512, the rotation adds ~2 ulp.
constexpr int    TRIG_N = 512;
constexpr double TRIG_STEP_1 = PI_1 / 256.0;        // 2pi/512 split like pi (power-of-two scaling is exact)
constexpr double TRIG_STEP_2 = PI_2 / 256.0;
constexpr double TRIG_STEP_3 = PI_3 / 256.0;
constexpr double TRIG_INV_STEP = 81.48733086305042;  // 512/(2pi)

struct sc_pair { double s, c; };

template <typename TablePtr>
SH_HD void sincos_tab(double t, TablePtr tab, double& s, double& c) {
    // m = nearest integer to t*N/2pi, obtained by adding 1.5*2^52: the low mantissa bits of m ARE that
    // integer (two's complement), so no rint / float->int conversion is needed; valid for |t*N/2pi| < 2^51
    union { double d; uint64_t u; } m;
    m.d = fma(t, TRIG_INV_STEP, 6755399441055744.0);
    const double fk = m.d - 6755399441055744.0;
    double r = fma(-fk, TRIG_STEP_1, t);       // two-term Cody-Waite: the third term is < 1e-19 for |fk| < 2^51
    r = fma(-fk, TRIG_STEP_2, r);
    const uint32_t k = (uint32_t)m.u & (uint32_t)(TRIG_N - 1);
    const double S = tab[k].s, C = tab[k].c;
    const double z = r * r;
    const double sr = fma(r * z, fma(z, 0.008333333333333333, -0.16666666666666666), r);
    const double cr = fma(z, fma(z, 0.041666666666666664, -0.5), 1.0);
    s = fma(S, cr, C * sr);
    c = fma(C, cr, -(S * sr));
}

// The same for N independent angles in lockstep: each stage is issued for all angles before the next
// stage starts, so the table reads are in flight together and the dependent float64 chains interleave.
template <int N, typename TablePtr>
SH_HD void sincos_tab_n(const double (&t)[N], TablePtr tab, double (&s)[N], double (&c)[N]) {
    double r[N], S[N], C[N];
    for (int j = 0; j < N; ++j) {
        union { double d; uint64_t u; } m;
        m.d = fma(t[j], TRIG_INV_STEP, 6755399441055744.0);
        const double fk = m.d - 6755399441055744.0;
        const uint32_t k = (uint32_t)m.u & (uint32_t)(TRIG_N - 1);
        S[j] = tab[k].s;
        C[j] = tab[k].c;
        r[j] = fma(-fk, TRIG_STEP_2, fma(-fk, TRIG_STEP_1, t[j]));
    }
    double z[N], sr[N], cr[N];
    for (int j = 0; j < N; ++j) z[j] = r[j] * r[j];
    for (int j = 0; j < N; ++j) sr[j] = fma(z[j], 0.008333333333333333, -0.16666666666666666);
    for (int j = 0; j < N; ++j) cr[j] = fma(z[j], 0.041666666666666664, -0.5);
    for (int j = 0; j < N; ++j) sr[j] = fma(r[j] * z[j], sr[j], r[j]);
    for (int j = 0; j < N; ++j) cr[j] = fma(z[j], cr[j], 1.0);
    for (int j = 0; j < N; ++j) s[j] = fma(S[j], cr[j], C[j] * sr[j]);
    for (int j = 0; j < N; ++j) c[j] = fma(C[j], cr[j], -(S[j] * sr[j]));
}

// ---- waveforms (formulas of oscillators.py, operation order preserved) -----------------

// Sawtooth: bias + amplitude*2.0*(t - floor(0.5+t))
SH_HD double saw_value(double t, double amp2, double bias) {
    double fl = floor(0.5 + t);
    return bias + amp2 * (t - fl);
}

// Square: (-amplitude if int(t*2) % 2 else amplitude) + bias ; int() truncates toward zero
SH_HD double square_value(double t, double amp, double bias) {
    double tr = trunc(t * 2.0);
    double h = tr * 0.5;
    bool odd = (h != rint(h)) && (fabs(tr) < 9007199254740992.0);
    return (odd ? -amp : amp) + bias;
}

// Pulse: (amplitude if t % 1.0 < pulsewidth else -amplitude) + bias ; Python float modulo
SH_HD double pulse_value(double t, double pw, double amp, double bias) {
    double m = t - trunc(t);          // fmod(t, 1.0), exact
    if (m < 0.0) m = m + 1.0;         // Python: result takes the sign of the divisor
    return ((m < pw) ? amp : -amp) + bias;
}

// Triangle: 4.0*amplitude*(abs((t+0.75) % 1.0 - 0.5) - 0.25) + bias ; Python float modulo
SH_HD double triangle_value(double t, double amp4, double bias) {
    double u = t + 0.75;
    double m = u - trunc(u);
    if (m < 0.0) m = m + 1.0;
    return amp4 * (fabs(m - 0.5) - 0.25) + bias;
}

}  // namespace shm
