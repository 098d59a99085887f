// libsynthhost.so: host-side table building in native code (include/synthhost.h).  Plain C++, no HIP.
//
// shh_phase_table is synthesizer_amd/phasetable.py build_phase_table statement by statement: the same float64 additions (x86-64
// SSE2 arithmetic is IEEE double, the file is compiled with -ffp-contract=off and without fast-math), frexp / ldexp are exact, the
// mantissas are 53-bit integers in int64.  tests/test_phasetable.py compares the two on random sequences, piece by piece.
#include "../../include/synthhost.h"
#include <cmath>

#ifndef SHH_SOURCE_HASH
#define SHH_SOURCE_HASH "unknown"
#endif

namespace {
constexpr int64_t LO = (int64_t)1 << 52, HI = ((int64_t)1 << 53) - 1;
constexpr int MAX_SINGLE_PIECES = 1 << 16;
}

extern "C" {

const char* shh_version(void) { return "synthhost 0.1 src:" SHH_SOURCE_HASH; }

int shh_phase_table(double t0, double inc, uint64_t n_limit, shh_segment* out, int cap) {
    const double TINY = std::ldexp(1.0, -1000);
    int count = 0;
    uint64_t n = 0;
    volatile double tv = t0;          // (volatile: every sum below is rounded to float64 before it is compared)
    double t = tv;
#define SHH_PUT(N0, T0, DT) do { if (count >= cap) return -1; out[count].n0 = (N0); out[count].t0 = (T0); out[count].dt = (DT); ++count; } while (0)
    while (n < n_limit) {
        volatile double s1 = t + inc;
        double t1 = s1;
        if (t1 == t) {                // inc == 0, or absorbed below half an ulp: constant forever
            SHH_PUT(n, t, 0.0);
            break;
        }
        if (t >= TINY || t <= -TINY) {
            int e;
            const double m = std::frexp(t, &e);
            const double hi = std::ldexp(1.0, e), lo = 0.5 * hi;      // the binade of t: lo <= |x| < hi, with t's sign
            const bool pos = t > 0;
            if (pos ? (lo <= t1 && t1 < hi) : (-hi < t1 && t1 <= -lo)) {
                volatile double s2 = t1 + inc;
                const double t2 = s2;
                volatile double dv = t1 - t;                           // exact: same binade
                const double d = dv;
                volatile double d2v = t2 - t1;
                if ((pos ? (lo <= t2 && t2 < hi) : (-hi < t2 && t2 <= -lo)) && d2v == d) {
                    // regular run: t + j*d for j = 0..k stays inside the binade
                    const int s = e - 53;
                    const int64_t T = (int64_t)std::ldexp(m, 53);
                    const int64_t D = (int64_t)std::ldexp(d, -s);
                    int64_t k;
                    if (T > 0) k = D > 0 ? (HI - T) / D : (T - LO) / (-D);
                    else k = D < 0 ? (HI + T) / (-D) : (-T - LO) / D;
                    SHH_PUT(n, t, d);
                    const double tk = std::ldexp((double)(T + k * D), s);     // exact: |T + k*D| < 2^53
                    volatile double so = tk + inc;                             // the step out of the binade: a real addition
                    t1 = so;
                    if (t1 == tk) {
                        SHH_PUT(n + (uint64_t)k, tk, 0.0);
                        break;
                    }
                    n += (uint64_t)k + 1;
                    t = t1;
                    continue;
                }
            }
        }
        // single step (entering / leaving a binade, crossing zero, irregular first step)
        volatile double dd = t1 - t;
        SHH_PUT(n, t, dd);
        n += 1;
        t = t1;
        if (count > MAX_SINGLE_PIECES) return -2;
    }
#undef SHH_PUT
    return count;
}

}  // extern "C"
