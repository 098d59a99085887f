// Host build of synthesizer_amd/csrc/devmath.hpp for tests/test_devmath.py (g++, no GPU).
#include "../synthesizer_amd/csrc/devmath.hpp"
#include <vector>

static shm::sc_pair g_tab[shm::TRIG_N];
static bool g_init = false;

static void init() {
    if (g_init) return;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < shm::TRIG_N; ++k) {
        long double a = two_pi * (long double)k / (long double)shm::TRIG_N;
        g_tab[k].s = (double)sinl(a);
        g_tab[k].c = (double)cosl(a);
    }
    g_init = true;
}

extern "C" {
void dm_sincos(const double* t, int n, double* s, double* c) { init(); for (int i = 0; i < n; ++i) shm::sincos_tab(t[i], g_tab, s[i], c[i]); }
void dm_sincos_poly(const double* t, int n, double* s, double* c) { for (int i = 0; i < n; ++i) shm::sincos_f64(t[i], s[i], c[i]); }
void dm_sin(const double* t, int n, double* s) { for (int i = 0; i < n; ++i) s[i] = shm::sin_f64(t[i]); }
void dm_cos(const double* t, int n, double* c) { for (int i = 0; i < n; ++i) c[i] = shm::cos_f64(t[i]); }
void dm_saw(const double* t, int n, double amp2, double bias, double* o) { for (int i = 0; i < n; ++i) o[i] = shm::saw_value(t[i], amp2, bias); }
void dm_square(const double* t, int n, double amp, double bias, double* o) { for (int i = 0; i < n; ++i) o[i] = shm::square_value(t[i], amp, bias); }
void dm_pulse(const double* t, int n, double pw, double amp, double bias, double* o) { for (int i = 0; i < n; ++i) o[i] = shm::pulse_value(t[i], pw, amp, bias); }
}
