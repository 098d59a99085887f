/*
 * oracle.c -- C restatement of the hot path, for parity checks at sizes the pure-Python oracle
 * (oracle/synth_oracle.py, oracle/pcm_oracle.py) cannot finish in seconds.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's cpu_baseline leg and
 * __graft_entry__.smoke() may load liboracle.so.  The product (synthesizer_amd/) never does.
 *
 * PARITY STATUS: the oscillator functions restate oracle/synth_oracle.py statement by statement
 * (sequential float64 loops, libm sin) and are checked against it in tests/test_oracle_c.py; the
 * reference tree itself is not mounted (/root/reference/README.md:1-2), so these rows stay
 * "parity unpinned".  or_add / or_ratecv restate CPython 3.10 Modules/audioop.c
 * (audioop_add_impl, audioop_ratecv_impl) and are checked against the live module.
 *
 * Build:  make -C oracle      (gcc -O2 -ffp-contract=off; no FMA contraction, like CPython's build)
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

enum { K_SINE = 0, K_SAW = 1, K_SQUARE = 2, K_PULSE = 3, K_HARM = 4, K_TRIANGLE = 5 };

/* Python float modulo by 1.0 */
static double pymod1(double t) {
    double m = fmod(t, 1.0);
    if (m != 0.0 && m < 0.0) m += 1.0;
    return m;
}

static double wave(int kind, double t, double amp, double bias, double pw, const double* hk, const double* ha, int nh) {
    switch (kind) {
    case K_SINE: return sin(t) * amp + bias;
    case K_SAW: return bias + amp * 2.0 * (t - floor(0.5 + t));
    case K_SQUARE: {
        double tr = trunc(t * 2.0);                 /* int(t*2) % 2 */
        int odd = fmod(fabs(tr), 2.0) == 1.0;
        return (odd ? -amp : amp) + bias;
    }
    case K_PULSE: return ((pymod1(t) < pw) ? amp : -amp) + bias;
    case K_TRIANGLE: return 4.0 * amp * (fabs(pymod1(t + 0.75) - 0.5) - 0.25) + bias;   /* (t+0.75) % 1.0: Python float modulo */
    default: {
        double h = 0.0;
        for (int k = 0; k < nh; ++k) h += sin(t * hk[k]) * ha[k];
        return h * amp + bias;
    }
    }
}

/* non-FM branch of blocks(): t accumulates `inc` from t0; samples [0, n) */
void or_osc_plain(int kind, double t0, double inc, double amp, double bias, double pw,
                  const double* hk, const double* ha, int nh, size_t n, double* out) {
    double t = t0;
    for (size_t i = 0; i < n; ++i) {
        out[i] = wave(kind, t, amp, bias, pw, hk, ha, nh);
        t += inc;
    }
}

/* the accumulated phase after n samples: the `t += inc` of blocks(), n times, from t0 (what or_osc_plain holds when it reaches sample n) */
double or_advance(double t0, double inc, size_t n) {
    volatile double t = t0;
    for (size_t i = 0; i < n; ++i) t = t + inc;
    return t;
}

/* FM branch with a plain Sine LFO (lfo_t0, lfo_inc, lfo_amp, lfo_bias) */
void or_osc_fm_sine(int kind, double frequency, double phase0, double inc, double amp, double bias, double pw,
                    const double* hk, const double* ha, int nh,
                    double lfo_t0, double lfo_inc, double lfo_amp, double lfo_bias, size_t n, double* out) {
    double phase_correction = phase0, freq_previous = frequency, t = 0.0, lt = lfo_t0;
    for (size_t i = 0; i < n; ++i) {
        double fm = sin(lt) * lfo_amp + lfo_bias;
        lt += lfo_inc;
        double freq = frequency * (1.0 + fm);
        phase_correction += (freq_previous - freq) * t;
        freq_previous = freq;
        out[i] = wave(kind, t * freq + phase_correction, amp, bias, pw, hk, ha, nh);
        t += inc;
    }
}

/* Linear: the level is emitted, then incremented while it lies strictly between min and max (synth_oracle.Linear) */
void or_linear(double start, double increment, double minv, double maxv, size_t n, double* out) {
    double value = start;
    for (size_t i = 0; i < n; ++i) {
        out[i] = value;
        if (minv < value && value < maxv) value += increment;
    }
}

/* WhiteNoise as this build defines it (synth_oracle.WhiteNoise): sample n holds draw h = n / cycles,
 * u = (splitmix64(seed + h * 0x9E3779B97F4A7C15) >> 11) * 2^-53, value = (a + (b - a) * u) + bias, a = -amplitude, b = amplitude */
static uint64_t splitmix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void or_white_noise(uint64_t cycles, double amplitude, double bias, uint64_t seed, size_t n, double* out) {
    const double a = -amplitude, b = amplitude;
    for (size_t i = 0; i < n; ++i) {
        const uint64_t h = (uint64_t)i / cycles;
        const double u = (double)(splitmix64(seed + h * 0x9E3779B97F4A7C15ull) >> 11) * 0x1.0p-53;
        out[i] = (a + (b - a) * u) + bias;
    }
}

/* EnvelopeFilter applied in place to n samples of a stream that started at sample 0 */
void or_envelope(double attack, double decay, double sustain, double sustain_level, double release,
                 int samplerate, size_t n, double* x) {
    double time = 0.0, increment = 1.0 / samplerate, amp, amp_change;
    double end_time_decay = attack + decay, end_time_sustain = end_time_decay + sustain,
           end_time_release = end_time_sustain + release;
    size_t i = 0;
    if (attack != 0.0) {
        amp_change = 1.0 / attack * increment;
        amp = 0.0;
        while (time < attack && i < n) { x[i++] *= amp; amp += amp_change; time += increment; }
    }
    if (decay != 0.0) {
        amp = 1.0;
        amp_change = (sustain_level - 1.0) / decay * increment;
        while (time < end_time_decay && i < n) { x[i++] *= amp; amp += amp_change; time += increment; }
    }
    while (time < end_time_sustain && i < n) { x[i++] *= sustain_level; time += increment; }
    if (release != 0.0) {
        amp = sustain_level;
        amp_change = (-sustain_level) / release * increment;
        while (time < end_time_release && i < n) { x[i++] *= amp; amp += amp_change; time += increment; }
        if (amp > 0.0 && i < n && !(time < end_time_release)) x[i++] *= amp;
    }
    while (i < n) x[i++] = 0.0;
}

/* bus[i] = (sum_v gl_v x_v[i], sum_v gr_v x_v[i]) in voice order; voices row-major [nv][n] */
void or_mix_bus(const double* voices, size_t nv, size_t n, const double* gl, const double* gr, double* bus) {
    for (size_t i = 0; i < n; ++i) {
        double l = 0.0, r = 0.0;
        for (size_t v = 0; v < nv; ++v) {
            l += gl[v] * voices[v * n + i];
            r += gr[v] * voices[v * n + i];
        }
        bus[2 * i] = l;
        bus[2 * i + 1] = r;
    }
}

/* int(scale*v) with truncation; returns the index of the first out-of-range value + 1, or 0 */
size_t or_quantise(const double* v, size_t n, double scale, int width, int32_t* out) {
    double lo = -ldexp(1.0, 8 * width - 1), hi = ldexp(1.0, 8 * width - 1) - 1.0;
    for (size_t i = 0; i < n; ++i) {
        double t = trunc(scale * v[i]);
        if (!(t >= lo && t <= hi)) return i + 1;
        out[i] = (int32_t)t;
    }
    return 0;
}

/* ---- audioop ------------------------------------------------------------------------------ */
static int32_t get_sample(const unsigned char* p, int width) {
    switch (width) {
    case 1: return (int8_t)p[0];
    case 2: return (int16_t)(p[0] | (p[1] << 8));
    case 3: { int32_t v = p[0] | (p[1] << 8) | (p[2] << 16); return (v & 0x800000) ? v - 0x1000000 : v; }
    default: return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
    }
}

static void set_sample(unsigned char* p, int width, int32_t v) {
    for (int b = 0; b < width; ++b) p[b] = (unsigned char)((uint32_t)v >> (8 * b));
}

/* audioop.add: saturating */
void or_add(const unsigned char* a, const unsigned char* b, size_t nbytes, int width, unsigned char* out) {
    double maxval = ldexp(1.0, 8 * width - 1) - 1.0, minval = -ldexp(1.0, 8 * width - 1);
    for (size_t i = 0; i + width <= nbytes; i += width) {
        double f = (double)get_sample(a + i, width) + (double)get_sample(b + i, width);
        if (f > maxval) f = maxval; else if (f < minval) f = minval;
        set_sample(out + i, width, (int32_t)floor(f));
    }
}

static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

/* audioop.ratecv(frames, width, nchannels, inrate, outrate, None): returns output frames written.
 * out must hold ((nframes-1)*outrate/inrate + 1) frames. */
size_t or_ratecv(const unsigned char* in, size_t nframes, int width, int nchannels, int inrate, int outrate,
                 unsigned char* out) {
    int g = gcd_i(inrate, outrate);
    inrate /= g;
    outrate /= g;
    int shift = 32 - 8 * width;
    int32_t prev_i[64], cur_i[64];
    if (nchannels > 64) return 0;
    memset(prev_i, 0, sizeof prev_i);
    memset(cur_i, 0, sizeof cur_i);
    int d = -outrate;
    size_t produced = 0;
    const unsigned char* cp = in;
    unsigned char* ncp = out;
    size_t len = nframes;
    for (;;) {
        while (d < 0) {
            if (len == 0) return produced;
            for (int c = 0; c < nchannels; ++c) {
                prev_i[c] = cur_i[c];
                cur_i[c] = (int32_t)((uint32_t)get_sample(cp, width) << shift);
                cp += width;
            }
            len--;
            d += outrate;
        }
        while (d >= 0) {
            for (int c = 0; c < nchannels; ++c) {
                int32_t cur_o = (int32_t)(((double)prev_i[c] * (double)d + (double)cur_i[c] * (double)(outrate - d)) / (double)outrate);
                set_sample(ncp, width, cur_o >> shift);
                ncp += width;
            }
            produced++;
            d -= inrate;
        }
    }
}

/* float32 PCM variant of the same index arithmetic ([SPEC] config 5) */
size_t or_ratecv_f32(const float* in, size_t nframes, int nchannels, int inrate, int outrate, float* out) {
    int g = gcd_i(inrate, outrate);
    inrate /= g;
    outrate /= g;
    double prev[64], cur[64];
    if (nchannels > 64) return 0;
    for (int c = 0; c < nchannels; ++c) prev[c] = cur[c] = 0.0;
    long long d = -outrate;
    size_t produced = 0, pos = 0;
    for (;;) {
        while (d < 0) {
            if (pos == nframes) return produced;
            for (int c = 0; c < nchannels; ++c) { prev[c] = cur[c]; cur[c] = (double)in[pos * nchannels + c]; }
            pos++;
            d += outrate;
        }
        while (d >= 0) {
            for (int c = 0; c < nchannels; ++c)
                out[produced * nchannels + c] = (float)((prev[c] * (double)d + cur[c] * (double)(outrate - d)) / (double)outrate);
            produced++;
            d -= inrate;
        }
    }
}
