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
//   k_prepare_chunks  one wavefront per 64 voices: resolve everything that depends on `start` (phase-table piece,
//                     envelope lines) into a 384-byte launch record per voice, and classify the voices for the render
//                     kernel (lean FastRec list / general index list per chunk).  In a stream the same code runs
//                     inside the previous block's render kernel (its first workgroups), not as a kernel of its own
//   k_prepare         the same records for a single voice (sh_osc_render)
//   k_generate        grid (frame tiles, voice groups): voice-major float32 PCM in HBM (4 B per voice-sample)
//   k_bank_render     fused generate-and-mix: block = W waves on one tile of 64*FPL frames; a wave walks its share of the
//                     lean list (record in SGPRs, one table lookup + rotations + Horner per voice) and of the general
//                     list (voice_block); float64 partial (L, R) per lane, LDS-staged sum across the waves
//   k_bus_combine     folds the voice groups' partial buses in group order, rounds to float32
//   k_mix_bus_f32     HBM-bound mixer over materialised voices (4N+8 B per frame)
#include "common.hpp"
#include "devmath.hpp"
#include <new>
#include <algorithm>
#include <type_traits>
#include <vector>
#include <stdlib.h>

namespace {

constexpr int SEG_MAX = 24;            // segments of a transition launch
// Unequal segments of a materialised row's head (n = 0: none): entry k holds the frames [first[k], first[k] + len[k]) and reads
// record set set[k] (the lists launch skips the segments the host can prove free of general voices)
struct SegTab { uint32_t n; uint32_t first[SEG_MAX]; uint32_t len[SEG_MAX]; uint32_t set[SEG_MAX]; };

struct BankPtrs {
    const sh_voice*   voices;
    const sh_segment* segs;
    const double*     coefs;
    const sh_partial* partials;
    uint32_t*         hint;       // per voice: table piece of the last prepared launch (streaming: same or next piece)
    const double2*    seg_rot;    // per table piece: (cos, sin)(64*dt), computed once on the host at bank creation
    const double2*    lfo_rot;    // per voice: (cos, sin)(64*lfo_d)
    // per launch (sh_bank_render_rows): float64 rows [row][row_stride], launch-relative -- the running LFO sum of an
    // SH_FM_BUFFER voice (fm_row), the pulse width per sample of a Pulse with a pwm_lfo (pwm_row), or the samples themselves of
    // an SH_BUFFER voice (fm_row); -1 = none.  NULL outside such launches.
    const double*     rows;
    size_t            row_stride;
    const int32_t*    fm_row;
    const int32_t*    pwm_row;
    // per launch (a transition launch cut into segments, RENDER_*_SEG): segment s holds the launch's frames
    // [seg_first[s], seg_first[s + 1]) and has a record set of its own (segment_set); nseg = 0 otherwise
    uint32_t          nseg;
    uint32_t          seg_first[SEG_MAX + 1];
    // RENDER_GENERAL_SEG: the FIRST segment holds every voice (several piece ends per tile, sloped envelope) and a workgroup's
    // walk through its group's list is the launch's critical path: there, gen_sub workgroups share a (tile, group) -- list
    // entries dealt round robin -- and write float64 slices [(group * gen_sub + sub) * seg_first[1] + frame] of gen_scratch,
    // which k_seg_combine adds in order into the group's general parts.  Later segments: sub 0 alone, straight into the parts.
    uint32_t          gen_sub;
    double2*          gen_scratch;
};

// Pointers to data that no thread of the running kernel writes are cast to the constant address space:
// a wave-uniform load through such a pointer is a scalar load (s_load_dwordxN into SGPRs) instead of a
// per-lane vector load.
#define SH_CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const T SH_CONST_AS* as_const(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const T SH_CONST_AS*)p;
#pragma clang diagnostic pop
}

// Per-launch voice record, written by k_prepare and read (as one batch of scalar loads) by the render
// kernels.  Everything that depends on `start` is resolved here once per voice instead of once per
// wave: the phase-table piece that holds `start`, and the envelope as lines in the launch-relative frame
// index i = n - start.  The scalar unit (one per CU) is the scarce resource of these kernels, so the
// record is laid out to cost the hot path one batch of loads and almost no scalar arithmetic.
constexpr uint32_t FL_KIND = 0xF, FL_FM_SHIFT = 4, FL_FM = 0x30, FL_DENSE = 0x40, FL_ENV_UNIFORM = 0x80,
                   FL_POLY = 0x100, FL_FOLDED = 0x200, FL_FLIP = 0x400, FL_SILENT = 0x800;
constexpr uint32_t NO_TAIL = 0xFFFFFFFFu;
constexpr int NXP = 12;                             // following phase-table pieces a launch record lists

struct alignas(16) VoiceLaunch {
    // ---- hot part (96 bytes) ----
    double   t_base, dt;          // table value at `start` and the increment of its piece
    uint32_t remain;              // frames from `start` that stay on the piece (saturated)
    uint32_t flags;               // kind | fm_mode << 4 | dense << 6 | env_uniform << 7 | poly << 8
    const double* harm;           // coefficients (dense / poly) or partials (sparse), absolute address
    double   amplitude, bias;
    double   gain_l, gain_r;
    double   g0u, slu;            // FL_ENV_UNIFORM: the whole launch lies on one envelope piece, gain(i) = fma(i, slu, g0u)
    uint32_t harm_cnt;
    uint32_t seg;                 // piece index (slow path resumes the walk here)
    uint32_t tail_i;              // launch-relative index of the post-release extra sample, or NO_TAIL
    uint32_t nx_count;            // following table pieces listed in nx_end
    double   poly[16];            // FL_POLY: the 16 polynomial coefficients, copied here so that the record
                                  // and the coefficients arrive in ONE batch of scalar loads
    double   rot_c, rot_s;        // cos / sin of 64*dt: a lane's second frame is 64 samples after its first, so its
                                  // (sin, cos) is one rotation of the first frame's instead of a second table lookup
    // ---- cold part: launches that cross an envelope boundary, Pulse ----
    uint32_t eb[4];               // launch-relative ends of attack / decay / sustain / release (saturated)
    double   g0[4], slope[4];     // gain(i) = fma(i, slope[p], g0[p]) on piece p; 0 after release
    double   tail_amp;
    double   pulsewidth;
    // the table pieces that follow inside the launch (the first second of a note runs through a dozen binades of the
    // phase sum): piece k (table index seg+1+k) holds the frames [nx_start(k), nx_end[k]), nx_start(0) = remain,
    // nx_start(k) = nx_end[k-1]; there t = fma(i - nx_start(k), dt, t0) with the piece's (t0, dt) from the table
    uint32_t nx_end[NXP];
};
static_assert(sizeof(VoiceLaunch) == 384, "VoiceLaunch layout");

struct alignas(16) VoiceFM {      // only read for FM voices
    double frequency, phase0, f_inc;      // theta = frequency*T + fma(f_inc, L, phase0)
    double lfo_a_rel, lfo_d, lfo_K, lfo_C0, lfo_bias;   // L(i) = K*(C0 - cos(a_rel + i*d)) + bias*(start+i)
    double lfo_rot_c, lfo_rot_s;          // cos / sin of 64*lfo_d: the LFO angle of a lane's next frame is one rotation away
};
static_assert(sizeof(VoiceFM) == 80, "VoiceFM layout");

// The common voice of an additive bank in steady state -- polynomial Harmonics, no FM, amplitude and (constant)
// envelope gain folded into the bus gains, the launch on one table piece or crossing one piece end -- needs only
// this much per launch.
// The render kernel walks these records in a loop of its own, with none of the general code's flag tests.
struct alignas(64) FastRec {
    // ---- the first 192 bytes are all a launch without a piece crossing needs: three s_load_dwordx16 ----
    double t_base, dt;            // frames i < remain: t(i) = fma(i, dt, t_base)
    double gain_l, gain_r;        // amplitude * envelope gain * bus gain
    double rot_c, rot_s;          // cos / sin of 64*dt
    double poly[16];              // sum_k a_k sin(k t) = sin(t) * P(cos t)
    uint32_t remain;              // 0xFFFFFFFF: no crossing in this launch
    uint32_t kind;                // LEAN_HARM: the fields as described; LEAN_FM: Sine carrier with a closed-form Sine LFO --
                                  // t is the accumulated TIME table, poly[0..10] = frequency, phase0, f_inc, lfo_a_rel, lfo_d,
                                  // lfo_K, lfo_C0, lfo_bias, lfo_rot_c, lfo_rot_s, (double)start (see VoiceFM)
    double off_b;                 // (double)remain
    // ---- read only when the launch crosses ONE end of a phase-table piece (a binade of the running sum), at `remain` ----
    double t0_b, dt_b;            // frames i >= remain: t(i) = fma(i - remain, dt_b, t0_b)
    double rot_c_b, rot_s_b;      // cos / sin of 64*dt_b
    // ---- k_generate_lists ----
    double amplitude, g0u;        // unfolded: the voice's own sample is ((x * amplitude) + 0) * g0u.  In the record sets of a
                                  // segmented launch (prepare_chunk<true>): the slopes of gain_l / gain_r per frame instead
    uint32_t vi;                  // the voice
    uint32_t pad1;
    double pad2;
};
static_assert(offsetof(FastRec, t0_b) == 192, "FastRec: common part is 192 bytes");
static_assert(sizeof(FastRec) == 256, "FastRec layout");
constexpr uint32_t LEAN_HARM = 0, LEAN_FM = 1,                 // ... and the plain waveforms without FM (t in turns, Sine: radians);
                   LEAN_SINE = 2, LEAN_SAW = 3, LEAN_SQUARE = 4, LEAN_TRIANGLE = 5, LEAN_PULSE = 6;    // Pulse: poly[0] = pulsewidth

// The accumulated t of a lane's frame j (64 samples apart) on the launch's first or second phase-table piece -- by VALUE
// members (a lambda capturing by reference kept its closure, and with it every captured scalar, in scratch memory).
struct LaneTheta {
    double   di0;                 // the lane's first frame
    double   t_base, dt, off;     // the tile's piece when it lies on one (uniform): t = fma(i - off, dt, t_base)
    double   ta, da, tb, db, ob;  // both pieces, for the tile that straddles the end of the first at `remain`
    uint32_t i0, remain;
    bool     straddle;
    __device__ __forceinline__ double operator()(int j) const {
        const double dd = di0 + (double)(j * 64);
        if (!straddle) return fma(dd - off, dt, t_base);
        return i0 + (uint32_t)j * 64u < remain ? fma(dd, da, ta) : fma(dd - ob, db, tb);
    }
};

// One set of per-launch data (double-buffered in the bank).  Voices are classified per chunk of 64 consecutive
// voices: the fast voices of chunk c get a FastRec, compacted at fast[64c ..], the others are listed by index in
// gen_idx[64c ..] (both in ascending voice order: the summation order is fixed), silent voices appear in neither.
struct LaunchSet {
    VoiceLaunch* launch;
    VoiceFM*     fm;
    FastRec*     fast;
    uint32_t*    gen_idx;
    uint32_t*    counts;          // per chunk: [4c] = lean voices, [4c+1] = general voices, [4c+2] = silent voices
};

struct PrepInfo {                 // what prepare_voice found, for the classification
    bool   fast, silent;
    double t_base, dt, gain_l, gain_r, rot_c, rot_s;
    double t0_b, dt_b, rot_c_b, rot_s_b;
    uint32_t remain;
    uint32_t kind;                // LEAN_HARM / LEAN_FM
    double   amplitude, g0u, pulsewidth;
    double   fmv[11];             // LEAN_FM: the values that go to FastRec::poly[0..10]
    double   slope_l, slope_r;    // a record set of sloped lean records (SLOPED): gain(i) = gain + i * slope on the envelope's line
    const double* harm;
};

// SLOPED (the record sets of a segmented transition launch): a polynomial-Harmonics voice on ONE envelope line of any slope is
// lean too -- its record carries the line folded into the bus gains (gain + i * slope) -- not only one on a constant gain.
template <bool SLOPED = false>
__device__ __forceinline__ void prepare_voice(const BankPtrs& B, uint32_t first, uint32_t vi, uint64_t start, uint32_t nframes,
                                              VoiceLaunch* __restrict__ out, VoiceFM* __restrict__ out_fm, PrepInfo& info) {
    // Fields are stored straight to the record (no local struct: a 368-byte private array would give
    // every wave of the render kernel a scratch allocation).
    const sh_voice& v = B.voices[first + vi];
    VoiceLaunch* __restrict__ o = out + vi;
    const bool fm = v.fm_mode != SH_FM_NONE;
    const uint32_t off = fm ? v.time_seg_offset : v.seg_offset;
    const uint32_t cnt = fm ? v.time_seg_count : v.seg_count;
    const sh_segment* tab = B.segs + off;
    uint32_t lo = B.hint[first + vi];
    if (lo < cnt && tab[lo].n0 <= start && lo + 2 < cnt && start < tab[lo + 2].n0) {
        if (start >= tab[lo + 1].n0) ++lo;                     // streaming: still on the piece, or on the next one
    } else {
        lo = 0;
        uint32_t hi = cnt - 1;
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (tab[mid].n0 <= start) lo = mid; else hi = mid - 1;
        }
    }
    B.hint[first + vi] = lo;
    const double dt = tab[lo].dt;
    const double t_base = fma((double)(start - tab[lo].n0), dt, tab[lo].t0);
    const uint64_t rem = (lo + 1 < cnt) ? (tab[lo + 1].n0 - start) : 0xFFFFFFFFull;
    o->t_base = t_base;
    o->dt = dt;
    o->remain = rem > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rem;
    o->seg = lo;
    {
        uint32_t count = 0;
        uint64_t piece_start = rem;                           // launch-relative first frame of the next piece
#pragma unroll 1
        for (int k = 0; k < NXP; ++k) {
            const uint32_t pi = lo + 1 + k;
            uint32_t end = 0xFFFFFFFFu;
            if (count == (uint32_t)k && pi < cnt && piece_start < (uint64_t)nframes) {      // still inside the launch
                const uint64_t e = (pi + 1 < cnt) ? (tab[pi + 1].n0 - start) : 0xFFFFFFFFull;
                end = e > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)e;
                piece_start = e;
                count = (uint32_t)k + 1;
            }
            o->nx_end[k] = end;
        }
        o->nx_count = count;
    }
    uint32_t flags = (uint32_t)v.kind | ((uint32_t)v.fm_mode << FL_FM_SHIFT) |
                     (v.harm_dense == 1 ? FL_DENSE : 0u) | (v.harm_dense == 2 ? FL_POLY : 0u) | (v.flip ? FL_FLIP : 0u);
    const double* harm = v.harm_dense ? (B.coefs + v.harm_offset) : reinterpret_cast<const double*>(B.partials + v.harm_offset);
    o->harm = harm;
    o->harm_cnt = v.harm_count;
    o->amplitude = v.amplitude;
    o->bias = v.bias;
    o->pulsewidth = v.pulsewidth;
    double gain_l = (double)v.gain_l, gain_r = (double)v.gain_r;
    // envelope as four lines in the launch-relative frame index
    uint32_t eb0, eb1, eb2, eb3, tail_i = NO_TAIL;
    double g00, g01, g02, g03, s0, s1, s2, s3, tail_amp = 0.0;
    const sh_envelope& e = v.env;
    if (e.enabled) {
        const uint64_t d0 = e.n_attack_end > start ? e.n_attack_end - start : 0;
        const uint64_t d1 = e.n_decay_end > start ? e.n_decay_end - start : 0;
        const uint64_t d2 = e.n_sustain_end > start ? e.n_sustain_end - start : 0;
        const uint64_t d3 = e.n_release_end > start ? e.n_release_end - start : 0;
        eb0 = d0 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d0;
        eb1 = d1 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d1;
        eb2 = d2 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d2;
        eb3 = d3 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d3;
        const double st = (double)start;
        g00 = st * e.attack_slope;                                      // gain(n) = n * attack_slope
        s0 = e.attack_slope;
        g01 = fma(st - (double)e.n_attack_end, e.decay_slope, 1.0);     // 1 + (n - nA) * decay_slope
        s1 = e.decay_slope;
        g02 = e.sustain_level;
        s2 = 0.0;
        g03 = fma(st - (double)e.n_sustain_end, e.release_slope, e.sustain_level);
        s3 = e.release_slope;
        if (e.has_tail && e.n_release_end >= start && e.n_release_end - start < 0xFFFFFFFFull) {
            tail_i = (uint32_t)(e.n_release_end - start);
            tail_amp = e.tail_amp;
        }
    } else {                       // no envelope: an endless sustain piece of gain 1
        eb0 = 0; eb1 = 0; eb2 = 0xFFFFFFFFu; eb3 = 0xFFFFFFFFu;
        g00 = g01 = g02 = g03 = 1.0;
        s0 = s1 = s2 = s3 = 0.0;
    }
    o->eb[0] = eb0; o->eb[1] = eb1; o->eb[2] = eb2; o->eb[3] = eb3;
    o->g0[0] = g00; o->g0[1] = g01; o->g0[2] = g02; o->g0[3] = g03;
    o->slope[0] = s0; o->slope[1] = s1; o->slope[2] = s2; o->slope[3] = s3;
    o->tail_i = tail_i;
    o->tail_amp = tail_amp;
    // does the whole launch [0, nframes) sit on one envelope piece?
    double g0u = 0.0, slu = 0.0;
    {
        const uint32_t last = nframes ? nframes - 1 : 0;
        const bool tail_here = tail_i != NO_TAIL && tail_i <= last;
        if (eb0 > 0) { if (last < eb0) { flags |= FL_ENV_UNIFORM; g0u = g00; slu = s0; } }
        else if (eb1 > 0) { if (last < eb1) { flags |= FL_ENV_UNIFORM; g0u = g01; slu = s1; } }
        else if (eb2 > 0) { if (last < eb2) { flags |= FL_ENV_UNIFORM; g0u = g02; slu = s2; } }
        else if (eb3 > 0) { if (last < eb3) { flags |= FL_ENV_UNIFORM; g0u = g03; slu = s3; } }
        else if (!tail_here) flags |= FL_ENV_UNIFORM | FL_SILENT;      // released before this launch: silent throughout
    }
    o->g0u = g0u;
    o->slu = slu;
    const double2 rot = B.seg_rot[off + lo];
    const double rc = rot.x, rs = rot.y;
    o->rot_c = rc;
    o->rot_s = rs;
    if (flags & FL_POLY) {
#pragma unroll
        for (int u = 0; u < 16; ++u) o->poly[u] = harm[u];
        // bias == 0 and a constant envelope gain: fold amplitude and envelope into the bus gains, so the
        // inner loop is h = P(c)*s; L += GL*h; R += GR*h.  (Differs from the unfolded order by float64
        // rounding only, ~1e-16 relative.)  Bank kernels only: k_generate needs the voice sample itself.
        if (v.bias == 0.0 && !v.flip && (flags & FL_ENV_UNIFORM) && slu == 0.0) {
            flags |= FL_FOLDED;
            gain_l = (v.amplitude * g0u) * gain_l;
            gain_r = (v.amplitude * g0u) * gain_r;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) o->poly[u] = 0.0;
    }
    o->gain_l = gain_l;
    o->gain_r = gain_r;
    o->flags = flags;
    // Other voices that can take the lean loop: a Sine carrier with a closed-form Sine LFO, and the plain waveforms without
    // FM.  Their record folds amplitude and envelope into the gains like FL_FOLDED does (the general code is not told: its
    // paths apply the amplitude themselves).
    const bool lean_env = v.bias == 0.0 && !v.flip && (flags & FL_ENV_UNIFORM) && slu == 0.0 && !(flags & FL_SILENT);
    const bool lean_fm = lean_env && v.kind == SH_SINE && v.fm_mode == SH_FM_SINE;
    const bool lean_plain = lean_env && v.fm_mode == SH_FM_NONE &&
                            (v.kind == SH_SINE || v.kind == SH_SAWTOOTH || v.kind == SH_SQUARE || v.kind == SH_TRIANGLE || v.kind == SH_PULSE);
    info.kind = lean_fm ? LEAN_FM
              : !lean_plain ? LEAN_HARM
              : v.kind == SH_SINE ? LEAN_SINE : v.kind == SH_SAWTOOTH ? LEAN_SAW : v.kind == SH_SQUARE ? LEAN_SQUARE
              : v.kind == SH_TRIANGLE ? LEAN_TRIANGLE : LEAN_PULSE;
    info.amplitude = v.amplitude;
    info.g0u = g0u;
    info.pulsewidth = v.pulsewidth;
    if (lean_fm || lean_plain) {
        gain_l = (v.amplitude * g0u) * gain_l;
        gain_r = (v.amplitude * g0u) * gain_r;
    }
    info.silent = (flags & FL_SILENT) != 0;
    // lean: the launch lies on the current table piece, or on it and the next one
    const bool one_piece = rem >= (uint64_t)nframes;
    const bool two_pieces = !one_piece && lo + 1 < cnt && ((lo + 2 < cnt) ? (tab[lo + 2].n0 - start >= (uint64_t)nframes) : true);
    bool lean_sloped = false;
    info.slope_l = 0.0;
    info.slope_r = 0.0;
    if (SLOPED && (flags & (FL_POLY | FL_FOLDED | FL_FM | FL_SILENT | FL_ENV_UNIFORM)) == (FL_POLY | FL_ENV_UNIFORM) && v.bias == 0.0 && !v.flip) {
        lean_sloped = true;                               // (not FL_FOLDED: slu != 0)
        gain_l = (v.amplitude * g0u) * (double)v.gain_l;
        gain_r = (v.amplitude * g0u) * (double)v.gain_r;
        info.slope_l = (v.amplitude * slu) * (double)v.gain_l;
        info.slope_r = (v.amplitude * slu) * (double)v.gain_r;
    }
    info.fast = ((flags & (FL_POLY | FL_FOLDED | FL_FM | FL_SILENT)) == (FL_POLY | FL_FOLDED) || lean_fm || lean_plain || lean_sloped) && (one_piece || two_pieces);
    info.remain = one_piece ? 0xFFFFFFFFu : (uint32_t)rem;
    info.t0_b = one_piece ? t_base : tab[lo + 1].t0;
    info.dt_b = one_piece ? dt : tab[lo + 1].dt;
    {
        const double2 rb = B.seg_rot[off + (one_piece ? lo : lo + 1)];
        info.rot_c_b = rb.x;
        info.rot_s_b = rb.y;
    }
    info.t_base = t_base;
    info.dt = dt;
    info.gain_l = gain_l;
    info.gain_r = gain_r;
    info.rot_c = rc;
    info.rot_s = rs;
    info.harm = harm;
    if (fm) {
        VoiceFM* __restrict__ f = out_fm + vi;
        f->frequency = v.frequency;
        f->phase0 = v.fm_phase0;
        f->f_inc = v.frequency * v.fm_inc;
        f->lfo_a_rel = fma((double)start - 0.5, v.lfo_d, v.lfo_a);     // arg(i) = a + (start + i - 0.5) * d
        f->lfo_d = v.lfo_d;
        f->lfo_K = v.lfo_K;
        f->lfo_C0 = v.lfo_C0;
        f->lfo_bias = v.lfo_bias;
        const double2 lrot = B.lfo_rot[first + vi];
        f->lfo_rot_c = lrot.x;
        f->lfo_rot_s = lrot.y;
        info.fmv[0] = v.frequency; info.fmv[1] = v.fm_phase0; info.fmv[2] = v.frequency * v.fm_inc;
        info.fmv[3] = fma((double)start - 0.5, v.lfo_d, v.lfo_a);
        info.fmv[4] = v.lfo_d; info.fmv[5] = v.lfo_K; info.fmv[6] = v.lfo_C0; info.fmv[7] = v.lfo_bias;
        info.fmv[8] = lrot.x; info.fmv[9] = lrot.y; info.fmv[10] = (double)start;
    }
}

__global__ void k_prepare(BankPtrs B, uint32_t first, uint32_t nvoices, uint64_t start, uint32_t nframes,
                          VoiceLaunch* __restrict__ out, VoiceFM* __restrict__ out_fm) {
    uint32_t vi = blockIdx.x * blockDim.x + threadIdx.x;
    PrepInfo info;
    if (vi < nvoices) prepare_voice(B, first, vi, start, nframes, out, out_fm, info);
}

// One wavefront resolves the launch records of one chunk of 64 consecutive voices (lane = voice) and classifies them
// (see LaunchSet); positions in the chunk's compacted lists come from wave ballots -- no inter-thread memory traffic,
// no barrier, chunks are independent of each other.
template <bool SLOPED = false>
__device__ __forceinline__ void prepare_chunk(const BankPtrs& B, const LaunchSet& S, uint32_t c, uint32_t nvoices,
                                              uint64_t start, uint32_t nframes) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t vi = c * 64 + lane;
    PrepInfo info;
    info.fast = false;
    info.silent = true;
    if (vi < nvoices) prepare_voice<SLOPED>(B, 0u, vi, start, nframes, S.launch, S.fm, info);
    const bool is_fast = vi < nvoices && info.fast;
    const bool is_gen = vi < nvoices && !info.fast && !info.silent;
    const uint64_t mf = __ballot(is_fast), mg = __ballot(is_gen);
    const uint64_t below = (1ull << lane) - 1ull;
    if (is_fast) {
        FastRec* __restrict__ f = S.fast + c * 64 + (uint32_t)__popcll(mf & below);
        f->t_base = info.t_base; f->dt = info.dt;
        f->gain_l = info.gain_l; f->gain_r = info.gain_r;
        f->rot_c = info.rot_c; f->rot_s = info.rot_s;
        if (info.kind == LEAN_FM) {
#pragma unroll
            for (int u = 0; u < 16; ++u) f->poly[u] = u < 11 ? info.fmv[u] : 0.0;
        } else if (info.kind != LEAN_HARM) {
#pragma unroll
            for (int u = 0; u < 16; ++u) f->poly[u] = u == 0 ? info.pulsewidth : 0.0;
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) f->poly[u] = info.harm[u];
        }
        f->t0_b = info.t0_b; f->dt_b = info.dt_b;
        f->rot_c_b = info.rot_c_b; f->rot_s_b = info.rot_s_b;
        f->off_b = (double)info.remain;
        f->remain = info.remain;
        f->kind = info.kind;
        f->amplitude = SLOPED ? info.slope_l : info.amplitude;        // (SLOPED sets are read by the RENDER_LEAN_*_SEG kernels only)
        f->g0u = SLOPED ? info.slope_r : info.g0u;
        f->vi = vi;
        f->pad1 = 0;
        f->pad2 = 0.0;
    }
    if (is_gen) S.gen_idx[c * 64 + (uint32_t)__popcll(mg & below)] = vi;
    // silent voices: listed from the END of the chunk's index slots (the render kernel never looks there; k_generate_lists
    // zero-fills their rows)
    const bool is_silent = vi < nvoices && info.silent;
    const uint64_t ms = __ballot(is_silent);
    if (is_silent) S.gen_idx[c * 64 + 63 - (uint32_t)__popcll(ms & below)] = vi;
    if (lane == 0) {
        S.counts[4 * c] = (uint32_t)__popcll(mf);
        S.counts[4 * c + 1] = (uint32_t)__popcll(mg);
        S.counts[4 * c + 2] = (uint32_t)__popcll(ms);
        S.counts[4 * c + 3] = 0;
    }
}

__global__ __launch_bounds__(64) void k_prepare_chunks(BankPtrs B, LaunchSet S, uint32_t nvoices, uint64_t start, uint32_t nframes) {
    prepare_chunk(B, S, blockIdx.x, nvoices, start, nframes);
}

// Segment s of a long materialisation: the frames [s * seg_frames, ...) of the launch get a record set of their own, laid out
// one after the other in arrays of nseg * nvoices records (counts: nseg * 4 * nchunks).
__device__ __forceinline__ LaunchSet segment_set(const LaunchSet& base, uint32_t s, uint32_t nvoices) {
    const uint32_t nchunks = (nvoices + 63) / 64;
    LaunchSet r;
    r.launch = base.launch + (size_t)s * nvoices;
    r.fm = base.fm + (size_t)s * nvoices;
    r.fast = base.fast + (size_t)s * nvoices;
    r.gen_idx = base.gen_idx + (size_t)s * nvoices;
    r.counts = base.counts + (size_t)s * 4 * nchunks;
    return r;
}

// grid = (chunks, segments): the launch records of every segment of a long materialisation in one launch.  (All segments
// read and write the per-voice piece hint; whatever value a lane finds there is checked before use.)
__global__ __launch_bounds__(64) void k_prepare_segments(BankPtrs B, LaunchSet base, uint32_t nvoices, uint64_t start, uint32_t nframes,
                                                         uint32_t seg_frames) {
    const uint32_t s = blockIdx.y;
    const uint32_t first = s * seg_frames;
    const uint32_t n = nframes - first < seg_frames ? nframes - first : seg_frames;
    prepare_chunk(B, segment_set(base, s, nvoices), blockIdx.x, nvoices, start + first, n);
}

// The same for the unequal segments of a transition launch (B.seg_first): grid = (chunks, segments).
// (SLOPED: the records of a render launch; the materialisation kernels need the unfolded amplitude and gain instead)
template <bool SLOPED>
__global__ __launch_bounds__(64) void k_prepare_segments_var(BankPtrs B, LaunchSet base, uint32_t nvoices, uint64_t start) {
    const uint32_t s = blockIdx.y;
    const uint32_t first = B.seg_first[s], n = B.seg_first[s + 1] - first;
    prepare_chunk<SLOPED>(B, segment_set(base, s, nvoices), blockIdx.x, nvoices, start + first, n);
}

struct VoiceRegs {                // the hot part of the launch record as plain scalars (SGPRs)
    double   t_base, dt;
    uint32_t remain, flags;
    const double SH_CONST_AS* harm;
    double   amplitude, bias, gain_l, gain_r, g0u, slu;
    uint32_t harm_cnt;
    double   poly[16];            // loaded unconditionally with the rest: one batch, one wait per voice
    double   rot_c, rot_s;
    const VoiceLaunch SH_CONST_AS* rec;   // cold fields are read through this where needed
};

__device__ __forceinline__ VoiceRegs load_record(const VoiceLaunch SH_CONST_AS* p) {
    VoiceRegs r;
    r.t_base = p->t_base; r.dt = p->dt;
    r.remain = p->remain; r.flags = p->flags;
    r.harm = as_const(p->harm);
    r.amplitude = p->amplitude; r.bias = p->bias;
    r.gain_l = p->gain_l; r.gain_r = p->gain_r;
    r.g0u = p->g0u; r.slu = p->slu;
    r.harm_cnt = p->harm_cnt;
#pragma unroll
    for (int u = 0; u < 16; ++u) r.poly[u] = p->poly[u];
    r.rot_c = p->rot_c; r.rot_s = p->rot_s;
    r.rec = p;
    return r;
}

typedef const shm::sc_pair* TrigTab;     // LDS

// FPL samples per lane of one voice (frames i[j], launch-relative), float64.  Every argument except
// i/di is wave-uniform, so the branches on kind / fm_mode / envelope do not diverge.
//   tile_last: last frame index any lane of this wave touches (uniform)
template <int FPL, bool BANK>
__device__ __forceinline__ void voice_block(const VoiceRegs& r, const VoiceFM* __restrict__ fmrec,
                                            const BankPtrs& B, const sh_voice* __restrict__ vfull,
                                            uint64_t start, uint32_t tile_last,
                                            const uint32_t (&i)[FPL], const double (&di)[FPL],
                                            const double* __restrict__ fm_cumsum, const double* __restrict__ pwm,
                                            TrigTab trig, double (&x)[FPL]) {
    double th[FPL];
    bool linear = false;          // th[j] = th[0] + 64*j*dt exactly (all frames of the tile on one table piece, no FM)
    double rot_c = r.rot_c, rot_s = r.rot_s;      // (cos, sin)(64*dt) of that piece
    // ---- phase: the reference's accumulated t at each frame ----
    if (tile_last < r.remain) {
        linear = (r.flags & FL_FM) == 0;
#pragma unroll
        for (int j = 0; j < FPL; ++j) th[j] = fma(di[j], r.dt, r.t_base);      // exact (stays on the piece)
    } else {
        // the launch crosses a binade of the running sum.  Tiles wholly on the NEXT piece use it directly;
        // anything else looks its piece up in the voice's table: once per wave for the tile's first frame
        // (scalar binary search), then per lane only for the lanes past that piece's end.
        const VoiceLaunch SH_CONST_AS* q = r.rec;
        const uint32_t tile_first = tile_last & ~(uint32_t)(64 * FPL - 1);
        // k = index of the following piece that holds the tile's first frame (uniform): the number of listed piece
        // ends at or before it
        uint32_t k = 0;
#pragma unroll
        for (int u = 0; u < NXP; ++u) k += tile_first >= q->nx_end[u];
        const uint32_t nx_count = q->nx_count;
        const bool fm_t = (r.flags & FL_FM) != 0;
        uint32_t k_end = 0, k_start = 0;
        if (k < nx_count) {
            k_end = q->nx_end[k];
            k_start = k ? q->nx_end[k - 1] : r.remain;
        }
        if (tile_first >= r.remain && k < nx_count && tile_last < k_end) {
            const sh_segment SH_CONST_AS* pc = as_const(B.segs) + (fm_t ? vfull->time_seg_offset : vfull->seg_offset) + q->seg + 1 + k;
            const double tb = pc->t0, dk = pc->dt, off = (double)k_start;
#pragma unroll
            for (int j = 0; j < FPL; ++j) th[j] = fma(di[j] - off, dk, tb);
            if ((r.flags & FL_FM) == 0) {                     // still one piece per tile: the rotation shortcut applies
                const double2 SH_CONST_AS* rt = as_const(B.seg_rot) + vfull->seg_offset + q->seg + 1 + k;
                rot_c = rt->x;
                rot_s = rt->y;
                linear = true;
            }
        } else {
            const bool fm = (r.flags & FL_FM) != 0;
            const sh_segment SH_CONST_AS* tab = as_const(B.segs) + (fm ? vfull->time_seg_offset : vfull->seg_offset);
            const uint32_t cnt = fm ? vfull->time_seg_count : vfull->seg_count;
            const uint64_t n_first = start + tile_first;
            uint32_t lo = 0, hi = cnt - 1;                      // uniform: last piece with n0 <= n_first
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (tab[mid].n0 <= n_first) lo = mid; else hi = mid - 1;
            }
            const uint64_t p_n0 = tab[lo].n0;
            const double p_t0 = tab[lo].t0, p_dt = tab[lo].dt;
            const uint64_t p_end = (lo + 1 < cnt) ? tab[lo + 1].n0 : ~0ull;
#pragma unroll
            for (int j = 0; j < FPL; ++j) {
                const uint64_t n = start + i[j];
                if (n < p_end) {
                    th[j] = fma((double)(n - p_n0), p_dt, p_t0);
                } else {                                        // lanes beyond the tile's first piece
                    uint32_t sgi = lo;
                    while (sgi + 1 < cnt && tab[sgi + 1].n0 <= n) ++sgi;
                    th[j] = fma((double)(n - tab[sgi].n0), tab[sgi].dt, tab[sgi].t0);
                }
            }
        }
    }
    if (r.flags & FL_FM) {
        const uint32_t fm_mode = (r.flags & FL_FM) >> FL_FM_SHIFT;
        const VoiceFM SH_CONST_AS* fp = as_const(fmrec);
        struct { double frequency, phase0, f_inc, lfo_a_rel, lfo_d, lfo_K, lfo_C0, lfo_bias; } f;
        f.frequency = fp->frequency; f.phase0 = fp->phase0; f.f_inc = fp->f_inc; f.lfo_a_rel = fp->lfo_a_rel;
        f.lfo_d = fp->lfo_d; f.lfo_K = fp->lfo_K; f.lfo_C0 = fp->lfo_C0; f.lfo_bias = fp->lfo_bias;
        if (fm_mode == SH_FM_SINE) {
            // LFO angle a_rel + i*d is exactly linear in i: table lookup for the lane's first frame, rotation
            // by 64*d for the following ones
            double ls, lc;
            shm::sincos_tab(fma(di[0], f.lfo_d, f.lfo_a_rel), trig, ls, lc);
            const double rc = fp->lfo_rot_c, rs = fp->lfo_rot_s;
#pragma unroll
            for (int j = 0; j < FPL; ++j) {
                if (j > 0) {
                    const double ns = fma(ls, rc, lc * rs), nc = fma(lc, rc, -(ls * rs));
                    ls = ns;
                    lc = nc;
                }
                const double Ln = fma(f.lfo_K, f.lfo_C0 - lc, f.lfo_bias * ((double)start + di[j]));
                th[j] = f.frequency * th[j] + fma(f.f_inc, Ln, f.phase0);      // t*freq + phase_correction
            }
        } else {
#pragma unroll
            for (int j = 0; j < FPL; ++j) th[j] = f.frequency * th[j] + fma(f.f_inc, fm_cumsum[i[j]], f.phase0);
        }
    }
    // ---- waveform ----
    if (r.flags & FL_POLY) {
        // Harmonics with k <= 16: sum_k a_k sin(k t) = sin(t) * P(cos t), P of degree 15 (coefficients
        // converted on the host in exact rational arithmetic), Horner: 15 FMAs instead of 32 Clenshaw ops
        double sn[FPL], cs[FPL], pv[FPL];
        if (FPL > 1 && linear) {
            shm::sincos_tab(th[0], trig, sn[0], cs[0]);
#pragma unroll
            for (int j = 1; j < FPL; ++j) {               // rotate by 64*dt: 4 float64 ops instead of 17
                sn[j] = fma(sn[j - 1], rot_c, cs[j - 1] * rot_s);
                cs[j] = fma(cs[j - 1], rot_c, -(sn[j - 1] * rot_s));
            }
        } else {
            shm::sincos_tab_n<FPL>(th, trig, sn, cs);
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) pv[j] = fma(r.poly[0], cs[j], r.poly[1]);
#pragma unroll
        for (int u = 2; u < 16; ++u) {                // Horner, the FPL chains interleaved
#pragma unroll
            for (int j = 0; j < FPL; ++j) pv[j] = fma(pv[j], cs[j], r.poly[u]);
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) x[j] = pv[j] * sn[j];
        if (BANK && (r.flags & FL_FOLDED)) return;          // amplitude and envelope live in the bus gains
#pragma unroll
        for (int j = 0; j < FPL; ++j) x[j] = x[j] * r.amplitude + r.bias;
    } else {
        switch (r.flags & FL_KIND) {
        case SH_SINE:
#pragma unroll
            for (int j = 0; j < FPL; ++j) {
                double sn, cs;
                shm::sincos_tab(th[j], trig, sn, cs);
                x[j] = sn * r.amplitude + r.bias;
            }
            break;
        case SH_SAWTOOTH:
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = shm::saw_value(th[j], r.amplitude * 2.0, r.bias);
            break;
        case SH_SQUARE:
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = shm::square_value(th[j], r.amplitude, r.bias);
            break;
        case SH_TRIANGLE:
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = shm::triangle_value(th[j], 4.0 * r.amplitude, r.bias);
            break;
        case SH_LINEAR:
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = th[j];
            break;
        case SH_BUFFER:                            // the samples were rendered elsewhere (a filter graph): row[i]
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = fm_cumsum ? fm_cumsum[i[j]] : 0.0;
            break;
        case SH_NOISE: {
            const uint64_t seed = vfull->noise_seed;
            const uint32_t hold = vfull->noise_hold;
            const double a2 = r.amplitude * 2.0;
#pragma unroll
            for (int j = 0; j < FPL; ++j) {
                const uint64_t n = start + i[j];
                const uint64_t h = n <= 0xFFFFFFFFull ? (uint64_t)((uint32_t)n / hold) : n / hold;
                uint64_t z = seed + h * 0x9E3779B97F4A7C15ull;                  // splitmix64 of the held-value counter
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z ^= z >> 31;
                const double u = (double)(z >> 11) * 0x1.0p-53;
                x[j] = (-r.amplitude + a2 * u) + r.bias;
            }
        } break;
        case SH_PULSE: {
            const double pw = r.rec->pulsewidth;
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = shm::pulse_value(th[j], pwm ? pwm[i[j]] : pw, r.amplitude, r.bias);
        } break;
        default: {   // SH_HARMONICS, general forms
            double h[FPL];
            if (r.flags & FL_DENSE) {
                // Clenshaw: b_k = a_k + 2cos(t) b_{k+1} - b_{k+2}; sum_k a_k sin(k t) = b_1 sin(t)
                double sn[FPL], c2[FPL], b1[FPL], b2[FPL];
#pragma unroll
                for (int j = 0; j < FPL; ++j) {
                    double c;
                    shm::sincos_tab(th[j], trig, sn[j], c);
                    c2[j] = c + c;
                    b1[j] = 0.0;
                    b2[j] = 0.0;
                }
                for (uint32_t k = 0; k < r.harm_cnt; k += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double ak = r.harm[k + u];
#pragma unroll
                        for (int j = 0; j < FPL; ++j) {
                            double bn = fma(c2[j], b1[j], ak - b2[j]);
                            b2[j] = b1[j];
                            b1[j] = bn;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < FPL; ++j) h[j] = b1[j] * sn[j];
            } else {
                const sh_partial SH_CONST_AS* p = reinterpret_cast<const sh_partial SH_CONST_AS*>(r.harm);
#pragma unroll
                for (int j = 0; j < FPL; ++j) h[j] = 0.0;
                for (uint32_t k = 0; k < r.harm_cnt; ++k) {
                    const double pk = p[k].k, pa = p[k].amp;
#pragma unroll
                    for (int j = 0; j < FPL; ++j) {
                        double sn, cs;
                        shm::sincos_tab(th[j] * pk, trig, sn, cs);
                        h[j] += sn * pa;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = h[j] * r.amplitude + r.bias;
        } break;
        }
    }
    if (r.flags & FL_FLIP) {                    // SawtoothH: "we have to flip the wave"
#pragma unroll
        for (int j = 0; j < FPL; ++j) x[j] = r.bias * 2.0 - x[j];
    }
    // ---- envelope ----
    if (r.flags & FL_ENV_UNIFORM) {            // the whole launch lies on one piece (the common case)
#pragma unroll
        for (int j = 0; j < FPL; ++j) x[j] = x[j] * fma(di[j], r.slu, r.g0u);
    } else {                                   // this launch crosses attack/decay/sustain/release ends
        const VoiceLaunch SH_CONST_AS* q = r.rec;
        const uint32_t e0 = q->eb[0], e1 = q->eb[1], e2 = q->eb[2], e3 = q->eb[3], ti = q->tail_i;
        const uint32_t tile_first = tile_last & ~(uint32_t)(64 * FPL - 1);
        const uint32_t p = (tile_first >= e0) + (tile_first >= e1) + (tile_first >= e2) + (tile_first >= e3);
        const uint32_t pend = p < 4 ? q->eb[p & 3] : 0xFFFFFFFFu;
        const bool tail_here = ti != NO_TAIL && ti >= tile_first && ti <= tile_last;
        if ((p == 4 || tile_last < pend) && !tail_here) {       // the tile lies on one piece: uniform line
            const double g0 = p < 4 ? q->g0[p & 3] : 0.0, sl = p < 4 ? q->slope[p & 3] : 0.0;
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = x[j] * fma(di[j], sl, g0);
        } else {                                                // the few tiles that straddle a piece end
            const double g00 = q->g0[0], g01 = q->g0[1], g02 = q->g0[2], g03 = q->g0[3];
            const double s0 = q->slope[0], s1 = q->slope[1], s2 = q->slope[2], s3 = q->slope[3];
            const double ta = q->tail_amp;
#pragma unroll
            for (int j = 0; j < FPL; ++j) {
                const uint32_t ii = i[j];
                const double g0 = ii < e0 ? g00 : ii < e1 ? g01 : ii < e2 ? g02 : g03;
                const double sl = ii < e0 ? s0 : ii < e1 ? s1 : ii < e2 ? s2 : s3;
                double g = fma(di[j], sl, g0);
                if (ii >= e3) g = (ii == ti) ? ta : 0.0;
                x[j] = x[j] * g;
            }
        }
    }
}

// voice-major materialisation: out[v*stride + i].  grid = (groups of 4 tiles, voice groups); block = 4
// waves on 4 consecutive tiles; each wave walks the voices of its group (so the 8 KB sin/cos table in LDS
// is filled once per block, not once per voice) and stores one coalesced row segment per voice.
template <int FPL>
__global__ __launch_bounds__(256, FPL >= 4 ? 4 : 6) void k_generate(BankPtrs B, const shm::sc_pair* __restrict__ trig_g, uint32_t first,
                                                  uint32_t nvoices, uint32_t voices_per_group,
                                                  const VoiceLaunch* __restrict__ launch,
                                                  const VoiceFM* __restrict__ launch_fm,
                                                  uint64_t start, uint32_t n,
                                                  const double* __restrict__ fm_cumsum,
                                                  const double* __restrict__ pwm,
                                                  float* __restrict__ out32, double* __restrict__ out64,
                                                  size_t stride) {
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    for (uint32_t k = threadIdx.x; k < shm::TRIG_N; k += 256) trig[k] = trig_g[k];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile0 = (blockIdx.x * 4 + wave) * (64 * FPL);
    if (tile0 >= n) return;
    uint32_t tile_last = tile0 + 64 * FPL - 1;
    if (tile_last > n - 1) tile_last = n - 1;
    uint32_t i[FPL];
    double di[FPL];
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        uint32_t raw = tile0 + j * 64 + lane;
        i[j] = raw < n ? raw : n - 1;
        di[j] = (double)i[j];
    }
    const uint32_t v0 = blockIdx.y * voices_per_group;
    uint32_t v1 = v0 + voices_per_group;
    if (v1 > nvoices) v1 = nvoices;
    const VoiceLaunch SH_CONST_AS* rp = as_const(launch) + v0;
    for (uint32_t vi = v0; vi < v1; ++vi, ++rp) {
        const VoiceRegs r = load_record(rp);
        double x[FPL];
        if (r.flags & FL_SILENT) {
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = 0.0;
        } else {
            voice_block<FPL, false>(r, launch_fm + vi, B, B.voices + first + vi, start, tile_last, i, di, fm_cumsum, pwm, trig, x);
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            uint32_t raw = tile0 + j * 64 + lane;
            if (raw < n) {
                if (out32) out32[(size_t)vi * stride + raw] = (float)x[j];
                if (out64) out64[(size_t)vi * stride + raw] = x[j];
            }
        }
    }
}

// Whole-bank materialisation through the launch's voice lists (see LaunchSet): grid = (groups of 4 tiles, 64-voice chunks);
// a wave owns one tile and walks the chunk's lean records with the loop of k_bank_render (the sample is rounded and
// stored instead of accumulated), then the general list through voice_block, then zero-fills the rows of silent voices.
// The lean arithmetic repeats the general code's order -- ((x * amplitude) + 0) * g0u.
// LEAN = false: only the general and the silent list (the lean records went through k_generate_lean_harm).
template <int FPL, bool LEAN>
__global__ __launch_bounds__(256, FPL >= 4 ? 4 : 6) void k_generate_lists(BankPtrs B, const shm::sc_pair* __restrict__ trig_g,
                                                                         uint32_t nvoices, LaunchSet cur, uint64_t start, uint32_t n,
                                                                         float* __restrict__ out32, size_t stride,
                                                                         SegTab tab = SegTab{0, {}, {}, {}}, uint32_t vsplit = 1) {
    // tab.n != 0: the unequal segments of a row's head in ONE launch -- grid.x runs over the groups of four tiles of all segments,
    // a workgroup finds its segment and from there on works relative to it (records, frames, output).  vsplit > 1: grid.y =
    // chunks x vsplit, the general and silent voices of a chunk dealt round robin to vsplit workgroups (rows are independent:
    // the first segment of a head holds EVERY voice, 64 per chunk, and a wave walking them all is the launch's critical path).
    uint32_t tg = blockIdx.x;
    if (tab.n) {
        uint32_t sidx = 0;
        for (;;) {
            const uint32_t groups_s = (tab.len[sidx] + 4 * 64 * FPL - 1) / (4 * 64 * FPL);
            if (tg < groups_s) break;
            tg -= groups_s;
            if (++sidx >= tab.n) return;
        }
        cur = segment_set(cur, tab.set[sidx], nvoices);
        start += tab.first[sidx];
        n = tab.len[sidx];
        out32 += tab.first[sidx];
    }
    const uint32_t c = blockIdx.y / vsplit, vsub = blockIdx.y % vsplit;
    if constexpr (!LEAN) {                     // nothing but lean voices in this chunk: leave before the table is staged
        const uint32_t SH_CONST_AS* cnt0 = as_const(cur.counts) + 4 * c;
        if (cnt0[1] + cnt0[2] == 0) return;
    }
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    for (uint32_t k = threadIdx.x; k < shm::TRIG_N; k += 256) trig[k] = trig_g[k];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile0 = (tg * 4 + wave) * (64 * FPL);
    if (tile0 >= n) return;
    uint32_t tile_last = tile0 + 64 * FPL - 1;
    if (tile_last > n - 1) tile_last = n - 1;
    uint32_t i[FPL];
    double di[FPL];
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        uint32_t raw = tile0 + j * 64 + lane;
        i[j] = raw < n ? raw : n - 1;
        di[j] = (double)i[j];
    }
    const uint32_t SH_CONST_AS* cnt = as_const(cur.counts) + 4 * c;
    const uint32_t nfast = (LEAN && vsub == 0) ? cnt[0] : 0u, ngen = cnt[1], nsilent = cnt[2];
    const FastRec SH_CONST_AS* q = as_const(cur.fast) + c * 64;
    for (uint32_t p = 0; p < nfast; ++p, ++q) {
        const uint32_t remain = q->remain, kind = q->kind, vi = q->vi;
        const double amp = q->amplitude, g0u = q->g0u;
        const double ta = q->t_base, da = q->dt, rca = q->rot_c, rsa = q->rot_s;
        const double tb = q->t0_b, db = q->dt_b, rcb = q->rot_c_b, rsb = q->rot_s_b, ob = q->off_b;
        double poly[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) poly[u] = q->poly[u];
        asm volatile("" :: "s"(amp), "s"(g0u), "s"(remain), "s"(kind), "s"(vi), "s"(ta), "s"(da), "s"(rca), "s"(rsa), "s"(tb), "s"(db),
                     "s"(rcb), "s"(rsb), "s"(ob), "s"(poly[0]), "s"(poly[1]), "s"(poly[2]), "s"(poly[3]), "s"(poly[4]), "s"(poly[5]),
                     "s"(poly[6]), "s"(poly[7]), "s"(poly[8]), "s"(poly[9]), "s"(poly[10]), "s"(poly[11]), "s"(poly[12]),
                     "s"(poly[13]), "s"(poly[14]), "s"(poly[15]));
        double x[FPL];
        if (kind == LEAN_FM) {
            double T[FPL];
            if (remain == 0xFFFFFFFFu || tile_last < remain) {
#pragma unroll
                for (int j = 0; j < FPL; ++j) T[j] = fma(di[j], da, ta);
            } else if (tile0 >= remain) {
#pragma unroll
                for (int j = 0; j < FPL; ++j) T[j] = fma(di[j] - ob, db, tb);
            } else {
#pragma unroll
                for (int j = 0; j < FPL; ++j) T[j] = i[j] < remain ? fma(di[j], da, ta) : fma(di[j] - ob, db, tb);
            }
            double ls, lc, th[FPL], sn[FPL], cs[FPL];
            shm::sincos_tab(fma(di[0], poly[4], poly[3]), trig, ls, lc);
#pragma unroll
            for (int j = 0; j < FPL; ++j) {
                if (j > 0) {
                    const double ns = fma(ls, poly[8], lc * poly[9]), nc = fma(lc, poly[8], -(ls * poly[9]));
                    ls = ns;
                    lc = nc;
                }
                const double Ln = fma(poly[5], poly[6] - lc, poly[7] * (poly[10] + di[j]));
                th[j] = poly[0] * T[j] + fma(poly[2], Ln, poly[1]);
            }
            shm::sincos_tab_n<FPL>(th, trig, sn, cs);
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = sn[j];
        } else if (kind >= LEAN_SAW) {                       // Sawtooth / Square / Triangle / Pulse at unit amplitude
            double th[FPL];
            if (remain == 0xFFFFFFFFu || tile_last < remain) {
#pragma unroll
                for (int j = 0; j < FPL; ++j) th[j] = fma(di[j], da, ta);
            } else if (tile0 >= remain) {
#pragma unroll
                for (int j = 0; j < FPL; ++j) th[j] = fma(di[j] - ob, db, tb);
            } else {
#pragma unroll
                for (int j = 0; j < FPL; ++j) th[j] = i[j] < remain ? fma(di[j], da, ta) : fma(di[j] - ob, db, tb);
            }
#pragma unroll
            for (int j = 0; j < FPL; ++j)
                x[j] = kind == LEAN_SAW ? shm::saw_value(th[j], 2.0, 0.0)
                     : kind == LEAN_SQUARE ? shm::square_value(th[j], 1.0, 0.0)
                     : kind == LEAN_TRIANGLE ? shm::triangle_value(th[j], 4.0, 0.0)
                     : shm::pulse_value(th[j], poly[0], 1.0, 0.0);
        } else {
            double sn[FPL], cs[FPL], pv[FPL];
            if (remain == 0xFFFFFFFFu || tile0 >= remain || tile_last < remain) {
                const bool on_b = remain != 0xFFFFFFFFu && tile0 >= remain;
                const double t_base = on_b ? tb : ta, dt = on_b ? db : da;
                const double rc = on_b ? rcb : rca, rs = on_b ? rsb : rsa;
                const double off = on_b ? ob : 0.0;
                shm::sincos_tab(fma(di[0] - off, dt, t_base), trig, sn[0], cs[0]);
#pragma unroll
                for (int j = 1; j < FPL; ++j) {
                    sn[j] = fma(sn[j - 1], rc, cs[j - 1] * rs);
                    cs[j] = fma(cs[j - 1], rc, -(sn[j - 1] * rs));
                }
            } else {
                double th[FPL];
#pragma unroll
                for (int j = 0; j < FPL; ++j) th[j] = i[j] < remain ? fma(di[j], da, ta) : fma(di[j] - ob, db, tb);
                shm::sincos_tab_n<FPL>(th, trig, sn, cs);
            }
#pragma unroll
            for (int j = 0; j < FPL; ++j) pv[j] = fma(poly[0], cs[j], poly[1]);
#pragma unroll
            for (int u = 2; u < 16; ++u) {
#pragma unroll
                for (int j = 0; j < FPL; ++j) pv[j] = fma(pv[j], cs[j], poly[u]);
            }
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = kind == LEAN_SINE ? sn[j] : pv[j] * sn[j];
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const uint32_t raw = tile0 + j * 64 + lane;
            if (raw < n) out32[(size_t)vi * stride + raw] = (float)((x[j] * amp + 0.0) * g0u);
        }
    }
    const uint32_t SH_CONST_AS* idx = as_const(cur.gen_idx) + c * 64;
    for (uint32_t p = vsub; p < ngen; p += vsplit) {
        const uint32_t vi = idx[p];
        const VoiceRegs r = load_record(as_const(cur.launch) + vi);
        double x[FPL];
        voice_block<FPL, false>(r, cur.fm + vi, B, B.voices + vi, start, tile_last, i, di, nullptr, nullptr, trig, x);
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const uint32_t raw = tile0 + j * 64 + lane;
            if (raw < n) out32[(size_t)vi * stride + raw] = (float)x[j];
        }
    }
    for (uint32_t p = vsub; p < nsilent; p += vsplit) {
        const uint32_t vi = idx[63 - p];
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const uint32_t raw = tile0 + j * 64 + lane;
            if (raw < n) out32[(size_t)vi * stride + raw] = 0.0f;
        }
    }
}

// Materialisation of the lean polynomial-Harmonics records of a launch (banks whose lean candidates are all of that kind):
// grid = (groups of 4 tiles of 64*FPL frames, 64-voice chunks); a wave owns one tile and walks the chunk's lean records with
// the arithmetic of the render kernel's lean loop -- one table lookup, one rotation, the three-term recurrence, Horner -- and
// stores FPL coalesced 256-byte row segments per record: base address of the row in SGPRs, one lane offset for all rows, the
// frame's 256*j bytes as the instruction's immediate.  No general code in this kernel (k_generate_lists<4, false> follows
// for the general and silent lists): 4 B written per voice-sample is what should bind it, not the scalar unit.
// The sample repeats the general code's order, ((x * amplitude) + 0) * g0u, before its one rounding to float32.
// A long row is cut into segments of seg_frames (a multiple of the tile) with a record set each (k_prepare_segments): a
// record describes at most two phase-table pieces, and ten seconds of a high voice run through more.
template <int FPL>
__global__ __launch_bounds__(256, 4) void k_generate_lean_harm(const shm::sc_pair* __restrict__ trig_g, LaunchSet base, uint32_t nvoices,
                                                               uint32_t total, uint32_t seg_frames,
                                                               float* __restrict__ out32_all, size_t stride, SegTab tab) {
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    for (uint32_t k = threadIdx.x; k < shm::TRIG_N; k += 256) trig[k] = trig_g[k];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t seg, seg_first, n, tile0;                                 // (uniform) the wave's segment, and everything relative to it
    if (tab.n) {                                                       // unequal segments: the head of rows that start with the notes
        uint32_t t = blockIdx.x * 4 + wave;
        seg = 0;
        for (;;) {
            const uint32_t tiles_s = (tab.len[seg] + 64 * FPL - 1) / (64 * FPL);
            if (t < tiles_s) break;
            t -= tiles_s;
            if (++seg >= tab.n) return;
        }
        seg_first = tab.first[seg];
        n = tab.len[seg];
        tile0 = t * (64 * FPL);
        seg = tab.set[seg];
    } else {
        const uint32_t abs0 = (blockIdx.x * 4 + wave) * (64 * FPL);   // the tile's first frame in the whole launch
        if (abs0 >= total) return;
        seg = abs0 / seg_frames;
        seg_first = seg * seg_frames;
        n = total - seg_first < seg_frames ? total - seg_first : seg_frames;
        tile0 = abs0 - seg_first;
    }
    const LaunchSet cur = segment_set(base, seg, nvoices);
    float* __restrict__ out32 = out32_all + seg_first;
    uint32_t tile_last = tile0 + 64 * FPL - 1;
    if (tile_last > n - 1) tile_last = n - 1;
    const uint32_t c = blockIdx.y;
    const uint32_t nfast = as_const(cur.counts)[4 * c];
    const FastRec SH_CONST_AS* q = as_const(cur.fast) + c * 64;
    const uint32_t i0 = tile0 + lane;
    const double di0 = (double)i0;
    float* __restrict__ col = out32 + i0;                      // + vi * stride per record (uniform), + 64 * j per frame
    for (uint32_t p = 0; p < nfast; ++p, ++q) {
        const uint32_t remain = q->remain, vi = q->vi;
        const double amp = q->amplitude, g0u = q->g0u;
        const double ta = q->t_base, da = q->dt, rca = q->rot_c, rsa = q->rot_s;
        double poly[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) poly[u] = q->poly[u];
        asm volatile("" :: "s"(amp), "s"(g0u), "s"(remain), "s"(vi), "s"(ta), "s"(da), "s"(rca), "s"(rsa),
                     "s"(poly[0]), "s"(poly[1]), "s"(poly[2]), "s"(poly[3]), "s"(poly[4]), "s"(poly[5]),
                     "s"(poly[6]), "s"(poly[7]), "s"(poly[8]), "s"(poly[9]), "s"(poly[10]), "s"(poly[11]), "s"(poly[12]),
                     "s"(poly[13]), "s"(poly[14]), "s"(poly[15]));
        double t_base = ta, dt = da, rc = rca, rs = rsa, off = 0.0, tb = ta, db = da, ob = 0.0;
        bool straddle = false;
        if (remain != 0xFFFFFFFFu && tile_last >= remain) {   // not wholly on the first piece
            tb = q->t0_b; db = q->dt_b; ob = q->off_b;
            const double rcb = q->rot_c_b, rsb = q->rot_s_b;
            straddle = tile0 < remain;
            if (!straddle) { t_base = tb; dt = db; rc = rcb; rs = rsb; off = ob; }
        }
        const LaneTheta theta{di0, t_base, dt, off, ta, da, tb, db, ob, i0, remain, straddle};
        double s0, c0, s1, c1;
        shm::sincos_tab(theta(0), trig, s0, c0);
        if (straddle) {
            shm::sincos_tab(theta(1), trig, s1, c1);
        } else {
            s1 = fma(s0, rc, c0 * rs);
            c1 = fma(c0, rc, -(s0 * rs));
        }
        const double k2 = rc + rc;
        float* __restrict__ row = col + (size_t)vi * stride;
        // amplitude and (constant) envelope gain scale the sine ONCE: the recurrence is linear, so every later frame's sine
        // arrives scaled and the sample is one product, p * s (differs from the general code's ((x * amplitude) + 0) * g0u
        // by float64 rounding only)
        const double ag = amp * g0u;
        s0 *= ag;
        s1 *= ag;
        auto frames = [&](auto full_tile) {
            constexpr bool FULL = decltype(full_tile)::value;
#pragma unroll
            for (int h = 0; h < FPL; h += 2) {
                double p0 = fma(poly[0], c0, poly[1]), p1 = fma(poly[0], c1, poly[1]);
#pragma unroll
                for (int u = 2; u < 16; ++u) {
                    p0 = fma(p0, c0, poly[u]);
                    p1 = fma(p1, c1, poly[u]);
                }
                // streaming stores: the rows are not read again by this kernel (1.97 GB per launch of the benchmark shape)
                if (FULL || i0 + (uint32_t)h * 64u < n) __builtin_nontemporal_store((float)(p0 * s0), row + h * 64);
                if (FULL || i0 + (uint32_t)(h + 1) * 64u < n) __builtin_nontemporal_store((float)(p1 * s1), row + (h + 1) * 64);
                if (h + 2 < FPL) {
                    if (straddle) {
                        shm::sincos_tab(theta(h + 2), trig, s0, c0);
                        shm::sincos_tab(theta(h + 3), trig, s1, c1);
                        s0 *= ag;
                        s1 *= ag;
                    } else {
                        const double s2 = fma(k2, s1, -s0), c2 = fma(k2, c1, -c0);
                        const double s3 = fma(k2, s2, -s1), c3 = fma(k2, c2, -c1);
                        s0 = s2; c0 = c2; s1 = s3; c1 = c3;
                    }
                }
            }
        };
        if (tile0 + 64 * FPL <= n) frames(std::true_type()); else frames(std::false_type());
    }
}

// One voice through the general code, accumulated into the lane's partial bus.  With more than four frames per lane
// (the lean loop likes eight: one table lookup serves them all) the general code still runs four at a time, so that its
// register needs stay those of the four-frame kernel.
template <int FPL>
__device__ __forceinline__ void general_voice(const VoiceRegs& r, const VoiceFM* __restrict__ fmrec, const BankPtrs& B,
                                              const sh_voice* __restrict__ vfull, uint64_t start, uint32_t tile0, uint32_t nframes,
                                              const uint32_t (&i)[FPL], const double (&di)[FPL], TrigTab trig,
                                              double (&accl)[FPL], double (&accr)[FPL]) {
    constexpr int GF = FPL > 4 ? 4 : FPL;
    static_assert(FPL % GF == 0, "frames per lane: 1, 2, 4 or a multiple of 4");
    const double* fm_p = nullptr;
    const double* pwm_p = nullptr;
    if (B.rows) {                                              // (uniform) this voice's rows of the launch's modulation matrix
        const uint32_t vidx = (uint32_t)(vfull - B.voices);
        const int32_t fr = as_const(B.fm_row)[vidx], pr = as_const(B.pwm_row)[vidx];
        if (fr >= 0) fm_p = B.rows + (size_t)fr * B.row_stride;
        if (pr >= 0) pwm_p = B.rows + (size_t)pr * B.row_stride;
    }
#pragma unroll
    for (int h = 0; h < FPL / GF; ++h) {
        const uint32_t first = tile0 + (uint32_t)h * 64 * GF;
        if (h > 0 && first >= nframes) break;                 // uniform: this part of the tile lies beyond the launch
        uint32_t last = first + 64 * GF - 1;
        if (last > nframes - 1) last = nframes - 1;
        uint32_t ih[GF];
        double dh[GF], x[GF];
#pragma unroll
        for (int j = 0; j < GF; ++j) { ih[j] = i[h * GF + j]; dh[j] = di[h * GF + j]; }
        voice_block<GF, true>(r, fmrec, B, vfull, start, last, ih, dh, fm_p, pwm_p, trig, x);
#pragma unroll
        for (int j = 0; j < GF; ++j) {
            accl[h * GF + j] = fma(r.gain_l, x[j], accl[h * GF + j]);
            accr[h * GF + j] = fma(r.gain_r, x[j], accr[h * GF + j]);
        }
    }
}

// fused generate-and-mix.  grid = (frame tiles, voice groups); block = WAVES waves on ONE tile of 64*FPL
// frames; wave w walks voices v0+w, v0+w+WAVES, ... of its group with the voice record in SGPRs; float64
// partial (L, R) per lane; LDS-staged sum across the waves; one store per frame.  With one group the block
// writes the final bus; with several it writes a float64 partial bus per group and k_bus_combine folds
// them in group order -- either way voices are summed in a fixed order (reproducible run to run).
// MODE (a static property of the bank): RENDER_DIRECT -- no voice could ever take the lean loop: walk the voice table
// directly; RENDER_LEAN_HARM -- lean loop for polynomial Harmonics only; RENDER_LEAN_ALL -- also for FM Sine voices (kept
// out of the Harmonics-only kernel: the extra branch and code cost its loop 5 %).
// One stereo frame of the float64 bus as saturated int16 PCM: what sh_quantize_clip_f32 makes of the float32 bus
// (rounded to float32 first, float64 product with the scale, truncation toward zero, clamp; NaN -> 0), packed (L | R << 16).
__device__ __forceinline__ uint32_t pcm16_frame(double l, double r, double scale) {
    double tl = trunc(scale * (double)(float)l), tr = trunc(scale * (double)(float)r);
    tl = tl != tl ? 0.0 : (tl > 32767.0 ? 32767.0 : (tl < -32768.0 ? -32768.0 : tl));
    tr = tr != tr ? 0.0 : (tr > 32767.0 ? 32767.0 : (tr < -32768.0 ? -32768.0 : tr));
    return ((uint32_t)(int)tl & 0xFFFFu) | ((uint32_t)(int)tr << 16);
}

// The lean Harmonics arithmetic for FPL frames of one lane, 64 samples apart: frame 0 by table lookup (s0, c0), frame 1 by
// one rotation (s1, c1), frames j >= 2 by the three-term recurrence x[j] = k2 x[j-1] - x[j-2], k2 = 2cos(64 dt) -- one FMA
// per value where a rotation takes two FMAs and two MULs -- consumed pair by pair, so that no array of FPL sines ever
// lives in registers and the accumulators are updated in place on every path.  The one tile per crossing that straddles a
// phase-table piece end (`straddle`, wave-uniform) cannot step that way: there every pair is looked up afresh from
// theta(frame), which picks the piece per lane.
// Rounding errors propagate through the recurrence like U_j(cos(64 dt)), i.e. grow at most linearly in j (FPL <= 8: < 25
// ulp worst case, ~2 ulp typically); the rounding of k2 itself shifts the step angle by <= 1.1e-16 / |sin(64 dt)| per
// step: below 1e-9 relative for all but ~1e-6 of voices, orders of magnitude inside the 1e-6 RMS contract in any case.
// SLOPED (segmented transition launches): the gains follow the envelope's line, gl + i * gls at the lane's frame i = dl + 64 j.
template <int FPL, bool SLOPED = false, typename Theta>
__device__ __forceinline__ void lean_harm_frames(double s0, double c0, double s1, double c1, double k2, bool straddle, Theta theta,
                                                 TrigTab trig, const double (&poly)[16], double gl, double gr,
                                                 double (&accl)[FPL], double (&accr)[FPL],
                                                 double gls = 0.0, double grs = 0.0, double dl = 0.0) {
#pragma unroll
    for (int h = 0; h < FPL; h += 2) {
        const bool two = h + 1 < FPL;                 // compile-time after unrolling (FPL = 1: a single frame)
        double p0 = fma(poly[0], c0, poly[1]), p1 = fma(poly[0], c1, poly[1]);
#pragma unroll
        for (int u = 2; u < 16; ++u) {
            p0 = fma(p0, c0, poly[u]);
            if (two) p1 = fma(p1, c1, poly[u]);
        }
        const double x0 = p0 * s0;
        if (SLOPED) {
            const double i0 = dl + (double)(h * 64);
            accl[h] = fma(fma(i0, gls, gl), x0, accl[h]);
            accr[h] = fma(fma(i0, grs, gr), x0, accr[h]);
        } else {
            accl[h] = fma(gl, x0, accl[h]);
            accr[h] = fma(gr, x0, accr[h]);
        }
        if (two) {
            const double x1 = p1 * s1;
            if (SLOPED) {
                const double i1 = dl + (double)((h + 1) * 64);
                accl[h + 1 < FPL ? h + 1 : h] = fma(fma(i1, gls, gl), x1, accl[h + 1 < FPL ? h + 1 : h]);
                accr[h + 1 < FPL ? h + 1 : h] = fma(fma(i1, grs, gr), x1, accr[h + 1 < FPL ? h + 1 : h]);
            } else {
                accl[h + 1 < FPL ? h + 1 : h] = fma(gl, x1, accl[h + 1 < FPL ? h + 1 : h]);
                accr[h + 1 < FPL ? h + 1 : h] = fma(gr, x1, accr[h + 1 < FPL ? h + 1 : h]);
            }
        }
        if (h + 2 < FPL) {
            if (straddle) {
                shm::sincos_tab(theta(h + 2), trig, s0, c0);
                if (h + 3 < FPL) shm::sincos_tab(theta(h + 3), trig, s1, c1);
            } else {
                const double s2 = fma(k2, s1, -s0), c2 = fma(k2, c1, -c0);
                const double s3 = fma(k2, s2, -s1), c3 = fma(k2, c2, -c1);
                s0 = s2; c0 = c2; s1 = s3; c1 = c3;
            }
        }
    }
}

// ... RENDER_LEAN_HARM_ONLY / RENDER_GENERAL_ONLY: the same launch as TWO kernels on one stream (banks whose lean candidates are all
// polynomial Harmonics, several voice groups).  The lean kernel compiles without the general code and so without its
// registers, scalar pressure and scratch (108 VGPRs at eight frames per lane where the combined kernel needs 128 + 280 bytes of
// scratch; 37 instead of 42 us per block); the general kernel walks the general lists with four frames per lane, writes its
// partial buses behind the lean kernel's (parts[groups + g]) and sets gen_valid[g] -- or, for a group without general voices
// (the steady state of a note), leaves after one scalar load.  The fold adds the general parts whose flag is set.
// RENDER_LEAN_HARM_SEG / RENDER_GENERAL_SEG: the split launch of a TRANSITION block (the first block of a note: the phase sum
// runs through a dozen binades, the envelope through attack and decay) cut into segments with a record set each.  Piece ends
// lie an octave apart (n = (2^k - t0) / inc), so a segment [a, b) with b <= 2a crosses at most one per voice -- which a lean
// record handles -- where the launch as a whole crosses ten and sends every voice through the general code.  grid.x runs
// over the tiles of all segments; a workgroup finds its segment first and from there on works in the segment's frame of
// reference (records, tile, clamps); only its stores are launch-relative again.
enum { RENDER_DIRECT = 0, RENDER_LEAN_HARM = 1, RENDER_LEAN_ALL = 2, RENDER_LEAN_HARM_ONLY = 3, RENDER_GENERAL_ONLY = 4, RENDER_LEAN_ALL_ONLY = 5,
       RENDER_LEAN_HARM_SEG = 6, RENDER_GENERAL_SEG = 7, RENDER_LEAN_ALL_SEG = 8 };
constexpr bool mode_lean_harm(int mode) { return mode == RENDER_LEAN_HARM || mode == RENDER_LEAN_HARM_ONLY || mode == RENDER_LEAN_HARM_SEG; }
constexpr bool mode_lean_only(int mode) { return mode == RENDER_LEAN_HARM_ONLY || mode == RENDER_LEAN_ALL_ONLY || mode == RENDER_LEAN_HARM_SEG || mode == RENDER_LEAN_ALL_SEG; }
constexpr bool mode_general(int mode) { return mode == RENDER_GENERAL_ONLY || mode == RENDER_GENERAL_SEG; }
constexpr bool mode_seg(int mode) { return mode == RENDER_LEAN_HARM_SEG || mode == RENDER_GENERAL_SEG || mode == RENDER_LEAN_ALL_SEG; }
constexpr bool mode_has_lean(int mode) { return mode != RENDER_DIRECT && !mode_general(mode); }
template <int WAVES, int FPL, int MINW, int MODE>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_bank_render(BankPtrs B, const shm::sc_pair* __restrict__ trig_g,
                                                                  uint32_t nvoices, uint32_t voices_per_group,
                                                                  LaunchSet cur, LaunchSet next, uint64_t next_start,
                                                                  uint64_t start, uint32_t nframes,
                                                                  float2* __restrict__ bus32,
                                                                  double2* __restrict__ bus64,
                                                                  double2* __restrict__ parts,
                                                                  const double2* __restrict__ prev_parts,
                                                                  float2* __restrict__ prev_bus32,
                                                                  double2* __restrict__ prev_bus64,
                                                                  uint32_t* __restrict__ pcm16, double pcm_scale,
                                                                  uint32_t* __restrict__ prev_pcm16, double prev_pcm_scale,
                                                                  uint32_t* __restrict__ gen_valid,
                                                                  const uint32_t* __restrict__ prev_gen_valid) {
    if constexpr (MODE == RENDER_GENERAL_ONLY) {       // (a segmented launch always writes its parts: see below)
        // a group without general voices in this launch: nothing to render, nothing to write (wave-uniform: scalar loads)
        const uint32_t c0g = (blockIdx.y * voices_per_group) / 64;
        uint32_t c1g = ((blockIdx.y + 1) * voices_per_group + 63) / 64;
        const uint32_t nch = (nvoices + 63) / 64;
        if (c1g > nch) c1g = nch;
        uint32_t total = 0;
        for (uint32_t c = c0g; c < c1g; ++c) total += as_const(cur.counts)[4 * c + 1];
        if (blockIdx.x == 0 && threadIdx.x == 0) gen_valid[blockIdx.y] = total ? 1u : 0u;
        if (total == 0) return;
    }
    if constexpr (MODE == RENDER_GENERAL_SEG) {
        // most (tile, group) pairs of a segmented launch hold no general voice: their parts are zeros, written before any set-up
        // (the flags of a segmented launch say "every group's general parts are valid": its general voices are the first segment's)
        // grid.x: the first segment's tiles gen_sub times over (tile-major), then the other segments' tiles
        const uint32_t tiles_0 = (B.seg_first[1] - B.seg_first[0] + 64 * FPL - 1) / (64 * FPL);
        uint32_t sidx = 0, tidx = blockIdx.x;
        if (tidx >= tiles_0 * B.gen_sub) {
            tidx -= tiles_0 * B.gen_sub;
            sidx = 1;
            for (;;) {
                const uint32_t n_s = B.seg_first[sidx + 1] - B.seg_first[sidx];
                const uint32_t tiles_s = (n_s + 64 * FPL - 1) / (64 * FPL);
                if (tidx < tiles_s || sidx + 1 >= B.nseg) break;
                tidx -= tiles_s;
                ++sidx;
            }
        }
        const uint32_t grp = blockIdx.y;
        if (sidx > 0) {
            const uint32_t nchs = (nvoices + 63) / 64;
            const uint32_t c0g = (grp * voices_per_group) / 64;
            uint32_t c1g = ((grp + 1) * voices_per_group + 63) / 64;
            if (c1g > nchs) c1g = nchs;
            const uint32_t SH_CONST_AS* cnt = as_const(cur.counts) + (size_t)sidx * 4 * nchs;
            uint32_t total = 0;
            for (uint32_t c = c0g; c < c1g; ++c) total += cnt[4 * c + 1];
            if (total == 0) {
                const uint32_t n_s = B.seg_first[sidx + 1] - B.seg_first[sidx];
                for (uint32_t f = threadIdx.x; f < 64 * FPL; f += WAVES * 64) {
                    const uint32_t raw = tidx * (64 * FPL) + f;
                    if (raw < n_s) parts[(size_t)(gridDim.y + grp) * nframes + B.seg_first[sidx] + raw] = make_double2(0.0, 0.0);
                }
                return;
            }
        }
    }
    // The previous launch of the stream (same shape) left its voice groups' partial buses unfolded: the workgroups of
    // group 0 fold their tile of it now, in group order, before their own work -- instead of a 5 us kernel between
    // every two render launches.
    if (!mode_general(MODE) && prev_parts && blockIdx.y == 0) {
        for (uint32_t f = threadIdx.x; f < 64 * FPL; f += WAVES * 64) {
            const uint32_t raw = blockIdx.x * (64 * FPL) + f;
            if (raw >= nframes) continue;
            double2 acc = prev_parts[raw];
            for (uint32_t g = 1; g < gridDim.y; ++g) {
                const double2 pp = prev_parts[(size_t)g * nframes + raw];
                acc.x += pp.x;
                acc.y += pp.y;
            }
            if (prev_gen_valid) {                              // the general kernel's parts of that launch, where it wrote any
                for (uint32_t g = 0; g < gridDim.y; ++g) {
                    if (as_const(prev_gen_valid)[g]) {
                        const double2 pp = prev_parts[(size_t)(gridDim.y + g) * nframes + raw];
                        acc.x += pp.x;
                        acc.y += pp.y;
                    }
                }
            }
            if (prev_bus32) prev_bus32[raw] = make_float2((float)acc.x, (float)acc.y);
            if (prev_bus64) prev_bus64[raw] = acc;
            if (prev_pcm16) prev_pcm16[raw] = pcm16_frame(acc.x, acc.y, prev_pcm_scale);
        }
    }
    // Sequential streaming is the common call pattern: this launch also resolves the launch records of the block
    // expected two launches on (next_start = start + 2 * nframes: the launch in between runs beside this one on the
    // other stream and got its records from this one's predecessor) into a free record set, so that launch needs no prepare
    // kernel of its own (a 15 us kernel + a launch boundary per block otherwise).
    // The chunks of 64 voices are spread over the first workgroups (one wavefront each, on different CUs): a single
    // workgroup doing all of it competes with three rendering workgroups for its CU and ends up as the launch's tail.
    if (!mode_general(MODE) && next.launch) {
        const uint32_t nchunks = (nvoices + 63) / 64, nblocks = gridDim.x * gridDim.y;
        const uint32_t bid = blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x < 64) {
            for (uint32_t c = bid; c < nchunks; c += nblocks) prepare_chunk(B, next, c, nvoices, next_start, nframes);
        }
    }
    __shared__ double red[WAVES][2][64 * FPL];
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    for (uint32_t k = threadIdx.x; k < shm::TRIG_N; k += WAVES * 64) trig[k] = trig_g[k];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a segmented launch: this workgroup's segment and its tile there (uniform); everything below up to the stores is
    // relative to the segment.  Otherwise the launch is its own single segment.
    uint32_t seg_off = 0, nfr = nframes, tile_index = blockIdx.x;
    uint64_t st0 = start;
    LaunchSet curS = cur;
    uint32_t sub = 0, nsub = 1;
    bool to_scratch = false;
    if constexpr (mode_seg(MODE)) {
        uint32_t sidx = 0;
        if constexpr (MODE == RENDER_GENERAL_SEG) {
            const uint32_t tiles_0 = (B.seg_first[1] - B.seg_first[0] + 64 * FPL - 1) / (64 * FPL);
            if (tile_index < tiles_0 * B.gen_sub) {            // the first segment: gen_sub workgroups per (tile, group)
                sub = tile_index % B.gen_sub;
                tile_index /= B.gen_sub;
                nsub = B.gen_sub;
                to_scratch = true;
            } else {
                tile_index -= tiles_0 * (B.gen_sub - 1);       // as if the first segment's tiles came once
            }
        }
        for (;;) {
            const uint32_t n_s = B.seg_first[sidx + 1] - B.seg_first[sidx];
            const uint32_t tiles_s = (n_s + 64 * FPL - 1) / (64 * FPL);
            if (tile_index < tiles_s || sidx + 1 >= B.nseg) break;
            tile_index -= tiles_s;
            ++sidx;
        }
        seg_off = B.seg_first[sidx];
        nfr = B.seg_first[sidx + 1] - seg_off;
        st0 = start + seg_off;
        curS = segment_set(cur, sidx, nvoices);
    }
    const uint32_t grp = blockIdx.y;                           // the voice group of this workgroup
    const uint32_t tile0 = tile_index * (64 * FPL);
    uint32_t tile_last = tile0 + 64 * FPL - 1;
    if (tile_last > nfr - 1) tile_last = nfr - 1;
    // this group's voices: chunks [c0, c1) of 64 voices (voices_per_group is a multiple of 64 unless there is one group)
    const uint32_t c0 = (grp * voices_per_group) / 64;
    uint32_t c1 = ((grp + 1) * voices_per_group + 63) / 64;
    const uint32_t nchunks = (nvoices + 63) / 64;
    if (c1 > nchunks) c1 = nchunks;
    uint32_t i[FPL];
    double di[FPL], accl[FPL], accr[FPL];
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        accl[j] = 0.0;
        accr[j] = 0.0;
    }
    // the frames of this lane, launch-relative (clamped into the launch: out-of-range lanes compute a valid sample and do not
    // store it).  The lean Harmonics loop does not use the arrays -- it works from the lane's first frame alone -- so in that
    // mode they are only built after it, for the general code: they would cost 3 registers per frame for the whole loop.
    auto build_frames = [&](uint32_t lane_) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            uint32_t raw = tile0 + j * 64 + lane_;
            i[j] = raw < nfr ? raw : nfr - 1;
            di[j] = (double)i[j];
        }
    };
    if constexpr (!mode_has_lean(MODE)) build_frames(lane);        // (the lean loops work from the lane's first frame alone)
    if constexpr (MODE == RENDER_DIRECT) {
        const uint32_t v0 = blockIdx.y * voices_per_group;
        uint32_t v1 = v0 + voices_per_group;
        if (v1 > nvoices) v1 = nvoices;
        const VoiceLaunch SH_CONST_AS* rp = as_const(curS.launch) + v0 + wave;
        for (uint32_t vi = v0 + wave; vi < v1; vi += WAVES, rp += WAVES) {
            const VoiceRegs r = load_record(rp);
            if (r.flags & FL_SILENT) continue;       // the note was released before this block: contributes exact zeros
            general_voice<FPL>(r, curS.fm + vi, B, B.voices + vi, st0, tile0, nfr, i, di, trig, accl, accr);
        }
    } else {
    // ---- fast voices: one table lookup, FPL-1 rotations, the Horner chains, two accumulations per frame ----
    // Wave w takes every WAVES-th list entry; the offset carries over from chunk to chunk so that the waves'
    // shares of the whole group differ by at most one voice.
    uint32_t first = wave + sub * WAVES;                      // position in the current chunk's list this wave starts at
    if constexpr (!mode_general(MODE)) {
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t nfast = as_const(curS.counts)[4 * c];
        const FastRec SH_CONST_AS* q = as_const(curS.fast) + c * 64 + first;
        uint32_t p = first;
        for (; p < nfast; p += WAVES, q += WAVES) {
            // the common 192 bytes in ONE batch of scalar loads: the empty asm makes these fields live here, so the
            // compiler cannot sink their loads behind the tests below (it did: three dependent round trips per voice).
            // The second piece's fields are NOT in the list: their loads stay inside the rare crossing branches.
            const double gl = q->gain_l, gr = q->gain_r;
            const uint32_t remain = q->remain, kind = q->kind;
            const double ta = q->t_base, da = q->dt, rca = q->rot_c, rsa = q->rot_s;
            double poly[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) poly[u] = q->poly[u];
            asm volatile("" :: "s"(gl), "s"(gr), "s"(remain), "s"(kind), "s"(ta), "s"(da), "s"(rca), "s"(rsa),
                         "s"(poly[0]), "s"(poly[1]), "s"(poly[2]), "s"(poly[3]), "s"(poly[4]), "s"(poly[5]),
                         "s"(poly[6]), "s"(poly[7]), "s"(poly[8]), "s"(poly[9]), "s"(poly[10]), "s"(poly[11]), "s"(poly[12]),
                         "s"(poly[13]), "s"(poly[14]), "s"(poly[15]));
            // Every lean kind works from the lane's FIRST frame alone (no per-frame index arrays): the piece of the phase table
            // (first / second of the launch) is chosen by scalar selects, the one tile per crossing that straddles the piece end by
            // a uniform flag, and the frames follow 64 samples apart.
            const uint32_t i0 = tile0 + lane;                         // the lane's first frame (< nfr + 64: harmless)
            const double di0 = (double)i0;
            double t_base = ta, dt = da, rc = rca, rs = rsa, off = 0.0, tb = ta, db = da, ob = 0.0;
            bool straddle = false;
            if constexpr (mode_seg(MODE)) {
                // a segment is cut so that (nearly) every voice crosses ONE piece end in it: half the tiles lie behind it, and the
                // second piece's fields belong in the first batch of loads (one round trip per record, not two)
                const double tb2 = q->t0_b, db2 = q->dt_b, ob2 = q->off_b, rcb = q->rot_c_b, rsb = q->rot_s_b;
                asm volatile("" :: "s"(tb2), "s"(db2), "s"(ob2), "s"(rcb), "s"(rsb));
                if (remain != 0xFFFFFFFFu && tile_last >= remain) {
                    tb = tb2; db = db2; ob = ob2;
                    straddle = tile0 < remain;
                    if (!straddle) { t_base = tb; dt = db; rc = rcb; rs = rsb; off = ob; }
                }
            } else
            if (remain != 0xFFFFFFFFu && tile_last >= remain) {       // not wholly on the first piece
                tb = q->t0_b; db = q->dt_b; ob = q->off_b;
                const double rcb = q->rot_c_b, rsb = q->rot_s_b;
                straddle = tile0 < remain;
                if (!straddle) { t_base = tb; dt = db; rc = rcb; rs = rsb; off = ob; }
            }
            const LaneTheta theta{di0, t_base, dt, off, ta, da, tb, db, ob, i0, remain, straddle};    // the accumulated t at frame j
            if (mode_lean_harm(MODE) || kind == LEAN_HARM) {
                // polynomial Harmonics: lookup + one rotation + the three-term recurrence (lean_harm_frames)
                double s0, c0, s1, c1;
                shm::sincos_tab(theta(0), trig, s0, c0);
                if (straddle) {
                    if (FPL > 1) shm::sincos_tab(theta(1), trig, s1, c1); else { s1 = s0; c1 = c0; }
                } else {
                    s1 = fma(s0, rc, c0 * rs);
                    c1 = fma(c0, rc, -(s0 * rs));
                }
                if constexpr (mode_seg(MODE)) {
                    const double gls = q->amplitude, grs = q->g0u;       // (a segmented launch's records: the gains' slopes per frame)
                    lean_harm_frames<FPL, true>(s0, c0, s1, c1, rc + rc, straddle, theta, trig, poly, gl, gr, accl, accr, gls, grs, di0);
                } else {
                    lean_harm_frames<FPL>(s0, c0, s1, c1, rc + rc, straddle, theta, trig, poly, gl, gr, accl, accr);
                }
                continue;
            }
            if constexpr (!mode_lean_harm(MODE)) {
            if (kind == LEAN_SINE) {
                // a plain Sine: the same recurrence on the sine alone (the amplitude lives in the gains)
                double s0, c0, s1, c1;
                shm::sincos_tab(theta(0), trig, s0, c0);
                if (straddle) {
                    if (FPL > 1) shm::sincos_tab(theta(1), trig, s1, c1); else { s1 = s0; c1 = c0; }
                } else {
                    s1 = fma(s0, rc, c0 * rs);
                }
                const double k2 = rc + rc;
#pragma unroll
                for (int j = 0; j < FPL; ++j) {
                    accl[j] = fma(gl, s0, accl[j]);
                    accr[j] = fma(gr, s0, accr[j]);
                    if (j + 1 < FPL) {
                        double s2;
                        if (straddle) { if (j + 2 < FPL) shm::sincos_tab(theta(j + 2), trig, s2, c1); else s2 = 0.0; }
                        else s2 = fma(k2, s1, -s0);
                        s0 = s1;
                        s1 = s2;
                    }
                }
                continue;
            }
            if (kind == LEAN_FM) {
                // Sine carrier, closed-form Sine LFO (the arithmetic of voice_block's FM path).  theta(j) is the accumulated TIME;
                // the LFO angle a_rel + i*d is exactly linear in i, and only its COSINE enters L(i) = K (C0 - cos) + bias*(start+i):
                // one table lookup for the lane's first frame, one rotation for the second, then cos[j] = 2cos(64d) cos[j-1] - cos[j-2].
                const double frequency = poly[0], phase0 = poly[1], f_inc = poly[2], lfo_a_rel = poly[3], lfo_d = poly[4];
                const double lfo_K = poly[5], lfo_C0 = poly[6], lfo_bias = poly[7], lrc = poly[8], lrs = poly[9], startd = poly[10];
                double ls0, lc0;
                shm::sincos_tab(fma(di0, lfo_d, lfo_a_rel), trig, ls0, lc0);
                double lc1 = fma(lc0, lrc, -(ls0 * lrs));
                const double lk2 = lrc + lrc;
#pragma unroll
                for (int h = 0; h < FPL; h += 4) {            // four carriers at a time: their table reads are in flight together
                    constexpr int Q = FPL < 4 ? FPL : 4;
                    double th[Q], sn[Q], cs[Q];
#pragma unroll
                    for (int jj = 0; jj < Q; ++jj) {
                        const int j = h + jj;
                        const double Ln = fma(lfo_K, lfo_C0 - lc0, lfo_bias * (startd + (di0 + (double)(j * 64))));
                        th[jj] = frequency * theta(j) + fma(f_inc, Ln, phase0);
                        const double lc2 = fma(lk2, lc1, -lc0);
                        lc0 = lc1;
                        lc1 = lc2;
                    }
                    shm::sincos_tab_n<Q>(th, trig, sn, cs);
#pragma unroll
                    for (int jj = 0; jj < Q; ++jj) {
                        accl[h + jj] = fma(gl, sn[jj], accl[h + jj]);
                        accr[h + jj] = fma(gr, sn[jj], accr[h + jj]);
                    }
                }
                continue;
            }
            // Sawtooth / Square / Triangle / Pulse at unit amplitude (the amplitude lives in the gains): t in turns, frame by frame
#pragma unroll
            for (int j = 0; j < FPL; ++j) {
                const double th = theta(j);
                const double x = kind == LEAN_SAW ? shm::saw_value(th, 2.0, 0.0)
                               : kind == LEAN_SQUARE ? shm::square_value(th, 1.0, 0.0)
                               : kind == LEAN_TRIANGLE ? shm::triangle_value(th, 4.0, 0.0)
                               : shm::pulse_value(th, poly[0], 1.0, 0.0);
                accl[j] = fma(gl, x, accl[j]);
                accr[j] = fma(gr, x, accr[j]);
            }
            }
        }
        first = p - nfast;                                    // 0 .. WAVES-1: where the stride lands in the next list
    }
    }
    // ---- every other sounding voice: the general code (the stride simply continues, so the extra voices go to the
    // waves that got one fast voice fewer) ----
    if constexpr (MODE == RENDER_LEAN_HARM || MODE == RENDER_LEAN_ALL) {
        uint32_t lane_late = lane;
        asm volatile("" : "+v"(lane_late));       // defined here, after the loop above: the arrays cannot be built earlier
        build_frames(lane_late);
    }
    if constexpr (!mode_lean_only(MODE)) {
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t ngen = as_const(curS.counts)[4 * c + 1];
        const uint32_t SH_CONST_AS* idx = as_const(curS.gen_idx) + c * 64;
        const uint32_t step = WAVES * nsub;                        // (nsub > 1: the first segment of a general segmented launch)
        uint32_t p = first;
        uint32_t vi_next = p < ngen ? idx[p] : 0u;
        for (; p < ngen; p += step) {
            const uint32_t vi = vi_next;
            vi_next = p + step < ngen ? idx[p + step] : 0u;       // in flight with this voice's record: one round trip less
            const VoiceRegs r = load_record(as_const(curS.launch) + vi);
            general_voice<FPL>(r, curS.fm + vi, B, B.voices + vi, st0, tile0, nfr, i, di, trig, accl, accr);
        }
        first = p - ngen;
    }
    }
    }
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        red[wave][0][j * 64 + lane] = accl[j];
        red[wave][1][j * 64 + lane] = accr[j];
    }
    __syncthreads();
    // each wave finishes 64-frame rows of the tile: rows wave, wave + WAVES, ...
    for (uint32_t row = wave; row < (uint32_t)FPL; row += WAVES) {
        const uint32_t f = row * 64 + lane;
        const uint32_t raw = tile0 + f;
        if (raw < nfr) {
            double l = red[0][0][f], rr = red[0][1][f];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) {
                l += red[w][0][f];
                rr += red[w][1][f];
            }
            const size_t at = (size_t)seg_off + raw;            // launch-relative frame
            if (MODE == RENDER_GENERAL_SEG && to_scratch) {
                B.gen_scratch[(size_t)(grp * nsub + sub) * nfr + raw] = make_double2(l, rr);
            } else if (parts) {
                const uint32_t slot = mode_general(MODE) ? gridDim.y + grp : grp;
                parts[(size_t)slot * nframes + at] = make_double2(l, rr);
            } else {
                if (bus32) bus32[at] = make_float2((float)l, (float)rr);
                if (bus64) bus64[at] = make_double2(l, rr);
                if (pcm16) pcm16[at] = pcm16_frame(l, rr, pcm_scale);
            }
        }
    }
}

// The first segment of a general segmented launch: its nsub slices per group, added in order into the group's general parts.
__global__ __launch_bounds__(256) void k_seg_combine(const double2* __restrict__ scratch, uint32_t nsub, uint32_t n0,
                                                     double2* __restrict__ gen_parts, uint32_t nframes) {
    const uint32_t f = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (f >= n0) return;
    double2 acc = scratch[(size_t)(g * nsub) * n0 + f];
    for (uint32_t k = 1; k < nsub; ++k) {
        const double2 x = scratch[(size_t)(g * nsub + k) * n0 + f];
        acc.x += x.x;
        acc.y += x.y;
    }
    gen_parts[(size_t)g * nframes + f] = acc;
}

__global__ __launch_bounds__(256) void k_bus_combine(const double2* __restrict__ parts, uint32_t ngroups, uint32_t nframes,
                                                     float2* __restrict__ bus32, double2* __restrict__ bus64,
                                                     uint32_t* __restrict__ pcm16, double pcm_scale,
                                                     const uint32_t* __restrict__ gen_valid) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= nframes) return;
    double2 s = parts[i];
    for (uint32_t g = 1; g < ngroups; ++g) {
        const double2 p = parts[(size_t)g * nframes + i];
        s.x += p.x;
        s.y += p.y;
    }
    if (gen_valid) {                                   // split launch: the general kernel's parts, same order as the in-kernel fold
        for (uint32_t g = 0; g < ngroups; ++g) {
            if (gen_valid[g]) {
                const double2 p = parts[(size_t)(ngroups + g) * nframes + i];
                s.x += p.x;
                s.y += p.y;
            }
        }
    }
    if (bus32) bus32[i] = make_float2((float)s.x, (float)s.y);
    if (bus64) bus64[i] = s;
    if (pcm16) pcm16[i] = pcm16_frame(s.x, s.y, pcm_scale);
}

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

// ---- elementwise filters over float64 blocks ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_ew_f64(int op, const double* a, const double* b, size_t n, double p0, double p1,
                                                double* out64, float* out32) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    double v;
    switch (op) {
    case SH_EW_ADD: v = a[i] + b[i]; break;
    case SH_EW_MUL: v = a[i] * b[i]; break;
    case SH_EW_CLIP: { double t = a[i] < p1 ? a[i] : p1; v = t > p0 ? t : p0; } break;    // max(min(v, maximum), minimum)
    case SH_EW_ABS: v = fabs(a[i]); break;
    case SH_EW_COPY: v = a[i]; break;
    case SH_EW_AXPY: { const double e = b[i] * p0; v = a[i] + e; } break;
    default: v = p0; break;
    }
    if (out64) out64[i] = v;
    if (out32) out32[i] = (float)v;
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

// The same scan for many rows at once (the modulators of a bank): blockIdx.y = row; x and out may be the same buffer;
// carry[row] is read as the row's carry-in and replaced by its carry-out, on the device: consecutive blocks chain without a
// host round trip.
__global__ __launch_bounds__(256) void k_scan_rows_tile_sums(const double* __restrict__ x, size_t stride, uint32_t n, uint32_t ntiles,
                                                             double* __restrict__ sums) {
    __shared__ double sh[256];
    const double* xr = x + (size_t)blockIdx.y * stride;
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (base + j < n) s += xr[base + j];
    double total;
    block_exclusive_scan_256(s, sh, total);
    if (threadIdx.x == 0) sums[(size_t)blockIdx.y * (ntiles + 1) + blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_scan_rows_sums(double* __restrict__ sums, uint32_t ntiles, double* __restrict__ carry) {
    __shared__ double sh[256];
    __shared__ double run;
    double* sr = sums + (size_t)blockIdx.x * (ntiles + 1);
    if (threadIdx.x == 0) run = carry[blockIdx.x];
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const double v = (i < ntiles) ? sr[i] : 0.0;
        double total;
        const double ex = block_exclusive_scan_256(v, sh, total);
        const double c = run;
        if (i < ntiles) sr[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) run = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) carry[blockIdx.x] = run;
}

__global__ __launch_bounds__(256) void k_scan_rows_apply(const double* x, size_t stride, uint32_t n, uint32_t ntiles,
                                                         const double* __restrict__ sums, double* out) {
    __shared__ double sh[256];
    const double* xr = x + (size_t)blockIdx.y * stride;
    double* outr = out + (size_t)blockIdx.y * stride;
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    double v[8];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = (base + j < n) ? xr[base + j] : 0.0;
        s += v[j];
    }
    double total;
    double ex = block_exclusive_scan_256(s, sh, total) + sums[(size_t)blockIdx.y * (ntiles + 1) + blockIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < n) outr[base + j] = ex;
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
    // launch records, four sets: launch n reads one while its first workgroups resolve the records of the block expected
    // two launches later (start + 2 * nframes) into another; launch n-1, on the other stream, holds two more
    static constexpr int NSETS = 4;
    VoiceLaunch* d_launch_buf[NSETS] = {};
    VoiceFM*    d_launch_fm_buf[NSETS] = {};
    FastRec*    d_fast_buf[NSETS] = {};
    uint32_t*   d_gen_idx_buf[NSETS] = {};
    uint32_t*   d_counts_buf[NSETS] = {};      // 4 per 64-voice chunk: lean, general, silent, -
    uint32_t*   d_hint = nullptr;
    // This bank's run of pipelined renders (same shape, consecutive blocks), the folds it still owes, and its ring of
    // partial-bus buffers: launch n writes ring slot n % 4, launch n + 2 (same stream) folds it.  Per bank, so that two banks
    // rendering turn by turn (or a bank beside a DistVoiceBank's) each keep their pipeline: only calls that can touch a bus
    // buffer end the runs (sh::flush_pending).
    bool        run_active = false;
    uint64_t    run_next_start = 0;
    uint32_t    run_nframes = 0, run_groups = 0, run_tile = 0, run_count = 0;
    sh::PendingCombine pending[2];
    int         npending = 0;
    void*       parts_buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t      parts_bytes[4] = {0, 0, 0, 0};
    LaunchSet   seg_set[2] = {};           // record sets of a segmented transition launch (one per stream), seg_cap[k] segments each
    uint32_t    seg_cap[2] = {0, 0};
    void*       seg_scratch[2] = {nullptr, nullptr};   // ... and the slices of its first segment's general parts (BankPtrs::gen_scratch)
    size_t      seg_scratch_bytes[2] = {0, 0};
    // modulation rows (sh_bank_set_rows / sh_bank_render_rows)
    int32_t*    d_fm_row = nullptr;
    int32_t*    d_pwm_row = nullptr;
    int32_t     fm_row_max = -1;
    bool        needs_rows = false;        // some voice reads a row: SH_FM_BUFFER, SH_BUFFER, or a pwm row was set
    const double* launch_rows = nullptr;   // set for the duration of one sh_bank_render_rows call
    size_t      launch_row_stride = 0;
    // record sets of a segmented materialisation (sh_bank_generate over long rows), gen_segs of them, allocated on demand
    LaunchSet   gen_set = {};
    uint32_t    gen_segs = 0;
    double2*    d_seg_rot = nullptr;       // (cos, sin)(64*dt) per table piece
    double2*    d_lfo_rot = nullptr;       // (cos, sin)(64*lfo_d) per voice
    VoiceLaunch* d_launch = nullptr;       // the set the next kernel reads
    VoiceFM*    d_launch_fm = nullptr;
    int         cur = 0;                   // the set the last launch read
    int         last_target = -1;          // the set the last render launch is filling (-1: none)
    struct Spec { bool valid = false; uint64_t start = 0; uint32_t nframes = 0; } spec[NSETS];   // what each set holds (or will)
    void        void_specs() { for (auto& q : spec) q.valid = false; last_target = -1; }
    uint32_t    lean_candidates = 0;      // voices that can take the lean loop in some launch (static properties)
    uint32_t    lean_fm_candidates = 0;   // ... of them other than polynomial Harmonics (FM Sine, plain waveforms)
    uint32_t    last_groups = 0;          // voice groups of the last sh_bank_render launch (sh_bank_launch_stats)
    // When can a launch hold NO general voice (so that a split launch needs no general-lists kernel)?  Conservative, from
    // static properties: every voice is a lean candidate, every envelope is on its sustain piece for the
    // whole launch, and no phase-table piece shorter than the launch ends after its start (a launch then crosses at most one
    // piece end per voice).  short_piece_end[k] = the largest end of any piece shorter than 2^k samples.
    bool        all_lean = false;          // every voice is a lean candidate (of any lean kind)
    uint64_t    env_flat_from = 0, env_flat_until = ~0ull;
    std::vector<uint64_t> env_corners;     // the distinct attack / decay / sustain / release ends of the voices, sorted (empty when there are many)
    uint64_t    short_piece_end[34] = {};
    bool        no_general_voice(uint64_t start, uint32_t nframes) const {
        if (!all_lean || start < env_flat_from || start + nframes > env_flat_until) return false;
        int k = 0;
        while ((1ull << k) < (uint64_t)nframes) ++k;
        return short_piece_end[k] <= start;
    }
    float2*     d_gains = nullptr;
    uint32_t    nsegs = 0, ncoefs = 0, npartials = 0;
    std::vector<sh_voice> h_voices;    // for validation of per-call arguments
};

static LaunchSet launch_set(const sh_bank* b, int k) {
    LaunchSet s;
    s.launch = b->d_launch_buf[k];
    s.fm = b->d_launch_fm_buf[k];
    s.fast = b->d_fast_buf[k];
    s.gen_idx = b->d_gen_idx_buf[k];
    s.counts = b->d_counts_buf[k];
    return s;
}

static BankPtrs ptrs(const sh_bank* b) {
    BankPtrs p;
    p.voices = b->d_voices;
    p.segs = b->d_segs;
    p.coefs = b->d_coefs;
    p.partials = b->d_partials;
    p.hint = b->d_hint;
    p.seg_rot = b->d_seg_rot;
    p.lfo_rot = b->d_lfo_rot;
    p.rows = b->launch_rows;
    p.row_stride = b->launch_row_stride;
    p.fm_row = b->d_fm_row;
    p.pwm_row = b->d_pwm_row;
    p.nseg = 0;
    for (int k = 0; k <= SEG_MAX; ++k) p.seg_first[k] = 0;
    p.gen_sub = 1;
    p.gen_scratch = nullptr;
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

static std::vector<sh_bank*>& live_banks() {
    static std::vector<sh_bank*> v;
    return v;
}

// Fold what one bank still owes (oldest first: a bus used for two blocks ends up holding the later one) on the main stream and
// end its run.  The caller has joined stream2 into the main stream.
static int fold_bank(sh_bank* b) {
    sh::State& S = sh::state();
    b->run_active = false;
    b->run_count = 0;
    const int n = b->npending;
    b->npending = 0;
    S.pending_total -= n;
    for (int k = 0; k < n; ++k) {
        const sh::PendingCombine& pc = b->pending[k];
        hipLaunchKernelGGL(k_bus_combine, sh::grid1d(pc.nframes, 256), dim3(256), 0, S.stream,
                           (const double2*)pc.parts, pc.groups, pc.nframes, (float2*)pc.o32, (double2*)pc.o64,
                           (uint32_t*)pc.o16, pc.scale, (const uint32_t*)pc.gen_valid);
        SH_CHECK_LAUNCH("k_bus_combine");
    }
    return SH_OK;
}

static int join_aux() {
    sh::State& S = sh::state();
    if (S.aux_busy) {
        S.aux_busy = false;
        SH_HIP(hipStreamWaitEvent(S.stream, S.ev_aux, 0));
    }
    return SH_OK;
}

namespace sh {
int flush_pending() {
    int rc = join_aux();
    if (rc) return rc;
    for (sh_bank* b : live_banks()) {                     // every bank's run ends here: the next one starts on `stream`
        rc = fold_bank(b);
        if (rc) return rc;
    }
    return SH_OK;
}

void free_render_buffers() {
    for (sh_bank* b : live_banks())
        for (int k = 0; k < 4; ++k)
            if (b->parts_buf[k]) { (void)hipFree(b->parts_buf[k]); b->parts_buf[k] = nullptr; b->parts_bytes[k] = 0; }
}

int bus_finalize_on(hipStream_t st, const double* in, size_t nvalues, float* out) {
    if (!nvalues) return SH_OK;
    hipLaunchKernelGGL(k_bus_finalize, sh::grid1d(nvalues, 256), dim3(256), 0, st, in, nvalues, out);
    SH_CHECK_LAUNCH("k_bus_finalize");
    return SH_OK;
}
}  // namespace sh

extern "C" {

int sh_bank_create(const sh_voice* voices, uint32_t nvoices, const sh_segment* segs, uint32_t nsegs,
                   const double* coefs, uint32_t ncoefs, const sh_partial* partials, uint32_t npartials,
                   sh_bank** out) {
    SH_REQUIRE_INIT();
    if (!out || !voices || nvoices == 0) return sh::set_error(SH_ERR_INVALID, "sh_bank_create: no voices");
    if (!segs || nsegs == 0) return sh::set_error(SH_ERR_INVALID, "sh_bank_create: no phase tables");
    for (uint32_t i = 0; i < nvoices; ++i) {
        const sh_voice& v = voices[i];
        if (v.kind < SH_SINE || v.kind > SH_BUFFER)
            return sh::set_error(SH_ERR_INVALID, "voice %u: unknown kind %d", i, v.kind);
        if (v.kind == SH_NOISE && v.noise_hold == 0)
            return sh::set_error(SH_ERR_INVALID, "voice %u: noise_hold must be >= 1", i);
        if ((v.kind == SH_NOISE || v.kind == SH_LINEAR || v.kind == SH_BUFFER) && v.fm_mode != SH_FM_NONE)
            return sh::set_error(SH_ERR_INVALID, "voice %u: kind %d has no FM form", i, v.kind);
        if (v.fm_mode < SH_FM_NONE || v.fm_mode > SH_FM_BUFFER)
            return sh::set_error(SH_ERR_INVALID, "voice %u: unknown fm_mode %d", i, v.fm_mode);
        uint32_t off = v.fm_mode ? v.time_seg_offset : v.seg_offset;
        uint32_t cnt = v.fm_mode ? v.time_seg_count : v.seg_count;
        if (cnt == 0 || off > nsegs || cnt > nsegs - off)
            return sh::set_error(SH_ERR_INVALID, "voice %u: phase table [%u,+%u) outside %u pieces", i, off, cnt, nsegs);
        if (segs[off].n0 != 0) return sh::set_error(SH_ERR_INVALID, "voice %u: phase table does not start at sample 0", i);
        if (v.kind == SH_HARMONICS) {
            if (v.harm_dense < 0 || v.harm_dense > 2) return sh::set_error(SH_ERR_INVALID, "voice %u: harm_dense %d not in {0,1,2}", i, v.harm_dense);
            if (v.harm_dense == 2 && v.harm_count != 16) return sh::set_error(SH_ERR_INVALID, "voice %u: polynomial form needs 16 coefficients", i);
            uint32_t lim = v.harm_dense ? ncoefs : npartials;
            if (v.harm_offset > lim || v.harm_count > lim - v.harm_offset)
                return sh::set_error(SH_ERR_INVALID, "voice %u: harmonics [%u,+%u) outside table of %u", i, v.harm_offset, v.harm_count, lim);
            if (v.harm_dense == 1 && (v.harm_count & 7))
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
    for (uint32_t i = 0; i < nvoices; ++i)
        if (voices[i].bias == 0.0 && !voices[i].flip &&
            ((voices[i].kind == SH_HARMONICS && voices[i].harm_dense == 2 && voices[i].fm_mode == SH_FM_NONE) ||
             (voices[i].kind == SH_SINE && voices[i].fm_mode == SH_FM_SINE) ||
             (voices[i].fm_mode == SH_FM_NONE && (voices[i].kind == SH_SINE || voices[i].kind == SH_SAWTOOTH || voices[i].kind == SH_SQUARE ||
                                                  voices[i].kind == SH_TRIANGLE || voices[i].kind == SH_PULSE)))) {
            b->lean_candidates += 1;
            if (voices[i].kind != SH_HARMONICS) b->lean_fm_candidates += 1;       // needs the kernel with all record kinds
        }
    {
        b->all_lean = b->lean_candidates == nvoices;
        for (uint32_t i = 0; i < nvoices && b->all_lean; ++i) {
            const sh_voice& v = voices[i];
            if (v.env.enabled) {
                if (v.env.n_decay_end > b->env_flat_from) b->env_flat_from = v.env.n_decay_end;
                if (v.env.n_attack_end > b->env_flat_from) b->env_flat_from = v.env.n_attack_end;
                if (v.env.n_sustain_end < b->env_flat_until) b->env_flat_until = v.env.n_sustain_end;
                if (b->env_corners.size() <= 16) {
                    for (uint64_t c : {v.env.n_attack_end, v.env.n_decay_end, v.env.n_sustain_end, v.env.n_release_end, v.env.n_release_end + 1})
                        if (c && std::find(b->env_corners.begin(), b->env_corners.end(), c) == b->env_corners.end()) b->env_corners.push_back(c);
                }
            }
            const uint32_t toff = v.fm_mode ? v.time_seg_offset : v.seg_offset, tcnt = v.fm_mode ? v.time_seg_count : v.seg_count;
            for (uint32_t k = 0; k + 1 < tcnt; ++k) {
                const uint64_t a = segs[toff + k].n0, e = segs[toff + k + 1].n0;
                const uint64_t len = e - a;
                for (int q = 0; q < 34; ++q)
                    if (len < (1ull << q) && e > b->short_piece_end[q]) b->short_piece_end[q] = e;
            }
        }
    }
    if (b->env_corners.size() > 16) b->env_corners.clear();       // voices with envelopes of their own: cut where they are all flat only
    std::sort(b->env_corners.begin(), b->env_corners.end());
    std::vector<float2> gains(nvoices);
    for (uint32_t i = 0; i < nvoices; ++i) gains[i] = make_float2(voices[i].gain_l, voices[i].gain_r);
    int rc = upload_array(&b->d_voices, voices, nvoices, st);
    if (!rc) rc = upload_array(&b->d_segs, segs, nsegs, st);
    if (!rc) rc = upload_array(&b->d_coefs, coefs, ncoefs, st);
    if (!rc) rc = upload_array(&b->d_partials, partials, npartials, st);
    if (!rc) rc = upload_array(&b->d_gains, gains.data(), nvoices, st);
    // rotations by 64 samples: constant per table piece / per voice, so they are computed here once (libm) instead of
    // in every launch's prepare step
    std::vector<double2> seg_rot(nsegs), lfo_rot(nvoices);
    for (uint32_t i = 0; i < nsegs; ++i) seg_rot[i] = make_double2(cos(64.0 * segs[i].dt), sin(64.0 * segs[i].dt));
    for (uint32_t i = 0; i < nvoices; ++i) lfo_rot[i] = make_double2(cos(64.0 * voices[i].lfo_d), sin(64.0 * voices[i].lfo_d));
    if (!rc) rc = upload_array(&b->d_seg_rot, seg_rot.data(), nsegs, st);
    if (!rc) rc = upload_array(&b->d_lfo_rot, lfo_rot.data(), nvoices, st);
    if (!rc) {
        hipError_t e = hipSuccess;
        for (int k = 0; k < sh_bank::NSETS && e == hipSuccess; ++k) {
            e = hipMalloc((void**)&b->d_launch_buf[k], sizeof(VoiceLaunch) * nvoices);
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_launch_fm_buf[k], sizeof(VoiceFM) * nvoices);
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_fast_buf[k], sizeof(FastRec) * nvoices);
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_gen_idx_buf[k], sizeof(uint32_t) * nvoices);
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_counts_buf[k], sizeof(uint32_t) * 4 * ((nvoices + 63) / 64));
        }
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_hint, sizeof(uint32_t) * nvoices);
        if (e == hipSuccess) e = hipMemsetAsync(b->d_hint, 0, sizeof(uint32_t) * nvoices, st);
        if (e != hipSuccess) rc = sh::hip_error(e, "hipMalloc(launch records)");
        b->d_launch = b->d_launch_buf[0];
        b->d_launch_fm = b->d_launch_fm_buf[0];
    }
    if (!rc) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = sh::hip_error(e, "upload voice table");
    }
    if (rc) {
        sh_bank_destroy(b);
        return rc;
    }
    live_banks().push_back(b);
    *out = b;
    return SH_OK;
}

int sh_bank_destroy(sh_bank* b) {
    if (!b) return SH_OK;
    SH_API_LOCK();
    if (sh::state().initialized) {
        if (sh::has_pending()) (void)sh::flush_pending();
        (void)hipStreamSynchronize(sh::state().stream);
        for (int k = 0; k < 4; ++k) if (b->parts_buf[k]) (void)hipFree(b->parts_buf[k]);
        for (int k = 0; k < 2; ++k) {
            LaunchSet& g = b->seg_set[k];
            if (g.launch) { (void)hipFree(g.launch); (void)hipFree(g.fm); (void)hipFree(g.fast); (void)hipFree(g.gen_idx); (void)hipFree(g.counts); }
            if (b->seg_scratch[k]) (void)hipFree(b->seg_scratch[k]);
        }
        if (b->d_voices) (void)hipFree(b->d_voices);
        if (b->d_segs) (void)hipFree(b->d_segs);
        if (b->d_coefs) (void)hipFree(b->d_coefs);
        if (b->d_partials) (void)hipFree(b->d_partials);
        for (int k = 0; k < sh_bank::NSETS; ++k) {
            if (b->d_launch_buf[k]) (void)hipFree(b->d_launch_buf[k]);
            if (b->d_launch_fm_buf[k]) (void)hipFree(b->d_launch_fm_buf[k]);
            if (b->d_fast_buf[k]) (void)hipFree(b->d_fast_buf[k]);
            if (b->d_gen_idx_buf[k]) (void)hipFree(b->d_gen_idx_buf[k]);
            if (b->d_counts_buf[k]) (void)hipFree(b->d_counts_buf[k]);
        }
        if (b->d_gains) (void)hipFree(b->d_gains);
        if (b->d_hint) (void)hipFree(b->d_hint);
        if (b->d_fm_row) (void)hipFree(b->d_fm_row);
        if (b->d_pwm_row) (void)hipFree(b->d_pwm_row);
        if (b->gen_set.launch) (void)hipFree(b->gen_set.launch);
        if (b->gen_set.fm) (void)hipFree(b->gen_set.fm);
        if (b->gen_set.fast) (void)hipFree(b->gen_set.fast);
        if (b->gen_set.gen_idx) (void)hipFree(b->gen_set.gen_idx);
        if (b->gen_set.counts) (void)hipFree(b->gen_set.counts);
        if (b->d_seg_rot) (void)hipFree(b->d_seg_rot);
        if (b->d_lfo_rot) (void)hipFree(b->d_lfo_rot);
    }
    {
        auto& v = live_banks();
        for (size_t k = 0; k < v.size(); ++k)
            if (v[k] == b) { v.erase(v.begin() + (long)k); break; }
    }
    delete b;
    return SH_OK;
}

uint32_t sh_bank_nvoices(const sh_bank* b) { return b ? b->nvoices : 0; }

int sh_bank_launch_stats(sh_bank* b, uint32_t* nfast, uint32_t* ngeneral) {
    SH_REQUIRE_INIT();
    if (!b) return sh::set_error(SH_ERR_INVALID, "sh_bank_launch_stats: NULL bank");
    if (!b->last_groups) return sh::set_error(SH_ERR_INVALID, "sh_bank_launch_stats: no sh_bank_render call yet");
    const uint32_t nchunks = (b->nvoices + 63) / 64;
    std::vector<uint32_t> c(4 * (size_t)nchunks);
    hipStream_t st = sh::state().stream;
    SH_HIP(hipMemcpyAsync(c.data(), b->d_counts_buf[b->cur], c.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SH_HIP(hipStreamSynchronize(st));
    uint32_t f = 0, g = 0;
    for (uint32_t k = 0; k < nchunks; ++k) { f += c[4 * k]; g += c[4 * k + 1]; }
    if (nfast) *nfast = f;
    if (ngeneral) *ngeneral = g;
    return SH_OK;
}

static const shm::sc_pair* trig_table() { return (const shm::sc_pair*)sh::state().trig; }

static int prepare(sh_bank* b, uint32_t first, uint32_t count, uint64_t start, uint32_t nframes) {
    // single-voice path (sh_osc_render): records go to slot 0 of the current set; any speculation is void
    b->void_specs();
    hipLaunchKernelGGL(k_prepare, sh::grid1d(count, 64), dim3(64), 0, sh::state().stream,
                       ptrs(b), first, count, start, nframes, b->d_launch, b->d_launch_fm);
    SH_CHECK_LAUNCH("k_prepare");
    return SH_OK;
}

// Whole-bank launches: use the records the previous render kernel prepared if the caller asks for the block
// that was predicted (sequential streaming), else run k_prepare.
// `launch_stream`: the stream the consuming kernel goes to.  `in_run`: the previous render launch of this bank may still
// be executing on the other stream -- its set (b->cur) and the set it is filling (b->last_target) must not be touched.
static int acquire_records(sh_bank* b, uint64_t start, uint32_t nframes, hipStream_t launch_stream, bool in_run) {
    sh::State& S = sh::state();
    for (int k = 0; k < sh_bank::NSETS; ++k) {
        if (b->spec[k].valid && b->spec[k].start == start && b->spec[k].nframes == nframes) {
            b->spec[k].valid = false;                    // consumed: the set is this launch's from here on
            b->cur = k;
            b->d_launch = b->d_launch_buf[k];
            b->d_launch_fm = b->d_launch_fm_buf[k];
            return SH_OK;
        }
    }
    int k = -1;
    for (int c = 0; c < sh_bank::NSETS && k < 0; ++c)   // a free set, preferably one that holds no resolved block
        if (!(in_run && (c == b->cur || c == b->last_target)) && !b->spec[c].valid) k = c;
    if (k < 0) {
        // every other set holds a resolved block (not reachable with the call patterns that keep a run alive; kept safe
        // anyway): an older launch on stream2 may still be filling the one taken here, so order the prepare after it
        if (in_run && S.aux_busy) SH_HIP(hipStreamWaitEvent(S.stream, S.ev_aux, 0));
        for (int c = 0; c < sh_bank::NSETS && k < 0; ++c)
            if (!(in_run && (c == b->cur || c == b->last_target))) k = c;
    }
    b->spec[k].valid = false;
    hipLaunchKernelGGL(k_prepare_chunks, sh::grid1d(b->nvoices, 64), dim3(64), 0, S.stream, ptrs(b), launch_set(b, k),
                       b->nvoices, start, nframes);
    SH_CHECK_LAUNCH("k_prepare_chunks");
    if (launch_stream != S.stream) {
        SH_HIP(hipEventRecord(S.ev_prep, S.stream));
        SH_HIP(hipStreamWaitEvent(launch_stream, S.ev_prep, 0));
    }
    b->cur = k;
    b->d_launch = b->d_launch_buf[k];
    b->d_launch_fm = b->d_launch_fm_buf[k];
    return SH_OK;
}

// The cuts of a transition launch / of the head of a materialised row (see RENDER_LEAN_HARM_SEG): with `corners`, the envelope
// corners the voices share (sloped records: a line of any slope is lean, a corner is not) -- without, or when the voices have
// envelopes of their own, the frame from which all of them are flat and the first sustain end -- and, between those, doubling
// positions (at most one piece end of the phase sum per voice in [pos, 2 pos)); no segment longer than max_len.
// seg_first[0 .. n] = launch-relative segment starts; returns n; the segments cover seg_first[n] <= nframes frames (fewer than
// nframes only when SEG_MAX segments do not reach the end).
static uint32_t plan_segments(const sh_bank* b, uint64_t start, uint32_t nframes, uint64_t T, uint64_t max_len, bool corners,
                              uint32_t* seg_first) {
    const uint64_t end = start + nframes;
    uint64_t cuts[SEG_MAX + 2];
    uint32_t nc = 0;
    uint64_t pos = start;
    cuts[nc++] = pos;
    const uint64_t flat = b->env_flat_from, rel = b->env_flat_until;       // last decay end, first sustain end
    const bool shared = corners && !b->env_corners.empty();
    if (!shared && pos < flat && flat < end && flat - pos <= max_len) { pos = flat; cuts[nc++] = pos; }
    static long seg_min = -1;
    if (seg_min < 0) { const char* e = getenv("SYNTHHIP_SEG_MIN"); seg_min = e ? atol(e) : 0; }
    while (pos < end && nc <= SEG_MAX) {
        uint64_t next = pos < T ? T : 2 * pos;
        if (pos == start && start < (uint64_t)seg_min && (uint64_t)seg_min < end) next = (uint64_t)seg_min;   // the dense first segment
        else if (shared) {
            for (uint64_t c : b->env_corners)
                if (c > pos && c < next) { next = c; break; }
        } else if (pos < rel && rel < next) {
            next = rel;
        }
        if (next - pos > max_len) next = pos + max_len;
        if (next >= end || (end - next <= next / 64 && !shared && end - pos <= max_len)) next = end;   // (a very short rest joins the last segment)
        pos = next;
        cuts[nc++] = pos;
    }
    for (uint32_t k = 0; k < nc; ++k) seg_first[k] = (uint32_t)(cuts[k] - start);
    return nc - 1;
}

static bool speculation_enabled() {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("SYNTHHIP_NO_SPECULATION"); enabled = (e && e[0] == '1') ? 0 : 1; }
    return enabled != 0;
}

static bool overlap_enabled() {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("SYNTHHIP_NO_OVERLAP"); enabled = (e && e[0] == '1') ? 0 : 1; }
    return enabled != 0;
}

int sh_osc_render(sh_bank* bank, uint32_t voice, const sh_buf* fm_cumsum, const sh_buf* pwm,
                  uint64_t start, uint32_t n, float* out_host, sh_buf* out_f32, size_t out_off, sh_buf* out_f64) {
    SH_REQUIRE_INIT();
    if (!bank || voice >= bank->nvoices) return sh::set_error(SH_ERR_INVALID, "sh_osc_render: bad bank/voice");
    if (!out_host && !out_f32 && !out_f64) return sh::set_error(SH_ERR_INVALID, "sh_osc_render: no destination");
    if (n == 0) return SH_OK;
    if (n > 0xFFFF0000u) return sh::set_error(SH_ERR_INVALID, "sh_osc_render: at most 2^32 - 65536 samples per call");
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
    int rc = prepare(bank, voice, 1, start, n);
    if (rc) return rc;
    hipLaunchKernelGGL(k_generate<1>, dim3(sh::div_up(n, 256), 1), dim3(256), 0, sh::state().stream,
                       ptrs(bank), trig_table(), voice, 1u, 1u, bank->d_launch, bank->d_launch_fm, start, n,
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
    if (b->launch_rows) return SH_OK;                      // sh_bank_render_rows supplies them
    for (uint32_t i = 0; i < b->nvoices; ++i)
        if (b->h_voices[i].fm_mode == SH_FM_BUFFER || b->h_voices[i].kind == SH_BUFFER)
            return sh::set_error(SH_ERR_INVALID, "%s: voice %u reads a modulation / sample row; render the bank with sh_bank_render_rows", who, i);
    if (b->needs_rows) return sh::set_error(SH_ERR_INVALID, "%s: the bank has modulation rows (sh_bank_set_rows); render it with sh_bank_render_rows", who);
    return SH_OK;
}

int sh_bank_generate(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* voices_out, size_t stride) {
    SH_REQUIRE_INIT();
    if (!b || !voices_out) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate: NULL argument");
    if (nframes == 0) return SH_OK;
    if (nframes > 0xFFFF0000u) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate: at most 2^32 - 65536 frames per call");
    if (stride < nframes) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate: stride < nframes");
    if (voices_out->bytes / 4 < (size_t)(b->nvoices - 1) * stride + nframes)
        return sh::set_error(SH_ERR_INVALID, "sh_bank_generate: output buffer too small");
    int rc = bank_check_plain(b, "sh_bank_generate");
    if (rc) return rc;
    // frames per lane: 4 for long rows (one sin/cos lookup + three rotations per voice, as in k_bank_render), else 2 / 1
    const int fpl = nframes >= 8192 ? 4 : (nframes >= 2048 ? 2 : 1);
    float* o = (float*)voices_out->ptr;
    if (fpl == 4 && b->lean_candidates != 0 && b->lean_fm_candidates == 0) {
        // Long rows, every lean candidate a polynomial Harmonics voice: the lean records by the recurrence kernel at sixteen frames
        // per lane, then the general and silent lists -- unless the segment provably has none.  Rows longer than a segment get
        // one record set per segment, all resolved by ONE prepare launch.
        // frames per lane of the lean kernel: sixteen on long rows (1024 x 480 000: 496 us, eight: 508), fewer when that leaves
        // the chip short of workgroups (1024 x 48 000 at sixteen: 12 x 16 = 192 workgroups of four 1024-frame tiles)
        int lf = 16;
        while (lf > 4 && (uint64_t)sh::div_up(nframes, 256 * lf) * sh::div_up(b->nvoices, 64) < 512) lf /= 2;
        {
            static int forced = -1;
            if (forced < 0) { const char* e = getenv("SYNTHHIP_GEN_LF"); forced = e ? atoi(e) : 0; }
            if (forced == 4 || forced == 8 || forced == 16) lf = forced;
        }
        const int LF = lf;
#define SH_GEN_LEAN(GRID_, ...) do { \
            if (LF == 16) hipLaunchKernelGGL(k_generate_lean_harm<16>, GRID_, dim3(256), 0, st, __VA_ARGS__); \
            else if (LF == 8) hipLaunchKernelGGL(k_generate_lean_harm<8>, GRID_, dim3(256), 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL(k_generate_lean_harm<4>, GRID_, dim3(256), 0, st, __VA_ARGS__); } while (0)
        constexpr uint32_t SEG = 65536;                      // frames per segment (a multiple of the 1024-frame tile)
        hipStream_t st = sh::state().stream;
        const uint32_t nchunks = sh::div_up(b->nvoices, 64);
        // Rows that start with the notes (attack, decay, a dozen binades of the phase sum in the first 65 536 frames): the head
        // is cut like a transition launch of the render path (plan_segments: where the envelopes are flat, then doubling
        // positions) so that its voices stay lean -- one prepare launch, one lean launch, ONE lists launch over all segments with
        // the general voices of a chunk dealt to eight workgroups; the rest of the row follows with equal segments.
        static int no_seg = -1;
        if (no_seg < 0) { const char* e = getenv("SYNTHHIP_NO_SEG"); no_seg = (e && e[0] == '1') ? 1 : 0; }
        if (start < SEG && b->all_lean && !no_seg && !b->no_general_voice(start, nframes < SEG ? nframes : SEG)) {
            uint32_t cut[SEG_MAX + 1];
            SegTab tab, ltab;
            tab.n = plan_segments(b, start, nframes, 64 * LF, SEG, false, cut);
            if (tab.n >= 2) {
                const uint32_t head = cut[tab.n];                        // frames the segments cover (all of them, or what SEG_MAX cuts reach)
                ltab.n = 0;
                for (uint32_t k = 0; k < tab.n; ++k) {
                    tab.first[k] = cut[k]; tab.len[k] = cut[k + 1] - cut[k]; tab.set[k] = k;
                    if (!b->no_general_voice(start + cut[k], cut[k + 1] - cut[k])) {     // (the lists launch: where general or silent voices can be)
                        ltab.first[ltab.n] = cut[k]; ltab.len[ltab.n] = cut[k + 1] - cut[k]; ltab.set[ltab.n] = k;
                        ++ltab.n;
                    }
                }
                if (b->gen_segs < tab.n) {
                    SH_HIP(hipStreamSynchronize(st));
                    LaunchSet& g = b->gen_set;
                    if (g.launch) { (void)hipFree(g.launch); (void)hipFree(g.fm); (void)hipFree(g.fast); (void)hipFree(g.gen_idx); (void)hipFree(g.counts); g = LaunchSet(); }
                    b->gen_segs = 0;
                    const size_t cap = tab.n;
                    SH_HIP(hipMalloc((void**)&g.launch, sizeof(VoiceLaunch) * cap * b->nvoices));
                    SH_HIP(hipMalloc((void**)&g.fm, sizeof(VoiceFM) * cap * b->nvoices));
                    SH_HIP(hipMalloc((void**)&g.fast, sizeof(FastRec) * cap * b->nvoices));
                    SH_HIP(hipMalloc((void**)&g.gen_idx, sizeof(uint32_t) * cap * b->nvoices));
                    SH_HIP(hipMalloc((void**)&g.counts, sizeof(uint32_t) * 4 * cap * nchunks));
                    b->gen_segs = tab.n;
                }
                const LaunchSet base = b->gen_set;
                BankPtrs P = ptrs(b);
                P.nseg = tab.n;
                uint32_t tiles = 0, list_groups = 0;
                for (uint32_t k = 0; k <= tab.n; ++k) P.seg_first[k] = cut[k];
                for (uint32_t k = 0; k < tab.n; ++k) tiles += sh::div_up(tab.len[k], 64 * LF);
                for (uint32_t k = 0; k < ltab.n; ++k) list_groups += sh::div_up(ltab.len[k], 4 * 64 * 4);
                hipLaunchKernelGGL(k_prepare_segments_var<false>, dim3(nchunks, tab.n), dim3(64), 0, st, P, base, b->nvoices, start);
                SH_CHECK_LAUNCH("k_prepare_segments_var");
                SH_GEN_LEAN(dim3(sh::div_up(tiles, 4), nchunks), trig_table(), base, b->nvoices, head, SEG, o, stride, tab);
                SH_CHECK_LAUNCH("k_generate_lean_harm");
                constexpr uint32_t VSPLIT = 8;
                if (ltab.n) {
                    hipLaunchKernelGGL((k_generate_lists<4, false>), dim3(list_groups, nchunks * VSPLIT), dim3(256), 0, st,
                                       ptrs(b), trig_table(), b->nvoices, base, start, head, o, stride, ltab, VSPLIT);
                    SH_CHECK_LAUNCH("k_generate_lists");
                }
                if (head == nframes) return SH_OK;
                sh_buf rest{(char*)voices_out->ptr + (size_t)head * 4, voices_out->bytes - (size_t)head * 4, false, 0};
                return sh_bank_generate(b, start + head, nframes - head, &rest, stride);
            }
        }
        const uint32_t nseg = sh::div_up(nframes, SEG);
        LaunchSet base;
        if (nseg == 1) {
            rc = acquire_records(b, start, nframes, st, false);
            if (rc) return rc;
            base = launch_set(b, b->cur);
        } else {
            if (b->gen_segs < nseg) {
                SH_HIP(hipStreamSynchronize(st));
                LaunchSet& g = b->gen_set;
                if (g.launch) { (void)hipFree(g.launch); (void)hipFree(g.fm); (void)hipFree(g.fast); (void)hipFree(g.gen_idx); (void)hipFree(g.counts); g = LaunchSet(); }
                b->gen_segs = 0;
                SH_HIP(hipMalloc((void**)&g.launch, sizeof(VoiceLaunch) * (size_t)nseg * b->nvoices));
                SH_HIP(hipMalloc((void**)&g.fm, sizeof(VoiceFM) * (size_t)nseg * b->nvoices));
                SH_HIP(hipMalloc((void**)&g.fast, sizeof(FastRec) * (size_t)nseg * b->nvoices));
                SH_HIP(hipMalloc((void**)&g.gen_idx, sizeof(uint32_t) * (size_t)nseg * b->nvoices));
                SH_HIP(hipMalloc((void**)&g.counts, sizeof(uint32_t) * 4 * (size_t)nseg * nchunks));
                b->gen_segs = nseg;
            }
            base = b->gen_set;
            hipLaunchKernelGGL(k_prepare_segments, dim3(nchunks, nseg), dim3(64), 0, st, ptrs(b), base, b->nvoices, start, nframes, SEG);
            SH_CHECK_LAUNCH("k_prepare_segments");
        }
        SegTab none;
        none.n = 0;
        SH_GEN_LEAN(dim3(sh::div_up(nframes, 256 * LF), nchunks), trig_table(), base, b->nvoices, nframes,
                    nseg == 1 ? (nframes + 1023u) / 1024u * 1024u : SEG, o, stride, none);
#undef SH_GEN_LEAN
        SH_CHECK_LAUNCH("k_generate_lean_harm");
        for (uint32_t sg = 0; sg < nseg; ++sg) {
            const uint32_t first = sg * SEG, n = nframes - first < SEG ? nframes - first : SEG;
            if (b->no_general_voice(start + first, n)) continue;
            LaunchSet cur = base;
            if (nseg > 1) {
                cur.launch += (size_t)sg * b->nvoices; cur.fm += (size_t)sg * b->nvoices; cur.fast += (size_t)sg * b->nvoices;
                cur.gen_idx += (size_t)sg * b->nvoices; cur.counts += (size_t)sg * 4 * nchunks;
            }
            hipLaunchKernelGGL((k_generate_lists<4, false>), dim3(sh::div_up(n, 1024), nchunks), dim3(256), 0, st,
                               ptrs(b), trig_table(), b->nvoices, cur, start + first, n, o + first, stride);
            SH_CHECK_LAUNCH("k_generate_lists");
        }
        return SH_OK;
    }
    rc = acquire_records(b, start, nframes, sh::state().stream, false);
    if (rc) return rc;
    const uint32_t tile_groups = sh::div_up(nframes, 256 * fpl);
    // voices per block: as many as keeps >= ~4096 blocks in flight (and gridDim.y <= 65535)
    uint32_t vpg = 1;
    while (vpg < 64 && (uint64_t)tile_groups * ((b->nvoices + 2 * vpg - 1) / (2 * vpg)) >= 4096) vpg *= 2;
    while ((b->nvoices + vpg - 1) / vpg > 65535) vpg *= 2;
    const uint32_t groups = (b->nvoices + vpg - 1) / vpg;
#define SH_GEN(F_) hipLaunchKernelGGL(k_generate<F_>, dim3(tile_groups, groups), dim3(256), 0, sh::state().stream,            \
                                      ptrs(b), trig_table(), 0u, b->nvoices, vpg, b->d_launch, b->d_launch_fm, start, nframes, \
                                      (const double*)nullptr, (const double*)nullptr, o, (double*)nullptr, stride)
    if (fpl == 4 && b->lean_candidates != 0) {
        // long rows of a bank with lean candidates: one workgroup column per 64-voice chunk, walking the launch's lists
        hipLaunchKernelGGL((k_generate_lists<4, true>), dim3(tile_groups, sh::div_up(b->nvoices, 64)), dim3(256), 0, sh::state().stream,
                           ptrs(b), trig_table(), b->nvoices, launch_set(b, b->cur), start, nframes, o, stride);
    } else if (fpl == 4) SH_GEN(4); else if (fpl == 2) SH_GEN(2); else SH_GEN(1);
#undef SH_GEN
    SH_CHECK_LAUNCH("k_generate");
    return SH_OK;
}

}  // extern "C"

// the render behind sh_bank_render (float32 / float64 bus) and sh_bank_render_pcm (int16 PCM straight from the fold)
static int bank_render(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* bus_f32, sh_buf* bus_f64, sh_buf* pcm_i16, double pcm_scale) {
    SH_REQUIRE_INIT_KEEP_PENDING();          // a pending fold of the previous render is taken over by this launch (below)
    const bool no_out = !bus_f32 && !bus_f64 && !pcm_i16;
    if (!b || nframes == 0 || no_out) {
        int rcp = sh::flush_pending();
        if (rcp) return rcp;
    }
    if (!b || no_out) return sh::set_error(SH_ERR_INVALID, "sh_bank_render: NULL argument");
    if (nframes == 0) return SH_OK;
    if (bus_f32 && bus_f32->bytes < (size_t)nframes * 8) return sh::set_error(SH_ERR_INVALID, "sh_bank_render: bus_f32 too small");
    if (bus_f64 && bus_f64->bytes < (size_t)nframes * 16) return sh::set_error(SH_ERR_INVALID, "sh_bank_render: bus_f64 too small");
    if (pcm_i16 && pcm_i16->bytes < (size_t)nframes * 4) return sh::set_error(SH_ERR_INVALID, "sh_bank_render_pcm: PCM buffer too small");
    int rc = bank_check_plain(b, "sh_bank_render");
    if (rc) return rc;
    float2* o32 = bus_f32 ? (float2*)bus_f32->ptr : nullptr;
    double2* o64 = bus_f64 ? (double2*)bus_f64->ptr : nullptr;
    uint32_t* o16 = pcm_i16 ? (uint32_t*)pcm_i16->ptr : nullptr;
    sh::State& S = sh::state();
    // variant = WAVES*100 + FPL*10 + MINW (SYNTHHIP_VARIANT overrides the tuned default)
    static int variant = -1;
    if (variant < 0) {
        const char* e = getenv("SYNTHHIP_VARIANT");
        variant = e ? atoi(e) : 0;
    }
    int var = variant;
    const int mode = b->lean_candidates == 0 ? RENDER_DIRECT : (b->lean_fm_candidates ? RENDER_LEAN_ALL : RENDER_LEAN_HARM);
    // mostly-lean banks take eight frames per lane on long blocks: the recurrences make every frame after the second cost one or
    // two FMAs of trigonometry (one table lookup per eight frames)
    if (var == 0) var = b->nvoices >= 128 ? (2 * b->lean_candidates >= b->nvoices ? (nframes >= 16384 ? 484 : 444) : 844)
                                          : (b->nvoices >= 64 ? 826 : (b->nvoices >= 8 ? 421 : 211));
    const int W = var / 100, F = (var / 10) % 10;
    // Voice groups: split the voices when the frame range alone gives too few tiles for 256 CUs -- up to ONE round of
    // resident workgroups (1024 slots of four waves), not beyond: 752 workgroups that all start at once beat 1504 whose
    // second round runs half empty (MI355X, 1024 voices x 48 000 frames: 4 groups of 188 tiles 46.7 us, 8 groups 47.2, 16
    // groups 50.1; eight frames per lane: 8 groups of 94 tiles 36.9, 16 groups 38.9, 4 groups 40.3).  SYNTHHIP_GROUPS overrides.
    const uint32_t tiles = sh::div_up(nframes, 64 * F);
    const uint32_t slots = 4096u / (uint32_t)W;
    uint32_t groups = 1;
    if (W == 4) {
        while (tiles * groups * 2 <= slots && b->nvoices / (groups * 2) >= (uint32_t)(4 * W)) groups *= 2;
    } else {                                                 // the eight-wave shapes of small / non-lean banks: cover the chip several times over
        while (tiles * groups < 1024 && b->nvoices / (groups * 2) >= (uint32_t)(4 * W)) groups *= 2;
    }
    {
        static int forced = -1;
        if (forced < 0) { const char* e = getenv("SYNTHHIP_GROUPS"); forced = e ? atoi(e) : 0; }
        if (forced > 0) groups = (uint32_t)forced;
    }
    uint32_t vpg = (b->nvoices + groups - 1) / groups;
    if (groups > 1) vpg = (vpg + 63) & ~63u;                // groups are made of whole 64-voice chunks (the lists' unit)
    groups = (b->nvoices + vpg - 1) / vpg;                  // no empty trailing groups
    // Does this launch continue the run of renders (same bank and shape, the next block)?  Then it alternates streams with
    // its predecessor and folds the partial buses of the launch before that; else what is outstanding is folded now and a
    // new run starts on `stream`.
    const bool cont = groups > 1 && b->run_active && b->run_nframes == nframes && b->run_groups == groups &&
                      b->run_tile == (uint32_t)(64 * F) && b->run_next_start == start;
    // an output buffer another bank still owes a fold to: the order of the two writes must be the callers' -- end every run
    for (sh_bank* o : live_banks()) {
        if (o == b) continue;
        for (int k = 0; k < o->npending; ++k) {
            const sh::PendingCombine& pc = o->pending[k];
            if ((o32 && pc.o32 == (void*)o32) || (o64 && pc.o64 == (void*)o64) || (o16 && pc.o16 == (void*)o16) ||
                (o32 && pc.o16 == (void*)o32) || (o16 && pc.o32 == (void*)o16)) {
                rc = sh::flush_pending();
                if (rc) return rc;
            }
        }
    }
    if (!cont) {
        // this bank's own leftovers are folded now; the other banks' runs go on
        if (b->npending || b->run_active) {
            rc = join_aux();
            if (!rc) rc = fold_bank(b);
            if (rc) return rc;
        }
        if (groups > 1) {
            b->run_active = true;
            b->run_nframes = nframes;
            b->run_groups = groups;
            b->run_tile = (uint32_t)(64 * F);
            b->run_count = 0;
            SH_HIP(hipEventRecord(S.ev_join, S.stream));    // everything enqueued so far: stream2's first launch of the run waits for it
        }
    }
    const uint32_t n = groups > 1 ? b->run_count : 0;
    const bool two_streams = groups > 1 && speculation_enabled() && overlap_enabled();
    const bool use_aux = two_streams && (n & 1);
    hipStream_t st = use_aux ? S.stream2 : S.stream;
    if (use_aux && n == 1) SH_HIP(hipStreamWaitEvent(S.stream2, S.ev_join, 0));
    const int prev_cur = b->cur;
    // a bank with lean candidates + several voice groups: the launch is split into a lean and a general kernel
    // (see the RENDER_* modes); SYNTHHIP_NO_SPLIT=1 keeps the combined kernel
    static int no_split = -1;
    if (no_split < 0) { const char* e = getenv("SYNTHHIP_NO_SPLIT"); no_split = (e && e[0] == '1') ? 1 : 0; }
    const bool split = mode != RENDER_DIRECT && groups > 1 && !no_split;
    // A transition launch (RENDER_*_SEG): cut where the envelopes become flat (every voice past its decay) and from there on
    // into segments no longer than their own distance from the note's start -- piece ends of the phase sum lie an octave apart,
    // so such a segment crosses at most one per voice; a cut, too, where the first voice leaves its sustain.  Tile-aligned cuts.
    uint32_t seg_first[SEG_MAX + 1];
    uint32_t nseg = 0;
    {
        static int no_seg = -1;
        if (no_seg < 0) { const char* e = getenv("SYNTHHIP_NO_SEG"); no_seg = (e && e[0] == '1') ? 1 : 0; }
        if (split && (mode == RENDER_LEAN_HARM || mode == RENDER_LEAN_ALL) && var == 484 && b->all_lean && !no_seg && !b->needs_rows && !b->no_general_voice(start, nframes)) {
            nseg = plan_segments(b, start, nframes, (uint64_t)(64 * F), ~0ull, true, seg_first);
            if (nseg < 2 || seg_first[nseg] != nframes) nseg = 0;        // nothing to cut, or more cuts than a launch carries
        }
    }
    if (nseg == 0) {
        rc = acquire_records(b, start, nframes, st, cont);
        if (rc) return rc;
    }
    // partial buses: ring slot n % 4 (last read by the fold in launch n - 2, which is this stream's previous launch)
    double2* parts = nullptr;
    if (groups > 1) {
        const int k = (int)(n & 3);
        // split launch: the general kernel's parts follow the lean kernel's, then one flag per group
        const size_t need = split ? 2 * (size_t)groups * nframes * sizeof(double2) + (size_t)groups * sizeof(uint32_t)
                                  : (size_t)groups * nframes * sizeof(double2);
        if (b->parts_bytes[k] < need) {
            if (b->parts_buf[k]) {
                SH_HIP(hipStreamSynchronize(S.stream));
                SH_HIP(hipStreamSynchronize(S.stream2));
                SH_HIP(hipFree(b->parts_buf[k]));
                b->parts_buf[k] = nullptr;
                b->parts_bytes[k] = 0;
            }
            SH_HIP(hipMalloc(&b->parts_buf[k], need));
            b->parts_bytes[k] = need;
        }
        parts = (double2*)b->parts_buf[k];
    }
    // ... and the general-lists kernel is only launched when the launch CAN hold a general voice
    static int always_general = -1;
    if (always_general < 0) { const char* e = getenv("SYNTHHIP_ALWAYS_GENERAL"); always_general = (e && e[0] == '1') ? 1 : 0; }
    const bool with_general = split && (always_general || !b->no_general_voice(start, nframes));
    uint32_t* gen_valid = with_general ? (uint32_t*)(parts + 2 * (size_t)groups * nframes) : nullptr;
    // Timing diagnostics only (wrong audio!): SYNTHHIP_DEBUG bit 0 drops the in-kernel prepare of the block two launches on
    // once the run is warm (the stale record sets are reused), bit 1 drops the in-kernel fold of the partial buses.
    static int debug = -1;
    if (debug < 0) { const char* e = getenv("SYNTHHIP_DEBUG"); debug = e ? atoi(e) : 0; }
    // the fold this launch takes over: the older of two outstanding ones (launch n - 2's)
    const bool take_over = b->npending == 2;
    const sh::PendingCombine prev = take_over ? b->pending[0] : sh::PendingCombine();
    const double2* pv_parts = (take_over && !(debug & 2)) ? (const double2*)prev.parts : nullptr;
    float2* pv32 = take_over ? (float2*)prev.o32 : nullptr;
    double2* pv64 = take_over ? (double2*)prev.o64 : nullptr;
    uint32_t* pv16 = take_over ? (uint32_t*)prev.o16 : nullptr;
    const double pv_scale = take_over ? prev.scale : 0.0;
    const uint32_t* pv_gen = (take_over && !(debug & 2)) ? (const uint32_t*)prev.gen_valid : nullptr;
    b->last_groups = groups;
    const LaunchSet cur = launch_set(b, b->cur);
    // the records of the block two launches on go to a set that is neither this launch's, nor its predecessor's (perhaps
    // still executing), nor the one holding the block in between
    const uint64_t next_start = start + 2 * (uint64_t)nframes;
    int target = -1;
    if (speculation_enabled()) {
        for (int c = 0; c < sh_bank::NSETS && target < 0; ++c) {
            const bool holds_between = b->spec[c].valid && b->spec[c].start == start + nframes && b->spec[c].nframes == nframes;
            if (c != b->cur && !(cont && (c == prev_cur || c == b->last_target)) && !holds_between) target = c;
        }
    }
    LaunchSet next = launch_set(b, target < 0 ? 0 : target);
    if (target < 0) next.launch = nullptr;
    if ((debug & 1) && n >= 8) next.launch = nullptr;
    if (nseg) {
        const int ks = use_aux ? 1 : 0;
        const uint32_t nchunks = sh::div_up(b->nvoices, 64);
        LaunchSet& g = b->seg_set[ks];
        if (b->seg_cap[ks] < nseg) {
            SH_HIP(hipStreamSynchronize(S.stream));
            SH_HIP(hipStreamSynchronize(S.stream2));
            if (g.launch) { (void)hipFree(g.launch); (void)hipFree(g.fm); (void)hipFree(g.fast); (void)hipFree(g.gen_idx); (void)hipFree(g.counts); g = LaunchSet(); }
            b->seg_cap[ks] = 0;
            const size_t cap = nseg;                          // (grows with the launches that need more: 724 bytes per voice and segment)
            SH_HIP(hipMalloc((void**)&g.launch, sizeof(VoiceLaunch) * cap * b->nvoices));
            SH_HIP(hipMalloc((void**)&g.fm, sizeof(VoiceFM) * cap * b->nvoices));
            SH_HIP(hipMalloc((void**)&g.fast, sizeof(FastRec) * cap * b->nvoices));
            SH_HIP(hipMalloc((void**)&g.gen_idx, sizeof(uint32_t) * cap * b->nvoices));
            SH_HIP(hipMalloc((void**)&g.counts, sizeof(uint32_t) * 4 * cap * nchunks));
            b->seg_cap[ks] = nseg;
        }
        BankPtrs P = ptrs(b);
        P.nseg = nseg;
        for (uint32_t k = 0; k <= nseg; ++k) P.seg_first[k] = seg_first[k];
        hipLaunchKernelGGL(k_prepare_segments_var<true>, dim3(nchunks, nseg), dim3(64), 0, st, P, g, b->nvoices, start);
        SH_CHECK_LAUNCH("k_prepare_segments_var");
        uint32_t tiles_lean = 0, tiles_gen = 0;
        for (uint32_t k = 0; k < nseg; ++k) {
            tiles_lean += sh::div_up(seg_first[k + 1] - seg_first[k], 64 * 8);
            tiles_gen += sh::div_up(seg_first[k + 1] - seg_first[k], 64 * 4);
        }
        if (mode == RENDER_LEAN_HARM)
            hipLaunchKernelGGL((k_bank_render<4, 8, 4, RENDER_LEAN_HARM_SEG>), dim3(tiles_lean, groups), dim3(256), 0, st, P,
                               trig_table(), b->nvoices, vpg, g, next, next_start, start, nframes, o32, o64, parts, pv_parts, pv32, pv64,
                               o16, pcm_scale, pv16, pv_scale, gen_valid, pv_gen);
        else
            hipLaunchKernelGGL((k_bank_render<4, 8, 4, RENDER_LEAN_ALL_SEG>), dim3(tiles_lean, groups), dim3(256), 0, st, P,
                               trig_table(), b->nvoices, vpg, g, next, next_start, start, nframes, o32, o64, parts, pv_parts, pv32, pv64,
                               o16, pcm_scale, pv16, pv_scale, gen_valid, pv_gen);
        SH_CHECK_LAUNCH("k_bank_render(lean, segments)");
        SH_HIP(hipMemsetAsync(gen_valid, 1, (size_t)groups * sizeof(uint32_t), st));         // every group's general parts are written
        // the first segment's groups are split SUB ways (BankPtrs::gen_sub): with sixteen waves per workgroup a wave walks two
        // or three of the 128 voices of its group
        static uint32_t SUB = 0;
        if (!SUB) { const char* e = getenv("SYNTHHIP_GEN_SUB"); SUB = e ? (uint32_t)atoi(e) : 4u; if (SUB < 1 || SUB > 16) SUB = 4; }
        const uint32_t n0 = seg_first[1];
        const size_t scratch_need = (size_t)groups * SUB * n0 * sizeof(double2);
        if (b->seg_scratch_bytes[ks] < scratch_need) {
            SH_HIP(hipStreamSynchronize(S.stream));
            SH_HIP(hipStreamSynchronize(S.stream2));
            if (b->seg_scratch[ks]) { (void)hipFree(b->seg_scratch[ks]); b->seg_scratch[ks] = nullptr; b->seg_scratch_bytes[ks] = 0; }
            SH_HIP(hipMalloc(&b->seg_scratch[ks], scratch_need));
            b->seg_scratch_bytes[ks] = scratch_need;
        }
        P.gen_sub = SUB;
        P.gen_scratch = (double2*)b->seg_scratch[ks];
        LaunchSet none = g;
        none.launch = nullptr;
        hipLaunchKernelGGL((k_bank_render<16, 4, 1, RENDER_GENERAL_SEG>), dim3(tiles_gen + (SUB - 1) * sh::div_up(n0, 64 * 4), groups), dim3(1024), 0, st, P,
                           trig_table(), b->nvoices, vpg, g, none, next_start, start, nframes, (float2*)nullptr, (double2*)nullptr, parts,
                           (const double2*)nullptr, (float2*)nullptr, (double2*)nullptr, (uint32_t*)nullptr, 0.0, (uint32_t*)nullptr, 0.0,
                           gen_valid, (const uint32_t*)nullptr);
        SH_CHECK_LAUNCH("k_bank_render(general, segments)");
        hipLaunchKernelGGL(k_seg_combine, dim3(sh::div_up(n0, 256), groups), dim3(256), 0, st, (const double2*)b->seg_scratch[ks], SUB, n0,
                           parts + (size_t)groups * nframes, nframes);
        SH_CHECK_LAUNCH("k_seg_combine");
    } else {
#define SH_LAUNCH_MODE(W_, F_, M_, MODE_)                                                                         \
    hipLaunchKernelGGL((k_bank_render<W_, F_, M_, MODE_>), dim3(tiles, groups), dim3(W_ * 64), 0, st, ptrs(b),    \
                       trig_table(), b->nvoices, vpg, cur, next, next_start, start, nframes, o32, o64, parts, pv_parts, pv32, pv64, \
                       o16, pcm_scale, pv16, pv_scale, gen_valid, pv_gen)
#define SH_LAUNCH_RENDER(W_, F_, M_)                                                 \
    do {                                                                             \
        if (split && mode == RENDER_LEAN_HARM) SH_LAUNCH_MODE(W_, F_, M_, RENDER_LEAN_HARM_ONLY); \
        else if (split) SH_LAUNCH_MODE(W_, F_, M_, RENDER_LEAN_ALL_ONLY);            \
        else if (mode == RENDER_LEAN_HARM) SH_LAUNCH_MODE(W_, F_, M_, RENDER_LEAN_HARM);  \
        else if (mode == RENDER_LEAN_ALL) SH_LAUNCH_MODE(W_, F_, M_, RENDER_LEAN_ALL); \
        else SH_LAUNCH_MODE(W_, F_, M_, RENDER_DIRECT);                              \
    } while (0)
    switch (var) {
    case 1621: SH_LAUNCH_RENDER(16, 2, 1); break;
    case 826: SH_LAUNCH_RENDER(8, 2, 6); break;
    case 828: SH_LAUNCH_RENDER(8, 2, 8); break;
    case 841: SH_LAUNCH_RENDER(8, 4, 1); break;
    case 844: SH_LAUNCH_RENDER(8, 4, 4); break;
    case 421: SH_LAUNCH_RENDER(4, 2, 1); break;
    case 411: SH_LAUNCH_RENDER(4, 1, 1); break;
    case 441: SH_LAUNCH_RENDER(4, 4, 1); break;
    case 444: SH_LAUNCH_RENDER(4, 4, 4); break;
    case 484: SH_LAUNCH_RENDER(4, 8, 4); break;
    case 884: SH_LAUNCH_RENDER(8, 8, 4); break;
    case 211: SH_LAUNCH_RENDER(2, 1, 1); break;
    case 221: SH_LAUNCH_RENDER(2, 2, 1); break;
    default: return sh::set_error(SH_ERR_INVALID, "sh_bank_render: unknown SYNTHHIP_VARIANT %d", var);
    }
#undef SH_LAUNCH_RENDER
#undef SH_LAUNCH_MODE
    SH_CHECK_LAUNCH("k_bank_render");
    if (with_general) {
        // the general lists of the same launch: four waves x four frames per lane whatever the lean kernel's shape (the parts
        // are indexed by frame), same voice groups, behind the lean kernel on the same stream
        LaunchSet none = cur;
        none.launch = nullptr;
        hipLaunchKernelGGL((k_bank_render<4, 4, 4, RENDER_GENERAL_ONLY>), dim3(sh::div_up(nframes, 256), groups), dim3(256), 0, st, ptrs(b),
                           trig_table(), b->nvoices, vpg, cur, none, next_start, start, nframes, (float2*)nullptr, (double2*)nullptr, parts,
                           (const double2*)nullptr, (float2*)nullptr, (double2*)nullptr, (uint32_t*)nullptr, 0.0, (uint32_t*)nullptr, 0.0,
                           gen_valid, (const uint32_t*)nullptr);
        SH_CHECK_LAUNCH("k_bank_render(general lists)");
    }
    }
    if (use_aux) {
        SH_HIP(hipEventRecord(S.ev_aux, S.stream2));
        S.aux_busy = true;
    }
    if (take_over) {                                        // folded by this launch
        b->pending[0] = b->pending[1];
        b->npending = 1;
        S.pending_total -= 1;
    }
    if (groups > 1) {                                       // this launch's partial buses: folded two launches on, or by the next other API call
        sh::PendingCombine& pc = b->pending[b->npending++];
        S.pending_total += 1;
        pc.parts = parts;
        pc.groups = groups;
        pc.nframes = nframes;
        pc.o32 = o32;
        pc.o64 = o64;
        pc.o16 = o16;
        pc.scale = pcm_scale;
        pc.gen_valid = gen_valid;
        b->run_count = n + 1;
        b->run_next_start = start + nframes;
    }
    b->last_target = target;
    if (target >= 0) {
        b->spec[target].valid = true;
        b->spec[target].start = next_start;
        b->spec[target].nframes = nframes;
    }
    return SH_OK;
}

// Renders of any length: the launch's x dimension is a tile count (tiles x threads per workgroup must stay below 2^32 work-items,
// which the shapes of small banks reach first: sixteen waves on 128 frames), and a bank with several voice groups keeps
// 32 bytes of partial buses per frame and group.  Long renders are therefore a run of launches of RENDER_MAX_FRAMES (87 s
// at 48 kHz; consecutive blocks of one shape: the two-stream pipeline applies) into views of the caller's buffers.
constexpr uint32_t RENDER_MAX_FRAMES = 1u << 22;
static int bank_render_any(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* bus_f32, sh_buf* bus_f64, sh_buf* pcm_i16, double pcm_scale) {
    if (nframes <= RENDER_MAX_FRAMES) return bank_render(b, start, nframes, bus_f32, bus_f64, pcm_i16, pcm_scale);
    SH_API_LOCK();                                           // (recursive) one call: nothing else gets between its launches
    if (bus_f32 && bus_f32->bytes < (size_t)nframes * 8) return sh::set_error(SH_ERR_INVALID, "sh_bank_render: bus_f32 too small");
    if (bus_f64 && bus_f64->bytes < (size_t)nframes * 16) return sh::set_error(SH_ERR_INVALID, "sh_bank_render: bus_f64 too small");
    if (pcm_i16 && pcm_i16->bytes < (size_t)nframes * 4) return sh::set_error(SH_ERR_INVALID, "sh_bank_render_pcm: PCM buffer too small");
    const double* rows = b ? b->launch_rows : nullptr;
    int rc = SH_OK;
    for (uint64_t off = 0; off < nframes && !rc; off += RENDER_MAX_FRAMES) {
        const uint32_t n = nframes - off < RENDER_MAX_FRAMES ? (uint32_t)(nframes - off) : RENDER_MAX_FRAMES;
        sh_buf v32{nullptr, 0, false, 0}, v64{nullptr, 0, false, 0}, v16{nullptr, 0, false, 0};
        if (bus_f32) { v32.ptr = (char*)bus_f32->ptr + off * 8; v32.bytes = bus_f32->bytes - off * 8; }
        if (bus_f64) { v64.ptr = (char*)bus_f64->ptr + off * 16; v64.bytes = bus_f64->bytes - off * 16; }
        if (pcm_i16) { v16.ptr = (char*)pcm_i16->ptr + off * 4; v16.bytes = pcm_i16->bytes - off * 4; }
        if (rows) b->launch_rows = rows + off;               // launch-relative rows (sh_bank_render_rows)
        rc = bank_render(b, start + off, n, bus_f32 ? &v32 : nullptr, bus_f64 ? &v64 : nullptr, pcm_i16 ? &v16 : nullptr, pcm_scale);
    }
    if (rows) b->launch_rows = rows;
    return rc;
}

extern "C" {

int sh_bank_render(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* bus_f32, sh_buf* bus_f64) {
    return bank_render_any(b, start, nframes, bus_f32, bus_f64, nullptr, 0.0);
}

int sh_bank_render_pcm(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* pcm_i16) {
    if (!pcm_i16) return sh::set_error(SH_ERR_INVALID, "sh_bank_render_pcm: NULL PCM buffer");
    return bank_render_any(b, start, nframes, nullptr, nullptr, pcm_i16, scale);
}

int sh_bank_set_rows(sh_bank* b, const int32_t* fm_row, const int32_t* pwm_row) {
    SH_REQUIRE_INIT();
    if (!b || !fm_row || !pwm_row) return sh::set_error(SH_ERR_INVALID, "sh_bank_set_rows: NULL argument");
    for (uint32_t i = 0; i < b->nvoices; ++i) {
        const sh_voice& v = b->h_voices[i];
        if ((v.fm_mode == SH_FM_BUFFER || v.kind == SH_BUFFER) && fm_row[i] < 0)
            return sh::set_error(SH_ERR_INVALID, "sh_bank_set_rows: voice %u (SH_FM_BUFFER / SH_BUFFER) needs a row", i);
        if (pwm_row[i] >= 0 && v.kind != SH_PULSE) return sh::set_error(SH_ERR_INVALID, "sh_bank_set_rows: voice %u has a pwm row but is no Pulse", i);
    }
    hipStream_t st = sh::state().stream;
    if (!b->d_fm_row) {
        SH_HIP(hipMalloc((void**)&b->d_fm_row, sizeof(int32_t) * b->nvoices));
        SH_HIP(hipMalloc((void**)&b->d_pwm_row, sizeof(int32_t) * b->nvoices));
    }
    SH_HIP(hipMemcpyAsync(b->d_fm_row, fm_row, sizeof(int32_t) * b->nvoices, hipMemcpyHostToDevice, st));
    SH_HIP(hipMemcpyAsync(b->d_pwm_row, pwm_row, sizeof(int32_t) * b->nvoices, hipMemcpyHostToDevice, st));
    SH_HIP(hipStreamSynchronize(st));
    b->needs_rows = true;
    b->fm_row_max = -1;
    for (uint32_t i = 0; i < b->nvoices; ++i) {
        if (fm_row[i] > b->fm_row_max) b->fm_row_max = fm_row[i];
        if (pwm_row[i] > b->fm_row_max) b->fm_row_max = pwm_row[i];
    }
    return SH_OK;
}

int sh_bank_render_rows(sh_bank* b, uint64_t start, uint32_t nframes, const sh_buf* rows_f64, size_t row_stride,
                        sh_buf* bus_f32, sh_buf* bus_f64) {
    if (!b || !rows_f64) return sh::set_error(SH_ERR_INVALID, "sh_bank_render_rows: NULL argument");
    SH_API_LOCK();                                           // held across bank_render (recursive): the rows belong to this call
    {
        if (!b->d_fm_row) return sh::set_error(SH_ERR_INVALID, "sh_bank_render_rows: sh_bank_set_rows has not been called");
        if (row_stride < nframes) return sh::set_error(SH_ERR_INVALID, "sh_bank_render_rows: row_stride < nframes");
        if (b->fm_row_max >= 0 && rows_f64->bytes / 8 < (size_t)b->fm_row_max * row_stride + nframes)
            return sh::set_error(SH_ERR_INVALID, "sh_bank_render_rows: rows buffer too small for row %d", b->fm_row_max);
        b->launch_rows = (const double*)rows_f64->ptr;
        b->launch_row_stride = row_stride;
    }
    int rc = bank_render_any(b, start, nframes, bus_f32, bus_f64, nullptr, 0.0);
    b->launch_rows = nullptr;
    b->launch_row_stride = 0;
    return rc;
}

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

int sh_ew_f64(int op, const sh_buf* a, size_t a_off, const sh_buf* b, size_t b_off, size_t n, double p0, double p1,
              sh_buf* out_f64, size_t out64_off, sh_buf* out_f32, size_t out32_off, float* out_host) {
    SH_REQUIRE_INIT();
    if (op < SH_EW_ADD || op > SH_EW_AXPY) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: unknown op %d", op);
    const bool need_a = op != SH_EW_FILL, need_b = op == SH_EW_ADD || op == SH_EW_MUL || op == SH_EW_AXPY;
    if (need_a && (!a || a_off > a->bytes / 8 || n > a->bytes / 8 - a_off)) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: operand a too small");
    if (need_b && (!b || b_off > b->bytes / 8 || n > b->bytes / 8 - b_off)) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: operand b too small");
    if (out_f64 && (out64_off > out_f64->bytes / 8 || n > out_f64->bytes / 8 - out64_off)) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: out_f64 too small");
    if (out_f32 && (out32_off > out_f32->bytes / 4 || n > out_f32->bytes / 4 - out32_off)) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: out_f32 too small");
    if (!out_f64 && !out_f32 && !out_host) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: no destination");
    if (!n) return SH_OK;
    float* d32 = out_f32 ? (float*)out_f32->ptr + out32_off : nullptr;
    if (!d32 && out_host) {
        int rc = sh::ensure_scratch(n * 4);
        if (rc) return rc;
        d32 = (float*)sh::state().scratch;
    }
    hipStream_t st = sh::state().stream;
    hipLaunchKernelGGL(k_ew_f64, sh::grid1d(n, 256), dim3(256), 0, st, op,
                       need_a ? (const double*)a->ptr + a_off : nullptr, need_b ? (const double*)b->ptr + b_off : nullptr,
                       n, p0, p1, out_f64 ? (double*)out_f64->ptr + out64_off : nullptr, d32);
    SH_CHECK_LAUNCH("k_ew_f64");
    if (out_host) {
        SH_HIP(hipMemcpyAsync(out_host, d32, n * 4, hipMemcpyDeviceToHost, st));
        SH_HIP(hipStreamSynchronize(st));
    }
    return SH_OK;
}

int sh_scan_rows_f64(sh_buf* rows, size_t row0, uint32_t nrows, uint32_t n, size_t row_stride, sh_buf* carry) {
    SH_REQUIRE_INIT();
    if (!rows || !carry) return sh::set_error(SH_ERR_INVALID, "sh_scan_rows_f64: NULL argument");
    if (!nrows || !n) return SH_OK;
    if (row_stride < n || rows->bytes / 8 < (row0 + nrows - 1) * row_stride + n) return sh::set_error(SH_ERR_INVALID, "sh_scan_rows_f64: rows buffer too small");
    if (carry->bytes / 8 < nrows) return sh::set_error(SH_ERR_INVALID, "sh_scan_rows_f64: carry buffer smaller than nrows doubles");
    if (nrows > 65535) return sh::set_error(SH_ERR_INVALID, "sh_scan_rows_f64: more than 65535 rows");
    const uint32_t ntiles = sh::div_up(n, SCAN_TILE);
    int rc = sh::ensure_scratch((size_t)nrows * (ntiles + 1) * 8);
    if (rc) return rc;
    double* sums = (double*)sh::state().scratch;
    double* x = (double*)rows->ptr + row0 * row_stride;
    hipStream_t st = sh::state().stream;
    hipLaunchKernelGGL(k_scan_rows_tile_sums, dim3(ntiles, nrows), dim3(256), 0, st, (const double*)x, row_stride, n, ntiles, sums);
    SH_CHECK_LAUNCH("k_scan_rows_tile_sums");
    hipLaunchKernelGGL(k_scan_rows_sums, dim3(nrows), dim3(256), 0, st, sums, ntiles, (double*)carry->ptr);
    SH_CHECK_LAUNCH("k_scan_rows_sums");
    hipLaunchKernelGGL(k_scan_rows_apply, dim3(ntiles, nrows), dim3(256), 0, st, (const double*)x, row_stride, n, ntiles, (const double*)sums, x);
    SH_CHECK_LAUNCH("k_scan_rows_apply");
    return SH_OK;
}

int sh_bank_generate_f64(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* rows_out, size_t row0, size_t row_stride) {
    SH_REQUIRE_INIT();
    if (!b || !rows_out) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_f64: NULL argument");
    if (nframes == 0) return SH_OK;
    if (nframes > 0xFFFF0000u) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_f64: at most 2^32 - 65536 frames per call");
    if (row_stride < nframes || rows_out->bytes / 8 < (row0 + b->nvoices - 1) * row_stride + nframes)
        return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_f64: rows buffer too small");
    int rc = bank_check_plain(b, "sh_bank_generate_f64");
    if (rc) return rc;
    rc = acquire_records(b, start, nframes, sh::state().stream, false);
    if (rc) return rc;
    const int fpl = nframes >= 8192 ? 4 : (nframes >= 2048 ? 2 : 1);
    const uint32_t tile_groups = sh::div_up(nframes, 256 * fpl);
    uint32_t vpg = 1;
    while (vpg < 64 && (uint64_t)tile_groups * ((b->nvoices + 2 * vpg - 1) / (2 * vpg)) >= 4096) vpg *= 2;
    while ((b->nvoices + vpg - 1) / vpg > 65535) vpg *= 2;
    const uint32_t groups = (b->nvoices + vpg - 1) / vpg;
    double* o = (double*)rows_out->ptr + row0 * row_stride;
#define SH_GEN64(F_) hipLaunchKernelGGL(k_generate<F_>, dim3(tile_groups, groups), dim3(256), 0, sh::state().stream,          \
                                        ptrs(b), trig_table(), 0u, b->nvoices, vpg, b->d_launch, b->d_launch_fm, start, nframes, \
                                        (const double*)nullptr, (const double*)nullptr, (float*)nullptr, o, row_stride)
    if (fpl == 4) SH_GEN64(4); else if (fpl == 2) SH_GEN64(2); else SH_GEN64(1);
#undef SH_GEN64
    SH_CHECK_LAUNCH("k_generate(f64 rows)");
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
