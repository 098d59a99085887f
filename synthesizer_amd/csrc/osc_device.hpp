// osc_device.hpp -- device-side records and arithmetic shared by the oscillator-bank kernels (gfx950).
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
// Translation units (each compiled on its own, no relocatable device code: a kernel lives where it is launched):
//   osc_bank.hip      the voice table in HBM (sh_bank), the launch records of a block (k_prepare*: everything that depends on
//                     `start` resolved once per voice, voices classified into lean / general / silent lists)
//   osc_render.hip    fused generate-and-mix (k_bank_render), the fold of the voice groups' partial buses, the two-stream
//                     pipeline of consecutive blocks
//   osc_generate.hip  voices materialised as PCM rows (k_generate*), single oscillators (sh_osc_render)
//   osc_mixbus.hip    the HBM-bound mixer over materialised float32 voices
//   osc_scan.hip      float64 running sums (arbitrary fm_lfo modulators) and the elementwise filter kernel
#pragma once
#include "common.hpp"
#include "devmath.hpp"
#include <type_traits>

namespace shosc {

constexpr int SEG_MAX = 24;            // segments of a transition launch
// Unequal segments of a materialised row's head (n = 0: none): entry k holds the frames [first[k], first[k] + len[k]) and reads
// record set set[k] (the lists launch skips the segments the host can prove free of general voices)
struct SegTab { uint32_t n; uint32_t first[SEG_MAX]; uint32_t len[SEG_MAX]; uint32_t set[SEG_MAX]; };

// A TILE-CLASSIFIED launch (banks whose notes do not move in lock-step: onsets and envelope corners of their own).  Whether a
// voice can take the lean loop is decided per (voice, tile of TILE_FRAMES frames), not per launch.  A pair is LEAN when the voice
// is a polynomial-Harmonics voice and the tile holds at most ONE corner of its envelope -- the onset counts as one: silence |
// attack -- and at most two ends of pieces of its phase table; then 128 bytes and the voice's polynomial (a table by voice) say all the lean arithmetic needs:
//   * the envelope over the tile is one line in front of the corner's frame and another from it on (no corner: the same line
//     twice) -- one compare and select per frame in the tiles that hold a corner, no list of corners.  (First version: the minimum
//     or maximum of the two lines, one operation -- wrong by up to one step of the slope in the corner's frame: the reference
//     switches lines at an INTEGER frame, where the lines need not have met yet; found by a bank of notes with ADSR times that are
//     no multiples of the sample period, tests/test_gpu_onsets.py.)  In front of an onset the line is the constant 0;
//   * piece ends inside the tile (the first tile of a note runs through several binades of its phase sum) make it a multi-piece
//     tile: every frame is looked up from the piece that holds it (up to three compares), no recurrence across the ends.
// A pair that sounds but holds two corners (or an onset without an attack) or more piece ends takes the general code for that
// tile only; the rest is silent.  One bit per pair in two masks per (tile, chunk of 64 voices): the kernels walk set bits.
constexpr uint32_t TILE_FRAMES = 512;         // = the lean render kernel's tile (four waves, eight frames per lane)
constexpr uint32_t TILE_MAX_PIECES = 3;
struct alignas(64) TileRec {
    double t0, dt;                // t(i) = fma(i - tile0, dt, t0): the accumulated phase at the tile's first frame, its piece's step
    double rc, rs;                // cos / sin of 64 dt (LEAN_FM: rc = the voice's own index of the tile's first frame; t0, dt: its TIME table)
    double ea0, ea1, eb0, eb1;    // envelope(i) = ea0 + (i - tile0) ea1 for i - tile0 < corner, eb0 + (i - tile0) eb1 from there on
    double GL, GR;                // amplitude * bus gain
    double tb[2], db[2];          // pieces 1, 2: frames i - tile0 >= split[k] lie on piece k + 1, t = fma(i - tile0 - split[k], db[k], tb[k])
    uint16_t split[2];            // 0xFFFF: no such piece
    uint16_t npieces;             // 0: a WALK pair -- more piece ends than a record lists (the first tiles of a note: its phase sum runs
                                  // through a binade every few frames) and / or an onset with the end of the attack behind it.  The lean
                                  // kernel reads the pieces from the voice's table itself: tb[0] = the voice's own index of the tile's first
                                  // frame (negative: the onset lies inside), tb[1] = (table index of the first piece | end of the voice's
                                  // table << 32) as bits; the frames in front of the onset get the angle 0 -- sin 0 = 0 -- and the envelope
                                  // is the minimum of the two lines behind the onset
    uint16_t corner;              // the first frame (relative to the tile's) of the second line; 0: no corner in the tile
    double pad_;                  // low 32 bits: LEAN_HARM (a plain Sine too: the series with one partial) or the waveform LEAN_SAW .. LEAN_PULSE;
                                  // high 32 bits: the voice's position in its chunk of 64 (the lean kernel finds its polynomial by it)
};
constexpr uint32_t TILE_WALK_PIECES = 16;     // pieces a walk pair may touch (lanes 0 .. 15 fetch one each)
static_assert(sizeof(TileRec) == 128 && offsetof(TileRec, GL) == 64 && offsetof(TileRec, tb) == 80 && offsetof(TileRec, split) == 112, "TileRec layout");
// lane `src` (wave-uniform) of a float64 vector value
__device__ __forceinline__ double readlane_f64(double v, uint32_t src) {
    union { double d; int u[2]; } a, b;
    a.d = v;
    b.u[0] = __builtin_amdgcn_readlane(a.u[0], (int)src);
    b.u[1] = __builtin_amdgcn_readlane(a.u[1], (int)src);
    return b.d;
}
// the position of the (p + 1)-th set bit of m (p < popcount(m)): wave-uniform scalar arithmetic, six halvings
__device__ __forceinline__ uint32_t nth_set_bit(uint64_t m, uint32_t p) {
    uint32_t pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const uint64_t low = m & ((1ull << w) - 1ull);
        const uint32_t cnt = (uint32_t)__popcll(low);
        if (p >= cnt) { p -= cnt; m >>= w; pos += (uint32_t)w; } else { m = low; }
    }
    return pos;
}
// the accumulated t of a lane's frame j (64 samples apart) in a multi-piece tile (by value members: see LaneTheta)
struct TileTheta {
    double   lane_d, t0, dt, tb0, tb1, db0, db1;
    uint32_t lane, s1, s2;
    __device__ __forceinline__ double operator()(int j) const {
        const uint32_t i = lane + (uint32_t)j * 64u;
        const double dd = lane_d + (double)(j * 64);
        double t = fma(dd, dt, t0);
        if (i >= s1) t = fma(dd - (double)s1, db0, tb0);
        if (i >= s2) t = fma(dd - (double)s2, db1, tb1);
        return t;
    }
};
struct TileSet {
    TileRec*  recs;               // [tile][slot]: the lean pairs of chunk c of a tile compacted at slots 64 (c - k0 groups) .. (their count: the low
                                  // half of the chunk's word in `lean`); a tile's row holds rec_chunks chunks -- the set's RANGE of chunks, not the table (a table of
                                  // 22 528 notes of which 30 chunks sound: 30 MB of records per set instead of 271)
    // masks: [tile][group][k], chunk c = group + k * groups (a voice group of a tile-classified launch = every groups-th chunk:
    // notes that sound together tend to be neighbours in the voice table, and a range of chunks would give one group all of
    // them); mask_k = masks per (tile, group), a multiple of 8: a workgroup fetches its masks in batches of eight scalar loads
    uint64_t* lean;               // the chunk's lean pairs in this tile: their number | the Harmonics pairs << 32 | the FM Sine pairs << 40 (the list's runs)
    uint64_t* gen;                // bit b = voice 64 c + b takes the general code in this tile
    uint32_t  groups, mask_k;
    // the set is resolved (and read) for the chunks [k0 groups, k1 groups) only: the masks of a group are its slots k0 .. k1 - 1 --
    // the chunks in front and behind are silent throughout the block (a table of notes in the order they start: a few dozen of
    // several hundred chunks sound in any one block), nobody writes or reads their masks
    uint32_t  k0, k1;
    uint32_t  rec_chunks;         // chunks per tile row of recs (>= (k1 - k0) groups)
};
__host__ __device__ __forceinline__ uint32_t tile_mask_k(uint32_t nvoices, uint32_t groups) {
    const uint32_t nchunks = (nvoices + 63) / 64;
    return ((nchunks + groups - 1) / groups + 7) & ~7u;
}

struct BankPtrs {
    const sh_voice*   voices;
    const sh_segment* segs;
    const double*     coefs;
    const sh_partial* partials;
    uint32_t*         hint;       // per voice: table piece of the last prepared launch (streaming: same or next piece)
    uint32_t*         lfo_hint;   // ... and the piece of an FM Sine voice's LFO table (oscillators.LfoTable)
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
    // per launch (tile-classified launches, RENDER_*_TILES): the launch's tile set; polys = [slot][16] polynomial coefficients by voice
    TileSet           tiles;
    const double*     polys;
    // per chunk of 64 voices: [first onset, last frame of sound + 1) of its voices (2^64 - 1: no end) -- a chunk silent throughout
    // a launch costs the classification one scalar load
    const uint64_t*   chunk_span;
    // ... and the set that launch resolves for the block expected two launches on (next_tiles.recs = NULL: none): next_tile_wgs
    // workgroups per row of the grid, behind the workgroups that resolve that block's launch records
    TileSet           next_tiles;
    uint32_t          next_tile_wgs, next_ntiles;
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
                   FL_POLY = 0x100, FL_FOLDED = 0x200, FL_FLIP = 0x400, FL_SILENT = 0x800, FL_ONSET = 0x1000;
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
    // a voice with an onset (sh_voice::start_frame): everything above is relative to the VOICE's sample index.  start_rel = the
    // voice's index at the launch's first frame it sounds in; onset_i > 0 only in the one launch that holds the onset: the voice
    // is silent in frames i < onset_i and its sample index there is i - onset_i (start_rel = 0)
    uint64_t start_rel;
    uint32_t onset_i;
    uint32_t pad_;
};
static_assert(sizeof(VoiceLaunch) == 400, "VoiceLaunch layout");

struct alignas(16) VoiceFM {      // only read for FM voices
    double frequency, phase0, f_inc;      // theta = frequency*T + fma(f_inc, L, phase0)
    double lfo_a_rel, lfo_d, lfo_K, lfo_C0, lfo_bias;   // L(i) = K*(C0 - cos(a_rel + i*d)) + bias*(start+i)
    double lfo_rot_c, lfo_rot_s;          // cos / sin of 64*lfo_d: the LFO angle of a lane's next frame is one rotation away
    // The LFO's own phase is an accumulated sum (oscillators.LfoTable): a_rel, d, K, C0 are those of the LFO's table piece that holds
    // the launch's first frame; if that piece ends inside the launch, frames i >= lfo_split take the next piece's (further ends inside
    // one launch exist only where t is small and an ulp of it with it: the second piece is followed beyond them)
    double lfo2_a_rel, lfo2_d, lfo2_K, lfo2_C0;
    uint32_t lfo_split, pad_[3];          // 0xFFFFFFFF: none
};
static_assert(sizeof(VoiceFM) == 128, "VoiceFM layout");

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
                                  // lfo_K, lfo_C0, lfo_bias, lfo_rot_c, lfo_rot_s, (double)start (see VoiceFM); poly[11..15], pad2 =
                                  // the LFO's next table piece (a_rel, d, K, C0, rot_c, rot_s) for the frames i >= pad1
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
    uint32_t*    counts;          // per chunk: [4c] = lean voices, [4c+1] = general voices, [4c+2] = silent voices, [4c+3] = Harmonics | FM Sine << 8 of the lean ones
};

struct PrepInfo {                 // what prepare_voice found, for the classification
    bool   fast, silent;
    double t_base, dt, gain_l, gain_r, rot_c, rot_s;
    double t0_b, dt_b, rot_c_b, rot_s_b;
    uint32_t remain;
    uint32_t kind;                // LEAN_HARM / LEAN_FM
    double   amplitude, g0u, pulsewidth;
    double   fmv[16];             // LEAN_FM: the values that go to FastRec::poly[0..15]
    double   lfo2_rot_s;          // ... and to pad2, pad1
    uint32_t lfo_split;
    double   slope_l, slope_r;    // a record set of sloped lean records (SLOPED): gain(i) = gain + i * slope on the envelope's line
    double   poly[16];            // FL_POLY: the coefficients (read once, with the table pieces; the lean record is written from here)
    // the int16 boundary guard of a lean polynomial-Harmonics record (sh_voice::guard_*): the voice's own list, and -- in one word --
    // its length (low 8 bits) under the launch's tolerance as a float32 rounded UP to 24 bits (guard_t * |t| at the launch's end + guard_c)
    uint64_t guard_ptr;
    uint32_t guard_word;
};

// SLOPED (the record sets of a segmented transition launch): a polynomial-Harmonics voice on ONE envelope line of any slope is
// lean too -- its record carries the line folded into the bus gains (gain + i * slope) -- not only one on a constant gain.
template <bool SLOPED = false>
__device__ __forceinline__ void prepare_voice(const BankPtrs& B, uint32_t first, uint32_t vi, uint64_t launch_start, uint32_t launch_nframes,
                                              VoiceLaunch* __restrict__ out, VoiceFM* __restrict__ out_fm, PrepInfo& info) {
    // Fields are stored straight to the record (no local struct: a 368-byte private array would give
    // every wave of the render kernel a scratch allocation).
    // The loads are arranged in BATCHES of independent requests -- the voice and its hint; then the table pieces around the
    // hint, their rotations and the polynomial; then (transition launches only) the ends of the following pieces -- because a
    // lane's chain of dependent round trips IS the time this step takes: one wavefront resolves 64 voices, nothing hides its
    // latency, and in a stream the step runs inside every render launch (eight dependent round trips took 8 us; three take 3).
    const sh_voice& v = B.voices[first + vi];
    VoiceLaunch* __restrict__ o = out + vi;
    // A voice with an onset (start_frame > 0) is silent before it and runs on its own sample index n - start_frame from there:
    // launches behind the onset are resolved at start - start_frame like any other; the one launch that HOLDS the onset is
    // resolved from the voice's sample 0 for the frames behind it (general code: the frames in front are zeros); launches in front
    // of it get a silent record.
    const uint64_t onset = v.start_frame;
    if (launch_start + (uint64_t)launch_nframes <= onset) {
        o->flags = (uint32_t)v.kind | FL_ENV_UNIFORM | FL_SILENT;
        o->onset_i = 0;
        o->start_rel = 0;
        info.fast = false;
        info.silent = true;
        return;
    }
    const uint32_t onset_i = launch_start < onset ? (uint32_t)(onset - launch_start) : 0u;
    const uint64_t start = launch_start < onset ? 0ull : launch_start - onset;
    const uint32_t nframes = launch_nframes - onset_i;
    o->onset_i = onset_i;
    o->start_rel = start;
    o->pad_ = 0;
    const bool fm = v.fm_mode != SH_FM_NONE;
    const uint32_t off = fm ? v.time_seg_offset : v.seg_offset;
    const uint32_t cnt = fm ? v.time_seg_count : v.seg_count;
    const sh_segment* tab = B.segs + off;
    const double2* rots = B.seg_rot + off;
    const uint32_t hint = B.hint[first + vi];
    const uint32_t last_piece = cnt - 1;
    const bool poly_form = v.harm_dense == 2;
    const double* harm = v.harm_dense ? (B.coefs + v.harm_offset) : reinterpret_cast<const double*>(B.partials + v.harm_offset);
    // ---- second batch: pieces hint .. hint + 3 (indices clamped into the table), rotations, polynomial ----
    const uint32_t h0 = hint < cnt ? hint : 0u;
    const uint32_t h1 = h0 + 1 < cnt ? h0 + 1 : last_piece, h2 = h0 + 2 < cnt ? h0 + 2 : last_piece, h3 = h0 + 3 < cnt ? h0 + 3 : last_piece;
    uint64_t p0_n0 = tab[h0].n0, p1_n0 = tab[h1].n0, p2_n0 = tab[h2].n0;
    const uint64_t p3_n0 = tab[h3].n0;
    double p0_t0 = tab[h0].t0, p0_dt = tab[h0].dt, p1_t0 = tab[h1].t0, p1_dt = tab[h1].dt;
    const double p2_t0 = tab[h2].t0, p2_dt = tab[h2].dt;
    double2 r0 = rots[h0], r1 = rots[h1];
    const double2 r2 = rots[h2];
#pragma unroll
    for (int u = 0; u < 16; ++u) info.poly[u] = poly_form ? harm[u] : 0.0;
    uint32_t lo;
    if (hint < cnt && p0_n0 <= start && hint + 2 < cnt && start < p2_n0) {
        lo = hint;
        if (start >= p1_n0) {                                  // streaming: still on the piece, or on the next one
            lo = hint + 1;
            p0_n0 = p1_n0; p0_t0 = p1_t0; p0_dt = p1_dt;
            p1_n0 = p2_n0; p1_t0 = p2_t0; p1_dt = p2_dt;
            p2_n0 = p3_n0;
            r0 = r1;
            r1 = r2;
        }
    } else {
        lo = 0;
        uint32_t hi = cnt - 1;
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (tab[mid].n0 <= start) lo = mid; else hi = mid - 1;
        }
        const uint32_t l1 = lo + 1 < cnt ? lo + 1 : last_piece, l2 = lo + 2 < cnt ? lo + 2 : last_piece;
        p0_n0 = tab[lo].n0; p0_t0 = tab[lo].t0; p0_dt = tab[lo].dt;
        p1_n0 = tab[l1].n0; p1_t0 = tab[l1].t0; p1_dt = tab[l1].dt;
        p2_n0 = tab[l2].n0;
        r0 = rots[lo];
        r1 = rots[l1];
    }
    // from here on: piece lo = (p0_n0, p0_t0, p0_dt) with rotation r0, piece lo + 1 = (p1_*) with r1 where it exists, and the
    // start of piece lo + 2 = p2_n0 where it exists
    B.hint[first + vi] = lo;
    const double dt = p0_dt;
    const double t_base = fma((double)(start - p0_n0), dt, p0_t0);
    const uint64_t rem = (lo + 1 < cnt) ? (p1_n0 - start) : 0xFFFFFFFFull;
    o->t_base = t_base;
    o->dt = dt;
    o->remain = rem > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rem;
    o->seg = lo;
    info.guard_ptr = 0;
    info.guard_word = 0;
    if (v.guard_count != 0 && v.guard_count <= 255u) {
        const double t_end = fabs(t_base) + (double)nframes * fabs(dt);     // (the pieces that follow inside the launch differ from dt by ulps)
        const float tol = __double2float_ru(fma(v.guard_t, t_end, v.guard_c) * 1.000001);
        info.guard_word = ((__float_as_uint(tol) + 0xFFu) & ~0xFFu) | v.guard_count;
        info.guard_ptr = (uint64_t)(uintptr_t)(B.partials + v.guard_offset);
    }
    if (rem >= (uint64_t)nframes) {                          // the launch stays on the piece: nothing follows inside it
#pragma unroll
        for (int k = 0; k < NXP; ++k) o->nx_end[k] = 0xFFFFFFFFu;
        o->nx_count = 0;
    } else {
        // ---- third batch: the starts of the pieces lo + 2 .. lo + 1 + NXP (the ends of the pieces that follow) ----
        uint64_t nxn[NXP];
#pragma unroll
        for (int k = 0; k < NXP; ++k) {
            const uint32_t q = lo + 2 + (uint32_t)k;
            nxn[k] = tab[q < cnt ? q : last_piece].n0;
        }
        uint32_t count = 0;
        uint64_t piece_start = rem;                           // launch-relative first frame of the next piece
#pragma unroll
        for (int k = 0; k < NXP; ++k) {
            const uint32_t pi = lo + 1 + k;
            uint32_t end = 0xFFFFFFFFu;
            if (count == (uint32_t)k && pi < cnt && piece_start < (uint64_t)nframes) {      // still inside the launch
                const uint64_t e = (pi + 1 < cnt) ? (nxn[k] - start) : 0xFFFFFFFFull;
                end = e > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)e;
                piece_start = e;
                count = (uint32_t)k + 1;
            }
            o->nx_end[k] = end;
        }
        o->nx_count = count;
    }
    uint32_t flags = (uint32_t)v.kind | ((uint32_t)v.fm_mode << FL_FM_SHIFT) | (onset_i ? FL_ONSET : 0u) |
                     (v.harm_dense == 1 ? FL_DENSE : 0u) | (v.harm_dense == 2 ? FL_POLY : 0u) | (v.flip ? FL_FLIP : 0u);
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
    const double rc = r0.x, rs = r0.y;
    o->rot_c = rc;
    o->rot_s = rs;
    if (flags & FL_POLY) {
#pragma unroll
        for (int u = 0; u < 16; ++u) o->poly[u] = info.poly[u];
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
    const bool two_pieces = !one_piece && lo + 1 < cnt && ((lo + 2 < cnt) ? (p2_n0 - start >= (uint64_t)nframes) : true);
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
    // the LFO of a closed-form FM voice: the piece of ITS table (pairs of records at seg_offset: oscillators.LfoTable) that holds the
    // launch's first frame -- a scalar-free binary search per lane, ~6 dependent loads, FM voices only -- and the next one's start
    double l_a = fma((double)start - 0.5, v.lfo_d, v.lfo_a), l_d = v.lfo_d, l_K = v.lfo_K, l_C0 = v.lfo_C0;      // (a constant LFO: no table)
    double l2_a = l_a, l2_d = l_d, l2_K = l_K, l2_C0 = l_C0;
    double2 l_rot = make_double2(1.0, 0.0), l2_rot = make_double2(1.0, 0.0);
    uint32_t l_split = 0xFFFFFFFFu;
    const bool lfo_pieces = v.fm_mode == SH_FM_SINE && v.seg_count >= 2;
    if (lfo_pieces) {
        const sh_segment* lt = B.segs + v.seg_offset;
        const uint32_t np_ = v.seg_count / 2;
        // (the piece of the last prepared launch, or the one behind it -- one batch of loads -- before the binary search's eight
        // dependent ones: a lane's chain of round trips IS the time this step takes)
        uint32_t lo_ = B.lfo_hint[first + vi];
        if (lo_ >= np_) lo_ = np_ - 1;
        {
            const uint64_t h0 = lt[2 * lo_].n0, h1 = lo_ + 1 < np_ ? lt[2 * lo_ + 2].n0 : ~0ull, h2 = lo_ + 2 < np_ ? lt[2 * lo_ + 4].n0 : ~0ull;
            if (h0 <= start && start < h1) {
            } else if (h1 <= start && start < h2) {
                lo_ += 1;
            } else {
                uint32_t hi_ = np_ - 1;
                lo_ = 0;
                while (lo_ < hi_) {
                    const uint32_t mid = (lo_ + hi_ + 1) >> 1;
                    if (lt[2 * mid].n0 <= start) lo_ = mid; else hi_ = mid - 1;
                }
            }
        }
        B.lfo_hint[first + vi] = lo_;
        const uint64_t q_n0 = lt[2 * lo_].n0;
        const double q_t0 = lt[2 * lo_].t0, q_dt = lt[2 * lo_].dt;
        l_a = fma((double)(start - q_n0) - 0.5, q_dt, q_t0);            // the angle at frame i: a + i d
        l_d = q_dt;
        l_K = lt[2 * lo_ + 1].t0;
        l_C0 = lt[2 * lo_ + 1].dt;
        l_rot = B.seg_rot[v.seg_offset + 2 * lo_];
        l2_a = l_a; l2_d = l_d; l2_K = l_K; l2_C0 = l_C0;
        if (lo_ + 1 < np_) {
            const uint64_t e_n0 = lt[2 * lo_ + 2].n0;
            if (e_n0 - start < (uint64_t)nframes) {
                l_split = (uint32_t)(e_n0 - start);
                const double e_t0 = lt[2 * lo_ + 2].t0, e_dt = lt[2 * lo_ + 2].dt;
                l2_a = fma(-(double)(e_n0 - start) - 0.5, e_dt, e_t0);    // (start - n0' - 1/2) d' + t0': the same frame index i
                l2_d = e_dt;
                l2_K = lt[2 * lo_ + 3].t0;
                l2_C0 = lt[2 * lo_ + 3].dt;
                l2_rot = B.seg_rot[v.seg_offset + 2 * lo_ + 2];
            }
        }
    } else if (fm) {
        l_rot = B.lfo_rot[first + vi];
    }
    info.fast = ((flags & (FL_POLY | FL_FOLDED | FL_FM | FL_SILENT)) == (FL_POLY | FL_FOLDED) || lean_fm || lean_plain || lean_sloped) &&
                (one_piece || two_pieces) && onset_i == 0;     // (the launch that holds the onset: general code)
    info.remain = one_piece ? 0xFFFFFFFFu : (uint32_t)rem;
    info.t0_b = one_piece ? t_base : p1_t0;
    info.dt_b = one_piece ? dt : p1_dt;
    info.rot_c_b = one_piece ? r0.x : r1.x;
    info.rot_s_b = one_piece ? r0.y : r1.y;
    info.t_base = t_base;
    info.dt = dt;
    info.gain_l = gain_l;
    info.gain_r = gain_r;
    info.rot_c = rc;
    info.rot_s = rs;
    if (fm) {
        VoiceFM* __restrict__ f = out_fm + vi;
        f->frequency = v.frequency;
        f->phase0 = v.fm_phase0;
        f->f_inc = v.frequency * v.fm_inc;
        f->lfo_a_rel = l_a;                   // arg(i) = a + (start + i - 0.5) * d on the LFO's piece
        f->lfo_d = l_d;
        f->lfo_K = l_K;
        f->lfo_C0 = l_C0;
        f->lfo_bias = v.lfo_bias;
        f->lfo_rot_c = l_rot.x;
        f->lfo_rot_s = l_rot.y;
        f->lfo2_a_rel = l2_a; f->lfo2_d = l2_d; f->lfo2_K = l2_K; f->lfo2_C0 = l2_C0;
        f->lfo_split = l_split;
        f->pad_[0] = 0; f->pad_[1] = 0; f->pad_[2] = 0;
        info.fmv[0] = v.frequency; info.fmv[1] = v.fm_phase0; info.fmv[2] = v.frequency * v.fm_inc;
        info.fmv[3] = l_a;
        info.fmv[4] = l_d; info.fmv[5] = l_K; info.fmv[6] = l_C0; info.fmv[7] = v.lfo_bias;
        info.fmv[8] = l_rot.x; info.fmv[9] = l_rot.y; info.fmv[10] = (double)start;
        // (the LFO's next piece, should this one end inside the launch: frames i >= lfo_split)
        info.fmv[11] = l2_a; info.fmv[12] = l2_d; info.fmv[13] = l2_K; info.fmv[14] = l2_C0; info.fmv[15] = l2_rot.x;
        info.lfo2_rot_s = l2_rot.y;
        // the lean loops change pieces at a TILE boundary: the first multiple of 1024 frames (a multiple of every shape's tile) at or
        // behind the piece's end.  The up to 1023 frames in between follow the first piece's line: their phase is off by at most
        // 1023 x (the difference of the two pieces' dt: an ulp of the LFO's t) -- 2e-9 rad at t = 8192, five minutes into a 5 Hz LFO --
        // and the carrier's angle, transiently, by f_inc amp 1023^2 / 2 times that difference: 2e-7 rad at depth 0.5 under a 3.5 kHz
        // carrier, below the rounding noise of the reference's own running sums by then (tools/probe.py fm-long).  The general
        // code (VoiceFM::lfo_split) changes at the exact frame.
        info.lfo_split = l_split == 0xFFFFFFFFu ? l_split : ((l_split + 1023u) & ~1023u);
    }
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
    // the lean list in three runs -- polynomial Harmonics, FM Sine, the rest (plain Sine and the waveforms) -- so that a bank of mixed
    // kinds is rendered by three loops one after the other, each as tight as the kernel of a bank of that kind alone (lean_lists)
    const uint64_t mh = __ballot(is_fast && info.kind == LEAN_HARM), mm = __ballot(is_fast && info.kind == LEAN_FM);
    const uint32_t n_harm = (uint32_t)__popcll(mh), n_fm = (uint32_t)__popcll(mm);
    if (is_fast) {
        const uint32_t pos = info.kind == LEAN_HARM ? (uint32_t)__popcll(mh & below)
                           : info.kind == LEAN_FM ? n_harm + (uint32_t)__popcll(mm & below)
                           : n_harm + n_fm + (uint32_t)__popcll(mf & ~mh & ~mm & below);
        FastRec* __restrict__ f = S.fast + c * 64 + pos;
        f->t_base = info.t_base; f->dt = info.dt;
        f->gain_l = info.gain_l; f->gain_r = info.gain_r;
        f->rot_c = info.rot_c; f->rot_s = info.rot_s;
        if (info.kind == LEAN_FM) {
#pragma unroll
            for (int u = 0; u < 16; ++u) f->poly[u] = info.fmv[u];
        } else if (info.kind != LEAN_HARM) {
#pragma unroll
            for (int u = 0; u < 16; ++u) f->poly[u] = u == 0 ? info.pulsewidth : 0.0;
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) f->poly[u] = info.poly[u];
        }
        f->t0_b = info.t0_b; f->dt_b = info.dt_b;
        f->rot_c_b = info.rot_c_b; f->rot_s_b = info.rot_s_b;
        f->off_b = (double)info.remain;
        f->remain = info.remain;
        f->kind = info.kind;
        f->amplitude = SLOPED ? info.slope_l : info.amplitude;        // (SLOPED sets are read by the RENDER_LEAN_*_SEG kernels only)
        f->g0u = SLOPED ? info.slope_r : info.g0u;
        f->vi = vi;
        // (LEAN_FM: where the LFO's table piece ends, 0xFFFFFFFF = not in this launch; LEAN_HARM: the int16 boundary guard, see PrepInfo)
        uint32_t gw = info.guard_word;
        if (info.kind == LEAN_HARM) {
            // a record with a NaN or an infinity in it (amplitude, gain, coefficients, phase) gets an infinite tolerance: the int16 kernels
            // then take every sample of it through the careful path, which raises the overflow flag for what is not a number
            double z = (info.amplitude + info.g0u + info.t_base + info.dt + info.t0_b + info.dt_b) * 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) z = fma(info.poly[u], 0.0, z);
            if (!(z == 0.0)) gw = 0x7F800000u | (gw & 0xFFu);
        }
        f->pad1 = info.kind == LEAN_FM ? info.lfo_split : info.kind == LEAN_HARM ? gw : 0u;
        f->pad2 = info.kind == LEAN_FM ? info.lfo2_rot_s : info.kind == LEAN_HARM ? __longlong_as_double((long long)info.guard_ptr) : 0.0;
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
        S.counts[4 * c + 3] = n_harm | (n_fm << 8);       // the runs of the lean list
    }
}

// Segment s of a long materialisation: the frames [s * seg_frames, ...) of the launch get a record set of their own, laid out
// one after the other in arrays of nseg * nvoices records (counts: nseg * 4 * nchunks).
// (A set holds whole chunks -- 64 * nchunks slots per array: the lists of a chunk are addressed by chunk, 64 c + position, and
// the silent list of the last, partly filled chunk grows down from slot 64 c + 63.  With nvoices slots per set that list wrote
// past the end of the set: into the next segment's first chunk, or, from the last segment, past the array -- found in round 3
// when the arrays of a segmented launch became neighbours in one block.)
__host__ __device__ __forceinline__ size_t set_slots(uint32_t nvoices) { return (size_t)((nvoices + 63) / 64) * 64; }
__device__ __forceinline__ LaunchSet segment_set(const LaunchSet& base, uint32_t s, uint32_t nvoices) {
    const uint32_t nchunks = (nvoices + 63) / 64;
    const size_t stride = set_slots(nvoices);
    LaunchSet r;
    r.launch = base.launch + (size_t)s * stride;
    r.fm = base.fm + (size_t)s * stride;
    r.fast = base.fast + (size_t)s * stride;
    r.gen_idx = base.gen_idx + (size_t)s * stride;
    r.counts = base.counts + (size_t)s * 4 * nchunks;
    return r;
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
//   tile_first, tile_last: first / last frame index any lane of this wave touches (uniform); `start`: the voice's sample index at
//   frame 0 (the record's start_rel)
template <int FPL, bool BANK>
__device__ __forceinline__ void voice_block(const VoiceRegs& r, const VoiceFM* __restrict__ fmrec,
                                            const BankPtrs& B, const sh_voice* __restrict__ vfull,
                                            uint64_t start, uint32_t tile_first, uint32_t tile_last,
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
        const uint32_t lfo_split = fp->lfo_split;
        if (fm_mode == SH_FM_SINE && lfo_split > tile_first && lfo_split <= tile_last) {
            // the LFO's table piece ends inside this tile: every frame from the piece that holds it, by lookup
            const double a2 = fp->lfo2_a_rel, d2 = fp->lfo2_d, K2 = fp->lfo2_K, C2 = fp->lfo2_C0;
#pragma unroll
            for (int j = 0; j < FPL; ++j) {
                const bool behind = i[j] >= lfo_split;
                double ls, lc;
                shm::sincos_tab(fma(di[j], behind ? d2 : f.lfo_d, behind ? a2 : f.lfo_a_rel), trig, ls, lc);
                const double Ln = fma(behind ? K2 : f.lfo_K, (behind ? C2 : f.lfo_C0) - lc, f.lfo_bias * ((double)start + di[j]));
                th[j] = f.frequency * th[j] + fma(f.f_inc, Ln, f.phase0);
            }
        } else if (fm_mode == SH_FM_SINE) {
            // LFO angle a_rel + i*d is exactly linear in i (on one piece of the LFO's table: the first of the launch, or, behind
            // lfo_split, the second): table lookup for the lane's first frame, rotation by 64*d for the following ones
            double rc = fp->lfo_rot_c, rs = fp->lfo_rot_s;
            if (tile_first >= lfo_split) {
                f.lfo_a_rel = fp->lfo2_a_rel; f.lfo_d = fp->lfo2_d; f.lfo_K = fp->lfo2_K; f.lfo_C0 = fp->lfo2_C0;
                shm::sincos_tab(64.0 * f.lfo_d, trig, rs, rc);
            }
            double ls, lc;
            shm::sincos_tab(fma(di[0], f.lfo_d, f.lfo_a_rel), trig, ls, lc);
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

// voice_block for the launch-relative frames of a tile, whatever the voice's onset: the record's start_rel is the voice's sample
// index at the launch's first sounding frame; in the launch that holds the onset (onset_i > 0) the frames in front of it are
// zeros and the voice's own index is i - onset_i.  (Voices that read rows of the launch's matrix have no onset: sh_bank_create.)
template <int FPL, bool BANK>
__device__ __forceinline__ void voice_block_at(const VoiceRegs& r, const VoiceFM* __restrict__ fmrec, const BankPtrs& B,
                                               const sh_voice* __restrict__ vfull, uint32_t tile_first, uint32_t tile_last,
                                               const uint32_t (&i)[FPL], const double (&di)[FPL],
                                               const double* __restrict__ fm_cumsum, const double* __restrict__ pwm,
                                               TrigTab trig, double (&x)[FPL]) {
    const uint64_t st = r.rec->start_rel;
    if (!(r.flags & FL_ONSET)) {
        voice_block<FPL, BANK>(r, fmrec, B, vfull, st, tile_first, tile_last, i, di, fm_cumsum, pwm, trig, x);
        return;
    }
    const uint32_t d = r.rec->onset_i;                      // (uniform)
    if (tile_last < d) {                                    // the whole tile lies in front of the onset
#pragma unroll
        for (int j = 0; j < FPL; ++j) x[j] = 0.0;
        return;
    }
    // the voice's own index of the lane's frames: integer form clamped at 0 for the frames in front of the onset (table walks,
    // row indices); float form NOT clamped -- a lane's frames stay 64 apart, which the rotation shortcuts rely on (the values
    // computed for negative indices are finite nonsense, and zeroed below)
    uint32_t ii[FPL];
    double dd[FPL];
    const double dneg = (double)d;
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        ii[j] = i[j] >= d ? i[j] - d : 0u;
        dd[j] = di[j] - dneg;
    }
    voice_block<FPL, BANK>(r, fmrec, B, vfull, st, tile_first > d ? tile_first - d : 0u, tile_last - d, ii, dd, fm_cumsum, pwm, trig, x);
#pragma unroll
    for (int j = 0; j < FPL; ++j)
        if (i[j] < d) x[j] = 0.0;
}

// ---- the int16 boundary guard (sh_voice::guard_*; include/synthhip.h) --------------------------------------------------------------
// The polynomial and Clenshaw forms of a Harmonics voice lie up to T = guard_t |t| + guard_c beside the reference's term-by-term sum;
// int(scale * v) of the two can differ only where an integer lies between them, i.e. where scale * v is within scale * T of one.
// guard_near: is any sample of the wave's tile that close (or NaN)?  With tq = |scale| T: fract(scale v + tq) <= 2 tq.
template <int FPL>
__device__ __forceinline__ bool guard_near(const double (&x)[FPL], double scale, double tq) {
    bool near = false;
    const double tq2 = tq + tq;
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        // (not around the integer 0: no boundary of a truncation toward zero, and where a waveform is flat at its zero crossing -- the
        //  1/k series of sixteen partials at t = pi -- one sample in 10^4 lies within reach of it: profiles/r06_guard_ab.txt)
        const double y = fma(scale, x[j], tq);
        near |= !(__builtin_amdgcn_fract(y) > tq2) && !(y >= 0.0 && y < 1.0);
    }
    return __ballot(near) != 0ull;                           // (uniform)
}
// |scale| T of a voice's tile that ends at launch-relative frame tile_last; 0 = the voice has no guard (FM, other kinds, no list)
__device__ __forceinline__ double guard_tq(const VoiceRegs& r, const sh_voice* __restrict__ vfull, uint32_t tile_last, double scale) {
    if ((r.flags & (FL_POLY | FL_DENSE)) == 0 || (r.flags & (FL_FM | FL_SILENT)) != 0) return 0.0;
    const sh_voice SH_CONST_AS* v = as_const(vfull);
    if (v->guard_count == 0) return 0.0;
    const double t_end = fabs(r.t_base) + ((double)tile_last + 1.0) * fabs(r.dt);
    return fabs(scale) * (fma(v->guard_t, t_end, v->guard_c) * 1.000001);
}
// the tile again, term by term from the voice's own list (the general code's sparse form: sin(fl(t k)) a_k in list order)
template <int FPL>
__device__ __forceinline__ void voice_block_exact(VoiceRegs r, const VoiceFM* __restrict__ fmrec, const BankPtrs& B,
                                                  const sh_voice* __restrict__ vfull, uint32_t tile_first, uint32_t tile_last,
                                                  const uint32_t (&i)[FPL], const double (&di)[FPL],
                                                  const double* __restrict__ fm_cumsum, const double* __restrict__ pwm,
                                                  TrigTab trig, double (&x)[FPL]) {
    const sh_voice SH_CONST_AS* v = as_const(vfull);
    r.flags &= ~(FL_POLY | FL_DENSE);
    r.harm = reinterpret_cast<const double SH_CONST_AS*>(as_const(B.partials) + v->guard_offset);
    r.harm_cnt = v->guard_count;
    voice_block_at<FPL, false>(r, fmrec, B, vfull, tile_first, tile_last, i, di, fm_cumsum, pwm, trig, x);
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
        voice_block_at<GF, true>(r, fmrec, B, vfull, first, last, ih, dh, fm_p, pwm_p, trig, x);
#pragma unroll
        for (int j = 0; j < GF; ++j) {
            accl[h * GF + j] = fma(r.gain_l, x[j], accl[h * GF + j]);
            accr[h * GF + j] = fma(r.gain_r, x[j], accr[h * GF + j]);
        }
    }
}

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
#ifndef SH_LEAN_CH
#define SH_LEAN_CH 4        // Horner chains of the lean Harmonics arithmetic in flight per wavefront (2: the pairs of rounds 2 and 3)
#endif
template <int FPL, bool SLOPED = false, typename Theta>
__device__ __forceinline__ void lean_harm_frames(double s0, double c0, double s1, double c1, double k2, bool straddle, Theta theta,
                                                 TrigTab trig, const double (&poly)[16], double gl, double gr,
                                                 double (&accl)[FPL], double (&accr)[FPL],
                                                 double gls = 0.0, double grs = 0.0, double dl = 0.0) {
    if constexpr (SH_LEAN_CH > 2 && FPL % SH_LEAN_CH == 0) {
        // Round 4: the Horner chains SH_LEAN_CH at a time instead of in pairs -- the same operations per frame in the same order
        // (bit-identical results), 2 % faster (a dependent float64 FMA issues as fast as an independent one on this chip:
        // profiles/r04_ubench_ilp.txt; what the wider interleave buys is slack for a wavefront that shares its SIMD with fewer
        // others).  The sines and cosines of HB frames at a time, the recurrence running on from the previous batch's last two values.
        constexpr int HB = FPL > 8 ? SH_LEAN_CH : (FPL < 8 ? FPL : 8);      // (sixteen frames per lane: the accumulators hold 64 registers -- four frames' sines at a time)
        static_assert(FPL % HB == 0 && HB % SH_LEAN_CH == 0, "frames per lane: a multiple of the chains, in halves of eight");
        double s[HB], c[HB];
#pragma unroll
        for (int h0 = 0; h0 < FPL; h0 += HB) {
            if (h0 == 0) {
                s[0] = s0; c[0] = c0;
                if (HB > 1) { s[1] = s1; c[1] = c1; }
                if (straddle) {
#pragma unroll
                    for (int j = 2; j < HB; ++j) shm::sincos_tab(theta(j), trig, s[j], c[j]);
                } else {
#pragma unroll
                    for (int j = 2; j < HB; ++j) {
                        s[j] = fma(k2, s[j - 1], -s[j - 2]);
                        c[j] = fma(k2, c[j - 1], -c[j - 2]);
                    }
                }
            } else {
                if (straddle) {
#pragma unroll
                    for (int j = 0; j < HB; ++j) shm::sincos_tab(theta(h0 + j), trig, s[j], c[j]);
                } else {
                    const double sa = s[HB - 2], sb = s[HB - 1], ca = c[HB - 2], cb = c[HB - 1];
                    s[0] = fma(k2, sb, -sa); c[0] = fma(k2, cb, -ca);
                    s[1] = fma(k2, s[0], -sb); c[1] = fma(k2, c[0], -cb);
#pragma unroll
                    for (int j = 2; j < HB; ++j) {
                        s[j] = fma(k2, s[j - 1], -s[j - 2]);
                        c[j] = fma(k2, c[j - 1], -c[j - 2]);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < HB; h += SH_LEAN_CH) {
                double p[SH_LEAN_CH];
#pragma unroll
                for (int jj = 0; jj < SH_LEAN_CH; ++jj) p[jj] = fma(poly[0], c[h + jj], poly[1]);
#pragma unroll
                for (int u = 2; u < 16; ++u) {
#pragma unroll
                    for (int jj = 0; jj < SH_LEAN_CH; ++jj) p[jj] = fma(p[jj], c[h + jj], poly[u]);
                }
#pragma unroll
                for (int jj = 0; jj < SH_LEAN_CH; ++jj) {
                    const double x = p[jj] * s[h + jj];
                    const int f = h0 + h + jj;
                    if (SLOPED) {
                        const double ij = dl + (double)(f * 64);
                        accl[f] = fma(fma(ij, gls, gl), x, accl[f]);
                        accr[f] = fma(fma(ij, grs, gr), x, accr[f]);
                    } else {
                        accl[f] = fma(gl, x, accl[f]);
                        accr[f] = fma(gr, x, accr[f]);
                    }
                }
            }
        }
        return;
    }
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

// The tile set of a tile-classified launch (see TileRec): one wavefront, lane = voice of chunk c, walks run `run` of
// TILES_PER_WAVE tiles in order, classifies each (voice, tile) pair, writes the 128-byte record of the lean ones and, by two
// ballots per tile, the masks of the chunk.
// A lane's chain of dependent round trips IS the time this step takes (nothing hides a lone wavefront's latency, and inside the
// render kernel a prepare workgroup holds a slot the tiles' workgroups want): so the loads come in three batches -- the voice and
// its hint; the starts of the pieces around the hint; a WINDOW of TILE_WIN pieces (start, t0, dt) from the piece that
// holds the run's first frame -- and the tiles of the run are classified from registers: the piece of a tile and the ends of
// pieces inside it are picks from the window, which slides (one more batch) only where a lane runs out of it -- the first
// tiles of a note, whose phase sum runs through a binade every few frames.  (Round 3, first version: a dependent load per
// piece end and tile, three tiles per wave: 32 us as a kernel of its own, and inside the render kernel its workgroups cost the
// launch 28 us.)
#ifndef SH_TPW
#define SH_TPW 3
#endif
constexpr uint32_t TILES_PER_WAVE = SH_TPW;
constexpr int TILE_WIN = 6;
template <typename V>
__device__ __forceinline__ V win_pick(const V (&a)[TILE_WIN], uint32_t r, V beyond) {
    V x = beyond;
#pragma unroll
    for (int k = 0; k < TILE_WIN; ++k) x = r == (uint32_t)k ? a[k] : x;
    return x;
}
// `recs` (a stream of blocks: the record set of that launch; NULL: the launch records have been resolved by a kernel of their own):
// only the general pairs of a tile-classified launch read launch records, so a lane resolves its voice's record HERE, and only if
// one of its tiles came out general -- for a table of notes that is next to never (as a step of its own, a wavefront per sounding
// chunk, the records cost the general kernel its longest chain: 10 .. 19 us where the classification takes 8).
__device__ __forceinline__ void prepare_tiles_wave(const BankPtrs& B, const TileSet& T, uint32_t nvoices, uint64_t start, uint32_t nframes,
                                                   uint32_t ntiles, uint32_t c, uint32_t run, const LaunchSet* recs = nullptr) {
    const uint32_t lane = threadIdx.x & 63, vi = c * 64 + lane;
    const size_t slots = (size_t)T.rec_chunks * 64;
    const uint32_t t_begin = run * TILES_PER_WAVE;
    if (t_begin >= ntiles) return;
    uint32_t t_end = t_begin + TILES_PER_WAVE;
    if (t_end > ntiles) t_end = ntiles;
    const bool valid = vi < nvoices;
    const uint64_t launch_end = start + (uint64_t)nframes;
    {   // a chunk whose voices are all silent throughout the launch (notes that have not started, notes that are over): empty masks
        const uint64_t span_lo = as_const(B.chunk_span)[2 * c], span_hi = as_const(B.chunk_span)[2 * c + 1];
        if (launch_end <= span_lo || start >= span_hi) {
            if (lane == 0)
                for (uint32_t t = t_begin; t < t_end; ++t) {
                    const size_t at = ((size_t)t * T.groups + c % T.groups) * T.mask_k + c / T.groups;
                    T.lean[at] = 0;
                    T.gen[at] = 0;
                }
            return;
        }
    }
    // ---- first batch: the voice, its hint ----
    const sh_voice& v = B.voices[valid ? vi : 0];
    const int32_t v_kind = v.kind, v_dense = v.harm_dense, v_fm = v.fm_mode, v_flip = v.flip;
    const double v_bias = v.bias, amp = v.amplitude;
    const uint32_t off_c = v.seg_offset, cnt_c = v.seg_count, off_t = v.time_seg_offset, cnt_t = v.time_seg_count;
    const uint64_t onset = v.start_frame;
    const uint64_t e_na = v.env.n_attack_end, e_nd = v.env.n_decay_end, e_ns = v.env.n_sustain_end, e_nr = v.env.n_release_end;
    const double e_as = v.env.attack_slope, e_ds = v.env.decay_slope, e_sl = v.env.sustain_level, e_rs = v.env.release_slope;
    const int32_t e_on = v.env.enabled, e_tail = v.env.has_tail;
    const double bgl = (double)v.gain_l, bgr = (double)v.gain_r;
    const uint32_t hint = B.hint[valid ? vi : 0];
    asm volatile("" :: "v"(v_kind), "v"(v_dense), "v"(v_fm), "v"(v_flip), "v"(v_bias), "v"(amp), "v"(off_c), "v"(cnt_c), "v"(off_t), "v"(cnt_t),
                 "v"(onset), "v"(e_na), "v"(e_nd), "v"(e_ns), "v"(e_nr), "v"(e_as), "v"(e_ds), "v"(e_sl), "v"(e_rs), "v"(e_on), "v"(e_tail),
                 "v"(bgl), "v"(bgr), "v"(hint));
    // (what a lean pair can be: polynomial Harmonics, a plain Sine -- the series with one partial -- or a plain waveform; see sh_bank::tile_all)
    const bool v_wave = v_kind == SH_SAWTOOTH || v_kind == SH_SQUARE || v_kind == SH_TRIANGLE || v_kind == SH_PULSE;
    const bool v_fm_sine = v_kind == SH_SINE && v_fm == SH_FM_SINE;
    const bool lean_capable = valid && v_bias == 0.0 && !v_flip &&
                              (v_fm_sine || (v_fm == SH_FM_NONE && ((v_kind == SH_HARMONICS && v_dense == 2) || v_kind == SH_SINE || v_wave)));
    const uint32_t rec_kind = v_fm_sine ? LEAN_FM : v_kind == SH_SAWTOOTH ? LEAN_SAW : v_kind == SH_SQUARE ? LEAN_SQUARE : v_kind == SH_TRIANGLE ? LEAN_TRIANGLE : v_kind == SH_PULSE ? LEAN_PULSE : LEAN_HARM;
    const bool fm = v_fm != SH_FM_NONE;
    const uint32_t off = fm ? off_t : off_c;
    const uint32_t cnt = fm ? cnt_t : cnt_c;
    const sh_segment* tab = B.segs + off;
    const double2* rots = B.seg_rot + off;
    // the voice's own index of the first frame of the run that it sounds in (0: its onset lies in the run or behind it)
    const uint64_t run0 = start + (uint64_t)t_begin * TILE_FRAMES;
    const uint64_t nn_first = run0 > onset ? run0 - onset : 0ull;
    // ---- second batch: the starts of the pieces hint - 1 .. hint + 3: the one that holds nn_first is (nearly always) among them ----
    uint32_t wb;                                  // the table index of the window's first piece
    {
        const uint32_t a = hint < cnt ? hint : cnt - 1;
        const uint32_t a0 = a > 0 ? a - 1 : 0;
        uint64_t s[5];
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) s[k] = a0 + k < cnt ? tab[a0 + k].n0 : ~0ull;
        const uint32_t r = (uint32_t)(s[1] <= nn_first) + (uint32_t)(s[2] <= nn_first) + (uint32_t)(s[3] <= nn_first) + (uint32_t)(s[4] <= nn_first);
        wb = a0 + r;
        if (s[0] > nn_first || (r == 4 && a0 + 5 < cnt)) {     // not among them (a jump, a cold start): search
            uint32_t lo_ = 0, hi = cnt - 1;
            while (lo_ < hi) {
                const uint32_t mid = (lo_ + hi + 1) >> 1;
                if (tab[mid].n0 <= nn_first) lo_ = mid; else hi = mid - 1;
            }
            wb = lo_;
        }
    }
    // ---- third batch: the window ----
    uint64_t w_n0[TILE_WIN];
    double w_t0[TILE_WIN], w_dt[TILE_WIN];
    auto load_window = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)TILE_WIN; ++k) {
            const bool have = wb + k < cnt;
            const uint32_t at = have ? wb + k : cnt - 1;
            const uint64_t n0_ = tab[at].n0;
            w_n0[k] = have ? n0_ : ~0ull;
            w_t0[k] = tab[at].t0;
            w_dt[k] = tab[at].dt;
        }
    };
    load_window();
    // an FM Sine voice's LFO has a table of its own (pairs of records at seg_offset: oscillators.LfoTable): the piece that holds the run's
    // first frame by binary search (FM voices only), then forwards tile by tile
    const uint32_t lnp = (v_fm_sine && cnt_c >= 2) ? cnt_c / 2 : 0u;
    const sh_segment* ltab = B.segs + off_c;
    uint32_t lp = 0;
    if (lnp) {
        uint32_t lo_ = 0, hi_ = lnp - 1;
        while (lo_ < hi_) {
            const uint32_t mid = (lo_ + hi_ + 1) >> 1;
            if (ltab[2 * mid].n0 <= nn_first) lo_ = mid; else hi_ = mid - 1;
        }
        lp = lo_;
    }
    bool any_general = false, any_sound = false;
    uint32_t last_piece = 0;
    for (uint32_t t = t_begin; t < t_end; ++t) {
        const uint64_t abs0 = start + (uint64_t)t * TILE_FRAMES;
        uint64_t abs1 = abs0 + TILE_FRAMES;
        if (abs1 > launch_end) abs1 = launch_end;
        bool is_lean = false, is_gen = false, is_walk = false;
        uint32_t lfo_split_rel = 0;
        double rec_t0 = 0.0, rec_ea0 = 0.0, rec_ea1 = 0.0, rec_eb0 = 0.0, rec_eb1 = 0.0;
        uint32_t rec_extra = 0, rec_corner = 0, r = 0;
        uint16_t rec_split[2] = {0xFFFFu, 0xFFFFu};
        // the voice's own indices [n0, n1) of the tile; n0 < 0: the onset lies inside (or the voice has not started)
        const long long n0 = (long long)abs0 - (long long)onset, n1 = (long long)abs1 - (long long)onset;
        const double dn0 = (double)n0;
        const uint64_t nn0 = n0 > 0 ? (uint64_t)n0 : 0ull;
        bool env_ok = true, walk_ok = true, sounds = valid && abs1 > onset;      // walk_ok: the envelope a WALK pair can carry (see TileRec)
        double wa0 = 1.0, wa1 = 0.0, wb0 = 1.0, wb1 = 0.0;
        uint32_t w_corner = 0;
        if (sounds) {
            // ---- envelope: the line that holds n0 (-1: the silence in front of the onset) and, if one corner lies inside the
            // tile, the line behind it ----
            double ea0 = 1.0, ea1 = 0.0, eb0 = 1.0, eb1 = 0.0;
            uint32_t corner_i = 0;                            // the first frame of the second line, relative to the tile's
            if (e_on) {
                const long long cn[4] = {(long long)e_na, (long long)e_nd, (long long)e_ns, (long long)e_nr};
                const int p = (n0 >= 0) + (n0 >= cn[0]) + (n0 >= cn[1]) + (n0 >= cn[2]) + (n0 >= cn[3]) - 1;
                const int p1 = (n1 - 1 >= 0) + (n1 - 1 >= cn[0]) + (n1 - 1 >= cn[1]) + (n1 - 1 >= cn[2]) + (n1 - 1 >= cn[3]) - 1;     // the tile's last frame
                // line k at the voice's index n, as value at n0 and slope: -1 silence, 0 attack, 1 decay, 2 sustain, 3 release, 4 silence
                auto line = [&](int k, double& v0, double& v1) {
                    if (k == 0) { v0 = dn0 * e_as; v1 = e_as; }
                    else if (k == 1) { v0 = fma(dn0 - (double)cn[0], e_ds, 1.0); v1 = e_ds; }
                    else if (k == 2) { v0 = e_sl; v1 = 0.0; }
                    else if (k == 3) { v0 = fma(dn0 - (double)cn[2], e_rs, e_sl); v1 = e_rs; }
                    else { v0 = 0.0; v1 = 0.0; }
                };
                if (p == 4) {                                // behind the release: silent, but for the one extra sample at its end
                    sounds = e_tail && n0 == cn[3];
                    env_ok = false;
                    walk_ok = false;
                } else if (p == -1 && p1 <= 1) {             // the onset inside: silence | attack [| decay]
                    line(-1, ea0, ea1);
                    line(0, eb0, eb1);
                    corner_i = (uint32_t)(-n0);
                    env_ok = p1 == 0;
                    line(0, wa0, wa1);                       // (a walk pair: the attack, the decay -- or the attack twice; the frames in
                    line(p1, wb0, wb1);                      //  front of the onset are silenced through their angle)
                    w_corner = p1 == 1 ? (uint32_t)(cn[0] - n0) : 0u;
                } else if (p1 == p) {
                    line(p, ea0, ea1);
                    eb0 = ea0; eb1 = ea1;
                } else if (p1 == p + 1) {                    // one corner inside (p >= 0): the reference changes lines at frame cn[p] -- and plays
                    line(p, ea0, ea1);                       // one more sample at the release's end if its accumulated amplitude is still
                    line(p + 1, eb0, eb1);                   // positive (has_tail): the release's line, one frame longer
                    corner_i = (uint32_t)(cn[p] - n0) + ((p == 3 && e_tail) ? 1u : 0u);
                } else {
                    env_ok = false;                          // two corners in one tile (lines of zero length included)
                    walk_ok = false;
                }
            } else if (n0 < 0) {
                env_ok = false;                              // an onset without an envelope is a step: not the meeting of two lines
            }                                                // (a walk pair carries it: the constant 1 behind the onset)
            rec_ea0 = ea0; rec_ea1 = ea1; rec_eb0 = eb0; rec_eb1 = eb1;
            rec_corner = corner_i;
            if (!(e_on && n0 < 0)) { wa0 = ea0; wa1 = ea1; wb0 = eb0; wb1 = eb1; w_corner = corner_i; }
        }
        // ---- phase table: the piece that holds nn0 and the ends of pieces inside the tile, from the window; a lane whose window
        // does not show the three pieces behind its own (and the table goes on) slides it -- with one batch of loads ----
        for (;;) {
            r = 0;
#pragma unroll
            for (int k = 1; k < TILE_WIN; ++k) r += (uint32_t)(w_n0[k] <= nn0);
            const bool slide = sounds && r > (uint32_t)(TILE_WIN - 4) && wb + TILE_WIN < cnt;
            if (__ballot(slide) == 0) break;
            if (slide) {
                wb += r;
                load_window();
            }
        }
        // (the piece's rotation: asked for here, needed at the record's stores)
        const double2 rot = rots[wb + r < cnt ? wb + r : cnt - 1];
        if (sounds) {
            const uint64_t p_n0 = win_pick<uint64_t>(w_n0, r, 0ull);
            const double p_t0 = win_pick<double>(w_t0, r, 0.0), p_dt = win_pick<double>(w_dt, r, 0.0);
            uint64_t starts[TILE_MAX_PIECES];
#pragma unroll
            for (uint32_t k = 0; k < TILE_MAX_PIECES; ++k) starts[k] = win_pick<uint64_t>(w_n0, r + 1 + k, ~0ull);
            uint32_t extra = 0;
#pragma unroll
            for (uint32_t k = 0; k < TILE_MAX_PIECES; ++k) extra += (uint32_t)(starts[k] < (uint64_t)n1);
            // (an FM Sine pair: the piece of the LFO's table that holds the tile's first frame rides in the record, and the frame at which
            //  the next one takes over, should that lie inside the tile)
            if (lnp) {
                while (lp + 1 < lnp && ltab[2 * lp + 2].n0 <= nn0) ++lp;
                lfo_split_rel = (lp + 1 < lnp && ltab[2 * lp + 2].n0 < (uint64_t)n1) ? (uint32_t)((long long)ltab[2 * lp + 2].n0 - n0) : 0u;
            }
            is_lean = lean_capable && env_ok && extra < TILE_MAX_PIECES;
            // more piece ends, or an onset with the attack's end behind it: a WALK pair, if the tile touches at most TILE_WALK_PIECES pieces
            if (lean_capable && !is_lean && walk_ok) {
                const uint32_t far = wb + r + TILE_WALK_PIECES;
                is_walk = far >= cnt || tab[far].n0 >= (uint64_t)n1;
            }
            is_gen = !is_lean && !is_walk;
            if (is_walk) {
                // (the record's fields, re-used: see TileRec)
                rec_ea0 = wa0; rec_ea1 = wa1; rec_eb0 = wb0; rec_eb1 = wb1;
                rec_corner = w_corner;
            }
            if (is_lean) {
                rec_t0 = fma(dn0 - (double)p_n0, p_dt, p_t0);
                rec_extra = extra;
#pragma unroll
                for (uint32_t k = 0; k < 2; ++k) rec_split[k] = k < extra ? (uint16_t)((long long)starts[k] - n0) : (uint16_t)0xFFFFu;
            }
        }
        any_general = any_general || is_gen;
        if (sounds) { any_sound = true; last_piece = wb + r; }
        const uint64_t ml = __ballot(is_lean || is_walk), mg = __ballot(is_gen);
        // the chunk's lean pairs in three runs, each in voice order -- polynomial Harmonics (plain Sines among them), FM Sine, the plain
        // waveforms -- which the render kernel walks with a loop each (tiles_lean; the lean lists of an ordinary launch: prepare_chunk)
        const uint64_t mh = __ballot((is_lean || is_walk) && rec_kind == LEAN_HARM), mm = __ballot((is_lean || is_walk) && rec_kind == LEAN_FM);
        const uint32_t n_harm = (uint32_t)__popcll(mh), n_fm = (uint32_t)__popcll(mm);
        if (is_lean || is_walk) {
            const uint64_t below = (1ull << lane) - 1ull;
            const uint32_t pos = rec_kind == LEAN_HARM ? (uint32_t)__popcll(mh & below)
                               : rec_kind == LEAN_FM ? n_harm + (uint32_t)__popcll(mm & below)
                               : n_harm + n_fm + (uint32_t)__popcll(ml & ~mh & ~mm & below);
            TileRec* __restrict__ q = T.recs + (size_t)t * slots + (c - T.k0 * T.groups) * 64 + pos;
            // (a corner at the tile's end -- the tail sample of a release is its last frame -- is no corner of this tile)
            const uint32_t corner = (rec_corner > 0 && rec_corner < TILE_FRAMES && !(rec_ea0 == rec_eb0 && rec_ea1 == rec_eb1)) ? rec_corner : 0u;
            if (!corner) { rec_eb0 = rec_ea0; rec_eb1 = rec_ea1; }
            double2* __restrict__ q2 = reinterpret_cast<double2*>(q);      // eight 16-byte stores
            q2[0] = make_double2(rec_t0, win_pick<double>(w_dt, r, 0.0));
            // (an FM pair: the voice's own index of the tile's first frame and the LFO's angle there, t0 + (n - n0 - 1/2) d on the LFO's piece)
            double lfo_angle = 0.0;
            if (rec_kind == LEAN_FM)
                lfo_angle = lnp ? fma(dn0 - (double)ltab[2 * lp].n0 - 0.5, ltab[2 * lp].dt, ltab[2 * lp].t0) : fma(dn0 - 0.5, v.lfo_d, v.lfo_a);
            q2[1] = rec_kind == LEAN_FM ? make_double2(dn0, lfo_angle) : rot;
            q2[2] = make_double2(rec_ea0, rec_ea1);
            q2[3] = make_double2(rec_eb0, rec_eb1);
            q2[4] = make_double2(amp * bgl, amp * bgr);
            union { uint64_t u; double d; } walk_bits;
            walk_bits.u = (uint64_t)(off + wb + r) | ((uint64_t)(off + cnt) << 32);
            q2[5] = is_walk ? make_double2(dn0, walk_bits.d)
                            : make_double2(rec_extra > 0 ? win_pick<double>(w_t0, r + 1, 0.0) : 0.0, rec_extra > 1 ? win_pick<double>(w_t0, r + 2, 0.0) : 0.0);
            q2[6] = make_double2(rec_extra > 0 ? win_pick<double>(w_dt, r + 1, 0.0) : 0.0, rec_extra > 1 ? win_pick<double>(w_dt, r + 2, 0.0) : 0.0);
            union { uint16_t h[4]; double d; } tail;
            tail.h[0] = rec_split[0]; tail.h[1] = rec_split[1]; tail.h[2] = is_walk ? (uint16_t)0 : (uint16_t)(1 + rec_extra); tail.h[3] = (uint16_t)corner;
            union { uint32_t u[2]; double d; } kind_bits;
            // (low byte: the kind; an FM pair: above it the index in `segs` of the LFO's piece, 0xFFFFFF = a constant LFO, none.
            //  high half: the voice's position in its chunk -- where its polynomial lives)
            kind_bits.u[0] = rec_kind | (rec_kind == LEAN_FM ? ((lnp ? off_c + 2 * lp : 0xFFFFFFu) << 8) : 0u);
            kind_bits.u[1] = lane | (lfo_split_rel << 8);      // (above the position: an FM pair's frame at which the LFO's next piece takes over, 0 = none)
            q2[7] = make_double2(tail.d, kind_bits.d);
        }
        if (lane == 0) {
            const size_t at = ((size_t)t * T.groups + c % T.groups) * T.mask_k + c / T.groups;
            T.lean[at] = (uint64_t)(uint32_t)__popcll(ml) | ((uint64_t)n_harm << 32) | ((uint64_t)n_fm << 40);
            T.gen[at] = mg;
        }
    }
    if (recs) {
        if (valid && any_general) {
            PrepInfo info;
            prepare_voice<false>(B, 0u, vi, start, nframes, recs->launch, recs->fm, info);       // (sets the voice's hint as well)
        } else if (valid && any_sound) {
            B.hint[vi] = last_piece;                             // where the next block's window starts looking
        }
        if (run == 0 && lane == 0) {                             // (sh_bank_launch_stats of a tile-classified launch: not by voice)
            recs->counts[4 * c] = 0; recs->counts[4 * c + 1] = 0; recs->counts[4 * c + 2] = 0; recs->counts[4 * c + 3] = 0;
        }
    }
}


}  // namespace shosc
