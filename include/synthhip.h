/*
 * synthhip.h -- C ABI of libsynthhip.so: the MI355X (gfx950) implementation of
 * synthplayer's oscillator-bank + sample-mixing hot path.
 *
 * The reference (irmen/synthesizer, package `synthplayer`) is pure Python and has
 * NO native/FFI boundary of its own; its "plugin interface" for this path is the
 * public Python class API.  The tree mounted at /root/reference holds only a
 * relocation notice (/root/reference/README.md:1-2), so the upstream symbols are
 * named here without line numbers:
 *
 *   synthplayer/oscillators.py  Oscillator.blocks(), Sine, Sawtooth, Square, Pulse,
 *                               Harmonics, fm_lfo= / pwm_lfo=, EnvelopeFilter
 *   synthplayer/sample.py       Sample.from_osc_block (quantise), Sample.mix,
 *                               Sample.mix_at, Sample.resample
 *   synthplayer/playback.py     mixer loop: repeated audioop.add over voice chunks
 *   CPython 3.10 Modules/audioop.c  audioop.add, audioop.ratecv (the arithmetic
 *                               Sample.mix / Sample.resample delegate to)
 *
 * Each entry point below says which of those it replaces.  INTEGRATION.md shows the
 * ctypes binding a synthplayer maintainer would add.
 *
 * Conventions
 *   - every function returns SH_OK (0) or a negative sh_status; nothing throws across
 *     the ABI; sh_last_error() returns a thread-local message for the last failure.
 *   - the caller owns all host memory; the library owns device memory behind sh_buf /
 *     sh_bank handles.  No callbacks.  Plain pointers and sizes only.
 *   - one HIP stream per process (created by sh_init).  Calls may come from any thread:
 *     each entry point runs under one process-wide lock (the real-time mixer is fed from
 *     one thread and drained from another), so calls serialise -- in the order the lock is
 *     won, which is also their order on the stream.  Kernel launches are asynchronous;
 *     functions that copy to host memory, and sh_sync(), synchronise (holding the lock).
 *   - "frame" = one sample period (all channels); PCM is interleaved, little endian.
 */
#ifndef SYNTHHIP_H
#define SYNTHHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sh_status {
    SH_OK = 0,
    SH_ERR_INVALID = -1,   /* bad argument */
    SH_ERR_HIP = -2,       /* HIP runtime error (message has the hipError string) */
    SH_ERR_NOMEM = -3,
    SH_ERR_NOTINIT = -4,   /* sh_init not called / no GPU */
    SH_ERR_OVERFLOW = -5,  /* quantise: a sample does not fit the PCM width (OverflowError upstream) */
    SH_ERR_RCCL = -6,
    SH_ERR_LENGTH = -7     /* audioop.add: "Lengths should be the same" */
} sh_status;

typedef enum sh_kind {     /* oscillators.py class names */
    SH_SINE = 0, SH_SAWTOOTH = 1, SH_SQUARE = 2, SH_PULSE = 3, SH_HARMONICS = 4, SH_TRIANGLE = 5,
    SH_LINEAR = 6,         /* Linear: the sample IS the accumulated value of the phase table (level += increment until
                            * it leaves (min, max); the host ends the table with a constant piece there) */
    SH_NOISE = 7,          /* WhiteNoise: sample-and-hold uniform noise from a counter-based generator, below */
    SH_BUFFER = 8          /* a voice whose float64 samples were rendered elsewhere (MixingFilter, EchoFilter, nested envelopes ...):
                            * row fm_row[voice] of sh_bank_render_rows' matrix; the fused envelope and the bus gains still apply */
} sh_kind;

typedef enum sh_fm_mode {
    SH_FM_NONE = 0,        /* non-FM branch of blocks(): t accumulates f/sr, phase table below */
    SH_FM_SINE = 1,        /* fm_lfo is a plain (non-FM) Sine: closed-form running sum of the LFO */
    SH_FM_BUFFER = 2       /* any fm_lfo: running sum supplied as a device buffer (sh_osc_render only) */
} sh_fm_mode;

/* One piece of an accumulated-phase table: for n0 <= n < next.n0 the reference's
 * running `t` (t += inc in float64) equals fma(n - n0, dt, t0) exactly. */
typedef struct sh_segment {
    uint64_t n0;
    double   t0;
    double   dt;
} sh_segment;

/* One (k, amp) pair of Harmonics(harmonics=[...]) -- used when the list is not dense. */
typedef struct sh_partial {
    double k;
    double amp;
} sh_partial;

/* EnvelopeFilter, reduced on the host to integer sample boundaries (the host replays the
 * reference's accumulated `time` exactly) and per-phase slopes. */
typedef struct sh_envelope {
    uint64_t n_attack_end;     /* [0, n_attack_end): gain = n*attack_slope */
    uint64_t n_decay_end;      /* [.., n_decay_end): gain = 1 + (n-n_attack_end)*decay_slope */
    uint64_t n_sustain_end;    /* [.., n_sustain_end): gain = sustain_level */
    uint64_t n_release_end;    /* [.., n_release_end): gain = sustain_level + (n-n_sustain_end)*release_slope */
    double   attack_slope, decay_slope, sustain_level, release_slope;
    double   tail_amp;         /* gain of the one extra sample at n_release_end when has_tail */
    int32_t  enabled;
    int32_t  has_tail;
} sh_envelope;

typedef struct sh_voice {
    int32_t  kind;             /* sh_kind */
    int32_t  fm_mode;          /* sh_fm_mode */
    double   amplitude, bias, pulsewidth;
    /* SH_FM_NONE: carrier phase table (t in radians for Sine/Harmonics, in turns otherwise).
     * SH_FM_SINE with a Sine LFO that moves (lfo_K != 0): the table of the LFO's running sum, TWO records per piece of the LFO's own
     * accumulated phase -- (n0, t0, dt) and (n0, K_p, C_p): on samples n0_p <= n < n0_{p+1},
     * L(n) = K_p * (C_p - cos(t0_p + (n - n0_p - 1/2) * dt_p)) + lfo_bias * n  (oscillators.LfoTable: the prefix sums of the pieces in
     * front are folded into C_p); seg_count = 0: a constant LFO, lfo_a .. lfo_C0 below are all there is. */
    uint32_t seg_offset, seg_count;
    /* Harmonics, by harm_dense:
     *   0 sparse  -> harm_count sh_partial at partial[harm_offset], summed term by term
     *   1 dense   -> harm_count Clenshaw coefficients (doubles, k = harm_count..1, count a multiple of 8)
     *                at coef[harm_offset]
     *   2 poly    -> 16 doubles at coef[harm_offset], highest power first: sum_k a_k sin(k t) =
     *                sin(t) * P(cos t), P of degree 15 (all k <= 16; converted exactly on the host) */
    uint32_t harm_offset, harm_count;
    int32_t  harm_dense;
    int32_t  flip;             /* SawtoothH: the sample is mirrored around the bias, value = bias*2.0 - value */
    /* FM: theta_n = fm_phase0 + frequency*T_n + frequency*fm_inc*L(n), T = shared time table
     * (t += fm_inc from 0), L(n) = sum_{j<n} lfo_j */
    double   frequency, fm_phase0, fm_inc;
    uint32_t time_seg_offset, time_seg_count;
    /* SH_FM_SINE: lfo_j = lfo_amp*sin(phase_j) + lfo_bias, phase_j the LFO's accumulated phase (lfo_a, += lfo_d per sample in float64:
     * piecewise exactly linear -- the table at seg_offset); on the ideal line phase_j = lfo_a + j*lfo_d (the first piece, and all
     * there is for a library older than ABI 5):
     * L(n) = lfo_K*(lfo_C0 - cos(lfo_a + (n-0.5)*lfo_d)) + lfo_bias*n,
     * lfo_K = lfo_amp/(2 sin(lfo_d/2)), lfo_C0 = cos(lfo_a - lfo_d/2).  ABI 5: lfo_bias must be 0 -- the bias rides in `frequency`
     * (f * (1 + bias)) -- and lfo_K != 0 needs the table; sh_bank_create rejects anything else. */
    double   lfo_a, lfo_d, lfo_amp, lfo_bias, lfo_K, lfo_C0;
    sh_envelope env;
    float    gain_l, gain_r;   /* stereo bus gains (bank only) */
    /* SH_NOISE: sample n holds value number h = n / noise_hold; u = (splitmix64(noise_seed + h * 0x9E3779B97F4A7C15) >> 11)
     * * 2^-53 in [0, 1); value = (-amplitude + (2*amplitude)*u) + bias   (random.uniform(-a, a) + bias) */
    uint64_t noise_seed;
    uint32_t noise_hold;       /* samples per held value, >= 1: int(samplerate / frequency) */
    uint32_t reserved0;
    /* Onset: the voice is silent in frames n < start_frame and plays its sample n - start_frame from there (phase tables,
     * envelope boundaries, FM sums and noise counters all count from the onset): upstream's DelayFilter(voice, seconds) with
     * seconds >= 0, start_frame = int(samplerate * seconds), fused into the voice.  0 for voices that read rows. */
    uint64_t start_frame;
    /* ABI 6 -- the int16 boundary guard of a Harmonics voice in the polynomial or Clenshaw form (harm_dense 2 / 1).  Those forms sum
     * a_k sin(k t) of the EXACT products k t where the reference rounds every `t * k` before its sine: the float64 samples part
     * ways by up to guard_t * |t| + guard_c (t: the accumulated phase), far inside the float contract -- but where int(scale * v)
     * of such a sample lies that close to an integer, the int16 routes (sh_bank_generate_i16, sh_bank_mixdown_i16) recompute it
     * term by term from the voice's own list, guard_count sh_partial at partial[guard_offset] in the order of the reference's
     * loop: equal integers whatever the time into the note.  guard_count = 0: no guard (the sample is quantised as it is). */
    double   guard_t, guard_c;
    uint32_t guard_offset, guard_count;
} sh_voice;

typedef struct sh_devinfo {
    char     name[128];
    char     arch[32];
    int32_t  compute_units;
    int32_t  clock_mhz;
    uint64_t hbm_bytes;
    int32_t  wavefront;
    int32_t  device;
} sh_devinfo;

typedef struct sh_buf  sh_buf;   /* device buffer */
typedef struct sh_bank sh_bank;  /* voice table resident in HBM */

/* ---- lifecycle -------------------------------------------------------------------- */
int  sh_init(int device);                 /* select GPU, create the stream; idempotent */
int  sh_shutdown(void);
int  sh_is_initialized(void);
int  sh_device_count(void);               /* 0 when no GPU is visible; never fails */
int  sh_device_info(sh_devinfo* out);
int  sh_device_pci(char* out, int n);     /* PCI bus id ("0000:c1:00.0") of the device sh_init selected, NUL-terminated */
const char* sh_last_error(void);
const char* sh_version(void);
/* The binary interface this library was compiled with, so that a binding can refuse a library whose structs it would mis-pack
 * (a stale .so loaded by path: same symbols, other layouts): out[0] = SH_ABI_VERSION, then sizeof of sh_segment, sh_partial,
 * sh_envelope, sh_voice, sh_devinfo, sh_counters.  Writes min(n, 7) words, returns 7.  Callable before sh_init, without a GPU. */
#define SH_ABI_VERSION 6
int  sh_abi(uint32_t* out, int n);
int  sh_sync(void);                       /* wait for the stream */

/* What the library did behind the caller's back since sh_init: driver allocations (hipMalloc: pool misses, growing blocks),
 * driver frees, host-side stream synchronisations it inserted on its own (NOT the caller's sh_sync / downloads / timers), and
 * buffers served from the pool.  A streaming caller in its steady state -- blocks of lengths it has rendered before, buffers
 * of sizes it has used before -- must leave the first three unchanged (tests/test_gpu_pipeline.py asserts it).  Then which shape
 * the bank renders took: launches cut into segments at shared envelope corners (RENDER_*_SEG), launches classified per (voice,
 * tile) (RENDER_*_TILES: banks whose notes do not move in lock-step), and of those the ones whose tile set had been resolved
 * two launches ahead -- so that a test can tell that the path it means to exercise is the one that ran. */
typedef struct sh_counters {
    uint64_t device_allocs, device_frees, stream_syncs, pool_hits;
    uint64_t segmented_launches, tiled_launches, tiled_predicted;
} sh_counters;
int  sh_debug_counters(sh_counters* out);

/* Measurement knobs: environment variables read ONCE, by sh_init.  None changes a result -- they choose between schedules of the same
 * arithmetic, so that A/B timings can be taken with one library (tests/test_gpu_pipeline.py::test_knobs_change_no_result).  The
 * fifteen others that rounds 1-3 accumulated selected paths that had lost their A/B and were removed with those paths in
 * round 4 (CHANGELOG.md keeps what each measured):
 *   SYNTHHIP_NO_OVERLAP=1         consecutive renders of a bank stay on one stream
 *   SYNTHHIP_NO_SPECULATION=1     launch records by a prepare kernel in front of every render (no records two launches ahead)
 *   SYNTHHIP_NO_SMALL_PIPELINE=1  single-group (small) banks render on one stream; several-group banks still alternate
 *   SYNTHHIP_NO_SPLIT=1           lean and general code in one render kernel (k_render_combined) instead of k_render_lean + k_render_general
 *   SYNTHHIP_NO_SEG=1             transition launches / the heads of materialised rows are not cut into segments
 *   SYNTHHIP_NO_TILES=1           banks whose notes do not move in lock-step (onsets, envelopes of their own) are not classified tile
 *                                 by tile: every voice that holds an onset or a corner in the launch takes the general code
 *   SYNTHHIP_VARIANT=WFM          render kernel shape: waves per workgroup, frames per lane, min waves per SIMD -- one of 4163, 484, 444,
 *                                 844, 821, 421, 211 (the shapes the library chooses between by itself; 4163 -- sixteen frames per lane --
 *                                 exists for split launches of polynomial-Harmonics banks and of FM Sine banks only)
 *   SYNTHHIP_GROUPS=n             voice groups of a render launch
 *   SYNTHHIP_POOL_FILL=0..255     device blocks that grow are filled with this byte first (diagnostics: a kernel that reads what it
 *                                 should have written shows)
 *   (round 6)
 *   SYNTHHIP_SELF=1|2|3           a render that stands alone resolves its records (1, 3) / folds its partial buses (1, 2) inside its one kernel
 *                                 (off: measured slower, profiles/r06_run_lengths.txt)
 *   SYNTHHIP_NO_LADDER=1          a render launch that nothing runs beside keeps its wavefronts at one priority (default: the priority falls as a
 *                                 wavefront gets on with its voice lists, profiles/r06_prio_ladder.txt)
 *   SYNTHHIP_RT_CUS=n             the last n compute units are kept for real-time lanes (sh_rt_*); 0 = off (profiles/r06_rt_lane.txt)
 *   SYNTHHIP_NO_PERIOD=1          16-bit mono resampling between rates with a short period (reduced outrate <= 2048: 44.1 / 48 / 96 kHz ...)
 *                                 goes through k_resample_small like any other pair of rates, not through k_resample_period_i16
 *   SYNTHHIP_PERIOD_CHUNKS=n      consecutive chunks per workgroup of k_resample_period_i16 (default: 1, 2 or 4 by the rates' ratio;
 *                                 profiles/r06_resample_period.txt)
 * (SYNTHHIP_LIB, read by the Python binding, names another build of this library to load; SYNTHHIP_ALLOW_STALE=1 lets it load a
 * library whose sources have changed when rebuilding fails.) */

/* Readings of the recalled reference arithmetic that the DEVICE has to know about (the others are host-side choices of
 * synthesizer_amd/params.py `variants`: increments, Square as a Pulse record, pulse widths, envelope boundaries -- they change the records,
 * not the kernels).  SH_OPT_QUANTISE_ROUND: 0 (default) Sample.from_osc_block quantises int(scale * v), truncation toward zero;
 * 1: round(scale * v), half to even (Python 3's round) -- sh_quantize_f32 / sh_quantize_f64 follow it (the saturating [SPEC] forms,
 * sh_quantize_clip_f32 and sh_bank_render_pcm, and the fused sh_bank_generate_i16 do not: under 1 the latter refuses and the caller
 * quantises float64 rows).  Process-wide, like the stream.  Returns SH_ERR_INVALID for an unknown option. */
typedef enum sh_option { SH_OPT_QUANTISE_ROUND = 1,
                         SH_INFO_LAST_MIXDOWN_FUSED = 2   /* read-only (sh_get_option): how many stretches of the last sh_bank_mixdown_i16
                                                           * call were folded where the samples are made (0: all through int16 rows) */
} sh_option;
int  sh_set_option(int option, int value);
int  sh_get_option(int option);

/* ---- device buffers ---------------------------------------------------------------- */
int    sh_buf_alloc(size_t bytes, sh_buf** out);
int    sh_buf_free(sh_buf* b);
/* a non-owning window [offset, offset+bytes) of `parent` (must be freed before the parent) */
int    sh_buf_view(sh_buf* parent, size_t offset, size_t bytes, sh_buf** out);
size_t sh_buf_size(const sh_buf* b);
void*  sh_buf_devptr(sh_buf* b);
int    sh_buf_upload(sh_buf* b, size_t offset, const void* host, size_t bytes);
int    sh_buf_download(const sh_buf* b, size_t offset, void* host, size_t bytes);
int    sh_buf_fill_zero(sh_buf* b, size_t offset, size_t bytes);
int    sh_buf_copy(sh_buf* dst, size_t dst_off, const sh_buf* src, size_t src_off, size_t bytes);

/* ---- timing (HIP events on the library's stream) ------------------------------------ */
int  sh_timer_start(void);
int  sh_timer_stop(float* elapsed_ms);    /* records, synchronises, returns elapsed */

/* ---- voice bank: the voice table (oscillator parameters + phase tables) resident in HBM.
 *      A bank of ONE voice backs a Python Oscillator object; a bank of N voices backs the
 *      mixer sum bus over oscillator voices. ------------------------------------------- */
int sh_bank_create(const sh_voice* voices, uint32_t nvoices,
                   const sh_segment* segs, uint32_t nsegs,
                   const double* coefs, uint32_t ncoefs,
                   const sh_partial* partials, uint32_t npartials,
                   sh_bank** out);
int sh_bank_destroy(sh_bank* b);
uint32_t sh_bank_nvoices(const sh_bank* b);
/* How the last sh_bank_render launch classified the voices: *nfast took the lean loop (polynomial Harmonics, no FM,
 * constant envelope over the launch, one phase-table piece), *ngeneral the general code; voices released before the
 * launch are in neither count.  Synchronises the stream.  (A tile-classified launch -- a table of notes -- classifies per (voice, tile),
 * not per voice: both counts are 0; sh_debug_counters tells which shape a launch took.) */
int sh_bank_launch_stats(sh_bank* bank, uint32_t* nfast, uint32_t* ngeneral);

/* ---- oscillators: replaces Oscillator.blocks() of Sine/Sawtooth/Square/Pulse/Harmonics
 *      (+fm_lfo, +pwm_lfo) and EnvelopeFilter.blocks() ---------------------------------
 * Renders samples [start, start+n) of voice `voice` of the bank.  The call is stateless: the
 * reference's generator state is (sample index, accumulated t), and the phase tables
 * reproduce the accumulated t from the index.
 *   fm_cumsum    SH_FM_BUFFER voices: device buffer of >= n doubles, L(start) .. L(start+n-1)
 *                (L(m) = sum_{j<m} lfo_j, see sh_scan_f64); else NULL
 *   pwm          Pulse with pwm_lfo: device buffer of n doubles (pulse width per sample); else NULL
 *   out_host     host destination, n floats, or NULL
 *   out_f32/off  device destination (float index `off`), or NULL
 *   out_f64      device destination receiving the samples as float64 (modulators), or NULL
 */
int sh_osc_render(sh_bank* bank, uint32_t voice,
                  const sh_buf* fm_cumsum, const sh_buf* pwm,
                  uint64_t start, uint32_t n,
                  float* out_host, sh_buf* out_f32, size_t out_off, sh_buf* out_f64);

/* ---- filters over rendered oscillator blocks: MixingFilter (a+b),
 *      AmpModulationFilter (a*b), ClipFilter (max(min(a, p1), p0)), AbsFilter (|a|), copy / constant fill.
 *      a, b, out_f64: float64 device buffers; out_f32 (+ element offset) / out_host: optional float32 copies */
typedef enum sh_ew_op { SH_EW_ADD = 0, SH_EW_MUL = 1, SH_EW_CLIP = 2, SH_EW_ABS = 3, SH_EW_COPY = 4, SH_EW_FILL = 5,
                        SH_EW_AXPY = 6 /* a + b*p0 (product rounded first): EchoFilter */,
                        SH_EW_NEXTUP = 7 /* the float64 successor of a: pulse widths under the `<=` reading (below) */ } sh_ew_op;
int sh_ew_f64(int op, const sh_buf* a, size_t a_off, const sh_buf* b, size_t b_off, size_t n, double p0, double p1,
              sh_buf* out_f64, size_t out64_off, sh_buf* out_f32, size_t out32_off, float* out_host);

/* exclusive running sum of n float64 values on the device (FM with an arbitrary fm_lfo):
 * out[i] = carry_in + x[0] + .. + x[i-1], i = 0..n-1 (n values).
 * carry_out (host, may be NULL) receives carry_in + x[0] + .. + x[n-1]. */
int sh_scan_f64(const sh_buf* x, uint32_t n, double carry_in, sh_buf* out, double* carry_out);

/* ---- voice bank rendering: N voices -> stereo bus (the "Mixer sum bus" over oscillator voices) */
/* materialise: voices_out[v*stride + i] = voice v at frame start+i (float32, voice-major); nframes <= 2^32 - 65536 */
int sh_bank_generate(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* voices_out, size_t stride);
/* fused generate-and-mix: bus_f32[i] = (sum_v gl_v x_v[i], sum_v gr_v x_v[i]), float32 x2 interleaved.
 * bus_f64 (optional) receives the float64 partial bus (frames x 2) used for the multi-GPU reduce.
 * Work is enqueued like every other call.  One detail matters to code that reads the bus buffers with its OWN
 * kernels through sh_buf_devptr: consecutive renders of a bank (same block length, the next block each time) form a
 * run that is pipelined (every bank has its own: banks rendering turn by turn do not end each other's runs) -- they
 * alternate between two HIP streams so that one launch fills the tail of the other, and
 * with several voice groups the last step of a render (folding the groups' partial buses into the bus) is done inside
 * the render kernel two launches later.  Any other call into the library (sh_sync, sh_buf_download, sh_buf_free, ...;
 * not sh_buf_alloc) ends the run first: it joins the streams and folds what is outstanding -- so such code must call
 * sh_sync() (or any other entry point) before touching the buffers.  Everything inside the library sees completed
 * buses.  To keep a run going while consuming its output, render into a ring of bus buffers and read a buffer only
 * after the run has been ended.
 * Any nframes is accepted: renders beyond 2^22 frames are carried out as a run of launches of 2^22 frames each into
 * consecutive parts of the buffers (a launch's grid holds fewer than 2^32 work-items per dimension, and the partial
 * buses of the voice groups are sized per launch). */
int sh_bank_render(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* bus_f32, sh_buf* bus_f64);
/* The same render delivered as saturated int16 stereo PCM (frames x 2, interleaved): sample = trunc(scale * bus) of the
 * float32 bus, clamped to the int16 range -- exactly what sh_quantize_clip_f32 makes of sh_bank_render's bus_f32 -- but
 * produced by the fold of the partial buses itself, so a stream of PCM blocks (what a player consumes: upstream's
 * synth -> Sample.from_osc_block -> mixer route) stays on the two-stream pipeline described above instead of ending
 * the run with a quantise kernel after every block.  The same rule about reading the buffer applies. */
int sh_bank_render_pcm(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* pcm_i16);

/* A run of blocks per call: blocks k = 0 .. nblocks - 1 of nframes frames each, block k = frames [start + k nframes, +nframes),
 * into bus_f32[k % nring] (float32 stereo) and / or pcm_i16[k % nring] (saturated int16 stereo, as sh_bank_render_pcm) -- what a
 * caller streaming blocks through a ring of buffers does with one sh_bank_render per block, with ONE crossing of the ABI, one
 * acquisition of the library's lock, and ONE LAUNCH for every stretch of blocks whose ring buffers lie back to back in memory
 * (windows of one allocation, sh_buf_view): a launch costs the host 2-4 us whatever it renders, and a small bank's one-second block
 * costs the device less (BASELINE config 2: 64 voices).  Blocks of a merged stretch are one render of the whole stretch: the same
 * samples as block-by-block renders up to the float64 summation order of the voice groups (a longer launch may use fewer groups).
 * Either ring may be NULL (not both).  The rule about reading the buffers is sh_bank_render's. */
int sh_bank_render_run(sh_bank* b, uint64_t start, uint32_t nframes, uint32_t nblocks, sh_buf* const* bus_f32, sh_buf* const* pcm_i16,
                       uint32_t nring, double pcm_scale);

/* ---- banks whose voices are modulated by arbitrary oscillators (fm_lfo= / pwm_lfo= any Oscillator, upstream oscillators.py)
 *      or ARE arbitrary oscillator graphs (filters): the modulators / sources are rendered as float64 rows of one matrix
 *      [row][row_stride] (launch-relative: element i = frame start + i) and the bank's general code reads them.
 * sh_bank_set_rows: per voice, the row holding L(start + i) = running sum of its fm_lfo (SH_FM_BUFFER voices) or its samples
 *   (SH_BUFFER voices) in fm_row, the row holding its pulse width per sample in pwm_row (Pulse with a pwm_lfo); -1 = none.
 * sh_bank_generate_f64: every voice of a bank of closed-form oscillators as float64 rows row0 .. row0 + nvoices - 1 (one launch
 *   for all the modulators of a bank).
 * sh_scan_rows_f64: rows row0 .. row0 + nrows - 1 replaced by their exclusive running sums, row r continuing from carry[r]
 *   (device buffer of nrows doubles, updated to the carry-out: consecutive blocks chain without a host round trip).
 * sh_bank_render_rows: sh_bank_render with that matrix. */
int sh_bank_set_rows(sh_bank* b, const int32_t* fm_row, const int32_t* pwm_row);
int sh_bank_generate_f64(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* rows_out, size_t row0, size_t row_stride);
int sh_scan_rows_f64(sh_buf* rows, size_t row0, uint32_t nrows, uint32_t n, size_t row_stride, sh_buf* carry);
int sh_bank_render_rows(sh_bank* b, uint64_t start, uint32_t nframes, const sh_buf* rows_f64, size_t row_stride,
                        sh_buf* bus_f32, sh_buf* bus_f64);
/* sh_bank_generate (the reference-shaped two-step route: every voice as a float32 row) for such a bank */
int sh_bank_generate_rows(sh_bank* b, uint64_t start, uint32_t nframes, const sh_buf* rows_f64, size_t row_stride,
                          sh_buf* voices_out, size_t stride);

/* The reference-shaped INTEGER route (upstream synthplayer/sample.py Sample.from_osc_block, [RECALL]: one oscillator block ->
 * int(amplitude_scale * v) per sample through array('h'), OverflowError where a value does not fit): every voice of the bank as a
 * row of int16 PCM, voices_out[v*stride + i] = int(scale * x_v[start + i]) -- the float64 sample, the float64 product, truncation
 * toward zero -- made where the sample is made: 2 bytes per voice-sample reach HBM instead of a float row (4 written) that a
 * quantise pass re-reads (4) and re-writes (2).  The rows are what sh_mix_chain_i16 / sh_mix_chain_pan_i16 fold (the mixer's
 * audioop.add chain).  stride must be even (rows are written as 32-bit pairs).  SH_ERR_OVERFLOW when a sample of ANY voice does not
 * fit (the rows are unspecified then); the call synchronises to learn it, like the quantisers.  _rows_: the same for a bank with
 * modulation rows (sh_bank_set_rows). */
int sh_bank_generate_i16(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* voices_out, size_t stride);
/* The streaming form: enqueues the same work and returns; a sample that does not fit raises a flag on the device that stays up until
 * sh_overflow_check() -- which waits for the stream, returns SH_ERR_OVERFLOW if any sh_bank_generate_i16_async since the last
 * check met one, and lowers the flag -- is called: once per batch of blocks instead of a host round trip per block.  The flag is
 * ONE word per process: the synchronous forms (sh_bank_generate_i16, sh_bank_mixdown_i16) read and lower it too, so a synchronous
 * call consumes -- and reports as its own -- an overflow still pending from an _async call of any bank; a synchronous call that
 * fails with another error lowers it as well, so that nothing it raised is left for the next call. */
int sh_bank_generate_i16_async(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* voices_out, size_t stride);
int sh_overflow_check(void);
/* The MONO mixdown the reference's mixer makes of the bank -- every voice quantised like sh_bank_generate_i16, then mixed =
 * audioop.add(mixed, voice, 2) down the voices in order -- without the rows: out[i], nframes int16 samples, equal byte for byte to
 * sh_mix_chain_i16 of sh_bank_generate_i16's rows.  Where every voice of a 65 536-frame stretch takes the lean polynomial-Harmonics loop
 * the samples enter the chain where they are made (a saturating chain over a range of voices is the map x -> clamp(x + a, L, U); such
 * maps compose in voice order: the kernel leaves one per frame and (64-voice chunk, half), a second small kernel applies them); the
 * other stretches go through int16 rows in a temporary and the chain kernel.  SH_ERR_OVERFLOW / _async / sh_overflow_check: as above. */
int sh_bank_mixdown_i16(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* out_i16);
int sh_bank_mixdown_i16_async(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* out_i16);
int sh_bank_generate_rows_i16(sh_bank* b, uint64_t start, uint32_t nframes, const sh_buf* rows_f64, size_t row_stride,
                              double scale, sh_buf* voices_out, size_t stride);

/* ---- mixer sum bus over materialised voices ------------------------------------------ */
/* float32: bus[i] = sum_v gains[v] * voices[v*stride+i]; gains = device buffer of nvoices x (l, r) floats */
int sh_mix_bus_f32(const sh_buf* voices, uint32_t nvoices, size_t stride, uint32_t nframes,
                   const sh_buf* gains_lr, sh_buf* bus_f32);
/* integer: the reference mixer's fold, mixed = add(add(c0, c1), c2) ... in voice order,
 * saturating at every step (playback.py mixer loop -> audioop.add).  chunks[v*stride + i] int16. */
int sh_mix_chain_i16(const sh_buf* chunks, uint32_t nvoices, size_t stride, uint32_t nsamples, sh_buf* out);
/* The same fold over MONO voice rows that enter it as Sample.stereo(left_factor, right_factor) of themselves (upstream
 * synthplayer/sample.py Sample.stereo -> audioop.tostereo(frames, 2, lf, rf), [RECALL]): out frame i = (L, R) with
 * L = add(add(tostereo_l(c0), tostereo_l(c1)), ...) -- per voice and channel fbound(sample * factor) (clamp, then floor), then the
 * saturating chain in voice order.  The stereo rows are never materialised: 2 bytes are read per voice-sample, 4 written per frame.
 * factors_lr: device buffer of nvoices x (left, right) doubles.  chunks[v*stride + i] int16 mono; out: nframes x 2 int16. */
int sh_mix_chain_pan_i16(const sh_buf* chunks, uint32_t nvoices, size_t stride, uint32_t nframes, const sh_buf* factors_lr, sh_buf* out);
/* The same fold with every chunk read where its sample lives (no staging copy): source v contributes
 * srcs[v][sample_offsets[v] .. +nsamples_each[v]) and silence after that, up to nsamples; the result goes to
 * out[out_sample_off ..).  This is one turn of the real-time mixer's chunk loop (upstream synthplayer/playback.py
 * RealTimeMixer.chunks, [RECALL]: next(chunk) of every active sample, short chunks padded with silence, then
 * mixed = audioop.add(mixed, chunk, 2) in the order the samples were added).  An in-place source (out among srcs)
 * is not allowed. */
int sh_mix_chain_gather_i16(const sh_buf* const* srcs, const size_t* sample_offsets, const uint32_t* nsamples_each, uint32_t nsrc,
                            uint32_t nsamples, sh_buf* out, size_t out_sample_off);

/* The same two folds for every sample width audioop.add takes (1, 2, 3, 4 bytes; offsets, strides and counts in SAMPLES;
 * width 2 = the two functions above): upstream's mixer calls audioop.add(mixed, chunk, samplewidth) with the width of its
 * output format, which need not be 16 bits. */
int sh_mix_chain(const sh_buf* chunks, uint32_t nvoices, size_t stride, uint32_t nsamples, int width, sh_buf* out);
int sh_mix_chain_gather(const sh_buf* const* srcs, const size_t* sample_offsets, const uint32_t* nsamples_each, uint32_t nsrc,
                        uint32_t nsamples, int width, sh_buf* out, size_t out_sample_off);

/* ---- the real-time lane -------------------------------------------------------------------------------------------------------
 * Replaces: the thread upstream's playback.py runs its mixer on (the output thread pulls RealTimeMixer.chunks() while other threads
 * make sound).  Every entry point above holds the library's one lock and enqueues on its one stream pair: a mixer turn from another
 * thread queues behind whatever the others have enqueued, and its download holds the lock while the stream drains.  A LANE is a
 * stream (high priority), a lock, a source-table buffer and a chunk buffer of its own: sh_rt_mix_turn -- the ordered saturating fold
 * of sh_mix_chain_gather over the sources' current chunks (bit-identical: the same kernels) + the chunk copied to out_host --
 * takes the lane's lock only and waits for the lane's stream only; it returns when out_host holds the chunk.  Order against the
 * library's streams is established once per source: sh_rt_acquire(lane, src) makes the lane's next turn wait (on the device) for
 * everything the library has been given so far -- the upload / resample / render that made `src` -- and folds first what a render
 * still owes into it.  After that the caller must not write `src` while the lane can read it, and must not free it before the last
 * turn that names it has returned (turns are synchronous: once sh_rt_mix_turn is back nothing of the lane is in flight).
 * One lane per mixer; distinct lanes are independent; sh_rt_create / sh_rt_destroy / sh_rt_acquire take the library's lock. */
typedef struct sh_rt sh_rt;
int sh_rt_create(size_t max_chunk_bytes, uint32_t max_sources, sh_rt** out);
int sh_rt_destroy(sh_rt* lane);
int sh_rt_acquire(sh_rt* lane, const sh_buf* src);
int sh_rt_mix_turn(sh_rt* lane, const sh_buf* const* srcs, const size_t* sample_offsets, const uint32_t* nsamples_each, uint32_t nsrc,
                   uint32_t nsamples, int width, void* out_host);

/* ---- Sample.from_osc_block: int(scale*v), truncation toward zero; SH_ERR_OVERFLOW if out of range */
int sh_quantize_f32(const sh_buf* in_f32, size_t in_off, size_t n, double scale, int width,
                    sh_buf* out_pcm, size_t out_off);
/* same for float64 input (blocks produced by a float64 generator, e.g. upstream's own oscillators) */
int sh_quantize_f64(const sh_buf* in_f64, size_t in_off, size_t n, double scale, int width,
                    sh_buf* out_pcm, size_t out_off);
/* float stereo bus -> int16 with saturation instead of OverflowError (bus epilogue, [SPEC]) */
int sh_quantize_clip_f32(const sh_buf* in_f32, size_t n, double scale, sh_buf* out_i16);

/* ---- Sample.mix -> audioop.add(frag1, frag2, width): saturating pairwise add, width 1/2/4 */
int sh_pcm_add(const sh_buf* a, size_t a_off, const sh_buf* b, size_t b_off, size_t nbytes, int width,
               sh_buf* out, size_t out_off);
int sh_pcm_add_host(const void* a, const void* b, size_t nbytes, int width, void* out);

/* ---- the other elementwise Sample operations that delegate to audioop ----
 * Sample.amplify / invert -> audioop.mul: fbound(sample * factor) (clamp, then floor) */
int sh_pcm_mul(const sh_buf* in, size_t in_off, size_t nbytes, int width, double factor, sh_buf* out, size_t out_off);
/* Sample.fadeout (fadeout != 0): int(sample_i * (1 - i*slope/numsamples)); Sample.fadein: int(sample_i *
 * (i*slope/numsamples + offset)); i = sample index in the range, numsamples = nbytes/width, truncation */
int sh_pcm_fade(const sh_buf* in, size_t in_off, size_t nbytes, int width, int fadeout, double slope, double offset,
                sh_buf* out, size_t out_off);
/* Sample.modulate_amp: out[i] = int(in[i] * mod[i mod nmod]) (float64 product, truncation toward zero); mod_f64 is
 * a device buffer of nmod float64 factors, cycled.  A product outside the sample range is SH_ERR_OVERFLOW
 * (upstream: OverflowError from the array store). */
int sh_pcm_modulate(const sh_buf* in, size_t nbytes, int width, const sh_buf* mod_f64, size_t nmod, sh_buf* out);
/* samples as float64: out[i] = in[i] / divisor (Sample.get_frames_as_floats uses 2^(bits-1); modulate_amp with a
 * waveform modulator uses its largest absolute value) */
int sh_pcm_to_f64(const sh_buf* in, size_t nsamples, int width, double divisor, sh_buf* out_f64);
/* Sample.pan(lfo=...) (upstream synthplayer/sample.py, [RECALL]): out frame i = (int(l*(1-p)/2), int(r*(1+p)/2)),
 * p = pan_f64[i] (one float64 per frame, device resident: an oscillator rendered in place, or uploaded values);
 * nchannels 1 (l = r = the mono sample) or 2; out is stereo.  SH_ERR_OVERFLOW where Python would raise. */
int sh_pcm_pan_lfo(const sh_buf* in, size_t nframes, int width, int nchannels, const sh_buf* pan_f64, sh_buf* out);
/* Sample.bias -> audioop.bias: wrapping add */
int sh_pcm_bias(const sh_buf* in, size_t nbytes, int width, int bias, sh_buf* out);
/* Sample.reverse -> audioop.reverse: sample order reversed (out must not alias in) */
int sh_pcm_reverse(const sh_buf* in, size_t nbytes, int width, sh_buf* out);
/* Sample.mono / left / right -> audioop.tomono; Sample.stereo (mono source) -> audioop.tostereo */
int sh_pcm_tomono(const sh_buf* in, size_t nframes, int width, double lfactor, double rfactor, sh_buf* out);
int sh_pcm_tostereo(const sh_buf* in, size_t nframes, int width, double lfactor, double rfactor, sh_buf* out);
/* Sample.normalize / make_16bit / make_32bit -> audioop.lin2lin */
int sh_pcm_lin2lin(const sh_buf* in, size_t nsamples, int width, int new_width, sh_buf* out);
/* audioop.max (maximum absolute sample) and the sum of squares audioop.rms takes the root of
 * (exact for widths 1 and 2; width 4 is summed in float64 in a fixed tree order) */
int sh_pcm_stats(const sh_buf* in, size_t nbytes, int width, uint32_t* max_abs, double* sum_squares);
/* The same for interleaved stereo, per channel in one pass: [0] = left, [1] = right.  Replaces the
 * audioop.tomono(frames, w, 1, 0) / (.., 0, 1) + audioop.max / audioop.rms sequence of Sample.level_db_peak /
 * level_db_rms and LevelMeter.update (upstream synthplayer/sample.py, [RECALL], tree not mounted). */
int sh_pcm_stats_stereo(const sh_buf* in, size_t nframes, int width, uint32_t max_abs[2], double sum_squares[2]);

/* ---- Sample.resample -> audioop.ratecv(frames, width, nchannels, inrate, outrate, None) */
size_t sh_resample_out_frames(size_t in_frames, int inrate, int outrate);
/* width 1/2/4 integer PCM (bit-exact audioop arithmetic); is_float!=0: float32 PCM (width must be 4) */
int sh_resample(const sh_buf* in, size_t in_frames, int nchannels, int width, int is_float,
                int inrate, int outrate, sh_buf* out, size_t* out_frames);
int sh_resample_host(const void* in, size_t in_frames, int nchannels, int width, int is_float,
                     int inrate, int outrate, void* out, size_t* out_frames);
/* Sharding Sample.resample by output-frame range (no collective, a halo of one input frame).
 * sh_resample_span: the input frames [*in_first, *in_first + *in_count) that output frames [out_first, out_first +
 * out_n) of a stream of in_total_frames read (*in_first rounded down to a multiple of 16).  Host-only arithmetic.
 * sh_resample_range: resample that output range from a buffer holding input frames [in_first, in_first + in_held)
 * into out[0 .. out_n); out_first and in_first must be multiples of 16 frames.  Concatenating the ranges of a
 * partition gives sh_resample's result bit for bit. */
int sh_resample_span(size_t in_total_frames, int inrate, int outrate, size_t out_first, size_t out_n,
                     size_t* in_first, size_t* in_count);
int sh_resample_range(const sh_buf* in, size_t in_first, size_t in_held, int nchannels, int width, int is_float,
                      int inrate, int outrate, size_t out_first, size_t out_n, sh_buf* out);

/* ---- multi-GPU: voice-sharded banks, partial buses summed by RCCL over xGMI ------------ */
#define SH_DIST_ID_BYTES 128
int sh_dist_unique_id(void* id128);                       /* rank 0: ncclGetUniqueId */
int sh_dist_init(int rank, int world, const void* id128); /* ncclCommInitRank on the library stream */
int sh_dist_shutdown(void);
int sh_dist_rank(void);
int sh_dist_world(void);
/* What RCCL itself says about the communicator (not what the caller passed to sh_dist_init): out[0] = 1 when a communicator exists,
 * out[1] = ncclCommCount, out[2] = ncclCommUserRank (-1 without a communicator), out[3] = ncclGetVersion (0 while librccl.so has not
 * been loaded), out[4] = 1 when librccl.so is loaded.  Writes min(n, 5) words, returns 5; never loads the library by itself. */
int sh_dist_comm_info(int32_t* out, int n);
/* sum the ranks' float64 partial buses into root's buffer (ncclReduce, ncclDouble, in place) */
int sh_dist_reduce_bus(sh_buf* bus_f64, size_t nvalues, int root);
int sh_dist_allreduce_bus(sh_buf* bus_f64, size_t nvalues);
int sh_dist_barrier(void);
/* pipelined form: the collective (and, on root, the float64 -> float32 rounding into bus_f32) is enqueued on
 * the library's communication stream behind everything already enqueued on the main stream, so block s is
 * reduced while block s+1 renders.  `slot` (0 .. sh_dist_slots()-1) names the buffer pair; call
 * sh_dist_wait_slot(slot) before the main stream overwrites that slot's buffers again.  sh_sync() waits
 * for both streams. */
int sh_dist_slots(void);
int sh_dist_reduce_bus_async(sh_buf* bus_f64, size_t nvalues, int root, sh_buf* bus_f32, int slot);
int sh_dist_wait_slot(int slot);
/* The two calls above end the run of pipelined renders they are made in (like every entry point that can touch a bus buffer): a
 * pipeline drain per batch.  These three do not.  sh_dist_mark_slot(slot): call once the slot's last render lies at least two
 * launches back in its bank's run -- every partial bus of the slot has been folded by then -- it records where both render streams
 * stand.  sh_dist_reduce_bus_lagged: call a few launches after the mark; the collective waits for the two marks only, and the run
 * goes on (without a mark, or if a fold into the buffers is still owed, it behaves like sh_dist_reduce_bus_async).
 * sh_dist_wait_slot_keep makes both render streams wait for the slot's collective. */
int sh_dist_mark_slot(int slot);
int sh_dist_reduce_bus_lagged(sh_buf* bus_f64, size_t nvalues, int root, sh_buf* bus_f32, int slot);
int sh_dist_wait_slot_keep(int slot);
/* float64 bus -> float32 bus after the reduce */
int sh_bus_finalize(const sh_buf* bus_f64, size_t nvalues, sh_buf* bus_f32);

#ifdef __cplusplus
}
#endif
#endif /* SYNTHHIP_H */
