// osc_bank.hip -- the voice table resident in HBM (sh_bank) and the launch records of a block.
//
// Kernels
//   k_prepare_chunks  one wavefront per 64 voices: resolve everything that depends on `start` (phase-table piece,
//                     envelope lines) into a 384-byte launch record per voice, and classify the voices for the render
//                     kernel (lean FastRec list / general index list per chunk).  In a stream the same code runs
//                     inside the previous block's render kernel, not as a kernel of its own
//   k_prepare         the same records for a single voice (sh_osc_render)
//   k_prepare_segments(_var)  a record set per segment of a long row / of a transition launch, all in one launch
#include "osc_host.hpp"
#include <new>
#include <stdlib.h>
#include <math.h>

namespace {

__global__ void k_prepare(BankPtrs B, uint32_t first, uint32_t nvoices, uint64_t start, uint32_t nframes,
                          VoiceLaunch* __restrict__ out, VoiceFM* __restrict__ out_fm) {
    uint32_t vi = blockIdx.x * blockDim.x + threadIdx.x;
    PrepInfo info;
    if (vi < nvoices) prepare_voice(B, first, vi, start, nframes, out, out_fm, info);
}

__global__ __launch_bounds__(64) void k_prepare_chunks(BankPtrs B, LaunchSet S, uint32_t nvoices, uint64_t start, uint32_t nframes) {
    prepare_chunk(B, S, blockIdx.x, nvoices, start, nframes);
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

// The tile set of a launch that was not predicted (the first launches of a run, a jump): a kernel in front of the render; in a stream
// of blocks the same wavefronts run inside the render kernel of the launch before last (prepare_tiles_wave, osc_device.hpp).
__global__ __launch_bounds__(256) void k_prepare_tiles(BankPtrs B, TileSet T, uint32_t nvoices, uint64_t start, uint32_t nframes, uint32_t ntiles,
                                                       LaunchSet recs, uint32_t with_recs) {
    const uint32_t c = T.k0 * T.groups + blockIdx.x;             // (the chunks of the set's range)
    if (c < (nvoices + 63) / 64)
        prepare_tiles_wave(B, T, nvoices, start, nframes, ntiles, c, blockIdx.y * 4 + (threadIdx.x >> 6), with_recs ? &recs : nullptr);
}

}  // namespace

namespace shosc {

int launch_prepare_tiles(hipStream_t st, const BankPtrs& P, const TileSet& T, uint32_t nvoices, uint64_t start, uint32_t nframes, const LaunchSet* recs) {
    const uint32_t ntiles = sh::div_up(nframes, TILE_FRAMES);
    if (T.k1 == T.k0) return SH_OK;                            // no chunk sounds in the block
    hipLaunchKernelGGL(k_prepare_tiles, dim3((T.k1 - T.k0) * T.groups, sh::div_up(ntiles, 4 * TILES_PER_WAVE)), dim3(256), 0, st, P, T, nvoices, start, nframes, ntiles,
                       recs ? *recs : LaunchSet(), recs ? 1u : 0u);
    SH_CHECK_LAUNCH("k_prepare_tiles");
    return SH_OK;
}

// The tile set of a launch of `ntiles` tiles and `groups` voice groups carved out of one pool-backed block (records, then the two
// mask arrays; the masks of chunks that do not exist -- the padding of a group's row to a multiple of eight -- are zeroed here).
int grow_tile_set(sh::Pooled& block, TileSet& T, uint32_t& carved_tiles, uint32_t ntiles, uint32_t nvoices, uint32_t groups, uint32_t range_chunks, hipStream_t st) {
    const uint32_t mask_k = tile_mask_k(nvoices, groups);
    if (block.ptr && T.recs == (TileRec*)block.ptr && T.groups == groups && T.mask_k == mask_k && carved_tiles == ntiles && T.rec_chunks >= range_chunks) return SH_OK;
    // (a quarter of headroom: the range of a stream of blocks breathes by a chunk or two)
    uint32_t rec_chunks = ((range_chunks + range_chunks / 4 + 8) + 7) & ~7u;
    const uint32_t all_chunks = (nvoices + 63) / 64 + groups;           // (a range is whole mask slots: up to groups - 1 chunks past the table)
    if (rec_chunks > all_chunks) rec_chunks = all_chunks;
    if (rec_chunks < range_chunks) rec_chunks = range_chunks;
    const size_t slots = (size_t)rec_chunks * 64;
    const size_t b_recs = sizeof(TileRec) * ntiles * slots, b_mask = ((sizeof(uint64_t) * ntiles * groups * mask_k) + 255) & ~(size_t)255;
    int rc = sh::grow_pooled(block, b_recs + 2 * b_mask);
    if (rc) return rc;
    carved_tiles = ntiles;
    char* p = (char*)block.ptr;
    T.recs = (TileRec*)p; p += b_recs;
    T.lean = (uint64_t*)p; p += b_mask;
    T.gen = (uint64_t*)p;
    T.groups = groups;
    T.mask_k = mask_k;
    T.k0 = 0;
    T.k1 = 0;
    T.rec_chunks = rec_chunks;
    if (((nvoices + 63) / 64) != groups * mask_k) SH_HIP(hipMemsetAsync(T.lean, 0, 2 * b_mask, st));     // (some slots have no chunk: they stay zero)
    return SH_OK;
}

LaunchSet launch_set(const sh_bank* b, int k) {
    LaunchSet s;
    s.launch = b->d_launch_buf[k];
    s.fm = b->d_launch_fm_buf[k];
    s.fast = b->d_fast_buf[k];
    s.gen_idx = b->d_gen_idx_buf[k];
    s.counts = b->d_counts_buf[k];
    return s;
}

BankPtrs ptrs(const sh_bank* b) {
    BankPtrs p;
    p.voices = b->d_voices;
    p.segs = b->d_segs;
    p.coefs = b->d_coefs;
    p.partials = b->d_partials;
    p.hint = b->d_hint;
    p.lfo_hint = b->d_hint + b->nvoices;
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
    p.tiles = TileSet{nullptr, nullptr, nullptr, 0, 0, 0};
    p.next_tiles = p.tiles;
    p.next_tile_wgs = 0;
    p.next_ntiles = 0;
    p.polys = b->d_polys;
    p.chunk_span = b->d_chunk_span;
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

std::vector<sh_bank*>& live_banks() {
    static std::vector<sh_bank*> v;
    return v;
}

const shm::sc_pair* trig_table() { return (const shm::sc_pair*)sh::state().trig; }

int prepare_single(sh_bank* b, uint32_t first, uint32_t count, uint64_t start, uint32_t nframes) {
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
int prepare_chunks_now(sh_bank* b, uint64_t start, uint32_t nframes, hipStream_t st) {
    hipLaunchKernelGGL(k_prepare_chunks, sh::grid1d(b->nvoices, 64), dim3(64), 0, st, ptrs(b), launch_set(b, b->cur), b->nvoices, start, nframes);
    SH_CHECK_LAUNCH("k_prepare_chunks");
    return SH_OK;
}

int acquire_records(sh_bank* b, uint64_t start, uint32_t nframes, hipStream_t launch_stream, bool in_run, bool accept_sparse, bool* deferred) {
    sh::State& S = sh::state();
    for (int k = 0; k < sh_bank::NSETS; ++k) {
        if (b->spec[k].valid && b->spec[k].start == start && b->spec[k].nframes == nframes && (accept_sparse || !b->spec[k].sparse)) {
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
        if (in_run && S.aux_busy) {
            SH_HIP(hipEventRecord(S.ev_aux, S.stream2));
            SH_HIP(hipStreamWaitEvent(S.stream, S.ev_aux, 0));
        }
        for (int c = 0; c < sh_bank::NSETS && k < 0; ++c)
            if (!(in_run && (c == b->cur || c == b->last_target))) k = c;
    }
    b->spec[k].valid = false;
    if (deferred) {
        *deferred = true;
        b->cur = k;
        b->d_launch = b->d_launch_buf[k];
        b->d_launch_fm = b->d_launch_fm_buf[k];
        return SH_OK;
    }
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
uint32_t plan_segments(const sh_bank* b, uint64_t start, uint32_t nframes, uint64_t T, uint64_t max_len, bool corners,
                              uint32_t* seg_first) {
    const uint64_t end = start + nframes;
    uint64_t cuts[SEG_MAX + 2];
    uint32_t nc = 0;
    uint64_t pos = start;
    cuts[nc++] = pos;
    const uint64_t flat = b->env_flat_from, rel = b->env_flat_until;       // last decay end, first sustain end
    const bool shared = corners && !b->env_corners.empty();
    if (!shared && pos < flat && flat < end && flat - pos <= max_len) { pos = flat; cuts[nc++] = pos; }
    while (pos < end && nc <= SEG_MAX) {
        uint64_t next = pos < T ? T : 2 * pos;
        if (shared) {
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

int grow_segment_sets(sh::Pooled& block, LaunchSet& g, uint32_t& cap, uint32_t nseg, uint32_t nvoices) {
    if (cap >= nseg && block.ptr) return SH_OK;
#ifdef SH_GUARD_SETS
    auto up = [](size_t x) { return ((x + 255) & ~(size_t)255) + 65536; };
#else
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
#endif
    const size_t nchunks = (nvoices + 63) / 64, n = (size_t)nseg * set_slots(nvoices);
    const size_t b_launch = up(sizeof(VoiceLaunch) * n), b_fm = up(sizeof(VoiceFM) * n), b_fast = up(sizeof(FastRec) * n),
                 b_idx = up(sizeof(uint32_t) * n), b_cnt = up(sizeof(uint32_t) * 4 * nseg * nchunks);
    cap = 0;
    int rc = sh::grow_pooled(block, b_launch + b_fm + b_fast + b_idx + b_cnt);     // (724 bytes per voice and segment)
    if (rc) return rc;
    char* p = (char*)block.ptr;
#ifdef SH_GUARD_SETS
    (void)hipMemsetAsync(block.ptr, 0, block.cap, sh::state().stream);
#endif
    g.launch = (VoiceLaunch*)p; p += b_launch;
    g.fm = (VoiceFM*)p; p += b_fm;
    g.fast = (FastRec*)p; p += b_fast;
    g.gen_idx = (uint32_t*)p; p += b_idx;
    g.counts = (uint32_t*)p;
    cap = nseg;
    return SH_OK;
}

int launch_prepare_segments(hipStream_t st, const BankPtrs& P, const LaunchSet& base, uint32_t nvoices, uint32_t nseg, uint64_t start,
                            uint32_t nframes, uint32_t seg_frames) {
    hipLaunchKernelGGL(k_prepare_segments, dim3(sh::div_up(nvoices, 64), nseg), dim3(64), 0, st, P, base, nvoices, start, nframes, seg_frames);
    SH_CHECK_LAUNCH("k_prepare_segments");
    return SH_OK;
}

int launch_prepare_segments_var(hipStream_t st, bool sloped, const BankPtrs& P, const LaunchSet& base, uint32_t nvoices, uint32_t nseg, uint64_t start) {
    if (sloped) hipLaunchKernelGGL(k_prepare_segments_var<true>, dim3(sh::div_up(nvoices, 64), nseg), dim3(64), 0, st, P, base, nvoices, start);
    else hipLaunchKernelGGL(k_prepare_segments_var<false>, dim3(sh::div_up(nvoices, 64), nseg), dim3(64), 0, st, P, base, nvoices, start);
    SH_CHECK_LAUNCH("k_prepare_segments_var");
    return SH_OK;
}

int bank_check_plain(const sh_bank* b, const char* who) {
    if (b->launch_rows) return SH_OK;                      // sh_bank_render_rows supplies them
    // (found once, at sh_bank_create: walking 22 528 voice records here cost every render call of a table of notes 15 us)
    if (b->first_row_voice >= 0)
        return sh::set_error(SH_ERR_INVALID, "%s: voice %u reads a modulation / sample row; render the bank with sh_bank_render_rows", who, (unsigned)b->first_row_voice);
    if (b->needs_rows) return sh::set_error(SH_ERR_INVALID, "%s: the bank has modulation rows (sh_bank_set_rows); render it with sh_bank_render_rows", who);
    return SH_OK;
}

}  // namespace shosc

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
        if (v.start_frame && (v.kind == SH_BUFFER || v.fm_mode == SH_FM_BUFFER))
            return sh::set_error(SH_ERR_INVALID, "voice %u: a voice that reads a row of the launch's matrix cannot have an onset (delay its sources)", i);
        if (v.start_frame > (1ull << 62)) return sh::set_error(SH_ERR_INVALID, "voice %u: start_frame beyond 2^62", i);
        if (v.fm_mode < SH_FM_NONE || v.fm_mode > SH_FM_BUFFER)
            return sh::set_error(SH_ERR_INVALID, "voice %u: unknown fm_mode %d", i, v.fm_mode);
        uint32_t off = v.fm_mode ? v.time_seg_offset : v.seg_offset;
        uint32_t cnt = v.fm_mode ? v.time_seg_count : v.seg_count;
        if (cnt == 0 || off > nsegs || cnt > nsegs - off)
            return sh::set_error(SH_ERR_INVALID, "voice %u: phase table [%u,+%u) outside %u pieces", i, off, cnt, nsegs);
        if (segs[off].n0 != 0) return sh::set_error(SH_ERR_INVALID, "voice %u: phase table does not start at sample 0", i);
        if (v.fm_mode == SH_FM_SINE) {
            // ABI 5: an LFO's bias rides in `frequency` (f (1 + bias): the angle's sum runs over the accumulated time), and an LFO that
            // moves brings its table (two records per piece at seg_offset); the lean loops are instantiated for lfo_bias = 0 alone
            if (v.lfo_bias != 0.0) return sh::set_error(SH_ERR_INVALID, "voice %u: lfo_bias must be 0 (fold it into frequency: f * (1 + bias))", i);
            if (v.seg_count & 1u) return sh::set_error(SH_ERR_INVALID, "voice %u: an LFO table holds two records per piece (seg_count %u)", i, v.seg_count);
            if (v.seg_count && (v.seg_offset > nsegs || v.seg_count > nsegs - v.seg_offset || segs[v.seg_offset].n0 != 0))
                return sh::set_error(SH_ERR_INVALID, "voice %u: LFO table [%u,+%u) outside %u records or not starting at sample 0", i, v.seg_offset, v.seg_count, nsegs);
            // (a tile record addresses the LFO's piece by its index in `segs` packed into 24 bits, 0xFFFFFF = a constant LFO:
            //  prepare_tiles in osc_device.hpp -- a table beyond that would be read at a truncated index, silently)
            if (v.seg_count && (uint64_t)v.seg_offset + v.seg_count >= 0xFFFFFFull)
                return sh::set_error(SH_ERR_INVALID, "voice %u: LFO table [%u,+%u) beyond the 2^24 - 1 records an FM voice's table may lie within (share LfoTables between voices)", i, v.seg_offset, v.seg_count);
            if (v.lfo_K != 0.0 && v.seg_count == 0) return sh::set_error(SH_ERR_INVALID, "voice %u: an LFO that moves (lfo_K != 0) needs its table", i);
        }
        if (v.kind == SH_HARMONICS) {
            if (v.harm_dense < 0 || v.harm_dense > 2) return sh::set_error(SH_ERR_INVALID, "voice %u: harm_dense %d not in {0,1,2}", i, v.harm_dense);
            if (v.harm_dense == 2 && v.harm_count != 16) return sh::set_error(SH_ERR_INVALID, "voice %u: polynomial form needs 16 coefficients", i);
            uint32_t lim = v.harm_dense ? ncoefs : npartials;
            if (v.harm_offset > lim || v.harm_count > lim - v.harm_offset)
                return sh::set_error(SH_ERR_INVALID, "voice %u: harmonics [%u,+%u) outside table of %u", i, v.harm_offset, v.harm_count, lim);
            if (v.harm_dense == 1 && (v.harm_count & 7))
                return sh::set_error(SH_ERR_INVALID, "voice %u: dense harmonic count %u is not a multiple of 8", i, v.harm_count);
        }
        if (v.guard_count) {           // ABI 6: the term-by-term list behind a polynomial / Clenshaw form (the int16 boundary guard)
            if (v.kind != SH_HARMONICS || v.harm_dense == 0 || v.fm_mode != SH_FM_NONE)
                return sh::set_error(SH_ERR_INVALID, "voice %u: a guard list belongs to a Harmonics voice in the polynomial or Clenshaw form without FM", i);
            if (v.guard_offset > npartials || v.guard_count > npartials - v.guard_offset)
                return sh::set_error(SH_ERR_INVALID, "voice %u: guard list [%u,+%u) outside table of %u", i, v.guard_offset, v.guard_count, npartials);
            if (!(v.guard_t >= 0.0 && v.guard_c >= 0.0 && v.guard_t < 1.0 && v.guard_c < 1.0))
                return sh::set_error(SH_ERR_INVALID, "voice %u: guard bounds must be finite, >= 0 and < 1", i);
        }
    }
    sh_bank* b = new (std::nothrow) sh_bank;
    if (!b) return sh::set_error(SH_ERR_NOMEM, "host allocation failed");
    hipStream_t st = sh::state().stream;
    b->nvoices = nvoices;
    for (uint32_t i = 0; i < nvoices; ++i) b->has_guard = b->has_guard || voices[i].guard_count != 0;
    b->nsegs = nsegs;
    b->ncoefs = ncoefs;
    b->npartials = npartials;
    b->h_voices.assign(voices, voices + nvoices);
    for (uint32_t i = 0; i < nvoices && b->first_row_voice < 0; ++i)
        if (voices[i].fm_mode == SH_FM_BUFFER || voices[i].kind == SH_BUFFER) b->first_row_voice = (long long)i;
    for (uint32_t i = 0; i < nvoices; ++i)
        if (voices[i].bias == 0.0 && !voices[i].flip &&
            ((voices[i].kind == SH_HARMONICS && voices[i].harm_dense == 2 && voices[i].fm_mode == SH_FM_NONE) ||
             (voices[i].kind == SH_SINE && voices[i].fm_mode == SH_FM_SINE) ||
             (voices[i].fm_mode == SH_FM_NONE && (voices[i].kind == SH_SINE || voices[i].kind == SH_SAWTOOTH || voices[i].kind == SH_SQUARE ||
                                                  voices[i].kind == SH_TRIANGLE || voices[i].kind == SH_PULSE)))) {
            b->lean_candidates += 1;
            if (voices[i].kind != SH_HARMONICS) b->lean_fm_candidates += 1;       // needs a kernel with other record kinds
            if (voices[i].kind == SH_SINE && voices[i].fm_mode == SH_FM_SINE) b->lean_fmsine_candidates += 1;
        }
    for (uint32_t i = 0; i < nvoices; ++i)
        if (voices[i].start_frame) b->has_onsets = true;
    uint32_t tile_capable = 0;
    for (uint32_t i = 0; i < nvoices; ++i) {
        const sh_voice& v = voices[i];
        const bool wave = v.kind == SH_SAWTOOTH || v.kind == SH_SQUARE || v.kind == SH_TRIANGLE || v.kind == SH_PULSE;
        const bool fm_sine = v.kind == SH_SINE && v.fm_mode == SH_FM_SINE;            // a Sine carrier with a closed-form Sine LFO
        const bool ok = v.bias == 0.0 && !v.flip && (fm_sine || (v.fm_mode == SH_FM_NONE && ((v.kind == SH_HARMONICS && v.harm_dense == 2) || v.kind == SH_SINE || wave)));
        if (ok) tile_capable += 1;
        if (wave || fm_sine) b->tile_waveforms = true;
    }
    // (a table of notes with the odd voice that only the general code can do -- noise, a bias, a sparse series -- is still a table of
    // notes: those voices are general pairs of every tile they sound in)
    b->tile_all = (uint64_t)tile_capable * 10 >= (uint64_t)nvoices * 9;
    {
        b->all_lean = b->lean_candidates == nvoices;
        for (uint32_t i = 0; i < nvoices && b->all_lean; ++i) {
            const sh_voice& v = voices[i];
            const uint64_t on = v.start_frame;               // (absolute frames: the voice's own boundaries + its onset)
            if (on > b->env_flat_from) b->env_flat_from = on;     // a launch in front of or across an onset is not "all voices on their sustain"
            if (v.env.enabled) {
                if (v.env.n_decay_end + on > b->env_flat_from) b->env_flat_from = v.env.n_decay_end + on;
                if (v.env.n_attack_end + on > b->env_flat_from) b->env_flat_from = v.env.n_attack_end + on;
                if (v.env.n_sustain_end + on < b->env_flat_until) b->env_flat_until = v.env.n_sustain_end + on;
                if (b->env_corners.size() <= 16) {
                    for (uint64_t c : {on, v.env.n_attack_end + on, v.env.n_decay_end + on, v.env.n_sustain_end + on, v.env.n_release_end + on, v.env.n_release_end + on + 1})
                        if (c && std::find(b->env_corners.begin(), b->env_corners.end(), c) == b->env_corners.end()) b->env_corners.push_back(c);
                }
            }
            const uint32_t toff = v.fm_mode ? v.time_seg_offset : v.seg_offset, tcnt = v.fm_mode ? v.time_seg_count : v.seg_count;
            for (uint32_t k = 0; k + 1 < tcnt; ++k) {
                const uint64_t a = segs[toff + k].n0, e = segs[toff + k + 1].n0;
                const uint64_t len = e - a;
                for (int q = 0; q < 34; ++q)
                    if (len < (1ull << q) && e + on > b->short_piece_end[q]) b->short_piece_end[q] = e + on;
            }
            // (the pieces of an FM Sine voice's LFO table -- oscillators.LfoTable -- do not count: the lean loop follows the first end
            // of them inside a launch at a tile boundary and a second one, rare and late, along the line of the piece in front of it:
            // never the general code)
        }
    }
    if (b->env_corners.size() > 16) {                              // voices with envelopes of their own: cut where they are all flat only
        b->env_corners.clear();
        b->own_envelopes = true;
    }
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
    {   // the polynomial of every polynomial-Harmonics voice by VOICE (tile-classified launches address it without a pointer chase)
        std::vector<double> polys(set_slots(nvoices) * 16, 0.0);
        // (a plain Sine is the series with one partial: sin(t) P(cos t), P = 1; a Pulse keeps its width in the first slot)
        for (uint32_t i = 0; i < nvoices; ++i) {
            if (voices[i].kind == SH_HARMONICS && voices[i].harm_dense == 2)
                for (int u = 0; u < 16; ++u) polys[(size_t)i * 16 + u] = coefs[voices[i].harm_offset + u];
            else if (voices[i].kind == SH_SINE && voices[i].fm_mode == SH_FM_SINE) {       // (the closed-form FM's constants by voice)
                const sh_voice& v = voices[i];
                const double row[8] = {v.frequency, v.fm_phase0, v.frequency * v.fm_inc, v.lfo_a, v.lfo_d, v.lfo_K, v.lfo_C0, v.lfo_bias};
                for (int u = 0; u < 8; ++u) polys[(size_t)i * 16 + u] = row[u];
                polys[(size_t)i * 16 + 8] = lfo_rot[i].x;                                  // (the LFO's rotation by 64 samples)
                polys[(size_t)i * 16 + 9] = lfo_rot[i].y;
            }
            else if (voices[i].kind == SH_SINE) polys[(size_t)i * 16 + 15] = 1.0;
            else if (voices[i].kind == SH_PULSE) polys[(size_t)i * 16] = voices[i].pulsewidth;
        }
        if (!rc) rc = upload_array(&b->d_polys, polys.data(), polys.size(), st);
        // ... and per chunk of 64 voices the frames in which any of them sounds (the extra sample sits AT the release's end)
        std::vector<uint64_t> span(2 * (size_t)sh::div_up(nvoices, 64));
        for (uint32_t c = 0; c < sh::div_up(nvoices, 64); ++c) {
            uint64_t lo = ~0ull, hi = 0;
            for (uint32_t i = c * 64; i < nvoices && i < c * 64 + 64; ++i) {
                const uint64_t on = voices[i].start_frame;
                const uint64_t end = voices[i].env.enabled ? on + voices[i].env.n_release_end + 1 : ~0ull;
                lo = on < lo ? on : lo;
                hi = (end < on || end > hi) ? (end < on ? ~0ull : end) : hi;
            }
            span[2 * c] = lo;
            span[2 * c + 1] = hi;
        }
        if (!rc) rc = upload_array(&b->d_chunk_span, span.data(), span.size(), st);
        b->chunk_span = span;
    }
    if (!rc) rc = upload_array(&b->d_seg_rot, seg_rot.data(), nsegs, st);
    if (!rc) rc = upload_array(&b->d_lfo_rot, lfo_rot.data(), nvoices, st);
    if (!rc) {
        hipError_t e = hipSuccess;
        for (int k = 0; k < sh_bank::NSETS && e == hipSuccess; ++k) {
            const size_t slots = set_slots(nvoices);           // whole chunks: see segment_set
            e = hipMalloc((void**)&b->d_launch_buf[k], sizeof(VoiceLaunch) * slots);
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_launch_fm_buf[k], sizeof(VoiceFM) * slots);
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_fast_buf[k], sizeof(FastRec) * slots);
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_gen_idx_buf[k], sizeof(uint32_t) * slots);
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_counts_buf[k], sizeof(uint32_t) * 4 * ((nvoices + 63) / 64));
        }
        // (carrier / time table pieces, then LFO table pieces, then the arrival counters of self-folding launches)
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_hint, sizeof(uint32_t) * (2 * (size_t)nvoices + sh_bank::SELF_TILES));
        if (e == hipSuccess) e = hipMemsetAsync(b->d_hint, 0, sizeof(uint32_t) * (2 * (size_t)nvoices + sh_bank::SELF_TILES), st);
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
        for (int k = 0; k < 4; ++k) sh::release_pooled(b->parts[k]);            // (back to the pool: stream-ordered reuse)
        for (int k = 0; k < 2; ++k) {
            sh::release_pooled(b->seg_block[k]);
            sh::release_pooled(b->seg_scratch[k]);
        }
        sh::release_pooled(b->gen_block);
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
        if (b->d_seg_rot) (void)hipFree(b->d_seg_rot);
        if (b->d_polys) (void)hipFree(b->d_polys);
        if (b->d_chunk_span) (void)hipFree(b->d_chunk_span);
        for (int k = 0; k < sh_bank::NTILESETS; ++k) sh::release_pooled(b->tile_block[k]);
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
    if (b->last_tiled) {                                     // classified per (voice, tile), not per voice: see sh_debug_counters
        if (nfast) *nfast = 0;
        if (ngeneral) *ngeneral = 0;
        return SH_OK;
    }
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

int sh_bank_set_rows(sh_bank* b, const int32_t* fm_row, const int32_t* pwm_row) {
    SH_REQUIRE_INIT();
    if (!b || !fm_row || !pwm_row) return sh::set_error(SH_ERR_INVALID, "sh_bank_set_rows: NULL argument");
    for (uint32_t i = 0; i < b->nvoices; ++i) {
        const sh_voice& v = b->h_voices[i];
        if ((v.fm_mode == SH_FM_BUFFER || v.kind == SH_BUFFER) && fm_row[i] < 0)
            return sh::set_error(SH_ERR_INVALID, "sh_bank_set_rows: voice %u (SH_FM_BUFFER / SH_BUFFER) needs a row", i);
        if (pwm_row[i] >= 0 && v.kind != SH_PULSE) return sh::set_error(SH_ERR_INVALID, "sh_bank_set_rows: voice %u has a pwm row but is no Pulse", i);
        if (pwm_row[i] >= 0 && v.start_frame) return sh::set_error(SH_ERR_INVALID, "sh_bank_set_rows: voice %u has an onset and cannot read a row", i);
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

}  // extern "C"
