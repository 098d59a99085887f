// osc_render.hip -- fused generate-and-mix of a voice bank into the stereo bus: the host side of a render call (which shape a launch
// takes, the two-stream pipeline, who folds whose partial buses) and the launches.  The kernels: osc_render_kernels.hpp.
#include "osc_host.hpp"
#include "osc_render_kernels.hpp"
#include <stdlib.h>


namespace {

// Memory that render launches write (directly, or through a fold they owe / carry out) while some bank's run is open.  The two
// render streams are ordered against each other only where a run starts (ev_join) and where one ends (ev_aux): a launch on
// stream2 is NOT behind what the main stream was given after its run began.  So every launch checks what it is about to
// write against this list -- another bank's entry: every run ends first (sh::flush_pending: the order of the two writes must
// be the callers'); an entry of its own bank written on the OTHER stream (single-group launches write the caller's buffers
// themselves): its own run ends first.  The list is emptied whenever the streams are fully joined.
struct WriteRec { const char* lo; const char* hi; const sh_bank* owner; uint8_t streams; };     // streams: bit 0 `stream`, bit 1 stream2
std::vector<WriteRec>& bank_writes() {
    static std::vector<WriteRec> v;
    return v;
}
bool overlaps_writes(const sh_bank* b, bool own, uint8_t streams, const void* p, size_t bytes) {
    if (!p) return false;
    const char* lo = (const char*)p;
    for (const WriteRec& r : bank_writes())
        if ((r.owner == b) == own && (!own || (r.streams & streams)) && lo < r.hi && r.lo < lo + bytes) return true;
    return false;
}
void note_write(const sh_bank* b, const void* p, size_t bytes, uint8_t stream_bit) {
    if (!p) return;
    const char* lo = (const char*)p;
    for (WriteRec& r : bank_writes())
        if (r.owner == b && r.lo == lo) { if (lo + bytes > r.hi) r.hi = lo + bytes; r.streams |= stream_bit; return; }
    bank_writes().push_back(WriteRec{lo, lo + bytes, b, stream_bit});
}

}  // namespace

namespace shosc {

// Fold what one bank still owes (oldest first: a bus used for two blocks ends up holding the later one) on the main stream and
// end its run.  The caller has joined stream2 into the main stream.
int fold_bank(sh_bank* b) {
    sh::State& S = sh::state();
    if (b->run_active) S.open_runs -= 1;
    b->run_active = false;
    b->run_count = 0;
    // (the caller has made the main stream wait for stream2, and this bank's next stream2 launch will wait for the main stream:
    // its own earlier writes are ordered from here on; with no run open anywhere, everybody's are)
    for (WriteRec& r : bank_writes()) if (r.owner == b) r.streams = 0;
    if (S.open_runs == 0) bank_writes().clear();
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

int join_aux() {
    sh::State& S = sh::state();
    if (S.aux_busy) {
        // recorded here, not behind every launch on stream2: an event recorded now stands for everything stream2 has been given
        // so far (0.7 us of host time per block saved for a small bank, whose blocks are host-bound)
        S.aux_busy = false;
        SH_HIP(hipEventRecord(S.ev_aux, S.stream2));
        SH_HIP(hipStreamWaitEvent(S.stream, S.ev_aux, 0));
    }
    return SH_OK;
}

}  // namespace shosc

namespace sh {
int flush_pending() {
    int rc = join_aux();
    if (rc) return rc;
    for (sh_bank* b : live_banks()) {                     // every bank's run ends here: the next one starts on `stream`
        rc = fold_bank(b);
        if (rc) return rc;
    }
    bank_writes().clear();
    return SH_OK;
}

bool fold_owed_into(const void* p, size_t bytes) {
    const char* lo = (const char*)p;
    for (sh_bank* b : live_banks())
        for (int k = 0; k < b->npending; ++k) {
            const PendingCombine& pc = b->pending[k];
            const struct { const void* q; size_t n; } outs[3] = {{pc.o32, (size_t)pc.nframes * 8}, {pc.o64, (size_t)pc.nframes * 16}, {pc.o16, (size_t)pc.nframes * 4}};
            for (const auto& o : outs)
                if (o.q && lo < (const char*)o.q + o.n && (const char*)o.q < lo + bytes) return true;
        }
    return false;
}

// Both render streams wait for each other (events only): whatever either has been given so far precedes whatever either is
// given from here on.  Used when a device block the launches of both streams use changes hands (grow_pooled).
int join_streams() {
    State& S = state();
    int rc = join_aux();
    if (rc) return rc;
    SH_HIP(hipEventRecord(S.ev_sync, S.stream));
    SH_HIP(hipStreamWaitEvent(S.stream2, S.ev_sync, 0));
    return SH_OK;
}

void free_render_buffers() {
    for (sh_bank* b : live_banks()) {
        for (int k = 0; k < 4; ++k) release_pooled(b->parts[k]);
        for (int k = 0; k < 2; ++k) {
            release_pooled(b->seg_block[k]);
            release_pooled(b->seg_scratch[k]);
            b->seg_cap[k] = 0;
        }
        release_pooled(b->gen_block);
        b->gen_segs = 0;
        for (int k = 0; k < sh_bank::NTILESETS; ++k) {
            release_pooled(b->tile_block[k]);
            b->tile_carved[k] = 0;
            b->tile_spec[k].valid = false;
        }
    }
}
}  // namespace sh

// What the launches of one bank render have in common (bank_render works it out; the three shapes of a launch -- tile-classified,
// segmented, plain -- enqueue their kernels from it).
struct RenderLaunch {
    sh_bank* b;
    hipStream_t st;
    uint64_t start, next_start;
    uint32_t nframes, tiles, groups, vpg, nchunks, prep_wgs;
    int mode, var;                    // mode: COMBINED_* (which lean kinds the bank can hold); var: waves * 100 + frames per lane * 10 + min waves per SIMD
    bool split, with_general, use_aux;
    bool self_prepare;                // nothing resolved the launch's records: the lean kernel's workgroups do (LaunchArgs::self_prepare)
    bool alone;                       // the launch does not continue a run: nothing renders beside it (LaunchArgs::alone)
    LaunchSet cur, next;
    BusOut out;                       // the caller's buses
    double2* parts;                   // this launch's partial buses (NULL: one voice group, the kernel writes `out` itself)
    FoldIn fold;                      // the fold this launch takes over
    uint32_t* gen_valid;
    uint32_t c_lo, c_hi;              // (tile-classified launches) the chunks that can sound in this block: [c_lo, c_hi)
    LaunchArgs args(const BankPtrs& P, const LaunchSet& set) const { return LaunchArgs{P, trig_table(), b->nvoices, vpg, set, start, nframes, self_prepare ? 1u : 0u, alone ? 1u : 0u}; }
    NextArgs next_args(const LaunchSet& nx, uint32_t wgs) const { return NextArgs{nx, next_start, wgs}; }
};

// A TILE-CLASSIFIED launch: the lean tiles kernel (which also resolves, in rows behind its voice groups, the tile set -- and the launch
// records that will be needed -- of the block two launches on) and the general pairs' kernel behind it, or the merged kernel for a
// short launch.
static int launch_tiled(const RenderLaunch& L, bool records_deferred) {
    sh_bank* b = L.b;
    hipStream_t st = L.st;
    const uint32_t nframes = L.nframes, tiles = L.tiles, groups = L.groups, nchunks = L.nchunks;
    const uint64_t start = L.start, next_start = L.next_start;
    int rc = SH_OK;
    const uint32_t ntiles = sh::div_up(nframes, TILE_FRAMES);
    // the chunks that can sound in a block, as a range of mask slots k (chunk c = group + k groups): [k0, k1)
    auto k_range = [&](uint64_t s0, uint32_t& k0, uint32_t& k1) {
        uint32_t c_lo = nchunks, c_hi = 0;
        const uint64_t s1 = s0 + (uint64_t)nframes;
        for (uint32_t c = 0; c < nchunks; ++c)
            if (!(s1 <= b->chunk_span[2 * c] || s0 >= b->chunk_span[2 * c + 1])) { c_lo = c < c_lo ? c : c_lo; c_hi = c + 1; }
        k0 = c_hi ? c_lo / groups : 0u;
        k1 = c_hi ? sh::div_up(c_hi, groups) : 0u;
    };
    const int ks = (int)(b->tile_count % sh_bank::NTILESETS);
    TileSet& T = b->tile_set[ks];
    sh_bank::TileSpec& sp = b->tile_spec[ks];
    BankPtrs P = ptrs(b);
    sh::counters().tiled_launches += 1;
    if (sp.valid && sp.start == start && sp.nframes == nframes && sp.groups == groups) sh::counters().tiled_predicted += 1;
    if (!(sp.valid && sp.start == start && sp.nframes == nframes && sp.groups == groups)) {
        // not predicted (the first launches of a run, a jump): resolve it in front of the render
        const uint32_t k0 = L.c_hi ? L.c_lo / groups : 0u, k1 = L.c_hi ? sh::div_up(L.c_hi, groups) : 0u;      // (found by bank_render)
        rc = grow_tile_set(b->tile_block[ks], T, b->tile_carved[ks], ntiles, b->nvoices, groups, (k1 - k0) * groups, st);
        if (rc) return rc;
        T.k0 = k0; T.k1 = k1;
        P.tiles = T;
        const LaunchSet own = launch_set(b, b->cur);
        rc = launch_prepare_tiles(st, P, T, b->nvoices, start, nframes, records_deferred ? &own : nullptr);
        if (rc) return rc;
    } else if (records_deferred) {
        rc = prepare_chunks_now(b, start, nframes, st);      // (a predicted tile set without its records: not a state the pipeline produces)
        if (rc) return rc;
    }
    sp.valid = false;
    // the tile set of the block expected two launches on: set (n + 2) % 4 -- read last by launch n - 2, the launch before on
    // this stream -- resolved by workgroups of this launch's lean kernel
    if (L.next.launch) {
        const int k2 = (int)((b->tile_count + 2) % sh_bank::NTILESETS);
        TileSet& T2 = b->tile_set[k2];
        uint32_t k0 = 0, k1 = 0;
        k_range(next_start, k0, k1);
        rc = grow_tile_set(b->tile_block[k2], T2, b->tile_carved[k2], ntiles, b->nvoices, groups, (k1 - k0) * groups, st);
        if (rc) return rc;
        T2.k0 = k0; T2.k1 = k1;
        P.next_tiles = T2;
        P.next_ntiles = ntiles;
        uint32_t in_range = (T2.k1 - T2.k0) * groups;                 // chunks of the range (the last slot's may not all exist)
        if (T2.k1 * groups > nchunks) in_range -= T2.k1 * groups - nchunks;
        P.next_tile_wgs = in_range * sh::div_up(sh::div_up(ntiles, TILES_PER_WAVE), 4);
        sh_bank::TileSpec& s2 = b->tile_spec[k2];
        s2.valid = true; s2.start = next_start; s2.nframes = nframes; s2.groups = groups;
    }
    P.tiles = T;
    const LaunchArgs A = L.args(P, L.cur);
    // A short launch (real-time chunks: a handful of tiles) is ONE kernel -- the lean tiles' workgroups and, in rows behind them, the
    // general pairs' and the ones that resolve the next-but-one tile set: the chip has room for all of them at once, the launch is
    // latency-bound, and a second kernel costs the host and the stream more than its work; the general code in the same kernel costs
    // the lean loop registers, which a long launch cannot afford and a short one does not notice.
    // (Measured in round 4: the merged kernel for LONG launches too -- no general kernel between two lean kernels of a stream -- 50.7
    // against 48.4 us per block of the staggered row; with the general workgroups' code behind a function call, so that it would not
    // cost the lean pairs' loop registers, 530 us: the call's stack frame makes every wavefront of the dispatch a scratch user.)
    // Waves per SIMD the kernels are compiled for (round 4, same-box A/Bs): the merged kernel 3 (168 registers: 116 B of scratch instead of
    // 328; a 4096-frame chunk 10.9 -> 10.6 us; at 2 -- no scratch at all -- 10.8), the long-launch kernel with the waveform branch 3 (a
    // staggered table of FM Sine notes 64.9 -> 63.0 us), the long-launch Harmonics kernel 4 (at 3: 45.5 instead of 41.0 us per block).
    const bool merged = tiles <= 16;
    if (merged) {
        const uint32_t behind = tiles * GEN_SPLIT + P.next_tile_wgs;   // general workgroups, then the tile-set prepare workgroups
        if (b->tile_waveforms)
            hipLaunchKernelGGL((k_render_tiles<4, 8, 3, true, true>), dim3(tiles, groups + sh::div_up(behind, tiles)), dim3(256), 0, st,
                               A, L.next_args(L.next, behind), L.fold, L.parts, L.gen_valid);
        else
            hipLaunchKernelGGL((k_render_tiles<4, 8, 3, false, true>), dim3(tiles, groups + sh::div_up(behind, tiles)), dim3(256), 0, st,
                               A, L.next_args(L.next, behind), L.fold, L.parts, L.gen_valid);
    } else {
        // (where the next-but-one tile set is resolved was moved three times in round 3 -- between the lean workgroups, a kernel of its
        // own on a third stream, the general kernel -- before it ended in rows behind the lean kernel's voice groups: CHANGELOG item 34)
        const uint32_t behind = P.next_tile_wgs;
        LaunchSet nx = L.next;
        if (!behind) nx.launch = nullptr;
        const dim3 grid(tiles, groups + sh::div_up(behind, tiles));
        if (b->tile_waveforms)
            hipLaunchKernelGGL((k_render_tiles<4, 8, 3, true, false>), grid, dim3(256), 0, st, A, L.next_args(nx, behind), L.fold, L.parts, L.gen_valid);
        else
            hipLaunchKernelGGL((k_render_tiles<4, 8, 4, false, false>), grid, dim3(256), 0, st, A, L.next_args(nx, behind), L.fold, L.parts, L.gen_valid);
    }
    SH_CHECK_LAUNCH("k_render_tiles");
    if (!merged) {   // the general pairs: GEN_SPLIT workgroups per 256-frame tile, each all voice groups' pairs of it
        // (on a stream of its own beside the lean kernel it was slower, 95 against 75 us per block: five streams share four
        // hardware queues, and a kernel that waits for an event holds up whatever shares its queue)
        hipLaunchKernelGGL((k_render_general<4, 4, 4, GEN_TILES>), dim3(sh::div_up(nframes, 256) * GEN_SPLIT), dim3(256), 0, st, A, L.parts, L.gen_valid);
        SH_CHECK_LAUNCH("k_render_general(tiles)");
    }
    b->tile_count += 1;
    return SH_OK;
}

// A transition launch cut into SEGMENTS: one batched prepare, the lean kernel over all segments, the general kernel (the first
// segment's groups split SUB ways) and the combine of its slices.
static int launch_segmented(const RenderLaunch& L, uint32_t nseg, const uint32_t* seg_first) {
    sh_bank* b = L.b;
    hipStream_t st = L.st;
    const uint32_t nframes = L.nframes, groups = L.groups;
    sh::counters().segmented_launches += 1;
    const int ks = L.use_aux ? 1 : 0;
    LaunchSet& g = b->seg_set[ks];
    int rc = grow_segment_sets(b->seg_block[ks], g, b->seg_cap[ks], nseg, b->nvoices);
    if (rc) return rc;
    BankPtrs P = ptrs(b);
    P.nseg = nseg;
    for (uint32_t k = 0; k <= nseg; ++k) P.seg_first[k] = seg_first[k];
    rc = launch_prepare_segments_var(st, true, P, g, b->nvoices, nseg, L.start);
    if (rc) return rc;
    uint32_t tiles_lean = 0, tiles_gen = 0;
    for (uint32_t k = 0; k < nseg; ++k) {
        tiles_lean += sh::div_up(seg_first[k + 1] - seg_first[k], 64 * 8);
        tiles_gen += sh::div_up(seg_first[k + 1] - seg_first[k], 64 * 4);
    }
    // the first segment's groups are split SUB ways (BankPtrs::gen_sub): with sixteen waves per workgroup a wave walks two
    // or three of the 128 voices of its group
    const uint32_t SUB = 4;
    const uint32_t n0 = seg_first[1];
    rc = sh::grow_pooled(b->seg_scratch[ks], (size_t)groups * SUB * n0 * sizeof(double2));
    if (rc) return rc;
    P.gen_sub = SUB;
    P.gen_scratch = (double2*)b->seg_scratch[ks].ptr;
    const LaunchArgs A = L.args(P, g);
    const NextArgs N = L.next_args(L.next, L.prep_wgs);
    const dim3 grid(tiles_lean, groups + sh::div_up(L.prep_wgs, tiles_lean));
    rc = launch_render_lean(484, L.mode == COMBINED_LEAN_HARM ? LEAN_K_HARM : LEAN_K_ALL, true, grid, st, A, N, L.fold, L.parts);
    if (rc) return rc;
    SH_HIP(hipMemsetAsync(L.gen_valid, 1, (size_t)groups * sizeof(uint32_t), st));         // every group's general parts are written
    hipLaunchKernelGGL((k_render_general<16, 4, 1, GEN_SEG>), dim3(tiles_gen + (SUB - 1) * sh::div_up(n0, 64 * 4), groups), dim3(1024), 0, st,
                       A, L.parts, L.gen_valid);
    SH_CHECK_LAUNCH("k_render_general(segments)");
    hipLaunchKernelGGL(k_seg_combine, dim3(sh::div_up(n0, 256), groups), dim3(256), 0, st, (const double2*)b->seg_scratch[ks].ptr, SUB, n0,
                       L.parts + (size_t)groups * nframes, nframes);
    SH_CHECK_LAUNCH("k_seg_combine");
    return SH_OK;
}

// A plain launch: the lean kernel of a split launch (+ the general lists' kernel where the launch can hold a general voice), or the
// combined kernel.  Shapes <waves, frames per lane, min waves per SIMD>: 484 / 444 banks of >= 128 mostly lean voices (long / short
// blocks), 844 other large banks, 821 / 421 / 211 small ones (64 .. 127, 8 .. 63, < 8 voices).
static int launch_plain(const RenderLaunch& L) {
    sh_bank* b = L.b;
    hipStream_t st = L.st;
    const LaunchArgs A = L.args(ptrs(b), L.cur);
    const NextArgs N = L.next_args(L.next, L.prep_wgs);
    const dim3 grid(L.tiles, L.groups + sh::div_up(L.prep_wgs, L.tiles));
    const bool lean_split = L.split;                     // (split implies lean candidates and several voice groups)
    const bool fm_only = b->lean_fmsine_candidates == b->lean_candidates;     // every lean candidate an FM Sine voice (BASELINE config 3)
    if (lean_split) {                                    // every k_render_lean lives in osc_render_lean.hip
        const int rc = launch_render_lean(L.var, L.mode == COMBINED_LEAN_HARM ? LEAN_K_HARM : (fm_only ? LEAN_K_FM : LEAN_K_ALL), false, grid, st, A, N, L.fold, L.parts);
        if (rc) return rc;
    } else {                                             // ... and every k_render_combined in osc_render_combined.hip
        const int rc = launch_render_combined(L.var, L.mode, grid, st, A, N, L.fold, L.parts, L.out);
        if (rc) return rc;
    }
    if (L.with_general) {
        // the general lists of the same launch: four waves x four frames per lane whatever the lean kernel's shape (the parts
        // are indexed by frame), same voice groups, behind the lean kernel on the same stream
        hipLaunchKernelGGL((k_render_general<4, 4, 4, GEN_LISTS>), dim3(sh::div_up(L.nframes, 256), L.groups), dim3(256), 0, st, A, L.parts, L.gen_valid);
        SH_CHECK_LAUNCH("k_render_general(lists)");
    }
    return SH_OK;
}


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
    const size_t n32 = (size_t)nframes * 8, n64 = (size_t)nframes * 16, n16 = (size_t)nframes * 4;
    sh::State& S = sh::state();
    const sh::Knobs& K = sh::knobs();
    // variant = WAVES*100 + FPL*10 + MINW (SYNTHHIP_VARIANT overrides the tuned default)
    int var = K.variant;
    const int mode = b->lean_candidates == 0 ? COMBINED_DIRECT : (b->lean_fm_candidates ? COMBINED_LEAN_ALL : COMBINED_LEAN_HARM);
    // mostly-lean banks take eight frames per lane on long blocks: the recurrences make every frame after the second cost one or
    // two FMAs of trigonometry (one table lookup per eight frames)
    if (var == 0) var = b->nvoices >= 128 ? (2 * b->lean_candidates >= b->nvoices ? (nframes >= 16384 ? 484 : 444) : 844)
                                          : (b->nvoices >= 64 ? 821 : (b->nvoices >= 8 ? 421 : 211));
    // a LONG launch of a small, mostly lean bank (a run of blocks in one launch: sh_bank_render_run; a whole piece rendered at once) is
    // throughput work, not a latency problem: 64 voices x 480 000 frames at two frames per lane are 3750 workgroups of eight waves that
    // each pay a table lookup per two frames -- at eight frames per lane the recurrences carry six of them (46.7 -> us per run of ten
    // one-second blocks of BASELINE config 2)
    if (K.variant == 0 && b->nvoices < 128 && b->nvoices >= 16 && nframes >= (1u << 18) && 2 * b->lean_candidates >= b->nvoices) var = 484;
    // A bank whose notes do not move in lock-step takes the tile-classified launch (below) at EVERY block length: in real-time
    // chunks (256 .. 4096 frames) the general code walked the whole table for every tile -- a table of 22 528 notes, ~780 of them
    // sounding: 60 .. 85 us per chunk where the arithmetic is 2 us.  The lean tiles kernel has one shape (four waves, 512-frame
    // tiles); a short launch is a few tiles of it, with more voice groups (but not hundreds: their partial buses are folded
    // frame by frame).
    const bool tile_candidate = K.variant == 0 && b->tile_all && b->nvoices >= 128 && mode != COMBINED_DIRECT && !K.no_tiles && !b->needs_rows && b->first_row_voice < 0 &&
                                (b->has_onsets || b->own_envelopes) && !b->no_general_voice(start, nframes);
    if (tile_candidate) var = 484;
    // SIXTEEN frames per lane (var 4163; round 4): what a (wave, voice) pair pays once -- the record's scalar loads, the table lookup,
    // the rotation: ~40 of the ~190 VALU instructions of eight frames -- is paid once per sixteen, ~10 % fewer instructions per
    // voice-sample.  The 32 float64 accumulator pairs do not fit 128 registers beside the loop (at 128 the compiler spills inside the
    // Horner chains: 63 us per block), so the shape runs at THREE waves per SIMD (167 registers): 41.6 instead of 46 us for a launch
    // alone, 35.4-35.7 instead of 35.8-35.9 us in a stream of launches (fewer instructions, but a CU holds three workgroups, not four:
    // less of the neighbouring launch beside it).  TWELVE frames per lane fit 128 registers without a spill in the loop -- 7 % fewer
    // instructions at four waves per SIMD -- and ran 36.5-36.7 us: slower than eight (36.0).  Fewer instructions do not buy time in
    // the stream of launches; the clock the chip holds falls as the pipe fills (2.38 GHz with one wavefront per SIMD, 2.12-2.18 with
    // two launches in flight: profiles/r04_headline_phases.md).  Tiles of 1024 frames need twice the voice groups to fill the chip (47 tiles x 16
    // groups of 64 voices for the headline: the same 752 workgroups), so: polynomial-Harmonics banks and FM Sine banks (the lean kernels
    // of these kinds are the ones instantiated at this shape; config 3: 1.8 % faster than at eight frames, 72 B of scratch), a split
    // launch, and enough 64-voice chunks for the groups.
    const bool fm_bank = mode == COMBINED_LEAN_ALL && b->lean_candidates > 0 && b->lean_fmsine_candidates == b->lean_candidates;
    if (K.variant == 0 && var == 484 && (mode == COMBINED_LEAN_HARM || fm_bank) && !tile_candidate && !K.no_split && K.groups == 0) {
        const uint32_t tiles16 = sh::div_up(nframes, 1024u);
        uint32_t g16 = 1;
        while (tiles16 * g16 * 2 <= 1024u && b->nvoices / (g16 * 2) >= 64u) g16 *= 2;         // (whole chunks per group)
        // a LONG launch (a run of blocks in one: sh_bank_render_run) has tiles enough for the chip with ONE voice group -- which would
        // make it a combined-kernel launch at eight frames per lane, general code and all (37.9 us per block where the stream of
        // one-block launches needs 36.4): two groups keep it a split launch of the sixteen-frame lean kernel
        if (g16 == 1 && tiles16 >= 640u && b->nvoices >= 128u) g16 = 2;
        if (g16 >= 2 && tiles16 * g16 >= 640u) var = 4163;
    }
    const int W = var >= 1000 ? var / 1000 : var / 100, F = var >= 1000 ? (var / 10) % 100 : (var / 10) % 10;
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
    if (tile_candidate && groups > 32) groups = 32;
    if (var == 4163 && K.variant == 0 && groups < 2) groups = 2;      // (the long launch above)
    // ... and the long launch of any other bank of >= 128 voices with lean candidates (mixed kinds; eight frames per lane): two groups, so
    // that it is a split launch -- lean lists in the lean kernel, the general lists beside it -- like its one-block launches
    if (K.variant == 0 && W == 4 && groups == 1 && mode != COMBINED_DIRECT && b->nvoices >= 128 && tiles >= 640 && !tile_candidate) groups = 2;
    if (K.groups > 0) groups = (uint32_t)K.groups;
    uint32_t vpg = (b->nvoices + groups - 1) / groups;
    if (groups > 1) vpg = (vpg + 63) & ~63u;                // groups are made of whole 64-voice chunks (the lists' unit)
    groups = (b->nvoices + vpg - 1) / vpg;                  // no empty trailing groups
    // A RUN: consecutive renders of a bank (same shape, the next block each time) alternate between two HIP streams so that the
    // tail of one launch -- its launch latency, for a small bank: nearly all of it -- is filled by the next; with several voice
    // groups a launch also folds the partial buses of the launch before its predecessor.  A single-group launch writes the
    // caller's buffers itself, so it may only run beside its predecessor when the two do not write the same memory (a ring of
    // bus buffers; a caller that renders every block into ONE buffer stays on one stream, as in round 2).
    const bool may_pipeline = !K.no_speculation && !K.no_overlap;
    const bool direct = groups == 1;
    bool pipelined = groups > 1 || (may_pipeline && !K.no_small_pipeline);
    bool cont = pipelined && b->run_active && b->run_nframes == nframes && b->run_groups == groups &&
                b->run_tile == (uint32_t)(64 * F) && b->run_next_start == start;
    if (direct && pipelined) {
        // memory the previous launch of this bank wrote directly: beside it (other stream) this launch must not write there
        const bool same = (o32 && b->overlaps_last_direct(o32, n32)) || (o64 && b->overlaps_last_direct(o64, n64)) ||
                          (o16 && b->overlaps_last_direct(o16, n16));
        if (same) { pipelined = false; cont = false; }
        // ... and memory an older launch of the run wrote on the OTHER stream (a ring with an odd number of buffers): new run
        else if (cont) {
            const uint8_t other = (b->run_count & 1) ? 1u : 2u;      // this launch goes to stream2 when run_count is odd: the other is `stream`
            if (overlaps_writes(b, true, other, o32, n32) || overlaps_writes(b, true, other, o64, n64) || overlaps_writes(b, true, other, o16, n16))
                cont = false;
        }
    }
    // an output buffer another bank's launches may still write (a fold it owes, a fold one of its launches is doing right now on
    // the other stream, a direct write): the order of the writes must be the callers' -- end every run
    if (overlaps_writes(b, false, 3, o32, n32) || overlaps_writes(b, false, 3, o64, n64) || overlaps_writes(b, false, 3, o16, n16) ||
        bank_writes().size() > 256) {
        rc = sh::flush_pending();
        if (rc) return rc;
        cont = false;
    }
    if (!cont) {
        // this bank's own leftovers are folded now; the other banks' runs go on
        if (b->npending || b->run_active) {
            rc = join_aux();
            if (!rc) rc = fold_bank(b);
            if (rc) return rc;
        }
        if (pipelined) {
            b->run_active = true;
            S.open_runs += 1;
            b->run_nframes = nframes;
            b->run_groups = groups;
            b->run_tile = (uint32_t)(64 * F);
            b->run_count = 0;
            SH_HIP(hipEventRecord(S.ev_join, S.stream));    // everything enqueued so far: stream2's first launch of the run waits for it
        }
    }
    const uint32_t n = pipelined ? b->run_count : 0;
    const bool two_streams = pipelined && may_pipeline;
    const bool use_aux = two_streams && (n & 1);
    hipStream_t st = use_aux ? S.stream2 : S.stream;
    if (use_aux && n == 1) SH_HIP(hipStreamWaitEvent(S.stream2, S.ev_join, 0));
    const int prev_cur = b->cur;
    // a bank with lean candidates + several voice groups: the launch is split into a lean and a general kernel
    // (k_render_lean + k_render_general); SYNTHHIP_NO_SPLIT=1 keeps the combined kernel
    // (the shapes of banks of fewer than 64 voices -- 421, 211 -- have no lean kernel: such banks never have two voice groups by
    //  themselves; forced on a larger bank by SYNTHHIP_VARIANT they render through the combined kernel)
    const bool split = mode != COMBINED_DIRECT && groups > 1 && !K.no_split && var != 421 && var != 211;
    // A transition launch (segmented): cut where the envelopes become flat (every voice past its decay) and from there on
    // into segments no longer than their own distance from the note's start -- piece ends of the phase sum lie an octave apart,
    // so such a segment crosses at most one per voice; a cut, too, where the first voice leaves its sustain.  Tile-aligned cuts.
    // A TILE-CLASSIFIED launch: banks whose notes do not move in lock-step -- onsets, envelope corners of their
    // own -- have no corners to cut a launch at, and nearly every voice holds a corner somewhere in a one-second block; but nearly
    // every (voice, 512-frame tile) pair lies on one envelope line and one piece of the phase table.  Those pairs take the lean
    // loop, the others the general code for that tile only (see TileRec).
    bool tiled = false;
    uint32_t tile_c_lo = 0, tile_c_hi = 0;
    if (split && tile_candidate && var == 484) {
        // (a tile set holds records for the RANGE of chunks that can sound in the block; a block in which more than 2 GB worth of them
        // do -- a million notes at once -- goes through the general code)
        tile_c_lo = sh::div_up(b->nvoices, 64);
        for (uint32_t c = 0; c < sh::div_up(b->nvoices, 64); ++c)
            if (!(start + nframes <= b->chunk_span[2 * c] || start >= b->chunk_span[2 * c + 1])) { tile_c_lo = c < tile_c_lo ? c : tile_c_lo; tile_c_hi = c + 1; }
        const uint64_t range = tile_c_hi > tile_c_lo ? (uint64_t)(tile_c_hi - tile_c_lo) + 2 * groups : 0;
        tiled = (uint64_t)sh::div_up(nframes, TILE_FRAMES) * range * 64 * sizeof(TileRec) <= ((uint64_t)1 << 31);
    }
    uint32_t seg_first[SEG_MAX + 1];
    uint32_t nseg = 0;
    if (!tiled && split && (var == 484 || var == 4163) && b->all_lean && !K.no_seg && !b->needs_rows && !b->no_general_voice(start, nframes)) {
        nseg = plan_segments(b, start, nframes, (uint64_t)(64 * F), ~0ull, true, seg_first);
        if (nseg < 2 || seg_first[nseg] != nframes) nseg = 0;        // nothing to cut, or more cuts than a launch carries
    }
    bool records_deferred = false;          // (a tile-classified launch without a resolved record set: its classification resolves what it needs)
    // A plain split launch whose records nothing has resolved (a render that stands alone, the first of a run, a jump) CAN resolve them in
    // the lean kernel itself (LaunchArgs::self_prepare: every workgroup its own voice group's records) and fold its partial buses itself
    // (FoldIn::self).  Both are OFF (SYNTHHIP_SELF = 1 both, 2 the fold, 3 the records): round 6 built them for VERDICT r05 item 3 (the
    // lone call, 56 -> <= 45 us) and measured 61 us with both, 63 with the fold (planes stored and loaded at agent scope; with fences at
    // agent scope instead: 123 us -- every workgroup writes back and invalidates its XCD's L2), 57 with the records (and +7 us per
    // launch of two or more blocks: hundreds of workgroups resolving the same sixteen chunks) against 57 for the prepare kernel +
    // k_bus_combine: profiles/r06_run_lengths.txt.
    const bool plain_split = split && !tiled && nseg == 0 && K.self != 0;
    bool self_prepare = false;
    if (nseg == 0) {
        rc = acquire_records(b, start, nframes, st, cont, tiled, tiled ? &records_deferred : (plain_split && K.self != 2) ? &self_prepare : nullptr);
        if (rc) return rc;
    }
    // partial buses: ring slot n % 4 (last read by the fold in launch n - 2, which is this stream's previous launch)
    double2* parts = nullptr;
    if (groups > 1) {
        const int k = (int)(n & 3);
        // split launch: the general kernel's parts follow the lean kernel's, then one flag per group
        const size_t need = split ? 2 * (size_t)groups * nframes * sizeof(double2) + (size_t)groups * sizeof(uint32_t)
                                  : (size_t)groups * nframes * sizeof(double2);
        rc = sh::grow_pooled(b->parts[k], need);
        if (rc) return rc;
        parts = (double2*)b->parts[k].ptr;
    }
    // ... and the general-lists kernel is only launched when the launch CAN hold a general voice
    const bool with_general = split && !b->no_general_voice(start, nframes);
    uint32_t* gen_valid = with_general ? (uint32_t*)(parts + 2 * (size_t)groups * nframes) : nullptr;
    // the fold this launch takes over: the older of two outstanding ones (launch n - 2's)
    const bool take_over = b->npending == 2;
    // A launch that does not continue a run and has nothing to take over folds ITSELF (FoldIn::self): the last workgroup of a tile to
    // store its plane folds the tile -- a render that stands alone is complete when its one kernel ends (no k_bus_combine behind it),
    // and the first launch of a run owes nobody a fold.  Only where the lean kernel writes every plane (no general-lists kernel behind it).
    const bool self_fold = plain_split && K.self != 3 && groups > 1 && !cont && b->npending == 0 && !with_general && tiles <= sh_bank::SELF_TILES;
    const sh::PendingCombine prev = take_over ? b->pending[0] : sh::PendingCombine();
    const double2* pv_parts = take_over ? (const double2*)prev.parts : nullptr;
    float2* pv32 = take_over ? (float2*)prev.o32 : nullptr;
    double2* pv64 = take_over ? (double2*)prev.o64 : nullptr;
    uint32_t* pv16 = take_over ? (uint32_t*)prev.o16 : nullptr;
    const double pv_scale = take_over ? prev.scale : 0.0;
    const uint32_t* pv_gen = take_over ? (const uint32_t*)prev.gen_valid : nullptr;
    b->last_groups = groups;
    b->last_tiled = tiled;
    const LaunchSet cur = launch_set(b, b->cur);
    // the records of the block two launches on go to a set that is neither this launch's, nor its predecessor's (perhaps
    // still executing), nor the one holding the block in between
    const uint64_t next_start = start + 2 * (uint64_t)nframes;
    int target = -1;
    if (!K.no_speculation) {
        for (int c = 0; c < sh_bank::NSETS && target < 0; ++c) {
            const bool holds_between = b->spec[c].valid && b->spec[c].start == start + nframes && b->spec[c].nframes == nframes;
            if (c != b->cur && !(cont && (c == prev_cur || c == b->last_target)) && !holds_between) target = c;
        }
    }
    LaunchSet next = launch_set(b, target < 0 ? 0 : target);
    if (target < 0) next.launch = nullptr;
    // the workgroups that resolve those records: one wavefront per chunk of 64 voices, in rows of the grid behind the voice groups'
    const uint32_t nchunks = sh::div_up(b->nvoices, 64);
    const uint32_t prep_wgs = next.launch ? nchunks : 0u;
    const BusOut out{o32, o64, o16, pcm_scale};
    const FoldIn fold = self_fold ? FoldIn{parts, nullptr, out, b->d_self()} : FoldIn{pv_parts, pv_gen, BusOut{pv32, pv64, pv16, pv_scale}, nullptr};
    const RenderLaunch L{b, st, start, next_start, nframes, tiles, groups, vpg, nchunks, prep_wgs, mode, var, split, with_general, use_aux,
                         self_prepare, !cont && !K.no_ladder, cur, next, out, parts, fold, gen_valid, tile_c_lo, tile_c_hi};
    rc = tiled ? launch_tiled(L, records_deferred) : nseg ? launch_segmented(L, nseg, seg_first) : launch_plain(L);
    if (rc) return rc;
    if (use_aux) S.aux_busy = true;                         // (join_aux records the event the main stream waits for)
    const uint8_t stream_bit = use_aux ? 2u : 1u;
    if (take_over) {                                        // folded by this launch -- which may be running for a while yet: the
        note_write(b, prev.o32, (size_t)prev.nframes * 8, stream_bit);           // buses it folds into stay on record (they are: from
        note_write(b, prev.o64, (size_t)prev.nframes * 16, stream_bit);          // the launch that rendered them; now with this stream)
        note_write(b, prev.o16, (size_t)prev.nframes * 4, stream_bit);
        b->pending[0] = b->pending[1];
        b->npending = 1;
        S.pending_total -= 1;
    }
    const bool writes_itself = direct || self_fold;        // the launch writes the caller's buses on its own stream
    if (groups > 1 && !self_fold) {                         // this launch's partial buses: folded two launches on, or by the next other API call
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
    }
    if (S.open_runs > 0) {
        // on record while any run is open (this bank's or another's): direct writes with their stream; the buses a pending fold
        // will write from now (another bank must not slip a write in between), the stream follows when a launch takes the fold over
        note_write(b, o32, n32, writes_itself ? stream_bit : 0);
        note_write(b, o64, n64, writes_itself ? stream_bit : 0);
        note_write(b, o16, n16, writes_itself ? stream_bit : 0);
    }
    if (pipelined) {
        b->run_count = n + 1;
        b->run_next_start = start + nframes;
    }
    b->last_direct[0] = b->last_direct[1] = b->last_direct[2] = sh_bank::Range{nullptr, nullptr};
    if (direct) {
        if (o32) b->last_direct[0] = sh_bank::Range{(const char*)o32, (const char*)o32 + n32};
        if (o64) b->last_direct[1] = sh_bank::Range{(const char*)o64, (const char*)o64 + n64};
        if (o16) b->last_direct[2] = sh_bank::Range{(const char*)o16, (const char*)o16 + n16};
    }
    b->last_target = target;
    if (target >= 0) {
        b->spec[target].valid = true;
        b->spec[target].start = next_start;
        b->spec[target].nframes = nframes;
        b->spec[target].sparse = tiled;
    }
    return SH_OK;
}

// Renders of any length: the launch's x dimension is a tile count (tiles x threads per workgroup must stay below 2^32 work-items,
// which the shapes of small banks reach first: sixteen waves on 128 frames), and a bank with several voice groups keeps
// 32 bytes of partial buses per frame and group.  Long renders are therefore a run of launches of RENDER_MAX_FRAMES (87 s
// at 48 kHz; consecutive blocks of one shape: the two-stream pipeline applies) into views of the caller's buffers.
constexpr uint32_t RENDER_MAX_FRAMES_ANY = 1u << 22;
static int bank_render_any(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* bus_f32, sh_buf* bus_f64, sh_buf* pcm_i16, double pcm_scale) {
    // A table of notes (tile-classified launches) is rendered in launches of at most 2^17 frames: a tile set holds a record per
    // (tile, sounding voice), and a launch of minutes of audio would not get one (it would fall back to the general code for
    // every voice of the table); everything else in launches of at most RENDER_MAX_FRAMES (the 32-bit work-item count of a dispatch).
    SH_API_LOCK();     // (recursive) taken BEFORE the bank is looked at -- sh_bank_set_rows on another thread changes needs_rows /
                       // first_row_voice -- and held across the launches of one call: nothing else gets between them
    const bool notes = b && b->tile_all && b->nvoices >= 128 && (b->has_onsets || b->own_envelopes) && !b->needs_rows && b->first_row_voice < 0 &&
                       !sh::knobs().no_tiles;
    const uint32_t RENDER_MAX_FRAMES = notes ? (1u << 17) : RENDER_MAX_FRAMES_ANY;
    if (nframes <= RENDER_MAX_FRAMES) return bank_render(b, start, nframes, bus_f32, bus_f64, pcm_i16, pcm_scale);
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

int sh_bank_render_run(sh_bank* b, uint64_t start, uint32_t nframes, uint32_t nblocks, sh_buf* const* bus_f32, sh_buf* const* pcm_i16,
                       uint32_t nring, double pcm_scale) {
    if (!b || (!bus_f32 && !pcm_i16) || nring == 0) return sh::set_error(SH_ERR_INVALID, "sh_bank_render_run: NULL argument");
    if (nframes == 0 || nblocks == 0) return SH_OK;
    for (uint32_t k = 0; k < nring && k < nblocks; ++k) {
        if (bus_f32 && (!bus_f32[k] || bus_f32[k]->bytes < (size_t)nframes * 8)) return sh::set_error(SH_ERR_INVALID, "sh_bank_render_run: bus_f32[%u] missing or too small", k);
        if (pcm_i16 && (!pcm_i16[k] || pcm_i16[k]->bytes < (size_t)nframes * 4)) return sh::set_error(SH_ERR_INVALID, "sh_bank_render_run: pcm_i16[%u] missing or too small", k);
    }
    SH_API_LOCK();                                            // one acquisition for the whole run
    // Blocks whose buffers lie back to back in memory (windows of one allocation: the usual shape of a ring) are ONE launch: a
    // launch costs the host 2-4 us whatever it renders, a small bank's block less than that on the device.  At most
    // RUN_MAX_FRAMES per launch, and never across the ring's wrap (slot nring - 1 is followed by slot 0 somewhere else).
    constexpr uint64_t RUN_MAX_FRAMES = 1u << 20;
    int rc = SH_OK;
    uint32_t k = 0;
    while (k < nblocks && !rc) {
        const uint32_t slot = k % nring;
        uint32_t m = 1;
        while (k + m < nblocks && slot + m < nring && (uint64_t)(m + 1) * nframes <= RUN_MAX_FRAMES) {
            const bool adj32 = !bus_f32 || (char*)bus_f32[slot + m]->ptr == (char*)bus_f32[slot]->ptr + (size_t)m * nframes * 8;
            const bool adj16 = !pcm_i16 || (char*)pcm_i16[slot + m]->ptr == (char*)pcm_i16[slot]->ptr + (size_t)m * nframes * 4;
            if (!adj32 || !adj16) break;
            ++m;
        }
        sh_buf v32{nullptr, 0, false, 0}, v16{nullptr, 0, false, 0};
        if (bus_f32) { v32.ptr = bus_f32[slot]->ptr; v32.bytes = (size_t)m * nframes * 8; }
        if (pcm_i16) { v16.ptr = pcm_i16[slot]->ptr; v16.bytes = (size_t)m * nframes * 4; }
        rc = bank_render_any(b, start + (uint64_t)k * nframes, m * nframes, bus_f32 ? &v32 : nullptr, nullptr, pcm_i16 ? &v16 : nullptr, pcm_scale);
        k += m;
    }
    return rc;
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

#ifdef SH_DIAG
int sh_debug_diag2(void* host, size_t bytes) {
    if (hipDeviceSynchronize() != hipSuccess) return SH_ERR_HIP;
    const size_t n = bytes < sizeof(g_diag2) ? bytes : sizeof(g_diag2);
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_diag2), n, 0, hipMemcpyDeviceToHost) == hipSuccess ? SH_OK : SH_ERR_HIP;
}
// (diagnostic builds only) the wavefronts' timestamps of the last launches: 4 banks x DIAG_WAVES x DIAG_SLOTS uint64
int sh_debug_diag(void* host, size_t bytes) {
    if (hipDeviceSynchronize() != hipSuccess) return SH_ERR_HIP;
    const size_t n = bytes < sizeof(g_diag) ? bytes : sizeof(g_diag);
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_diag), n, 0, hipMemcpyDeviceToHost) == hipSuccess ? SH_OK : SH_ERR_HIP;
}
#endif

}  // extern "C"
