// osc_host.hpp -- host side of the oscillator bank shared by osc_*.hip: the bank object and what the translation units
// call across each other.
#pragma once
#include "osc_device.hpp"
#include <vector>
#include <algorithm>

using namespace shosc;

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
    bool        has_guard = false;         // some voice carries a guard list (sh_voice::guard_count): the int16 kernels with the boundary check
    // one arrival counter per tile of a SELF-FOLDING launch (a render that stands alone: the last workgroup of a tile to store its
    // partial bus folds the tile's planes itself -- no k_bus_combine behind the launch); behind d_hint's 2 * nvoices words, zero
    // between launches (the folding workgroup resets its counter)
    static constexpr uint32_t SELF_TILES = 16384;
    uint32_t*   d_self() const { return d_hint + 2 * (size_t)nvoices; }
    // This bank's run of pipelined renders (same shape, consecutive blocks), the folds it still owes, and its ring of
    // partial-bus buffers: launch n writes ring slot n % 4, launch n + 2 (same stream) folds it.  Per bank, so that two banks
    // rendering turn by turn (or a bank beside a DistVoiceBank's) each keep their pipeline: only calls that can touch a bus
    // buffer end the runs (sh::flush_pending).
    bool        run_active = false;
    uint64_t    run_next_start = 0;
    uint32_t    run_nframes = 0, run_groups = 0, run_tile = 0, run_count = 0;
    sh::PendingCombine pending[2];
    int         npending = 0;
    // Device blocks that only grow, from the buffer pool (sh::Pooled): a render call never allocates or synchronises once the
    // shapes it sees have been seen (sh_debug_counters)
    sh::Pooled  parts[4];                  // the ring of partial-bus buffers
    sh::Pooled  seg_block[2];              // record sets of a segmented transition launch (one per stream), seg_cap[k] segments each
    LaunchSet   seg_set[2] = {};
    uint32_t    seg_cap[2] = {0, 0};
    sh::Pooled  seg_scratch[2];            // ... and the slices of its first segment's general parts (BankPtrs::gen_scratch)
    // The tile sets of tile-classified launches: a ring of four.  Launch n uses set n % 4 and resolves, in workgroups of its own
    // render kernel, the set of the block expected two launches on (tile_spec says for which block): set (n + 2) % 4, last read by
    // launch n - 2 -- the launch before on the same stream.
    static constexpr int NTILESETS = 4;
    sh::Pooled  tile_block[NTILESETS];
    TileSet     tile_set[NTILESETS] = {};
    uint32_t    tile_carved[NTILESETS] = {0, 0, 0, 0};   // tiles the set was carved for
    struct TileSpec { bool valid = false; uint64_t start = 0; uint32_t nframes = 0, groups = 0; } tile_spec[NTILESETS];
    uint32_t    tile_count = 0;            // tile-classified launches so far
    struct Range { const char* lo; const char* hi; };
    Range       last_direct[3] = {};       // what the last launch wrote itself (single-group launches: float32 / float64 / PCM bus)
    bool        overlaps_last_direct(const void* p, size_t bytes) const {
        const char* lo = (const char*)p;
        for (const Range& r : last_direct) if (r.lo && lo < r.hi && r.lo < lo + bytes) return true;
        return false;
    }
    // modulation rows (sh_bank_set_rows / sh_bank_render_rows)
    int32_t*    d_fm_row = nullptr;
    int32_t*    d_pwm_row = nullptr;
    int32_t     fm_row_max = -1;
    bool        needs_rows = false;        // some voice reads a row: SH_FM_BUFFER, SH_BUFFER, or a pwm row was set
    const double* launch_rows = nullptr;   // set for the duration of one sh_bank_render_rows call
    size_t      launch_row_stride = 0;
    // record sets of a segmented materialisation (sh_bank_generate over long rows), gen_segs of them, allocated on demand
    sh::Pooled  gen_block;
    LaunchSet   gen_set = {};
    uint32_t    gen_segs = 0;
    double2*    d_seg_rot = nullptr;       // (cos, sin)(64*dt) per table piece
    std::vector<uint64_t> chunk_span;      // (host copy)
    uint64_t*   d_chunk_span = nullptr;    // [chunk][2]: first onset, last frame of sound + 1 of the chunk's voices
    double*     d_polys = nullptr;         // [slot][16]: the polynomial of a polynomial-Harmonics voice, by voice
    double2*    d_lfo_rot = nullptr;       // (cos, sin)(64*lfo_d) per voice
    VoiceLaunch* d_launch = nullptr;       // the set the next kernel reads
    VoiceFM*    d_launch_fm = nullptr;
    int         cur = 0;                   // the set the last launch read
    int         last_target = -1;          // the set the last render launch is filling (-1: none)
    // what each set holds (or will); sparse: resolved for a tile-classified launch -- the chunks whose voices are all silent in
    // the block were skipped (nobody reads them there), so only such a launch may take the set
    struct Spec { bool valid = false; uint64_t start = 0; uint32_t nframes = 0; bool sparse = false; } spec[NSETS];
    void        void_specs() { for (auto& q : spec) q.valid = false; last_target = -1; }
    uint32_t    lean_candidates = 0;      // voices that can take the lean loop in some launch (static properties)
    uint32_t    lean_fm_candidates = 0;   // ... of them other than polynomial Harmonics (FM Sine, plain waveforms)
    uint32_t    lean_fmsine_candidates = 0;   // ... of them FM Sine voices (all of the lean candidates: the FM-only lean kernel)
    bool        last_tiled = false;       // ... and whether it was tile-classified (no classification by voice then)
    uint32_t    last_groups = 0;          // voice groups of the last sh_bank_render launch (sh_bank_launch_stats)
    // When can a launch hold NO general voice (so that a split launch needs no general-lists kernel)?  Conservative, from
    // static properties: every voice is a lean candidate, every envelope is on its sustain piece for the
    // whole launch, and no phase-table piece shorter than the launch ends after its start (a launch then crosses at most one
    // piece end per voice).  short_piece_end[k] = the largest end of any piece shorter than 2^k samples.
    bool        all_lean = false;          // every voice is a lean candidate (of any lean kind)
    bool        has_onsets = false;        // some voice starts late (sh_voice::start_frame)
    bool        own_envelopes = false;     // the voices' envelope corners are too many to cut launches at (more than 16 distinct ones)
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
    // (nearly: nine in ten) every voice can be a lean pair of a tile-classified launch: polynomial Harmonics or a plain Sine / Sawtooth / Square /
    // Triangle / Pulse, no FM, no bias, not mirrored; tile_waveforms: some of them are no Harmonics / Sine (the tiles kernel with the
    // waveform branch)
    bool        tile_all = false, tile_waveforms = false;
    long long   first_row_voice = -1;  // the first voice that reads a modulation / sample row (SH_FM_BUFFER, SH_BUFFER); -1: none
};

namespace shosc {
LaunchSet launch_set(const sh_bank* b, int k);
BankPtrs ptrs(const sh_bank* b);
std::vector<sh_bank*>& live_banks();
const shm::sc_pair* trig_table();
// osc_bank.hip
int prepare_single(sh_bank* b, uint32_t first, uint32_t count, uint64_t start, uint32_t nframes);     // k_prepare: one voice (sh_osc_render)
// `deferred` (tile-classified launches): when no resolved set is found, one is chosen but NOT resolved -- *deferred = true -- because
// the classification of such a launch resolves the records it needs itself (launch_prepare_tiles with the set)
int acquire_records(sh_bank* b, uint64_t start, uint32_t nframes, hipStream_t launch_stream, bool in_run, bool accept_sparse = false, bool* deferred = nullptr);
int prepare_chunks_now(sh_bank* b, uint64_t start, uint32_t nframes, hipStream_t st);     // the set b->cur, on stream st
uint32_t plan_segments(const sh_bank* b, uint64_t start, uint32_t nframes, uint64_t T, uint64_t max_len, bool corners, uint32_t* seg_first);
int bank_check_plain(const sh_bank* b, const char* who);
// `nseg` record sets for `nvoices` voices carved out of one pool-backed block (grown when it is too small; *cap = sets it holds)
int grow_segment_sets(sh::Pooled& block, LaunchSet& g, uint32_t& cap, uint32_t nseg, uint32_t nvoices);
int launch_prepare_tiles(hipStream_t st, const BankPtrs& P, const TileSet& T, uint32_t nvoices, uint64_t start, uint32_t nframes, const LaunchSet* recs = nullptr);
// (`range_chunks`: the chunks a tile's row of records has to hold -- the set's range, see TileSet; the rows grow with headroom, never shrink)
int grow_tile_set(sh::Pooled& block, TileSet& T, uint32_t& carved_tiles, uint32_t ntiles, uint32_t nvoices, uint32_t groups, uint32_t range_chunks, hipStream_t st);
int launch_prepare_segments(hipStream_t st, const BankPtrs& P, const LaunchSet& base, uint32_t nvoices, uint32_t nseg, uint64_t start,
                            uint32_t nframes, uint32_t seg_frames);
int launch_prepare_segments_var(hipStream_t st, bool sloped, const BankPtrs& P, const LaunchSet& base, uint32_t nvoices, uint32_t nseg, uint64_t start);
// osc_render.hip
int fold_bank(sh_bank* b);
int join_aux();
}  // namespace shosc
