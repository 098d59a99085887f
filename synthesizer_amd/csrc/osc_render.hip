// osc_render.hip -- fused generate-and-mix of a voice bank into the stereo bus.
//
// Kernels
//   k_bank_render     block = W waves on one tile of 64*FPL frames; a wave walks its share of the
//                     lean list (record in SGPRs, one table lookup + rotations + Horner per voice) and of the general
//                     list (voice_block); float64 partial (L, R) per lane, LDS-staged sum across the waves
//   k_bus_combine     folds the voice groups' partial buses in group order, rounds to float32
//   k_seg_combine     the slices of a segmented launch's first segment
#include "osc_host.hpp"
#include <stdlib.h>

namespace {

// fused generate-and-mix.  grid = (frame tiles, voice groups); block = WAVES waves on ONE tile of 64*FPL
// frames; wave w walks voices v0+w, v0+w+WAVES, ... of its group with the voice record in SGPRs; float64
// partial (L, R) per lane; LDS-staged sum across the waves; one store per frame.  With one group the block
// writes the final bus; with several it writes a float64 partial bus per group and k_bus_combine folds
// them in group order -- either way voices are summed in a fixed order (reproducible run to run).
// MODE (a static property of the bank): RENDER_DIRECT -- no voice could ever take the lean loop: walk the voice table
// directly; RENDER_LEAN_HARM -- lean loop for polynomial Harmonics only; RENDER_LEAN_ALL -- also for FM Sine voices (kept
// out of the Harmonics-only kernel: the extra branch and code cost its loop 5 %).
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
       RENDER_LEAN_HARM_SEG = 6, RENDER_GENERAL_SEG = 7, RENDER_LEAN_ALL_SEG = 8,
       // RENDER_LEAN_TILES / RENDER_GENERAL_TILES: the split launch of a TILE-CLASSIFIED block (see TileRec): the lean kernel walks the
       // set bits of its tile's lean masks -- a 64-byte record per (voice, tile) + the voice's polynomial, the sloped lean arithmetic,
       // no piece ends, no corners -- the general kernel the set bits of the general masks, through the launch records.
       RENDER_LEAN_TILES = 9, RENDER_GENERAL_TILES = 10,
       // RENDER_LEAN_TILES_ALL: the lean tiles kernel of a bank that holds plain Sawtooth / Square / Triangle / Pulse voices too (the
       // waveform branch costs the Harmonics loop registers: an instantiation of its own, as with RENDER_LEAN_ALL)
       RENDER_LEAN_TILES_ALL = 11,
       // RENDER_TILES_MERGED: a SHORT tile-classified launch (real-time chunks: a handful of tiles) as ONE kernel -- the lean tiles'
       // workgroups, and in rows behind them the general pairs' workgroups and the ones that resolve the tile set of the block two
       // launches on.  The chip has room for all of them at once, the launch is latency-bound, and a second kernel costs the host
       // and the stream more than its work; the general code in the same kernel costs the lean loop registers, which a long
       // launch cannot afford (RENDER_LEAN_TILES + RENDER_GENERAL_TILES) and a short one does not notice.
       RENDER_TILES_MERGED = 12 };
constexpr bool mode_merged(int mode) { return mode == RENDER_TILES_MERGED; }
constexpr bool mode_tiles(int mode) { return mode == RENDER_LEAN_TILES || mode == RENDER_LEAN_TILES_ALL || mode == RENDER_TILES_MERGED; }
constexpr bool mode_lean_harm(int mode) { return mode == RENDER_LEAN_HARM || mode == RENDER_LEAN_HARM_ONLY || mode == RENDER_LEAN_HARM_SEG || mode_tiles(mode); }
constexpr bool mode_lean_only(int mode) { return mode == RENDER_LEAN_HARM_ONLY || mode == RENDER_LEAN_ALL_ONLY || mode == RENDER_LEAN_HARM_SEG || mode == RENDER_LEAN_ALL_SEG || mode_tiles(mode); }
constexpr bool mode_general(int mode) { return mode == RENDER_GENERAL_ONLY || mode == RENDER_GENERAL_SEG || mode == RENDER_GENERAL_TILES; }
constexpr bool mode_seg(int mode) { return mode == RENDER_LEAN_HARM_SEG || mode == RENDER_GENERAL_SEG || mode == RENDER_LEAN_ALL_SEG; }
constexpr bool mode_has_lean(int mode) { return mode != RENDER_DIRECT && !mode_general(mode); }
constexpr uint32_t GEN_SPLIT = 2;      // general workgroups per tile of a tile-classified launch (each writes a plane of general parts; <= groups)

// -DSH_DIAG (SYNTHHIP_BUILD_FLAGS, never the shipped library): every wavefront of a render launch leaves the 100 MHz timestamps of
// its phases and the SIMD it ran on in g_diag (four banks by block number: launches of a stream of blocks overlap pairwise);
// sh_debug_diag copies them out.  tools/headline_phases.py turns them into profiles/rNN_headline_phases.md.
#ifdef SH_DIAG
constexpr uint32_t DIAG_WAVES = 4096, DIAG_SLOTS = 10;
__device__ uint64_t g_diag[4 * DIAG_WAVES * DIAG_SLOTS];
__device__ __forceinline__ void diag_stamp(uint64_t start_, uint32_t nframes_, uint32_t slot) {
    const uint32_t w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && w < DIAG_WAVES) {
        const uint32_t bank = (uint32_t)((double)start_ * __builtin_amdgcn_rcp((double)nframes_) + 0.5);     // the block's number (a stream of equal blocks)
        uint64_t* d = g_diag + ((size_t)(bank & 3) * DIAG_WAVES + w) * DIAG_SLOTS;
        d[slot] = __builtin_amdgcn_s_memrealtime();
        if (slot == 0) {
            uint32_t hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[6] = (uint64_t)hw | ((uint64_t)xcc << 32);
            d[7] = __builtin_amdgcn_s_memtime();
            d[9] = ((uint64_t)gridDim.x << 32) | gridDim.y;
        }
        if (slot == 5) d[8] = __builtin_amdgcn_s_memtime();
    }
}
__device__ uint64_t g_diag2[64 * 8];     // -DSH_DIAG2: s_memtime at five points of the first voices of one wavefront's lean loop
#define SH_STAMP(slot) diag_stamp(start, nframes, slot)
#else
#define SH_STAMP(slot) ((void)0)
#endif

// lean_harm_frames for a (voice, tile) pair of a tile-classified launch with a corner of the envelope inside: the envelope of
// frame i is one line in front of frame ci (tile-relative) and another from there on, applied to the sample before the (constant)
// bus gains.
template <int FPL, typename Theta>
__device__ __forceinline__ void lean_tile_frames(double s0, double c0, double s1, double c1, double k2, bool straddle, Theta theta,
                                                 TrigTab trig, const double (&poly)[16], double GL, double GR,
                                                 double ea0, double ea1, double eb0, double eb1, double ci, double dl,
                                                 double (&accl)[FPL], double (&accr)[FPL]) {
    static_assert(FPL % 2 == 0, "frames in pairs");
#pragma unroll
    for (int h = 0; h < FPL; h += 2) {
        double p0 = fma(poly[0], c0, poly[1]), p1 = fma(poly[0], c1, poly[1]);
#pragma unroll
        for (int u = 2; u < 16; ++u) {
            p0 = fma(p0, c0, poly[u]);
            p1 = fma(p1, c1, poly[u]);
        }
        const double i0 = dl + (double)(h * 64), i1 = dl + (double)((h + 1) * 64);
        const double e0 = i0 < ci ? fma(i0, ea1, ea0) : fma(i0, eb1, eb0), e1 = i1 < ci ? fma(i1, ea1, ea0) : fma(i1, eb1, eb0);
        const double x0 = (p0 * s0) * e0, x1 = (p1 * s1) * e1;
        accl[h] = fma(GL, x0, accl[h]);
        accr[h] = fma(GR, x0, accr[h]);
        accl[h + 1] = fma(GL, x1, accl[h + 1]);
        accr[h + 1] = fma(GR, x1, accr[h + 1]);
        if (h + 2 < FPL) {
            if (straddle) {
                shm::sincos_tab(theta(h + 2), trig, s0, c0);
                shm::sincos_tab(theta(h + 3), trig, s1, c1);
            } else {
                const double s2 = fma(k2, s1, -s0), c2 = fma(k2, c1, -c0);
                const double s3 = fma(k2, s2, -s1), c3 = fma(k2, c2, -c1);
                s0 = s2; c0 = c2; s1 = s3; c1 = c3;
            }
        }
    }
}

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
                                                                  const uint32_t* __restrict__ prev_gen_valid,
                                                                  uint32_t prep_wgs) {
    // prep_wgs > 0 (lean / direct / combined modes only): prep_wgs workgroups in rows of their own BEHIND the voice groups' rows do not render.
    // Sequential streaming is the common call pattern: a launch also resolves the launch records of the block expected two
    // launches on (next_start = start + 2 * nframes: the launch in between runs beside this one on the other stream and got
    // its records from this one's predecessor) into a free record set, so that launch needs no prepare kernel of its own
    // (a 13 us kernel + a launch boundary per block otherwise).  One wavefront per chunk of 64 voices, in workgroups of their own:
    // beside the rendering workgroups, not in front of one (as the first step of the first tiles' workgroups -- round 2 -- the
    // step was the whole launch's critical path for small banks: 8 of 13 us) -- and dispatched LAST: a workgroup that leaves at
    // once between the rendering ones upsets their placement (the dispatcher does not refill the slot evenly: some CUs end up
    // with four rendering workgroups where others hold two, and the launch lasts as long as its fullest CU -- 63 against 53 us
    // for a bank of 352 chunks, most of them silent).
    if constexpr (!mode_general(MODE)) SH_STAMP(0);
    const uint32_t prep_rows = (!mode_general(MODE) && prep_wgs) ? (prep_wgs + gridDim.x - 1) / gridDim.x : 0u;
    const uint32_t ngroups = gridDim.y - prep_rows;              // the voice groups of the launch
    bool is_gen_wg = false;                   // RENDER_TILES_MERGED: this workgroup renders general pairs (a row behind the voice groups')
    uint32_t gen_unit = 0;
    if (!mode_general(MODE) && prep_wgs) {
        if (blockIdx.y >= ngroups) {
            const uint32_t unit = (blockIdx.y - ngroups) * gridDim.x + blockIdx.x;
            if constexpr (mode_tiles(MODE)) {
                // the rows behind the voice groups': (merged kernel) GEN_SPLIT general workgroups per tile; then the workgroups that
                // resolve the tile set of the block two launches on
                const uint32_t gen_total = mode_merged(MODE) ? gridDim.x * GEN_SPLIT : 0u;
                if (unit >= prep_wgs) return;
                if (unit >= gen_total) {
                    const uint32_t runs = (B.next_ntiles + TILES_PER_WAVE - 1) / TILES_PER_WAVE, wgs_per_chunk = (runs + 3) / 4;
                    const uint32_t u = unit - gen_total;
                    const uint32_t c = B.next_tiles.k0 * B.next_tiles.groups + u / wgs_per_chunk, run = (u % wgs_per_chunk) * 4 + (threadIdx.x >> 6);
                    if (B.next_tile_wgs && c < (nvoices + 63) / 64) {
                        __builtin_amdgcn_s_setprio(3);                       // (latency-bound, beside wavefronts that fill every issue slot)
                        prepare_tiles_wave(B, B.next_tiles, nvoices, next_start, nframes, B.next_ntiles, c, run, next.launch ? &next : nullptr);
                    }
                    return;
                }
                is_gen_wg = true;
                gen_unit = unit;
            } else
            if (next.launch && unit < prep_wgs && threadIdx.x < 64) {       // (ONE wavefront per workgroup works: spread over the CUs)
                __builtin_amdgcn_s_setprio(3);                               // (latency-bound, beside wavefronts that fill every issue slot)
                // (a tile-classified launch: the chunks of the next set's range only)
                const uint32_t c = (mode_tiles(MODE) ? B.next_tiles.k0 * B.next_tiles.groups : 0u) + unit;
                if (c < (nvoices + 63) / 64) {
                    if constexpr (mode_tiles(MODE)) {
                        // a tile-classified launch reads the records of the voices in its masks only: a chunk whose voices are all
                        // silent in that block is not resolved at all (a table of notes is mostly such chunks: 352 chunks, ~30
                        // sounding, cost the launch 11 of 64 us) -- the set is marked sparse on the host
                        const uint64_t span_lo = as_const(B.chunk_span)[2 * c], span_hi = as_const(B.chunk_span)[2 * c + 1];
                        if (next_start + (uint64_t)nframes <= span_lo || next_start >= span_hi) {
                            if (threadIdx.x == 0) {
                                const uint32_t in_chunk = nvoices - c * 64 < 64u ? nvoices - c * 64 : 64u;
                                next.counts[4 * c] = 0; next.counts[4 * c + 1] = 0; next.counts[4 * c + 2] = in_chunk; next.counts[4 * c + 3] = 0;
                            }
                            return;
                        }
                    }
                    prepare_chunk(B, next, c, nvoices, next_start, nframes);
                }
            }
            if (!mode_merged(MODE)) return;
        }
    }
    // RENDER_GENERAL_TILES with prep_wgs > 0 (SYNTHHIP_PREP_IN_GENERAL=1: where the step lived for most of round 3): the last
    // B.next_tile_wgs workgroups of the grid resolve the TILE SET of the block two launches on: four wavefronts each, a run of
    // TILES_PER_WAVE tiles of one chunk per wavefront (prepare_tiles_wave).  The step has been in four places: between the lean
    // kernel's workgroups (it cost the launch 16 of 69 us: workgroups that leave at once upset the placement of the others), a
    // kernel of its own on a third stream (which shared a hardware queue with a render stream), this kernel (68 + 25 us per launch
    // on its stream), and -- the default -- rows of the lean kernel's grid BEHIND its voice groups (80 + 13 us: see above).
    if constexpr (MODE == RENDER_GENERAL_TILES) {
        // This kernel is a few hundred latency-bound wavefronts that run beside the other stream's lean kernel, whose wavefronts
        // fill every issue slot they are given: without priority the two do not overlap at all -- the lean kernel runs at the speed
        // it has alone and this one takes 60 us instead of 23 (rocprofv3 kernel trace of a stream of blocks).
        __builtin_amdgcn_s_setprio(3);
        // (a one-dimensional grid: the general workgroups first -- theirs is the longer job and the other stream's lean kernel
        // leaves this one few slots -- then next_tile_wgs prepare workgroups over the chunks of the next set's range)
        // (the launch records of that block: resolved by the same wavefronts, for the voices that need one -- prepare_tiles_wave)
        if (prep_wgs && blockIdx.x >= gridDim.x - B.next_tile_wgs) {      // (prep_wgs = 0: the lean kernel of this launch resolves that set)
            const uint32_t runs = (B.next_ntiles + TILES_PER_WAVE - 1) / TILES_PER_WAVE, wgs_per_chunk = (runs + 3) / 4;
            const uint32_t unit = blockIdx.x - (gridDim.x - B.next_tile_wgs);
            const uint32_t c = B.next_tiles.k0 * B.next_tiles.groups + unit / wgs_per_chunk, run = (unit % wgs_per_chunk) * 4 + (threadIdx.x >> 6);
            if (c < (nvoices + 63) / 64) prepare_tiles_wave(B, B.next_tiles, nvoices, next_start, nframes, B.next_ntiles, c, run, next.launch ? &next : nullptr);
            return;
        }
    }
    // ... and workgroup (x, y) behind them renders the general pairs of ONE tile of 64 FPL frames, ALL voice groups' -- part
    // gen_part of GEN_SPLIT of them -- into plane gen_part of the general parts.
    uint32_t gen_part = 0;
    uint32_t bx_ = blockIdx.x;
    if constexpr (MODE == RENDER_GENERAL_TILES || mode_merged(MODE))
    if (MODE == RENDER_GENERAL_TILES || is_gen_wg) {
        if (mode_merged(MODE)) bx_ = gen_unit;
        gen_part = bx_ % GEN_SPLIT;
        bx_ /= GEN_SPLIT;
        if (bx_ * (64 * FPL) >= nframes) return;
        if (bx_ == 0 && gen_part == 0 && threadIdx.x < B.tiles.groups) gen_valid[threadIdx.x] = threadIdx.x < GEN_SPLIT ? 1u : 0u;   // GEN_SPLIT general planes
        {   // a tile without a general pair (nearly all of them, since the first tiles of a note are walk pairs of the lean kernel):
            // zeros into this workgroup's share of the plane, and out -- before the table, the barrier, the reduction
            const uint32_t kw = B.tiles.k1 - B.tiles.k0, nmask = B.tiles.groups * kw;
            const uint64_t* __restrict__ grow = B.tiles.gen + (size_t)((bx_ * (64 * FPL)) / TILE_FRAMES) * (B.tiles.groups * B.tiles.mask_k);
            bool any = false;
            for (uint32_t base = 0; base < nmask; base += 64) {
                const uint32_t mi = base + (threadIdx.x & 63);
                const uint64_t mine = mi < nmask ? grow[(mi / kw) * B.tiles.mask_k + B.tiles.k0 + mi % kw] : 0ull;
                any = any || __ballot(mine != 0ull) != 0ull;
            }
            if (!any) {
                for (uint32_t f = threadIdx.x; f < 64 * FPL; f += WAVES * 64) {
                    const uint32_t raw = bx_ * (64 * FPL) + f;
                    if (raw < nframes) parts[(size_t)(B.tiles.groups + gen_part) * nframes + raw] = make_double2(0.0, 0.0);
                }
                return;
            }
        }
    }
    const uint32_t bx = bx_;       // the tile (or, segmented: the tile counter) of this workgroup
    if constexpr (MODE == RENDER_GENERAL_ONLY) {       // (a segmented launch always writes its parts: see below)
        // a group without general voices in this launch: nothing to render, nothing to write (wave-uniform: scalar loads)
        const uint32_t c0g = (blockIdx.y * voices_per_group) / 64;
        uint32_t c1g = ((blockIdx.y + 1) * voices_per_group + 63) / 64;
        const uint32_t nch = (nvoices + 63) / 64;
        if (c1g > nch) c1g = nch;
        uint32_t total = 0;
        for (uint32_t c = c0g; c < c1g; ++c) total += as_const(cur.counts)[4 * c + 1];
        if (bx == 0 && threadIdx.x == 0) gen_valid[blockIdx.y] = total ? 1u : 0u;
        if (total == 0) return;
    }
    if constexpr (MODE == RENDER_GENERAL_SEG) {
        // most (tile, group) pairs of a segmented launch hold no general voice: their parts are zeros, written before any set-up
        // (the flags of a segmented launch say "every group's general parts are valid": its general voices are the first segment's)
        // grid.x: the first segment's tiles gen_sub times over (tile-major), then the other segments' tiles
        const uint32_t tiles_0 = (B.seg_first[1] - B.seg_first[0] + 64 * FPL - 1) / (64 * FPL);
        uint32_t sidx = 0, tidx = bx;
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
                    if (raw < n_s) parts[(size_t)(ngroups + grp) * nframes + B.seg_first[sidx] + raw] = make_double2(0.0, 0.0);
                }
                return;
            }
        }
    }
    // The previous launch of the stream (same shape) left its voice groups' partial buses unfolded: this launch's workgroups
    // fold it now, in group order, before their own work -- instead of a 5 us kernel between every two render launches.  With up
    // to sixteen groups the workgroups of group 0 fold their tile (whole 4 KB runs of every plane per load instruction); a short
    // launch of a table of notes -- 32 groups, a handful of tiles -- would leave 34 planes to eight workgroups: there the workgroups
    // of ALL the groups share the tile's frames, a slice each (A/B in one call: 1024 frames 15.7 -> 13.7 us, 4096 frames 20.0 ->
    // 18.8 us; but 16 384 frames -- 32 tiles -- 30.8 -> 36.3 us and the headline 37.5 -> 38.0 us: every workgroup then starts
    // with a round trip of loads, and a load instruction moves 64 lanes' worth instead of 256).
    const bool fold_shared = ngroups >= 32 && gridDim.x <= 12;         // (many groups, few tiles)
    if (!mode_general(MODE) && prev_parts && !is_gen_wg && (fold_shared || blockIdx.y == 0)) {
        const bool shared = fold_shared;
        const uint32_t slice = shared ? (64 * FPL + ngroups - 1) / ngroups : (uint32_t)(64 * FPL), f_lo = shared ? blockIdx.y * slice : 0u;
        const uint32_t f_hi = f_lo + slice < (uint32_t)(64 * FPL) ? f_lo + slice : (uint32_t)(64 * FPL);
        for (uint32_t f = f_lo + threadIdx.x; f < f_hi; f += WAVES * 64) {
            const uint32_t raw = bx * (64 * FPL) + f;
            if (raw >= nframes) continue;
            double2 acc = prev_parts[raw];
            uint32_t g = 1;
            // (eight loads in flight, added in group order: a short launch of a table of notes has 32 voice groups and few tiles --
            // a load per addition made the fold, 34 round trips, the longest thing in the launch: 25 us for 4096 frames)
            for (; g + 8 <= ngroups; g += 8) {
                double2 pp[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) pp[k] = prev_parts[(size_t)(g + k) * nframes + raw];
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    acc.x += pp[k].x;
                    acc.y += pp[k].y;
                }
            }
            for (; g < ngroups; ++g) {
                const double2 pp = prev_parts[(size_t)g * nframes + raw];
                acc.x += pp.x;
                acc.y += pp.y;
            }
            if (prev_gen_valid) {                              // the general kernel's parts of that launch, where it wrote any
                for (uint32_t g = 0; g < ngroups; ++g) {
                    if (as_const(prev_gen_valid)[g]) {
                        const double2 pp = prev_parts[(size_t)(ngroups + g) * nframes + raw];
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
    if constexpr (!mode_general(MODE)) SH_STAMP(1);
    // (SYNTHHIP_PREPARE_IN_TILE=1, for A/B timings: the round-2 placement of the prepare step -- the chunks of 64 voices spread
    // over the first tile workgroups, one wavefront each, in front of their own work)
    if (!mode_general(MODE) && next.launch && !prep_wgs) {
        const uint32_t nchunks = (nvoices + 63) / 64, nblocks = gridDim.x * ngroups;
        const uint32_t bid = blockIdx.y * gridDim.x + bx;
        if (threadIdx.x < 64) {
            for (uint32_t c = bid; c < nchunks; c += nblocks) prepare_chunk(B, next, c, nvoices, next_start, nframes);
        }
    }
    __shared__ double red[WAVES][2][64 * FPL];
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    for (uint32_t k = threadIdx.x; k < shm::TRIG_N; k += WAVES * 64) trig[k] = trig_g[k];
    __syncthreads();
    if constexpr (!mode_general(MODE)) SH_STAMP(2);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a segmented launch: this workgroup's segment and its tile there (uniform); everything below up to the stores is
    // relative to the segment.  Otherwise the launch is its own single segment.
    uint32_t seg_off = 0, nfr = nframes, tile_index = bx;
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
    if constexpr (mode_merged(MODE)) { if (is_gen_wg) build_frames(lane); }
    if constexpr (MODE == RENDER_GENERAL_TILES || mode_merged(MODE))
    if (MODE == RENDER_GENERAL_TILES || is_gen_wg) {
        // the (voice, tile) pairs of this tile that sound but are not lean, all voice groups': the masks of the classification tile
        // are one contiguous row -- a lane fetches one mask per pass, the non-zero ones are handed round by ballot and readlane --
        // and the pairs are dealt to the workgroups of the tile and their waves by their ordinal; each goes through the launch
        // record and the general code
        const uint32_t kw = B.tiles.k1 - B.tiles.k0, nmask = B.tiles.groups * kw;
        const uint64_t* __restrict__ grow = B.tiles.gen + (size_t)(tile0 / TILE_FRAMES) * (B.tiles.groups * B.tiles.mask_k);
        uint32_t ord = 0;
        for (uint32_t base = 0; base < nmask; base += 64) {
            const uint32_t mi = base + lane;
            const uint64_t mine = mi < nmask ? grow[(mi / kw) * B.tiles.mask_k + B.tiles.k0 + mi % kw] : 0ull;
            uint64_t have = __ballot(mine != 0ull);
            while (have) {
                const uint32_t src = (uint32_t)__builtin_ctzll(have);
                have &= have - 1;
                uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), (int)src) << 32) |
                             (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, (int)src);
                const uint32_t idx = base + src;                          // (group, k - k0)
                const uint32_t c = idx / kw + (B.tiles.k0 + idx % kw) * B.tiles.groups;
                while (m) {
                    const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    if ((ord++ % (uint32_t)(WAVES * GEN_SPLIT)) != wave * GEN_SPLIT + gen_part) continue;
                    const uint32_t vi = c * 64 + bit;
                    const VoiceRegs r = load_record(as_const(curS.launch) + vi);
                    general_voice<FPL>(r, curS.fm + vi, B, B.voices + vi, st0, tile0, nfr, i, di, trig, accl, accr);
                }
            }
        }
    }
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
    if constexpr (mode_tiles(MODE)) {
        if (!is_gen_wg) {
        // this tile's lean pairs: per chunk of the group a compacted list of 256-byte records (the count: the bits of the chunk's
        // mask), walked like the lean lists of an ordinary launch -- wave w takes every WAVES-th entry, the offset carries over
        static_assert(64 * FPL == TILE_FRAMES, "the lean kernel's tile is the tile of the classification");
        const size_t slots = (size_t)B.tiles.rec_chunks * 64;
        const TileRec SH_CONST_AS* trow = as_const(B.tiles.recs) + (size_t)tile_index * slots;
        // The voice groups of a tile-classified launch do not partition the CHUNKS but every chunk's list: entry p of a list goes
        // to group p / WAVES mod groups, wave p mod WAVES (the offset carries over from list to list) -- notes that sound together are
        // neighbours in the voice table, whole chunks of them, and any deal of whole chunks leaves one group with twice the work of
        // another.  The masks of the tile (one contiguous row) are fetched 64 at a time, one per lane; the lists that are not empty
        // are handed round by ballot and readlane.
        // (only the masks k0 .. k1 - 1 of every group: the chunks of the set's range -- see TileSet)
        const uint32_t kw = B.tiles.k1 - B.tiles.k0, nmask = ngroups * kw, stride = ngroups * WAVES;
        const uint64_t* __restrict__ lrow = B.tiles.lean + (size_t)tile_index * (ngroups * B.tiles.mask_k);
        uint32_t firstp = grp * WAVES + wave;
        for (uint32_t base = 0; base < nmask; base += 64) {
            const uint32_t mi = base + lane;
            const uint64_t mymask = mi < nmask ? lrow[(mi / kw) * B.tiles.mask_k + B.tiles.k0 + mi % kw] : 0ull;
            uint64_t have = __ballot(mymask != 0ull);
            while (have) {
            const uint32_t src = (uint32_t)__builtin_ctzll(have);
            have &= have - 1;
            const uint64_t cmask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mymask >> 32), (int)src) << 32) |
                                   (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mymask, (int)src);
            const uint32_t npairs = (uint32_t)__popcll(cmask);
            const uint32_t idx = base + src;                              // (group, k - k0) of the prepare step's layout
            const uint32_t c = idx / kw + (B.tiles.k0 + idx % kw) * ngroups;
            const TileRec SH_CONST_AS* q = trow + (c - B.tiles.k0 * ngroups) * 64 + firstp;
            uint32_t p = firstp;
            for (; p < npairs; p += stride, q += stride) {
                const double t0 = q->t0, dt = q->dt, rc = q->rc, rs = q->rs, ea0 = q->ea0, ea1 = q->ea1, GL = q->GL, GR = q->GR;
                const uint32_t pc = *reinterpret_cast<const uint32_t SH_CONST_AS*>(&q->npieces);      // npieces | corner << 16
                uint32_t lane_again = lane;                               // (converted per entry: two registers less across the loop)
                asm volatile("" : "+v"(lane_again));
                const double lane_d = (double)lane_again;
                // the voice of the list's entry p: the chunk's (p + 1)-th set bit -- its polynomial comes from the table by voice (read by
                // every tile's workgroups: it lives in L2), at an address that does not wait for the record
                const double SH_CONST_AS* pp = as_const(B.polys) + (size_t)(c * 64 + nth_set_bit(cmask, p)) * 16;
                double poly[16];
#pragma unroll
                for (int v_ = 0; v_ < 16; ++v_) poly[v_] = pp[v_];
                asm volatile("" :: "s"(t0), "s"(dt), "s"(rc), "s"(rs), "s"(ea0), "s"(ea1), "s"(GL), "s"(GR), "s"(pc),
                             "s"(poly[0]), "s"(poly[1]), "s"(poly[2]), "s"(poly[3]), "s"(poly[4]), "s"(poly[5]),
                             "s"(poly[6]), "s"(poly[7]), "s"(poly[8]), "s"(poly[9]), "s"(poly[10]), "s"(poly[11]), "s"(poly[12]),
                             "s"(poly[13]), "s"(poly[14]), "s"(poly[15]));
                const LaneTheta none{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0u, 0u, false};
                if constexpr (MODE == RENDER_LEAN_TILES_ALL || mode_merged(MODE)) {
                    // a plain Sawtooth / Square / Triangle / Pulse (unit amplitude, t in turns: the amplitude lives in the gains): every
                    // frame from its accumulated t on the piece that holds it -- the record's pieces, or a walk along the voice's
                    // table -- the envelope's line of the frame, and nothing in front of an onset
                    const uint32_t wkind = *reinterpret_cast<const uint32_t SH_CONST_AS*>(&q->pad_);
                    if (wkind != LEAN_HARM) {
                        const double eb0 = q->eb0, eb1 = q->eb1, ci = (double)(pc >> 16);
                        double th[FPL], on = 0.0;
                        if ((pc & 0xFFFFu) == 0u) {
                            const double dn0 = q->tb[0];
                            const uint64_t wbits = *reinterpret_cast<const uint64_t SH_CONST_AS*>(&q->tb[1]);
                            const uint32_t seg_first = (uint32_t)wbits, seg_end = (uint32_t)(wbits >> 32);
                            const uint32_t kidx = seg_first + (lane & (TILE_WALK_PIECES - 1));
                            const bool have_piece = lane < TILE_WALK_PIECES && kidx < seg_end;
                            const sh_segment* sp = B.segs + (have_piece ? kidx : seg_first);
                            const uint64_t pn0 = sp->n0;
                            const double pt0 = sp->t0, pdt = sp->dt;
                            const double prel = (double)(long long)pn0 - dn0;
                            const uint32_t npc = (uint32_t)__popcll(__ballot(have_piece && prel < (double)TILE_FRAMES));
                            on = -dn0;                                    // (<= 0: the voice started before the tile)
#pragma unroll
                            for (int j = 0; j < FPL; ++j) th[j] = 0.0;
                            for (uint32_t k = 0; k < npc; ++k) {
                                const double rk = readlane_f64(prel, k), tk = readlane_f64(pt0, k), dk = readlane_f64(pdt, k);
#pragma unroll
                                for (int j = 0; j < FPL; ++j) {
                                    const double x = lane_d + (double)(j * 64);
                                    th[j] = x >= rk ? fma(x - rk, dk, tk) : th[j];
                                }
                            }
                        } else {
                            const TileTheta theta{lane_d, t0, dt, q->tb[0], q->tb[1], q->db[0], q->db[1], lane, q->split[0], q->split[1]};
#pragma unroll
                            for (int j = 0; j < FPL; ++j) th[j] = theta(j);
                        }
                        if (wkind == LEAN_FM) {
                            // a Sine carrier with a closed-form Sine LFO: th[] is the accumulated TIME; the carrier's angle from the
                            // running sum of the LFO, L(n) = K (C0 - cos(a + (n - 1/2) d)) + bias n, at the voice's own index n
                            const double fr = poly[0], phase0 = poly[1], f_inc = poly[2], lfo_a = poly[3], lfo_d = poly[4];
                            const double lfo_K = poly[5], lfo_C0 = poly[6], lfo_bias = poly[7], n_first = rc;
                            const double a_rel = fma(n_first - 0.5, lfo_d, lfo_a);
#pragma unroll
                            for (int j = 0; j < FPL; ++j) {
                                const double x = lane_d + (double)(j * 64);
                                double ls, lc, sj, cj;
                                shm::sincos_tab(fma(x, lfo_d, a_rel), trig, ls, lc);
                                const double Ln = fma(lfo_K, lfo_C0 - lc, lfo_bias * (n_first + x));
                                shm::sincos_tab(fr * th[j] + fma(f_inc, Ln, phase0), trig, sj, cj);
                                const double ej = x < ci ? fma(x, ea1, ea0) : fma(x, eb1, eb0);
                                const double w = x >= on ? sj * ej : 0.0;
                                accl[j] = fma(GL, w, accl[j]);
                                accr[j] = fma(GR, w, accr[j]);
                            }
                            continue;
                        }
#pragma unroll
                        for (int j = 0; j < FPL; ++j) {
                            const double x = lane_d + (double)(j * 64);
                            double w = wkind == LEAN_SAW ? shm::saw_value(th[j], 2.0, 0.0)
                                     : wkind == LEAN_SQUARE ? shm::square_value(th[j], 1.0, 0.0)
                                     : wkind == LEAN_TRIANGLE ? shm::triangle_value(th[j], 4.0, 0.0)
                                     : shm::pulse_value(th[j], poly[0], 1.0, 0.0);
                            const double ej = x < ci ? fma(x, ea1, ea0) : fma(x, eb1, eb0);
                            w = x >= on ? w * ej : 0.0;
                            accl[j] = fma(GL, w, accl[j]);
                            accr[j] = fma(GR, w, accr[j]);
                        }
                        continue;
                    }
                }
                if (pc == 1u) {
                    // one piece, one line (seven pairs of eight): the lean arithmetic of an ordinary launch with the line folded into the gains
                    double s0, c0s, s1, c1s;
                    shm::sincos_tab(fma(lane_d, dt, t0), trig, s0, c0s);
                    s1 = fma(s0, rc, c0s * rs);
                    c1s = fma(c0s, rc, -(s0 * rs));
                    if (ea1 == 0.0)          // a flat line (the sustain: two thirds of a note's life): the headline's loop, two operations per frame less
                        lean_harm_frames<FPL>(s0, c0s, s1, c1s, rc + rc, false, none, trig, poly, GL * ea0, GR * ea0, accl, accr);
                    else
                        lean_harm_frames<FPL, true>(s0, c0s, s1, c1s, rc + rc, false, none, trig, poly, GL * ea0, GR * ea0, accl, accr, GL * ea1, GR * ea1, lane_d);
                } else if ((pc & 0xFFFFu) == 1u) {
                    // one piece, a corner: the envelope changes lines at frame pc >> 16
                    const double eb0 = q->eb0, eb1 = q->eb1;
                    double s0, c0s, s1, c1s;
                    shm::sincos_tab(fma(lane_d, dt, t0), trig, s0, c0s);
                    s1 = fma(s0, rc, c0s * rs);
                    c1s = fma(c0s, rc, -(s0 * rs));
                    lean_tile_frames<FPL>(s0, c0s, s1, c1s, rc + rc, false, none, trig, poly, GL, GR, ea0, ea1, eb0, eb1, (double)(pc >> 16), lane_d, accl, accr);
                } else if ((pc & 0xFFFFu) == 0u) {
                    // a WALK pair (see TileRec): lanes 0 .. 15 fetch a piece of the voice's table each -- one round trip -- and every
                    // frame takes the angle of the last piece that starts at or in front of it; a frame in front of them all (the
                    // onset lies inside the tile) keeps the angle 0, whose sine is 0
                    const double eb0 = q->eb0, eb1 = q->eb1, dn0 = q->tb[0];
                    const uint64_t wbits = *reinterpret_cast<const uint64_t SH_CONST_AS*>(&q->tb[1]);
                    const uint32_t seg_first = (uint32_t)wbits, seg_end = (uint32_t)(wbits >> 32);
                    const uint32_t kidx = seg_first + (lane & (TILE_WALK_PIECES - 1));
                    const bool have_piece = lane < TILE_WALK_PIECES && kidx < seg_end;
                    const sh_segment* sp = B.segs + (have_piece ? kidx : seg_first);
                    const uint64_t pn0 = sp->n0;
                    const double pt0 = sp->t0, pdt = sp->dt;
                    const double prel = (double)(long long)pn0 - dn0;             // the piece's first frame, relative to the tile's
                    const uint32_t npc = (uint32_t)__popcll(__ballot(have_piece && prel < (double)TILE_FRAMES));
                    double th[FPL];
#pragma unroll
                    for (int j = 0; j < FPL; ++j) th[j] = 0.0;
                    for (uint32_t k = 0; k < npc; ++k) {
                        const double rk = readlane_f64(prel, k), tk = readlane_f64(pt0, k), dk = readlane_f64(pdt, k);
#pragma unroll
                        for (int j = 0; j < FPL; ++j) {
                            const double x = lane_d + (double)(j * 64);
                            th[j] = x >= rk ? fma(x - rk, dk, tk) : th[j];
                        }
                    }
                    // (the polynomial once more, behind the walk: its sixteen coefficients would sit in scalar registers through a loop
                    // that needs those for the pieces -- and what the scalar file cannot hold costs vector registers this kernel lacks)
                    const uint64_t pp_bits = (uint64_t)pp;
                    const double SH_CONST_AS* pp2 = (const double SH_CONST_AS*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pp_bits >> 32)) << 32) |
                                                                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pp_bits));
                    double poly2[16];
#pragma unroll
                    for (int v_ = 0; v_ < 16; ++v_) poly2[v_] = pp2[v_];
#pragma unroll
                    for (int j = 0; j < FPL; ++j) {
                        const double x = lane_d + (double)(j * 64);
                        double sj, cj;
                        shm::sincos_tab(th[j], trig, sj, cj);
                        double pj = fma(poly2[0], cj, poly2[1]);
#pragma unroll
                        for (int u = 2; u < 16; ++u) pj = fma(pj, cj, poly2[u]);
                        const double ej = x < (double)(pc >> 16) ? fma(x, ea1, ea0) : fma(x, eb1, eb0);
                        const double xj = (pj * sj) * ej;
                        accl[j] = fma(GL, xj, accl[j]);
                        accr[j] = fma(GR, xj, accr[j]);
                    }
                } else {
                    // piece ends inside the tile: every frame by lookup from the piece that holds it
                    const double eb0 = q->eb0, eb1 = q->eb1;
                    const TileTheta theta{lane_d, t0, dt, q->tb[0], q->tb[1], q->db[0], q->db[1], lane, q->split[0], q->split[1]};
                    double s0, c0s, s1, c1s;
                    shm::sincos_tab(theta(0), trig, s0, c0s);
                    shm::sincos_tab(theta(1), trig, s1, c1s);
                    lean_tile_frames<FPL>(s0, c0s, s1, c1s, 0.0, true, theta, trig, poly, GL, GR, ea0, ea1, eb0, eb1, (double)(pc >> 16), lane_d, accl, accr);
                }
            }
            firstp = p - npairs;
            }
        }
        }   // (!is_gen_wg)
    } else
    if constexpr (!mode_general(MODE)) {
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t nfast = as_const(curS.counts)[4 * c];
        const FastRec SH_CONST_AS* q = as_const(curS.fast) + c * 64 + first;
        uint32_t p = first;
        for (; p < nfast; p += WAVES, q += WAVES) {
#ifdef SH_DIAG2
            const bool d2on = blockIdx.x == 1 && blockIdx.y == 1 && wave == 0 && c == c0 + 1;
            uint64_t d2t[5] = {0, 0, 0, 0, 0};
            if (d2on) d2t[0] = __builtin_amdgcn_s_memtime();
#endif
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
#ifdef SH_DIAG2
                { double th0 = theta(0); asm volatile("" : "+v"(th0)); if (d2on) d2t[1] = __builtin_amdgcn_s_memtime(); asm volatile("" : "+v"(th0)); }
#endif
                shm::sincos_tab(theta(0), trig, s0, c0);
#ifdef SH_DIAG2
                asm volatile("" : "+v"(s0), "+v"(c0)); if (d2on) d2t[2] = __builtin_amdgcn_s_memtime(); asm volatile("" : "+v"(s0), "+v"(c0));
#endif
                if (straddle) {
                    if (FPL > 1) shm::sincos_tab(theta(1), trig, s1, c1); else { s1 = s0; c1 = c0; }
                } else {
                    s1 = fma(s0, rc, c0 * rs);
                    c1 = fma(c0, rc, -(s0 * rs));
                }
#ifdef SH_DIAG2
                asm volatile("" : "+v"(s1), "+v"(c1)); if (d2on) d2t[3] = __builtin_amdgcn_s_memtime(); asm volatile("" : "+v"(s1), "+v"(c1));
#endif
                if constexpr (mode_seg(MODE)) {
                    const double gls = q->amplitude, grs = q->g0u;       // (a segmented launch's records: the gains' slopes per frame)
                    lean_harm_frames<FPL, true>(s0, c0, s1, c1, rc + rc, straddle, theta, trig, poly, gl, gr, accl, accr, gls, grs, di0);
                } else {
                    lean_harm_frames<FPL>(s0, c0, s1, c1, rc + rc, straddle, theta, trig, poly, gl, gr, accl, accr);
                }
#ifdef SH_DIAG2
                asm volatile("" : "+v"(accl[FPL - 1]), "+v"(accr[FPL - 1]), "+v"(accl[0]));
                if (d2on) {
                    d2t[4] = __builtin_amdgcn_s_memtime();
                    const uint32_t it = (p - first) / WAVES;
                    if (lane == 0 && it < 12) {
#pragma unroll
                        for (int k_ = 0; k_ < 5; ++k_) g_diag2[it * 8 + k_] = d2t[k_];
                    }
                }
#endif
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
                // an LFO without bias (the usual modulator; BASELINE config 3): the linear term of L is 0 * (start + i) = +0, and
                // fma(K, C0 - cos, +0) IS the rounded product -- three operations per frame less, the same bits.  (The two loops are
                // written out: behind a lambda the accumulators are no longer scalarised -- they went to LDS and scratch.)
#define SH_FM_FRAMES(LN_EXPR, THETA_EXPR)                                                                                \
                _Pragma("unroll")                                                                                        \
                for (int h = 0; h < FPL; h += 4) {            /* four carriers at a time: their table reads are in flight together */ \
                    constexpr int Q = FPL < 4 ? FPL : 4;                                                                 \
                    double th[Q], sn[Q], cs[Q];                                                                          \
                    _Pragma("unroll")                                                                                    \
                    for (int jj = 0; jj < Q; ++jj) {                                                                     \
                        const int j = h + jj;                                                                            \
                        const double Ln = LN_EXPR;                                                                       \
                        th[jj] = frequency * (THETA_EXPR) + fma(f_inc, Ln, phase0);                                      \
                        const double lc2 = fma(lk2, lc1, -lc0);                                                          \
                        lc0 = lc1;                                                                                       \
                        lc1 = lc2;                                                                                       \
                    }                                                                                                    \
                    shm::sincos_tab_n<Q>(th, trig, sn, cs);                                                              \
                    _Pragma("unroll")                                                                                    \
                    for (int jj = 0; jj < Q; ++jj) {                                                                     \
                        accl[h + jj] = fma(gl, sn[jj], accl[h + jj]);                                                    \
                        accr[h + jj] = fma(gr, sn[jj], accr[h + jj]);                                                    \
                    }                                                                                                    \
                }
                // (Measured and dropped: the same split by `straddle` -- theta(j) asks it per frame, a uniform branch per frame -- with
                // the one-piece tile's angle written out: straight-line code, four chains interleaved, and 58.8 instead of 51.6 us
                // per block of BASELINE config 3.)
                if (lfo_bias == 0.0) {
                    SH_FM_FRAMES(lfo_K * (lfo_C0 - lc0), theta(j))
                } else {
                    SH_FM_FRAMES(fma(lfo_K, lfo_C0 - lc0, lfo_bias * (startd + (di0 + (double)(j * 64)))), theta(j))
                }
#undef SH_FM_FRAMES
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
    if constexpr (MODE == RENDER_GENERAL_TILES) {
        // (rendered above)
    } else
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
    if constexpr (!mode_general(MODE)) SH_STAMP(3);
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        red[wave][0][j * 64 + lane] = accl[j];
        red[wave][1][j * 64 + lane] = accr[j];
    }
    __syncthreads();
    if constexpr (!mode_general(MODE)) SH_STAMP(4);
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
                const uint32_t slot = (MODE == RENDER_GENERAL_TILES || is_gen_wg) ? B.tiles.groups + gen_part : (mode_general(MODE) ? ngroups + grp : grp);
                parts[(size_t)slot * nframes + at] = make_double2(l, rr);
            } else {
                if (bus32) bus32[at] = make_float2((float)l, (float)rr);
                if (bus64) bus64[at] = make_double2(l, rr);
                if (pcm16) pcm16[at] = pcm16_frame(l, rr, pcm_scale);
            }
        }
    }
    if constexpr (!mode_general(MODE)) SH_STAMP(5);
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

}  // namespace

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
    const sh::Knobs& K;
    uint64_t start, next_start;
    uint32_t nframes, tiles, groups, vpg, nchunks, prep_wgs;
    int mode, var;
    bool split, with_general, use_aux;
    LaunchSet cur, next;
    float2* o32; double2* o64; uint32_t* o16;
    double2* parts;
    const double2* pv_parts; float2* pv32; double2* pv64; uint32_t* pv16;
    double pcm_scale, pv_scale;
    uint32_t* gen_valid; const uint32_t* pv_gen;
    uint32_t c_lo, c_hi;              // (tile-classified launches) the chunks that can sound in this block: [c_lo, c_hi)
};
#define SH_RL_UNPACK(L)                                                                                                             \
    sh_bank* b = (L).b; hipStream_t st = (L).st; const sh::Knobs& K = (L).K; const uint64_t start = (L).start, next_start = (L).next_start; \
    const uint32_t nframes = (L).nframes, tiles = (L).tiles, groups = (L).groups, vpg = (L).vpg, nchunks = (L).nchunks, prep_wgs = (L).prep_wgs; \
    const int mode = (L).mode, var = (L).var; const bool split = (L).split, with_general = (L).with_general, use_aux = (L).use_aux;    \
    const LaunchSet cur = (L).cur, next = (L).next; float2* o32 = (L).o32; double2* o64 = (L).o64; uint32_t* o16 = (L).o16;          \
    double2* parts = (L).parts; const double2* pv_parts = (L).pv_parts; float2* pv32 = (L).pv32; double2* pv64 = (L).pv64;            \
    uint32_t* pv16 = (L).pv16; const double pcm_scale = (L).pcm_scale, pv_scale = (L).pv_scale; uint32_t* gen_valid = (L).gen_valid;  \
    const uint32_t* pv_gen = (L).pv_gen; int rc = SH_OK;                                                                            \
    (void)b; (void)st; (void)K; (void)start; (void)next_start; (void)nframes; (void)tiles; (void)groups; (void)vpg; (void)nchunks;    \
    (void)prep_wgs; (void)mode; (void)var; (void)split; (void)with_general; (void)use_aux; (void)cur; (void)next; (void)o32; (void)o64; \
    (void)o16; (void)parts; (void)pv_parts; (void)pv32; (void)pv64; (void)pv16; (void)pcm_scale; (void)pv_scale; (void)gen_valid;    \
    (void)pv_gen; (void)rc

// A TILE-CLASSIFIED launch: the lean tiles kernel and the general kernel behind it (which also resolves the tile set -- and the launch
// records that will be needed -- of the block two launches on), or the merged kernel for a short launch.
static int launch_tiled(const RenderLaunch& L, bool records_deferred) {
    SH_RL_UNPACK(L);
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
    // this stream -- resolved by workgroups of this launch's general kernel
    if (next.launch) {
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
    // (both prepare steps of the block two launches on -- tile set, and launch records where a voice needs one -- ride in the
    // general kernel)
    LaunchSet no_next = next;
    no_next.launch = nullptr;
    const bool merged = tiles <= 16 && !K.no_merged;                  // a short launch: ONE kernel (RENDER_TILES_MERGED)
    if (merged) {
        const uint32_t behind = tiles * GEN_SPLIT + P.next_tile_wgs;   // general workgroups, then the tile-set prepare workgroups
        hipLaunchKernelGGL((k_bank_render<4, 8, 4, RENDER_TILES_MERGED>), dim3(tiles, groups + sh::div_up(behind, tiles)), dim3(256), 0, st, P,
                           trig_table(), b->nvoices, vpg, cur, next, next_start, start, nframes, o32, o64, parts, pv_parts, pv32, pv64,
                           o16, pcm_scale, pv16, pv_scale, gen_valid, pv_gen, behind);
    } else {
        // The tile set of the block two launches on is resolved in rows behind the lean kernel's voice groups (as in the merged
        // kernel); the general kernel behind it is then the general pairs' alone.  With the classification in the general kernel
        // (SYNTHHIP_PREP_IN_GENERAL=1: the shape this path had for most of round 3) the two kernels took 68 + 25 us on their stream, so
        // 80 + 13: the sum is what the chip can do, but a block comes out at 48.4 instead of 50.4 us.
        const uint32_t in_lean = K.prep_in_general ? 0u : P.next_tile_wgs;
        const LaunchSet& nx = in_lean ? next : no_next;
        if (b->tile_waveforms)
            hipLaunchKernelGGL((k_bank_render<4, 8, 4, RENDER_LEAN_TILES_ALL>), dim3(tiles, groups + sh::div_up(in_lean, tiles)), dim3(256), 0, st, P,
                               trig_table(), b->nvoices, vpg, cur, nx, next_start, start, nframes, o32, o64, parts, pv_parts, pv32, pv64,
                               o16, pcm_scale, pv16, pv_scale, gen_valid, pv_gen, in_lean);
        else
            hipLaunchKernelGGL((k_bank_render<4, 8, 4, RENDER_LEAN_TILES>), dim3(tiles, groups + sh::div_up(in_lean, tiles)), dim3(256), 0, st, P,
                               trig_table(), b->nvoices, vpg, cur, nx, next_start, start, nframes, o32, o64, parts, pv_parts, pv32, pv64,
                               o16, pcm_scale, pv16, pv_scale, gen_valid, pv_gen, in_lean);
    }
    SH_CHECK_LAUNCH("k_bank_render(lean, tiles)");
    if (!merged) {   // the general pairs: GEN_SPLIT workgroups per 256-frame tile, each all voice groups' pairs of it
        const uint32_t gen_wgs = sh::div_up(nframes, 256) * GEN_SPLIT;
        // (on a stream of its own beside the lean kernel it was slower, 95 against 75 us per block: five streams share four
        // hardware queues, and a kernel that waits for an event holds up whatever shares its queue)
        const uint32_t g_prep = K.prep_in_general ? P.next_tile_wgs : 0u;
        hipLaunchKernelGGL((k_bank_render<4, 4, 4, RENDER_GENERAL_TILES>), dim3(gen_wgs + g_prep), dim3(256), 0, st, P,
                           trig_table(), b->nvoices, vpg, cur, next, next_start, start, nframes, (float2*)nullptr, (double2*)nullptr, parts,
                           (const double2*)nullptr, (float2*)nullptr, (double2*)nullptr, (uint32_t*)nullptr, 0.0, (uint32_t*)nullptr, 0.0,
                           gen_valid, (const uint32_t*)nullptr, g_prep);
        SH_CHECK_LAUNCH("k_bank_render(general, tiles)");
    }
    b->tile_count += 1;
    return SH_OK;
}

// A transition launch cut into SEGMENTS (RENDER_*_SEG): one batched prepare, the lean kernel over all segments, the general kernel
// (the first segment's groups split SUB ways) and the combine of its slices.
static int launch_segmented(const RenderLaunch& L, uint32_t nseg, const uint32_t* seg_first) {
    SH_RL_UNPACK(L);
    sh::counters().segmented_launches += 1;
    const int ks = use_aux ? 1 : 0;
    LaunchSet& g = b->seg_set[ks];
    rc = grow_segment_sets(b->seg_block[ks], g, b->seg_cap[ks], nseg, b->nvoices);
    if (rc) return rc;
    BankPtrs P = ptrs(b);
    P.nseg = nseg;
    for (uint32_t k = 0; k <= nseg; ++k) P.seg_first[k] = seg_first[k];
    rc = launch_prepare_segments_var(st, true, P, g, b->nvoices, nseg, start);
    if (rc) return rc;
    uint32_t tiles_lean = 0, tiles_gen = 0;
    for (uint32_t k = 0; k < nseg; ++k) {
        tiles_lean += sh::div_up(seg_first[k + 1] - seg_first[k], 64 * 8);
        tiles_gen += sh::div_up(seg_first[k + 1] - seg_first[k], 64 * 4);
    }
    if (mode == RENDER_LEAN_HARM)
        hipLaunchKernelGGL((k_bank_render<4, 8, 4, RENDER_LEAN_HARM_SEG>), dim3(tiles_lean, groups + sh::div_up(prep_wgs, tiles_lean)), dim3(256), 0, st, P,
                           trig_table(), b->nvoices, vpg, g, next, next_start, start, nframes, o32, o64, parts, pv_parts, pv32, pv64,
                           o16, pcm_scale, pv16, pv_scale, gen_valid, pv_gen, prep_wgs);
    else
        hipLaunchKernelGGL((k_bank_render<4, 8, 4, RENDER_LEAN_ALL_SEG>), dim3(tiles_lean, groups + sh::div_up(prep_wgs, tiles_lean)), dim3(256), 0, st, P,
                           trig_table(), b->nvoices, vpg, g, next, next_start, start, nframes, o32, o64, parts, pv_parts, pv32, pv64,
                           o16, pcm_scale, pv16, pv_scale, gen_valid, pv_gen, prep_wgs);
    SH_CHECK_LAUNCH("k_bank_render(lean, segments)");
    SH_HIP(hipMemsetAsync(gen_valid, 1, (size_t)groups * sizeof(uint32_t), st));         // every group's general parts are written
    // the first segment's groups are split SUB ways (BankPtrs::gen_sub): with sixteen waves per workgroup a wave walks two
    // or three of the 128 voices of its group
    const uint32_t SUB = (uint32_t)K.gen_sub;
    const uint32_t n0 = seg_first[1];
    rc = sh::grow_pooled(b->seg_scratch[ks], (size_t)groups * SUB * n0 * sizeof(double2));
    if (rc) return rc;
    P.gen_sub = SUB;
    P.gen_scratch = (double2*)b->seg_scratch[ks].ptr;
    LaunchSet none = g;
    none.launch = nullptr;
    hipLaunchKernelGGL((k_bank_render<16, 4, 1, RENDER_GENERAL_SEG>), dim3(tiles_gen + (SUB - 1) * sh::div_up(n0, 64 * 4), groups), dim3(1024), 0, st, P,
                       trig_table(), b->nvoices, vpg, g, none, next_start, start, nframes, (float2*)nullptr, (double2*)nullptr, parts,
                       (const double2*)nullptr, (float2*)nullptr, (double2*)nullptr, (uint32_t*)nullptr, 0.0, (uint32_t*)nullptr, 0.0,
                       gen_valid, (const uint32_t*)nullptr, 0u);
    SH_CHECK_LAUNCH("k_bank_render(general, segments)");
    hipLaunchKernelGGL(k_seg_combine, dim3(sh::div_up(n0, 256), groups), dim3(256), 0, st, (const double2*)b->seg_scratch[ks].ptr, SUB, n0,
                       parts + (size_t)groups * nframes, nframes);
    SH_CHECK_LAUNCH("k_seg_combine");
    return SH_OK;
}

// A plain launch: the render kernel of the bank's shape (lean + general lists in one kernel, or the split pair).
static int launch_plain(const RenderLaunch& L) {
    SH_RL_UNPACK(L);
#define SH_LAUNCH_MODE(W_, F_, M_, MODE_)                                                                         \
hipLaunchKernelGGL((k_bank_render<W_, F_, M_, MODE_>), dim3(tiles, groups + sh::div_up(prep_wgs, tiles)), dim3(W_ * 64), (size_t)K.lds_pad, st, ptrs(b),    \
                   trig_table(), b->nvoices, vpg, cur, next, next_start, start, nframes, o32, o64, parts, pv_parts, pv32, pv64, \
                   o16, pcm_scale, pv16, pv_scale, gen_valid, pv_gen, prep_wgs)
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
case 1611: SH_LAUNCH_RENDER(16, 1, 1); break;
case 826: SH_LAUNCH_RENDER(8, 2, 6); break;
case 821: SH_LAUNCH_RENDER(8, 2, 1); break;
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
                       gen_valid, (const uint32_t*)nullptr, 0u);
    SH_CHECK_LAUNCH("k_bank_render(general lists)");
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
    const int mode = b->lean_candidates == 0 ? RENDER_DIRECT : (b->lean_fm_candidates ? RENDER_LEAN_ALL : RENDER_LEAN_HARM);
    // mostly-lean banks take eight frames per lane on long blocks: the recurrences make every frame after the second cost one or
    // two FMAs of trigonometry (one table lookup per eight frames)
    if (var == 0) var = b->nvoices >= 128 ? (2 * b->lean_candidates >= b->nvoices ? (nframes >= 16384 ? 484 : 444) : 844)
                                          : (b->nvoices >= 64 ? 821 : (b->nvoices >= 8 ? 421 : 211));
    // A bank whose notes do not move in lock-step takes the tile-classified launch (below) at EVERY block length: in real-time
    // chunks (256 .. 4096 frames) the general code walked the whole table for every tile -- a table of 22 528 notes, ~780 of them
    // sounding: 60 .. 85 us per chunk where the arithmetic is 2 us.  The lean tiles kernel has one shape (four waves, 512-frame
    // tiles); a short launch is a few tiles of it, with more voice groups (but not hundreds: their partial buses are folded
    // frame by frame).
    const bool tile_candidate = K.variant == 0 && b->tile_all && b->nvoices >= 128 && mode != RENDER_DIRECT && !K.no_tiles && !b->needs_rows && b->first_row_voice < 0 &&
                                (b->has_onsets || b->own_envelopes || K.tiles_for_all == 1) && !b->no_general_voice(start, nframes);
    if (tile_candidate) var = 484;
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
    if (tile_candidate && groups > 32) groups = 32;
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
    // (see the RENDER_* modes); SYNTHHIP_NO_SPLIT=1 keeps the combined kernel
    const bool split = mode != RENDER_DIRECT && groups > 1 && !K.no_split;
    // A transition launch (RENDER_*_SEG): cut where the envelopes become flat (every voice past its decay) and from there on
    // into segments no longer than their own distance from the note's start -- piece ends of the phase sum lie an octave apart,
    // so such a segment crosses at most one per voice; a cut, too, where the first voice leaves its sustain.  Tile-aligned cuts.
    // A TILE-CLASSIFIED launch (RENDER_*_TILES): banks whose notes do not move in lock-step -- onsets, envelope corners of their
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
    if (!tiled && split && (mode == RENDER_LEAN_HARM || mode == RENDER_LEAN_ALL) && var == 484 && b->all_lean && !K.no_seg && !b->needs_rows && !b->no_general_voice(start, nframes)) {
        nseg = plan_segments(b, start, nframes, (uint64_t)(64 * F), ~0ull, true, seg_first);
        if (nseg < 2 || seg_first[nseg] != nframes) nseg = 0;        // nothing to cut, or more cuts than a launch carries
    }
    bool records_deferred = false;          // (a tile-classified launch without a resolved record set: its classification resolves what it needs)
    if (nseg == 0) {
        rc = acquire_records(b, start, nframes, st, cont, tiled, tiled ? &records_deferred : nullptr);
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
    const bool with_general = split && (K.always_general || !b->no_general_voice(start, nframes));
    uint32_t* gen_valid = with_general ? (uint32_t*)(parts + 2 * (size_t)groups * nframes) : nullptr;
    // the fold this launch takes over: the older of two outstanding ones (launch n - 2's)
    const bool take_over = b->npending == 2;
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
    const uint32_t prep_wgs = (next.launch && !K.prepare_in_tile) ? nchunks : 0u;
    const RenderLaunch L{b, st, K, start, next_start, nframes, tiles, groups, vpg, nchunks, prep_wgs, mode, var, split, with_general, use_aux,
                         cur, next, o32, o64, o16, parts, pv_parts, pv32, pv64, pv16, pcm_scale, pv_scale, gen_valid, pv_gen, tile_c_lo, tile_c_hi};
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
    }
    if (S.open_runs > 0) {
        // on record while any run is open (this bank's or another's): direct writes with their stream; the buses a pending fold
        // will write from now (another bank must not slip a write in between), the stream follows when a launch takes the fold over
        note_write(b, o32, n32, direct ? stream_bit : 0);
        note_write(b, o64, n64, direct ? stream_bit : 0);
        note_write(b, o16, n16, direct ? stream_bit : 0);
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
