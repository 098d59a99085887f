// osc_render_kernels.hpp -- the render kernels of a voice bank: fused generate-and-mix into the stereo bus.  Included by
// osc_render.hip only (the kernels live in the translation unit that launches them: no relocatable device code).
//
// Every render kernel has the same skeleton -- a workgroup of WAVES wavefronts owns ONE tile of 64*FPL frames and ONE voice group;
// wave w walks its share of the group's voices with the voice record in SGPRs; float64 partial (L, R) per lane; LDS-staged sum
// across the waves; one store per frame -- and the steps are __device__ inlines shared by four __global__ functions, so that no kernel
// carries code or parameters it never uses (round 3 had ONE 750-line kernel with thirteen MODE values):
//
//   k_render_lean      the lean lists of a SPLIT launch (several voice groups): polynomial Harmonics by lookup + rotation + three-term
//                      recurrence + Horner (ALL: also FM Sine and the plain waveforms); SEG: a transition launch cut into segments
//                      with a record set each.  Compiles without the general code and so without its registers and scratch.
//   k_render_general   what the lean kernel leaves: the general LISTS of a split launch, of a SEGMENTED launch (first segment split
//                      gen_sub ways), or the general PAIRS of a tile-classified launch -- all through voice_block, four frames per lane.
//   k_render_tiles     the lean (voice, tile) pairs of a TILE-CLASSIFIED launch (banks whose notes do not move in lock-step); MERGED: a
//                      short launch as one kernel -- the general pairs in rows of the grid behind the voice groups'.
//   k_render_combined  lean and general lists (or the voice table directly) in one kernel: single-group launches (small banks write
//                      the caller's bus themselves), banks without lean candidates, SYNTHHIP_NO_SPLIT.
//
// Beside its own work a lean / tiles / combined launch (a) FOLDS the partial buses its stream's previous launch left (fold_previous)
// and (b) RESOLVES, in workgroups of their own in rows of the grid BEHIND the voice groups' rows, the launch records (tile set) of the
// block expected two launches on (prep_rows_*): a steady stream is ONE kernel per block.
#pragma once
#include "osc_device.hpp"
#include <type_traits>

// The launch structures are shared by the two translation units that instantiate these kernels (osc_render.hip: combined / general /
// tiles / the folds; osc_render_lean.hip: every k_render_lean -- compiled side by side, the lean kernels are half of the ISA), so they
// have names with linkage; everything else in this header stays local to the unit that includes it.
namespace shosc {
// A final bus: float32 and / or float64 frames, and / or saturated int16 PCM (any may be NULL).
struct BusOut {
    float2*   bus32;
    double2*  bus64;
    uint32_t* pcm16;
    double    pcm_scale;
};
// The partial buses the previous launch of this stream left unfolded (parts == NULL: none) and where their sum goes.
struct FoldIn {
    const double2*  parts;
    const uint32_t* gen_valid;      // which groups' general parts (behind the lean ones) hold anything; NULL: no general parts
    BusOut          out;
    // non-NULL: a SELF-FOLDING launch (a render that stands alone: nothing to take over, nobody behind it to leave the fold to) --
    // `parts` are this launch's OWN partial buses, `out` its own buses, and self[tile] counts the voice groups' workgroups that have
    // stored their plane of the tile: the last one folds the tile, in group order like k_bus_combine (bit-identical sums)
    uint32_t*       self;
};
// What every render kernel needs of the launch.
struct LaunchArgs {
    BankPtrs              B;
    const shm::sc_pair*   trig_g;
    uint32_t              nvoices, voices_per_group;
    LaunchSet             cur;
    uint64_t              start;
    uint32_t              nframes;
    // 1: nothing resolved this launch's records (a render that stands alone, the first of a run): every workgroup resolves the
    // records of its OWN voice group's chunks first -- the same values from every tile's workgroup of the group, three dependent round
    // trips at the head of the launch instead of a prepare kernel (9-13 us) and a launch boundary in front of it (k_render_lean)
    uint32_t              self_prepare;
    // 1: no other render launch runs beside this one (it does not continue a run): its wavefronts lower their priority as they get on
    // with their lists (lean_lists), so the wavefronts of a SIMD end together
    uint32_t              alone;
};
// The block expected two launches on (next_start = start + 2 * nframes: the launch in between runs beside this one on the other stream
// and got its records from this one's predecessor): its record set, and how many workgroups of this grid resolve it (0: nobody).
struct NextArgs {
    LaunchSet next;
    uint64_t  next_start;
    uint32_t  prep_wgs;
};
enum { LEAN_K_HARM = 0, LEAN_K_ALL = 1, LEAN_K_FM = 2, LEAN_K_REST = 3 };     // which kinds of lean record a bank can hold (static); _REST: a run of a list
// osc_render_lean.hip: k_render_lean<W, F, M, kinds, seg> of shape var = W F M (4163, 484, 444, 844, 821) on `st`; SH_ERR_INVALID for a
// shape that has no lean kernel (421, 211: banks of fewer than 64 voices never split a launch)
int launch_render_lean(int var, int kinds, bool seg, dim3 grid, hipStream_t st, const LaunchArgs& A, const NextArgs& N, const FoldIn& F, double2* parts);
// osc_render_combined.hip: k_render_combined<W, F, M, mode> of shape var (484, 444, 844, 821, 421, 211), mode = COMBINED_DIRECT / _LEAN_HARM / _LEAN_ALL
int launch_render_combined(int var, int mode, dim3 grid, hipStream_t st, const LaunchArgs& A, const NextArgs& N, const FoldIn& F, double2* parts, const BusOut& out);
}  // namespace shosc

namespace {

constexpr uint32_t GEN_SPLIT = 2;      // general workgroups per tile of a tile-classified launch (each writes a plane of general parts; <= groups)

// -DSH_DIAG (tools/ab.py build; never the shipped library): every wavefront of a render launch leaves the 100 MHz timestamps of
// its phases and the SIMD it ran on in g_diag (four banks by block number: launches of a stream of blocks overlap pairwise);
// sh_debug_diag copies them out.  tools/headline_phases.py turns them into profiles/r04_headline_phases.md.
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
#define SH_STAMP(A_, slot) diag_stamp((A_).start, (A_).nframes, slot)
#else
#define SH_STAMP(A_, slot) ((void)0)
#endif

// ---- the tile a workgroup owns -------------------------------------------------------------------------------------------------
// A segmented launch: the workgroup's segment and its tile there; everything up to the stores is relative to the segment (records,
// tile, clamps), only the stores are launch-relative again (seg_off).  Otherwise the launch is its own single segment.
struct TileCtx {
    uint32_t lane, wave;          // wave: uniform
    uint32_t grp, ngroups;        // the voice group of this workgroup, of the launch
    uint32_t tile_index, tile0, tile_last, nfr;     // segment-relative; nfr: frames of the segment
    uint32_t seg_off;             // the segment's first frame in the launch
    uint64_t st0;                 // ... and absolute
    uint32_t c0, c1;              // the group's chunks of 64 voices: [c0, c1)
    LaunchSet set;                // the record set (of the segment)
    uint32_t alone;               // LaunchArgs::alone
};

template <int FPL>
__device__ __forceinline__ void tile_bounds(TileCtx& T) {
    T.tile0 = T.tile_index * (64 * FPL);
    T.tile_last = T.tile0 + 64 * FPL - 1;
    if (T.tile_last > T.nfr - 1) T.tile_last = T.nfr - 1;
}

// this group's voices: chunks [c0, c1) of 64 voices (voices_per_group is a multiple of 64 unless there is one group)
__device__ __forceinline__ void group_chunks(TileCtx& T, const LaunchArgs& A) {
    T.c0 = (T.grp * A.voices_per_group) / 64;
    T.c1 = ((T.grp + 1) * A.voices_per_group + 63) / 64;
    const uint32_t nchunks = (A.nvoices + 63) / 64;
    if (T.c1 > nchunks) T.c1 = nchunks;
}

template <int FPL>
__device__ __forceinline__ TileCtx plain_tile(const LaunchArgs& A, uint32_t bx, uint32_t grp, uint32_t ngroups) {
    TileCtx T;
    T.lane = threadIdx.x & 63;
    T.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    T.grp = grp;
    T.ngroups = ngroups;
    T.tile_index = bx;
    T.nfr = A.nframes;
    T.seg_off = 0;
    T.st0 = A.start;
    T.set = A.cur;
    T.alone = A.alone;
    tile_bounds<FPL>(T);
    group_chunks(T, A);
    return T;
}

// the segment that holds tile counter T.tile_index (tiles of all segments run through grid.x): T becomes segment-relative
template <int FPL>
__device__ __forceinline__ void enter_segment(TileCtx& T, const LaunchArgs& A) {
    uint32_t sidx = 0;
    for (;;) {
        const uint32_t n_s = A.B.seg_first[sidx + 1] - A.B.seg_first[sidx];
        const uint32_t tiles_s = (n_s + 64 * FPL - 1) / (64 * FPL);
        if (T.tile_index < tiles_s || sidx + 1 >= A.B.nseg) break;
        T.tile_index -= tiles_s;
        ++sidx;
    }
    T.seg_off = A.B.seg_first[sidx];
    T.nfr = A.B.seg_first[sidx + 1] - T.seg_off;
    T.st0 = A.start + T.seg_off;
    T.set = segment_set(A.cur, sidx, A.nvoices);
    tile_bounds<FPL>(T);
}

// the lane's frames, launch- (segment-) relative, clamped into the launch: out-of-range lanes compute a valid sample and do not store it
template <int FPL>
__device__ __forceinline__ void build_frames(const TileCtx& T, uint32_t lane_, uint32_t (&i)[FPL], double (&di)[FPL]) {
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        const uint32_t raw = T.tile0 + j * 64 + lane_;
        i[j] = raw < T.nfr ? raw : T.nfr - 1;
        di[j] = (double)i[j];
    }
}

__device__ __forceinline__ void load_trig(shm::sc_pair* trig, const shm::sc_pair* __restrict__ trig_g, uint32_t nthreads) {
    for (uint32_t k = threadIdx.x; k < shm::TRIG_N; k += nthreads) trig[k] = trig_g[k];
    __syncthreads();
}

// ---- rows of the grid behind the voice groups': the records of the block two launches on ------------------------------------------
// Sequential streaming is the common call pattern, so a launch resolves the launch records of the block expected two launches on into
// a free record set, and that launch needs no prepare kernel of its own (a 13 us kernel + a launch boundary per block otherwise).
// One wavefront per chunk of 64 voices, in workgroups of their own BESIDE the rendering ones, not in front of one (as the first step
// of the first tiles' workgroups -- round 2 -- the step was the whole launch's critical path for small banks: 8 of 13 us) -- and
// dispatched LAST: a workgroup that leaves at once between the rendering ones upsets their placement (the dispatcher does not refill
// the slot evenly; 63 against 53 us for a bank of 352 chunks, most of them silent).
__device__ __forceinline__ uint32_t groups_of_grid(uint32_t prep_wgs) {
    return gridDim.y - (prep_wgs ? (prep_wgs + gridDim.x - 1) / gridDim.x : 0u);
}
// true: this workgroup sat in a row behind the voice groups' (and has done what there was to do): the kernel returns
__device__ __forceinline__ bool prep_rows_lists(const LaunchArgs& A, const NextArgs& N, uint32_t ngroups) {
    if (!N.prep_wgs || blockIdx.y < ngroups) return false;
    const uint32_t unit = (blockIdx.y - ngroups) * gridDim.x + blockIdx.x;
    if (N.next.launch && unit < N.prep_wgs && threadIdx.x < 64) {           // (ONE wavefront per workgroup works: spread over the CUs)
        __builtin_amdgcn_s_setprio(3);                                     // (latency-bound, beside wavefronts that fill every issue slot)
        if (unit < (A.nvoices + 63) / 64) prepare_chunk(A.B, N.next, unit, A.nvoices, N.next_start, A.nframes);
    }
    return true;
}
// The same for a tile-classified launch: (MERGED) GEN_SPLIT general workgroups per tile first, then the workgroups that resolve the
// TILE SET of the block two launches on -- four wavefronts each, a run of TILES_PER_WAVE tiles of one chunk per wavefront
// (prepare_tiles_wave, which also resolves the launch records of the voices that need one).  0: a lean workgroup; 1: done, return;
// 2: a general workgroup of the merged kernel (its unit in gen_unit).
template <bool MERGED>
__device__ __forceinline__ int prep_rows_tiles(const LaunchArgs& A, const NextArgs& N, uint32_t ngroups, uint32_t& gen_unit) {
    if (!N.prep_wgs || blockIdx.y < ngroups) return 0;
    const uint32_t unit = (blockIdx.y - ngroups) * gridDim.x + blockIdx.x;
    const uint32_t gen_total = MERGED ? gridDim.x * GEN_SPLIT : 0u;
    if (unit >= N.prep_wgs) return 1;
    if (unit >= gen_total) {
        const BankPtrs& B = A.B;
        const uint32_t runs = (B.next_ntiles + TILES_PER_WAVE - 1) / TILES_PER_WAVE, wgs_per_chunk = (runs + 3) / 4;
        const uint32_t u = unit - gen_total;
        const uint32_t c = B.next_tiles.k0 * B.next_tiles.groups + u / wgs_per_chunk, run = (u % wgs_per_chunk) * 4 + (threadIdx.x >> 6);
        if (B.next_tile_wgs && c < (A.nvoices + 63) / 64) {
            __builtin_amdgcn_s_setprio(3);
            prepare_tiles_wave(B, B.next_tiles, A.nvoices, N.next_start, A.nframes, B.next_ntiles, c, run, N.next.launch ? &N.next : nullptr);
        }
        return 1;
    }
    gen_unit = unit;
    return 2;
}

// ---- the fold of the previous launch's partial buses --------------------------------------------------------------------------
// The previous launch of the stream (same shape) left its voice groups' partial buses unfolded: this launch's workgroups fold them
// now, in group order, before their own work -- instead of a 5 us kernel between every two render launches.  With up to sixteen
// groups the workgroups of group 0 fold their tile (whole 4 KB runs of every plane per load instruction); a short launch of a table
// of notes -- 32 groups, a handful of tiles -- would leave 34 planes to eight workgroups: there the workgroups of ALL the groups share
// the tile's frames, a slice each (1024 frames 15.7 -> 13.7 us, 4096 frames 20.0 -> 18.8 us; but 32 tiles 30.8 -> 36.3 us and the
// headline 37.5 -> 38.0 us: every workgroup then starts with a round trip of loads).  Measured in round 4 (profiles/
// r04_headline_phases.md): 4.4 us for the 94 workgroups of group 0 of the headline launch, 0.5 us for everybody else -- not what
// bounds the launch.
template <int WAVES, int FPL>
__device__ __forceinline__ void fold_previous(const FoldIn& F, uint32_t ngroups, uint32_t nframes, uint32_t bx) {
    const bool shared = ngroups >= 32 && gridDim.x <= 12;         // (many groups, few tiles)
    if (!F.parts || F.self || !(shared || blockIdx.y == 0)) return;
    const double2* __restrict__ prev_parts = F.parts;
    const uint32_t slice = shared ? (64 * FPL + ngroups - 1) / ngroups : (uint32_t)(64 * FPL), f_lo = shared ? blockIdx.y * slice : 0u;
    const uint32_t f_hi = f_lo + slice < (uint32_t)(64 * FPL) ? f_lo + slice : (uint32_t)(64 * FPL);
    for (uint32_t f = f_lo + threadIdx.x; f < f_hi; f += WAVES * 64) {
        const uint32_t raw = bx * (64 * FPL) + f;
        if (raw >= nframes) continue;
        double2 acc = prev_parts[raw];
        uint32_t g = 1;
        // (eight loads in flight, added in group order: a load per addition made the fold, 34 round trips, the longest thing in a
        // short launch of a table of notes: 25 us for 4096 frames)
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
        if (F.gen_valid) {                                 // the general kernel's parts of that launch, where it wrote any
            for (uint32_t g2 = 0; g2 < ngroups; ++g2) {
                if (as_const(F.gen_valid)[g2]) {
                    const double2 pp = prev_parts[(size_t)(ngroups + g2) * nframes + raw];
                    acc.x += pp.x;
                    acc.y += pp.y;
                }
            }
        }
        if (F.out.bus32) F.out.bus32[raw] = make_float2((float)acc.x, (float)acc.y);
        if (F.out.bus64) F.out.bus64[raw] = acc;
        if (F.out.pcm16) F.out.pcm16[raw] = pcm16_frame(acc.x, acc.y, F.out.pcm_scale);
    }
}

// ---- a launch that stands alone ----------------------------------------------------------------------------------------------------
// SELF-FOLD (FoldIn::self): behind its store of the tile's plane a workgroup takes a ticket of the tile's counter; the one that draws
// the last ticket -- every group's plane of the tile is in memory then (release fence before the ticket, acquire fence behind it: the
// planes were written through other XCDs' L2s) -- folds the tile in group order, the sum k_bus_combine makes, and resets the counter.
// The idea: a render that nothing follows pays the fold in its own tail instead of a 5.6 us kernel and a launch boundary behind it.
// Measured (profiles/r06_run_lengths.txt): 63 us per lone block against 57 -- the planes must cross XCDs inside the kernel -- so the
// host never asks for it unless SYNTHHIP_SELF says so.
template <int WAVES, int FPL>
__device__ __forceinline__ void fold_own_tile(const FoldIn& F, uint32_t ngroups, uint32_t nframes, uint32_t bx, uint32_t* ticket_lds) {
    // (ticket_lds: a word of the staging buffer, free once the planes are stored -- a word of its own would be the 160 KB + 4 bytes that
    //  take the kernels of four workgroups per CU down to three)
    // The planes of a self-folding launch are stored and loaded at AGENT scope (sc1: through the XCD's L2 to memory and from there) --
    // a release / acquire fence pair at agent scope writes back and invalidates the whole L2 of the XCD, from every one of the launch's
    // 752 workgroups: 123 us per block instead of 56 (profiles/r06_run_lengths.txt).  The stores have been acknowledged when the
    // wavefront's counter is back at zero; then the ticket.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *ticket_lds = atomicAdd(F.self + bx, 1u);
    __syncthreads();
    const uint32_t ticket = *ticket_lds;
    if (ticket != ngroups - 1) return;
    if (threadIdx.x == 0) F.self[bx] = 0;                  // (the next self-folding launch of the bank starts from zero)
    const double* __restrict__ own = reinterpret_cast<const double*>(F.parts);
    auto plane = [&](uint32_t g, uint32_t raw) {
        const double* q = own + 2 * ((size_t)g * nframes + raw);
        return make_double2(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    for (uint32_t f = threadIdx.x; f < (uint32_t)(64 * FPL); f += WAVES * 64) {
        const uint32_t raw = bx * (64 * FPL) + f;
        if (raw >= nframes) continue;
        double2 acc = plane(0, raw);
        uint32_t g = 1;
        for (; g + 8 <= ngroups; g += 8) {
            double2 pp[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) pp[k] = plane(g + k, raw);
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                acc.x += pp[k].x;
                acc.y += pp[k].y;
            }
        }
        for (; g < ngroups; ++g) {
            const double2 pp = plane(g, raw);
            acc.x += pp.x;
            acc.y += pp.y;
        }
        if (F.out.bus32) F.out.bus32[raw] = make_float2((float)acc.x, (float)acc.y);
        if (F.out.bus64) F.out.bus64[raw] = acc;
        if (F.out.pcm16) F.out.pcm16[raw] = pcm16_frame(acc.x, acc.y, F.out.pcm_scale);
    }
}

// SELF-PREPARE (LaunchArgs::self_prepare): the records of this workgroup's voice group, resolved by its own wavefronts (a chunk of 64
// voices per wavefront) into the launch's record set -- every tile's workgroup of the group stores the same values there -- and read
// back through the scalar cache, which is invalidated first (it may hold what an earlier launch kept in this set).
template <int WAVES>
__device__ __forceinline__ void prepare_own_group(const LaunchArgs& A, uint32_t grp) {
    const uint32_t nchunks = (A.nvoices + 63) / 64;
    const uint32_t c0 = (grp * A.voices_per_group) / 64;
    uint32_t c1 = ((grp + 1) * A.voices_per_group + 63) / 64;
    if (c1 > nchunks) c1 = nchunks;
    for (uint32_t c = c0 + (threadIdx.x >> 6); c < c1; c += WAVES) prepare_chunk(A.B, A.cur, c, A.nvoices, A.start, A.nframes);
    // (the workgroup reads its OWN stores back: acknowledged -- in its XCD's L2 -- when the counter is at zero; no fence at agent scope,
    //  which would write back and invalidate that whole L2 from every workgroup of the launch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_dcache_inv();
    __syncthreads();
}

// ---- the sum across the waves of the workgroup, and the store ---------------------------------------------------------------------
// dst: a plane of partial buses (launch-relative frames) -- or NULL: the final bus `out` (single-group launches).  The staging buffer
// holds RED_ROWS(FPL) rows of 64 frames per wave and channel: eight at most (32 KB at four waves -- four workgroups per CU with the
// trig table); sixteen frames per lane go through it in two passes.
constexpr int red_rows(int fpl) { return fpl <= 8 ? fpl : (fpl % 8 == 0 ? 8 : (fpl % 6 == 0 ? 6 : 4)); }
template <int WAVES, int FPL>
__device__ __forceinline__ void reduce_store(double (*red)[2][64 * red_rows(FPL)], const TileCtx& T, const double (&accl)[FPL], const double (&accr)[FPL],
                                             double2* __restrict__ dst, size_t dst_off, const BusOut& out, bool agent_scope = false) {
    constexpr int RR = red_rows(FPL);
#pragma unroll
    for (int r0 = 0; r0 < FPL; r0 += RR) {
        if (r0) __syncthreads();                          // (the previous pass's rows have been read)
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            red[T.wave][0][j * 64 + T.lane] = accl[r0 + j];
            red[T.wave][1][j * 64 + T.lane] = accr[r0 + j];
        }
        __syncthreads();
        // each wave finishes 64-frame rows of the tile: rows wave, wave + WAVES, ...
        for (uint32_t row = T.wave; row < (uint32_t)RR; row += WAVES) {
            const uint32_t f = row * 64 + T.lane;
            const uint32_t raw = T.tile0 + (uint32_t)r0 * 64 + f;
            if (raw < T.nfr) {
                double l = red[0][0][f], rr = red[0][1][f];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) {
                    l += red[w][0][f];
                    rr += red[w][1][f];
                }
                const size_t at = dst_off + raw;
                if (dst && agent_scope) {                  // (a self-folding launch: another XCD's workgroup reads the plane in this kernel)
                    double* q = reinterpret_cast<double*>(dst + at);
                    __hip_atomic_store(q, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(q + 1, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (dst) {
                    dst[at] = make_double2(l, rr);
                } else {
                    if (out.bus32) out.bus32[at] = make_float2((float)l, (float)rr);
                    if (out.bus64) out.bus64[at] = make_double2(l, rr);
                    if (out.pcm16) out.pcm16[at] = pcm16_frame(l, rr, out.pcm_scale);
                }
            }
        }
    }
}

// The lean Harmonics arithmetic as VALUES: out[j] = P(cos) * sin of the lane's frame j (lean_harm_frames without the accumulation; same
// operations, same order).  The tiles kernel's five kinds of pair all end in ONE accumulation (tiles_lean): with an accumulation inside
// every kind the compiler shuffled the sixteen accumulators between the kinds' register assignments at every join -- ~34 register
// copies per pair (profiles/r04_summary.md).
template <int FPL, typename Theta>
__device__ __forceinline__ void lean_harm_values(double s0, double c0, double s1, double c1, double k2, bool straddle, Theta theta,
                                                 TrigTab trig, const double (&poly)[16], double (&out)[FPL]) {
    static_assert(FPL % SH_LEAN_CH == 0 && FPL <= 8, "frames per lane");
    double s[FPL], c[FPL];
    s[0] = s0; c[0] = c0; s[1] = s1; c[1] = c1;
    if (straddle) {
#pragma unroll
        for (int j = 2; j < FPL; ++j) shm::sincos_tab(theta(j), trig, s[j], c[j]);
    } else {
#pragma unroll
        for (int j = 2; j < FPL; ++j) {
            s[j] = fma(k2, s[j - 1], -s[j - 2]);
            c[j] = fma(k2, c[j - 1], -c[j - 2]);
        }
    }
#pragma unroll
    for (int h = 0; h < FPL; h += SH_LEAN_CH) {
        double p[SH_LEAN_CH];
#pragma unroll
        for (int jj = 0; jj < SH_LEAN_CH; ++jj) p[jj] = fma(poly[0], c[h + jj], poly[1]);
#pragma unroll
        for (int u = 2; u < 16; ++u) {
#pragma unroll
            for (int jj = 0; jj < SH_LEAN_CH; ++jj) p[jj] = fma(p[jj], c[h + jj], poly[u]);
        }
#pragma unroll
        for (int jj = 0; jj < SH_LEAN_CH; ++jj) out[h + jj] = p[jj] * s[h + jj];
    }
}
// envelope of the lane's frame j of a tile: one line in front of frame ci (tile-relative), another from there on
__device__ __forceinline__ double tile_envelope(double x, double ci, double ea0, double ea1, double eb0, double eb1) {
    return x < ci ? fma(x, ea1, ea0) : fma(x, eb1, eb0);
}

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

// sin(t) for N angles in lockstep: shm::sincos_tab_n's sine, bit for bit (same operations in the same order; the cosine outputs are not
// computed).  `magic` (1.5 * 2^52) and `s5` (1/120) arrive as VGPR-RESIDENT values: a float64 VALU instruction reads ONE scalar operand,
// so fma(t, INV_STEP, MAGIC) and fma(z, 1/120, -1/6) cost a v_mov_b64 each, per angle, wherever the compiler lacks the registers to
// keep the constants around -- 7 v_mov_b64 per frame in round 3's FM loop (profiles/r04_fm_isa.md).
template <int N>
__device__ __forceinline__ void sin_tab_n(const double (&t)[N], TrigTab tab, double magic, double s5, double (&s)[N]) {
    double r[N], S[N], C[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        union { double d; uint64_t u; } m;
        m.d = fma(t[j], shm::TRIG_INV_STEP, magic);
        const double fk = m.d - magic;
        const uint32_t k = (uint32_t)m.u & (uint32_t)(shm::TRIG_N - 1);
        S[j] = tab[k].s;
        C[j] = tab[k].c;
        r[j] = fma(-fk, shm::TRIG_STEP_2, fma(-fk, shm::TRIG_STEP_1, t[j]));
    }
    double z[N], sr[N], cr[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = r[j] * r[j];
#pragma unroll
    for (int j = 0; j < N; ++j) sr[j] = fma(z[j], s5, -0.16666666666666666);
#pragma unroll
    for (int j = 0; j < N; ++j) cr[j] = fma(z[j], 0.041666666666666664, -0.5);
#pragma unroll
    for (int j = 0; j < N; ++j) sr[j] = fma(r[j] * z[j], sr[j], r[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) cr[j] = fma(z[j], cr[j], 1.0);
#pragma unroll
    for (int j = 0; j < N; ++j) s[j] = fma(S[j], cr[j], C[j] * sr[j]);
}

// The lean FM arithmetic for FPL frames of one lane, 64 samples apart: a Sine carrier with a closed-form Sine LFO (the arithmetic of
// voice_block's FM path).  The LFO angle a_rel + i*d is exactly linear in i, and only its COSINE enters L(i) = K (C0 - cos) + bias*(start+i):
// one table lookup for the lane's first frame, one rotation for the second, then cos[j] = 2cos(64d) cos[j-1] - cos[j-2].  The carrier's
// angle is not linear in i: one table lookup per frame, SH_FM_Q carriers at a time so that their table reads are in flight together.
// BIASED = false: an LFO without bias (the usual modulator; BASELINE config 3) -- the linear term of L is 0 * (start + i) = +0, and
// fma(K, C0 - cos, +0) IS the rounded product: three operations per frame less, the same bits.
// The accumulated TIME of the lane's frame j: LINEAR -- the tile lies on ONE phase-table piece, t_j = fma(i_j - off, dt, t_base): the
// true value is a float64 by the table's construction, so t_first + j * t_step (t_step = 64 dt, a power-of-two multiple: exact) summed
// step by step gives the same bits with ONE addition per frame and one scalar operand; else time(j) picks the piece per lane
// (LaneTheta: the one tile per crossing that straddles a piece end).  The sines go to out[]: the caller accumulates them in ONE place
// behind the four forms of this function (with the accumulation inside each, the compiler shuffled the accumulators between the forms'
// register assignments: ~25 register copies per eight frames).
#ifndef SH_FM_Q
#define SH_FM_Q 2        // carriers looked up at a time (four: 64 B of scratch in the FM-only lean kernel, 2 % slower; 260 B in the tiles kernel)
#endif
template <int FPL, bool BIASED, bool LINEAR, int QMAX = SH_FM_Q, typename TimeFn>
__device__ __forceinline__ void lean_fm_frames(const double (&poly)[16], double lfo_a_rel, double startd, double lfo_d, double lfo_K, double lfo_C0,
                                               double lrc, double lrs, double di0, TimeFn time,
                                               double t_first, double t_step, TrigTab trig, double (&out)[FPL]) {
    // (lfo_a_rel: the LFO's angle at the launch's -- or tile's -- frame 0; startd: the voice's own index of that frame; lfo_d, lfo_K,
    //  lfo_C0, (lrc, lrs) = (cos, sin)(64 lfo_d): the constants of the piece of the LFO's table the frames lie on)
    const double frequency = poly[0], f_inc = poly[2], lfo_bias = poly[7];
    double magic = 6755399441055744.0, s5 = 0.008333333333333333, phase0 = poly[1];
    asm volatile("" : "+v"(magic), "+v"(s5), "+v"(phase0));          // (VGPR-resident: see sin_tab_n)
    double ls0, lc0;
    shm::sincos_tab(fma(di0, lfo_d, lfo_a_rel), trig, ls0, lc0);
    double lc1 = fma(lc0, lrc, -(ls0 * lrs));
    const double lk2 = lrc + lrc;
    double tj = t_first;
    constexpr int Q = FPL < QMAX ? FPL : QMAX;      // (carriers looked up at a time: their temporaries are most of the loop's registers)
#pragma unroll
    for (int h = 0; h < FPL; h += Q) {
        double th[Q], sn[Q];
#pragma unroll
        for (int jj = 0; jj < Q; ++jj) {
            const int j = h + jj;
            const double Ln = BIASED ? fma(lfo_K, lfo_C0 - lc0, lfo_bias * (startd + (di0 + (double)(j * 64)))) : lfo_K * (lfo_C0 - lc0);
            const double tt = LINEAR ? tj : time(j);
            th[jj] = frequency * tt + fma(f_inc, Ln, phase0);
            if (LINEAR) tj = tj + t_step;
            const double lc2 = fma(lk2, lc1, -lc0);
            lc0 = lc1;
            lc1 = lc2;
        }
        sin_tab_n<Q>(th, trig, magic, s5, sn);
#pragma unroll
        for (int jj = 0; jj < Q; ++jj) out[h + jj] = sn[jj];
    }
}

// ---- the lean lists ----------------------------------------------------------------------------------------------------------------
// One table lookup, FPL-1 recurrence steps, the Horner chains, two accumulations per frame.  Wave w takes every WAVES-th list entry;
// the offset carries over from chunk to chunk (`first`: in and out) so that the waves' shares of the whole group differ by at most one
// voice.  KINDS: LEAN_K_HARM -- every lean record is a polynomial Harmonics voice (FM Sine and the plain waveforms are kept out of that
// instantiation: the extra branches and code cost its loop 5 %); LEAN_K_FM -- every one is an FM Sine voice (BASELINE config 3);
// LEAN_K_ALL -- anything.  SEG: the records of a segmented launch (sloped gains; the second
// piece's fields belong in the first batch of loads -- half the tiles lie behind the crossing).
template <int WAVES, int FPL, int KINDS, bool SEG>
__device__ __forceinline__ void lean_lists(const TileCtx& T, uint32_t& first, TrigTab trig, double (&accl)[FPL], double (&accr)[FPL]) {
    const uint32_t lane = T.lane, tile0 = T.tile0, tile_last = T.tile_last;
    for (uint32_t c = T.c0; c < T.c1; ++c) {
        const uint32_t nfast = as_const(T.set.counts)[4 * c];
        // (one list entry; CLS: the run of the list it lies in -- LEAN_K_HARM, LEAN_K_FM or LEAN_K_REST)
        auto entry = [&](auto cls_, const FastRec SH_CONST_AS* q, uint32_t p) __attribute__((always_inline)) {
            constexpr int CLS = decltype(cls_)::value;
            (void)p;
#ifdef SH_DIAG2
            const bool d2on = blockIdx.x == 1 && blockIdx.y == 1 && T.wave == 0 && c == T.c0 + 1;
            uint64_t d2t[5] = {0, 0, 0, 0, 0};
            if (d2on) d2t[0] = __builtin_amdgcn_s_memtime();
#endif
            // the common 192 bytes in ONE batch of scalar loads: the empty asm makes these fields live here, so the
            // compiler cannot sink their loads behind the tests below (it did: three dependent round trips per voice).
            // The second piece's fields are NOT in the list: their loads stay inside the rare crossing branches.
            const double gl = q->gain_l, gr = q->gain_r;
            const uint32_t remain = q->remain, kind = q->kind;
            const uint32_t lsplit = CLS == LEAN_K_FM ? q->pad1 : 0xFFFFFFFFu;      // (FM Sine: where the LFO's table piece ends; same batch of loads)
            const double ta = q->t_base, da = q->dt, rca = q->rot_c, rsa = q->rot_s;
            double poly[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) poly[u] = q->poly[u];
            if constexpr (CLS == LEAN_K_FM) asm volatile("" :: "s"(lsplit));
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
            if constexpr (SEG) {
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
            if constexpr (CLS == LEAN_K_FM) {
                // every lean record of the bank is an FM Sine voice: the FM code alone (no polynomial path, no waveform branches: the kernel
                // with all kinds needs 128 VGPRs and spills; round 4)
                double sn[FPL];
                // the LFO's table piece: the launch's first, or -- the tiles from FastRec::pad1 on, a multiple of 1024 frames and so of every
                // shape's tile: the first tile boundary behind the piece's end -- the next one (prepare_voice says what that costs)
                double l_a = poly[3], l_d = poly[4], l_K = poly[5], l_C = poly[6], l_rc = poly[8], l_rs = poly[9];
                if (tile0 >= lsplit) { l_a = poly[11]; l_d = poly[12]; l_K = poly[13]; l_C = poly[14]; l_rc = poly[15]; l_rs = q->pad2; }
                if (straddle) {
                    lean_fm_frames<FPL, false, false>(poly, l_a, poly[10], l_d, l_K, l_C, l_rc, l_rs, di0, theta, 0.0, 0.0, trig, sn);
                } else {
                    const double t_first = fma(di0 - off, dt, t_base), t_step = 64.0 * dt;
                    lean_fm_frames<FPL, false, true>(poly, l_a, poly[10], l_d, l_K, l_C, l_rc, l_rs, di0, theta, t_first, t_step, trig, sn);
                }
#pragma unroll
                for (int j = 0; j < FPL; ++j) {
                    accl[j] = fma(gl, sn[j], accl[j]);
                    accr[j] = fma(gr, sn[j], accr[j]);
                }
                return;
            }
            if constexpr (CLS == LEAN_K_HARM) {
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
                if constexpr (SEG) {
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
                return;
            }
            if constexpr (CLS == LEAN_K_REST) {
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
                return;
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
        };
        const FastRec SH_CONST_AS* q = as_const(T.set.fast) + c * 64 + first;
        uint32_t p = first;
        if constexpr (KINDS == LEAN_K_ALL) {
            // the list's three runs (prepare_chunk), one loop each: a loop over a switch of the kinds cost the Harmonics entries 16 % and
            // the FM Sine entries 25-100 % of what the kernels of unmixed banks need (accumulator copies where the branches join, spills)
            const uint32_t runs = as_const(T.set.counts)[4 * c + 3];
            const uint32_t end_harm = runs & 0xFFu, end_fm = end_harm + ((runs >> 8) & 0xFFu);
            for (; p < end_harm; p += WAVES, q += WAVES) entry(std::integral_constant<int, LEAN_K_HARM>{}, q, p);
            for (; p < end_fm; p += WAVES, q += WAVES) entry(std::integral_constant<int, LEAN_K_FM>{}, q, p);
            for (; p < nfast; p += WAVES, q += WAVES) entry(std::integral_constant<int, LEAN_K_REST>{}, q, p);
        } else {
            // A SIMD serves its OLDEST wavefront first: of the three wavefronts it holds one ends at 25 us, one at 33, one at 42 -- and a
            // launch that nothing runs beside (a render that stands alone, a long launch of a run of blocks) leaves every SIMD with two
            // and then ONE wavefront, which issues at half rate (profiles/r04_headline_phases.md).  There a wavefront's priority FALLS as
            // it gets on with its lists (s_setprio 3 .. 0), the ones behind catch up and the three end together: a lone block
            // 57.1 -> 54.1 us, one launch of 2 / 4 / 8 blocks 48.6 -> 46.8 / 42.0 -> 40.4 / 38.4 -> 37.4 us per block.  Not beside another
            // launch (the stream of blocks: its tail is the successor's head; + 0.6 us per block with the ladder):
            // profiles/r06_prio_ladder.txt.
            const uint32_t span = (T.c1 - T.c0) * 64u;
            for (; p < nfast; p += WAVES, q += WAVES) {
                if (T.alone) {                                   // (uniform; ONE loop body: the entry is 5 KB of straight-line code)
                    // GEOMETRIC steps -- half, a quarter, an eighth, an eighth of the group's voices: the wavefronts of a SIMD can be one step
                    // apart at most, and what matters is how far apart they END (quarters: lone 53.0, one launch of 16 blocks 33.6 us per
                    // block; geometric: 52.3 and 33.0)
                    const uint32_t at8 = ((c - T.c0) * 64u + p) * 8u;      // (compares, no division)
                    if (at8 < 4 * span) __builtin_amdgcn_s_setprio(3);
                    else if (at8 < 6 * span) __builtin_amdgcn_s_setprio(2);
                    else if (at8 < 7 * span) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                entry(std::integral_constant<int, KINDS>{}, q, p);
            }
        }
        first = p - nfast;                                    // 0 .. WAVES-1: where the stride lands in the next list
    }
}

// ---- the general lists: every other sounding voice, through voice_block -------------------------------------------------------------
// (the stride simply continues from the lean lists, so the extra voices go to the waves that got one lean voice fewer; nsub > 1: the
// first segment of a general segmented launch, its list entries dealt to nsub workgroups)
template <int WAVES, int FPL>
__device__ __forceinline__ void general_lists(const LaunchArgs& A, const TileCtx& T, uint32_t& first, uint32_t nsub, const uint32_t (&i)[FPL],
                                              const double (&di)[FPL], TrigTab trig, double (&accl)[FPL], double (&accr)[FPL]) {
    for (uint32_t c = T.c0; c < T.c1; ++c) {
        const uint32_t ngen = as_const(T.set.counts)[4 * c + 1];
        const uint32_t SH_CONST_AS* idx = as_const(T.set.gen_idx) + c * 64;
        const uint32_t step = WAVES * nsub;
        uint32_t p = first;
        uint32_t vi_next = p < ngen ? idx[p] : 0u;
        for (; p < ngen; p += step) {
            const uint32_t vi = vi_next;
            vi_next = p + step < ngen ? idx[p + step] : 0u;       // in flight with this voice's record: one round trip less
            const VoiceRegs r = load_record(as_const(T.set.launch) + vi);
            general_voice<FPL>(r, T.set.fm + vi, A.B, A.B.voices + vi, T.st0, T.tile0, T.nfr, i, di, trig, accl, accr);
        }
        first = p - ngen;
    }
}

// (total general voices of the group's chunks in a record set: wave-uniform scalar loads)
__device__ __forceinline__ uint32_t general_count(const uint32_t* counts, uint32_t c0, uint32_t c1) {
    uint32_t total = 0;
    for (uint32_t c = c0; c < c1; ++c) total += as_const(counts)[4 * c + 1];
    return total;
}

// ---- tile-classified launches: the lean pairs of a tile --------------------------------------------------------------------------------
// Per chunk of the group a compacted list of 128-byte records (the count: the bits of the chunk's mask), walked like the lean lists of
// an ordinary launch.  The voice groups of a tile-classified launch do not partition the CHUNKS but every chunk's list: entry p of a
// list goes to group p / WAVES mod groups, wave p mod WAVES (the offset carries over from list to list) -- notes that sound together
// are neighbours in the voice table, whole chunks of them, and any deal of whole chunks leaves one group with twice the work of
// another.  The masks of the tile (one contiguous row) are fetched 64 at a time, one per lane; the lists that are not empty are handed
// round by ballot and readlane.  WAVEFORMS: the bank holds plain Sawtooth / Square / Triangle / Pulse / FM Sine voices too (the
// waveform branch costs the Harmonics loop registers: an instantiation of its own).
enum { TILE_RUN_HARM = 0, TILE_RUN_FM = 1, TILE_RUN_REST = 2 };      // the runs of a (tile, chunk) list
template <int WAVES, int FPL, bool WAVEFORMS>
__device__ __forceinline__ void tiles_lean(const BankPtrs& B, const TileCtx& T, TrigTab trig, double (&accl)[FPL], double (&accr)[FPL]) {
    static_assert(64 * FPL == TILE_FRAMES, "the lean kernel's tile is the tile of the classification");
    const uint32_t lane = T.lane, ngroups = T.ngroups, tile_index = T.tile_index;
    const size_t slots = (size_t)B.tiles.rec_chunks * 64;
    const TileRec SH_CONST_AS* trow = as_const(B.tiles.recs) + (size_t)tile_index * slots;
    // (only the masks k0 .. k1 - 1 of every group: the chunks of the set's range -- see TileSet)
    const uint32_t kw = B.tiles.k1 - B.tiles.k0, nmask = ngroups * kw, stride = ngroups * WAVES;
    const uint64_t* __restrict__ lrow = B.tiles.lean + (size_t)tile_index * (ngroups * B.tiles.mask_k);
    // (everything about WHICH pair comes next is wave-uniform -- ballots, set bits, popcounts -- and is kept in scalar registers by
    // hand: left to itself the compiler did the arithmetic of nth_set_bit on the vector unit, ~50 VALU instructions per pair of the
    // ~275: profiles/r04_summary.md, 44 % of the tiles kernel's VALU instructions were not float64)
    uint32_t firstp = __builtin_amdgcn_readfirstlane(T.grp * WAVES + T.wave);
    for (uint32_t base = 0; base < nmask; base += 64) {
        const uint32_t mi = base + lane;
        const uint64_t mymask = mi < nmask ? lrow[(mi / kw) * B.tiles.mask_k + B.tiles.k0 + mi % kw] : 0ull;
        const uint64_t have0 = __ballot(mymask != 0ull);
        uint64_t have = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(have0 >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)have0);
        while (have) {
            const uint32_t src = (uint32_t)__builtin_ctzll(have);
            have &= have - 1;
            const uint64_t cmask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mymask >> 32), (int)src) << 32) |
                                   (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mymask, (int)src);
            const uint32_t npairs = (uint32_t)cmask;                      // (the chunk's word: pairs | Harmonics pairs << 32 | FM Sine pairs << 40)
            const uint32_t idx = base + src;                              // (group, k - k0) of the prepare step's layout
            const uint32_t c = idx / kw + (B.tiles.k0 + idx % kw) * ngroups;
            // (one pair; CLS: the run of the chunk's list it lies in)
            auto pair = [&](auto cls_, const TileRec SH_CONST_AS* q) __attribute__((always_inline)) {
                constexpr int CLS = decltype(cls_)::value;
                const double t0 = q->t0, dt = q->dt, rc = q->rc, rs = q->rs, ea0 = q->ea0, ea1 = q->ea1, GL = q->GL, GR = q->GR;
                const uint32_t pc = *reinterpret_cast<const uint32_t SH_CONST_AS*>(&q->npieces);      // npieces | corner << 16
                uint32_t lane_again = lane;                               // (converted per entry: two registers less across the loop)
                asm volatile("" : "+v"(lane_again));
                const double lane_d = (double)lane_again;
                // the voice of the list's entry p: its position in the chunk rides in the record (round 3 found it as the chunk's (p + 1)-th
                // set bit -- an address that did not wait for the record, at ~50 VALU instructions per pair: with the scalar registers this
                // kernel has left, the compiler did that uniform arithmetic on the vector unit); its polynomial comes from the table by
                // voice (read by every tile's workgroups: it lives in L2)
                const uint32_t vword = *reinterpret_cast<const uint32_t SH_CONST_AS*>(reinterpret_cast<const char SH_CONST_AS*>(&q->pad_) + 4);
                const uint32_t vbit = vword & 0xFFu;
                const double SH_CONST_AS* pp = as_const(B.polys) + (size_t)(c * 64 + vbit) * 16;
                double poly[16];
#pragma unroll
                for (int v_ = 0; v_ < 16; ++v_) poly[v_] = pp[v_];
                asm volatile("" :: "s"(t0), "s"(dt), "s"(rc), "s"(rs), "s"(ea0), "s"(ea1), "s"(GL), "s"(GR), "s"(pc),
                             "s"(poly[0]), "s"(poly[1]), "s"(poly[2]), "s"(poly[3]), "s"(poly[4]), "s"(poly[5]),
                             "s"(poly[6]), "s"(poly[7]), "s"(poly[8]), "s"(poly[9]), "s"(poly[10]), "s"(poly[11]), "s"(poly[12]),
                             "s"(poly[13]), "s"(poly[14]), "s"(poly[15]));
                const LaneTheta none{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0u, 0u, false};
                if constexpr (CLS != TILE_RUN_HARM) {
                    // (an FM pair: the constants of the piece of the LFO's table the tile lies on -- its index in `segs` rides above the kind;
                    //  0xFFFFFF: a constant LFO, the voice's own constants)
                    double lfo_d = 0.0, lfo_K = 0.0, lfo_C0 = 0.0, lrc = 1.0, lrs = 0.0, lfo2_a = 0.0, lfo2_d = 0.0, lfo2_K = 0.0, lfo2_C0 = 0.0;
                    uint32_t lfo_split = 0;                          // (the LFO's next piece takes over at this frame of the tile; 0: not in this tile)
                    if constexpr (CLS == TILE_RUN_FM) {
                        const uint32_t lfo_idx = *reinterpret_cast<const uint32_t SH_CONST_AS*>(&q->pad_) >> 8;
                        const sh_segment SH_CONST_AS* lpc_ = as_const(B.segs) + (lfo_idx != 0xFFFFFFu ? lfo_idx : 0u);
                        if (lfo_idx != 0xFFFFFFu) {
                            const sh_segment SH_CONST_AS* lpc = lpc_;
                            const double2 SH_CONST_AS* lro = as_const(B.seg_rot) + lfo_idx;
                            lfo_d = lpc[0].dt; lfo_K = lpc[1].t0; lfo_C0 = lpc[1].dt; lrc = lro->x; lrs = lro->y;
                        } else {
                            lfo_d = poly[4]; lfo_K = poly[5]; lfo_C0 = poly[6]; lrc = poly[8]; lrs = poly[9];
                        }
                        lfo_split = vword >> 8;
                        if (lfo_split) {
                            // ... whose constants follow in `segs`; the angle of frame x on it: t0' + (x - split - 1/2) d'
                            lfo2_d = lpc_[2].dt; lfo2_a = fma(-(double)lfo_split - 0.5, lfo2_d, lpc_[2].t0); lfo2_K = lpc_[3].t0; lfo2_C0 = lpc_[3].dt;
                        }
                        if (pc == 1u && lfo_split == 0u) {
                            // an FM Sine pair on one piece under one line (the sustain, the release): the arithmetic of the lean lists --
                            // the time by one addition per frame, the LFO's cosine by recurrence (lean_fm_frames); the record's rc is the
                            // voice's own index of the tile's first frame, rs the LFO's angle there
                            double sn[FPL];
                            const double t_first = fma(lane_d, dt, t0), t_step = 64.0 * dt;
                            lean_fm_frames<FPL, false, true>(poly, rs, rc, lfo_d, lfo_K, lfo_C0, lrc, lrs, lane_d, none, t_first, t_step, trig, sn);
                            double gl_e = GL, gr_e = GR;
                            if (ea1 == 0.0) {
                                gl_e = GL * ea0;
                                gr_e = GR * ea0;
                            } else {
#pragma unroll
                                for (int j = 0; j < FPL; ++j) sn[j] = sn[j] * fma(lane_d + (double)(j * 64), ea1, ea0);
                            }
#pragma unroll
                            for (int j = 0; j < FPL; ++j) {
                                accl[j] = fma(gl_e, sn[j], accl[j]);
                                accr[j] = fma(gr_e, sn[j], accr[j]);
                            }
                            return;
                        }
                    }
                    // an FM Sine pair of another form, or a plain Sawtooth / Square / Triangle / Pulse (unit amplitude, t in turns: the
                    // amplitude lives in the gains): every frame from its accumulated t on the piece that holds it -- the record's
                    // pieces, or a walk along the voice's table -- the envelope's line of the frame, and nothing in front of an onset
                    const uint32_t wkind = *reinterpret_cast<const uint32_t SH_CONST_AS*>(&q->pad_) & 0xFFu;
                    {
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
                        if constexpr (CLS == TILE_RUN_FM) {
                            // a Sine carrier with a closed-form Sine LFO: th[] is the accumulated TIME; the carrier's angle from the
                            // running sum of the LFO, L(n) = K (C0 - cos(a + (n - 1/2) d)) + bias n, at the voice's own index n
                            // (a pair with a corner, further pieces or a walk: every frame by two lookups)
                            const double fr = poly[0], phase0 = poly[1], f_inc = poly[2], lfo_bias = poly[7], n_first = rc;
                            const double a_rel = rs;             // (the LFO's angle at the tile's first frame, on its piece: prepare_tiles_wave)
#pragma unroll
                            for (int j = 0; j < FPL; ++j) {
                                const double x = lane_d + (double)(j * 64);
                                double ls, lc, sj, cj;
                                const bool behind = lfo_split != 0u && x >= (double)lfo_split;
                                shm::sincos_tab(behind ? fma(x, lfo2_d, lfo2_a) : fma(x, lfo_d, a_rel), trig, ls, lc);
                                const double Ln = fma(behind ? lfo2_K : lfo_K, (behind ? lfo2_C0 : lfo_C0) - lc, lfo_bias * (n_first + x));
                                shm::sincos_tab(fr * th[j] + fma(f_inc, Ln, phase0), trig, sj, cj);
                                const double ej = x < ci ? fma(x, ea1, ea0) : fma(x, eb1, eb0);
                                const double w = x >= on ? sj * ej : 0.0;
                                accl[j] = fma(GL, w, accl[j]);
                                accr[j] = fma(GR, w, accr[j]);
                            }
                            return;
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
                        return;
                    }
                }
                if constexpr (CLS == TILE_RUN_HARM) {
                // ---- a polynomial-Harmonics pair: w[j] = sample x envelope of the lane's frame j by the pair's kind, then ONE accumulation ----
                double w[FPL], gl_e = GL, gr_e = GR;
                if (pc == 1u) {
                    // one piece, one line (seven pairs of eight): the lean arithmetic of an ordinary launch
                    double s0, c0s, s1, c1s;
                    shm::sincos_tab(fma(lane_d, dt, t0), trig, s0, c0s);
                    s1 = fma(s0, rc, c0s * rs);
                    c1s = fma(c0s, rc, -(s0 * rs));
                    lean_harm_values<FPL>(s0, c0s, s1, c1s, rc + rc, false, none, trig, poly, w);
                    if (ea1 == 0.0) {        // a flat line (the sustain: two thirds of a note's life): the line folded into the gains
                        gl_e = GL * ea0;
                        gr_e = GR * ea0;
                    } else {
#pragma unroll
                        for (int j = 0; j < FPL; ++j) w[j] = w[j] * fma(lane_d + (double)(j * 64), ea1, ea0);
                    }
                } else if ((pc & 0xFFFFu) == 1u) {
                    // one piece, a corner: the envelope changes lines at frame pc >> 16
                    const double eb0 = q->eb0, eb1 = q->eb1, ci = (double)(pc >> 16);
                    double s0, c0s, s1, c1s;
                    shm::sincos_tab(fma(lane_d, dt, t0), trig, s0, c0s);
                    s1 = fma(s0, rc, c0s * rs);
                    c1s = fma(c0s, rc, -(s0 * rs));
                    lean_harm_values<FPL>(s0, c0s, s1, c1s, rc + rc, false, none, trig, poly, w);
#pragma unroll
                    for (int j = 0; j < FPL; ++j) w[j] = w[j] * tile_envelope(lane_d + (double)(j * 64), ci, ea0, ea1, eb0, eb1);
                } else if ((pc & 0xFFFFu) == 0u) {
                    // a WALK pair (see TileRec): lanes 0 .. 15 fetch a piece of the voice's table each -- one round trip -- and every
                    // frame takes the angle of the last piece that starts at or in front of it; a frame in front of them all (the
                    // onset lies inside the tile) keeps the angle 0, whose sine is 0
                    const double eb0 = q->eb0, eb1 = q->eb1, dn0 = q->tb[0], ci = (double)(pc >> 16);
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
#pragma unroll
                    for (int j = 0; j < FPL; ++j) {
                        double sj, cj;
                        shm::sincos_tab(th[j], trig, sj, cj);
                        double pj = fma(poly[0], cj, poly[1]);
#pragma unroll
                        for (int u = 2; u < 16; ++u) pj = fma(pj, cj, poly[u]);
                        w[j] = (pj * sj) * tile_envelope(lane_d + (double)(j * 64), ci, ea0, ea1, eb0, eb1);
                    }
                } else {
                    // piece ends inside the tile: every frame by lookup from the piece that holds it
                    const double eb0 = q->eb0, eb1 = q->eb1, ci = (double)(pc >> 16);
                    const TileTheta theta{lane_d, t0, dt, q->tb[0], q->tb[1], q->db[0], q->db[1], lane, q->split[0], q->split[1]};
                    double s0, c0s, s1, c1s;
                    shm::sincos_tab(theta(0), trig, s0, c0s);
                    shm::sincos_tab(theta(1), trig, s1, c1s);
                    lean_harm_values<FPL>(s0, c0s, s1, c1s, 0.0, true, theta, trig, poly, w);
#pragma unroll
                    for (int j = 0; j < FPL; ++j) w[j] = w[j] * tile_envelope(lane_d + (double)(j * 64), ci, ea0, ea1, eb0, eb1);
                }
#pragma unroll
                for (int j = 0; j < FPL; ++j) {
                    accl[j] = fma(gl_e, w[j], accl[j]);
                    accr[j] = fma(gr_e, w[j], accr[j]);
                }
                }
            };
            const TileRec SH_CONST_AS* q = trow + (c - B.tiles.k0 * ngroups) * 64 + firstp;
            uint32_t p = firstp;
            if constexpr (WAVEFORMS) {
                // the list's three runs (prepare_tiles_wave), one loop each -- see lean_lists
                const uint32_t end_harm = (uint32_t)(cmask >> 32) & 0xFFu, end_fm = end_harm + ((uint32_t)(cmask >> 40) & 0xFFu);
                for (; p < end_harm; p += stride, q += stride) pair(std::integral_constant<int, TILE_RUN_HARM>{}, q);
                for (; p < end_fm; p += stride, q += stride) pair(std::integral_constant<int, TILE_RUN_FM>{}, q);
                for (; p < npairs; p += stride, q += stride) pair(std::integral_constant<int, TILE_RUN_REST>{}, q);
            } else {
                for (; p < npairs; p += stride, q += stride) pair(std::integral_constant<int, TILE_RUN_HARM>{}, q);
            }
            firstp = __builtin_amdgcn_readfirstlane(p - npairs);
        }
    }
}

// ---- tile-classified launches: the general pairs of a tile ----------------------------------------------------------------------------
// The (voice, tile) pairs of this tile that sound but are not lean, ALL voice groups': the masks of the classification tile are one
// contiguous row -- a lane fetches one mask per pass, the non-zero ones are handed round by ballot and readlane -- and the pairs are
// dealt to the GEN_SPLIT workgroups of the tile and their waves by their ordinal; each goes through the launch record and voice_block.
// false: the tile holds no general pair at all (nearly every tile: the first tiles of a note are walk pairs of the lean kernel).
template <int WAVES>
__device__ __forceinline__ bool tile_has_general(const BankPtrs& B, uint32_t class_tile) {
    const uint32_t kw = B.tiles.k1 - B.tiles.k0, nmask = B.tiles.groups * kw;
    const uint64_t* __restrict__ grow = B.tiles.gen + (size_t)class_tile * (B.tiles.groups * B.tiles.mask_k);
    bool any = false;
    for (uint32_t base = 0; base < nmask; base += 64) {
        const uint32_t mi = base + (threadIdx.x & 63);
        const uint64_t mine = mi < nmask ? grow[(mi / kw) * B.tiles.mask_k + B.tiles.k0 + mi % kw] : 0ull;
        any = any || __ballot(mine != 0ull) != 0ull;
    }
    return any;
}
template <int WAVES, int FPL>
__device__ __forceinline__ void tiles_general(const LaunchArgs& A, const TileCtx& T, uint32_t gen_part, const uint32_t (&i)[FPL], const double (&di)[FPL],
                                              TrigTab trig, double (&accl)[FPL], double (&accr)[FPL]) {
    const BankPtrs& B = A.B;
    const uint32_t kw = B.tiles.k1 - B.tiles.k0, nmask = B.tiles.groups * kw;
    const uint64_t* __restrict__ grow = B.tiles.gen + (size_t)(T.tile0 / TILE_FRAMES) * (B.tiles.groups * B.tiles.mask_k);
    uint32_t ord = 0;
    for (uint32_t base = 0; base < nmask; base += 64) {
        const uint32_t mi = base + T.lane;
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
                if ((ord++ % (uint32_t)(WAVES * GEN_SPLIT)) != T.wave * GEN_SPLIT + gen_part) continue;
                const uint32_t vi = c * 64 + bit;
                const VoiceRegs r = load_record(as_const(T.set.launch) + vi);
                general_voice<FPL>(r, T.set.fm + vi, B, B.voices + vi, T.st0, T.tile0, T.nfr, i, di, trig, accl, accr);
            }
        }
    }
}
// zeros into a workgroup's share of a plane of general parts (a tile / a (tile, group) without a general voice), before any set-up
template <int WAVES, int FPL>
__device__ __forceinline__ void zero_tile(double2* __restrict__ plane, uint32_t tile_first, uint32_t nfr) {
    for (uint32_t f = threadIdx.x; f < 64 * FPL; f += WAVES * 64) {
        const uint32_t raw = tile_first + f;
        if (raw < nfr) plane[raw] = make_double2(0.0, 0.0);
    }
}

template <int FPL>
__device__ __forceinline__ void clear_acc(double (&accl)[FPL], double (&accr)[FPL]) {
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        accl[j] = 0.0;
        accr[j] = 0.0;
    }
}

// =====================================================================================================================================
// k_render_lean: grid = (tiles [of all segments], voice groups + rows that resolve the next-but-one block's records)
// =====================================================================================================================================
template <int WAVES, int FPL, int MINW, int KINDS, bool SEG>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_render_lean(LaunchArgs A, NextArgs N, FoldIn F, double2* __restrict__ parts) {
    SH_STAMP(A, 0);
    const uint32_t ngroups = groups_of_grid(N.prep_wgs);
    if (prep_rows_lists(A, N, ngroups)) return;
    fold_previous<WAVES, FPL>(F, ngroups, A.nframes, blockIdx.x);
    if constexpr (!SEG) {
        if (A.self_prepare) prepare_own_group<WAVES>(A, blockIdx.y);     // (first: nothing of the tile's own state is alive yet)
    }
    SH_STAMP(A, 1);
    __shared__ double red[WAVES][2][64 * red_rows(FPL)];
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    load_trig(trig, A.trig_g, WAVES * 64);
    SH_STAMP(A, 2);
    TileCtx T = plain_tile<FPL>(A, blockIdx.x, blockIdx.y, ngroups);
    if constexpr (SEG) enter_segment<FPL>(T, A);
    double accl[FPL], accr[FPL];
    clear_acc(accl, accr);
    uint32_t first = T.wave;
    lean_lists<WAVES, FPL, KINDS, SEG>(T, first, trig, accl, accr);
    SH_STAMP(A, 3);
    SH_STAMP(A, 4);
    reduce_store<WAVES, FPL>(red, T, accl, accr, parts + (size_t)T.grp * A.nframes, T.seg_off, BusOut{}, !SEG && F.self != nullptr);
    SH_STAMP(A, 5);
    if constexpr (!SEG) {
        if (F.self) fold_own_tile<WAVES, FPL>(F, ngroups, A.nframes, blockIdx.x, reinterpret_cast<uint32_t*>(&red[0][0][0]));
    }
}

// =====================================================================================================================================
// k_render_general: the general lists / pairs behind a lean kernel on the same stream; writes the general parts (planes ngroups ..) and
// gen_valid.  Latency-bound wavefronts that run beside the other stream's lean kernel, whose wavefronts fill every issue slot they are
// given: at raised priority (without it the two do not overlap at all: 60 us instead of 23 for the tiles' general kernel).
// =====================================================================================================================================
enum { GEN_LISTS = 0, GEN_SEG = 1, GEN_TILES = 2 };
template <int WAVES, int FPL, int MINW, int KIND>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_render_general(LaunchArgs A, double2* __restrict__ parts, uint32_t* __restrict__ gen_valid) {
    const BankPtrs& B = A.B;
    uint32_t gen_part = 0, sub = 0, nsub = 1;
    bool to_scratch = false;
    TileCtx T;
    if constexpr (KIND == GEN_TILES) {
        // a one-dimensional grid: GEN_SPLIT workgroups per tile of 64 FPL frames, workgroup u renders part u % GEN_SPLIT of ALL voice
        // groups' general pairs of tile u / GEN_SPLIT into plane gen_part of the general parts
        __builtin_amdgcn_s_setprio(3);
        gen_part = blockIdx.x % GEN_SPLIT;
        const uint32_t bx = blockIdx.x / GEN_SPLIT;
        if (bx * (64 * FPL) >= A.nframes) return;
        if (bx == 0 && gen_part == 0 && threadIdx.x < B.tiles.groups) gen_valid[threadIdx.x] = threadIdx.x < GEN_SPLIT ? 1u : 0u;   // GEN_SPLIT general planes
        double2* plane = parts + (size_t)(B.tiles.groups + gen_part) * A.nframes;
        if (!tile_has_general<WAVES>(B, (bx * (64 * FPL)) / TILE_FRAMES)) {
            zero_tile<WAVES, FPL>(plane, bx * (64 * FPL), A.nframes);
            return;
        }
        T = plain_tile<FPL>(A, bx, 0, B.tiles.groups);
    } else if constexpr (KIND == GEN_LISTS) {
        T = plain_tile<FPL>(A, blockIdx.x, blockIdx.y, gridDim.y);
        // a group without general voices in this launch: nothing to render, nothing to write (wave-uniform: scalar loads)
        const uint32_t total = general_count(A.cur.counts, T.c0, T.c1);
        if (blockIdx.x == 0 && threadIdx.x == 0) gen_valid[blockIdx.y] = total ? 1u : 0u;
        if (total == 0) return;
    } else {
        // a segmented launch: grid.x = the first segment's tiles gen_sub times over (tile-major: gen_sub workgroups share a (tile,
        // group), list entries dealt round robin, slices into gen_scratch), then the other segments' tiles.  Its flags say "every
        // group's general parts are valid" (the host sets them), so every (tile, group) writes -- zeros where it holds no voice.
        T = plain_tile<FPL>(A, blockIdx.x, blockIdx.y, gridDim.y);
        const uint32_t tiles_0 = (B.seg_first[1] - B.seg_first[0] + 64 * FPL - 1) / (64 * FPL);
        if (T.tile_index < tiles_0 * B.gen_sub) {
            sub = T.tile_index % B.gen_sub;
            T.tile_index /= B.gen_sub;
            nsub = B.gen_sub;
            to_scratch = true;
        } else {
            T.tile_index -= tiles_0 * (B.gen_sub - 1);       // as if the first segment's tiles came once
        }
        enter_segment<FPL>(T, A);
        if (!to_scratch && general_count(T.set.counts, T.c0, T.c1) == 0) {
            zero_tile<WAVES, FPL>(parts + (size_t)(T.ngroups + T.grp) * A.nframes + T.seg_off, T.tile0, T.nfr);
            return;
        }
    }
    __shared__ double red[WAVES][2][64 * red_rows(FPL)];
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    load_trig(trig, A.trig_g, WAVES * 64);
    uint32_t i[FPL];
    double di[FPL], accl[FPL], accr[FPL];
    clear_acc(accl, accr);
    build_frames<FPL>(T, T.lane, i, di);
    if constexpr (KIND == GEN_TILES) {
        tiles_general<WAVES, FPL>(A, T, gen_part, i, di, trig, accl, accr);
        reduce_store<WAVES, FPL>(red, T, accl, accr, parts + (size_t)(B.tiles.groups + gen_part) * A.nframes, 0, BusOut{});
    } else {
        uint32_t first = T.wave + sub * WAVES;
        general_lists<WAVES, FPL>(A, T, first, nsub, i, di, trig, accl, accr);
        if (KIND == GEN_SEG && to_scratch)
            reduce_store<WAVES, FPL>(red, T, accl, accr, B.gen_scratch + (size_t)(T.grp * nsub + sub) * T.nfr, 0, BusOut{});
        else
            reduce_store<WAVES, FPL>(red, T, accl, accr, parts + (size_t)(T.ngroups + T.grp) * A.nframes, T.seg_off, BusOut{});
    }
}

// =====================================================================================================================================
// k_render_tiles: grid = (tiles, voice groups + rows behind them: [MERGED: GEN_SPLIT general workgroups per tile,] the workgroups that
// resolve the tile set of the block two launches on)
// =====================================================================================================================================
template <int WAVES, int FPL, int MINW, bool WAVEFORMS, bool MERGED>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_render_tiles(LaunchArgs A, NextArgs N, FoldIn F, double2* __restrict__ parts,
                                                                    uint32_t* __restrict__ gen_valid) {
    const BankPtrs& B = A.B;
    const uint32_t ngroups = groups_of_grid(N.prep_wgs);
    uint32_t gen_unit = 0;
    const int role = prep_rows_tiles<MERGED>(A, N, ngroups, gen_unit);
    if (role == 1) return;
    const bool is_gen_wg = MERGED && role == 2;
    uint32_t gen_part = 0, bx = blockIdx.x;
    if constexpr (MERGED) {
        if (is_gen_wg) {
            gen_part = gen_unit % GEN_SPLIT;
            bx = gen_unit / GEN_SPLIT;
            if (bx * (64 * FPL) >= A.nframes) return;
            if (bx == 0 && gen_part == 0 && threadIdx.x < B.tiles.groups) gen_valid[threadIdx.x] = threadIdx.x < GEN_SPLIT ? 1u : 0u;
            if (!tile_has_general<WAVES>(B, (bx * (64 * FPL)) / TILE_FRAMES)) {
                zero_tile<WAVES, FPL>(parts + (size_t)(B.tiles.groups + gen_part) * A.nframes, bx * (64 * FPL), A.nframes);
                return;
            }
        }
    }
    if (!is_gen_wg) fold_previous<WAVES, FPL>(F, ngroups, A.nframes, bx);
    __shared__ double red[WAVES][2][64 * red_rows(FPL)];
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    load_trig(trig, A.trig_g, WAVES * 64);
    TileCtx T = plain_tile<FPL>(A, bx, blockIdx.y, ngroups);
    double accl[FPL], accr[FPL];
    clear_acc(accl, accr);
    if constexpr (MERGED) {
        if (is_gen_wg) {
            uint32_t i[FPL];
            double di[FPL];
            build_frames<FPL>(T, T.lane, i, di);
            tiles_general<WAVES, FPL>(A, T, gen_part, i, di, trig, accl, accr);
            reduce_store<WAVES, FPL>(red, T, accl, accr, parts + (size_t)(B.tiles.groups + gen_part) * A.nframes, 0, BusOut{});
            return;
        }
    }
    tiles_lean<WAVES, FPL, WAVEFORMS>(B, T, trig, accl, accr);
    reduce_store<WAVES, FPL>(red, T, accl, accr, parts + (size_t)T.grp * A.nframes, 0, BusOut{});
}

// =====================================================================================================================================
// k_render_combined: lean and general lists in one kernel (MODE: which lean kinds the bank can hold; DIRECT: none -- the voice table is
// walked directly).  One voice group: the workgroup writes the caller's bus itself; several (banks without lean candidates,
// SYNTHHIP_NO_SPLIT): partial buses, folded by the launch two on like a split launch's.
// =====================================================================================================================================
enum { COMBINED_DIRECT = 0, COMBINED_LEAN_HARM = 1, COMBINED_LEAN_ALL = 2 };
template <int WAVES, int FPL, int MINW, int MODE>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_render_combined(LaunchArgs A, NextArgs N, FoldIn F, double2* __restrict__ parts, BusOut out) {
    const uint32_t ngroups = groups_of_grid(N.prep_wgs);
    if (prep_rows_lists(A, N, ngroups)) return;
    fold_previous<WAVES, FPL>(F, ngroups, A.nframes, blockIdx.x);
    __shared__ double red[WAVES][2][64 * red_rows(FPL)];
    __shared__ shm::sc_pair trig[shm::TRIG_N];
    load_trig(trig, A.trig_g, WAVES * 64);
    const TileCtx T = plain_tile<FPL>(A, blockIdx.x, blockIdx.y, ngroups);
    uint32_t i[FPL];
    double di[FPL], accl[FPL], accr[FPL];
    clear_acc(accl, accr);
    if constexpr (MODE == COMBINED_DIRECT) {
        build_frames<FPL>(T, T.lane, i, di);
        const uint32_t v0 = blockIdx.y * A.voices_per_group;
        uint32_t v1 = v0 + A.voices_per_group;
        if (v1 > A.nvoices) v1 = A.nvoices;
        const VoiceLaunch SH_CONST_AS* rp = as_const(T.set.launch) + v0 + T.wave;
        for (uint32_t vi = v0 + T.wave; vi < v1; vi += WAVES, rp += WAVES) {
            const VoiceRegs r = load_record(rp);
            if (r.flags & FL_SILENT) continue;       // the note was released before this block: contributes exact zeros
            general_voice<FPL>(r, T.set.fm + vi, A.B, A.B.voices + vi, T.st0, T.tile0, T.nfr, i, di, trig, accl, accr);
        }
    } else {
        uint32_t first = T.wave;
        lean_lists<WAVES, FPL, MODE == COMBINED_LEAN_HARM ? LEAN_K_HARM : LEAN_K_ALL, false>(T, first, trig, accl, accr);
        // The lean loops work from the lane's first frame alone, so the per-frame index arrays are only built behind them, for the
        // general code: they would cost 3 registers per frame for the whole lean loop.
        uint32_t lane_late = T.lane;
        asm volatile("" : "+v"(lane_late));       // defined here, after the loop above: the arrays cannot be built earlier
        build_frames<FPL>(T, lane_late, i, di);
        general_lists<WAVES, FPL>(A, T, first, 1u, i, di, trig, accl, accr);
    }
    reduce_store<WAVES, FPL>(red, T, accl, accr, parts ? parts + (size_t)T.grp * A.nframes : nullptr, 0, out);
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

// The fold as a kernel of its own (the end of a run of renders; single renders): the voice groups' partial buses in group order,
// then the general parts whose flag is set -- the same order as fold_previous -- rounded to float32 / saturated to int16 once.
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
