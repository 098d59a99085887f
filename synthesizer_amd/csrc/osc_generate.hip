// osc_generate.hip -- voices materialised as PCM rows in HBM, and single oscillators.
//
// Kernels
//   k_generate            grid (frame tiles, voice groups): voice-major float32 / float64 rows (4 B per voice-sample)
//   k_generate_lists      whole-bank materialisation through the launch's lean / general / silent lists
//   k_generate_lean_harm  the lean polynomial-Harmonics records by the recurrence, sixteen frames per lane
#include "osc_host.hpp"
#include <stdlib.h>

namespace {

typedef int int2v __attribute__((ext_vector_type(2)));     // a plane element of the fused mixdown: (a, L | U << 16)

// ---- the int16 forms (sh_bank_generate_i16): a row element is int(scale * v) of the float64 sample v -- Sample.from_osc_block's
// quantiser (upstream synthplayer/sample.py; truncation toward zero, OverflowError where the value does not fit) applied where the
// sample is made, so that a voice reaches HBM as 2 bytes instead of 4 (float32 row) + 4 read + 2 written by a quantise pass.
template <typename OutT> struct RowOut;
template <> struct RowOut<float> {
    static constexpr bool I16 = false;
    __device__ __forceinline__ static float make(double x, double, bool&) { return (float)x; }
};
template <> struct RowOut<short> {
    static constexpr bool I16 = true;
    __device__ __forceinline__ static short make(double x, double scale, bool& bad) {
        const double t = trunc(scale * x);                   // float64 product, like the Python expression
        const bool ok = t >= -32768.0 && t <= 32767.0;       // (false for NaN too)
        bad |= !ok;
        return ok ? (short)(int)t : (short)0;
    }
};

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
                                                  size_t stride, short* __restrict__ out16 = nullptr, double scale16 = 0.0,
                                                  int* __restrict__ flag = nullptr) {
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
            const double* fm_p = fm_cumsum;
            const double* pwm_p = pwm;
            if (B.rows) {                                          // (uniform) a bank with modulation rows: this voice's rows of the launch's matrix
                const int32_t fr = as_const(B.fm_row)[first + vi], pr = as_const(B.pwm_row)[first + vi];
                fm_p = fr >= 0 ? B.rows + (size_t)fr * B.row_stride : nullptr;
                pwm_p = pr >= 0 ? B.rows + (size_t)pr * B.row_stride : nullptr;
            }
            voice_block_at<FPL, false>(r, launch_fm + vi, B, B.voices + first + vi, tile0, tile_last, i, di, fm_p, pwm_p, trig, x);
            if (out16) {       // the int16 boundary guard: a polynomial / Clenshaw sample whose int(scale * v) is in doubt is summed term by term
                const double tq = guard_tq(r, B.voices + first + vi, tile_last, scale16);
                if (tq != 0.0 && guard_near<FPL>(x, scale16, tq))
                    voice_block_exact<FPL>(r, launch_fm + vi, B, B.voices + first + vi, tile0, tile_last, i, di, fm_p, pwm_p, trig, x);
            }
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            uint32_t raw = tile0 + j * 64 + lane;
            if (raw < n) {
                if (out32) out32[(size_t)vi * stride + raw] = (float)x[j];
                if (out64) out64[(size_t)vi * stride + raw] = x[j];
                if (out16) {
                    bool bad = false;
                    out16[(size_t)vi * stride + raw] = RowOut<short>::make(x[j], scale16, bad);
                    if (bad) *flag = 1;
                }
            }
        }
    }
}

// Whole-bank materialisation through the launch's voice lists (see LaunchSet): grid = (groups of 4 tiles, 64-voice chunks);
// a wave owns one tile and walks the chunk's lean records with the loop of k_bank_render (the sample is rounded and
// stored instead of accumulated), then the general list through voice_block, then zero-fills the rows of silent voices.
// The lean arithmetic repeats the general code's order -- ((x * amplitude) + 0) * g0u.
// LEAN = false: only the general and the silent list (the lean records went through k_generate_lean_harm).
template <int FPL, bool LEAN, typename OutT = float>
__global__ __launch_bounds__(256, FPL >= 4 ? 4 : 6) void k_generate_lists(BankPtrs B, const shm::sc_pair* __restrict__ trig_g,
                                                                         uint32_t nvoices, LaunchSet cur, uint64_t start, uint32_t n,
                                                                         OutT* __restrict__ out32, size_t stride,
                                                                         SegTab tab = SegTab{0, {}, {}, {}}, uint32_t vsplit = 1,
                                                                         double scale = 0.0, int* __restrict__ flag = nullptr) {
    bool bad = false;                          // (int16 rows: a sample that does not fit)
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
            // the LFO's table piece (FastRec: the launch's first in poly[3..9], the next one in poly[11..15], pad2 for the tiles from
            // frame pad1 on -- a multiple of 1024, so of this kernel's tile: prepare_voice)
            {
                const bool second = tile0 >= q->pad1;
                const double l_a = second ? poly[11] : poly[3], l_d = second ? poly[12] : poly[4], l_K = second ? poly[13] : poly[5];
                const double l_C = second ? poly[14] : poly[6], l_rc = second ? poly[15] : poly[8], l_rs = second ? q->pad2 : poly[9];
                shm::sincos_tab(fma(di[0], l_d, l_a), trig, ls, lc);
#pragma unroll
                for (int j = 0; j < FPL; ++j) {
                    if (j > 0) {
                        const double ns = fma(ls, l_rc, lc * l_rs), nc = fma(lc, l_rc, -(ls * l_rs));
                        ls = ns;
                        lc = nc;
                    }
                    const double Ln = fma(l_K, l_C - lc, poly[7] * (poly[10] + di[j]));
                    th[j] = poly[0] * T[j] + fma(poly[2], Ln, poly[1]);
                }
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
            if (RowOut<OutT>::I16 && kind == LEAN_SINE) {
                // Where int16 parity hangs on ONE ulp: a Sine whose frequency divides the sample rate has samples on its peaks; the
                // reference's accumulated t lies some delta (1e-13 early in a note, 1e-6 minutes in) beside pi/2 + 2 pi k there and
                // math.sin returns the correctly rounded 1 - delta^2 / 2: exactly 1.0 while delta <= 1.05e-8, an ulp or two less just
                // beyond.  With amplitude x scale an integer -- the oscillators' default amplitude 1.0 at scale 32767 -- the sample lies
                // ON a truncation boundary: 32767 if the sine is 1.0, 32766 if it is 1 - 2^-53.  A sine rotated from its neighbour is an
                // ulp off either way about half the time.  But the cosine beside it IS delta (to 1e-16 absolute, 1e-8 of itself), so
                // next to a peak the sine is taken from it: +-fma(-c / 2, c, 1), ONE rounding of 1 - c^2 / 2 (c^4 / 8 < 5e-22 inside
                // |c| < 2^-17; beyond, the sample is 1e-6 of a step away from the boundary and an ulp cannot carry it across).
                // (The table lookup of a lane's first frame, and of every frame of the per-oscillator path, does the same by
                // construction: its node at pi/2 is (1, 6e-17) and the residual enters as 1 - r^2 / 2.)
#pragma unroll
                for (int j = 0; j < FPL; ++j) sn[j] = fabs(cs[j]) < 0x1p-17 ? copysign(fma(-0.5 * cs[j], cs[j], 1.0), sn[j]) : sn[j];
            }
#pragma unroll
            for (int j = 0; j < FPL; ++j) x[j] = kind == LEAN_SINE ? sn[j] : pv[j] * sn[j];
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) x[j] = (x[j] * amp + 0.0) * g0u;
        if constexpr (RowOut<OutT>::I16) {
            // the int16 boundary guard of a polynomial-Harmonics record (FastRec::pad1: the launch's tolerance above the list's length)
            const uint32_t gword = q->pad1;
            if (kind == LEAN_HARM && (gword & 0xFFu) != 0) {
                const double tq = fabs(scale) * (double)__uint_as_float(gword & ~0xFFu);
                if (guard_near<FPL>(x, scale, tq)) {
                    const VoiceRegs r = load_record(as_const(cur.launch) + vi);
                    voice_block_exact<FPL>(r, cur.fm + vi, B, B.voices + vi, tile0, tile_last, i, di, nullptr, nullptr, trig, x);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const uint32_t raw = tile0 + j * 64 + lane;
            if (raw < n) out32[(size_t)vi * stride + raw] = RowOut<OutT>::make(x[j], scale, bad);
        }
    }
    const uint32_t SH_CONST_AS* idx = as_const(cur.gen_idx) + c * 64;
    for (uint32_t p = vsub; p < ngen; p += vsplit) {
        const uint32_t vi = idx[p];
        const VoiceRegs r = load_record(as_const(cur.launch) + vi);
        double x[FPL];
        voice_block_at<FPL, false>(r, cur.fm + vi, B, B.voices + vi, tile0, tile_last, i, di, nullptr, nullptr, trig, x);
        if constexpr (RowOut<OutT>::I16) {
            const double tq = guard_tq(r, B.voices + vi, tile_last, scale);
            if (tq != 0.0 && guard_near<FPL>(x, scale, tq))
                voice_block_exact<FPL>(r, cur.fm + vi, B, B.voices + vi, tile0, tile_last, i, di, nullptr, nullptr, trig, x);
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const uint32_t raw = tile0 + j * 64 + lane;
            if (raw < n) out32[(size_t)vi * stride + raw] = RowOut<OutT>::make(x[j], scale, bad);
        }
    }
    for (uint32_t p = vsub; p < nsilent; p += vsplit) {
        const uint32_t vi = idx[63 - p];
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const uint32_t raw = tile0 + j * 64 + lane;
            if (raw < n) out32[(size_t)vi * stride + raw] = (OutT)0;
        }
    }
    if (RowOut<OutT>::I16 && bad) *flag = 1;
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
//
// OutT = short (sh_bank_generate_i16): the sample leaves as int(scale * v), two bytes.  A lane's frames lie 64 apart, so a lane's own
// pair of consecutive values (frames h, h + 1 of its sixteen) are NOT neighbours in the row; lanes 2p and 2p + 1 hold the four frames
// (2p, 2p + 1) + 64 h and (2p, 2p + 1) + 64 (h + 1).  They swap one packed pair (a DPP quad_perm within the register file, no LDS), a
// v_perm_b32 with a per-lane selector picks the two int16 that are neighbours in memory, and every lane stores ONE 32-bit word: the
// even lanes write the 128 bytes of frames [64 h, 64 h + 64), the odd lanes the 128 bytes behind them -- 256 contiguous bytes per
// store instruction, as in the float32 form, for two frames per lane instead of one.  Per pair of samples: two products, two
// conversions, the range check (min3 / max3), pack, swap, select -- against the 42 float64 operations that made the two samples.
//
// FOLD (sh_bank_mixdown_i16): the int16 samples are not stored at all -- they enter the reference mixer's chain, mixed =
// audioop.add(mixed, voice, 2) down the voices in order, where they are made.  A saturating chain over a RANGE of voices is the map
// x -> clamp(x + a, L, U) (a: the exact sum, L / U: what the rails have done to the bounds), and such maps compose in voice order:
// a wave keeps (a, L, U) for its sixteen frames per lane in registers (a += s; L = add_sat(L, s); U = add_sat(U, s): one packed
// instruction per two frames for each bound), walks its share of a chunk's records, and leaves ONE (a, L | U) pair per frame --
// 8 bytes -- in plane blockIdx.y of `parts`; k_mixdown_combine applies the planes in order.  Where the int16 rows and the chain
// kernel move 2 B + 2 B per voice-sample through HBM this moves 16 B per frame and plane (32 planes for 1024 voices: 0.5 B per
// voice-sample) -- and the launch is bound by the same float64 arithmetic as the materialisation, so the chain pass disappears.
#ifndef SH_GEN_MINW
#define SH_GEN_MINW 4              // wavefronts per SIMD the compiler budgets registers for (tools/ab.py build NAME -DSH_GEN_MINW=3: the A/B of round 6)
#endif
template <int FPL, typename OutT = float, bool FOLD = false, bool GUARD = true>
__global__ __launch_bounds__(256, (FOLD || (RowOut<OutT>::I16 && FPL == 16)) ? 3 : SH_GEN_MINW) void k_generate_lean_harm(const shm::sc_pair* __restrict__ trig_g, LaunchSet base, uint32_t nvoices,
                                                               uint32_t total, uint32_t seg_frames,
                                                               OutT* __restrict__ out32_all, size_t stride, SegTab tab, uint32_t rec_split,
                                                               double scale = 0.0, int* __restrict__ flag = nullptr,
                                                               int2v* __restrict__ parts = nullptr, size_t plane = 0) {
    constexpr bool I16 = RowOut<OutT>::I16;
    static_assert(!FOLD || I16, "the fold is over int16 samples");
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
    OutT* __restrict__ out32 = out32_all + seg_first;
    uint32_t tile_last = tile0 + 64 * FPL - 1;
    if (tile_last > n - 1) tile_last = n - 1;
    // (rec_split workgroups share a chunk's records: a wave that walks all 64 of them over 1024 frames lives a third of the launch --
    // 7552 such waves on 5120 wave slots are a full round and a half-empty one)
    const uint32_t c = blockIdx.y / rec_split, part = blockIdx.y % rec_split;
    const uint32_t nfast_all = as_const(cur.counts)[4 * c];
    const uint32_t p_lo = nfast_all * part / rec_split, nfast = nfast_all * (part + 1) / rec_split;
    const FastRec SH_CONST_AS* q = as_const(cur.fast) + c * 64 + p_lo;
    const uint32_t i0 = tile0 + lane;
    const double di0 = (double)i0;
    OutT* __restrict__ col = out32 + i0;                       // + vi * stride per record (uniform), + 64 * j per frame
    // (int16 rows) the lane's word of a pair of frames: 32-bit index 64 (h / 2) + 32 (lane & 1) + (lane >> 1) from the tile's first frame
    const uint32_t pair_word = ((lane & 1u) << 5) + (lane >> 1);
    const uint32_t pair_sel = (lane & 1u) ? 0x07060302u : 0x01000504u;      // v_perm_b32(own, neighbour's): odd lanes (n.hi, own.hi), even (own.lo, n.lo)
    int mx = 0, mn = 0;                                         // the largest / smallest integer any sample of this wave became
    // (FOLD) the chain over this wave's records as x -> clamp(x + fa, fL, fU), per frame of the lane: the identity on int16 to begin with
    typedef short short2p __attribute__((ext_vector_type(2)));
    int fa[FOLD ? FPL : 1];
    short2p fL[FOLD ? FPL / 2 : 1], fU[FOLD ? FPL / 2 : 1];
    if constexpr (FOLD) {
#pragma unroll
        for (int j = 0; j < FPL; ++j) fa[j] = 0;
#pragma unroll
        for (int m = 0; m < FPL / 2; ++m) { fL[m] = (short2p){-32768, -32768}; fU[m] = (short2p){32767, 32767}; }
    }
    // ---- int16 (rows and FOLD): a record's sixteen samples are quantised into registers first -- w[m] = frames 2m, 2m + 1 of the lane,
    // packed -- checked against the boundary guard (below), and only then stored / folded.
    constexpr int NQ = I16 ? (FPL / 4) : 1;                     // a record's frames in quarters of four per lane: what the guard redoes at a time
    short2p w[I16 ? FPL / 2 : 1];
    int qmx[NQ], qmn[NQ];                                       // largest / smallest integer of each quarter's (valid) samples
    uint64_t nearm[NQ];                                         // (uniform) lanes with a sample of quarter h within the guard's reach of an integer
    uint32_t nearv[NQ];                                         // the smallest frac(scale v + tq) * 2^32 among the lane's frames of quarter h
    double tq = 0.0;                                            // (uniform) the record's guard distance in integer units
    double tqm = 0.0;                                           // (uniform) tq + 1.5 * 2^20
    uint32_t near_lo = 0;                                       // (uniform) 2 tq in units of 2^-32, plus two; 0xFFFFFFFF: every sample (a record that holds a NaN)
    const bool full_tile = tile0 + 64 * FPL <= n;
    // frames 2m, 2m + 1 of the lane: int(scale * v) -- as int(scale v + tq): the same integer unless scale v lies within tq of one, which
    // is what frac(scale v + tq) <= 2 tq says; those samples are redone below (the boundary guard)
    auto quant_pair = [&](int m, double v0, double v1, bool ok0, bool ok1) {
        if constexpr (I16) {
            const double s0q = fma(scale, v0, tq), s1q = fma(scale, v1, tq);
            const int a = (int)s0q, b = (int)s1q;               // float64 product, truncation toward zero
            if constexpr (GUARD) {       // (GUARD = false: a bank built without guard lists -- params.int16_guard = False -- quantises as rounds 1-5 did)
            // The fraction of scale v + tq as an integer: added to 1.5 * 2^20 (an ulp there is 2^-32) the sum's LOW WORD is
            // frac(scale v + tq) * 2^32 -- one FMA per sample, no v_fract_f64 (a quarter-rate instruction: with it the check cost 13-17 %
            // of the kernel, a compare per frame into a scalar register pair 18 %; profiles/r06_guard_ab.txt).  The SMALLEST low word of
            // the quarter's samples is kept -- one v_min3_u32 per pair -- and compared once per record with 2 tq * 2^32 (+ 2: rounding).
            // ... except around the integer 0, which is no boundary of a truncation toward zero -- and a waveform that is FLAT at its zero
            // crossing (the 1/k series of an even number of partials at t = pi: value, slope and curvature vanish) spends one sample in
            // 10^4 within reach of it: 2 % of the quarters 5 s into the benchmark's notes, 5.5 % after 300 s, against 10^-7 of the samples in
            // reach of a real boundary (tools/guard_count.py).  floor(scale v + tq) = 0 is the high word 0x41380000 of the same sum.
            const double w0 = fma(scale, v0, tqm), w1 = fma(scale, v1, tqm);
            uint32_t h0 = (uint32_t)__double2hiint(w0), h1 = (uint32_t)__double2hiint(w1);
            // (the high words as opaque 32-bit values: left to itself the compiler compares the masked 64-bit patterns -- v_mov, v_cmp_eq_u64,
            //  v_cmp_ne_u64 and two selects per sample, 76 instructions per record instead of 40)
            asm volatile("" : "+v"(h0), "+v"(h1));
            uint32_t k0 = h0 == 0x41380000u ? 0xFFFFFFFFu : (uint32_t)__double2loint(w0);
            uint32_t k1 = h1 == 0x41380000u ? 0xFFFFFFFFu : (uint32_t)__double2loint(w1);
            // (the running minimum -- and the maxima below -- as ONE three-operand instruction per pair, v_min3_u32 / v_max3_i32 / v_min3_i32:
            //  left to itself the compiler balances the trees into a two-operand instruction per pair and a three-operand one per quarter --
            //  twelve instructions of a record's ninety-two beside its float64 ones; rows 0.528 -> 0.522 ms, fused mixdown 0.492 -> 0.481)
            asm volatile("" : "+v"(k0), "+v"(k1));
            uint32_t nv_ = min(min(nearv[m / 2], k0), k1);
            asm volatile("" : "+v"(nv_));
            nearv[m / 2] = nv_;
            }
            int mx_ = max(max(qmx[m / 2], ok0 ? a : 0), ok1 ? b : 0), mn_ = min(min(qmn[m / 2], ok0 ? a : 0), ok1 ? b : 0);
            asm volatile("" : "+v"(mx_), "+v"(mn_));
            qmx[m / 2] = mx_;
            qmn[m / 2] = mn_;
            w[m] = __builtin_amdgcn_cvt_pk_i16(a, b);
        }
    };
    for (uint32_t p = p_lo; p < nfast; ++p, ++q) {
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
        OutT* __restrict__ row = col + (size_t)vi * stride;
        OutT* __restrict__ row_tile = row - lane;                  // (the tile's first frame in this record's row: 4-byte aligned, stride is even)
        // amplitude and (constant) envelope gain scale the sine ONCE: the recurrence is linear, so every later frame's sine
        // arrives scaled and the sample is one product, p * s (differs from the general code's ((x * amplitude) + 0) * g0u
        // by float64 rounding only)
        const double ag = amp * g0u;
        s0 *= ag;
        s1 *= ag;
        uint32_t gword = 0;
        if constexpr (I16) {
            gword = q->pad1;                                       // the boundary guard: tolerance (float32, 24 bits, rounded up) | length of the list
            const double tol = fabs(scale) * (double)__uint_as_float(gword & ~0xFFu);
            if constexpr (!GUARD) {
                tq = 0.0;                                          // int(scale * v) as it is
                near_lo = 0;
            } else if (tol < 0.125) {
                tq = tol >= 0x1p-31 ? tol : 0x1p-31;               // (at least two steps of the 2^-32 grid the check works on)
                near_lo = (uint32_t)((tq + tq) * 0x1p32) + 2u;
            } else {
                // a record with a non-finite value carries an infinite tolerance (prepare_chunk), and a tolerance of 1/8 or more cannot be
                // told from one: every sample is redone, and what is not a number is flagged there
                tq = 0.0;
                near_lo = 0xFFFFFFFFu;
            }
            tqm = tq + 0x1.8p20;
#ifdef SH_AB_COUNT
            (void)0;
#endif
#pragma unroll
            for (int h = 0; h < NQ; ++h) { qmx[h] = 0; qmn[h] = 0; nearv[h] = 0xFFFFFFFFu; }
        }
        auto frames = [&](auto full_tile_t) {
            constexpr bool FULL = decltype(full_tile_t)::value;
#pragma unroll
            for (int h = 0; h < FPL; h += 2) {
                double p0 = fma(poly[0], c0, poly[1]), p1 = fma(poly[0], c1, poly[1]);
#pragma unroll
                for (int u = 2; u < 16; ++u) {
                    p0 = fma(p0, c0, poly[u]);
                    p1 = fma(p1, c1, poly[u]);
                }
                // streaming stores: the rows are not read again by this kernel (1.97 GB per launch of the benchmark shape)
                if constexpr (!I16) {
                    if (FULL || i0 + (uint32_t)h * 64u < n) __builtin_nontemporal_store((float)(p0 * s0), row + h * 64);
                    if (FULL || i0 + (uint32_t)(h + 1) * 64u < n) __builtin_nontemporal_store((float)(p1 * s1), row + (h + 1) * 64);
                } else {                                           // (the last tile of a row: frames behind its end are quantised too, never range-checked or written)
                    quant_pair(h / 2, p0 * s0, p1 * s1, FULL || i0 + (uint32_t)h * 64u < n, FULL || i0 + (uint32_t)(h + 1) * 64u < n);
                }
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
        if (!straddle && full_tile && FPL % 4 == 0) {
            // the common case (a whole tile on one phase-table piece) as straight-line code (round 4): four Horner chains at a time, no
            // branch between the frames -- the general form below asks `straddle` after every pair of frames, eight uniform branches
            // per record that end a basic block each.  0.476 -> 0.458 ms per 1024 x 480 000 samples (profiles/r04_generate_ab.txt); at five
            // waves per SIMD (94 registers, 36 bytes of scratch) 0.466
            double sa = s0, sb = s1, ca = c0, cb = c1;
#pragma unroll
            for (int h = 0; h < FPL; h += 4) {
                double sv[4], cv[4];
                if (h == 0) {
                    sv[0] = sa; cv[0] = ca; sv[1] = sb; cv[1] = cb;
                } else {
                    sv[0] = fma(k2, sb, -sa); cv[0] = fma(k2, cb, -ca);
                    sv[1] = fma(k2, sv[0], -sb); cv[1] = fma(k2, cv[0], -cb);
                }
                sv[2] = fma(k2, sv[1], -sv[0]); cv[2] = fma(k2, cv[1], -cv[0]);
                sv[3] = fma(k2, sv[2], -sv[1]); cv[3] = fma(k2, cv[2], -cv[1]);
                sa = sv[2]; sb = sv[3]; ca = cv[2]; cb = cv[3];
                double pv[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) pv[jj] = fma(poly[0], cv[jj], poly[1]);
#pragma unroll
                for (int u = 2; u < 16; ++u) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) pv[jj] = fma(pv[jj], cv[jj], poly[u]);
                }
                if constexpr (I16) {
                    quant_pair(h / 2, pv[0] * sv[0], pv[1] * sv[1], true, true);
                    quant_pair(h / 2 + 1, pv[2] * sv[2], pv[3] * sv[3], true, true);
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) __builtin_nontemporal_store((float)(pv[jj] * sv[jj]), row + (h + jj) * 64);
                }
            }
        } else if (full_tile) {
            frames(std::true_type());
        } else {
            frames(std::false_type());
        }
        if constexpr (I16) {
            // ---- the boundary guard (sh_voice::guard_*): the polynomial form is sum a_k sin(k t) of the EXACT products k t, the reference
            // rounds every t * k before its sine -- up to guard_t |t| + guard_c apart, and where scale * v lies that close to an integer the two
            // truncate differently.  Such a sample (one record in ~10^4 of the benchmark's voices ten seconds into their notes) is summed
            // term by term from the voice's own list, like the reference's loop: a quarter of the record (four frames per lane) at a time,
            // every frame from a table lookup of its own, the lanes in reach of an integer through the list. ----
            uint64_t any_near = 0;
            if constexpr (GUARD) {
#pragma unroll
                for (int h = 0; h < NQ; ++h) {
                    nearm[h] = __builtin_amdgcn_ballot_w64(nearv[h] <= near_lo);
                    any_near |= nearm[h];
                }
            } else if ((gword & 0x7F800000u) == 0x7F800000u) {
                mx = 0x7FFFFFFF;                                  // a record that holds a NaN or an infinity (prepare_chunk): OverflowError, as on the general path
            }
            if (GUARD && any_near != 0) {
                const uint32_t glen = gword & 0xFFu;
                const sh_partial SH_CONST_AS* gl = reinterpret_cast<const sh_partial SH_CONST_AS*>((uintptr_t)__double_as_longlong(q->pad2));
#pragma unroll 1
                for (int h = 0; h < NQ; ++h) {
                    uint64_t nm = 0;
#pragma unroll
                    for (int hh = 0; hh < NQ; ++hh) nm = h == hh ? nearm[hh] : nm;
                    if (nm == 0) continue;
#ifdef SH_AB_COUNT             // (tools/ab.py build count -DSH_AB_COUNT: how often the careful path runs -- words 4.. of the flag block)
                    if (lane == 0) { atomicAdd(flag + 4, 1); atomicAdd(flag + 5, (int)__popcll(nm)); }
#endif
                    int aa[4], lo = 0, hi = 0;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = 4 * h + jj;
                        const double t = theta(j);
                        double sn, cs;
                        shm::sincos_tab(t, trig, sn, cs);
                        double pv = fma(poly[0], cs, poly[1]);
#pragma unroll
                        for (int u = 2; u < 16; ++u) pv = fma(pv, cs, poly[u]);
                        const double sq = fma(scale, pv * (sn * ag), tq);
                        // (an integer in reach -- but not 0: truncation toward zero has no boundary there, and a waveform that is FLAT at
                        //  its zero crossing, like the 1/k series of an even number of partials at t = pi, spends one sample in 10^4 within
                        //  reach of it: 2-5 % of the quarters come here for that alone, tools/guard_count.py)
                        const double wq = fma(scale, pv * (sn * ag), tqm);
                        const bool nr = ((uint32_t)__double2loint(wq) <= near_lo && __double2hiint(wq) != 0x41380000) || !(sq == sq);
                        int a = (int)sq;
                        if (__ballot(nr) != 0ull) {               // (uniform) some lane's frame j is in reach of an integer
                            if (nr) {
                                double hs = 0.0;                   // the reference's loop: sum of sin(fl(t * k)) * a_k in list order
                                for (uint32_t k = 0; k < glen; ++k) {
                                    double ks, kc;
                                    shm::sincos_tab(t * gl[k].k, trig, ks, kc);
                                    hs += ks * gl[k].amp;
                                }
                                const double ex = glen ? scale * (((hs * amp) + 0.0) * g0u) : scale * (pv * (sn * ag));
                                a = (ex >= -2147483000.0 && ex <= 2147483000.0) ? (int)ex : 0x7FFFFFFF;     // (NaN, like anything beyond int16: OverflowError)
                            }
                        }
#ifdef SH_AB_COUNT
                        if (nr && atomicCAS(flag + 14, 0, 1) == 0) {      // the first sample the careful path finds near: what does it look like?
                            const double vv = pv * (sn * ag), ww = fma(scale, vv, tqm);
                            flag[7] = __double2loint(ww); flag[8] = __double2hiint(ww); flag[9] = __double2loint(vv); flag[10] = __double2hiint(vv);
                            flag[11] = (int)near_lo; flag[12] = (int)lane; flag[13] = j;
                        }
                        if (nr && lane == (uint32_t)__ffsll((long long)__ballot(nr)) - 1u) atomicAdd(flag + 6, 1);
#endif
                        aa[jj] = a;
                        const bool ok = full_tile || i0 + (uint32_t)j * 64u < n;
                        hi = max(hi, ok ? a : 0);
                        lo = min(lo, ok ? a : 0);
                    }
                    const short2p w0 = __builtin_amdgcn_cvt_pk_i16(aa[0], aa[1]), w1 = __builtin_amdgcn_cvt_pk_i16(aa[2], aa[3]);
#pragma unroll
                    for (int hh = 0; hh < NQ; ++hh) {
                        if (h == hh) { w[2 * hh] = w0; w[2 * hh + 1] = w1; qmx[hh] = hi; qmn[hh] = lo; }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < NQ; ++h) { mx = max(mx, qmx[h]); mn = min(mn, qmn[h]); }
            // ---- the record's samples leave: folded into the chain (FOLD), or as 256 contiguous bytes per store instruction -- lanes 2p and
            // 2p + 1 swap one packed pair (DPP) and a v_perm_b32 picks the two int16 that are neighbours in memory -- or, on the last tile
            // of a row, frame by frame
#pragma unroll
            for (int m = 0; m < FPL / 2; ++m) {
                if constexpr (FOLD) {
                    fa[2 * m] += (int)w[m].x;
                    fa[2 * m + 1] += (int)w[m].y;
                    fL[m] = __builtin_elementwise_add_sat(fL[m], w[m]);
                    fU[m] = __builtin_elementwise_add_sat(fU[m], w[m]);
                } else if (full_tile) {
                    union { short2p v; uint32_t u; } ww;
                    ww.v = w[m];
                    const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)ww.u, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
                    __builtin_nontemporal_store((int)__builtin_amdgcn_perm(ww.u, nb, pair_sel), reinterpret_cast<int*>(row_tile) + 64 * m + pair_word);
                } else {
                    if (i0 + (uint32_t)(2 * m) * 64u < n) row[2 * m * 64] = w[m].x;
                    if (i0 + (uint32_t)(2 * m + 1) * 64u < n) row[(2 * m + 1) * 64] = w[m].y;
                }
            }
        }
    }
    if (I16 && (mx > 32767 || mn < -32768)) *flag = 1;
    if constexpr (FOLD) {
        // this wave's map for its tile, into the plane of its (chunk, part): frame f of the launch at parts[plane_index * plane + f]
        int2v* __restrict__ dst = parts + (size_t)blockIdx.y * plane + seg_first + i0;
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            if (i0 + (uint32_t)j * 64u < n) {
                const uint32_t lu = (uint32_t)(uint16_t)fL[j / 2][j & 1] | ((uint32_t)(uint16_t)fU[j / 2][j & 1] << 16);
                __builtin_nontemporal_store((int2v){fa[j], (int)lu}, dst + j * 64);
            }
        }
    }
}

// The planes of k_generate_lean_harm<.., FOLD> applied in order (plane = 64-voice chunk x part: voice order): out[f] = the chain's result.
__global__ __launch_bounds__(256) void k_mixdown_combine(const int2v* __restrict__ parts, uint32_t nplanes, size_t plane, uint32_t first, uint32_t n,
                                                         short* __restrict__ out) {
    const uint32_t f = (uint32_t)(sh::block_id() * 256 + threadIdx.x);
    if (f >= n) return;
    const int2v* __restrict__ p = parts + first + f;
    int x = 0;
    uint32_t k = 0;
    for (; k + 4 <= nplanes; k += 4) {                          // four planes in flight
        int2v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)(k + u) * plane);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int L = (int)(short)(uint16_t)((uint32_t)v[u].y & 0xFFFFu), U = (int)(short)(uint16_t)((uint32_t)v[u].y >> 16);
            x = min(max(x + v[u].x, L), U);
        }
    }
    for (; k < nplanes; ++k) {
        const int2v v = p[(size_t)k * plane];
        const int L = (int)(short)(uint16_t)((uint32_t)v.y & 0xFFFFu), U = (int)(short)(uint16_t)((uint32_t)v.y >> 16);
        x = min(max(x + v.x, L), U);
    }
    out[first + f] = (short)x;
}

// (Measured and dropped in round 3: the same materialisation walked ROW by row -- one record per workgroup held in SGPRs, contiguous runs
// of a row per workgroup: 0.515 of HBM against 0.52; CHANGELOG item 38.)

}  // namespace

extern "C" {

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
    int rc = prepare_single(bank, voice, 1, start, n);
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

}  // extern "C"

namespace {

// every voice of the bank as a row of OutT (float: the sample rounded to float32; short: int(scale * sample), *flag set where one does not fit)
template <typename OutT>
int generate_rows(sh_bank* b, uint64_t start, uint32_t nframes, OutT* o, size_t stride, double scale, int* flag) {
    int rc;
    // frames per lane: 4 for long rows (one sin/cos lookup + three rotations per voice, as in k_bank_render), else 2 / 1
    const int fpl = nframes >= 8192 ? 4 : (nframes >= 2048 ? 2 : 1);
    const bool with_rows = b->launch_rows != nullptr;             // sh_bank_generate_rows: every voice through the general kernel, which reads the rows
    if (!with_rows && fpl == 4 && b->lean_candidates != 0 && b->lean_fm_candidates == 0) {
        // Long rows, every lean candidate a polynomial Harmonics voice: the lean records by the recurrence kernel at sixteen frames
        // per lane, then the general and silent lists -- unless the segment provably has none.  Rows longer than a segment get
        // one record set per segment, all resolved by ONE prepare launch.
        // frames per lane of the lean kernel: sixteen on long rows (1024 x 480 000: 496 us, eight: 508), fewer when that leaves
        // the chip short of workgroups (1024 x 48 000 at sixteen: 12 x 16 = 192 workgroups of four 1024-frame tiles)
        int lf = 16;
        while (lf > 4 && (uint64_t)sh::div_up(nframes, 256 * lf) * sh::div_up(b->nvoices, 64) < 512) lf /= 2;
        const int LF = lf;
        // workgroups per chunk of records: enough waves for several rounds of the chip's wave slots (1, 4, 8 measured: CHANGELOG item 38)
        const uint32_t rsplit = 2;
        // (the int16 kernels in two forms: with the boundary guard's check, and -- a bank none of whose voices carries a guard list -- without)
        constexpr bool I16_ = RowOut<OutT>::I16;
        const bool guard = !I16_ || b->has_guard;
#define SH_GEN_LEAN(GRID_, ...) do { \
            if (!I16_ || guard) { \
                if (LF == 16) hipLaunchKernelGGL((k_generate_lean_harm<16, OutT>), GRID_, dim3(256), 0, st, __VA_ARGS__, scale, flag); \
                else if (LF == 8) hipLaunchKernelGGL((k_generate_lean_harm<8, OutT>), GRID_, dim3(256), 0, st, __VA_ARGS__, scale, flag); \
                else hipLaunchKernelGGL((k_generate_lean_harm<4, OutT>), GRID_, dim3(256), 0, st, __VA_ARGS__, scale, flag); \
            } else { \
                if (LF == 16) hipLaunchKernelGGL((k_generate_lean_harm<16, OutT, false, !I16_>), GRID_, dim3(256), 0, st, __VA_ARGS__, scale, flag); \
                else if (LF == 8) hipLaunchKernelGGL((k_generate_lean_harm<8, OutT, false, !I16_>), GRID_, dim3(256), 0, st, __VA_ARGS__, scale, flag); \
                else hipLaunchKernelGGL((k_generate_lean_harm<4, OutT, false, !I16_>), GRID_, dim3(256), 0, st, __VA_ARGS__, scale, flag); \
            } } while (0)
        constexpr uint32_t SEG = 65536;                      // frames per segment (a multiple of the 1024-frame tile)
        hipStream_t st = sh::state().stream;
        const uint32_t nchunks = sh::div_up(b->nvoices, 64);
        // Rows that start with the notes (attack, decay, a dozen binades of the phase sum in the first 65 536 frames): the head
        // is cut like a transition launch of the render path (plan_segments: where the envelopes are flat, then doubling
        // positions) so that its voices stay lean -- one prepare launch, one lean launch, ONE lists launch over all segments with
        // the general voices of a chunk dealt to eight workgroups; the rest of the row follows with equal segments.
        const bool no_seg = sh::knobs().no_seg;
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
                rc = grow_segment_sets(b->gen_block, b->gen_set, b->gen_segs, tab.n, b->nvoices);
                if (rc) return rc;
                const LaunchSet base = b->gen_set;
                BankPtrs P = ptrs(b);
                P.nseg = tab.n;
                uint32_t tiles = 0, list_groups = 0;
                for (uint32_t k = 0; k <= tab.n; ++k) P.seg_first[k] = cut[k];
                for (uint32_t k = 0; k < tab.n; ++k) tiles += sh::div_up(tab.len[k], 64 * LF);
                for (uint32_t k = 0; k < ltab.n; ++k) list_groups += sh::div_up(ltab.len[k], 4 * 64 * 4);
                rc = launch_prepare_segments_var(st, false, P, base, b->nvoices, tab.n, start);
                if (rc) return rc;
                SH_GEN_LEAN(dim3(sh::div_up(tiles, 4), nchunks * rsplit), trig_table(), base, b->nvoices, head, SEG, o, stride, tab, rsplit);
                SH_CHECK_LAUNCH("k_generate_lean_harm");
                constexpr uint32_t VSPLIT = 8;
                if (ltab.n) {
                    hipLaunchKernelGGL((k_generate_lists<4, false, OutT>), dim3(list_groups, nchunks * VSPLIT), dim3(256), 0, st,
                                       ptrs(b), trig_table(), b->nvoices, base, start, head, o, stride, ltab, VSPLIT, scale, flag);
                    SH_CHECK_LAUNCH("k_generate_lists");
                }
                if (head == nframes) return SH_OK;
                return generate_rows<OutT>(b, start + head, nframes - head, o + head, stride, scale, flag);
            }
        }
        const uint32_t nseg = sh::div_up(nframes, SEG);
        LaunchSet base;
        if (nseg == 1) {
            rc = acquire_records(b, start, nframes, st, false);
            if (rc) return rc;
            base = launch_set(b, b->cur);
        } else {
            rc = grow_segment_sets(b->gen_block, b->gen_set, b->gen_segs, nseg, b->nvoices);
            if (rc) return rc;
            base = b->gen_set;
            rc = launch_prepare_segments(st, ptrs(b), base, b->nvoices, nseg, start, nframes, SEG);
            if (rc) return rc;
        }
        SegTab none;
        none.n = 0;
        SH_GEN_LEAN(dim3(sh::div_up(nframes, 256 * LF), nchunks * rsplit), trig_table(), base, b->nvoices, nframes,
                    nseg == 1 ? (nframes + 1023u) / 1024u * 1024u : SEG, o, stride, none, rsplit);
#undef SH_GEN_LEAN
        SH_CHECK_LAUNCH("k_generate_lean_harm");
        for (uint32_t sg = 0; sg < nseg; ++sg) {
            const uint32_t first = sg * SEG, n = nframes - first < SEG ? nframes - first : SEG;
            if (b->no_general_voice(start + first, n)) continue;
            LaunchSet cur = base;
            if (nseg > 1) {
                const size_t stride = set_slots(b->nvoices);          // (segment_set's layout)
                cur.launch += (size_t)sg * stride; cur.fm += (size_t)sg * stride; cur.fast += (size_t)sg * stride;
                cur.gen_idx += (size_t)sg * stride; cur.counts += (size_t)sg * 4 * nchunks;
            }
            SegTab no_tab;
            no_tab.n = 0;
            hipLaunchKernelGGL((k_generate_lists<4, false, OutT>), dim3(sh::div_up(n, 1024), nchunks), dim3(256), 0, st,
                               ptrs(b), trig_table(), b->nvoices, cur, start + first, n, o + first, stride, no_tab, 1u, scale, flag);
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
    float* o32 = nullptr;
    short* o16 = nullptr;
    if constexpr (RowOut<OutT>::I16) o16 = o; else o32 = o;
#define SH_GEN(F_) hipLaunchKernelGGL(k_generate<F_>, dim3(tile_groups, groups), dim3(256), 0, sh::state().stream,            \
                                      ptrs(b), trig_table(), 0u, b->nvoices, vpg, b->d_launch, b->d_launch_fm, start, nframes, \
                                      (const double*)nullptr, (const double*)nullptr, o32, (double*)nullptr, stride, o16, scale, flag)
    if (!with_rows && fpl == 4 && b->lean_candidates != 0) {
        // long rows of a bank with lean candidates: one workgroup column per 64-voice chunk, walking the launch's lists
        SegTab no_tab;
        no_tab.n = 0;
        hipLaunchKernelGGL((k_generate_lists<4, true, OutT>), dim3(tile_groups, sh::div_up(b->nvoices, 64)), dim3(256), 0, sh::state().stream,
                           ptrs(b), trig_table(), b->nvoices, launch_set(b, b->cur), start, nframes, o, stride, no_tab, 1u, scale, flag);
    } else if (fpl == 4) SH_GEN(4); else if (fpl == 2) SH_GEN(2); else SH_GEN(1);
#undef SH_GEN
    SH_CHECK_LAUNCH("k_generate");
    return SH_OK;
}

int generate_check(const char* who, sh_bank* b, uint32_t nframes, const sh_buf* voices_out, size_t stride, size_t elem) {
    if (!b || !voices_out) return sh::set_error(SH_ERR_INVALID, "%s: NULL argument", who);
    if (nframes > 0xFFFF0000u) return sh::set_error(SH_ERR_INVALID, "%s: at most 2^32 - 65536 frames per call", who);
    if (stride < nframes) return sh::set_error(SH_ERR_INVALID, "%s: stride < nframes", who);
    if (nframes && voices_out->bytes / elem < (size_t)(b->nvoices - 1) * stride + nframes)
        return sh::set_error(SH_ERR_INVALID, "%s: output buffer too small", who);
    return bank_check_plain(b, who);
}

}  // namespace

extern "C" {

int sh_bank_generate(sh_bank* b, uint64_t start, uint32_t nframes, sh_buf* voices_out, size_t stride) {
    SH_REQUIRE_INIT();
    int rc = generate_check("sh_bank_generate", b, nframes, voices_out, stride, 4);
    if (rc || nframes == 0) return rc;
    return generate_rows<float>(b, start, nframes, (float*)voices_out->ptr, stride, 0.0, nullptr);
}

int sh_bank_generate_i16_async(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* voices_out, size_t stride) {
    SH_REQUIRE_INIT();
    int rc = generate_check("sh_bank_generate_i16", b, nframes, voices_out, stride, 2);
    if (rc || nframes == 0) return rc;
    if ((stride & 1u) || ((uintptr_t)voices_out->ptr & 3u))
        return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_i16: stride must be even (rows are written as 32-bit pairs of samples)");
    if (sh::state().quantise_round)
        return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_i16: the fused quantiser truncates; under SH_OPT_QUANTISE_ROUND quantise float64 rows (sh_bank_generate_f64 + sh_quantize_f64)");
    return generate_rows<short>(b, start, nframes, (short*)voices_out->ptr, stride, scale, sh::state().flag + 1);     // (word 0: the quantisers of pcm.hip)
}

#ifdef SH_AB_COUNT
int sh_debug_flag_words(int* out, int n) {       // (diagnostic builds only: the flag block's words, then cleared)
    sh::State& st = sh::state();
    if (hipMemcpy(out, st.flag, sizeof(int) * (n < 16 ? n : 16), hipMemcpyDeviceToHost) != hipSuccess) return SH_ERR_HIP;
    (void)hipMemset(st.flag + 2, 0, sizeof(int) * 14);
    return SH_OK;
}
#endif

int sh_overflow_check(void) {
    SH_REQUIRE_INIT();
    sh::State& st = sh::state();
    SH_HIP(hipMemcpyAsync(st.flag_host + 1, st.flag + 1, sizeof(int), hipMemcpyDeviceToHost, st.stream));
    SH_HIP(hipStreamSynchronize(st.stream));
    if (st.flag_host[1]) {
        SH_HIP(hipMemsetAsync(st.flag + 1, 0, sizeof(int), st.stream));
        return sh::set_error(SH_ERR_OVERFLOW, "signed integer out of range for sample width 2");
    }
    return SH_OK;
}

namespace {
// A synchronous call that fails part-way (kernels already enqueued may have raised the flag) must not leave the process-wide word up
// for the next, unrelated call: lower it on the stream, behind those kernels.  (The word is shared with the _async calls: a synchronous
// call consumes whatever overflow is pending -- include/synthhip.h says so.)
int sync_call_failed(int rc) {
    sh::State& st = sh::state();
    if (st.flag) (void)hipMemsetAsync(st.flag + 1, 0, sizeof(int), st.stream);
    return rc;
}
}  // namespace

int sh_bank_generate_i16(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* voices_out, size_t stride) {
    SH_API_LOCK();
    const int rc = sh_bank_generate_i16_async(b, start, nframes, scale, voices_out, stride);
    if (rc) return sync_call_failed(rc);
    if (nframes == 0) return rc;
    return sh_overflow_check();
}

// ---- the reference's mono mixdown without the rows (see k_generate_lean_harm, FOLD) ---------------------------------------------------
namespace {

// frames [f0, f0 + len) of the call through int16 rows in a temporary and the chain kernel: where a voice needs the general code (the
// attack and decay of the notes, voices of other kinds), whose rows the fold would have to take in voice order between the lean ones
int mixdown_two_step(sh_bank* b, uint64_t start, uint32_t len, double scale, short* out, int* flag) {
    const size_t stride = ((size_t)len + 63) & ~(size_t)63;
    sh::Temp rows;
    int rc = rows.alloc((size_t)b->nvoices * stride * 2);
    if (rc) return rc;
    rc = generate_rows<short>(b, start, len, (short*)rows.buf.ptr, stride, scale, flag);
    if (rc) return rc;
    sh_buf o{out, (size_t)len * 2, false, 0};
    return sh_mix_chain_i16(&rows.buf, b->nvoices, stride, len, &o);
}

int mixdown_fused(sh_bank* b, uint64_t start, uint32_t len, double scale, short* out, int* flag) {
    constexpr uint32_t SEG = 65536;
    hipStream_t st = sh::state().stream;
    const uint32_t nchunks = sh::div_up(b->nvoices, 64), rsplit = 2, nplanes = nchunks * rsplit;
    int lf = 16;
    while (lf > 4 && (uint64_t)sh::div_up(len, 256 * lf) * nchunks < 512) lf /= 2;
    const uint32_t nseg = sh::div_up(len, SEG);
    LaunchSet base;
    int rc;
    if (nseg == 1) {
        rc = acquire_records(b, start, len, st, false);
        if (rc) return rc;
        base = launch_set(b, b->cur);
    } else {
        rc = grow_segment_sets(b->gen_block, b->gen_set, b->gen_segs, nseg, b->nvoices);
        if (rc) return rc;
        base = b->gen_set;
        rc = launch_prepare_segments(st, ptrs(b), base, b->nvoices, nseg, start, len, SEG);
        if (rc) return rc;
    }
    sh::Temp parts;
    rc = parts.alloc((size_t)nplanes * len * sizeof(int2v));
    if (rc) return rc;
    SegTab none;
    none.n = 0;
    const dim3 grid(sh::div_up(len, 256 * lf), nplanes);
    const uint32_t seg_frames = nseg == 1 ? (len + 1023u) / 1024u * 1024u : SEG;
#define SH_MIXDOWN(F_) do { if (b->has_guard) hipLaunchKernelGGL((k_generate_lean_harm<F_, short, true>), grid, dim3(256), 0, st, trig_table(), base, b->nvoices, len, seg_frames, \
                                          (short*)nullptr, (size_t)0, none, rsplit, scale, flag, (int2v*)parts.buf.ptr, (size_t)len); \
                            else hipLaunchKernelGGL((k_generate_lean_harm<F_, short, true, false>), grid, dim3(256), 0, st, trig_table(), base, b->nvoices, len, seg_frames, \
                                          (short*)nullptr, (size_t)0, none, rsplit, scale, flag, (int2v*)parts.buf.ptr, (size_t)len); } while (0)
    if (lf == 16) SH_MIXDOWN(16); else if (lf == 8) SH_MIXDOWN(8); else SH_MIXDOWN(4);
#undef SH_MIXDOWN
    SH_CHECK_LAUNCH("k_generate_lean_harm(fold)");
    hipLaunchKernelGGL(k_mixdown_combine, sh::grid1d(len, 256), dim3(256), 0, st, (const int2v*)parts.buf.ptr, nplanes, (size_t)len, 0u, len, out);
    SH_CHECK_LAUNCH("k_mixdown_combine");
    return SH_OK;
}

}  // namespace

int sh_bank_mixdown_i16_async(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* out_i16) {
    SH_REQUIRE_INIT();
    if (!b || !out_i16) return sh::set_error(SH_ERR_INVALID, "sh_bank_mixdown_i16: NULL argument");
    if (nframes > 0xFFFF0000u) return sh::set_error(SH_ERR_INVALID, "sh_bank_mixdown_i16: at most 2^32 - 65536 frames per call");
    if (out_i16->bytes / 2 < nframes) return sh::set_error(SH_ERR_INVALID, "sh_bank_mixdown_i16: output buffer too small");
    int rc = bank_check_plain(b, "sh_bank_mixdown_i16");
    if (rc || nframes == 0) return rc;
    if (b->nvoices > 32768) return sh::set_error(SH_ERR_INVALID, "sh_bank_mixdown_i16: at most 32768 voices");
    if (sh::state().quantise_round)
        return sh::set_error(SH_ERR_INVALID, "sh_bank_mixdown_i16: the fused quantiser truncates; under SH_OPT_QUANTISE_ROUND quantise float64 rows and fold them (sh_mix_chain_i16)");
    int* flag = sh::state().flag + 1;
    short* out = (short*)out_i16->ptr;
    // Stretches of whole 65 536-frame segments in which every voice takes the lean polynomial-Harmonics loop (its records then hold the
    // bank in voice order, silent voices -- which add nothing to a chain -- left out) are folded where the samples are made; the others
    // (the notes' attack and decay, banks with other kinds of voice, short calls) go through int16 rows and the chain kernel.
    constexpr uint32_t SEG = 65536;
    const bool lean_bank = !b->launch_rows && b->lean_candidates != 0 && b->lean_fm_candidates == 0 && b->all_lean && nframes >= 8192;
    uint32_t f0 = 0;
    sh::state().last_mixdown_fused = 0;
    while (f0 < nframes && !rc) {
        const uint32_t n0 = nframes - f0 < SEG ? nframes - f0 : SEG;
        const bool fused = lean_bank && b->no_general_voice(start + f0, n0);
        uint32_t f1 = f0 + n0;
        while (f1 < nframes) {                                // extend the stretch while the next segment is of the same sort
            const uint32_t n1 = nframes - f1 < SEG ? nframes - f1 : SEG;
            if ((lean_bank && b->no_general_voice(start + f1, n1)) != fused) break;
            if (!fused && f1 - f0 >= 4 * SEG) break;          // (two-step stretches: a temporary of nvoices x 2 B per frame each)
            if (fused && f1 - f0 >= 16 * SEG) break;          // (fused stretches: planes of 16 B per frame each -- 1024 voices: 0.5 GB per 2^20 frames; the chain is per frame, so cutting changes nothing)
            f1 += n1;
        }
        rc = fused ? mixdown_fused(b, start + f0, f1 - f0, scale, out + f0, flag) : mixdown_two_step(b, start + f0, f1 - f0, scale, out + f0, flag);
        if (fused) sh::state().last_mixdown_fused += 1;
        f0 = f1;
    }
    return rc;
}

int sh_bank_mixdown_i16(sh_bank* b, uint64_t start, uint32_t nframes, double scale, sh_buf* out_i16) {
    SH_API_LOCK();
    const int rc = sh_bank_mixdown_i16_async(b, start, nframes, scale, out_i16);
    if (rc) return sync_call_failed(rc);
    if (nframes == 0) return rc;
    return sh_overflow_check();
}

int sh_bank_generate_rows(sh_bank* b, uint64_t start, uint32_t nframes, const sh_buf* rows_f64, size_t row_stride,
                          sh_buf* voices_out, size_t stride) {
    if (!b || !rows_f64) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_rows: NULL argument");
    SH_API_LOCK();                                           // held across sh_bank_generate (recursive): the rows belong to this call
    if (!b->d_fm_row) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_rows: sh_bank_set_rows has not been called");
    if (row_stride < nframes) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_rows: row_stride < nframes");
    if (b->fm_row_max >= 0 && rows_f64->bytes / 8 < (size_t)b->fm_row_max * row_stride + nframes)
        return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_rows: rows buffer too small for row %d", b->fm_row_max);
    b->launch_rows = (const double*)rows_f64->ptr;
    b->launch_row_stride = row_stride;
    const int rc = sh_bank_generate(b, start, nframes, voices_out, stride);
    b->launch_rows = nullptr;
    b->launch_row_stride = 0;
    return rc;
}

int sh_bank_generate_rows_i16(sh_bank* b, uint64_t start, uint32_t nframes, const sh_buf* rows_f64, size_t row_stride,
                              double scale, sh_buf* voices_out, size_t stride) {
    if (!b || !rows_f64) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_rows_i16: NULL argument");
    SH_API_LOCK();
    if (!b->d_fm_row) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_rows_i16: sh_bank_set_rows has not been called");
    if (row_stride < nframes) return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_rows_i16: row_stride < nframes");
    if (b->fm_row_max >= 0 && rows_f64->bytes / 8 < (size_t)b->fm_row_max * row_stride + nframes)
        return sh::set_error(SH_ERR_INVALID, "sh_bank_generate_rows_i16: rows buffer too small for row %d", b->fm_row_max);
    b->launch_rows = (const double*)rows_f64->ptr;
    b->launch_row_stride = row_stride;
    const int rc = sh_bank_generate_i16(b, start, nframes, scale, voices_out, stride);
    b->launch_rows = nullptr;
    b->launch_row_stride = 0;
    return rc;
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

}  // extern "C"
