"""
Oscillators with the synthplayer API, rendered by HIP kernels on an MI355X.

Drop-in for the hot-path subset of upstream ``synthplayer/oscillators.py`` (tree not mounted at
/root/reference): ``Sine``, ``Sawtooth``, ``Square``, ``Pulse``, ``Harmonics`` with
``fm_lfo=`` / ``pwm_lfo=``, and ``EnvelopeFilter``.  Same constructor signatures, same ``blocks()``
generator protocol (lists of ``params.norm_osc_blocksize`` Python floats).  The samples are computed
on the GPU -- one thread per output sample -- through libsynthhip.so; there is no CPU path.

Host code here only *describes* a voice: it evaluates the same float64 expressions the reference
evaluates once per oscillator (increment, start phase, envelope slopes), turns the running sums the
reference carries per sample into exact tables (phasetable.py) and packs everything into the
``sh_voice`` record the kernels read.

Beyond the reference API each oscillator has ``render(nframes, start=None) -> numpy float32`` (bulk, the
float32 PCM storage format) and ``render_f64`` (the float64 values ``blocks()`` yields and the quantiser
consumes: float32 is a storage format only, never an intermediate on the way to integer PCM), and can be
put in a ``mixer.VoiceBank``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from fractions import Fraction
from functools import lru_cache
from math import pi, sin, cos, sqrt
from typing import Dict, Generator, List, Optional, Sequence, Tuple

import numpy as np

from . import params
from . import _native as N
from .phasetable import PhaseTable

__all__ = ["Oscillator", "Sine", "Triangle", "Sawtooth", "Square", "Pulse", "Harmonics", "SquareH", "SawtoothH",
           "Linear", "WhiteNoise", "EnvelopeFilter", "MixingFilter", "AmpModulationFilter", "ClipFilter", "AbsFilter",
           "NullFilter", "DelayFilter", "EchoFilter"]

_SUPERBLOCK = 128          # blocks rendered per kernel launch behind blocks()


class _SourceExhausted(RuntimeError):
    """The source of an envelope ended inside the envelope's phases: upstream's generator ends with next()'s StopIteration,
    which Python (PEP 479) hands to the consumer as RuntimeError("generator raised StopIteration").  A class of its own so
    that blocks() can tell it from a device error (SynthHipError is a RuntimeError too)."""
_DENSE_MAX_K = 4096        # Harmonics: use the Clenshaw (dense) form when max k <= min(this, 8*len+64)


@lru_cache(maxsize=4096)
def _table(t0: float, inc: float) -> PhaseTable:
    return PhaseTable(t0, inc)


class LfoTable:
    """What an FM carrier's angle owes its Sine LFO, as the reference's loop forms it, piece by piece.

    The reference evaluates sin(t * freq + phase_correction) with freq_j = f (1 + lfo_j), phase_correction += (freq_{j-1} - freq_j) t_j
    and t += inc: by Abel's summation the angle at sample n is  sum_{j<n} freq_j (t_{j+1} - t_j)  =  f (1 + bias) t_n  +  f sum_{j<n}
    amp sin(phi_j) (t_{j+1} - t_j)  -- with t the ACCUMULATED time and phi the ACCUMULATED phase of the LFO (an oscillator of its own),
    both piecewise exactly linear like every running sum here (phasetable.py) and both drifting from n inc and a + n d as they grow.
    Rounds 1-3 summed along the ideal lines (f inc sum amp sin(a + j d) + f inc bias n): against the oracle a 3.5 kHz carrier under a
    5 Hz LFO of depth 0.5 was 1e-5 off after 30 s and 1e-3 after 300 s -- a t^2 law no test looked at.

    On a joint piece g (between consecutive piece starts of EITHER table: samples n0_g .. n0_{g+1} - 1, LFO phase u_g + (j - n0_g) dl_g,
    time step tau_g) the sum has the closed form

        sum_{j<n} amp sin(phi_j) (t_{j+1} - t_j) = W_g + tau_g K_g (cos(u_g - dl_g / 2) - cos(u_g + (n - n0_g - 1/2) dl_g)),

    K_g = amp / (2 sin(dl_g / 2)), W_g = the pieces in front (each with its own tau, plus r amp sin(phi) of the one sample at a time
    piece's end whose step is not tau: r = the next piece's first t minus what the line gives).  The device evaluates
    frequency (t_n + inc K'_g (C'_g - cos(...))) with frequency = f (1 + bias): K'_g = tau_g K_g / ((1 + bias) inc), C'_g =
    cos(u_g - dl_g / 2) + W_g / (tau_g K_g) -- extended precision here, float64 in the records.  `records`: two sh_segment-shaped
    records per joint piece, (n0, u, dl) and (n0, K', C'), addressed through an FM voice's seg_offset / seg_count (its carrier runs on
    the TIME table: its own slots are free)."""

    def __init__(self, a: float, d: float, amp: float, bias: float, inc: float) -> None:
        ld = np.longdouble
        lt = PhaseTable(a, d).records
        tt = _table(0.0, inc).records
        ln0, tn0 = lt["n0"].astype(np.uint64), tt["n0"].astype(np.uint64)
        n0 = np.union1d(ln0, tn0)                                                   # sorted, unique: the joint pieces' starts
        G = len(n0)
        p = np.searchsorted(ln0, n0, side="right") - 1                              # the LFO's piece, the time table's piece of each
        q = np.searchsorted(tn0, n0, side="right") - 1
        dl, tau = lt["dt"][p].astype(ld), tt["dt"][q].astype(ld)
        u = lt["t0"][p].astype(ld) + (n0 - ln0[p]).astype(ld) * dl                   # (a table value: a float64, exactly)
        count = np.empty(G, dtype=ld)
        count[:-1] = (n0[1:] - n0[:-1]).astype(ld)
        count[-1] = ld(1.0)                                                          # (the last piece has no end: nothing lies behind it)
        u_last = u + (count - ld(1.0)) * dl
        half = np.sin(dl / ld(2.0))
        dead = (half == 0) | (tau == 0)                                              # (2^60 samples on: a sum that no longer moves)
        K = np.where(dead, ld(0.0), ld(amp) / (ld(2.0) * np.where(dead, ld(1.0), half)))
        cos_b = np.cos(u - dl / ld(2.0))
        piece = tau * K * (cos_b - np.cos(u_last + dl / ld(2.0)))                    # the sum over the whole piece, every step tau
        # the one step at a time piece's end that is not tau: t0 of the next time piece - (the line at the piece's last sample)
        r = np.zeros(G, dtype=ld)
        ends_time = np.zeros(G, dtype=bool)
        ends_time[:-1] = np.isin(n0[1:], tn0)
        nxt = np.minimum(q + 1, len(tn0) - 1)
        t_last = tt["t0"][q].astype(ld) + (n0 + count.astype(np.float64).astype(np.uint64) - np.uint64(1) - tn0[q]).astype(ld) * tau
        r[ends_time] = (tt["t0"][nxt].astype(ld) - t_last - tau)[ends_time]
        piece = piece + r * ld(amp) * np.sin(u_last)
        W = np.concatenate([[ld(0.0)], np.cumsum(piece[:-1])])
        Kp = tau * K / (ld(1.0 + bias) * ld(inc))
        Cp = cos_b + np.where(dead, ld(0.0), W / np.where(dead, ld(1.0), tau * K))
        out = np.zeros(2 * G, dtype=N.SEGMENT_DTYPE)
        out["n0"][0::2] = n0
        out["t0"][0::2] = u.astype(np.float64)
        out["dt"][0::2] = dl.astype(np.float64)
        out["n0"][1::2] = n0
        out["t0"][1::2] = Kp.astype(np.float64)
        out["dt"][1::2] = Cp.astype(np.float64)
        self.records = out
        self.pieces = G


@lru_cache(maxsize=4096)
def _lfo_table(a: float, d: float, amp: float, bias: float, inc: float) -> LfoTable:
    return LfoTable(a, d, amp, bias, inc)


def time_step_weights(inc: float, start: int, n: int) -> List[Tuple[int, int, float]]:
    """For the samples [start, start + n) of an FM carrier whose time is the accumulated t += inc: runs (offset, count, w - 1) with
    w = (t_{j+1} - t_j) / inc, the ACTUAL step of the accumulated time over the ideal one -- constant on a piece of the time table,
    another value for the one sample at a piece's end.  The reference's angle is sum freq_j (t_{j+1} - t_j) (LfoTable): a modulator
    that is summed by a scan (the buffer path) is weighted with w first, so that f inc cumsum(w m) is that sum."""
    tab = _table(0.0, inc)
    rec = tab.records
    n0s = rec["n0"]
    ld = np.longdouble
    out: List[Tuple[int, int, float]] = []
    pos, end = start, start + n
    q = int(np.searchsorted(n0s, np.uint64(pos), side="right") - 1)
    while pos < end:
        tau = ld(rec["dt"][q])
        nxt = int(n0s[q + 1]) if q + 1 < len(n0s) else None
        stop = end if nxt is None else min(end, nxt - 1)              # (the piece's last sample steps into the next piece)
        if stop > pos:
            out.append((pos - start, stop - pos, float((tau - ld(inc)) / ld(inc))))
            pos = stop
        if pos < end and nxt is not None and pos == nxt - 1:
            t_last = ld(rec["t0"][q]) + ld(pos - int(n0s[q])) * tau
            step = ld(rec["t0"][q + 1]) - t_last
            out.append((pos - start, 1, float((step - ld(inc)) / ld(inc))))
            pos += 1
            q += 1
    return out


@lru_cache(maxsize=None)
def _cheb_u(n: int) -> Tuple[Tuple[int, ...], ...]:
    """Monomial coefficients (ascending powers) of the Chebyshev polynomials U_0 .. U_n."""
    U = [[1], [0, 2]]
    for _ in range(2, n + 1):
        a = [0] + [2 * c for c in U[-1]]
        b = U[-2] + [0] * (len(a) - len(U[-2]))
        U.append([x - y for x, y in zip(a, b)])
    return tuple(tuple(u) for u in U[:n + 1])


@lru_cache(maxsize=1024)
def series_polynomial(coef_by_k: Tuple[float, ...]) -> Optional[Tuple[float, ...]]:
    """sum_{k=1..16} a_k sin(k t) = sin(t) * P(cos t): the 16 coefficients of P, highest power first,
    converted in exact rational arithmetic (coef_by_k[k-1] = a_k).  None when the monomial form would
    lose accuracy (its coefficients grow like 2^k; bounded here so that the float64 Horner evaluation
    stays ~1e-11 of the largest partial)."""
    assert len(coef_by_k) == 16
    U = _cheb_u(15)
    p = [Fraction(0)] * 16
    for k in range(1, 17):
        a = Fraction(coef_by_k[k - 1])
        if a:
            for j, c in enumerate(U[k - 1]):
                p[j] += a * c
    pf = [float(x) for x in p]
    scale = max((abs(a) for a in coef_by_k), default=0.0)
    if scale == 0.0 or max(abs(x) for x in pf) > scale * float(1 << 20):
        return None
    return tuple(reversed(pf))


@dataclass(frozen=True)
class EnvelopeSpec:
    n_attack_end: int
    n_decay_end: int
    n_sustain_end: int
    n_release_end: int
    attack_slope: float
    decay_slope: float
    sustain_level: float
    release_slope: float
    tail_amp: float
    has_tail: bool
    stop_at_end: bool

    @property
    def length(self) -> int:
        return self.n_release_end + (1 if self.has_tail else 0)


def envelope_spec(attack: float, decay: float, sustain: float, sustain_level: float, release: float,
                  samplerate: int, stop_at_end: bool = False) -> EnvelopeSpec:
    """Replay EnvelopeFilter's accumulated ``time`` exactly and return the sample indices at which
    its four ``while time < ...`` loops end (upstream: oscillators.py EnvelopeFilter)."""
    return _envelope_spec(attack, decay, sustain, sustain_level, release, samplerate, stop_at_end, params.variants["envelope"] == "le")


@lru_cache(maxsize=4096)          # a bank's voices usually share one ADSR: the four boundary searches are done once
def _envelope_spec(attack: float, decay: float, sustain: float, sustain_level: float, release: float,
                   samplerate: int, stop_at_end: bool, inclusive: bool) -> EnvelopeSpec:
    increment = 1.0 / samplerate
    tt = _table(0.0, increment)

    def first_ge(x: float) -> int:
        # the number of samples a loop `while time < x` delivers: the first index whose accumulated time is >= x; inclusive (the other
        # reading, `while time <= x`): the first index whose time is > x, i.e. >= the successor of x
        if inclusive:
            return 0 if x < 0.0 else tt.first_index_ge(float(np.nextafter(x, np.inf)))
        return 0 if x <= 0.0 else tt.first_index_ge(x)

    end_time_decay = attack + decay
    end_time_sustain = end_time_decay + sustain
    end_time_release = end_time_sustain + release
    n_a = first_ge(attack) if attack else 0
    n_d = max(n_a, first_ge(end_time_decay)) if decay else n_a
    n_s = max(n_d, first_ge(end_time_sustain))
    attack_slope = 1.0 / attack * increment if attack else 0.0
    decay_slope = (sustain_level - 1.0) / decay * increment if decay else 0.0
    release_slope = 0.0
    tail_amp = 0.0
    has_tail = False
    n_r = n_s
    if release:
        release_slope = (-sustain_level) / release * increment
        n_r = max(n_s, first_ge(end_time_release))
        # the reference's accumulated amp after the release loop decides the extra sample
        tail_amp = _table(sustain_level, release_slope).value(n_r - n_s)
        has_tail = tail_amp > 0.0
    return EnvelopeSpec(n_a, n_d, n_s, n_r, attack_slope, decay_slope, sustain_level, release_slope,
                        tail_amp, has_tail, stop_at_end)


@dataclass
class VoiceSpec:
    kind: int
    amplitude: float
    bias: float
    pulsewidth: float = 0.0
    fm_mode: int = N.SH_FM_NONE
    carrier: Optional[PhaseTable] = None          # SH_FM_NONE
    time_table: Optional[PhaseTable] = None       # FM
    frequency: float = 0.0
    fm_phase0: float = 0.0
    fm_inc: float = 0.0
    lfo: Tuple[float, float, float, float, float, float] = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)  # a, d, amp, bias, K, C0
    lfo_table: Optional[LfoTable] = None          # SH_FM_SINE with a Sine LFO that moves: its running sum piece by piece
    harm_poly: Optional[Tuple[float, ...]] = None       # 16 polynomial coefficients (all k <= 16), highest power first
    harm_dense: Optional[Tuple[float, ...]] = None      # Clenshaw coefficients, k = K..1, len % 8 == 0
    harm_sparse: Optional[Tuple[Tuple[float, float], ...]] = None
    # the voice's own list (k, a_k) in the reference's order, carried beside a polynomial / Clenshaw form: the int16 routes recompute a
    # sample term by term from it where int(scale * v) of the fast form lies within the forms' distance of an integer (sh_voice::guard_*)
    harm_guard: Optional[Tuple[Tuple[float, float], ...]] = None
    env: Optional[EnvelopeSpec] = None
    needs_pwm: bool = False
    flip: bool = False                                  # SawtoothH mirrors the wave around the bias
    noise_seed: int = 0                                 # WhiteNoise
    noise_hold: int = 0
    start_frame: int = 0                                # onset: silent before, the voice's own sample 0 at this frame (DelayFilter)


def _with(sp: VoiceSpec, **changes) -> VoiceSpec:
    """A copy of `sp` with some fields changed (dataclasses.replace re-runs __init__ with introspection: 10 us per call, two calls per
    note of a table of notes)."""
    new = VoiceSpec.__new__(VoiceSpec)
    new.__dict__ = {**sp.__dict__, **changes}
    return new


def guard_bounds(partials, poly, dense) -> Tuple[float, float]:
    """(per unit |t|, constant) bound, at unit amplitude, of |fast form - the reference's term-by-term sum| for a harmonic list:
    the reference rounds every product t * k (half an ulp of it: <= |t k| 2^-53 of phase, a_k of that in the sum; doubled here), the
    fast forms evaluate sum a_k sin(k t) of the exact products -- plus what their own evaluation leaves: Horner in the monomial basis
    (2^-50 of the coefficient mass: measured 1.6e-17 of it for a_k = 1/k, k <= 16), Clenshaw (2^-46 sum |a_k| k^2), and the rotations
    that carry (sin t, cos t) from frame to frame (< 50 ulp of angle: 2^-46 sum |a_k k|).  tests/test_gpu_guard.py holds the measured
    distance under half of this."""
    A = sum(abs(a) * abs(k) for k, a in partials)
    per_t = A * 2.0 ** -52
    if poly is not None:
        const = 2.0 ** -50 * sum(abs(c) for c in poly) + 2.0 ** -46 * A
    else:
        const = 2.0 ** -46 * sum(abs(a) * k * k for k, a in partials) + 2.0 ** -46 * A
    return per_t, const


def pack_voices(specs: Sequence[VoiceSpec], gains: Optional[Sequence[Tuple[float, float]]] = None):
    """VoiceSpec list -> (voices, segs, coefs, partials) arrays in the C layout.  Tables and harmonic
    lists shared by several voices are stored once."""
    voices = np.zeros(len(specs), dtype=N.VOICE_DTYPE)
    seg_chunks: List[np.ndarray] = []
    seg_index: Dict[int, Tuple[int, int]] = {}
    nsegs = 0
    coef_list: List[float] = []
    coef_index: Dict[Tuple[float, ...], int] = {}
    part_list: List[Tuple[float, float]] = []
    part_index: Dict[Tuple[Tuple[float, float], ...], int] = {}

    def add_table(tab: PhaseTable) -> Tuple[int, int]:
        nonlocal nsegs
        key = id(tab)
        if key not in seg_index:
            arr = tab.records                       # (the C layout already: phasetable.phase_table_records)
            seg_index[key] = (nsegs, len(arr))
            seg_chunks.append(arr)
            nsegs += len(arr)
        return seg_index[key]

    # columns first (one assignment per field of the record array, not one per voice and field: a table of 22 528 notes packs in
    # 60 ms instead of 250), then what needs a look-up per voice: the shared tables and harmonic lists, FM and envelope records
    n = len(specs)
    voices["kind"] = [s.kind for s in specs]
    voices["fm_mode"] = [s.fm_mode for s in specs]
    voices["amplitude"] = [s.amplitude for s in specs]
    voices["bias"] = [s.bias for s in specs]
    voices["pulsewidth"] = [s.pulsewidth for s in specs]
    voices["flip"] = [1 if s.flip else 0 for s in specs]
    voices["noise_seed"] = np.array([s.noise_seed & 0xFFFFFFFFFFFFFFFF for s in specs], dtype=np.uint64)
    voices["noise_hold"] = [s.noise_hold for s in specs]
    voices["start_frame"] = np.array([s.start_frame for s in specs], dtype=np.uint64)
    if gains is None:
        voices["gain_l"] = 1.0
        voices["gain_r"] = 1.0
    else:
        voices["gain_l"] = [gains[i][0] for i in range(n)]
        voices["gain_r"] = [gains[i][1] for i in range(n)]
    seg_off, seg_cnt = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    tseg_off, tseg_cnt = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    h_off, h_cnt, h_dense = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    for i, s in enumerate(specs):
        if s.fm_mode == N.SH_FM_NONE:
            seg_off[i], seg_cnt[i] = add_table(s.carrier)
        else:
            tseg_off[i], tseg_cnt[i] = add_table(s.time_table)
            if s.lfo_table is not None:
                seg_off[i], seg_cnt[i] = add_table(s.lfo_table)     # (the LFO's pieces, two records each: LfoTable)
            v = voices[i]
            v["frequency"] = s.frequency
            v["fm_phase0"] = s.fm_phase0
            v["fm_inc"] = s.fm_inc
            (v["lfo_a"], v["lfo_d"], v["lfo_amp"], v["lfo_bias"], v["lfo_K"], v["lfo_C0"]) = s.lfo
        if s.harm_poly is not None:
            if s.harm_poly not in coef_index:
                coef_index[s.harm_poly] = len(coef_list)
                coef_list.extend(s.harm_poly)
            h_off[i], h_cnt[i], h_dense[i] = coef_index[s.harm_poly], 16, 2
        elif s.harm_dense is not None:
            if s.harm_dense not in coef_index:
                coef_index[s.harm_dense] = len(coef_list)
                coef_list.extend(s.harm_dense)
            h_off[i], h_cnt[i], h_dense[i] = coef_index[s.harm_dense], len(s.harm_dense), 1
        elif s.harm_sparse is not None:
            if s.harm_sparse not in part_index:
                part_index[s.harm_sparse] = len(part_list)
                part_list.extend(s.harm_sparse)
            h_off[i], h_cnt[i], h_dense[i] = part_index[s.harm_sparse], len(s.harm_sparse), 0
    voices["seg_offset"], voices["seg_count"] = seg_off, seg_cnt
    voices["time_seg_offset"], voices["time_seg_count"] = tseg_off, tseg_cnt
    voices["harm_offset"], voices["harm_count"], voices["harm_dense"] = h_off, h_cnt, h_dense
    # the int16 boundary guard of the polynomial / Clenshaw voices (not under FM: there the contract is a bound, not equality)
    g_off, g_cnt = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    g_t, g_c = np.zeros(n), np.zeros(n)
    bound_cache: Dict[int, Tuple[float, float]] = {}
    for i, s in enumerate(specs):
        if s.harm_guard is None or s.fm_mode != N.SH_FM_NONE or (s.harm_poly is None and s.harm_dense is None):
            continue
        if s.harm_guard not in part_index:
            part_index[s.harm_guard] = len(part_list)
            part_list.extend(s.harm_guard)
        g_off[i], g_cnt[i] = part_index[s.harm_guard], len(s.harm_guard)
        key = id(s.harm_poly) if s.harm_poly is not None else id(s.harm_dense)
        if key not in bound_cache:
            bound_cache[key] = guard_bounds(s.harm_guard, s.harm_poly, s.harm_dense)
        per_t, const = bound_cache[key]
        gmax = max(1.0, abs(s.env.sustain_level)) if s.env is not None else 1.0
        a = abs(s.amplitude)
        gt = gmax * a * per_t
        gc = gmax * (a * const + 2.0 ** -50 * (3.0 * abs(s.bias) + a * sum(abs(x[1]) for x in s.harm_guard)))
        if not (gt < 1.0 and gc < 1.0):          # (NaN / huge amplitudes: no guard -- such a voice overflows any integer format anyway)
            g_off[i], g_cnt[i] = 0, 0
            continue
        g_t[i], g_c[i] = gt, gc
    voices["guard_offset"], voices["guard_count"], voices["guard_t"], voices["guard_c"] = g_off, g_cnt, g_t, g_c
    # envelopes: a table of notes shares a handful of EnvelopeSpec objects -- one record per object, copied to its voices
    env_rows: Dict[int, List[int]] = {}
    env_objs: Dict[int, EnvelopeSpec] = {}
    for i, s in enumerate(specs):
        if s.env is not None:
            env_rows.setdefault(id(s.env), []).append(i)
            env_objs[id(s.env)] = s.env
    env_col = voices["env"]
    for key, rows in env_rows.items():
        ev = env_objs[key]
        rec = np.zeros((), dtype=env_col.dtype)
        rec["n_attack_end"], rec["n_decay_end"] = ev.n_attack_end, ev.n_decay_end
        rec["n_sustain_end"], rec["n_release_end"] = ev.n_sustain_end, ev.n_release_end
        rec["attack_slope"], rec["decay_slope"] = ev.attack_slope, ev.decay_slope
        rec["sustain_level"], rec["release_slope"] = ev.sustain_level, ev.release_slope
        rec["tail_amp"] = ev.tail_amp
        rec["enabled"] = 1
        rec["has_tail"] = 1 if ev.has_tail else 0
        env_col[np.array(rows, dtype=np.int64)] = rec
    segs = np.concatenate(seg_chunks) if seg_chunks else np.zeros(0, dtype=N.SEGMENT_DTYPE)
    coefs = np.array(coef_list, dtype=np.float64)
    partials = np.zeros(len(part_list), dtype=N.PARTIAL_DTYPE)
    for i, (k, a) in enumerate(part_list):
        partials[i] = (k, a)
    return voices, segs, coefs, partials


class Oscillator:
    """Oscillator base class (upstream: oscillators.py class Oscillator)."""

    def __init__(self, samplerate: int = 0) -> None:
        self.samplerate = samplerate or params.norm_samplerate
        self._bank: Optional[N.Bank] = None
        self._spec_cache: Optional[VoiceSpec] = None
        self._pos = 0                 # next sample index for render() without start / blocks()
        self._fm_pos = 0              # general FM: running LFO sum up to _fm_pos
        self._fm_carry = 0.0

    # -- description ----------------------------------------------------------------------------
    def _make_spec(self) -> VoiceSpec:
        raise NotImplementedError

    def _fm_source(self) -> Optional["Oscillator"]:
        return getattr(self, "fm", None)

    def _pwm_source(self) -> Optional["Oscillator"]:
        return getattr(self, "pwm", None)

    def spec(self) -> VoiceSpec:
        if self._spec_cache is None:
            self._spec_cache = self._make_spec()
        return self._spec_cache

    @property
    def length(self) -> Optional[int]:
        """Number of samples the stream has, or None when infinite (EnvelopeFilter stop_at_end)."""
        s = self.spec()
        if s.env is not None and s.env.stop_at_end:
            return s.env.length
        return None

    def _get_bank(self) -> N.Bank:
        """The one-voice bank a single oscillator renders through.  A Harmonics voice in a fast form (polynomial / Clenshaw) is summed
        term by term here unless params.exact_harmonics is False: one voice is launch-bound whatever its form, and a float64 block
        equal to the reference's to an ulp is what Sample.from_osc_block's truncation needs (params.py)."""
        if self._bank is None:
            sp = self.spec()
            if params.exact_harmonics is None and sp.harm_guard is not None and sp.fm_mode == N.SH_FM_NONE and sp.harm_sparse is None:
                sp = _with(sp, harm_poly=None, harm_dense=None, harm_sparse=sp.harm_guard, harm_guard=None)
            self._bank = N.Bank(*pack_voices([sp]))
        return self._bank

    # -- rendering ------------------------------------------------------------------------------
    def _fm_cumsum(self, start: int, n: int) -> N.DeviceBuffer:
        """L(start) .. L(start+n-1) on the device for an arbitrary fm_lfo (sequential access)."""
        lfo = self._fm_source()
        if start != self._fm_pos:
            if start < self._fm_pos:
                self._fm_pos, self._fm_carry = 0, 0.0
            while self._fm_pos < start:          # catch up (random access into a recurrence)
                step = min(1 << 20, start - self._fm_pos)
                self._advance_fm(lfo, self._fm_pos, step, keep=False)
        return self._advance_fm(lfo, start, n, keep=True)

    def _advance_fm(self, lfo: "Oscillator", start: int, n: int, keep: bool) -> Optional[N.DeviceBuffer]:
        mod = lfo._render_f64_device(start, n)
        # the modulator's samples weighted with the accumulated time's actual steps over inc (time_step_weights): m += m (w - 1), in place
        for off, cnt, wm1 in time_step_weights(self.spec().fm_inc, start, n):
            if wm1 != 0.0:
                N.check(N.lib().sh_ew_f64(N.SH_EW_AXPY, mod.handle, off, mod.handle, off, cnt, wm1, 0.0, mod.handle, off, None, 0, None))
        cum = N.DeviceBuffer(n * 8)
        carry = C.c_double()
        N.check(N.lib().sh_scan_f64(mod.handle, n, self._fm_carry, cum.handle, C.byref(carry)))
        self._fm_pos, self._fm_carry = start + n, carry.value
        mod.free()
        if keep:
            return cum
        cum.free()
        return None

    def _render_device(self, start: int, n: int, out_host: Optional[np.ndarray] = None,
                       out_f32: Optional[N.DeviceBuffer] = None, out_off: int = 0,
                       out_f64: Optional[N.DeviceBuffer] = None) -> None:
        s = self.spec()
        bank = self._get_bank()
        fm_buf = self._fm_cumsum(start, n) if s.fm_mode == N.SH_FM_BUFFER else None
        pwm_buf = None
        if s.needs_pwm:
            pwm_buf = self._pwm_source()._render_f64_device(start, n)
            _pwm_widths(pwm_buf, 0, n)
        N.check(N.lib().sh_osc_render(
            bank.handle, 0,
            fm_buf.handle if fm_buf else None, pwm_buf.handle if pwm_buf else None,
            start, n,
            out_host.ctypes.data if out_host is not None else None,
            out_f32.handle if out_f32 is not None else None, out_off,
            out_f64.handle if out_f64 is not None else None))
        # the modulator buffers go back to the pool; whoever gets them next is ordered behind this kernel by the stream
        if fm_buf is not None:
            fm_buf.free()
        if pwm_buf is not None:
            pwm_buf.free()

    def _render_f64_device(self, start: int, n: int) -> N.DeviceBuffer:
        """This oscillator's samples as float64 in HBM (it is someone's modulator)."""
        out = N.DeviceBuffer(n * 8)
        limit = self.length
        if limit is not None and start + n > limit:
            raise ValueError("modulator stream ended (stop_at_end) before the carrier")
        self._render_device(start, n, out_f64=out)
        return out

    def render(self, nframes: int, start: Optional[int] = None) -> np.ndarray:
        """Samples [start, start+nframes) as float32 (truncated at the end of a finite stream)."""
        if start is None:
            start = self._pos
        limit = self.length
        if limit is not None:
            nframes = max(0, min(nframes, limit - start))
        out = np.empty(nframes, dtype=np.float32)
        if nframes:
            self._render_device(start, nframes, out_host=out)
        self._pos = start + nframes
        return out

    def render_f64(self, nframes: int, start: Optional[int] = None) -> np.ndarray:
        """Samples [start, start+nframes) as float64 -- the values the reference's generator yields (Python floats),
        before any rounding to the float32 storage format (truncated at the end of a finite stream)."""
        if start is None:
            start = self._pos
        limit = self.length
        if limit is not None:
            nframes = max(0, min(nframes, limit - start))
        if nframes == 0:
            self._pos = start
            return np.empty(0, dtype=np.float64)
        buf = self._render_f64_device(start, nframes)
        out = buf.download(np.float64, nframes)
        buf.free()
        self._pos = start + nframes
        return out

    def render_to(self, buf: N.DeviceBuffer, offset: int, nframes: int, start: int) -> None:
        """Samples into an existing float32 device buffer at element ``offset`` (no host copy)."""
        if nframes:
            self._render_device(start, nframes, out_f32=buf, out_off=offset)

    def blocks(self) -> Generator[List[float], None, None]:
        """Upstream protocol: an endless (or, with stop_at_end, finite) stream of blocks."""
        bs = params.norm_osc_blocksize
        pos = 0
        limit = self.length
        single = False                                         # fell back to one block per launch
        while True:
            want = bs if single else bs * _SUPERBLOCK
            if limit is not None:
                want = min(want, limit - pos)
                if want <= 0:
                    return
            try:
                chunk = self.render_f64(want, start=pos)       # float64, like upstream's Python floats
            except _SourceExhausted:
                # a source that ends inside an envelope's phases (see EnvelopeFilter): upstream delivers every block before
                # the one that needs the missing sample -- go on block by block until that one raises
                if want <= bs:
                    raise
                single = True
                chunk = self.render_f64(bs, start=pos)
            pos += len(chunk)
            values = chunk.tolist()
            for i in range(0, len(values), bs):
                yield values[i:i + bs]

    def __iter__(self):
        return self.blocks()


def _increment(frequency: float, samplerate: int, radians: bool) -> float:
    """The per-sample phase step of the non-FM branch of the reference's blocks(), under params.variants["increment"]."""
    if params.variants["increment"] == "div":
        rate = samplerate / frequency
        return 2.0 * pi / rate if radians else 1.0 / rate
    return 2.0 * pi * frequency / samplerate if radians else frequency / samplerate


def _pwm_widths(buf: N.DeviceBuffer, offset: int, n: int) -> None:
    """Per-sample pulse widths (a rendered pwm_lfo) as the device's strict comparison needs them: under params.variants["pulse"] ==
    "le" every width becomes its float64 successor, in place."""
    if params.variants["pulse"] == "le" and n:
        N.check(N.lib().sh_ew_f64(N.SH_EW_NEXTUP, buf.handle, offset, None, 0, n, 0.0, 0.0, buf.handle, offset, None, 0, None))


def _closed_form_lfo(lfo: Optional[Oscillator]) -> bool:
    return type(lfo) is Sine and lfo.fm is None


class _Carrier(Oscillator):
    """Shared constructor/state of the five waveform oscillators."""

    KIND = -1
    RADIANS = False        # Sine/Harmonics keep t in radians, the others in turns

    def __init__(self, frequency: float, amplitude: float = 1.0, phase: float = 0.0, bias: float = 0.0,
                 fm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        super().__init__(samplerate)
        self.frequency = frequency
        self.amplitude = amplitude
        self.bias = bias
        self.fm = fm_lfo
        self._phase = phase

    def _phase_fields(self, force_fm: bool = False) -> dict:
        """Phase bookkeeping: the expressions of the reference's blocks() preamble."""
        sr = self.samplerate
        if self.fm or force_fm:
            if self.RADIANS:
                phase0, inc = self._phase * 2.0 * pi, 2.0 * pi / sr
            else:
                phase0, inc = self._phase, 1.0 / sr
            out = dict(time_table=_table(0.0, inc), frequency=float(self.frequency), fm_phase0=phase0, fm_inc=inc)
            lfo = self.fm
            if lfo is None:
                out.update(fm_mode=N.SH_FM_SINE, lfo=(0.0, 0.0, 0.0, 0.0, 0.0, 0.0))
            elif _closed_form_lfo(lfo) and abs(1.0 + float(lfo.bias)) >= 2.0 ** -10:
                # (a bias at or next to -1 leaves nothing to fold it into -- the closed form carries f (1 + bias) as its frequency and
                # divides the sine part by it: such an LFO takes the buffer path below, which needs no division)
                a = lfo._phase * 2.0 * pi
                d = _increment(lfo.frequency, lfo.samplerate, True)
                half = sin(d / 2.0)
                if half == 0.0 or lfo.amplitude == 0.0:
                    # a constant LFO, lfo_j = c: the angle is f (1 + c) t_n (the sum of freq_j over the ACCUMULATED time steps)
                    c = sin(a) * lfo.amplitude + lfo.bias
                    out.update(fm_mode=N.SH_FM_SINE, frequency=float(self.frequency) * (1.0 + c), lfo=(0.0, 0.0, 0.0, 0.0, 0.0, 0.0))
                else:
                    # (the bias rides in the frequency, f (1 + bias) t_n; the sine part in the table: LfoTable.  lfo = what a library
                    # older than ABI 5 would have read)
                    K = lfo.amplitude / (2.0 * half)
                    out.update(fm_mode=N.SH_FM_SINE, frequency=float(self.frequency) * (1.0 + float(lfo.bias)),
                               lfo=(a, d, float(lfo.amplitude), 0.0, K / (1.0 + float(lfo.bias)), cos(a - d / 2.0)),
                               lfo_table=_lfo_table(a, d, float(lfo.amplitude), float(lfo.bias), inc))
            else:
                out.update(fm_mode=N.SH_FM_BUFFER)
            return out
        increment = _increment(self.frequency, sr, self.RADIANS)
        t0 = self._phase * 2.0 * pi if self.RADIANS else self._phase
        return dict(fm_mode=N.SH_FM_NONE, carrier=_table(t0, increment))

    def _make_spec(self) -> VoiceSpec:
        return VoiceSpec(kind=self.KIND, amplitude=float(self.amplitude), bias=float(self.bias), **self._phase_fields())

    # What "held to the 1e-6 contract" means for an FM voice (not upstream: INTEGRATION.md).  The reference forms the carrier's angle
    # as t * freq + phase_correction with phase_correction += (freq_previous - freq) * t per sample: a running float64 sum whose every
    # addition rounds at ulp(f * depth * t) -- a random walk no closed form can retrace.  Measured against the C
    # oracle (tests/test_gpu_fm_contract.py; profiles/r04_late_parity.txt): RMS error <= FM_WALK_C * ulp(f depth t) * sqrt(n), n = t * sr
    # samples into the note, ulp(x) = 2^-52 x (the smooth envelope of the binade steps).
    FM_WALK_C = 0.5

    def fm_error_bound(self, seconds: float) -> float:
        """Bound on the RMS distance of this voice from the reference's samples ``seconds`` into the note: the random walk of the
        reference's own phase_correction roundings for an FM voice (grows like t^1.5), 0.0 for one without an fm_lfo (whose
        accumulated phase is reproduced bit for bit)."""
        lfo = self.fm
        if lfo is None or seconds <= 0.0:
            return 0.0
        depth = abs(float(getattr(lfo, "amplitude", 1.0))) + abs(float(getattr(lfo, "bias", 0.0)))
        x = abs(float(self.frequency)) * depth * (2.0 * pi if self.RADIANS else 1.0) * seconds     # the size phase_correction has grown to
        rad = (1.0 if self.RADIANS else 2.0 * pi)                                                   # error of the angle in radians
        return self.FM_WALK_C * (2.0 ** -52) * x * sqrt(seconds * self.samplerate) * rad * abs(float(self.amplitude))

    def fm_contract_horizon(self, tolerance: float = 1.0e-6) -> float:
        """Seconds into a note until ``fm_error_bound`` reaches ``tolerance`` (the 1e-6 RMS float contract): how long an FM voice can
        sound before its distance from the reference's own rounding walk may exceed the contract; ``inf`` without an fm_lfo."""
        if self.fm is None or self.fm_error_bound(1.0) == 0.0:
            return float("inf")
        return (tolerance / self.fm_error_bound(1.0)) ** (2.0 / 3.0)                                # the bound grows like t^1.5


class Sine(_Carrier):
    """Sine wave oscillator (upstream: oscillators.py class Sine)."""
    KIND = N.SH_SINE
    RADIANS = True


class Triangle(_Carrier):
    """Perfect triangle wave (upstream: oscillators.py class Triangle)."""
    KIND = N.SH_TRIANGLE


class Sawtooth(_Carrier):
    """Sawtooth oscillator, naive form (upstream: oscillators.py class Sawtooth)."""
    KIND = N.SH_SAWTOOTH


class Square(_Carrier):
    """Perfect square wave (upstream: oscillators.py class Square)."""
    KIND = N.SH_SQUARE

    def _make_spec(self) -> VoiceSpec:
        if params.variants["square"] == "mod1":
            # the other reading, a if t % 1.0 < 0.5 else -a (equal for t >= 0, not for a negative phase: int() truncates toward zero):
            # literally a Pulse of width 0.5 -- the record the device already knows
            return VoiceSpec(kind=N.SH_PULSE, amplitude=float(self.amplitude), bias=float(self.bias), pulsewidth=0.5, **self._phase_fields())
        return super()._make_spec()


class Pulse(_Carrier):
    """Pulse of a given width, optionally pulse-width modulated (upstream: oscillators.py class Pulse)."""
    KIND = N.SH_PULSE

    def __init__(self, frequency: float, amplitude: float = 1.0, phase: float = 0.0, pulsewidth: float = 0.1,
                 bias: float = 0.0, fm_lfo: Optional[Oscillator] = None, pwm_lfo: Optional[Oscillator] = None,
                 samplerate: int = 0) -> None:
        assert 0 <= pulsewidth <= 1
        super().__init__(frequency, amplitude, phase, bias, fm_lfo, samplerate)
        self.pulsewidth = pulsewidth
        self.pwm = pwm_lfo

    def _make_spec(self) -> VoiceSpec:
        # with a pwm_lfo (and no fm_lfo) the reference still runs its modulated loop: t += 1/sr, tt = t*f + phase
        fields = self._phase_fields(force_fm=self.pwm is not None)
        pw = float(self.pulsewidth)
        if params.variants["pulse"] == "le":
            pw = float(np.nextafter(pw, np.inf))      # m <= w  <=>  m < successor(w) for float64 m, w (pwm rows: _pwm_widths)
        return VoiceSpec(kind=self.KIND, amplitude=float(self.amplitude), bias=float(self.bias),
                         pulsewidth=pw, needs_pwm=self.pwm is not None, **fields)


class _StoppedTable(PhaseTable):
    """A running sum that stops changing at sample `stop`: the pieces of `base` before it, then a constant piece."""

    def __init__(self, base: PhaseTable, stop: int) -> None:      # noqa: super().__init__ not called on purpose
        self.t0, self.inc = base.t0, base.inc
        self.segments = [seg for seg in base.segments if seg[0] < stop] + [(stop, base.value(stop), 0.0)]
        self._starts = [seg[0] for seg in self.segments]


class Linear(Oscillator):
    """Linear ramp; a constant if the increment is 0 (upstream: oscillators.py class Linear).  The level grows by
    `increment` per sample (accumulated in float64, like upstream's running sum) for as long as it lies strictly
    between min_value and max_value, then stays where it is."""

    def __init__(self, startlevel: float, increment: float = 0.0, min_value: float = -1.0, max_value: float = 1.0,
                 samplerate: int = 0) -> None:
        super().__init__(samplerate)
        self._value = startlevel
        self._increment = increment
        self._min = min_value
        self._max = max_value

    def _make_spec(self) -> VoiceSpec:
        v, inc, lo, hi = float(self._value), float(self._increment), float(self._min), float(self._max)
        table = _table(v, inc)
        stop = None
        if not lo < v < hi:
            stop = 0
        elif inc > 0.0:
            try:
                stop = table.first_index_ge(hi)
            except OverflowError:
                stop = None                      # the sum saturates below max_value
        elif inc < 0.0:
            try:
                stop = _table(-v, -inc).first_index_ge(-lo)        # float addition is symmetric under negation
            except OverflowError:
                stop = None
        if stop is not None and stop < (1 << 62):
            table = _StoppedTable(table, stop)
        return VoiceSpec(kind=N.SH_LINEAR, amplitude=1.0, bias=0.0, fm_mode=N.SH_FM_NONE, carrier=table)


class WhiteNoise(Oscillator):
    """White noise: a new uniform random value in [-amplitude, amplitude) + bias every int(samplerate/frequency)
    samples, held in between (upstream: oscillators.py class WhiteNoise, which draws from Python's global Mersenne
    Twister).  Here the values come from a counter-based generator -- splitmix64 of (seed + value index) -- so that any
    sample range can be rendered in parallel and reproducibly; `seed` is this build's addition."""

    def __init__(self, frequency: float, amplitude: float = 1.0, bias: float = 0.0, samplerate: int = 0,
                 seed: int = 0) -> None:
        super().__init__(samplerate)
        self.frequency = frequency
        self.amplitude = amplitude
        self.bias = bias
        self.seed = seed

    def _make_spec(self) -> VoiceSpec:
        cycles = int(self.samplerate / self.frequency)
        if cycles < 1:
            raise ValueError("whitenoise frequency cannot be bigger than the sample rate")
        return VoiceSpec(kind=N.SH_NOISE, amplitude=float(self.amplitude), bias=float(self.bias), fm_mode=N.SH_FM_NONE,
                         carrier=_table(0.0, 0.0), noise_seed=int(self.seed), noise_hold=cycles)


@lru_cache(maxsize=4096)
def _harmonic_forms(harmonics, exact=False):
    """(polynomial, dense Clenshaw, sparse) form of a harmonic list [(k, amplitude)]: exactly one of them is not None.
    exact (params.exact_harmonics): the term-by-term form whatever the list -- the reference's own loop, sin(fl(t k)) a_k in order."""
    dense = None
    sparse = None
    poly = None
    ks = [k for k, _ in harmonics]
    integral = not exact and all(float(k) == int(k) for k in ks)
    kmax = max((abs(int(k)) for k in ks), default=0) if integral else 0
    if integral and 0 < kmax <= min(_DENSE_MAX_K, 8 * len(ks) + 64):
        n = (kmax + 7) // 8 * 8
        coef = [0.0] * (n + 1)                       # index k
        for k, a in harmonics:
            k = int(k)
            if k > 0:
                coef[k] += float(a)
            elif k < 0:
                coef[-k] -= float(a)                 # sin(-k t) = -sin(k t)
        if kmax <= 16:
            poly = series_polynomial(tuple((coef[1:] + [0.0] * 16)[:16]))
        if poly is None:
            dense = tuple(coef[n:0:-1])              # k = n .. 1
    else:
        sparse = tuple((float(k), float(a)) for k, a in harmonics)
        if not sparse:
            sparse = ((0.0, 0.0),)
    return poly, dense, sparse


@lru_cache(maxsize=4096)
def _guard_list(harmonics):
    """The list as the term-by-term loop takes it (one shared tuple per distinct list: pack_voices stores it once)."""
    return tuple((float(k), float(a)) for k, a in harmonics) or ((0.0, 0.0),)


class Harmonics(_Carrier):
    """Additive sine series sum_k a_k sin(k*t) (upstream: oscillators.py class Harmonics)."""
    KIND = N.SH_HARMONICS
    RADIANS = True

    def __init__(self, frequency: float, harmonics: Sequence[Tuple[int, float]], amplitude: float = 1.0,
                 phase: float = 0.0, bias: float = 0.0, fm_lfo: Optional[Oscillator] = None,
                 samplerate: int = 0) -> None:
        super().__init__(frequency, amplitude, phase, bias, fm_lfo, samplerate)
        self.harmonics = list(harmonics)

    def _make_spec(self) -> VoiceSpec:
        # (the three forms of a harmonic list depend on the list alone: a table of notes shares a handful of lists among thousands
        # of voices)
        if type(self) in (Harmonics, SquareH) and len(self.harmonics) == 1 and tuple(self.harmonics[0]) == (1, 1.0):
            # one partial, k = 1, a = 1.0: the reference's sum is 0 + sin(t * 1) * 1.0 -- a Sine, bit for bit; as a Sine record it takes
            # the exact treatment of the peaks (tests/test_gpu_int_mixdown.py::test_sine_peaks_on_rational_frequencies)
            return VoiceSpec(kind=N.SH_SINE, amplitude=float(self.amplitude), bias=float(self.bias), **self._phase_fields())
        poly, dense, sparse = _harmonic_forms(tuple(map(tuple, self.harmonics)), bool(params.exact_harmonics))
        guard = _guard_list(tuple(map(tuple, self.harmonics))) if sparse is None and params.int16_guard else None
        if guard is not None and len(guard) > 255:           # (the lean records hold the list's length in eight bits: a longer list is summed term by term)
            poly, dense, sparse, guard = None, None, guard, None
        return VoiceSpec(kind=self.KIND, amplitude=float(self.amplitude), bias=float(self.bias),
                         harm_poly=poly, harm_dense=dense, harm_sparse=sparse, harm_guard=guard, **self._phase_fields())


class SquareH(Harmonics):
    """Square wave built from odd sine harmonics, 1/n amplitudes (upstream: oscillators.py class SquareH)."""

    def __init__(self, frequency: float, num_harmonics: int = 16, amplitude: float = 0.9999, phase: float = 0.0,
                 bias: float = 0.0, fm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        harmonics = [(n, 1.0 / n) for n in range(1, num_harmonics * 2, 2)]
        super().__init__(frequency, harmonics, amplitude, phase, bias, fm_lfo, samplerate)


class SawtoothH(Harmonics):
    """Sawtooth built from all sine harmonics, 1/n amplitudes, flipped (upstream: oscillators.py class SawtoothH)."""

    def __init__(self, frequency: float, num_harmonics: int = 16, amplitude: float = 0.9999, phase: float = 0.0,
                 bias: float = 0.0, fm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        harmonics = [(n, 1.0 / n) for n in range(1, num_harmonics + 1)]
        super().__init__(frequency, harmonics, amplitude, phase + 0.5, bias, fm_lfo, samplerate)

    def _make_spec(self) -> VoiceSpec:
        return _with(super()._make_spec(), flip=True)          # value -> bias*2.0 - value


class EnvelopeFilter(Oscillator):
    """ADSR volume envelope over an oscillator (upstream: oscillators.py class EnvelopeFilter).
    A, D, S, R in seconds, sustain_level an amplitude factor.  Over a waveform source (the usual patch) the envelope is
    fused into the source's kernel; over anything else -- a filter graph, another envelope -- the source is rendered as a
    float64 block and multiplied by the gain curve, which is itself rendered by the fused path (the envelope of a constant 1:
    exactly the reference's accumulated amplitude), one elementwise kernel."""

    def __init__(self, source: Oscillator, attack: float, decay: float, sustain: float, sustain_level: float,
                 release: float, stop_at_end: bool = False, cycle: bool = False) -> None:
        assert attack >= 0 and decay >= 0 and sustain >= 0 and release >= 0
        assert 0 <= sustain_level <= 1
        super().__init__(source.samplerate)
        self._source = source
        self._attack = attack
        self._decay = decay
        self._sustain = sustain
        self._sustain_level = sustain_level
        self._release = release
        self._stop_at_end = stop_at_end
        self._cycle = bool(cycle)
        self._fused = isinstance(source, (_Carrier, Linear, WhiteNoise)) and not self._cycle
        self._gain: Optional["EnvelopeFilter"] = None
        self._cycle_gain: Optional[N.DeviceBuffer] = None      # cycle=True: whole periods of the gain curve, kept in HBM
        self._cycle_frames = 0
        if not self._fused:
            # cycle=True (SURVEY 8(a) row a8's signature): the four phases start over when the release is done -- ``time`` and
            # ``amp`` from zero, the source running on -- so the gain is ONE pass of the phases (stop_at_end: its length is the
            # period) repeated; stop_at_end itself is never reached
            self._gain = EnvelopeFilter(Linear(1.0, samplerate=source.samplerate), attack, decay, sustain, sustain_level, release,
                                        True if self._cycle else stop_at_end)
            if self._cycle and self._gain.spec().env.length == 0:
                raise ValueError("a cycling envelope needs a phase of at least one sample")

    def _fm_source(self):
        return self._source._fm_source() if self._fused else None

    def _pwm_source(self):
        return self._source._pwm_source() if self._fused else None

    def _make_spec(self) -> VoiceSpec:
        if not self._fused:
            raise NotImplementedError("an envelope over a filter graph is rendered block by block, it is not a single voice record")
        env = envelope_spec(self._attack, self._decay, self._sustain, self._sustain_level, self._release,
                            self.samplerate, self._stop_at_end)
        return _with(self._source.spec(), env=env)

    @property
    def length(self) -> Optional[int]:
        if self._fused:
            return super().length
        # the envelope's own phases decide: attack .. release (+ the extra sample) with stop_at_end, endless silence
        # after them without -- the source is not consulted once the release is over
        return None if self._cycle else self._gain.length

    def _render_cycling(self, start: int, n: int) -> N.DeviceBuffer:
        """cycle=True: source[start + i] * gain[(start + i) mod period].  One period of the gain comes from the fused path (the
        envelope of a constant 1) once, laid out back to back until the buffer holds 65 536 frames, so a block is a few
        multiplications of slices whatever the period."""
        period = self._gain.spec().env.length
        src_len = self._source.length
        if src_len is not None and start + n > src_len:
            raise _SourceExhausted("generator raised StopIteration")
        if self._cycle_gain is None:
            reps = max(1, -(-65536 // period))
            one = self._gain._render_f64_device(0, period)
            buf = N.DeviceBuffer(8 * period * reps)
            N.check(N.lib().sh_buf_copy(buf.handle, 0, one.handle, 0, 8 * period))
            one.free()
            have = 1                                  # periods laid out so far; doubled per copy: O(log reps) device copies, not
            while have < reps:                        # reps of them (a period of a few samples made tens of thousands: ADVICE r03)
                cnt = min(have, reps - have)
                N.check(N.lib().sh_buf_copy(buf.handle, 8 * period * have, buf.handle, 0, 8 * period * cnt))
                have += cnt
            self._cycle_gain, self._cycle_frames = buf, period * reps
        out = self._source._render_f64_device(start, n)
        done = 0
        while done < n:
            pos = (start + done) % period
            cnt = min(n - done, self._cycle_frames - pos)
            N.check(N.lib().sh_ew_f64(N.SH_EW_MUL, out.handle, done, self._cycle_gain.handle, pos, cnt, 0.0, 0.0,
                                      out.handle, done, None, 0, None))
            done += cnt
        return out

    def close(self) -> None:
        """Give the periods of a cycling envelope's gain curve back to the buffer pool (also done when the object is collected)."""
        if self._cycle_gain is not None:
            self._cycle_gain.free()
            self._cycle_gain, self._cycle_frames = None, 0

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown: the library may be gone
            pass

    def _render_f64_device(self, start: int, n: int) -> N.DeviceBuffer:
        if self._fused:
            return super()._render_f64_device(start, n)
        if self._cycle:
            return self._render_cycling(start, n)
        limit = self.length
        if limit is not None and start + n > limit:
            raise ValueError("stream ended before the requested range")
        # upstream's generator pulls one source sample per envelope sample up to the end of the release (+ the extra
        # sample) and none after; a source that ends inside that range ends the generator with next()'s StopIteration,
        # which Python (PEP 479) hands to the consumer as RuntimeError
        phases = self._gain.spec().env.length
        n_src = max(0, min(start + n, phases) - start)
        src_len = self._source.length
        if src_len is not None and n_src and start + n_src > src_len:
            raise _SourceExhausted("generator raised StopIteration")
        gain = self._gain._render_f64_device(start, n)               # exactly 0.0 from the end of the phases on
        if n_src:
            src = self._source._render_f64_device(start, n_src)
            N.check(N.lib().sh_ew_f64(N.SH_EW_MUL, src.handle, 0, gain.handle, 0, n_src, 0.0, 0.0, gain.handle, 0, None, 0, None))
            src.free()
        return gain

    def _render_device(self, start, n, out_host=None, out_f32=None, out_off=0, out_f64=None) -> None:
        if self._fused:
            return super()._render_device(start, n, out_host=out_host, out_f32=out_f32, out_off=out_off, out_f64=out_f64)
        buf = self._render_f64_device(start, n)
        N.check(N.lib().sh_ew_f64(N.SH_EW_COPY, buf.handle, 0, None, 0, n, 0.0, 0.0,
                                  out_f64.handle if out_f64 is not None else None, 0,
                                  out_f32.handle if out_f32 is not None else None, out_off,
                                  out_host.ctypes.data if out_host is not None else None))
        buf.free()


# ---------------------------------------------------------------------------------------------------
# Filters over oscillator blocks.  They are not voices of a bank: each
# renders its sources as float64 blocks in HBM and combines them with an elementwise kernel.
# ---------------------------------------------------------------------------------------------------

class _Filter(Oscillator):
    def __init__(self, sources: Sequence[Oscillator]) -> None:
        super().__init__(sources[0].samplerate)
        self._sources = list(sources)

    def spec(self) -> VoiceSpec:
        raise NotImplementedError("%s is a filter over rendered blocks, not a voice of a bank" % type(self).__name__)

    @property
    def length(self) -> Optional[int]:
        lens = [s.length for s in self._sources if s.length is not None]
        return min(lens) if lens else None

    def _f64(self, start: int, n: int) -> N.DeviceBuffer:
        raise NotImplementedError

    def _render_f64_device(self, start: int, n: int) -> N.DeviceBuffer:
        limit = self.length
        if limit is not None and start + n > limit:
            raise ValueError("stream ended before the requested range")
        return self._f64(start, n)

    def _render_device(self, start, n, out_host=None, out_f32=None, out_off=0, out_f64=None) -> None:
        buf = self._f64(start, n)
        N.check(N.lib().sh_ew_f64(N.SH_EW_COPY, buf.handle, 0, None, 0, n, 0.0, 0.0,
                                  out_f64.handle if out_f64 is not None else None, 0,
                                  out_f32.handle if out_f32 is not None else None, out_off,
                                  out_host.ctypes.data if out_host is not None else None))
        buf.free()

    @staticmethod
    def _ew(op: int, a: N.DeviceBuffer, b: Optional[N.DeviceBuffer], n: int, p0: float = 0.0, p1: float = 0.0) -> N.DeviceBuffer:
        """a = op(a, b) in place."""
        N.check(N.lib().sh_ew_f64(op, a.handle, 0, b.handle if b is not None else None, 0, n, p0, p1,
                                  a.handle, 0, None, 0, None))
        return a


class MixingFilter(_Filter):
    """Mixes (adds) the waves of several sources (upstream: oscillators.py class MixingFilter).
    Per sample ``sum(values)``: left to right, like Python's sum."""

    def __init__(self, *sources: Oscillator) -> None:
        super().__init__(sources)

    def _f64(self, start: int, n: int) -> N.DeviceBuffer:
        acc = self._sources[0]._render_f64_device(start, n)
        for src in self._sources[1:]:
            other = src._render_f64_device(start, n)
            self._ew(N.SH_EW_ADD, acc, other, n)
            other.free()
        return acc


class AmpModulationFilter(_Filter):
    """Modulates the amplitude of the source by another oscillator (upstream: class AmpModulationFilter)."""

    def __init__(self, source: Oscillator, modulator: Oscillator) -> None:
        super().__init__([source, modulator])

    def _f64(self, start: int, n: int) -> N.DeviceBuffer:
        acc = self._sources[0]._render_f64_device(start, n)
        mod = self._sources[1]._render_f64_device(start, n)
        self._ew(N.SH_EW_MUL, acc, mod, n)
        mod.free()
        return acc


class ClipFilter(_Filter):
    """Clips the source between a minimum and a maximum: max(min(v, maximum), minimum) (upstream: class ClipFilter)."""

    def __init__(self, source: Oscillator, minimum: float = -1.0, maximum: float = 1.0) -> None:
        super().__init__([source])
        self.min = minimum
        self.max = maximum

    def _f64(self, start: int, n: int) -> N.DeviceBuffer:
        return self._ew(N.SH_EW_CLIP, self._sources[0]._render_f64_device(start, n), None, n, float(self.min), float(self.max))


class AbsFilter(_Filter):
    """Returns the absolute value of the source (upstream: class AbsFilter)."""

    def __init__(self, source: Oscillator) -> None:
        super().__init__([source])

    def _f64(self, start: int, n: int) -> N.DeviceBuffer:
        return self._ew(N.SH_EW_ABS, self._sources[0]._render_f64_device(start, n), None, n)


class NullFilter(_Filter):
    """Passes the source through unchanged (upstream: class NullFilter)."""

    def __init__(self, source: Oscillator) -> None:
        super().__init__([source])

    def _f64(self, start: int, n: int) -> N.DeviceBuffer:
        return self._sources[0]._render_f64_device(start, n)


class DelayFilter(_Filter):
    """Delays the source by a number of seconds (zeros first), or skips into it when negative
    (upstream: class DelayFilter)."""

    def __init__(self, source: Oscillator, seconds: float) -> None:
        super().__init__([source])
        self._seconds = seconds

    def spec(self) -> VoiceSpec:
        """A delayed single-record voice (a waveform, a fused envelope over one; no row-reading modulators) IS a single record:
        the same voice with an onset (sh_voice::start_frame) -- silent before, its own sample 0 there.  So a bank can hold notes
        that start at different times without rendering each into a row of its own."""
        src = self._sources[0]
        if self._seconds < 0.0:
            raise NotImplementedError("a negative delay skips into the source: rendered block by block")
        sp = src.spec()                              # NotImplementedError for filter graphs: rendered block by block
        if sp.fm_mode == N.SH_FM_BUFFER or sp.needs_pwm or sp.kind == N.SH_BUFFER:
            raise NotImplementedError("a delayed voice that reads modulator rows is rendered block by block")
        if sp.env is not None and sp.env.stop_at_end:
            raise NotImplementedError("a delayed finite stream is rendered block by block")
        return _with(sp, start_frame=sp.start_frame + self._shift)

    @property
    def _shift(self) -> int:
        if self._seconds > 0.0:
            return int(self.samplerate * self._seconds)
        return -int(-self.samplerate * self._seconds) if self._seconds < 0.0 else 0

    @property
    def length(self) -> Optional[int]:
        src = self._sources[0].length
        return None if src is None else max(0, src + self._shift)

    def _f64(self, start: int, n: int) -> N.DeviceBuffer:
        d = self._shift
        out = N.DeviceBuffer(n * 8)
        nzero = min(n, max(0, d - start))                 # output samples that fall in the leading silence
        L = N.lib()
        if nzero:
            N.check(L.sh_ew_f64(N.SH_EW_FILL, None, 0, None, 0, nzero, 0.0, 0.0, out.handle, 0, None, 0, None))
        if n > nzero:
            src = self._sources[0]._render_f64_device(start + nzero - d, n - nzero)
            N.check(L.sh_ew_f64(N.SH_EW_COPY, src.handle, 0, None, 0, n - nzero, 0.0, 0.0, out.handle, nzero, None, 0, None))
            src.free()
        return out


class EchoFilter(_Filter):
    """Mixes `amount` echos of the source into itself, starting `after` seconds in: echo i (1-based) is the source
    from that point on, delayed by int(samplerate * (delay + ... + delay)) samples (i terms, accumulated) and scaled
    by decay**i (a running product); per sample the values are summed left to right
    (upstream: oscillators.py class EchoFilter)."""

    def __init__(self, source: Oscillator, after: float, amount: int, delay: float, decay: float) -> None:
        super().__init__([source])
        if decay < 0 or decay > 1:
            raise ValueError("decay should be 0-1")
        self._after = after
        self._amount = amount
        self._delay = delay
        self._decay = decay
        self.echo_duration = self._after + self._amount * self._delay

    def _taps(self) -> List[Tuple[int, float]]:
        """(first sample index the echo sounds at - its source index offset, amplitude) per echo."""
        taps = []
        amp = self._decay
        echo_delay = self._delay
        for _ in range(max(0, self._amount)):
            taps.append((int(self.samplerate * echo_delay), amp))
            echo_delay += self._delay
            amp *= self._decay
        return taps

    def _f64(self, start: int, n: int) -> N.DeviceBuffer:
        src = self._sources[0]
        acc = src._render_f64_device(start, n)
        after = int(self.samplerate * self._after)
        L = N.lib()
        for shift, amp in self._taps():
            first = max(start, after + shift)             # the echo is silent before `after` + its delay
            count = start + n - first
            if count <= 0:
                continue
            echo = src._render_f64_device(first - shift, count)
            N.check(L.sh_ew_f64(N.SH_EW_AXPY, acc.handle, first - start, echo.handle, 0, count, float(amp), 0.0,
                                acc.handle, first - start, None, 0, None))
            echo.free()
        return acc
