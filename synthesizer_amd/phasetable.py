"""
Exact closed form of a float64 running sum ``t += inc``.

The reference oscillators (synthplayer/oscillators.py, every ``blocks()`` loop)
never recompute the phase from the sample index: they carry ``t`` and add a
constant increment once per sample, in float64.  The rounding of that addition
is deterministic, and inside one binade [2^e, 2^(e+1)) it is *constant*: ``t``
is a multiple of the binade's ulp ``u``, so ``fl(t + inc) = t + q*u`` with the
same integer ``q`` for every step after the first one in the binade (a tie
``inc mod u == u/2`` is resolved to the even neighbour, after which the parity
of ``t/u`` and therefore the choice stays fixed).

So the whole sequence t_0, t_1, ... is piecewise *exactly* linear in the sample
index, with one piece per binade (plus a few single-step pieces where a binade
is entered).  ``build_phase_table`` returns those pieces; a GPU thread that owns
sample ``n`` evaluates ``fma(n - n0, dt, t0)`` on its piece and obtains the very
same double the reference's sequential loop holds at that sample -- which is
what decides the side of a Square/Pulse edge and the 1e-5 rad drift of a
long-running Sine.

Host logic only.  tests/test_phasetable.py checks the table against brute-force
accumulation, and the native form of the same loop (libsynthhost.so, include/synthhost.h:
what a PhaseTable is built by when the library can be had) against this one.
"""
from __future__ import annotations

import ctypes as C
from bisect import bisect_right
from fractions import Fraction
from math import frexp, ceil, ldexp
from typing import List, Optional, Tuple

import numpy as np

Segment = Tuple[int, float, float]      # (n0, t0, dt): t_n = t0 + (n-n0)*dt exactly, n0 <= n < next n0

N_LIMIT = 1 << 62
_TINY = 2.0 ** -1000


_LO, _HI = 1 << 52, (1 << 53) - 1
_MAX_PIECES = 1 << 16


def build_phase_table(t0: float, inc: float, n_limit: int = N_LIMIT) -> List[Segment]:
    """Pieces of the sequence t_0 = t0, t_{n+1} = fl(t_n + inc), covering [0, n_limit).
    (The binade tests are written out as comparisons against the binade's bounds: this loop is most of what creating a bank costs
    on the host -- 150 pieces per distinct (phase, frequency) -- and calls of Python helpers were most of the loop.)"""
    segs: List[Segment] = []
    append = segs.append
    n, t = 0, float(t0)
    inc = float(inc)
    while n < n_limit:
        t1 = t + inc
        if t1 == t:                       # inc == 0, or absorbed below half an ulp: constant forever
            append((n, t, 0.0))
            break
        if t >= _TINY or t <= -_TINY:     # (zero and tiny values are never part of a run)
            m, e = frexp(t)
            # the binade of t: lo <= |x| < hi, with t's sign (|t| >= 2^1023: hi = 2^1024 is no float64 -- math.ldexp raises where C's
            # ldexp returns inf; with inf, as in the native loop, no step stays "inside" and the table goes on by single steps)
            hi = ldexp(1.0, e) if e < 1024 else float("inf")
            lo = 0.5 * hi
            pos = t > 0
            if (lo <= t1 < hi) if pos else (-hi < t1 <= -lo):
                t2 = t1 + inc
                d = t1 - t                # exact: same binade
                if ((lo <= t2 < hi) if pos else (-hi < t2 <= -lo)) and (t2 - t1) == d:
                    # regular run: t + j*d for j = 0..k stays inside the binade.  t = T * 2**s with 2^52 <= |T| < 2^53;
                    # d in units of 2**s: d is a multiple of the binade's ulp and |d| < 2**(s + 53), so the scaling is exact
                    s = e - 53
                    T = int(ldexp(m, 53))
                    D = int(ldexp(d, -s))
                    if T > 0:
                        k = (_HI - T) // D if D > 0 else (T - _LO) // (-D)
                    else:
                        k = (_HI + T) // (-D) if D < 0 else (-T - _LO) // D
                    append((n, t, d))
                    tk = ldexp(float(T + k * D), s)       # exact: |T + k*D| < 2^53
                    t1 = tk + inc                          # the step out of the binade: a real addition
                    if t1 == tk:
                        append((n + k, tk, 0.0))
                        break
                    n, t = n + k + 1, t1
                    continue
        # single step (entering/leaving a binade, crossing zero, irregular first step)
        append((n, t, t1 - t))
        n, t = n + 1, t1
        if len(segs) > _MAX_PIECES:
            # (a denormal increment from a denormal start walks 2^70 single steps below the smallest binade this table keeps)
            raise OverflowError("phase table of t0=%r, increment=%r does not close: more than %d single-step pieces (a denormal "
                                "increment from a denormal start, or values that are not finite)" % (t0, inc, _MAX_PIECES))
    return segs


# ---- the same loop in native code (libsynthhost.so, include/synthhost.h): ~2 us per table instead of ~110 ------------------------
_SEGMENT_DTYPE = np.dtype([("n0", "<u8"), ("t0", "<f8"), ("dt", "<f8")], align=True)      # = _native.SEGMENT_DTYPE = shh_segment
_host = None           # None: not tried yet; False: unavailable (the Python loop is used, said once); else the CDLL


def _host_lib():
    global _host
    if _host is None:
        try:
            from . import build as B
            lib = C.CDLL(str(B.build_host()))
            lib.shh_phase_table.restype = C.c_int
            lib.shh_phase_table.argtypes = [C.c_double, C.c_double, C.c_uint64, C.c_void_p, C.c_int]
            lib.shh_version.restype = C.c_char_p
            _host = lib
        except Exception as exc:      # no compiler, a read-only tree ...: host logic has a tested Python statement to fall back on
            import warnings
            warnings.warn("libsynthhost.so unavailable (%s): phase tables are built by the Python loop" % (exc,))
            _host = False
    return _host


def phase_table_records(t0: float, inc: float, n_limit: int = N_LIMIT) -> np.ndarray:
    """The pieces of build_phase_table as an array of (n0, t0, dt) records in the C layout, from the native loop when the host
    library can be had (identical pieces: tests/test_phasetable.py), else from the Python one."""
    lib = _host_lib()
    if lib:
        cap = 256
        while True:
            out = np.empty(cap, dtype=_SEGMENT_DTYPE)
            cnt = lib.shh_phase_table(float(t0), float(inc), min(int(n_limit), (1 << 64) - 1), out.ctypes.data, cap)
            if cnt >= 0:
                return out[:cnt].copy()
            if cnt == -2:
                raise OverflowError("phase table of t0=%r, increment=%r does not close: more than %d single-step pieces (a denormal "
                                    "increment from a denormal start, or values that are not finite)" % (t0, inc, _MAX_PIECES))
            if cap > (1 << 17):
                raise OverflowError("phase table of t0=%r, increment=%r needs more than %d pieces" % (t0, inc, cap))
            cap *= 16
    segs = build_phase_table(t0, inc, n_limit)
    out = np.empty(len(segs), dtype=_SEGMENT_DTYPE)
    for i, seg in enumerate(segs):
        out[i] = seg
    return out


class PhaseTable:
    """Host-side view used for envelope boundaries, by the packing of a bank's tables (`records`) and by the tests."""

    def __init__(self, t0: float, inc: float, n_limit: int = N_LIMIT) -> None:
        self.t0 = float(t0)
        self.inc = float(inc)
        self._records: Optional[np.ndarray] = phase_table_records(t0, inc, n_limit)
        self._segments: Optional[List[Segment]] = None
        self._starts_cache: Optional[List[int]] = None

    @property
    def records(self) -> np.ndarray:
        """(n0, t0, dt) records in the C layout (sh_segment)."""
        if self._records is None:
            out = np.empty(len(self._segments), dtype=_SEGMENT_DTYPE)
            for i, seg in enumerate(self._segments):
                out[i] = seg
            self._records = out
        return self._records

    @property
    def segments(self) -> List[Segment]:
        if self._segments is None:
            r = self._records
            self._segments = list(zip(r["n0"].tolist(), r["t0"].tolist(), r["dt"].tolist()))
        return self._segments

    @segments.setter
    def segments(self, value: List[Segment]) -> None:
        self._segments = list(value)
        self._records = None
        self._starts_cache = None

    @property
    def _starts(self) -> List[int]:
        if self._starts_cache is None:
            self._starts_cache = [s[0] for s in self.segments]
        return self._starts_cache

    @_starts.setter
    def _starts(self, value: List[int]) -> None:
        self._starts_cache = list(value)

    def __len__(self) -> int:
        return len(self._records) if self._records is not None else len(self._segments)

    def value(self, n: int) -> float:
        """t_n, exactly (what the sequential loop holds when it emits sample n)."""
        i = bisect_right(self._starts, n) - 1
        n0, t0, dt = self.segments[i]
        return float(Fraction(t0) + (n - n0) * Fraction(dt))

    def first_index_ge(self, x: float) -> int:
        """Smallest n with t_n >= x (needs inc > 0: the sequence is non-decreasing)."""
        if self.inc <= 0:
            raise ValueError("first_index_ge needs a positive increment")
        if self.t0 >= x:
            return 0
        X = Fraction(x)
        x = float(x)
        lo, hi = 0, len(self.segments) - 1          # last piece whose start value is < x (doubles compare exactly)
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if self.segments[mid][1] < x:
                lo = mid
            else:
                hi = mid - 1
        n0, t0, dt = self.segments[lo]
        if dt == 0.0:
            raise OverflowError("sequence never reaches %r" % x)
        n = n0 + ceil((X - Fraction(t0)) / Fraction(dt))
        if lo + 1 < len(self.segments):
            n = min(n, self.segments[lo + 1][0])
        return n
