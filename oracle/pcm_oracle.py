"""
CPU ORACLE for the integer PCM rows: Sample.mix and Sample.resample.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/synth_oracle.py).

Upstream synthplayer delegates both operations to a third-party dependency that
is NOT under /root/reference: the CPython standard-library C module ``audioop``
(Modules/audioop.c).  Pinned version: the interpreter of this image,
**CPython 3.10.12** (audioop was removed in 3.13).  Call sites upstream
(uncitable, tree not mounted): sample.py ``Sample.mix`` -> ``audioop.add``;
``Sample.resample`` -> ``audioop.ratecv(frames, width, nchannels, inrate,
outrate, None)``; playback.py mixer -> repeated ``audioop.add``.

PARITY STATUS: **pinned against the live dependency**.  tests/test_oracle_pcm.py
checks every function below against ``audioop`` itself (available on the GPU box
too - same image) and against the golden vectors in tests/golden/ that
tests/golden/make_golden.py generated from ``audioop`` in this container.

Restated algorithms (CPython 3.10 Modules/audioop.c):

* ``audioop_add_impl``: per sample, width < 4: ``int newval = val1 + val2``
  clamped to [minval, maxval]; width 4: the sum is formed in double and clamped
  by ``fbound``.  Both fragments must have the same length.
* ``audioop_ratecv_impl``: rates are divided by their gcd; samples are widened
  to 32 bits (``GETSAMPLE32``: value << (32 - 8*width)); the state ``d`` starts
  at ``-outrate``; each consumed input frame adds ``outrate``; while ``d >= 0``
  an output frame ``(int)((prev*(double)d + cur*(double)(outrate-d)) /
  (double)outrate)`` is emitted (then narrowed with an arithmetic right shift)
  and ``d -= inrate``.  With the default weightA=1, weightB=0 the "simple
  digital filter" is the identity.
"""
from __future__ import annotations

from math import gcd
from typing import List, Sequence, Tuple

import numpy as np

_DT = {1: np.int8, 2: np.int16, 4: np.int32}


def _decode(frames: bytes, width: int) -> np.ndarray:
    """little-endian signed PCM -> int64 array of sample values."""
    if width == 3:
        b = np.frombuffer(frames, dtype=np.uint8).reshape(-1, 3).astype(np.int64)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        return np.where(v >= 1 << 23, v - (1 << 24), v)
    return np.frombuffer(frames, dtype=np.dtype(_DT[width]).newbyteorder("<")).astype(np.int64)


def _encode(vals: np.ndarray, width: int) -> bytes:
    if width == 3:
        v = vals.astype(np.int64) & 0xFFFFFF
        b = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=1).astype(np.uint8)
        return b.tobytes()
    return vals.astype(np.dtype(_DT[width]).newbyteorder("<")).tobytes()


def add(frag1: bytes, frag2: bytes, width: int) -> bytes:
    """audioop.add: saturating per-sample sum (audioop.c audioop_add_impl)."""
    if len(frag1) != len(frag2):
        raise ValueError("Lengths should be the same")
    if len(frag1) % width:
        raise ValueError("not a whole number of frames")
    a = _decode(frag1, width)
    b = _decode(frag2, width)
    lo, hi = -(1 << (8 * width - 1)), (1 << (8 * width - 1)) - 1
    return _encode(np.clip(a + b, lo, hi), width)


def add_chain(frags: Sequence[bytes], width: int) -> bytes:
    """The real-time mixer's fold: mixed = add(add(add(f0, f1), f2), ...) in order
    (upstream: playback.py mixer loop).  Saturation makes the order significant."""
    mixed = frags[0]
    for f in frags[1:]:
        mixed = add(mixed, f, width)
    return bytes(mixed)


def ratecv_sequential(frames: bytes, width: int, nchannels: int, inrate: int, outrate: int) -> bytes:
    """audioop.ratecv(frames, width, nchannels, inrate, outrate, None)[0], restated
    statement by statement (state machine with ``d``).  Pure Python: small inputs only."""
    if len(frames) % (width * nchannels):
        raise ValueError("not a whole number of frames")
    g = gcd(inrate, outrate)
    inrate //= g
    outrate //= g
    x = _decode(frames, width).reshape(-1, nchannels)
    shift = 32 - 8 * width
    nframes = x.shape[0]
    prev_i = [0] * nchannels
    cur_i = [0] * nchannels
    d = -outrate
    out: List[int] = []
    pos = 0
    while True:
        while d < 0:
            if pos == nframes:
                return _encode(np.array(out, dtype=np.int64), width)
            for c in range(nchannels):
                prev_i[c] = cur_i[c]
                cur_i[c] = int(x[pos, c]) << shift
            pos += 1
            d += outrate
        while d >= 0:
            for c in range(nchannels):
                cur_o = int((float(prev_i[c]) * float(d) + float(cur_i[c]) * float(outrate - d)) / float(outrate))
                out.append(cur_o >> shift)
            d -= inrate


def ratecv_out_frames(in_frames: int, inrate: int, outrate: int) -> int:
    """Number of output frames audioop.ratecv produces from a fresh state."""
    if in_frames <= 0:
        return 0
    g = gcd(inrate, outrate)
    inrate //= g
    outrate //= g
    # output m exists iff ceil(m*inrate/outrate) <= in_frames-1
    return ((in_frames - 1) * outrate) // inrate + 1


def ratecv(frames: bytes, width: int, nchannels: int, inrate: int, outrate: int) -> bytes:
    """Closed form of ratecv_sequential, vectorised with numpy (same arithmetic:
    float64 products, float64 sum, float64 division, truncation, arithmetic shift).
    Output m reads input frames j-1 and j with j = ceil(m*inrate/outrate) and
    d = j*outrate - m*inrate."""
    if len(frames) % (width * nchannels):
        raise ValueError("not a whole number of frames")
    g = gcd(inrate, outrate)
    inr = inrate // g
    outr = outrate // g
    x = _decode(frames, width).reshape(-1, nchannels)
    nframes = x.shape[0]
    nout = ratecv_out_frames(nframes, inrate, outrate)
    if nout == 0:
        return b""
    shift = 32 - 8 * width
    m = np.arange(nout, dtype=np.int64)
    j = (m * inr + outr - 1) // outr
    d = j * outr - m * inr
    cur = (x[j] << shift).astype(np.float64)
    prev = np.where((j > 0)[:, None], x[np.maximum(j - 1, 0)] << shift, 0).astype(np.float64)
    dd = d.astype(np.float64)[:, None]
    val = (prev * dd + cur * (float(outr) - dd)) / float(outr)
    o = np.trunc(val).astype(np.int64) >> shift
    return _encode(o.reshape(-1), width)


def ratecv_f32(x: np.ndarray, inrate: int, outrate: int) -> np.ndarray:
    """float32 PCM resample for BASELINE config 5 ([SPEC]; upstream Sample is integer
    only, so this row has no upstream oracle): identical index arithmetic (j, d) to
    ratecv, value = (prev*d + cur*(outrate-d))/outrate in float64, rounded to float32.
    x: [frames, nchannels] float32."""
    g = gcd(inrate, outrate)
    inr = inrate // g
    outr = outrate // g
    nframes = x.shape[0]
    nout = ratecv_out_frames(nframes, inrate, outrate)
    if nout == 0:
        return np.zeros((0, x.shape[1]), dtype=np.float32)
    m = np.arange(nout, dtype=np.int64)
    j = (m * inr + outr - 1) // outr
    d = j * outr - m * inr
    cur = x[j].astype(np.float64)
    prev = np.where((j > 0)[:, None], x[np.maximum(j - 1, 0)], 0).astype(np.float64)
    dd = d.astype(np.float64)[:, None]
    val = (prev * dd + cur * (float(outr) - dd)) / float(outr)
    return val.astype(np.float32)


def ratecv_f32_window(get_frames, inrate: int, outrate: int, m_first: int, m_count: int) -> np.ndarray:
    """The same closed form for output frames m_first .. m_first + m_count - 1 only, of an input too long to hold twice:
    get_frames(idx) returns the input frames idx (int64 array) as [len(idx), nchannels] float32.  Used to check the tail
    of BASELINE config 5 at its full size (57.6 M input frames)."""
    g = gcd(inrate, outrate)
    inr = inrate // g
    outr = outrate // g
    m = np.arange(m_first, m_first + m_count, dtype=np.int64)
    j = (m * inr + outr - 1) // outr
    d = j * outr - m * inr
    cur = get_frames(j).astype(np.float64)
    prev = np.where((j > 0)[:, None], get_frames(np.maximum(j - 1, 0)), 0).astype(np.float64)
    dd = d.astype(np.float64)[:, None]
    val = (prev * dd + cur * (float(outr) - dd)) / float(outr)
    return val.astype(np.float32)


# ---------------------------------------------------------------------------------------------------
# the other audioop functions Sample delegates to, restated from
# CPython 3.10 Modules/audioop.c and checked against the live module in tests/test_oracle_pcm.py
# ---------------------------------------------------------------------------------------------------

def _fbound(val: np.ndarray, width: int) -> np.ndarray:
    """audioop.c fbound(): clamp to [minval, maxval] (values below minval+1 go to minval), then floor."""
    lo, hi = float(-(1 << (8 * width - 1))), float((1 << (8 * width - 1)) - 1)
    v = np.where(val > hi, hi, np.where(val < lo + 1.0, lo, val))
    return np.floor(v).astype(np.int64)


def mul(frames: bytes, width: int, factor: float) -> bytes:
    return _encode(_fbound(_decode(frames, width).astype(np.float64) * float(factor), width), width)


def bias(frames: bytes, width: int, bias_value: int) -> bytes:
    v = (_decode(frames, width) + int(bias_value)) & ((1 << (8 * width)) - 1)
    v = np.where(v >= 1 << (8 * width - 1), v - (1 << (8 * width)), v)
    return _encode(v, width)


def reverse(frames: bytes, width: int) -> bytes:
    return _encode(_decode(frames, width)[::-1], width)


def tomono(frames: bytes, width: int, lfactor: float, rfactor: float) -> bytes:
    x = _decode(frames, width).reshape(-1, 2).astype(np.float64)
    return _encode(_fbound(x[:, 0] * float(lfactor) + x[:, 1] * float(rfactor), width), width)


def tostereo(frames: bytes, width: int, lfactor: float, rfactor: float) -> bytes:
    x = _decode(frames, width).astype(np.float64)
    out = np.stack([_fbound(x * float(lfactor), width), _fbound(x * float(rfactor), width)], axis=1)
    return _encode(out.reshape(-1), width)


def lin2lin(frames: bytes, width: int, new_width: int) -> bytes:
    v32 = _decode(frames, width) << (32 - 8 * width)
    return _encode(v32 >> (32 - 8 * new_width), new_width)


def amax(frames: bytes, width: int) -> int:
    x = _decode(frames, width)
    return int(np.max(np.abs(x))) if len(x) else 0


def rms(frames: bytes, width: int) -> int:
    x = _decode(frames, width)
    if len(x) == 0:
        return 0
    total = 0.0
    for v in x.tolist():            # audioop: sequential float64 sum of squares
        total += float(v) * float(v)
    return int(np.sqrt(total / float(len(x))))


def fade(frames: bytes, width: int, fadeout: bool, slope: float, offset: float) -> bytes:
    """Sample.fadeout / fadein inner expression (upstream sample.py, recalled): per sample index i,
    int(sample * (1.0 - i*slope/numsamples)) or int(sample * (i*slope/numsamples + offset))."""
    x = _decode(frames, width)
    numsamples = len(frames) / width
    out = []
    for i, v in enumerate(x.tolist()):
        f = (1.0 - i * slope / numsamples) if fadeout else (i * slope / numsamples + offset)
        out.append(int(v * f))
    return _encode(np.array(out, dtype=np.int64), width)
