"""ctypes front end of oracle/liboracle.so (oracle/oracle.c): the same restatement as
oracle/synth_oracle.py / oracle/pcm_oracle.py, in C, for sizes Python cannot finish in seconds.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  ``render(osc, n)`` takes an *oracle* oscillator object
(synth_oracle.Sine / Sawtooth / Square / Triangle / Pulse / Harmonics, optionally FM'd by a plain Sine; Linear;
WhiteNoise; any of them optionally inside an EnvelopeFilter) and returns its first n samples as float64.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from math import pi
from pathlib import Path

import numpy as np

from . import synth_oracle as O

HERE = Path(__file__).resolve().parent
_lib = None
_D = C.POINTER(C.c_double)
_KIND = {O.Sine: 0, O.Sawtooth: 1, O.Square: 2, O.Pulse: 3, O.Harmonics: 4, O.Triangle: 5}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        so = HERE / "liboracle.so"
        if not so.exists() or so.stat().st_mtime < (HERE / "oracle.c").stat().st_mtime:
            subprocess.run(["make", "-C", str(HERE), "-s"], check=True)
        _lib = C.CDLL(str(so))
        _lib.or_ratecv.restype = C.c_size_t
        _lib.or_ratecv_f32.restype = C.c_size_t
        _lib.or_quantise.restype = C.c_size_t
    return _lib


def _dp(a):
    return a.ctypes.data_as(_D)


def _c_code_knows_the_variants(osc) -> bool:
    """oracle.c restates the DEFAULT readings of synth_oracle.VARIANTS (the increment is computed here, so both of its readings are
    served); a case that touches another reading goes through the Python generators."""
    v = O.VARIANTS
    inner = osc._source if isinstance(osc, O.EnvelopeFilter) else osc
    if isinstance(osc, O.EnvelopeFilter) and v["envelope"] != "lt":
        return False
    if isinstance(inner, O.Square) and v["square"] != "int2":
        return False
    if isinstance(inner, O.Pulse) and v["pulse"] != "lt":
        return False
    return True


def render(osc, n: int) -> np.ndarray:
    if not _c_code_knows_the_variants(osc):
        return np.array(osc.take(n), dtype=np.float64)
    env = None
    if isinstance(osc, O.EnvelopeFilter):
        env, osc = osc, osc._source
    if isinstance(osc, (O.Linear, O.WhiteNoise)):
        out = np.empty(n, dtype=np.float64)
        if isinstance(osc, O.Linear):
            lib().or_linear(C.c_double(osc._value), C.c_double(osc._increment), C.c_double(osc._min), C.c_double(osc._max),
                            C.c_size_t(n), _dp(out))
        else:
            cycles = int(osc.samplerate / osc.frequency)
            if cycles < 1:
                raise ValueError("whitenoise frequency cannot be bigger than the sample rate")
            lib().or_white_noise(C.c_uint64(cycles), C.c_double(osc.amplitude), C.c_double(osc.bias),
                                 C.c_uint64(osc.seed & ((1 << 64) - 1)), C.c_size_t(n), _dp(out))
        if env is not None:
            lib().or_envelope(C.c_double(env._attack), C.c_double(env._decay), C.c_double(env._sustain), C.c_double(env._sustain_level),
                              C.c_double(env._release), C.c_int(osc.samplerate), C.c_size_t(n), _dp(out))
        return out
    kind = _KIND[type(osc)]
    radians = kind in (0, 4)
    sr = osc.samplerate
    out = np.empty(n, dtype=np.float64)
    hk = np.array([float(k) for k, _ in getattr(osc, "harmonics", [])] or [0.0], dtype=np.float64)
    ha = np.array([float(a) for _, a in getattr(osc, "harmonics", [])] or [0.0], dtype=np.float64)
    nh = len(getattr(osc, "harmonics", []))
    pw = float(getattr(osc, "pulsewidth", 0.0))
    L = lib()
    if osc.fm is None and getattr(osc, "pwm", None) is None:
        inc = O._increment(osc.frequency, sr, radians)
        t0 = osc._phase * 2.0 * pi if radians else osc._phase
        L.or_osc_plain(kind, C.c_double(t0), C.c_double(inc), C.c_double(osc.amplitude), C.c_double(osc.bias),
                       C.c_double(pw), _dp(hk), _dp(ha), nh, C.c_size_t(n), _dp(out))
    else:
        lfo = osc.fm
        assert type(lfo) is O.Sine and lfo.fm is None and getattr(osc, "pwm", None) is None
        if radians:
            phase0, inc = osc._phase * 2.0 * pi, 2.0 * pi / sr
        else:
            phase0, inc = osc._phase, 1.0 / sr
        L.or_osc_fm_sine(kind, C.c_double(osc.frequency), C.c_double(phase0), C.c_double(inc), C.c_double(osc.amplitude),
                         C.c_double(osc.bias), C.c_double(pw), _dp(hk), _dp(ha), nh,
                         C.c_double(lfo._phase * 2.0 * pi), C.c_double(O._increment(lfo.frequency, lfo.samplerate, True)),
                         C.c_double(lfo.amplitude), C.c_double(lfo.bias), C.c_size_t(n), _dp(out))
    if env is not None:
        L.or_envelope(C.c_double(env._attack), C.c_double(env._decay), C.c_double(env._sustain),
                      C.c_double(env._sustain_level), C.c_double(env._release), int(env.samplerate), C.c_size_t(n), _dp(out))
    return out


def render_window(osc, start: int, n: int) -> np.ndarray:
    """Samples [start, start + n) of a non-FM waveform oscillator, optionally inside an EnvelopeFilter whose SUSTAIN holds the whole
    window: the generator's loop entered at sample `start` -- its accumulated phase brought there by `start` additions (or_advance),
    the envelope's sustain loop (`next(src) * sustain_level`) applied -- instead of `start` samples rendered and thrown away."""
    env = None
    if isinstance(osc, O.EnvelopeFilter):
        env, osc = osc, osc._source
        sr = osc.samplerate
        assert (start - 2) / sr > env._attack + env._decay and (start + n + 2) / sr < env._attack + env._decay + env._sustain, "window outside the sustain"
        assert not env._cycle
    assert osc.fm is None and getattr(osc, "pwm", None) is None and O.VARIANTS["increment"] in ("mul", "div")
    kind = _KIND[type(osc)]
    radians = kind in (0, 4)
    L = lib()
    L.or_advance.restype = C.c_double
    inc = O._increment(osc.frequency, osc.samplerate, radians)
    t0 = osc._phase * 2.0 * pi if radians else osc._phase
    t = L.or_advance(C.c_double(t0), C.c_double(inc), C.c_size_t(start))
    hk = np.array([float(k) for k, _ in getattr(osc, "harmonics", [])] or [0.0], dtype=np.float64)
    ha = np.array([float(a) for _, a in getattr(osc, "harmonics", [])] or [0.0], dtype=np.float64)
    out = np.empty(n, dtype=np.float64)
    L.or_osc_plain(kind, C.c_double(t), C.c_double(inc), C.c_double(osc.amplitude), C.c_double(osc.bias),
                   C.c_double(float(getattr(osc, "pulsewidth", 0.0))), _dp(hk), _dp(ha), len(getattr(osc, "harmonics", [])), C.c_size_t(n), _dp(out))
    if env is not None:
        out *= env._sustain_level
    return out


def mix_bus(voices: np.ndarray, gains) -> np.ndarray:
    voices = np.ascontiguousarray(voices, dtype=np.float64)
    nv, n = voices.shape
    gl = np.array([g[0] for g in gains], dtype=np.float64)
    gr = np.array([g[1] for g in gains], dtype=np.float64)
    bus = np.empty((n, 2), dtype=np.float64)
    lib().or_mix_bus(_dp(voices), C.c_size_t(nv), C.c_size_t(n), _dp(gl), _dp(gr), _dp(bus))
    return bus


def add(a: bytes, b: bytes, width: int) -> bytes:
    assert len(a) == len(b)
    out = C.create_string_buffer(len(a))
    lib().or_add(a, b, C.c_size_t(len(a)), width, out)
    return out.raw


def ratecv(frames: bytes, width: int, nch: int, inrate: int, outrate: int) -> bytes:
    nfr = len(frames) // (width * nch)
    from .pcm_oracle import ratecv_out_frames
    cap = ratecv_out_frames(nfr, inrate, outrate)
    out = C.create_string_buffer(max(1, cap * width * nch))
    got = lib().or_ratecv(frames, C.c_size_t(nfr), width, nch, inrate, outrate, out)
    assert got == cap
    return out.raw[:got * width * nch]


def ratecv_f32(x: np.ndarray, inrate: int, outrate: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    nfr, nch = x.shape
    from .pcm_oracle import ratecv_out_frames
    cap = ratecv_out_frames(nfr, inrate, outrate)
    out = np.empty((cap, nch), dtype=np.float32)
    got = lib().or_ratecv_f32(x.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(nfr), nch, inrate, outrate,
                              out.ctypes.data_as(C.POINTER(C.c_float)))
    assert got == cap
    return out


def quantise(v: np.ndarray, scale: float = 32767.0, width: int = 2) -> np.ndarray:
    if O.VARIANTS["quantise"] != "trunc":
        return np.array(O.quantise(np.asarray(v, dtype=np.float64).tolist(), width, scale), dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.empty(len(v), dtype=np.int32)
    bad = lib().or_quantise(_dp(v), C.c_size_t(len(v)), C.c_double(scale), width, out.ctypes.data_as(C.POINTER(C.c_int32)))
    if bad:
        raise OverflowError("sample %d does not fit %d bytes" % (bad - 1, width))
    return out
