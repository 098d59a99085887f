"""GPU parity: HIP oscillators / envelope / quantise vs the CPU oracle (oracle/synth_oracle.py).

Tolerance ([SPEC] BASELINE.json north_star): float32 oscillator blocks within 1e-6 RMS of the
reference's float64 samples; int16 quantisation bit-exact.  In practice the HIP path reproduces the
float64 value to ~1e-15 before its single rounding to float32, so most checks below are far tighter
than 1e-6: Square/Pulse blocks must be *equal* to float32(oracle) (an edge sample on the wrong side
would cost 2*amplitude), the others within 1 float32 ulp.
"""
import itertools
import math

import numpy as np
import pytest

from oracle import synth_oracle as O
from tests.helpers import rms, accumulated, ulp32_diff

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-6          # the contract
SR = 48000


def _pair(cls_name, *args, **kw):
    from synthesizer_amd import oscillators as G
    return getattr(G, cls_name)(*args, **kw), getattr(O, cls_name)(*args, **kw)


def _check(g, o, n, exact=False, max_ulp=1):
    got = g.render(n, start=0)
    want = np.array(o.take(n), dtype=np.float64)
    assert got.dtype == np.float32 and got.shape == (n,)
    assert rms(got, want) <= RMS_TOL
    want32 = want.astype(np.float32)
    if exact:
        assert np.array_equal(got, want32), int(np.sum(got != want32))
    else:
        # 1 ulp of float32 where the value is not tiny; absolute 1e-7 near zero crossings
        bad = (ulp32_diff(got, want32) > max_ulp) & (np.abs(got.astype(np.float64) - want) > 1e-7)
        assert not bad.any(), int(bad.sum())
    return got, want


def test_config1_sine_440_1s_44k1(gpu):
    """BASELINE configs[0]: single 440 Hz Sine, 1 s @ 44.1 kHz mono."""
    g, o = _pair("Sine", 440, samplerate=44100)
    got, want = _check(g, o, 44100)
    golden = np.load("tests/golden/osc_sine440_44k1.npy")      # samples [0:4096] and [40004:44100]
    assert rms(got[np.r_[0:4096, 40004:44100]], golden) <= RMS_TOL


@pytest.mark.parametrize("kind", ["Sine", "Sawtooth", "Square", "Pulse"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_plain_oscillators(gpu, kind, seed):
    rng = np.random.default_rng(seed)
    f = float(np.exp(rng.uniform(np.log(55), np.log(3520))))
    kw = dict(amplitude=float(rng.uniform(0.1, 1.0)), phase=float(rng.uniform(0, 1)), bias=float(rng.uniform(-0.2, 0.2)),
              samplerate=SR)
    if kind == "Pulse":
        kw["pulsewidth"] = float(rng.uniform(0.05, 0.95))
    g, o = _pair(kind, f, **kw)
    _check(g, o, 30000, exact=kind in ("Square", "Pulse"))


@pytest.mark.parametrize("kind,f,sr", [("Square", 1000, 48000), ("Square", 440, 44100), ("Pulse", 1000, 48000),
                                        ("Sawtooth", 1000, 48000), ("Square", 12000, 48000), ("Pulse", 100, 8000)])
def test_edges_on_rational_frequencies(gpu, kind, f, sr):
    """Frequencies whose edges fall *exactly* on sample instants: the side is decided by the rounding
    of the reference's accumulated t, which the phase tables reproduce."""
    kw = dict(samplerate=sr)
    if kind == "Pulse":
        kw["pulsewidth"] = 0.25
    g, o = _pair(kind, f, **kw)
    _check(g, o, 3 * sr, exact=True)


def test_negative_phase_and_frequency(gpu):
    for kind in ("Square", "Pulse", "Sawtooth", "Sine"):
        g, o = _pair(kind, 333.3, phase=-0.3, samplerate=SR)
        _check(g, o, 5000, exact=kind in ("Square", "Pulse"))
        g, o = _pair(kind, -250.0, phase=0.1, samplerate=SR)
        _check(g, o, 5000, exact=kind in ("Square", "Pulse"))


def test_harmonics_poly_dense_and_sparse(gpu):
    """The three evaluation forms of Harmonics: degree-15 polynomial in cos t (all k <= 16), Clenshaw
    (larger integer k), term by term (non-integer or very sparse k)."""
    partials = [(k, 1.0 / k) for k in range(1, 17)]
    g, o = _pair("Harmonics", 220.0, partials, amplitude=0.4, phase=0.2, bias=0.01, samplerate=SR)
    assert g.spec().harm_poly is not None
    _check(g, o, 20000, max_ulp=2)
    odd = [(1, 1.0), (3, 0.3), (5, 0.2), (9, -0.1), (3, 0.05), (-2, 0.1), (0, 0.5)]
    g, o = _pair("Harmonics", 330.0, odd, samplerate=SR)
    assert g.spec().harm_poly is not None
    _check(g, o, 20000, max_ulp=2)
    big = [(k, 1.0) for k in range(1, 17)]                       # worst case for the monomial form
    g, o = _pair("Harmonics", 97.0, big, amplitude=0.1, samplerate=SR)
    assert g.spec().harm_poly is not None
    _check(g, o, 20000, max_ulp=2)
    many = [(k, 1.0 / k) for k in range(1, 41)]
    g, o = _pair("Harmonics", 110.0, many, amplitude=0.4, samplerate=SR)
    assert g.spec().harm_poly is None and g.spec().harm_dense is not None
    _check(g, o, 20000, max_ulp=2)
    sparse = [(1, 1.0), (1000, 0.05), (2.5, 0.1)]
    g, o = _pair("Harmonics", 5.0, sparse, samplerate=SR)
    assert g.spec().harm_sparse is not None
    _check(g, o, 20000, max_ulp=2)


@pytest.mark.parametrize("kind", ["Sine", "Sawtooth", "Square", "Pulse", "Harmonics"])
def test_fm_sine_lfo_closed_form(gpu, kind):
    from synthesizer_amd import oscillators as G
    rng = np.random.default_rng(7)
    for _ in range(2):
        fm, depth, pm = float(rng.uniform(0.5, 8)), float(rng.uniform(0.005, 0.05)), float(rng.uniform(0, 1))
        f = float(rng.uniform(110, 1760))
        args = (f,) if kind != "Harmonics" else (f, [(k, 1.0 / k) for k in range(1, 9)])
        kw = dict(amplitude=0.8, phase=float(rng.uniform(0, 1)), samplerate=SR)
        g = getattr(G, kind)(*args, fm_lfo=G.Sine(fm, depth, phase=pm, samplerate=SR), **kw)
        o = getattr(O, kind)(*args, fm_lfo=O.Sine(fm, depth, phase=pm, samplerate=SR), **kw)
        assert g.spec().fm_mode == 1
        got = g.render(48000)
        want = np.array(o.take(48000))
        if kind in ("Square", "Pulse"):
            # an FM'd edge can land within rounding of a sample instant only by accident
            assert np.sum(got != want.astype(np.float32)) <= 1
        else:
            assert rms(got, want) <= RMS_TOL
            assert np.max(np.abs(got - want)) < 2e-6


def test_fm_general_modulator_and_pwm(gpu):
    from synthesizer_amd import oscillators as G
    # sawtooth LFO -> buffer path (modulator rendered in float64 on the GPU, scanned, fed to the carrier)
    g = G.Sine(440, fm_lfo=G.Sawtooth(3, 0.03, samplerate=SR), samplerate=SR)
    o = O.Sine(440, fm_lfo=O.Sawtooth(3, 0.03, samplerate=SR), samplerate=SR)
    assert g.spec().fm_mode == 2
    got = np.concatenate([g.render(10000), g.render(7000), g.render(15000)])      # carried state across calls
    want = np.array(o.take(32000))
    assert rms(got, want) <= RMS_TOL
    # FM'd LFO (two levels)
    g = G.Sawtooth(300, fm_lfo=G.Sine(6, 0.05, fm_lfo=G.Sine(0.7, 0.5, samplerate=SR), samplerate=SR), samplerate=SR)
    o = O.Sawtooth(300, fm_lfo=O.Sine(6, 0.05, fm_lfo=O.Sine(0.7, 0.5, samplerate=SR), samplerate=SR), samplerate=SR)
    got, want = g.render(24000), np.array(o.take(24000))
    near_edge = np.abs(np.abs(want) - 1.0) < 1e-3
    assert rms(got[~near_edge], want[~near_edge]) <= RMS_TOL
    # PWM
    g = G.Pulse(220, pulsewidth=0.3, pwm_lfo=G.Sine(2, 0.2, bias=0.5, samplerate=SR), samplerate=SR)
    o = O.Pulse(220, pulsewidth=0.3, pwm_lfo=O.Sine(2, 0.2, bias=0.5, samplerate=SR), samplerate=SR)
    got, want = g.render(24000), np.array(o.take(24000))
    assert np.sum(got != want.astype(np.float32)) <= 2
    # random access into a recurrence: restart from an arbitrary position
    g = G.Sine(440, fm_lfo=G.Sawtooth(3, 0.03, samplerate=SR), samplerate=SR)
    got = g.render(5000, start=20000)
    want = np.array(O.Sine(440, fm_lfo=O.Sawtooth(3, 0.03, samplerate=SR), samplerate=SR).take(25000))[20000:]
    assert rms(got, want) <= RMS_TOL


@pytest.mark.parametrize("adsr", [(0.01, 0.05, 0.5, 0.6, 0.2), (0.0, 0.05, 0.1, 0.6, 0.1), (0.02, 0.0, 0.0, 1.0, 0.05),
                                   (0.013, 0.007, 0.0, 0.3, 0.0), (0.0, 0.0, 0.01, 0.5, 0.0), (0.05, 0.05, 0.05, 0.0, 0.05)])
def test_envelope_filter(gpu, adsr):
    from synthesizer_amd import oscillators as G
    a, d, s, sl, r = adsr
    n = int((a + d + s + r) * SR) + 3000
    g = G.EnvelopeFilter(G.Sine(440, samplerate=SR), a, d, s, sl, r)
    o = O.EnvelopeFilter(O.Sine(440, samplerate=SR), a, d, s, sl, r)
    got, want = g.render(n), np.array(o.take(n))
    assert rms(got, want) <= RMS_TOL
    assert np.max(np.abs(got - want)) < 2e-7
    # stop_at_end: same stream length as the reference generator
    g = G.EnvelopeFilter(G.Square(300, samplerate=SR), a, d, s, sl, r, stop_at_end=True)
    o = O.EnvelopeFilter(O.Square(300, samplerate=SR), a, d, s, sl, r, stop_at_end=True)
    want = np.array(list(itertools.chain.from_iterable(o.blocks())))
    got = np.array(list(itertools.chain.from_iterable(g.blocks())))
    assert got.shape == want.shape
    assert rms(got, want) <= RMS_TOL


@pytest.mark.parametrize("adsr", [(0.01, 0.02, 0.03, 0.6, 0.04), (0.0, 0.005, 0.0, 0.3, 0.011), (0.002, 0.0, 0.001, 1.0, 0.0)])
def test_envelope_filter_cycle(gpu, adsr):
    """cycle=True (SURVEY 8(a) row a8's signature): the phases start over after the release, the source runs on.  Against the
    oracle's generator over several periods, from frame 0, from the middle of a period, block by block, and as a bank voice."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    a, d, s, sl, r = adsr
    period = G.envelope_spec(a, d, s, sl, r, SR, True).length
    n = max(5 * period + 777, 8 * 512 + 5)
    g = G.EnvelopeFilter(G.Sine(440, samplerate=SR), a, d, s, sl, r, cycle=True)
    o = O.EnvelopeFilter(O.Sine(440, samplerate=SR), a, d, s, sl, r, cycle=True)
    want = np.array(o.take(n))
    assert g.length is None
    got = g.render(n, start=0)
    assert rms(got, want) <= RMS_TOL and np.max(np.abs(got - want)) < 2e-7
    f64 = g.render_f64(n, start=0)
    assert np.max(np.abs(f64 - want)) <= 1e-12
    assert np.max(np.abs(want[period:2 * period])) > 0.0            # the second period sounds: this is not stop / silence
    start = 2 * period + period // 3
    assert np.array_equal(g.render(1000, start=start), got[start:start + 1000])
    blocks = list(itertools.islice(g.blocks(), 8))
    flat = np.array(list(itertools.chain.from_iterable(blocks)))
    assert np.max(np.abs(flat - want[:flat.size])) <= 1e-12
    # stop_at_end is never reached while cycling; a delayed cycling envelope goes block by block
    g2 = G.EnvelopeFilter(G.Square(300, samplerate=SR), a, d, s, sl, r, stop_at_end=True, cycle=True)
    o2 = O.EnvelopeFilter(O.Square(300, samplerate=SR), a, d, s, sl, r, stop_at_end=True, cycle=True)
    assert rms(g2.render(n, start=0), np.array(o2.take(n))) <= RMS_TOL
    # a voice of a bank (a row of the launch's matrix) beside a fused one
    voices = [g, G.EnvelopeFilter(G.Sawtooth(220, samplerate=SR), a, d, s, sl, r)]
    gains = [(0.25, 0.75), (0.5, 0.125)]
    bus = VoiceBank(voices, gains=gains).render(n)
    w1 = np.array(O.EnvelopeFilter(O.Sawtooth(220, samplerate=SR), a, d, s, sl, r).take(n))
    ref = np.array(O.mix_bus([want, w1], gains))
    assert rms(bus, ref) <= RMS_TOL
    with pytest.raises(ValueError):
        G.EnvelopeFilter(G.Sine(440, samplerate=SR), 0, 0, 0, 0.5, 0, cycle=True)


def test_late_window_phase_precision(gpu):
    """Samples 600 s into the stream: float32 phase would be off by radians here, and the float64
    accumulation has drifted ~1e-5 rad from the ideal n*inc -- the tables follow the drift."""
    from synthesizer_amd import oscillators as G
    start, n = 600 * SR, 4096
    f = 3519.77
    inc = 2.0 * math.pi * f / SR
    t = accumulated(0.25 * 2.0 * math.pi, inc, start, n)
    want = np.sin(t) * 0.9 + 0.0
    got = G.Sine(f, 0.9, phase=0.25, samplerate=SR).render(n, start=start)
    assert rms(got, want) <= RMS_TOL
    assert np.max(np.abs(got - want)) < 1.5e-7
    ideal = np.sin(0.25 * 2 * math.pi + np.arange(start, start + n, dtype=np.float64) * inc) * 0.9
    assert np.max(np.abs(ideal - want)) > np.max(np.abs(got - want))      # the drift is real and we track it
    # square wave: every sample on the reference's side of its edge
    t = accumulated(0.0, 1000.0 / SR, start, n)
    want = np.where(np.trunc(t * 2) % 2 == 1, -1.0, 1.0)
    got = G.Square(1000.0, samplerate=SR).render(n, start=start)
    assert np.array_equal(got, want.astype(np.float32))


def test_blocks_protocol(gpu):
    """blocks() yields float64 (upstream: lists of Python floats), not float32-rounded values."""
    from synthesizer_amd import oscillators as G, params
    g = G.Sawtooth(100, samplerate=SR)
    o = O.Sawtooth(100, samplerate=SR)
    gb, ob = g.blocks(), o.blocks()
    for _ in range(3):
        b, w = next(gb), next(ob)
        assert isinstance(b, list) and len(b) == params.norm_osc_blocksize == len(w)
        assert isinstance(b[0], float)
        assert np.max(np.abs(np.array(b) - np.array(w))) <= 1e-12
    assert len(next(iter(g))) == params.norm_osc_blocksize
    # transcendental kinds: float64 all the way (a float32-rounded block would sit at ~3e-8)
    for name, args in (("Sine", (440.0, 0.8)), ("Harmonics", (220.0, [(1, 1.0), (2, 0.5), (5, 0.2)], 0.4)), ("Triangle", (330.0,))):
        gg, oo = _pair(name, *args, samplerate=SR)
        gb, ob = gg.blocks(), oo.blocks()
        for _ in range(130):                         # crosses the 128-block superblock of one kernel launch
            b, w = next(gb), next(ob)
        assert np.max(np.abs(np.array(b) - np.array(w))) <= 1e-12, name
    # Square / Pulse / Linear: equal
    gg, oo = _pair("Square", 1000.0, 0.7, samplerate=SR)
    assert next(gg.blocks()) == next(oo.blocks())
    got = G.Sine(440.0, samplerate=SR).render_f64(1000, start=5)
    assert got.dtype == np.float64 and np.max(np.abs(got - np.array(O.Sine(440.0, samplerate=SR).take(1005))[5:])) <= 1e-12


def test_quantise_bit_exact_and_overflow(gpu):
    """Sample.from_osc_block: int(scale*v) in float64, truncation toward zero, OverflowError."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.sample import Sample
    block = G.Sine(997.0, 0.9999, samplerate=SR).render(50000)
    s = Sample.from_osc_block(block, SR)
    want = O.quantise([float(v) for v in block])
    assert s.samplewidth == 2 and s.nchannels == 1 and len(s) == 50000
    assert list(s.get_frame_array()) == want
    # values that sit a hair below an integer after scaling (float32 product would round up)
    tricky = np.array([1234.0 / 32767.0, -1234.0 / 32767.0, 0.99999994, -1.0, 1.0, 0.0, -0.0, 3.0517578e-05], dtype=np.float32)
    assert list(Sample.from_osc_block(tricky, SR).get_frame_array()) == O.quantise([float(v) for v in tricky])
    # float64 input (e.g. blocks from a float64 generator) takes the float64 kernel
    blk64 = np.array(O.Sine(440, 0.7, samplerate=SR).take(2000))
    assert list(Sample.from_osc_block(list(blk64), SR).get_frame_array()) == O.quantise(blk64)
    # other widths and explicit scale
    assert list(Sample.from_osc_block(block[:1000], SR, samplewidth=4).get_frame_array()) == O.quantise([float(v) for v in block[:1000]], 4)
    assert list(Sample.from_osc_block(block[:1000], SR, amplitude_scale=1000.0).get_frame_array()) == O.quantise([float(v) for v in block[:1000]], 2, 1000.0)
    with pytest.raises(OverflowError):
        Sample.from_osc_block(np.array([0.5, 1.01], dtype=np.float32), SR)
    with pytest.raises(OverflowError):
        O.quantise([0.5, 1.01])
    # the overflow flag does not stick
    assert list(Sample.from_osc_block(np.array([0.5], dtype=np.float32), SR).get_frame_array()) == [16383]


def test_quantise_f64_offsets_tails_and_overflow_position(gpu):
    """sh_quantize_f64 (the WaveSynth.to_sample route): the 8-per-thread kernel, its scalar tail, ranges that start off a
    16-byte boundary, and an out-of-range value anywhere in the range -> OverflowError."""
    from synthesizer_amd import _native as N
    L = N.lib()
    rng = np.random.default_rng(226)
    n = 70001
    vals = rng.uniform(-1.0, 1.0, n)
    vals[:8] = [1.0, -1.0, 0.0, -0.0, 1234.0 / 32767.0, -1234.0 / 32767.0, 0.9999999999, -0.9999999999]
    src = N.DeviceBuffer.from_array(vals)
    dst = N.DeviceBuffer(2 * (n + 16))
    for in_off, out_off, cnt in ((0, 0, n), (0, 0, 4096), (1, 0, 5000), (0, 3, 5000), (2, 8, 8191), (5, 1, 7), (0, 0, 8), (16, 16, 64)):
        N.check(L.sh_quantize_f64(src.handle, in_off, cnt, 32767.0, 2, dst.handle, out_off))
        got = dst.download(np.int16, cnt, 2 * out_off)
        assert list(got) == O.quantise(vals[in_off:in_off + cnt]), (in_off, out_off, cnt)
    for pos in (0, 7, 8, 4095, 65535, n - 1):
        bad = vals.copy()
        bad[pos] = 1.0001 if pos % 2 else float("nan")
        b = N.DeviceBuffer.from_array(bad)
        assert L.sh_quantize_f64(b.handle, 0, n, 32767.0, 2, dst.handle, 0) == N.SH_ERR_OVERFLOW, pos
    N.check(L.sh_quantize_f64(src.handle, 0, 16, 32767.0, 2, dst.handle, 0))       # the flag does not stick
    # the float32 entry points through the same cases (vectors of four samples, 8-byte stores)
    v32 = vals.astype(np.float32)
    src32 = N.DeviceBuffer.from_array(v32)
    for in_off, out_off, cnt in ((0, 0, n), (0, 0, 2048), (1, 0, 5000), (0, 3, 5000), (4, 4, 8191), (4, 2, 4099), (5, 1, 7), (0, 0, 3)):
        N.check(L.sh_quantize_f32(src32.handle, in_off, cnt, 32767.0, 2, dst.handle, out_off))
        assert list(dst.download(np.int16, cnt, 2 * out_off)) == O.quantise([float(x) for x in v32[in_off:in_off + cnt]]), (in_off, out_off, cnt)
    loud = (v32 * np.float32(1.7)).astype(np.float32)
    loud[[3, 600, n - 1]] = [np.float32("nan"), np.float32(5.0), np.float32(-5.0)]
    b = N.DeviceBuffer.from_array(loud)
    for cnt in (n, 1027, 2):
        N.check(L.sh_quantize_clip_f32(b.handle, cnt, 32767.0, dst.handle))
        p = 32767.0 * loud[:cnt].astype(np.float64)
        want = np.where(np.isnan(p), 0.0, np.clip(np.trunc(p), -32768.0, 32767.0)).astype(np.int16)
        assert np.array_equal(dst.download(np.int16, cnt), want), cnt
    assert L.sh_quantize_f32(b.handle, 0, n, 32767.0, 2, dst.handle, 0) == N.SH_ERR_OVERFLOW


def test_triangle_and_band_limited_harmonics(gpu):
    """Triangle, SquareH, SawtoothH."""
    from synthesizer_amd import oscillators as G
    for args, kw in (((440.0,), dict(amplitude=0.8, phase=0.3, bias=0.1)), ((1000.0,), {}), ((-300.0,), dict(phase=-0.4))):
        g, o = _pair("Triangle", *args, samplerate=SR, **kw)
        _check(g, o, 20000)
    g = G.Triangle(330.0, fm_lfo=G.Sine(4.0, 0.03, samplerate=SR), samplerate=SR)
    o = O.Triangle(330.0, fm_lfo=O.Sine(4.0, 0.03, samplerate=SR), samplerate=SR)
    assert rms(g.render(20000), np.array(o.take(20000))) <= RMS_TOL
    for cls in ("SquareH", "SawtoothH"):
        for nh in (4, 16):
            g, o = _pair(cls, 220.0, nh, samplerate=SR, bias=0.05, phase=0.1)
            _check(g, o, 20000, max_ulp=2)
        g = getattr(G, cls)(220.0, fm_lfo=G.Sine(3.0, 0.02, samplerate=SR), samplerate=SR)
        o = getattr(O, cls)(220.0, fm_lfo=O.Sine(3.0, 0.02, samplerate=SR), samplerate=SR)
        assert rms(g.render(20000), np.array(o.take(20000))) <= RMS_TOL
    # in a bank
    from synthesizer_amd.mixer import VoiceBank
    gv = [G.Triangle(200.0, 0.3, samplerate=SR), G.SawtoothH(150.0, 8, 0.3, samplerate=SR), G.SquareH(100.0, 6, 0.3, samplerate=SR)]
    ov = [O.Triangle(200.0, 0.3, samplerate=SR), O.SawtoothH(150.0, 8, 0.3, samplerate=SR), O.SquareH(100.0, 6, 0.3, samplerate=SR)]
    gains = [(0.5, 0.25), (1.0, 0.0), (0.125, 0.75)]
    got = VoiceBank(gv, gains=gains).render(5000)
    want = np.array(O.mix_bus([v.take(5000) for v in ov], gains))
    assert rms(got, want) <= RMS_TOL


def test_filters(gpu):
    """MixingFilter, AmpModulationFilter, ClipFilter, AbsFilter, NullFilter, DelayFilter."""
    from synthesizer_amd import oscillators as G

    def build(M):
        a = M.Sine(440.0, 0.6, samplerate=SR)
        b = M.Sawtooth(111.0, 0.5, phase=0.2, samplerate=SR)
        c = M.EnvelopeFilter(M.Square(50.0, 0.4, samplerate=SR), 0.01, 0.02, 0.05, 0.5, 0.02)
        return {
            "mix": M.MixingFilter(a, b, c),
            "am": M.AmpModulationFilter(a, M.Sine(3.0, 0.5, bias=0.5, samplerate=SR)),
            "clip": M.ClipFilter(M.MixingFilter(a, b), -0.5, 0.7),
            "abs": M.AbsFilter(b),
            "null": M.NullFilter(a),
            "delay": M.DelayFilter(a, 0.01),
            "advance": M.DelayFilter(b, -0.02),
            "nested": M.AbsFilter(M.ClipFilter(M.AmpModulationFilter(M.MixingFilter(a, b), c), -0.2, 0.2)),
        }

    g, o = build(G), build(O)
    n = 6000
    for name in g:
        got = g[name].render(n)
        want = np.array(o[name].take(n))
        assert got.shape == (n,), name
        assert rms(got, want) <= RMS_TOL, name
        assert np.max(np.abs(got - want)) < 3e-7, name
    # random access and the blocks() protocol go through the same path
    assert np.array_equal(g["mix"].render(1000, start=2500), g["mix"].render(n, start=0)[2500:3500])
    assert np.array_equal(g["delay"].render(300, start=400), g["delay"].render(n, start=0)[400:700])
    assert rms(next(g["clip"].blocks()), next(o["clip"].blocks())) <= RMS_TOL
    # a filter graph as the voice of a bank: rendered into a row, mixed with the bus gains
    from synthesizer_amd.mixer import VoiceBank
    bus = VoiceBank([g["mix"]], gains=[(0.5, 0.25)]).render(2000)
    mixed = np.array(o["mix"].take(2000))
    assert rms(bus[:, 0], 0.5 * mixed) <= RMS_TOL and rms(bus[:, 1], 0.25 * mixed) <= RMS_TOL


def test_wavesynth_facade(gpu):
    """WaveSynth methods = oscillator (float64 block) + Sample.from_osc_block: bit-exact int16."""
    from synthesizer_amd.synth import WaveSynth
    ws = WaveSynth(samplerate=22050, samplewidth=2)
    dur = 0.25
    n = int(22050 * dur)
    cases = {
        "sine": (ws.sine(440, dur), O.Sine(440, 0.9999, samplerate=22050)),
        "square": (ws.square(440, dur, amplitude=0.5), O.Square(440, 0.5, samplerate=22050)),
        "triangle": (ws.triangle(220, dur), O.Triangle(220, 0.9999, samplerate=22050)),
        "sawtooth": (ws.sawtooth(330, dur, phase=0.2), O.Sawtooth(330, 0.75, phase=0.2, samplerate=22050)),
        "pulse": (ws.pulse(110, dur, pulsewidth=0.3), O.Pulse(110, 0.75, pulsewidth=0.3, samplerate=22050)),
        "square_h": (ws.square_h(220, dur, 5, amplitude=0.5), O.SquareH(220, 5, 0.5, samplerate=22050)),
        "sawtooth_h": (ws.sawtooth_h(220, dur, 6, amplitude=0.3), O.SawtoothH(220, 6, 0.3, samplerate=22050)),
        "harmonics": (ws.harmonics(220, dur, [(1, 1.0), (2, 0.5)], amplitude=0.4), O.Harmonics(220, [(1, 1.0), (2, 0.5)], 0.4, samplerate=22050)),
        "fm": (ws.sine(440, dur, fm_lfo=ws.sine_gen(5, 0.05)), O.Sine(440, 0.9999, fm_lfo=O.Sine(5, 0.05, samplerate=22050), samplerate=22050)),
        "white_noise": (ws.white_noise(2205, dur, amplitude=0.5, seed=11), O.WhiteNoise(2205, 0.5, samplerate=22050, seed=11)),
        "linear": (ws.linear(dur, -0.5, 1e-4, -1.0, 0.25), O.Linear(-0.5, 1e-4, -1.0, 0.25, samplerate=22050)),
    }
    for name, (sample, osc) in cases.items():
        assert sample.samplerate == 22050 and sample.nchannels == 1 and sample.samplewidth == 2 and len(sample) == n, name
        got = np.array(sample.get_frame_array())
        want = np.array(O.quantise(osc.take(n)))
        # float64 block -> int(scale * v): the reference's own route, no float32 on the way
        assert np.array_equal(got, want), (name, int(np.sum(got != want)))
    # 32-bit samples: every one of the 31 magnitude bits comes from the float64 block
    ws4 = WaveSynth(samplerate=22050, samplewidth=4)
    for name, sample, osc in (("sine32", ws4.sine(440, dur), O.Sine(440, 0.9999, samplerate=22050)),
                              ("harm32", ws4.harmonics(220, dur, [(1, 1.0), (3, 0.3)], amplitude=0.4),
                               O.Harmonics(220, [(1, 1.0), (3, 0.3)], 0.4, samplerate=22050)),
                              ("saw32", ws4.sawtooth(330, dur), O.Sawtooth(330, 0.75, samplerate=22050))):
        got = np.array(sample.get_frame_array(), dtype=np.int64)
        want = np.array(O.quantise(osc.take(n), 4), dtype=np.int64)
        assert sample.samplewidth == 4
        # scale 2^31-1 magnifies the ~2e-16 difference between the table-driven sin and libm's to ~1e-6 of an LSB:
        # a sample within that of an integer may land on the other side
        assert np.max(np.abs(got - want)) <= 1 and np.mean(got != want) < 1e-4, (name, int(np.sum(got != want)))
    assert np.array_equal(np.array(ws4.square(440, dur, amplitude=0.5).get_frame_array()),
                          np.array(O.quantise(O.Square(440, 0.5, samplerate=22050).take(n), 4)))


def test_randomised_voices_vs_c_oracle(gpu):
    """Seeded sweep over kinds x FM x envelope x parameters, 30 000 samples each, against the C oracle."""
    from oracle import c_oracle as CO
    from synthesizer_amd import oscillators as G
    import os
    rng = np.random.default_rng(int(os.environ.get("SYNTHHIP_FUZZ_SEED", "777")))
    kinds = ["Sine", "Sawtooth", "Square", "Pulse", "Harmonics"]
    n = 30000
    worst = 0.0
    for case in range(60):
        kind = kinds[case % 5]
        sr = int(rng.choice([22050, 44100, 48000, 96000]))
        f = float(np.exp(rng.uniform(np.log(20), np.log(0.45 * sr))))
        kw = dict(amplitude=float(rng.uniform(0.05, 1.0)), phase=float(rng.uniform(-1, 1)), bias=float(rng.uniform(-0.3, 0.3)), samplerate=sr)
        args = [f]
        if kind == "Pulse":
            kw["pulsewidth"] = float(rng.uniform(0.02, 0.98))
        if kind == "Harmonics":
            nh = int(rng.integers(1, 25))
            args.append([(int(k), float(rng.uniform(-1, 1))) for k in rng.choice(np.arange(1, 40), size=nh, replace=False)])
        fm = case % 3 == 1
        lfo_args = (float(rng.uniform(0.1, 30)), float(rng.uniform(0, 0.2)), float(rng.uniform(0, 1)), float(rng.uniform(-0.05, 0.05)))

        def make(M):
            lfo = M.Sine(lfo_args[0], lfo_args[1], phase=lfo_args[2], bias=lfo_args[3], samplerate=sr) if fm else None
            osc = getattr(M, kind)(*args, fm_lfo=lfo, **kw)
            if case % 4 == 2:
                a, d, s, sl, r = (float(x) for x in rng2.uniform(0, 0.1, 5))
                osc = M.EnvelopeFilter(osc, a, d, s, min(sl * 10, 1.0), r)
            return osc

        rng2 = np.random.default_rng(case)
        g = make(G)
        rng2 = np.random.default_rng(case)
        o = make(O)
        got = g.render(n)
        want = CO.render(o, n)
        if kind in ("Square", "Pulse"):
            flips = int(np.sum(got != want.astype(np.float32)))
            assert flips <= (2 if fm else 0), (case, kind, flips)
        elif kind == "Sawtooth" and fm:
            ok = np.abs(np.abs(want - kw["bias"]) - kw["amplitude"]) > 1e-3 * kw["amplitude"]      # away from the wrap
            assert rms(got[ok], want[ok]) <= RMS_TOL, case
        else:
            e = rms(got, want)
            worst = max(worst, e)
            assert e <= RMS_TOL, (case, kind, e)
    assert worst < 1e-7


@pytest.mark.parametrize("args", [(0.0, 0.3, -1.0, 1.0), (0.5, -0.4, -1.0, 1.0), (0.0, 1e-5, -1.0, 1.0), (0.25, 0.0, -1.0, 1.0),
                                  (0.9, -3.3e-6, 0.0, 1.0), (2.0, 0.1, -1.0, 1.0), (-0.5, 1.0 / 48000, -1.0, 0.25),
                                  (1e-3, 1e-9, -1.0, 1.0), (0.0, 0.1, -1.0, 1e30)])
def test_linear_is_bit_exact(gpu, args):
    """Linear.  The level is a float64 running sum; the phase table reproduces it exactly, so the
    float32 output equals the oracle's sequential loop bit for bit -- also deep into the stream and past the stop."""
    g, o = _pair("Linear", *args, samplerate=SR)
    n = 250000
    want = np.array(o.take(n), dtype=np.float64)
    got = g.render(n, start=0)
    assert np.array_equal(got, want.astype(np.float32)), int(np.sum(got != want.astype(np.float32)))
    assert np.array_equal(g.render(777, start=199000), want[199000:199777].astype(np.float32))
    # as a modulator (float64 path) and inside a bank with an envelope
    from synthesizer_amd import oscillators as G
    am_g = G.AmpModulationFilter(G.Sine(100.0, samplerate=SR), g)
    am_o = O.AmpModulationFilter(O.Sine(100.0, samplerate=SR), O.Linear(*args, samplerate=SR))
    ref = np.array(am_o.take(5000))
    assert rms(am_g.render(5000), ref) <= RMS_TOL * max(1.0, float(np.abs(ref).max()))      # float32 output: relative


def test_white_noise_counter_based(gpu):
    """WhiteNoise with a counter-based generator: bit-exact against the oracle's restatement,
    random access equals streaming, different seeds differ, the values are uniform over [-a, a) + bias."""
    from synthesizer_amd import oscillators as G
    for freq, amp, bias, sr, seed in ((48000.0, 1.0, 0.0, 48000, 0), (4410.0, 0.5, 0.1, 44100, 3), (1000.0, 0.25, -0.5, 48000, 2 ** 63 + 5),
                                      (7.0, 1.0, 0.0, 8000, 12345)):
        g = G.WhiteNoise(freq, amp, bias, samplerate=sr, seed=seed)
        o = O.WhiteNoise(freq, amp, bias, samplerate=sr, seed=seed)
        n = 30000
        want = np.array(o.take(n), dtype=np.float64)
        got = g.render(n, start=0)
        assert np.array_equal(got, want.astype(np.float32))
        assert np.array_equal(g.render(1234, start=20000), got[20000:21234])
        hold = int(sr / freq)
        assert np.all(got[:hold] == got[0])
        assert want.min() >= -amp + bias and want.max() < amp + bias
    a = G.WhiteNoise(48000.0, samplerate=48000, seed=1).render(200000)
    b = G.WhiteNoise(48000.0, samplerate=48000, seed=2).render(200000)
    assert not np.array_equal(a, b)
    assert abs(float(a.mean())) < 0.01 and abs(float(a.astype(np.float64).var()) - 1.0 / 3.0) < 0.01
    hist = np.histogram(a, bins=16, range=(-1, 1))[0]
    assert hist.min() > 0.9 * len(a) / 16
    # far into the stream (sample index above 2^32): the 64-bit counter path
    g = G.WhiteNoise(12000.0, samplerate=48000, seed=9)
    far = (1 << 33) + 12345
    want = []
    for k in range(64):
        h = (far + k) // 4
        u = (O.splitmix64(9 + h * 0x9E3779B97F4A7C15) >> 11) * 2.0 ** -53
        want.append((-1.0 + 2.0 * u) + 0.0)
    assert np.array_equal(g.render(64, start=far), np.array(want).astype(np.float32))
    with pytest.raises(ValueError):
        G.WhiteNoise(50000.0, samplerate=48000).render(10)
    # noise voices in a bank, with an envelope
    from synthesizer_amd.mixer import VoiceBank
    env_g = G.EnvelopeFilter(G.WhiteNoise(6000.0, 0.5, samplerate=SR, seed=4), 0.01, 0.01, 0.02, 0.5, 0.02)
    env_o = O.EnvelopeFilter(O.WhiteNoise(6000.0, 0.5, samplerate=SR, seed=4), 0.01, 0.01, 0.02, 0.5, 0.02)
    bus = VoiceBank([env_g], gains=[(1.0, 0.5)]).render(4000)
    ref = np.array(env_o.take(4000))
    assert rms(bus[:, 0], ref) <= RMS_TOL and rms(bus[:, 1], ref * 0.5) <= RMS_TOL


def test_echo_filter(gpu):
    """EchoFilter."""
    from synthesizer_amd import oscillators as G

    def build(M, after, amount, delay, decay):
        src = M.EnvelopeFilter(M.Sine(440.0, 0.6, samplerate=SR), 0.005, 0.01, 0.01, 0.5, 0.02)
        return M.EchoFilter(src, after, amount, delay, decay)

    for after, amount, delay, decay in ((0.01, 3, 0.02, 0.5), (0.0, 5, 0.003, 0.9), (0.05, 1, 0.0, 1.0), (0.02, 0, 0.1, 0.5),
                                        (0.0301, 4, 0.0107, 0.33)):
        g, o = build(G, after, amount, delay, decay), build(O, after, amount, delay, decay)
        n = 9000
        got = g.render(n)
        want = np.array(o.take(n))
        assert rms(got, want) <= RMS_TOL, (after, amount, delay, decay)
        assert np.max(np.abs(got - want)) < 3e-7
        # random access: the same samples up to float64 rounding of the source (its evaluation depends on the launch start)
        assert np.max(np.abs(g.render(700, start=3000) - got[3000:3700])) < 1e-7
        assert g.echo_duration == o.echo_duration
    with pytest.raises(ValueError):
        G.EchoFilter(G.Sine(1.0), 0.1, 2, 0.1, 1.5)


def test_linear_and_noise_at_full_size_vs_c_oracle(gpu):
    """100 s of WhiteNoise and Linear at 48 kHz (4.8 M samples) against the C restatement of the oracle
    (oracle/oracle.c: or_white_noise / or_linear, themselves checked against the Python oracle in tests/test_oracle_c.py)."""
    from oracle import c_oracle as CO
    from synthesizer_amd import oscillators as G
    n = 4_800_000
    for args in ((6000.0, 0.7, 0.05, 7), (48000.0, 1.0, 0.0, 2 ** 64 - 3), (3.0, 0.5, -0.2, 1)):
        freq, amp, bias, seed = args
        want = CO.render(O.WhiteNoise(freq, amp, bias, samplerate=SR, seed=seed), n).astype(np.float32)
        got = G.WhiteNoise(freq, amp, bias, samplerate=SR, seed=seed).render(n)
        assert np.array_equal(got, want), args
    for args in ((-1.0 + 1e-9, 4.1e-7, -1.0, 1.0), (0.5, -3.0e-7, -0.75, 1.0), (0.0, 1e-12, -1.0, 1.0)):
        want = CO.render(O.Linear(*args, samplerate=SR), n).astype(np.float32)
        got = G.Linear(*args, samplerate=SR).render(n)
        assert np.array_equal(got, want), args
