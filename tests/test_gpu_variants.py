"""One GPU test per reading of the recalled arithmetic (oracle.synth_oracle.VARIANTS == synthesizer_amd.params.variants): flipping a
reading the day tools/pin_oracle.py --variants names it is a change of a default, not of a kernel -- the device reproduces whatever
increments, records, widths, boundaries and rounding the host code describes.  Every test checks (a) that its case can tell the two
readings apart (the oracle's output changes) and (b) that the product under the reading equals the oracle under the reading."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 48000


class both_readings:
    """Oracle and product under the same reading for the duration of a block; both tables restored afterwards."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        from oracle import synth_oracle as O
        from synthesizer_amd import params
        self.old_o = O.set_variants(**self.kw)
        self.old_p = params.set_variants(**self.kw)
        return self

    def __exit__(self, *exc):
        from oracle import synth_oracle as O
        from synthesizer_amd import params
        O.set_variants(**self.old_o)
        params.set_variants(**self.old_p)
        return False


def _take(osc, n):
    return np.array(osc.take(n), dtype=np.float64)


def test_increment_by_division(gpu):
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    n = 4096
    base_sine = _take(O.Sine(440.0, 0.8, samplerate=SR), n)
    with both_readings(increment="div"):
        assert not np.array_equal(_take(O.Sine(440.0, 0.8, samplerate=SR), n), base_sine)            # the readings part (phase 0: from the first samples)
        for f in (440.0, 220.0, 3520.0):
            want = CO.render(O.Sine(f, 0.8, samplerate=SR), (1 << 20) + n)
            got0 = G.Sine(f, 0.8, samplerate=SR).render_f64(n, start=0)
            got1 = G.Sine(f, 0.8, samplerate=SR).render_f64(n, start=1 << 20)
            assert np.max(np.abs(got0 - want[:n])) <= 4e-16 and np.max(np.abs(got1 - want[1 << 20:])) <= 4e-16, f
        # a turn-based kind whose edges the accumulated t decides: 441 Hz at 48 kHz (f / sr != 1 / (sr / f)), a million samples in
        # (the increments differ by an ulp; whether an edge moves within these samples is not the point: the product follows the oracle)
        want = CO.render(O.Square(441.0, 0.8, samplerate=SR), (1 << 20) + (1 << 16))
        got = G.Square(441.0, 0.8, samplerate=SR).render_f64(1 << 16, start=1 << 20)
        assert np.array_equal(got, want[1 << 20:])
        # FM: the LFO's own increment follows the reading too
        o = O.Sine(440.0, 0.7, fm_lfo=O.Sine(220.0, 0.1, samplerate=SR), samplerate=SR)
        g = G.Sine(440.0, 0.7, fm_lfo=G.Sine(220.0, 0.1, samplerate=SR), samplerate=SR)
        assert np.sqrt(np.mean((g.render_f64(n, start=48000) - CO.render(o, 48000 + n)[48000:]) ** 2)) <= 1e-10


def test_square_by_modulo(gpu):
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    n = 4096
    make = lambda m: m.Square(440.0, 0.8, -0.3, 0.05, samplerate=SR)          # a negative phase: int() truncates toward zero, % does not
    base = _take(make(O), n)
    with both_readings(square="mod1"):
        want = _take(make(O), n)
        assert not np.array_equal(want, base)
        assert np.array_equal(make(G).render_f64(n), want)
        assert next(make(G).blocks()) == want[:512].tolist()
        # ... and in a bank (float32 bus of one voice at unit gain == float32 of the samples)
        bus = VoiceBank([make(G)], gains=[(1.0, 1.0)]).render(n)
        assert np.array_equal(bus[:, 0], want.astype(np.float32))
    assert np.array_equal(make(G).render_f64(n), base)                        # back to the default reading


def test_pulse_inclusive_comparison(gpu):
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    n = 2048
    make = lambda m: m.Pulse(375.0, 0.8, 0.0, 0.25, 0.1, samplerate=SR)      # 375 / 48000 = 1 / 128: t % 1 lands ON the width every 128 samples
    base = _take(make(O), n)
    with both_readings(pulse="le"):
        want = _take(make(O), n)
        assert not np.array_equal(want, base)
        assert np.array_equal(make(G).render_f64(n), want)
        # a pwm_lfo whose widths land on t % 1 as well: a constant modulator (Linear without increment)
        pw = lambda m: m.Pulse(375.0, 0.8, 0.0, 0.25, 0.0, pwm_lfo=m.Linear(0.25, samplerate=SR), samplerate=SR)
        want_pwm = _take(pw(O), n)
        assert np.array_equal(pw(G).render_f64(n), want_pwm)
    assert np.array_equal(make(G).render_f64(n), base)


def test_quantise_by_rounding(gpu):
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.sample import Sample
    halves = [0.375, 0.625, -0.375, -0.625, 0.874, 0.126, -0.126, 0.0, 0.5, 1.5, 2.5, -2.5, 3.4999, 8191.75, -8192.0]
    block = np.array(halves + (0.9 * np.sin(0.01 * np.arange(5000))).tolist(), dtype=np.float64)
    base = O.quantise(block.tolist(), 2, 4.0)
    with both_readings(quantise="round"):
        want = O.quantise(block.tolist(), 2, 4.0)
        assert want != base and want[:4] == [2, 2, -2, -2]                    # half to even
        assert list(Sample.from_osc_block(block, SR, amplitude_scale=4.0).get_frame_array()) == want
        assert list(Sample.from_osc_block(block.astype(np.float32), SR, amplitude_scale=4.0).get_frame_array()) == \
            O.quantise(block.astype(np.float32).astype(np.float64).tolist(), 2, 4.0)
        with pytest.raises(OverflowError):
            Sample.from_osc_block(np.array([8191.9]), SR, amplitude_scale=4.0)     # rounds to 32768
        # the int16 materialisation of a bank: through float64 rows and the rounding quantiser
        gv = [G.Sine(440.0, 0.5, samplerate=SR), G.Harmonics(220.0, [(1, 1.0), (2, 0.5)], 0.4, samplerate=SR)]
        ov = [O.Sine(440.0, 0.5, samplerate=SR), O.Harmonics(220.0, [(1, 1.0), (2, 0.5)], 0.4, samplerate=SR)]
        rows, stride = VoiceBank(gv).generate_i16_device(3000)
        got = rows.download(np.int16, 2 * stride).reshape(2, stride)[:, :3000]
        for k in range(2):
            w = np.array(O.quantise(ov[k].take(3000)), dtype=np.int16)
            assert np.count_nonzero(got[k] != w) <= 1
    assert list(Sample.from_osc_block(block, SR, amplitude_scale=4.0).get_frame_array()) == base


def test_envelope_inclusive_boundaries(gpu):
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    sr = 32768                                                                 # 1 / sr is a float64: the accumulated time is exact, boundaries fall ON samples
    make = lambda m, **kw: m.EnvelopeFilter(m.Sawtooth(512.0, 0.9, samplerate=sr), 1.0 / 64, 1.0 / 128, 1.0 / 64, 0.5, 1.0 / 128, **kw)
    n = 2048
    base = _take(make(O), n)
    with both_readings(envelope="le"):
        want = _take(make(O), n)
        assert not np.array_equal(want, base)
        assert np.array_equal(make(G).render_f64(n), want)
        stop = make(O, stop_at_end=True).take(5000)
        g = make(G, stop_at_end=True)
        assert g.length == len(stop) and np.array_equal(g.render_f64(5000), np.array(stop))
        bus = VoiceBank([make(G)], gains=[(1.0, 1.0)]).render(n)
        assert np.array_equal(bus[:, 0], want.astype(np.float32))
    assert np.array_equal(make(G).render_f64(n), base)
