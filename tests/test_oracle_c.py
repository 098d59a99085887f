"""oracle/oracle.c (the C restatement used for large-size checks) against oracle/synth_oracle.py and the
live audioop module: the two oracles must agree bit for bit before either is trusted."""
import audioop

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import pcm_oracle as P
from oracle import synth_oracle as O

SR = 48000


@pytest.mark.parametrize("make", [
    lambda: O.Sine(440.0, 0.7, phase=0.3, bias=0.1, samplerate=SR),
    lambda: O.Sawtooth(1000.0, 0.7, phase=-0.3, samplerate=SR),
    lambda: O.Square(1000.0, samplerate=SR),
    lambda: O.Square(-250.0, phase=0.1, samplerate=SR),
    lambda: O.Pulse(441.0, pulsewidth=0.25, phase=-0.2, samplerate=SR),
    lambda: O.Harmonics(220.0, [(k, 1.0 / k) for k in range(1, 17)], 0.5, phase=0.2, samplerate=SR),
    lambda: O.Sine(440.0, fm_lfo=O.Sine(5.0, 0.03, phase=0.4, samplerate=SR), samplerate=SR),
    lambda: O.Square(300.0, fm_lfo=O.Sine(2.0, 0.05, samplerate=SR), samplerate=SR),
    lambda: O.Harmonics(110.0, [(1, 1.0), (2, 0.5)], fm_lfo=O.Sine(3.0, 0.02, samplerate=SR), samplerate=SR),
    lambda: O.EnvelopeFilter(O.Sine(440.0, samplerate=SR), 0.01, 0.05, 0.1, 0.6, 0.05),
    lambda: O.EnvelopeFilter(O.Sawtooth(300.0, samplerate=SR), 0.0, 0.02, 0.0, 0.3, 0.02),
    lambda: O.EnvelopeFilter(O.Pulse(100.0, samplerate=SR), 0.02, 0.0, 0.01, 1.0, 0.0),
    lambda: O.Triangle(441.0, 0.6, phase=-0.4, bias=0.05, samplerate=SR),
    lambda: O.Triangle(-250.0, 0.5, phase=0.3, samplerate=SR),
    lambda: O.Triangle(300.0, fm_lfo=O.Sine(2.0, 0.05, samplerate=SR), samplerate=SR),
    lambda: O.Linear(-0.5, 0.00013, samplerate=SR),
    lambda: O.Linear(0.9, 0.001, min_value=-2.0, max_value=1.0, samplerate=SR),
    lambda: O.Linear(0.2, -0.0007, samplerate=SR),
    lambda: O.WhiteNoise(4000.0, 0.8, bias=0.1, samplerate=SR, seed=12345),
    lambda: O.WhiteNoise(48000.0, 1.0, samplerate=SR, seed=(1 << 63) + 7),
    lambda: O.EnvelopeFilter(O.WhiteNoise(1000.0, samplerate=SR, seed=3), 0.01, 0.02, 0.05, 0.5, 0.05),
])
def test_c_oscillators_equal_python_oracle(make):
    n = 12000
    assert np.array_equal(CO.render(make(), n), np.array(make().take(n)))


def test_c_pcm_equals_audioop():
    rng = np.random.default_rng(0)
    for width, dt in ((1, np.int8), (2, np.int16), (4, np.int32)):
        info = np.iinfo(dt)
        a = rng.integers(info.min, info.max + 1, 5000, dtype=np.int64).astype(dt).tobytes()
        b = rng.integers(info.min, info.max + 1, 5000, dtype=np.int64).astype(dt).tobytes()
        assert CO.add(a, b, width) == audioop.add(a, b, width)
        for nch in (1, 2):
            for (i, o) in ((96000, 44100), (44100, 48000), (3, 7)):
                x = a[:(len(a) // (width * nch)) * width * nch]
                assert CO.ratecv(x, width, nch, i, o) == audioop.ratecv(x, width, nch, i, o, None)[0]
    x = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    assert np.array_equal(CO.ratecv_f32(x, 96000, 44100), P.ratecv_f32(x, 96000, 44100))
    v = rng.uniform(-1, 1, 4000)
    assert CO.quantise(v).tolist() == O.quantise(v.tolist())
    with pytest.raises(OverflowError):
        CO.quantise(np.array([1.5]))
    voices = rng.uniform(-1, 1, (7, 300))
    gains = [(float(l), float(r)) for l, r in rng.uniform(0, 1, (7, 2))]
    assert np.array_equal(CO.mix_bus(voices, gains), np.array(O.mix_bus(voices.tolist(), gains)))


@pytest.mark.parametrize("make", [
    lambda: O.Harmonics(440.0, [(k, 1.0 / k) for k in range(1, 17)], 0.5, phase=0.2, samplerate=SR),
    lambda: O.EnvelopeFilter(O.Harmonics(313.7, [(1, 1.0), (3, 0.3), (7, -0.2)], 0.8, phase=0.9, samplerate=SR), 0.01, 0.05, 1.0e6, 0.6, 0.2),
    lambda: O.EnvelopeFilter(O.Sawtooth(300.0, samplerate=SR), 0.01, 0.02, 1000.0, 0.3, 0.02),
])
def test_render_window_enters_the_loop_late(make):
    """c_oracle.render_window (the generator's loop entered at `start`: the phase brought there by `start` additions, the envelope's
    sustain applied) equals the tail of a render from sample 0."""
    start, n = 2 * SR + 12345, 5000
    assert np.array_equal(CO.render_window(make(), start, n), CO.render(make(), start + n)[start:])
