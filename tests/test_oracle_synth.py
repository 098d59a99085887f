"""Sanity of oracle/synth_oracle.py (the restatement of the oscillator path).  Parity of these rows is
UNPINNED: the upstream tree is not mounted, so the golden files only guard the oracle against accidental
edits; the closed-form checks below tie it to the formulas written in SURVEY.md section 8(a)."""
import itertools
import math

import numpy as np
import pytest

from oracle import synth_oracle as O


def test_golden_vectors_reproduce():
    g = np.load("tests/golden/osc_misc.npz")
    sr = 48000
    assert np.array_equal(g["saw"], O.Sawtooth(1000, 0.8, phase=0.1, bias=0.05, samplerate=sr).take(2048))
    assert np.array_equal(g["square"], O.Square(1000, samplerate=sr).take(2048))
    assert np.array_equal(g["pulse"], O.Pulse(441, pulsewidth=0.25, samplerate=sr).take(2048))
    assert np.array_equal(g["harm"], O.Harmonics(220, [(k, 1.0 / k) for k in range(1, 17)], 0.5, samplerate=sr).take(2048))
    assert np.array_equal(g["fm_sine"], O.Sine(440, fm_lfo=O.Sine(5, 0.03, samplerate=sr), samplerate=sr).take(2048))
    assert np.array_equal(g["adsr"], O.EnvelopeFilter(O.Sine(440, samplerate=sr), 0.01, 0.01, 0.01, 0.6, 0.01).take(2048))
    assert np.array_equal(g["adsr_cycle"], O.EnvelopeFilter(O.Sine(440, samplerate=sr), 0.004, 0.003, 0.005, 0.6, 0.006, cycle=True).take(2048))
    assert np.array_equal(g["quant"], O.quantise(g["harm"] * 0.5))
    s = np.array(O.Sine(440, samplerate=44100).take(44100))
    assert np.array_equal(np.load("tests/golden/osc_sine440_44k1.npy"), s[np.r_[0:4096, 40004:44100]])


def test_closed_forms():
    sr = 48000
    n = np.arange(4000)
    s = np.array(O.Sine(440, 0.5, phase=0.25, bias=0.1, samplerate=sr).take(4000))
    assert np.max(np.abs(s - (0.5 * np.sin(2 * np.pi * (440 * n / sr + 0.25)) + 0.1))) < 1e-10
    x = 100 * n / sr + 0.2
    w = np.array(O.Sawtooth(100, samplerate=sr, phase=0.2).take(4000))
    fr = (x + 0.5) - np.floor(x + 0.5)
    safe = (fr > 1e-9) & (fr < 1 - 1e-9)
    assert np.max(np.abs(w - 2 * (x - np.floor(x + 0.5)))[safe]) < 1e-9
    q = np.array(O.Square(100, samplerate=sr, phase=0.2).take(4000))
    frac = x - np.floor(x)
    safe = (np.abs(frac - 0.5) > 1e-9) & (frac > 1e-9) & (frac < 1 - 1e-9)
    assert np.array_equal(q[safe], np.where(frac < 0.5, 1.0, -1.0)[safe])
    p = np.array(O.Pulse(100, pulsewidth=0.3, samplerate=sr, phase=0.2).take(4000))
    safe = (np.abs(frac - 0.3) > 1e-9) & (frac > 1e-9) & (frac < 1 - 1e-9)
    assert np.array_equal(p[safe], np.where(frac < 0.3, 1.0, -1.0)[safe])
    h = np.array(O.Harmonics(50, [(1, 1.0), (2, 0.5)], samplerate=sr).take(4000))
    th = 2 * np.pi * 50 * n / sr
    assert np.max(np.abs(h - (np.sin(th) + 0.5 * np.sin(2 * th)))) < 1e-9


def test_fm_with_silent_lfo_equals_plain_within_rounding():
    sr = 48000
    a = np.array(O.Sine(440, fm_lfo=O.Sine(3, 0.0, samplerate=sr), samplerate=sr).take(5000))
    b = np.array(O.Sine(440, samplerate=sr).take(5000))
    assert np.max(np.abs(a - b)) < 1e-9
    # FM is the running sum of the instantaneous frequency
    lfo = np.array(O.Sine(3, 0.1, samplerate=sr).take(5000))
    c = np.array(O.Sine(440, fm_lfo=O.Sine(3, 0.1, samplerate=sr), samplerate=sr).take(5000))
    theta = 2 * np.pi / sr * np.concatenate([[0.0], np.cumsum(440 * (1 + lfo))[:-1]])
    assert np.max(np.abs(c - np.sin(theta))) < 1e-8


def test_envelope_shape_and_length():
    sr = 8000
    e = O.EnvelopeFilter(O.Sine(0.0, phase=0.25, samplerate=sr), 0.1, 0.1, 0.2, 0.5, 0.1, stop_at_end=True)
    x = np.array(list(itertools.chain.from_iterable(e.blocks())))
    assert abs(len(x) - 4000) <= 4
    assert abs(x[400] - 0.5) < 2e-3 and abs(x[799] - 1.0) < 3e-3          # attack ramp (source is the constant 1)
    assert abs(x[1700] - 0.5) < 1e-12 and abs(x[3600] - 0.25) < 2e-3      # sustain, mid release
    forever = O.EnvelopeFilter(O.Sine(0.0, phase=0.25, samplerate=sr), 0.1, 0.1, 0.2, 0.5, 0.1)
    tail = forever.take(5000)[4100:]
    assert not any(tail)


def test_envelope_cycle_repeats_the_gain_and_keeps_the_source_running():
    """cycle=True: the phases start over after the release; the host side's period (envelope_spec(...).length) is the
    generator's, the source is pulled once per sample throughout."""
    from synthesizer_amd.oscillators import envelope_spec
    sr = 8000
    adsr = (0.01, 0.02, 0.03, 0.5, 0.04)
    period = envelope_spec(*adsr, sr, True).length
    one = np.array(list(itertools.chain.from_iterable(
        O.EnvelopeFilter(O.Linear(1.0, samplerate=sr), *adsr, stop_at_end=True).blocks())))
    assert one.size == period
    gain = np.array(O.EnvelopeFilter(O.Linear(1.0, samplerate=sr), *adsr, cycle=True).take(3 * period + 17))
    assert np.array_equal(gain, np.tile(one, 4)[:gain.size])
    src = np.array(O.Sine(440, samplerate=sr).take(gain.size))
    x = np.array(O.EnvelopeFilter(O.Sine(440, samplerate=sr), *adsr, stop_at_end=True, cycle=True).take(gain.size))
    assert np.array_equal(x, src * gain)


def test_quantise_rule():
    assert O.quantise([0.5, -0.5, 0.99999, -1.0, 1.0, 3.05e-5, -3.06e-5]) == [16383, -16383, 32766, -32767, 32767, 0, -1]
    assert O.quantise([1.0], 1) == [127] and O.quantise([-1.0], 4) == [-(2 ** 31 - 1)]
    with pytest.raises(OverflowError):
        O.quantise([1.0001])


def test_splitmix64_known_answers_and_noise():
    """The counter-based generator behind WhiteNoise: SplitMix64's published first outputs for seed 0."""
    assert O.splitmix64(0x9E3779B97F4A7C15) == 0xE220A8397B1DCDAF
    assert O.splitmix64((2 * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)) == 0x6E789E6AA1B965F4
    x = O.WhiteNoise(4410.0, 0.5, 0.1, samplerate=44100, seed=3).take(2000)
    assert all(x[i] == x[i - i % 10] for i in range(2000))            # held for int(44100/4410) = 10 samples
    assert len(set(x)) == 200 and -0.4 <= min(x) and max(x) < 0.6
    assert x == O.WhiteNoise(4410.0, 0.5, 0.1, samplerate=44100, seed=3).take(2000)
    assert x != O.WhiteNoise(4410.0, 0.5, 0.1, samplerate=44100, seed=4).take(2000)


def test_linear_and_echo_known_answers():
    assert O.Linear(0.0, 0.25, -1, 1).take(7) == [0.0, 0.25, 0.5, 0.75, 1.0, 1.0, 1.0]
    assert O.Linear(0.5, -0.5, -1, 1).take(6) == [0.5, 0.0, -0.5, -1.0, -1.0, -1.0]
    assert O.Linear(3.0, 0.5, -1, 1).take(3) == [3.0, 3.0, 3.0]
    assert O.Linear(0.25, 0.0).take(3) == [0.25, 0.25, 0.25]
    # a constant source: plays alone for 3 samples, then echos at +2 and +4 samples at 0.5 and 0.25
    e = O.EchoFilter(O.Linear(1.0, 0.0, samplerate=10), 0.3, 2, 0.2, 0.5)
    assert e.take(10) == [1.0, 1.0, 1.0, 1.0, 1.0, 1.5, 1.5, 1.75, 1.75, 1.75]
    assert e.echo_duration == 0.3 + 2 * 0.2
    assert O.EchoFilter(O.Linear(0.0, 1.0, -1e9, 1e9, samplerate=10), 0.0, 1, 0.1, 1.0).take(5) == [0.0, 1.0, 3.0, 5.0, 7.0]
