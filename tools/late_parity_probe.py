"""Every oscillator kind far into a note (300 s at 48 kHz: 1.4e7 samples of accumulated phase) against the C oracle's float64 values: the plain
kinds, Harmonics in its three forms, FM under a Sine LFO on a turn-based carrier, a Pulse with a pwm_lfo, an envelope with a long sustain.
usage (GPU box): python tools/late_parity_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import synth_oracle as O
from oracle import c_oracle as CO
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
N.ensure_init(0)
SR, blk = 48000, 16384
first = 300 * SR


def cases(m):
    h16 = [(k, 1.0 / k) for k in range(1, 17)]
    return [("Sine 1234.5 Hz", m.Sine(1234.5, 0.8, phase=0.1, samplerate=SR)),
            ("Sawtooth 1000 Hz (edges on samples)", m.Sawtooth(1000.0, 0.8, samplerate=SR)),
            ("Square 1000 Hz (edges on samples)", m.Square(1000.0, 0.8, samplerate=SR)),
            ("Pulse 777.7 Hz", m.Pulse(777.7, 0.8, pulsewidth=0.3, samplerate=SR)),
            ("Triangle 432.1 Hz", m.Triangle(432.1, 0.8, phase=0.4, samplerate=SR)),
            ("Harmonics x16 (polynomial)", m.Harmonics(440.0, h16, 0.5, samplerate=SR)),
            ("Harmonics 1 + 33 (Clenshaw)", m.Harmonics(200.0, [(1, 1.0), (33, 0.2)], 0.5, samplerate=SR)),
            ("Harmonics sparse (1, 7.5, 1000 non-integer)", m.Harmonics(100.0, [(1, 1.0), (7.5, 0.3)], 0.5, samplerate=SR)),
            ("Sawtooth under a Sine LFO (turn-based FM)", m.Sawtooth(880.0, 0.5, fm_lfo=m.Sine(5.0, 0.1, bias=0.01, samplerate=SR), samplerate=SR)),
            ("Pulse with pwm_lfo", m.Pulse(300.0, 0.5, pulsewidth=0.5, pwm_lfo=m.Sine(0.7, 0.3, bias=0.5, samplerate=SR), samplerate=SR)),
            ("Sine under an envelope with a 400 s sustain", m.EnvelopeFilter(m.Sine(660.0, 0.9, samplerate=SR), 0.01, 0.05, 400.0, 0.6, 0.2))]


for (name, g), (_n, o) in zip(cases(G), cases(O)):
    try:
        want = CO.render(o, first + blk)[first:]
    except Exception as e:                                   # (what the C oracle does not know: the pure-Python one, 30 s in)
        short = 30 * SR
        want = np.array(o.take(short + blk), dtype=np.float64)[short:]
        got = g.render_f64(blk, start=short)
        print("%-48s  30 s in (Python oracle): max |err| %.3e  differing float32 %d of %d" % (name, float(np.max(np.abs(got - want))), int(np.sum(got.astype(np.float32) != want.astype(np.float32))), blk))
        continue
    got = g.render_f64(blk, start=first)
    print("%-48s 300 s in: max |err| %.3e  differing float32 %d of %d" % (name, float(np.max(np.abs(got - want))), int(np.sum(got.astype(np.float32) != want.astype(np.float32))), blk))
