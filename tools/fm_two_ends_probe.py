"""One launch that holds TWO ends of pieces of an FM voice's joint LFO / time table late in a note (the LFO's phase crosses 8192 rad 0.3 s
before the accumulated time crosses 2048 rad, 326 s in): the lean loops and the general code follow the first end exactly (the lean ones at a
tile boundary) and the second along the line of the piece in front of it.  A single oscillator (general code) and a 160-voice bank (lean
lists) against the C oracle, the one-second block that holds both ends and the block behind it.  usage (GPU box): python tools/fm_two_ends_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import synth_oracle as O
from oracle import c_oracle as CO
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
SR = 48000
n_time = int(2048.0 / (2 * np.pi / SR))                    # the time table's piece end (t crosses 2048 rad)
n_lfo = n_time - 15000
a = 0.3 * 2 * np.pi
rate = (8192.0 - a) / n_lfo * SR / (2 * np.pi)
first = n_lfo - 9000
tab = G.LfoTable(a, 2 * np.pi * rate / SR, 0.5, 0.0, 2 * np.pi / SR)
ends = [int(x) for x in tab.records["n0"][0::2] if first < int(x) < first + SR]
print("LFO rate %.6f Hz; ends of joint pieces inside the block [%d, %d): %s" % (rate, first, first + SR, ends))
for depth in (0.05, 0.5):
    g = G.Sine(3520.0, 1.0, phase=0.2, fm_lfo=G.Sine(rate, depth, phase=0.3, samplerate=SR), samplerate=SR)
    o = O.Sine(3520.0, 1.0, phase=0.2, fm_lfo=O.Sine(rate, depth, phase=0.3, samplerate=SR), samplerate=SR)
    want = CO.render(o, first + 2 * SR)[first:]
    for k in range(2):
        got = g.render_f64(SR, start=first + k * SR)
        w = want[k * SR:(k + 1) * SR]
        print("depth %.2f single oscillator block %d: max |err| %.3e rms %.3e" % (depth, k, float(np.max(np.abs(got - w))), float(np.sqrt(np.mean((got - w) ** 2)))))
    nv = 160
    gains = [(1.0 / 8, 1.0 / 8)] * nv
    bank = VoiceBank([G.Sine(3520.0, 1.0 / 20, phase=0.2, fm_lfo=G.Sine(rate, depth, phase=0.3, samplerate=SR), samplerate=SR) for _ in range(nv)], gains=gains)
    for k in range(2):
        got = bank.render(SR, start=first + k * SR)[:, 0].astype(np.float64)
        w = want[k * SR:(k + 1) * SR] * (nv / 20 / 8)
        print("depth %.2f bank (lean lists) block %d: max |err| %.3e rms %.3e (float32 bus: ~3e-8)" % (depth, k, float(np.max(np.abs(got - w))), float(np.sqrt(np.mean((got - w) ** 2)))))
