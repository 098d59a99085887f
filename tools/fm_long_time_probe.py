"""How long does an FM Sine voice stay within the contract (1e-6 RMS) of the reference's generator?  The reference adds
phase_correction += (freq_previous - freq) * t sample by sample and evaluates sin(t * freq + phase_correction): two terms of ~f t radians each
whose rounding (ulp(f t) / 2 per addition, sqrt(n) of them) is part of ITS output; the closed form here has no such noise.  One carrier
(440 Hz and 3520 Hz) with a 5 Hz Sine LFO, render_f64 against the C oracle at 1 .. 300 s into the note.
usage (GPU box): python tools/fm_long_time_probe.py"""
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
for f in (440.0, 3520.0):
    for depth in (0.05, 0.5):
        g = G.Sine(f, 1.0, phase=0.2, fm_lfo=G.Sine(5.0, depth, phase=0.3, samplerate=SR), samplerate=SR)
        o = O.Sine(f, 1.0, phase=0.2, fm_lfo=O.Sine(5.0, depth, phase=0.3, samplerate=SR), samplerate=SR)
        want_all = CO.render(o, 300 * SR + blk)
        for secs in (1, 10, 30, 100, 300):
            first = secs * SR
            got = g.render_f64(blk, start=first)
            w = want_all[first:first + blk]
            print("carrier %6.0f Hz depth %.2f, %3d s in: max |err| %.3e rms %.3e" % (f, depth, secs, float(np.max(np.abs(got - w))), float(np.sqrt(np.mean((got - w) ** 2)))))
