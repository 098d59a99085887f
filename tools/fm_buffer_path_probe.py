"""FM through the BUFFER path (any modulator that is no plain Sine: here a Triangle LFO, unbiased and biased) against the pure-Python oracle,
1 .. 30 s into the note: the running sum of the modulator is taken over ideal time steps inc (DESIGN 2 says what that costs).
usage (GPU box): python tools/fm_buffer_path_probe.py"""
import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/oracle") else ".")
import numpy as np
from oracle import synth_oracle as O
from oracle import c_oracle as CO
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
N.ensure_init(0)
SR, blk = 48000, 16384
for f in (440.0, 3520.0):
    for depth, bias in ((0.05, 0.0), (0.5, 0.0), (0.05, 0.01)):
        g = G.Sine(f, 1.0, phase=0.2, fm_lfo=G.Triangle(5.0, depth, phase=0.3, bias=bias, samplerate=SR), samplerate=SR)
        o = O.Sine(f, 1.0, phase=0.2, fm_lfo=O.Triangle(5.0, depth, phase=0.3, bias=bias, samplerate=SR), samplerate=SR)
        want_all = np.array(o.take(30 * SR + blk), dtype=np.float64)          # (the pure-Python oracle: the C one knows Sine LFOs only)
        for secs in (1, 10, 30):
            first = secs * SR
            got = g.render_f64(blk, start=first)
            w = want_all[first:first + blk]
            print("carrier %6.0f Hz triangle LFO depth %.2f bias %.2f, %3d s in: max |err| %.3e rms %.3e" % (f, depth, bias, secs, float(np.max(np.abs(got - w))), float(np.sqrt(np.mean((got - w) ** 2)))))
