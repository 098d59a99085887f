"""Fused render of a 1024-voice bank of MIXED lean kinds (Harmonics x16, FM Sine, Sine, Sawtooth, Square, Pulse; one-second blocks, steady
state): the lean kernel instantiated for all kinds (k_render_lean<.., LEAN_K_ALL, false>).  Prints us per block and a checksum."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
SR = 48000
rng = np.random.default_rng(0)
f = np.exp(rng.uniform(np.log(55), np.log(3520), 1024))
gains = [(0.01, 0.02)] * 1024
harm = [(j, 1.0 / j) for j in range(1, 17)]
makers = [lambda k: G.Harmonics(float(f[k]), harm, 0.5, samplerate=SR),
          lambda k: G.Sine(float(f[k]), 0.5, fm_lfo=G.Sine(5.0, 0.02, samplerate=SR), samplerate=SR),
          lambda k: G.Sine(float(f[k]), 0.5, samplerate=SR),
          lambda k: G.Sawtooth(float(f[k]), 0.5, samplerate=SR),
          lambda k: G.Square(float(f[k]), 0.5, samplerate=SR),
          lambda k: G.Pulse(float(f[k]), 0.5, pulsewidth=0.3, samplerate=SR)]
CASES = (("mixed six kinds", lambda k: makers[k % 6](k)), ("harmonics + fm", lambda k: makers[k % 2](k)), ("saw + square + pulse", lambda k: makers[3 + k % 3](k)),
                   ("harmonics only", lambda k: makers[0](k)), ("harmonics, one saw", lambda k: makers[3 if k == 500 else 0](k)),
                   ("fm only", lambda k: makers[1](k)), ("fm, one saw", lambda k: makers[3 if k == 500 else 1](k)),
                   ("harmonics | fm halves", lambda k: makers[0 if k < 512 else 1](k)))
for name, pick in CASES:
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    bank = VoiceBank([pick(k) for k in range(1024)], gains=gains)
    ring = [N.DeviceBuffer(SR * 8) for _ in range(4)]
    for s in range(40):
        bank.render_device(SR, (100 + s) * SR, bus_f32=ring[s & 3])
    N.sync()
    best = 1e9
    for rep in range(5):
        N.timer_start()
        for s in range(40, 140):
            bank.render_device(SR, (100 + s) * SR, bus_f32=ring[s & 3])
        best = min(best, N.timer_stop() / 100)
    N.sync()
    got = ring[3].download(np.float32, SR * 2).astype(np.float64)
    print("%-24s %6.1f us per block   checksum %.9f" % (name, best * 1e3, float(np.abs(got).sum())))
