"""Fused render throughput of 1024-voice banks of one oscillator kind each (1-s blocks, steady state)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
SR = 48000
rng = np.random.default_rng(0)
f = np.exp(rng.uniform(np.log(55), np.log(3520), 1024))
gains = [(0.01, 0.02)] * 1024
bus = N.DeviceBuffer(SR * 8)
kinds = {
    "Sine": lambda k: G.Sine(float(f[k]), 0.5, samplerate=SR),
    "Sawtooth": lambda k: G.Sawtooth(float(f[k]), 0.5, samplerate=SR),
    "Square": lambda k: G.Square(float(f[k]), 0.5, samplerate=SR),
    "Triangle": lambda k: G.Triangle(float(f[k]), 0.5, samplerate=SR),
    "Pulse": lambda k: G.Pulse(float(f[k]), 0.5, pulsewidth=0.3, samplerate=SR),
    "Harmonics x16": lambda k: G.Harmonics(float(f[k]), [(j, 1.0 / j) for j in range(1, 17)], 0.5, samplerate=SR),
    "Harmonics x16 biased": lambda k: G.Harmonics(float(f[k]), [(j, 1.0 / j) for j in range(1, 17)], 0.5, bias=0.01, samplerate=SR),
    "Harmonics x64 (Clenshaw)": lambda k: G.Harmonics(float(f[k]), [(j, 1.0 / j) for j in range(1, 65)], 0.5, samplerate=SR),
    "Sine + FM": lambda k: G.Sine(float(f[k]), 0.5, fm_lfo=G.Sine(5.0, 0.02, samplerate=SR), samplerate=SR),
    "WhiteNoise": lambda k: G.WhiteNoise(4800.0, 0.5, samplerate=SR, seed=k),
}
for name, make in kinds.items():
    bank = VoiceBank([make(k) for k in range(1024)], gains=gains)
    for s in range(20):
        bank.render_device(SR, s * SR, bus_f32=bus)
    N.sync()
    N.timer_start()
    for s in range(20, 70):
        bank.render_device(SR, s * SR, bus_f32=bus)
    ms = N.timer_stop() / 50
    print("%-26s %6.1f us per second of audio  %6.0f G voice-samples/s" % (name, ms * 1e3, 1024 * SR / ms / 1e6))
