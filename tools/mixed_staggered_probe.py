"""bench.py's staggered_notes table (1024 players x 22 rounds, literal ADSR, tile-classified launches) with the players' instruments
varied: all Harmonics x16, all FM Sine, every other player FM Sine, and Harmonics / FM Sine / Sawtooth in turn.  us per one-second block."""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd import workloads as W
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
SR, slots, notes, nblocks = 48000, 1024, 22, 20
_, f, amp, phase, gains = W._voice_params(slots, 0)
rng = np.random.default_rng(3)
fm = rng.uniform(0.5, 8.0, slots)
depth = rng.uniform(0.0, 0.05, slots)
harm = [(k, 1.0 / k) for k in range(1, 17)]
e = W.ADSR


def instrument(s, which):
    if which == 0:
        return G.Harmonics(float(f[s]), harm, amplitude=float(amp[s]), phase=float(phase[s]), samplerate=SR)
    if which == 1:
        return G.Sine(float(f[s]), float(amp[s]), phase=float(phase[s]), fm_lfo=G.Sine(float(fm[s]), float(depth[s]), samplerate=SR), samplerate=SR)
    return G.Sawtooth(float(f[s]), float(amp[s]), phase=float(phase[s]), samplerate=SR)


for name, pick in (("all Harmonics", lambda s: 0), ("all FM Sine", lambda s: 1), ("Harmonics | FM Sine", lambda s: s & 1),
                   ("Harmonics | FM Sine | Sawtooth", lambda s: s % 3)):
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    voices, vgains = [], []
    for k in range(notes):
        for s in range(slots):
            onset = (s / slots + k) * 1.0
            osc = G.EnvelopeFilter(instrument(s, pick(s)), e["attack"], e["decay"], e["sustain"], e["sustain_level"], e["release"])
            voices.append(G.DelayFilter(osc, onset) if onset else osc)
            vgains.append(gains[s])
    bank = VoiceBank(voices, gains=vgains)
    ring = [N.DeviceBuffer(SR * 8) for _ in range(4)]

    def loop():
        for k in range(1, 3):
            bank.render_device(SR, k * SR, bus_f32=ring[k & 3])
        N.timer_start()
        for k in range(3, nblocks + 1):
            bank.render_device(SR, k * SR, bus_f32=ring[k & 3])
        return N.timer_stop() / (nblocks - 2)
    loop()
    N.sync()
    got = [loop() for _ in range(9)]
    N.sync()
    x = ring[0].download(np.float32, SR * 2).astype(np.float64)
    print("%-32s %6.1f us per block   checksum %.9f" % (name, statistics.median(got) * 1e3, float(np.abs(x).sum())))
    for b in ring:
        b.free()
    bank.close() if hasattr(bank, "close") else None
