"""A table of notes of EVERY lean kind (1024 players x 22 rounds as in workloads.staggered_notes, the kind by player: Harmonics x16,
Sine, Sawtooth, Square, Triangle, Pulse, Sine with a Sine LFO): microseconds per block at 48 000 and 4096 frames."""
import sys
sys.path.insert(0, ".")
import numpy as np
import bench
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank

N.ensure_init(0)
SR, slots, notes = 48000, 1024, 22
rng = np.random.default_rng(0)
f = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), slots))
amp = rng.uniform(0.1, 1.0, slots) / np.sqrt(slots)
phase = rng.uniform(0.0, 1.0, slots)
pan = rng.uniform(-1.0, 1.0, slots)
harm = [(k, 1.0 / k) for k in range(1, 17)]
voices, gains = [], []
for k in range(notes):
    for s in range(slots):
        kind = s % 7
        a, fr, ph = float(amp[s]), float(f[s]), float(phase[s])
        osc = (G.Harmonics(fr, harm, amplitude=a, phase=ph, samplerate=SR) if kind == 0 else G.Sine(fr, a, phase=ph, samplerate=SR) if kind == 1
               else G.Sawtooth(fr, a, phase=ph, samplerate=SR) if kind == 2 else G.Square(fr, a, phase=ph, samplerate=SR) if kind == 3
               else G.Triangle(fr, a, phase=ph, samplerate=SR) if kind == 4 else G.Pulse(fr, a, phase=ph, pulsewidth=0.3, samplerate=SR) if kind == 5
               else G.Sine(fr, a, phase=ph, fm_lfo=G.Sine(5.0, 0.02, samplerate=SR), samplerate=SR))
        osc = G.EnvelopeFilter(osc, 0.01, 0.05, 0.5, 0.6, 0.2)
        onset = (s / slots + k) * 1.0
        voices.append(G.DelayFilter(osc, onset) if onset else osc)
        gains.append(((1.0 - pan[s]) / 2.0, (1.0 + pan[s]) / 2.0))
bank = VoiceBank(voices, gains=gains)
for frames in (4096, 48000):
    ring = [N.DeviceBuffer(frames * 8) for _ in range(4)]
    nblocks = (19 * SR) // frames
    pos = [0]

    def step():
        k = pos[0] % nblocks
        bank.render_device(frames, SR + k * frames, bus_f32=ring[k & 3])
        pos[0] += 1
    for _ in range(20):
        step()
    ms = bench.steady(N, step, min_seconds=0.1, reps=min(200, nblocks))
    print("block %6d frames: %7.1f us per block   (tile-classified launches so far: %d)" % (frames, ms * 1e3, N.debug_counters()["tiled_launches"]), flush=True)
