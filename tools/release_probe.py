"""Kernel time of the block in which the notes are released (sustain ends at 2.06 s: block 2 of 48 000 frames)."""
import sys
sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices
N.ensure_init(0)
SR = 48000
bus = N.DeviceBuffer(SR * 8)
voices, gains = additive_voices(G, 1024, SR, seed=0, adsr={"sustain": 2.0})
bank = VoiceBank(voices, gains=gains)
for rep in range(3):
    out = []
    for s in range(5):
        N.sync()
        N.timer_start()
        bank.render_device(SR, s * SR, bus_f32=bus)
        out.append(round(N.timer_stop() * 1e3, 1))
print("blocks 0..4 us (release in block 2, silence from block 3):", out)
