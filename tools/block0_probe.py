"""Kernel time of the first blocks of a note, with and without the envelope (what makes block 0 slow?)."""
import sys
sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices
N.ensure_init(0)
SR = 48000
bus = N.DeviceBuffer(SR * 8)
for env in (False, True):
    voices, gains = additive_voices(G, 1024, SR, seed=0, envelope=env, adsr={"sustain": 100.0})
    bank = VoiceBank(voices, gains=gains)
    for rep in range(3):
        out = []
        for s in range(4):
            N.sync()
            N.timer_start()
            bank.render_device(SR, s * SR, bus_f32=bus)
            out.append(round(N.timer_stop() * 1e3, 1))
    print("envelope", env, "blocks 0..3 us (incl. launch + prepare when not speculated):", out)
