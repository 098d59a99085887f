"""BASELINE's literal job: 10 s of the 1024-voice additive bank from frame 0 in blocks of 48 000 (the first block is the note's
attack, decay and a dozen binades of the phase sum), timed as a whole -- next to the steady state bench.py's passes measure."""
import sys
sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices
N.ensure_init(0)
SR = 48000
voices, gains = additive_voices(G, 1024, SR, seed=0, adsr={"sustain": 1e6})
bank = VoiceBank(voices, gains=gains)
ring = [N.DeviceBuffer(SR * 8) for _ in range(4)]
# clocks up on other frames
for k in range(400):
    bank.render_device(SR, (100 + k) * SR, bus_f32=ring[k & 3])
N.sync()
best = 1e9
for rep in range(20):
    N.timer_start()
    for k in range(10):
        bank.render_device(SR, k * SR, bus_f32=ring[k & 3])
    best = min(best, N.timer_stop())
    for k in range(50):                        # (keeps the clocks up between repetitions, on other frames)
        bank.render_device(SR, (600 + k) * SR, bus_f32=ring[k & 3])
    N.sync()
print("10 s job from frame 0: %.1f us  = %.3f T voice-samples/s" % (best * 1e3, 1024 * 10 * SR / best / 1e9))
