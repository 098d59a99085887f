"""Time sh_bank_generate on the two-step bench shape (1024 voices x 480000 frames from block 5)."""
import sys
sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices
N.ensure_init(0)
v, g = additive_voices(G, 1024, 48000, seed=0, adsr={"sustain": 1e6})
bank = VoiceBank(v, gains=g)
F = 480000
buf = N.DeviceBuffer(1024 * F * 4)
for _ in range(30):
    bank.generate_device(F, 5 * 48000, out=buf)
N.sync()
N.timer_start()
for _ in range(20):
    bank.generate_device(F, 5 * 48000, out=buf)
print("generate 1024 x %d: %.1f us" % (F, N.timer_stop() / 20 * 1e3))
