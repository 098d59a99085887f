"""sh_bank_generate from the start of the notes against the steady state: 1024 voices, one-second and ten-second rows."""
import sys
sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices
N.ensure_init(0)
v, g = additive_voices(G, 1024, 48000, seed=0, adsr={"sustain": 1e6})
bank = VoiceBank(v, gains=g)
for F in (48000, 480000):
    buf = N.DeviceBuffer(1024 * F * 4)
    for start in (0, 5 * 48000):
        for _ in range(200 if F == 48000 else 30):
            bank.generate_device(F, start, out=buf)
        N.sync()
        N.timer_start()
        reps = 100 if F == 48000 else 20
        for _ in range(reps):
            bank.generate_device(F, start, out=buf)
        print("generate 1024 x %6d from %6d: %.1f us" % (F, start, N.timer_stop() / reps * 1e3))
    buf.free()
