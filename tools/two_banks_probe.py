"""Two 1024-voice banks rendering turn by turn (a, b, a, b, ...): time per block, against one bank alone."""
import sys
sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices
N.ensure_init(0)
F = 48000
banks = [VoiceBank(*additive_voices(G, 1024, F, seed=s, adsr={"sustain": 1e6})[:1], gains=additive_voices(G, 1024, F, seed=s)[1]) for s in (0, 1)]
ring = [[N.DeviceBuffer(F * 8) for _ in range(4)] for _ in banks]


def run(nbanks, steps):
    for s in range(steps):
        for i in range(nbanks):
            banks[i].render_device(F, (1000 + s) * F, bus_f32=ring[i][s & 3])


for nb in (1, 2):
    run(nb, 400)
    N.sync()
    N.timer_start()
    run(nb, 400)
    ms = N.timer_stop()
    print("%d bank(s) turn by turn: %.1f us per block" % (nb, ms / (400 * nb) * 1e3))
