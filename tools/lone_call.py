import os, sys, statistics
sys.path.insert(0, os.getcwd())
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd import workloads as W
from synthesizer_amd.mixer import VoiceBank
import numpy as np
N.ensure_init(0)
SR = 48000
v, g = W.additive_voices(G, 1024, SR, seed=0, partials=16, adsr={"sustain": 1e6})
bank = VoiceBank(v, gains=g)
ring4 = [N.DeviceBuffer(SR * 8) for _ in range(4)]
pos = 100
for _ in range(100):
    bank.render_device(SR, pos * SR, bus_f32=ring4[pos & 3]); pos += 1
N.sync()
mode = sys.argv[1] if len(sys.argv) > 1 else "seq"
got = []
for rep in range(60):
    if mode == "probe":
        for _ in range(3):
            bank.render_device(SR, pos * SR, bus_f32=ring4[pos & 3]); pos += 1
        N.sync()
    N.timer_start()
    bank.render_device(SR, pos * SR, bus_f32=ring4[pos & 3])
    ms = N.timer_stop()
    if mode == "seq-download":
        ring4[pos & 3].download(np.float32, 16)
    pos += 1
    got.append(ms * 1e3)
print(mode, "median %.2f us, min %.2f" % (statistics.median(got), min(got)))
