"""A LONG piece: 1024 players x 200 rounds = 204 800 notes in one table (a tile set sized by the table would need 2.4 GB, times four);
microseconds per one-second block a minute into it, and the memory the bank's render path holds."""
import sys
sys.path.insert(0, ".")
import time
import bench
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd import workloads as W
from synthesizer_amd.mixer import VoiceBank

N.ensure_init(0)
SR = 48000
t0 = time.perf_counter()
voices, gains = W.staggered_notes(G, 1024, SR, seed=0, partials=16, period=1.0, notes=200)
bank = VoiceBank(voices, gains=gains)
print("voices", len(voices), "host build %.1f s" % (time.perf_counter() - t0), flush=True)
ring = [N.DeviceBuffer(SR * 8) for _ in range(4)]
pos = [0]


def step():
    k = 60 + pos[0] % 100
    bank.render_device(SR, k * SR, bus_f32=ring[k & 3])
    pos[0] += 1
for _ in range(10):
    step()
ms = bench.steady(N, step, min_seconds=0.1, reps=50)
c = N.debug_counters()
print("one-second block: %.1f us   tile-classified launches %d (on sets resolved ahead %d)" % (ms * 1e3, c["tiled_launches"], c["tiled_predicted"]))
