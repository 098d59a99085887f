"""The staggered-notes table (1024 players x 22 rounds) rendered in SHORT blocks (real-time chunk sizes): microseconds per block."""
import sys
sys.path.insert(0, ".")
import bench
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd import workloads as W
from synthesizer_amd.mixer import VoiceBank

N.ensure_init(0)
SR = 48000
voices, gains = W.staggered_notes(G, 1024, SR, seed=0, partials=16, period=1.0, notes=22)
bank = VoiceBank(voices, gains=gains)
for frames in (256, 1024, 4096, 16384, 48000):
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
    import time
    N.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    host_us = (time.perf_counter() - t0) / 200 * 1e6
    N.sync()
    c = N.debug_counters()
    print("block %6d frames: %7.1f us per block, %7.0f x real time   host enqueue %5.1f us   (tile-classified launches so far: %d)" % (frames, ms * 1e3, frames / SR / (ms / 1e3), host_us, c["tiled_launches"]), flush=True)
