"""Time per block of the fused render for short blocks (real-time chunk sizes), 1024 additive voices."""
import sys
sys.path.insert(0, ".")
import bench
from synthesizer_amd import _native as N
from synthesizer_amd.mixer import VoiceBank

N.ensure_init(0)
voices, gains = bench.build_voices(1024)                 # (the bench bank: its sustain covers every block timed below)
bank = VoiceBank(list(voices), gains=list(gains))
for frames in (256, 1024, 4096, 16384, 48000):
    ring = [N.DeviceBuffer(frames * 8) for _ in range(4)]
    base = 48000 * 5
    for s in range(20):
        bank.render_device(frames, base + s * frames, bus_f32=ring[s & 3])
    N.sync()
    pos = [20]

    def step():
        bank.render_device(frames, base + pos[0] * frames, bus_f32=ring[pos[0] & 3])
        pos[0] += 1
    ms = bench.steady(N, step, min_seconds=0.1, reps=200)       # (steady clocks: loops of 200 blocks, the median loop)
    print("block %6d frames: %7.1f us per block, %6.0f G voice-samples/s, %7.0f x real time" %
          (frames, ms * 1e3, 1024 * frames / ms / 1e6, frames / 48000 / (ms / 1e3)), flush=True)
