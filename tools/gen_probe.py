"""Probe: k_generate throughput vs block length (row stride) on the GPU box."""
import sys
sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices

N.ensure_init(0)
nv = 1024
voices, gains = additive_voices(G, nv, 48000, seed=0, adsr={"sustain": 100.0})
bank = VoiceBank(voices, gains=gains)
for frames in (4096, 48000, 120000, 480000):
    buf = N.DeviceBuffer(nv * frames * 4)
    bus = N.DeviceBuffer(frames * 8)
    for _ in range(2):
        bank.generate_device(frames, 48000, out=buf)
    N.sync()
    N.timer_start()
    for _ in range(5):
        bank.generate_device(frames, 48000, out=buf)
    ms = N.timer_stop() / 5
    N.timer_start()
    for _ in range(5):
        bank.render_device(frames, 48000, bus_f32=bus)
    ms2 = N.timer_stop() / 5
    N.timer_start()
    for _ in range(5):
        bank.mix_device(buf, frames, bus_f32=bus)
    ms3 = N.timer_stop() / 5
    print("frames %7d: generate %.3f ms = %.0f G samples/s (%.0f GB/s written) | fused %.3f ms = %.0f G/s | mix %.3f ms = %.0f GB/s"
          % (frames, ms, nv * frames / ms / 1e6, nv * frames * 4 / ms / 1e6, ms2, nv * frames / ms2 / 1e6, ms3, (4 * nv + 8) * frames / ms3 / 1e6), flush=True)
    buf.free()
    bus.free()
