"""Throughput of the BASELINE configs other than the headline, on one MI355X (numbers quoted in DESIGN.md)."""
import json
import sys
import time

sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices, fm_voices

N.ensure_init(0)
SR = 48000
out = {}


def time_bank(bank, frames, steps=60, warm=4):
    bus = N.DeviceBuffer(frames * 8)
    for s in range(warm):
        bank.render_device(frames, s * frames, bus_f32=bus)
    N.sync()
    t0 = time.perf_counter()
    N.timer_start()
    for s in range(steps):
        bank.render_device(frames, (warm + s) * frames, bus_f32=bus)
    ms = N.timer_stop() / steps
    wall = (time.perf_counter() - t0) / steps
    return ms, wall


# config 1: single 440 Hz sine, 1 s @ 44.1 kHz (host buffer out: includes the D2H copy)
osc = G.Sine(440, samplerate=44100)
osc.render(44100)
t0 = time.perf_counter()
for _ in range(50):
    osc.render(44100, start=0)
out["config1_sine_1s_44k1_ms_incl_copy"] = (time.perf_counter() - t0) / 50 * 1e3

for name, (voices, gains) in (("config2_additive64_adsr", additive_voices(G, 64, SR, seed=0, adsr={"sustain": 200.0})),
                              ("config3_fm1024", fm_voices(G, 1024, SR, seed=1)),
                              ("headline_additive1024", additive_voices(G, 1024, SR, seed=0, adsr={"sustain": 200.0}))):
    bank = VoiceBank(voices, gains=gains)
    ms, wall = time_bank(bank, SR)
    nv = len(voices)
    out[name] = {"ms_per_1s_block": ms, "voice_Msamples_per_s": nv * SR / ms / 1e3, "realtime_factor": 1e3 / ms}
print(json.dumps(out, indent=1))
