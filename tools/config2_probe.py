"""Where a small bank's block goes (BASELINE configs[1]: 64 additive voices + ADSR, 48 kHz, blocks of one second).

    python tools/config2_probe.py [label]

Prints one JSON line: microseconds per block (HIP events over runs of back-to-back renders, steady clocks) for the bench's
64-voice bank, for the same bank with every voice silent (the launch's fixed cost), for block lengths 512 .. 48 000, and
the per-launch floor of the stream (a one-element elementwise kernel back to back).  Environment knobs of the library
(SYNTHHIP_*) are latched per process: run once per setting, the label names it.
"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices

SR = 48000
N.ensure_init(0)
L = N.lib()


def steady(call, min_seconds=0.05, reps=50):
    call()
    N.sync()
    loops, total = [], 0.0
    while (total < min_seconds or len(loops) < 5) and len(loops) < 400:
        N.timer_start()
        for _ in range(reps):
            call()
        ms = N.timer_stop()
        loops.append(ms / reps)
        total += ms / 1e3
    return statistics.median(loops) * 1e3        # us


def bank_us(nv, frames, adsr, reps=50):
    v, g = additive_voices(G, nv, SR, seed=0, partials=16, adsr=adsr)
    bank = VoiceBank(v, gains=g)
    ring = [N.DeviceBuffer(frames * 8) for _ in range(4)]
    pos = [20]

    def step():
        bank.render_device(frames, pos[0] * frames, bus_f32=ring[pos[0] & 3])
        pos[0] += 1
    for _ in range(16):
        step()
    us = steady(step, reps=reps)
    t0 = time.perf_counter()
    for _ in range(2000):
        step()
    host = (time.perf_counter() - t0) / 2000 * 1e6
    N.sync()
    return us, host


out = {"label": sys.argv[1] if len(sys.argv) > 1 else "default",
       "env": {k: v for k, v in os.environ.items() if k.startswith("SYNTHHIP_")}}
# warm the clocks
bank_us(1024, SR, {"sustain": 1.0e6}, reps=20)
a = N.DeviceBuffer(64)
out["tiny_kernel_us"] = steady(lambda: N.check(L.sh_ew_f64(N.SH_EW_FILL, None, 0, None, 0, 1, 0.0, 0.0, a.handle, 0, None, 0, None)), reps=200)
us, host = bank_us(64, SR, {"sustain": 1.0e6})
out["config2_us"] = us
out["config2_host_enqueue_us"] = host
out["config2_silent_us"] = bank_us(64, SR, {"sustain": 0.0, "release": 0.0})[0]      # every voice released long ago
out["by_frames_64v"] = {str(f): bank_us(64, f, {"sustain": 1.0e6})[0] for f in (512, 4096, 16384)}
out["by_voices_1s"] = {str(nv): bank_us(nv, SR, {"sustain": 1.0e6})[0] for nv in (8, 32, 128, 256)}
print(json.dumps(out), flush=True)
