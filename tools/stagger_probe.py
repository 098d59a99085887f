"""The staggered-notes bench row alone (microseconds per one-second block), beside the headline bank on the same box."""
import json
import os
import sys

sys.path.insert(0, ".")
import bench
from synthesizer_amd import _native as N

N.ensure_init(0)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("SYNTHHIP_")}}
from synthesizer_amd import dist
voices, gains = bench.build_voices(1024)
bank = dist.DistVoiceBank(voices, gains, 0, 1)
pos = [5]


def step():
    bank.render_device(48000, pos[0] * 48000)
    pos[0] += 1


for _ in range(10):
    step()
out["headline_us"] = bench.steady(N, step, min_seconds=0.15, reps=20) * 1e3
r = bench.staggered_row(N, 48000)
out["staggered_us"] = r["ms_per_step"] * 1e3
out["standing_start_us"] = r["from_a_standing_start_ms_per_step"] * 1e3
out["ratio"] = out["staggered_us"] / out["headline_us"]
out["host_build_s"] = r["host_build_s"]
print(json.dumps(out))
