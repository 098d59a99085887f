"""16-bit mono resample rows under a knob of the resample kernels (SYNTHHIP_RESAMPLE_SPLIT, ...) vs the default, 900 MB of input, steady
clocks; the whole output's CRC so that two runs (knob off / on) can be compared bit for bit."""
import json
import os
import sys
import zlib

sys.path.insert(0, ".")
import numpy as np
import bench
from synthesizer_amd import _native as N

N.ensure_init(0)
L = N.lib()
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("SYNTHHIP_")}}
frames = 450_000_000
src = N.DeviceBuffer(frames * 2)
chunk = (np.random.default_rng(1).integers(-32768, 32768, 1 << 24)).astype(np.int16)
for off in range(0, src.nbytes, chunk.nbytes):
    src.upload(chunk[:min(len(chunk), (src.nbytes - off) // 2)], off)
for inr, outr, frames_in in ((44100, 48000, frames), (96000, 44100, frames), (48000, 44100, frames), (44100, 48000, 1_000_003), (8000, 48000, 777_777)):
    nout = L.sh_resample_out_frames(frames_in, inr, outr)
    dst = N.DeviceBuffer(nout * 2)
    call = lambda: N.check(L.sh_resample(src.handle, frames_in, 1, 2, 0, inr, outr, dst.handle, None))
    ms = bench.steady(N, call, min_seconds=0.03, reps=3) if frames_in == frames else None
    call()
    crc = 0
    step = 1 << 26
    for off in range(0, nout * 2, step):
        crc = zlib.crc32(dst.download_bytes(min(step, nout * 2 - off), off), crc)
    row = {"crc": crc, "frames_out": int(nout)}
    if ms is not None:
        row.update(ms=ms, frac_hbm=(frames_in + nout) * 2 / (ms / 1e3) / 8e12)
    out["%d_to_%d_%d" % (inr, outr, frames_in)] = row
    dst.free()
print(json.dumps(out))
