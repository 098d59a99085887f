"""16-bit mono resample rows with the packed dot2 form on / off (SYNTHHIP_RESAMPLE_PK), 900 MB of input, steady clocks."""
import json
import os
import sys

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
for inr, outr in ((44100, 48000), (96000, 44100), (48000, 44100)):
    nout = L.sh_resample_out_frames(frames, inr, outr)
    dst = N.DeviceBuffer(nout * 2)
    ms = bench.steady(N, lambda: N.check(L.sh_resample(src.handle, frames, 1, 2, 0, inr, outr, dst.handle, None)), min_seconds=0.03, reps=3)
    head = dst.download(np.int16, 1 << 16)
    out["%d_to_%d" % (inr, outr)] = {"ms": ms, "frac_hbm": (frames + nout) * 2 / (ms / 1e3) / 8e12, "checksum": int(head.astype(np.int64).sum())}
    dst.free()
print(json.dumps(out))
