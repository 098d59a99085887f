"""k_generate_lean_harm against its row-major variant (SYNTHHIP_GEN_ROWS): 1024 voices x 480 000 frames, five seconds into the notes;
prints milliseconds per call and checks the two against each other through a checksum of the rows."""
import json
import os
import sys

sys.path.insert(0, ".")
import numpy as np
import bench
from synthesizer_amd import _native as N
from synthesizer_amd.mixer import VoiceBank

N.ensure_init(0)
voices, gains = bench.build_voices(1024)
bank = VoiceBank(list(voices), gains=list(gains))
F2 = 480000
vbuf = N.DeviceBuffer(1024 * F2 * 4)
ms = bench.steady(N, lambda: bank.generate_device(F2, 5 * 48000, out=vbuf), min_seconds=0.05, reps=3)
rows = vbuf.download(np.float32, 4 * F2, offset=0)
tail = vbuf.download(np.float32, F2, offset=1023 * F2 * 4)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("SYNTHHIP_")}, "generate_ms": ms,
                  "frac_hbm": 1024 * F2 * 4 / (ms / 1e3) / 8e12,
                  "checksum": float(np.abs(rows.astype(np.float64)).sum() + np.abs(tail.astype(np.float64)).sum())}))
