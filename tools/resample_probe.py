"""Run one resample layout a few times (for rocprofv3 PMC passes).  usage: resample_probe.py [nch] [width] [inrate] [outrate] [MB]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from synthesizer_amd import _native as N

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
width = int(sys.argv[2]) if len(sys.argv) > 2 else 2
inr = int(sys.argv[3]) if len(sys.argv) > 3 else 44100
outr = int(sys.argv[4]) if len(sys.argv) > 4 else 48000
mb = int(sys.argv[5]) if len(sys.argv) > 5 else 900
N.ensure_init()
frames = mb * 1000000 // (nch * width)
nout = N.lib().sh_resample_out_frames(frames, inr, outr)
a = N.DeviceBuffer(frames * nch * width)
b = N.DeviceBuffer(nout * nch * width)
chunk = np.random.default_rng(0).integers(-128, 127, 1 << 24, dtype=np.int8)
a.zero()
a.upload(chunk)
import ctypes
got = ctypes.c_size_t()
for _ in range(5):
    N.check(N.lib().sh_resample(a.handle, frames, nch, width, 0, inr, outr, b.handle, ctypes.byref(got)))
N.check(N.lib().sh_sync())
t = []
for _ in range(5):
    N.check(N.lib().sh_timer_start())
    N.check(N.lib().sh_resample(a.handle, frames, nch, width, 0, inr, outr, b.handle, ctypes.byref(got)))
    ms = ctypes.c_float()
    N.check(N.lib().sh_timer_stop(ctypes.byref(ms)))
    t.append(ms.value)
byts = (frames + nout) * nch * width
print("ms", min(t), "GB/s", byts / min(t) / 1e6, "frac", byts / min(t) / 1e6 / 8000)
