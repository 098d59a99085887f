"""A/B of the 16-bit mono / stereo resample rows (bench.py's shapes)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import _native as N
N.ensure_init(0)
L = N.lib()
src = N.DeviceBuffer(900_000_000)
x = np.random.default_rng(0).integers(-32768, 32768, 1 << 24, dtype=np.int64).astype(np.int16)
for off in range(0, src.nbytes, x.nbytes):
    src.upload(x[:min(len(x), (src.nbytes - off) // 2)], off)
for name, nch, inr, outr in (("mono 44.1->48", 1, 44100, 48000), ("stereo 44.1->48", 2, 44100, 48000), ("stereo 96->44.1", 2, 96000, 44100),
                             ("mono 96->44.1", 1, 96000, 44100), ("mono 22.05->44.1", 1, 22050, 44100)):
    frames = 450_000_000 // nch
    nout = L.sh_resample_out_frames(frames, inr, outr)
    dst = N.DeviceBuffer(nout * 2 * nch)
    for _ in range(10):
        N.check(L.sh_resample(src.handle, frames, nch, 2, 0, inr, outr, dst.handle, None))
    N.sync()
    best = 1e9
    for _ in range(5):
        N.timer_start()
        for _ in range(5):
            N.check(L.sh_resample(src.handle, frames, nch, 2, 0, inr, outr, dst.handle, None))
        best = min(best, N.timer_stop() / 5)
    nbytes = (frames + nout) * 2 * nch
    print("%-18s %.4f ms  %.2f TB/s (%.3f of 8)" % (name, best, nbytes / best / 1e9, nbytes / best / 1e9 / 8))
    dst.free()
# the many-channel shapes (generic kernel, one thread per 16-byte channel group) and float32 mono / stereo (LDS kernel)
for name, nch, width, isf in (("8ch f32 96->44.1", 8, 4, 1), ("8ch i16 96->44.1", 8, 2, 0), ("stereo f32 96->44.1", 2, 4, 1), ("mono f32 44.1->48", 1, 4, 1)):
    inr, outr = (96000, 44100) if "96" in name else (44100, 48000)
    frames = 800_000_000 // (nch * width)
    nout = L.sh_resample_out_frames(frames, inr, outr)
    dst = N.DeviceBuffer(nout * width * nch)
    for _ in range(10):
        N.check(L.sh_resample(src.handle, frames, nch, width, isf, inr, outr, dst.handle, None))
    N.sync()
    best = 1e9
    for _ in range(5):
        N.timer_start()
        for _ in range(5):
            N.check(L.sh_resample(src.handle, frames, nch, width, isf, inr, outr, dst.handle, None))
        best = min(best, N.timer_stop() / 5)
    nbytes = (frames + nout) * width * nch
    print("%-18s %.4f ms  %.2f TB/s (%.3f of 8)" % (name, best, nbytes / best / 1e9, nbytes / best / 1e9 / 8))
    dst.free()
