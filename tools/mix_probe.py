"""Time sh_mix_chain_i16 / sh_mix_bus_f32 on shapes from cache-resident to HBM-streaming (A/B of load hints)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import _native as N
N.ensure_init(0)
L = N.lib()
rng = np.random.default_rng(0)
for nv, ns in ((16, 1 << 20), (64, 1 << 20), (64, 96000), (1024, 96000), (256, 1 << 20), (1024, 1 << 20), (1024, 960000)):
    x = rng.integers(-3000, 3000, 1 << 22, dtype=np.int64).astype(np.int16)
    src = N.DeviceBuffer(nv * ns * 2)
    for off in range(0, src.nbytes, x.nbytes):
        src.upload(x[:min(len(x), (src.nbytes - off) // 2)], off)
    dst = N.DeviceBuffer(ns * 2)
    for _ in range(20):
        N.check(L.sh_mix_chain_i16(src.handle, nv, ns, ns, dst.handle))
    N.sync()
    best = 1e9
    for _ in range(5):
        N.timer_start()
        for _ in range(10):
            N.check(L.sh_mix_chain_i16(src.handle, nv, ns, ns, dst.handle))
        best = min(best, N.timer_stop() / 10)
    nbytes = (nv + 1) * ns * 2
    print("chain i16 %5d x %8d (%6.1f MB): %.4f ms  %.2f TB/s" % (nv, ns, nbytes / 1e6, best, nbytes / best / 1e9))
    src.free(); dst.free()
for nv, nf in ((16, 1 << 19), (64, 1 << 19), (64, 48000), (1024, 48000), (256, 1 << 19), (1024, 480000)):
    y = rng.uniform(-1, 1, 1 << 22).astype(np.float32)
    src = N.DeviceBuffer(nv * nf * 4)
    for off in range(0, src.nbytes, y.nbytes):
        src.upload(y[:min(len(y), (src.nbytes - off) // 4)], off)
    gains = N.DeviceBuffer.from_array(rng.uniform(0, 1, nv * 2).astype(np.float32))
    dst = N.DeviceBuffer(nf * 8)
    for _ in range(20):
        N.check(L.sh_mix_bus_f32(src.handle, nv, nf, nf, gains.handle, dst.handle))
    N.sync()
    best = 1e9
    for _ in range(5):
        N.timer_start()
        for _ in range(10):
            N.check(L.sh_mix_bus_f32(src.handle, nv, nf, nf, gains.handle, dst.handle))
        best = min(best, N.timer_stop() / 10)
    nbytes = nv * nf * 4 + nf * 8
    print("bus f32   %5d x %8d (%6.1f MB): %.4f ms  %.2f TB/s" % (nv, nf, nbytes / 1e6, best, nbytes / best / 1e9))
    src.free(); dst.free()
