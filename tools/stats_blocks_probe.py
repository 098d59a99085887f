import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import _native as N
N.ensure_init(0)
n = 900_000_000
a = N.DeviceBuffer(n)
a.zero()
a.upload(np.random.default_rng(0).integers(-128, 127, 1 << 24, dtype=np.int8))
L = N.lib()
mx, sq = C.c_uint32(), C.c_double()
for _ in range(3):
    N.check(L.sh_pcm_stats(a.handle, n, 2, C.byref(mx), C.byref(sq)))
N.timer_start()
for _ in range(10):
    N.check(L.sh_pcm_stats(a.handle, n, 2, C.byref(mx), C.byref(sq)))
ms = N.timer_stop() / 10
print("ms", ms, "frac", n / ms / 1e6 / 8000)
