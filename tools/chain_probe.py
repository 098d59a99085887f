import sys, time
sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd.sample import Sample
N.ensure_init(0)
x = (np.random.default_rng(0).integers(-20000, 20000, 48000 * 2)).astype(np.int16)
s = Sample.from_raw_frames(x.tobytes(), 2, 48000, 2).to_device()
s.amplify(0.99); N.sync()
t0 = time.perf_counter()
for _ in range(300):
    s.amplify(0.999).bias(1).reverse()
N.sync()
t1 = time.perf_counter()
print("900 chained Sample ops on a 1 s stereo sample: %.1f us per op" % ((t1 - t0) / 900 * 1e6))
t0 = time.perf_counter()
e = Sample.from_raw_frames(x.tobytes(), 2, 48000, 2).echo(0.5, 30, 0.05, 0.85)
N.sync()
print("echo x30: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
