"""what would float32 arithmetic after a float64 phase cost in accuracy?  CPU emulation (numpy) on the
bench workload: 1024 additive voices (Harmonics x16, a_k = 1/k), phase accumulated and range-reduced in float64 as on the
device, THEN (a) float32 sin/cos + float32 Clenshaw, (b) float32 sin/cos + float32 Horner of sin*P(cos), bus summed in
float64; error of the stereo bus against the all-float64 evaluation, RMS, to be set against the 1e-6 contract.

    python tools/mixed_precision_probe.py [frames]
"""
import sys
from fractions import Fraction

import numpy as np

sys.path.insert(0, ".")
from synthesizer_amd.oscillators import series_polynomial  # noqa: E402  (exact polynomial coefficients, host code)

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nv, sr = 1024, 48000
rng = np.random.default_rng(0)
f = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), nv))
amp = rng.uniform(0.1, 1.0, nv) / np.sqrt(nv)
phase = rng.uniform(0.0, 1.0, nv)
pan = rng.uniform(-1.0, 1.0, nv)
gl, gr = (1.0 - pan) / 2.0, (1.0 + pan) / 2.0
a = np.array([1.0 / k for k in range(1, 17)])
poly = np.array(series_polynomial(tuple(a)))                 # highest power first
n = np.arange(5 * sr, 5 * sr + frames, dtype=np.float64)      # a window five seconds into the note

bus64 = np.zeros((frames, 2))
busC = np.zeros((frames, 2))
busH = np.zeros((frames, 2))
for v in range(nv):
    t = 2 * np.pi * phase[v] + n * (2 * np.pi * f[v] / sr)    # float64 phase (the device reproduces the accumulated sum exactly)
    r = np.remainder(t, 2 * np.pi)                            # float64 range reduction
    s64, c64 = np.sin(r), np.cos(r)
    x64 = s64 * np.polyval(poly, c64)
    s32, c32 = np.sin(r.astype(np.float32)), np.cos(r.astype(np.float32))      # float32 from here on
    # (a) Clenshaw in float32: b_k = a_k + 2c b_{k+1} - b_{k+2}
    b1 = np.zeros(frames, np.float32)
    b2 = np.zeros(frames, np.float32)
    c2 = (c32 + c32).astype(np.float32)
    for k in range(16, 0, -1):
        b1, b2 = (np.float32(a[k - 1]) + c2 * b1 - b2).astype(np.float32), b1
    xC = (b1 * s32).astype(np.float32)
    # (b) Horner of the power-basis polynomial in float32
    p = np.full(frames, np.float32(poly[0]), np.float32)
    for coef in poly[1:]:
        p = (p * c32 + np.float32(coef)).astype(np.float32)
    xH = (p * s32).astype(np.float32)
    for bus, x in ((bus64, x64), (busC, xC.astype(np.float64)), (busH, xH.astype(np.float64))):
        bus[:, 0] += gl[v] * amp[v] * x
        bus[:, 1] += gr[v] * amp[v] * x
rms = lambda e: float(np.sqrt(np.mean(e ** 2)))
print("bus rms %.4f" % rms(bus64))
print("float32 Clenshaw: rms error %.3e (contract 1e-6, margin %.2fx)" % (rms(busC - bus64), 1e-6 / rms(busC - bus64)))
print("float32 Horner:   rms error %.3e (contract 1e-6, margin %.2fx)" % (rms(busH - bus64), 1e-6 / rms(busH - bus64)))
