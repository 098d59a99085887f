"""The lean FM loop against the general code ON THE DEVICE, no oracle: the float64 bus of a bank of FM Sine voices (lean lists) minus the
gain-weighted sum of the same voices rendered one by one with render_f64 (sh_osc_render: the general code), at 30, 300, 3000 s into the
notes, banks of 24 and 256 voices (other kernel shapes).  usage (GPU box): python tools/fm_lean_vs_general.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
SR, blk = 48000, 16384
rng = np.random.default_rng(5)
for nv in (24, 256):
    f = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), nv))
    ph = rng.uniform(0, 1, nv)
    gains = [(1.0 / nv, 1.0 / nv)] * nv
    for rate, depth in ((10.0, 0.05), (1.0, 0.5), (0.1, 0.05)):
        voices = [G.Sine(float(f[i]), 1.0, phase=float(ph[i]), fm_lfo=G.Sine(rate * (1 + 0.1 * i / nv), depth, phase=0.3, samplerate=SR), samplerate=SR) for i in range(nv)]
        bank = VoiceBank(voices, gains=gains)
        for secs in (30, 300, 3000):
            first = secs * SR
            b = N.DeviceBuffer(blk * 16)
            bank.render_device(blk, first, bus_f64=b)
            got = b.download(np.float64, blk * 2).reshape(blk, 2)[:, 0]
            b.free()
            want = np.zeros(blk)
            for v in voices:
                want += v.render_f64(blk, start=first) / nv
            print("voices %3d lfo %5.1f Hz depth %.2f, %4d s in: lean - general max %.3e rms %.3e" %
                  (nv, rate, depth, secs, float(np.max(np.abs(got - want))), float(np.sqrt(np.mean((got - want) ** 2)))))
