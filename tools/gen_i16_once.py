#!/usr/bin/env python
"""A few int16 materialisations and fused mixdowns of the benchmark bank (1024 x 480 000), for a profiler to look at:
    rocprofv3 --kernel-trace --pmc ... -- python tools/gen_i16_once.py   (SYNTHHIP_LIB selects a variant library)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd import workloads as W
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
SR, NV, F2 = 48000, 1024, 480000
v, g = W.additive_voices(G, NV, SR, seed=0, partials=16, adsr={"sustain": 1.0e6})
bank = VoiceBank(v, gains=g)
rows = N.DeviceBuffer(NV * F2 * 2)
mono = N.DeviceBuffer(F2 * 2)
for _ in range(6):
    bank.generate_i16_device(F2, 5 * SR, out=rows, stride=F2, check=False)
    bank.mixdown_i16_device(F2, 5 * SR, out=mono, check=False)
N.sync()
bank.overflow_check()
