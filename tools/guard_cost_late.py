#!/usr/bin/env python
"""The int16 routes of the benchmark bank 5 s and 300 s into the notes (the guard's reach grows with t), ms per 1024 x 480 000, with the
boundary guard (default) and for a bank built without guard lists (params.int16_guard = False: the kernels without the check):
    python tools/guard_cost_late.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd import workloads as W
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
SR, NV, F2 = 48000, 1024, 480000
from synthesizer_amd import params
banks = {}
for guard in (True, False):
    params.int16_guard = guard
    try:
        v, g = W.additive_voices(G, NV, SR, seed=0, partials=16, adsr={"sustain": 1.0e6})
        banks[guard] = VoiceBank(v, gains=g)
    finally:
        params.int16_guard = True
rows = N.DeviceBuffer(NV * F2 * 2)
mono = N.DeviceBuffer(F2 * 2)
def steady(call):
    for _ in range(20): call()
    N.sync(); loops = []
    for _ in range(12):
        N.timer_start()
        for _ in range(5): call()
        loops.append(N.timer_stop() / 5)
    return statistics.median(loops)
for guard, bank in banks.items():
    for sec in (5, 300):
        a = steady(lambda: bank.generate_i16_device(F2, sec * SR, out=rows, stride=F2, check=False))
        b = steady(lambda: bank.mixdown_i16_device(F2, sec * SR, out=mono, check=False))
        print("%-14s %3d s in: int16 rows %.4f ms, fused mixdown %.4f ms" % ("guard" if guard else "no guard lists", sec, a, b), flush=True)
    bank.overflow_check()
