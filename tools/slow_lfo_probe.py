"""FM Sine voices whose LFO is very slow: the lean FM loop takes the LFO's cosine by the three-term recurrence cos[j] = 2cos(64 d) cos[j-1] - cos[j-2],
whose error grows like j * 1.1e-16 / sin(64 d) and reaches the carrier's angle multiplied by f_inc * K, K ~ amplitude / d.  Max error of a
256-voice bank and of single voices against the C oracle, LFO rates 10 Hz .. 0.0001 Hz.  usage (GPU box): python tools/slow_lfo_probe.py [seconds into the notes = 30] [voices = 256]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import synth_oracle as O
from oracle import c_oracle as CO
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
SR, blk = 48000, 16384
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 256
first = SR * (int(sys.argv[1]) if len(sys.argv) > 1 else 30)          # seconds into the notes
rng = np.random.default_rng(5)
f = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), nv))
ph = rng.uniform(0, 1, nv)
gains = [(1.0 / nv, 1.0 / nv)] * nv
for rate in (10.0, 1.0, 0.1, 0.01, 0.001, 0.0001):
    for depth in (0.05, 0.5):
        def build(m):
            return [m.Sine(float(f[i]), 1.0, phase=float(ph[i]), fm_lfo=m.Sine(rate * (1 + 0.1 * i / nv), depth, phase=0.3, samplerate=SR), samplerate=SR) for i in range(nv)]
        gv, ov = build(G), build(O)
        want = CO.mix_bus(np.stack([CO.render(v, first + blk)[first:] for v in ov]), gains)
        got = VoiceBank(gv, gains=gains).render(blk, start=first)
        one = gv[7].render(blk, start=first) if hasattr(gv[7], "render") else None      # (a single oscillator: the general code)
        w1 = CO.render(ov[7], first + blk)[first:]
        e1 = float(np.max(np.abs(np.asarray(one, dtype=np.float64).reshape(-1)[:blk] - w1))) if one is not None else float("nan")
        print("lfo %8.4f Hz depth %.2f: bank max |err| %.3e rms %.3e   single voice (general path) max |err| %.3e" %
              (rate, depth, float(np.max(np.abs(got - want))), float(np.sqrt(np.mean((got - want) ** 2))), e1))
