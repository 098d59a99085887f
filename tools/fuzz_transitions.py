"""Random additive banks (shared or per-voice ADSRs, negative phases, silent and endless voices) rendered over random launches
around their transitions -- long launches take the segmented path (csrc/osc.hip RENDER_*_SEG) -- against the same frames rendered
as launches of 8192 frames (never segmented: below the eight-frames-per-lane shape).  usage: python tools/fuzz_transitions.py [seed] [cases]"""
import sys

sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
SR = 48000
bad = 0
for case in range(cases):
    nv = int(rng.choice([128, 192, 320, 512, 1024]))
    shared = rng.random() < 0.5
    base = (float(rng.choice([0.0, 0.004, 0.01, 0.05])), float(rng.choice([0.0, 0.02, 0.3])), float(rng.choice([0.0, 0.2, 1.0, 50.0])),
            float(rng.choice([0.3, 0.6, 1.0])), float(rng.choice([0.0, 0.05, 0.4])))
    voices, gains = [], []
    for i in range(nv):
        f = float(np.exp(rng.uniform(np.log(40.0), np.log(4000.0))))
        npart = int(rng.choice([1, 4, 16]))
        harm = [(k, 1.0 / k) for k in range(1, npart + 1)]
        phase = float(rng.uniform(-0.5, 1.0)) if rng.random() < 0.2 else float(rng.uniform(0.0, 1.0))
        osc = G.Harmonics(f, harm, amplitude=float(rng.uniform(0.1, 1.0)) / np.sqrt(nv), phase=phase, samplerate=SR)
        r = rng.random()
        if r < 0.05:
            pass                                              # no envelope at all
        elif shared:
            osc = G.EnvelopeFilter(osc, *base)
        else:
            osc = G.EnvelopeFilter(osc, float(rng.uniform(0, 0.05)), float(rng.uniform(0, 0.3)), float(rng.uniform(0, 1.5)),
                                   float(rng.uniform(0.2, 1.0)), float(rng.uniform(0, 0.4)))
        voices.append(osc)
        gains.append((float(rng.uniform(0, 1)), float(rng.uniform(0, 1))))
    bank = VoiceBank(voices, gains=gains)
    for _ in range(3):
        n = int(rng.choice([16384, 20000, 48000, 65536, 100001]))
        start = int(rng.choice([0, 0, 0, 100, 4096, 40000, 48000, int(1.0 * SR) - 5000, int(rng.integers(0, 3 * SR))]))
        got = bank.render(n, start=start)
        want = np.concatenate([bank.render(min(8192, n - o), start=start + o) for o in range(0, n, 8192)])
        scale = max(1e-3, float(np.max(np.abs(want))))
        err = float(np.max(np.abs(got.astype(np.float64) - want))) / scale
        frac = float(np.mean(got != want))
        if err > 2e-7 or frac > 5e-3:
            bad += 1
            print("case", case, "nv", nv, "shared", shared, "n", n, "start", start, "MISMATCH max", err, "fraction", frac)
print("cases", cases, "launches", 3 * cases, "mismatches", bad)
