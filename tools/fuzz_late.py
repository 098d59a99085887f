"""Random single oscillators -- the five waveforms and Harmonics, plain or under a Sine LFO, some under an envelope with a long sustain --
with parameters from the edges of their ranges (0.01 Hz .. 0.49 sr, phases of either sign and beyond 1, biases, LFOs of 0.003 .. 300 Hz and
depths to 0.95, four sample rates), rendered at a random position 60 .. 600 s into the note against the C oracle's float64 values.
FM cases are held to the contract (1e-6 RMS: the reference's own phase_correction sum carries rounding noise by then), the others to 1e-9.
usage (GPU box): python tools/fuzz_late.py [seed] [cases]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import synth_oracle as O
from oracle import c_oracle as CO
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
N.ensure_init(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
blk, bad, worst_plain, worst_fm = 8192, 0, 0.0, 0.0
for c in range(cases):
    sr = int(rng.choice([22050, 44100, 48000, 96000]))
    kind = str(rng.choice(["Sine", "Sawtooth", "Square", "Triangle", "Pulse", "Harmonics"]))
    f = float(np.exp(rng.uniform(np.log(0.01), np.log(0.49 * sr))))
    amp, ph = float(rng.uniform(0.05, 1.5)), float(rng.uniform(-1.5, 2.5))
    bias = float(rng.choice([0.0, rng.uniform(-0.5, 0.5)]))
    fm = rng.random() < 0.5
    lf, ld, lp, lb = float(np.exp(rng.uniform(np.log(0.003), np.log(300.0)))), float(rng.uniform(0.0, 0.95)), float(rng.uniform(-1, 1)), float(rng.choice([0.0, rng.uniform(-0.1, 0.1)]))
    env = rng.random() < 0.3
    nh = int(rng.integers(1, 17))

    def make(m):
        kw = dict(samplerate=sr)
        if fm:
            kw["fm_lfo"] = m.Sine(lf, ld, phase=lp, bias=lb, samplerate=sr)
        if kind == "Harmonics":
            h = [(k, 1.0 / k) for k in range(1, nh + 1)]
            if f * nh > 0.49 * sr:
                h = h[:1]
            o = m.Harmonics(f, h, amp, phase=ph, bias=bias, **kw)
        elif kind == "Pulse":
            o = m.Pulse(f, amp, phase=ph, bias=bias, pulsewidth=0.37, **kw)
        else:
            o = getattr(m, kind)(f, amp, phase=ph, bias=bias, **kw)
        if env:
            o = m.EnvelopeFilter(o, 0.01, 0.05, 900.0, 0.6, 0.2)
        return o
    first = int(rng.uniform(60, 600) * sr)
    try:
        g, o = make(G), make(O)
        want = CO.render(o, first + blk)[first:]
    except Exception as e:
        print("case", c, "skipped:", repr(e)[:80])
        continue
    got = g.render_f64(blk, start=first)
    scale = max(1.0, float(np.max(np.abs(want))))
    err = float(np.sqrt(np.mean((got - want) ** 2))) / scale
    if fm:
        worst_fm = max(worst_fm, err)
    else:
        worst_plain = max(worst_plain, err)
    if err > (1e-6 if fm else 1e-9):
        bad += 1
        print("MISMATCH case %d: %s f=%.6g sr=%d amp=%.3g ph=%.3g bias=%.3g fm=%s lfo=(%.5g Hz, %.3g, ph %.3g, bias %.3g) env=%s start=%d rms/scale %.3e" %
              (c, kind, f, sr, amp, ph, bias, fm, lf, ld, lp, lb, env, first, err))
print("seed", seed, "cases", cases, "mismatches", bad, "worst rms/scale plain %.3e fm %.3e" % (worst_plain, worst_fm))
