"""Random oscillator graphs -- plain waveforms with random fm_lfo / pwm_lfo sub-oscillators, wrapped in random filters -- built
identically from synthesizer_amd.oscillators and from the oracle, rendered (float64 blocks) and compared.  Also the same graphs as
voices of a VoiceBank against the oracle's bus.  usage: python tools/fuzz_osc.py [seed] [cases]"""
import sys

sys.path.insert(0, ".")
import numpy as np
from oracle import synth_oracle as O
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(seed)
SR = 22050


def leaf(depth):
    """-> a recipe (callable: module -> oscillator) for a waveform, possibly frequency- or pulse-width-modulated."""
    kind = rng.choice(["Sine", "Triangle", "Sawtooth", "Square", "Pulse", "Harmonics", "SquareH", "SawtoothH", "Linear", "WhiteNoise"])
    f = float(rng.choice([0.0, 0.25, 3.0, 55.0, 441.0, 997.3, 5000.0, 11025.0]) * rng.uniform(0.9, 1.1))
    amp = float(rng.choice([1.0, 0.5, 0.01, 2.5]))
    phase = float(rng.choice([0.0, 0.25, 0.9, -0.3]))
    bias = float(rng.choice([0.0, 0.0, 0.1, -0.4]))
    fm = leaf(depth + 1) if depth < 2 and rng.random() < 0.4 and kind not in ("Linear", "WhiteNoise") else None
    fm_scale = float(rng.choice([0.001, 0.02, 0.3]))
    if kind == "Linear":
        a, inc = float(rng.uniform(-1, 1)), float(rng.choice([0.0, 1e-4, -3e-4]))
        return lambda M: M.Linear(a, inc, -1.0, 1.0, samplerate=SR)
    if kind == "WhiteNoise":
        fr = float(rng.choice([100.0, 2000.0, SR]))
        sd = int(rng.integers(0, 1 << 30))
        return lambda M: M.WhiteNoise(fr, amp, bias, seed=sd, samplerate=SR)
    harm = [(int(k), float(rng.uniform(0.05, 1.0) / k)) for k in sorted(rng.choice(np.arange(1, 40), int(rng.integers(1, 9)), replace=False))]
    nh = int(rng.integers(1, 12))
    pw = float(rng.choice([0.1, 0.5, 0.93]))
    pwm = leaf(depth + 1) if kind == "Pulse" and depth < 2 and rng.random() < 0.4 else None

    def make(M):
        lfo = None
        if fm is not None:
            lfo = M.AmpModulationFilter(fm(M), M.Linear(fm_scale, samplerate=SR)) if rng_choice_fixed else fm(M)
        if kind == "Harmonics":
            return M.Harmonics(f, harm, amp, phase, bias, fm_lfo=lfo, samplerate=SR)
        if kind in ("SquareH", "SawtoothH"):
            return getattr(M, kind)(f, nh, amp, phase, bias, fm_lfo=lfo, samplerate=SR)
        if kind == "Pulse":
            pl = None
            if pwm is not None:
                pl = M.ClipFilter(M.MixingFilter(M.AmpModulationFilter(pwm(M), M.Linear(0.3, samplerate=SR)), M.Linear(0.5, samplerate=SR)), 0.02, 0.98)
            return M.Pulse(f, amp, phase, bias, pulsewidth=pw, fm_lfo=lfo, pwm_lfo=pl, samplerate=SR)
        return getattr(M, kind)(f, amp, phase, bias, fm_lfo=lfo, samplerate=SR)
    rng_choice_fixed = bool(rng.random() < 0.7)        # scale the modulator down (a raw +-1 modulator means f * (1 + lfo) in 0 .. 2f)
    return make


def tree(depth=0):
    r = rng.random()
    if depth >= 3 or r < 0.35:
        return leaf(depth)
    op = rng.choice(["env", "mix", "ampmod", "clip", "abs", "null", "delay", "echo"])
    a = tree(depth + 1)
    if op == "env":
        args = (float(rng.choice([0.0, 0.01, 0.03])), float(rng.choice([0.0, 0.02])), float(rng.choice([0.0, 0.05, 10.0])),
                float(rng.choice([0.0, 0.6, 1.0])), float(rng.choice([0.0, 0.04])))
        stop = bool(rng.random() < 0.3)
        return lambda M: M.EnvelopeFilter(a(M), *args, stop_at_end=stop)
    if op == "mix":
        b = tree(depth + 1)
        return lambda M: M.MixingFilter(a(M), b(M))
    if op == "ampmod":
        b = leaf(depth + 1)
        return lambda M: M.AmpModulationFilter(a(M), b(M))
    if op == "clip":
        lo, hi = float(rng.choice([-1.0, -0.2])), float(rng.choice([0.3, 1.0]))
        return lambda M: M.ClipFilter(a(M), lo, hi)
    if op == "abs":
        return lambda M: M.AbsFilter(a(M))
    if op == "null":
        return lambda M: M.NullFilter(a(M))
    if op == "delay":
        sec = float(rng.choice([0.0, 0.013, -0.007, 0.2]))
        return lambda M: M.DelayFilter(a(M), sec)
    after, amount, delay, decay = float(rng.choice([0.0, 0.01])), int(rng.integers(1, 4)), float(rng.choice([0.005, 0.02])), float(rng.choice([0.5, 0.9]))
    return lambda M: M.EchoFilter(a(M), after, amount, delay, decay)


bad = 0
worst = 0.0
recipes = []
compared = refused = 0
kinds = {}
for case in range(cases):
    state = rng.bit_generator.state
    recipe = tree()
    n = int(rng.choice([1, 511, 512, 513, 3000, 7001]))
    try:
        want = np.array(recipe(O).take(n), dtype=np.float64)
    except Exception as e:                      # the oracle refuses the graph (e.g. an envelope over a source that ends): so must we
        try:
            recipe(G).render_f64(n)
            print("case", case, "oracle raised", type(e).__name__, "but the GPU path rendered")
            bad += 1
        except Exception as e2:
            refused += 1
            if type(e2) is not type(e):
                kinds[(type(e).__name__, type(e2).__name__)] = kinds.get((type(e).__name__, type(e2).__name__), 0) + 1
        continue
    got = recipe(G).render_f64(len(want) if len(want) < n else n)
    if len(got) != len(want):
        print("case", case, "LENGTH", len(got), len(want))
        bad += 1
        continue
    if len(want) == 0:
        continue
    scale = max(1.0, float(np.max(np.abs(want))))
    err = float(np.sqrt(np.mean((got - want) ** 2))) / scale
    frac = float(np.mean(np.abs(got - want) > 1e-9 * scale))
    worst = max(worst, err if frac < 0.02 else 0.0)
    # discontinuous waveforms under modulation may flip single samples at an edge; everything else is at rounding level
    if err > 1e-6 and frac > 0.02:
        print("case", case, "MISMATCH rms", err, "fraction off", frac, "n", n)
        bad += 1
    recipes.append(recipe)
    compared += 1
print("exception types that differ (oracle, here):", kinds)
print("graphs", cases, "compared", compared, "refused by both", refused, "mismatches", bad, "worst rms (continuous)", worst)

# the same graphs as voices of a bank: bus against the oracle's float sum
bad_bank = 0
for lo in range(0, min(len(recipes), 96), 12):
    part = recipes[lo:lo + 12]
    n = 2500
    try:
        rows = [np.array(r(O).take(n), dtype=np.float64) for r in part]
    except Exception:
        continue
    if any(len(x) < n for x in rows):
        continue
    gains = [(float(rng.uniform(0, 1)), float(rng.uniform(0, 1))) for _ in part]
    bank = VoiceBank([r(G) for r in part], gains=gains)
    got = bank.render(n)
    want = np.zeros((n, 2))
    for x, (gl, gr) in zip(rows, gains):
        want[:, 0] += np.float64(np.float32(gl)) * x
        want[:, 1] += np.float64(np.float32(gr)) * x
    scale = max(1.0, float(np.max(np.abs(want))))
    err = float(np.sqrt(np.mean((got - want) ** 2))) / scale
    frac = float(np.mean(np.abs(got - want) > 1e-5 * scale))
    if err > 2e-6 and frac > 0.02:
        print("bank", lo, "MISMATCH rms", err, "fraction off", frac)
        bad_bank += 1
print("banks", (min(len(recipes), 96) + 11) // 12, "mismatches", bad_bank)
