#!/usr/bin/env python
"""Randomised parity sweeps against the oracle / live audioop (GPU box, from the repo root): python tools/fuzz.py <name> [seed] [cases]

    osc          Random oscillator graphs -- plain waveforms with random fm_lfo / pwm_lfo sub-oscillators, wrapped in random filters -- built
    tiles        Random banks of NOTES -- every voice with an onset and an ADSR of its own (zero-length phases, envelopes that end, voices without
    late         Random single oscillators -- the five waveforms and Harmonics, plain or under a Sine LFO, some under an envelope with a long sustain --
    mix          One-off sweep: mixer.mix_samples (the saturating fold in voice order) and Sample.mix against the live audioop.add.
    ratecv       One-off sweep: Sample.resample against the live audioop.ratecv over random rates, widths, layouts and lengths.
    period       One-off sweep: 16-bit mono / stereo resampling between rates with a short period (k_resample_period_i16), whole calls and ranges.
    int16        Random additive banks through the integer routes (generate_i16 rows, fused mixdown) against the C oracle's quantised samples: equal
    transitions  Random additive banks (shared or per-voice ADSRs, negative phases, silent and endless voices) rendered over random launches

Exit status 1 when a sweep found a mismatch.  (profiles/r04_fuzz.txt: what they found in round 4.)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def fuzz_osc(argv):
    """Random oscillator graphs -- plain waveforms with random fm_lfo / pwm_lfo sub-oscillators, wrapped in random filters -- built
identically from synthesizer_amd.oscillators and from the oracle, rendered (float64 blocks) and compared.  Also the same graphs as
voices of a VoiceBank against the oracle's bus.  usage: python tools/fuzz_osc.py [seed] [cases]"""
    import sys

    import numpy as np
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank

    seed = int(argv[0]) if len(argv) > 0 else 0
    cases = int(argv[1]) if len(argv) > 1 else 150
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
    return int(bool(locals().get("bad", 0) or locals().get("bad_bank", 0)))



def fuzz_int16(argv):
    """Random additive banks through the INTEGER routes -- generate_i16 rows and the fused mixdown -- against the C oracle's quantised samples and the
live audioop chain over them: strict equality (round 6: the int16 boundary guard).  Harmonic lists: 1/k series of even and odd length (flat and
not flat at their zero crossing), random amplitudes, repeated and negative k, dense lists beyond 16 (Clenshaw), one partial; amplitudes from
quiet to nearly full scale; no envelope, a shared ADSR, ADSRs of their own; windows at the start of the notes (attack / decay: the general
code), seconds and minutes in; row lengths that take every materialisation kernel.  usage: python tools/fuzz.py int16 [seed] [cases]"""
    import audioop
    import numpy as np
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    seed = int(argv[0]) if len(argv) > 0 else 0
    cases = int(argv[1]) if len(argv) > 1 else 40
    rng = np.random.default_rng(seed)
    SR = 48000
    bad = 0
    total = 0
    for case in range(cases):
        nv = int(rng.choice([1, 3, 16, 64, 65, 130]))
        kind = int(rng.integers(0, 6))
        def harm():
            if kind == 0:
                return [(k, 1.0 / k) for k in range(1, 17)]
            if kind == 1:
                return [(k, 1.0 / k) for k in range(1, int(rng.integers(2, 16)))]
            if kind == 2:
                return [(k, float(rng.uniform(-1, 1))) for k in range(1, 17)]
            if kind == 3:
                return [(int(rng.integers(-16, 17)) or 1, float(rng.uniform(-0.5, 0.5))) for _ in range(int(rng.integers(1, 24)))]
            if kind == 4:
                return [(k, 1.0 / k) for k in range(1, int(rng.integers(18, 48)))]
            return [(int(rng.integers(1, 17)), 1.0)]
        lists = [harm() for _ in range(3)]
        env_mode = int(rng.integers(0, 3))                 # 0 none, 1 shared, 2 own
        late = bool(rng.integers(0, 2)) and env_mode != 2
        start = int(rng.choice([5, 30, 300, 1800])) * SR + int(rng.integers(0, 5000)) if late else int(rng.choice([0, 0, 777, 30000]))
        n = int(rng.choice([700, 3000, 9000, 20000, 70001, 140000])) if nv <= 16 else int(rng.choice([700, 9000, 20000]))
        f = np.exp(rng.uniform(np.log(30.0), np.log(6000.0), nv))
        peak = [sum(abs(a) for _k, a in h) for h in lists]
        which = rng.integers(0, 3, nv)
        amp = rng.uniform(0.02, 0.95, nv) / np.array([max(peak[w], 1e-9) for w in which])
        ph = rng.uniform(-0.5, 1.0, nv)
        shared = (0.01, 0.05, 1.0e6 if late else float(rng.uniform(0.05, 0.5)), 0.6, 0.2)
        own = [(float(rng.uniform(0, 0.02)), float(rng.uniform(0, 0.05)), float(rng.uniform(0.01, 0.4)), float(rng.uniform(0.1, 1.0)), float(rng.uniform(0, 0.1))) for _ in range(nv)]

        def make(M):
            out = []
            for i in range(nv):
                o = M.Harmonics(float(f[i]), lists[which[i]], amplitude=float(amp[i]), phase=float(ph[i]), samplerate=SR)
                if env_mode == 1:
                    o = M.EnvelopeFilter(o, *shared)
                elif env_mode == 2:
                    o = M.EnvelopeFilter(o, *own[i])
                out.append(o)
            return out
        ov = make(O)
        if late:
            want = np.stack([CO.quantise(CO.render_window(o, start, n)).astype(np.int16) for o in ov])
        else:
            want = np.stack([CO.quantise(CO.render(o, start + n)[start:]).astype(np.int16) for o in ov])
        bank = VoiceBank(make(G))
        rows, stride = bank.generate_i16_device(n, start)
        got = rows.download(np.int16, nv * stride).reshape(nv, stride)[:, :n]
        nd = int(np.count_nonzero(got != want))
        chain = want[0].tobytes()
        for r in want[1:]:
            chain = audioop.add(chain, r.tobytes(), 2)
        mono = bank.mixdown_i16_device(n, start).download_bytes(n * 2)
        md = int(np.count_nonzero(np.frombuffer(mono, dtype=np.int16) != np.frombuffer(chain, dtype=np.int16)))
        total += want.size
        if nd or md:
            bad += 1
            v, j = (np.argwhere(got != want)[0] if nd else (0, 0))
            print("case %d: %d voices, list kind %d, envelope mode %d, start %d, %d frames: %d row samples differ (first: voice %d frame %d: %d != %d), %d mixdown samples differ"
                  % (case, nv, kind, env_mode, start, n, nd, v, j, got[v, j] if nd else 0, want[v, j] if nd else 0, md), flush=True)
    print("int16 fuzz seed %d: %d cases, %d voice-samples, %d cases with a difference" % (seed, cases, total, bad), flush=True)
    return 1 if bad else 0

def fuzz_tiles(argv):
    """Random banks of NOTES -- every voice with an onset and an ADSR of its own (zero-length phases, envelopes that end, voices without
an envelope), fundamentals up to 12 kHz, Harmonics of 1 .. 16 partials or all the plain kinds -- rendered as streams of launches of 256 .. 48 000 frames, which take the
tile-classified path (csrc/osc_render.hip RENDER_*_TILES: lean / corner / multi-piece / walk pairs, chunk ranges that move with the
block), against the float64 buses of the same frames rendered by sub-banks of 100 voices (never tile-classified: below 128 voices)
added up in float64.
usage: python tools/fuzz_tiles.py [seed] [cases]"""
    import sys

    import numpy as np
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank

    seed = int(argv[0]) if len(argv) > 0 else 0
    cases = int(argv[1]) if len(argv) > 1 else 30
    rng = np.random.default_rng(seed)
    SR = 48000
    bad = launches = 0
    N.ensure_init(0)
    c0 = N.debug_counters()
    for case in range(cases):
        nv = int(rng.choice([130, 200, 520, 1024, 2100]))
        span = float(rng.choice([0.2, 1.0, 3.0]))                       # the notes start within this many seconds
        order = rng.random() < 0.7                                       # in the order they start (chunk ranges) or shuffled
        odd_voices = rng.random() < 0.4                                  # a few voices that are no lean pairs at all
        mixed_kinds = rng.random() < 0.5                                 # Harmonics only, or all the plain kinds (the waveform branch)
        onsets = rng.integers(0, int(span * SR), nv)
        if order:
            onsets = np.sort(onsets)
        if rng.random() < 0.5:
            onsets[: nv // 8] = 0                                        # a block of notes that start with the piece
        voices, gains = [], []
        for i in range(nv):
            f = float(np.exp(rng.uniform(np.log(25.0), np.log(12000.0))))
            npart = int(rng.integers(1, 17))
            harm = [(k, 1.0 / k) for k in range(1, npart + 1)]
            phase = float(rng.uniform(-0.5, 1.0)) if rng.random() < 0.2 else float(rng.uniform(0.0, 1.0))
            amp = float(rng.uniform(0.1, 1.0)) / np.sqrt(nv)
            kind = int(rng.integers(0, 7)) if mixed_kinds else 0
            if kind == 6:
                lfo = G.Sine(float(rng.uniform(0.3, 12.0)), float(rng.uniform(0.0, 0.08)), phase=float(rng.uniform(0.0, 1.0)), samplerate=SR)
                osc = G.Sine(f, amp, phase=phase, fm_lfo=lfo, samplerate=SR)
            elif kind == 0:
                osc = G.Harmonics(f, harm, amplitude=amp, phase=phase, samplerate=SR)
            elif kind == 1:
                osc = G.Sine(f, amp, phase=phase, samplerate=SR)
            elif kind == 2:
                osc = G.Sawtooth(f, amp, phase=phase, samplerate=SR)
            elif kind == 3:
                osc = G.Square(f, amp, phase=phase, samplerate=SR)
            elif kind == 4:
                osc = G.Triangle(f, amp, phase=phase, samplerate=SR)
            else:
                osc = G.Pulse(f, amp, phase=phase, pulsewidth=float(rng.uniform(0.02, 0.98)), samplerate=SR)
            if odd_voices and rng.random() < 0.06:                       # what only the general code can do: general pairs of every tile
                which = int(rng.integers(0, 3))
                osc = (G.Harmonics(f, harm, amplitude=amp, phase=phase, bias=0.01, samplerate=SR) if which == 0
                       else G.Harmonics(min(f, 500.0), [(1, 1.0), (5, 0.3), (40, 0.1)], amplitude=amp, phase=phase, samplerate=SR) if which == 1
                       else G.WhiteNoise(float(rng.uniform(200.0, 8000.0)), amp, samplerate=SR, seed=int(rng.integers(1, 1 << 30))))
            r = rng.random()
            if r < 0.15:
                pass                                                     # no envelope: the onset is a step
            else:
                z = lambda hi: 0.0 if rng.random() < 0.15 else float(rng.uniform(0.0, hi))
                osc = G.EnvelopeFilter(osc, z(0.02), z(0.1), z(0.8), float(rng.uniform(0.2, 1.0)), z(0.3))
            d = int(onsets[i])
            voices.append(G.DelayFilter(osc, d / SR) if d else osc)
            gains.append((float(rng.uniform(0, 1)), float(rng.uniform(0, 1))))
        bank = VoiceBank(voices, gains=gains)
        refs = [VoiceBank(voices[a:a + 100], gains=gains[a:a + 100]) for a in range(0, nv, 100)]
        n = int(rng.choice([256, 1000, 4096, 16384, 20000, 48000]))
        ref64 = N.DeviceBuffer(n * 16)
        first = int(rng.choice([0, 0, 1, 2]))
        ring = [N.DeviceBuffer(n * 8) for _ in range(4)]
        nblocks = int(rng.integers(3, 7)) if n >= 16384 else int(rng.integers(6, 40))
        plan = list(range(first, first + nblocks))
        if rng.random() < 0.3:
            plan += [first, first + 1]                                   # a jump back
        for k in plan:
            bank.render_device(n, k * n, bus_f32=ring[k & 3])
            if k < plan[-1] - 3 or k == plan[-1] or rng.random() < 0.5:  # (not every block is read at once: the pipeline stays up)
                got = ring[k & 3].download(np.float32, n * 2).reshape(n, 2)
                want = np.zeros((n, 2))
                for r_ in refs:
                    r_.render_device(n, k * n, bus_f32=None, bus_f64=ref64)
                    want += ref64.download(np.float64, n * 2).reshape(n, 2)
                scale = max(1e-3, float(np.max(np.abs(want))))
                err = float(np.max(np.abs(got.astype(np.float64) - want))) / scale
                launches += 1
                if err > 3e-7:
                    bad += 1
                    at = int(np.argmax(np.abs(got.astype(np.float64) - want).max(axis=1)))
                    print("case", case, "nv", nv, "n", n, "block", k, "MISMATCH max", err, "at frame", at + k * n)
    c1 = N.debug_counters()
    print("seed", seed, "cases", cases, "launches checked", launches, "mismatches", bad, "tile-classified launches", c1["tiled_launches"] - c0["tiled_launches"],
          "of them on sets resolved ahead", c1["tiled_predicted"] - c0["tiled_predicted"])
    return int(bool(locals().get("bad", 0) or locals().get("bad_bank", 0)))


def fuzz_late(argv):
    """Random single oscillators -- the five waveforms and Harmonics, plain or under a Sine LFO, some under an envelope with a long sustain --
with parameters from the edges of their ranges (0.01 Hz .. 0.49 sr, phases of either sign and beyond 1, biases, LFOs of 0.003 .. 300 Hz and
depths to 0.95, four sample rates), rendered at a random position 60 .. 600 s into the note against the C oracle's float64 values.
FM cases are held to the contract (1e-6 RMS: the reference's own phase_correction sum carries rounding noise by then), the others to 1e-9.
usage (GPU box): python tools/fuzz_late.py [seed] [cases]"""
    import os
    import sys
    import numpy as np
    from oracle import synth_oracle as O
    from oracle import c_oracle as CO
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    N.ensure_init(0)
    seed = int(argv[0]) if len(argv) > 0 else 0
    cases = int(argv[1]) if len(argv) > 1 else 40
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
    return int(bool(locals().get("bad", 0) or locals().get("bad_bank", 0)))


def fuzz_mix(argv):
    """One-off sweep: mixer.mix_samples (the saturating fold in voice order) and Sample.mix against the live audioop.add."""
    import audioop
    import sys

    import numpy as np
    from synthesizer_amd.mixer import mix_samples
    from synthesizer_amd.sample import Sample

    rng = np.random.default_rng(int(argv[0]) if len(argv) > 0 else 0)
    DT = {1: np.int8, 2: np.int16, 4: np.int32}
    bad = 0
    for case in range(200):
        width = int(rng.choice([1, 2, 2, 2, 4]))
        nv = int(rng.choice([1, 2, 3, 7, 8, 9, 31, 64, 65, 200]))
        n = int(rng.choice([1, 2, 7, 8, 9, 63, 511, 512, 513, 4097, 20001]))
        info = np.iinfo(DT[width])
        scale = float(rng.choice([1.0, 0.5, 0.05]))
        chunks = [(rng.integers(info.min, info.max + 1, n, dtype=np.int64) * scale).astype(DT[width]) for _ in range(nv)]
        for c in chunks[:3]:
            c[:min(n, 4)] = np.array([info.max, info.min, info.max, info.min], dtype=DT[width])[:min(n, 4)]
        want = chunks[0].tobytes()
        for c in chunks[1:]:
            want = audioop.add(want, c.tobytes(), width)
        got = mix_samples([Sample.from_raw_frames(c.tobytes(), width, 8000, 1) for c in chunks])
        if bytes(got.view_frame_data()) != want:
            bad += 1
            print("MISMATCH chain", width, nv, n)
        a = Sample.from_raw_frames(chunks[0].tobytes(), width, 8000, 1)
        if nv > 1:
            a.mix(Sample.from_raw_frames(chunks[1].tobytes(), width, 8000, 1))
            if bytes(a.view_frame_data()) != audioop.add(chunks[0].tobytes(), chunks[1].tobytes(), width):
                bad += 1
                print("MISMATCH add", width, n)
    print("cases 200 mismatches", bad)
    return int(bool(locals().get("bad", 0) or locals().get("bad_bank", 0)))


def fuzz_ratecv(argv):
    """One-off sweep: Sample.resample against the live audioop.ratecv over random rates, widths, layouts and lengths."""
    import audioop
    import sys

    import numpy as np
    from synthesizer_amd.sample import Sample

    rng = np.random.default_rng(int(argv[0]) if len(argv) > 0 else 0)
    DT = {1: np.int8, 2: np.int16, 4: np.int32}
    bad = 0
    for case in range(400):
        width = int(rng.choice([1, 2, 2, 2, 4]))
        nch = int(rng.choice([1, 1, 2, 2, 3, 4, 5, 6, 8]))
        if rng.random() < 0.5:
            i, o = (int(x) for x in rng.choice([8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000], 2))
        else:
            i, o = int(rng.integers(1, 200000)), int(rng.integers(1, 200000))
        frames = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 2047, 2049, 5000, 30011, 100003]))
        if frames * o / i > 3e6:
            frames = max(1, int(3e6 * i / o))
        info = np.iinfo(DT[width])
        x = rng.integers(info.min, info.max + 1, frames * nch, dtype=np.int64).astype(DT[width])
        want = audioop.ratecv(x.tobytes(), width, nch, i, o, None)[0]
        got = bytes(Sample.from_raw_frames(x.tobytes(), width, max(i, 2), nch).resample(o).view_frame_data()) if i >= 2 else want
        if got != want:
            bad += 1
            print("MISMATCH", width, nch, i, o, frames, len(got), len(want))
    print("cases 400 mismatches", bad)
    return int(bool(locals().get("bad", 0) or locals().get("bad_bank", 0)))


def fuzz_period(argv):
    """One-off sweep of the short-period resample kernel (k_resample_period_i16): 16-bit mono / stereo, rates whose reduced outrate is <= 2048,
    whole calls and ranges (sh_resample_range at random 16-frame-aligned starts), against the live audioop.ratecv."""
    import audioop
    import ctypes as C

    import numpy as np
    from synthesizer_amd import _native as N

    N.ensure_init(0)
    L = N.lib()
    rng = np.random.default_rng(int(argv[0]) if len(argv) > 0 else 0)
    ncases = int(argv[1]) if len(argv) > 1 else 400
    bad = 0
    eligible = 0
    for case in range(ncases):
        nch = int(rng.choice([1, 2]))
        kind = rng.random()
        if kind < 0.4:
            i, o = (int(x) for x in rng.choice([8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000], 2))
        elif kind < 0.8:
            i, o = int(rng.integers(1, 3000)), int(rng.integers(1, 2049))
        else:
            g = int(rng.integers(1, 60))
            i, o = g * int(rng.integers(1, 700)), g * int(rng.integers(1, 2049))
        frames = int(rng.choice([1, 2, 9, 1000, 3675, 4097, 8192, 20011, 65536, 100003, 300007]))
        if frames * o / i > 4e6:
            frames = max(1, int(4e6 * i / o))
        x = rng.integers(-32768, 32768, frames * nch).astype(np.int16)
        if rng.random() < 0.3:
            x[:] = rng.choice(np.array([-32768, 32767, -1, 0, 1], dtype=np.int16), size=x.size)
        want = audioop.ratecv(x.tobytes(), 2, nch, i, o, None)[0]
        nout = L.sh_resample_out_frames(frames, i, o)
        gg = int(np.gcd(i, o))
        eligible += int(o // gg <= 2048 and i // gg < 65536)
        src = N.DeviceBuffer.from_array(x)
        dst = N.DeviceBuffer(max(nout, 1) * 2 * nch)
        N.check(L.sh_resample(src.handle, frames, nch, 2, 0, i, o, dst.handle, None))
        got = dst.download_bytes(nout * 2 * nch)
        ok = got == want
        if ok and nout > 64:
            for _ in range(3):
                out_first = int(rng.integers(0, nout // 16)) * 16
                out_n = int(rng.integers(1, nout - out_first + 1))
                a, b = C.c_size_t(), C.c_size_t()
                N.check(L.sh_resample_span(frames, i, o, out_first, out_n, C.byref(a), C.byref(b)))
                part = N.DeviceBuffer.from_array(x[a.value * nch:(a.value + b.value) * nch])
                od = N.DeviceBuffer(out_n * 2 * nch)
                N.check(L.sh_resample_range(part.handle, a.value, b.value, nch, 2, 0, i, o, out_first, out_n, od.handle))
                if od.download_bytes(out_n * 2 * nch) != want[out_first * 2 * nch:(out_first + out_n) * 2 * nch]:
                    ok = False
                    print("RANGE MISMATCH", nch, i, o, frames, out_first, out_n)
                part.free(); od.free()
        if not ok:
            bad += 1
            print("MISMATCH", nch, i, o, frames)
        src.free(); dst.free()
    print("cases %d (short period: %d) mismatches %d" % (ncases, eligible, bad))
    return int(bool(bad))


def fuzz_transitions(argv):
    """Random additive banks (shared or per-voice ADSRs, negative phases, silent and endless voices) rendered over random launches
around their transitions -- long launches take the segmented path (csrc/osc.hip RENDER_*_SEG) -- against the same frames rendered
as launches of 8192 frames (never segmented: below the eight-frames-per-lane shape).  usage: python tools/fuzz_transitions.py [seed] [cases]"""
    import sys

    import numpy as np
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank

    seed = int(argv[0]) if len(argv) > 0 else 0
    cases = int(argv[1]) if len(argv) > 1 else 40
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
    return int(bool(locals().get("bad", 0) or locals().get("bad_bank", 0)))


FUZZERS = {"osc": fuzz_osc, "int16": fuzz_int16, "tiles": fuzz_tiles, "late": fuzz_late, "mix": fuzz_mix, "ratecv": fuzz_ratecv, "period": fuzz_period, "transitions": fuzz_transitions}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in FUZZERS:
        sys.exit(__doc__)
    sys.exit(FUZZERS[sys.argv[1]](sys.argv[2:]) or 0)
