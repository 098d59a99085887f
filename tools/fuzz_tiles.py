"""Random banks of NOTES -- every voice with an onset and an ADSR of its own (zero-length phases, envelopes that end, voices without
an envelope), fundamentals up to 12 kHz, Harmonics of 1 .. 16 partials or all the plain kinds -- rendered as streams of launches of 256 .. 48 000 frames, which take the
tile-classified path (csrc/osc_render.hip RENDER_*_TILES: lean / corner / multi-piece / walk pairs, chunk ranges that move with the
block), against the float64 buses of the same frames rendered by sub-banks of 100 voices (never tile-classified: below 128 voices)
added up in float64.
usage: python tools/fuzz_tiles.py [seed] [cases]"""
import sys

sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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
