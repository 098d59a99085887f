"""Synthetic workloads of SURVEY.md section 8(d), shared by tests and bench.py (definitions only -- no arithmetic).

``mod`` is either ``synthesizer_amd.oscillators`` (GPU) or ``oracle.synth_oracle`` (CPU): both expose
the same class names and constructor signatures, so one builder yields the two sides of a parity
check from the same seed.
"""
import numpy as np

ADSR = dict(attack=0.01, decay=0.05, sustain=0.5, sustain_level=0.6, release=0.2)


def _voice_params(n, seed):
    rng = np.random.default_rng(seed)
    f = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), n))
    amp = rng.uniform(0.1, 1.0, n) / np.sqrt(n)
    phase = rng.uniform(0.0, 1.0, n)
    pan = rng.uniform(-1.0, 1.0, n)
    gains = [((1.0 - p) / 2.0, (1.0 + p) / 2.0) for p in pan]
    # gains go through float32 on the device; use the float32 values on both sides of a parity check
    gains = [(float(np.float32(l)), float(np.float32(r))) for l, r in gains]
    return rng, f, amp, phase, gains


def additive_voices(mod, n, samplerate, seed=0, partials=16, envelope=True, adsr=None):
    """n additive voices: Harmonics with `partials` partials a_k = 1/k, ADSR envelope."""
    _, f, amp, phase, gains = _voice_params(n, seed)
    harm = [(k, 1.0 / k) for k in range(1, partials + 1)]
    e = dict(ADSR)
    if adsr:
        e.update(adsr)
    voices = []
    for i in range(n):
        osc = mod.Harmonics(float(f[i]), harm, amplitude=float(amp[i]), phase=float(phase[i]), samplerate=samplerate)
        if envelope:
            osc = mod.EnvelopeFilter(osc, e["attack"], e["decay"], e["sustain"], e["sustain_level"], e["release"])
        voices.append(osc)
    return voices, gains


def fm_voices(mod, n, samplerate, seed=0):
    """n FM voices: Sine carrier, Sine modulator f_m ~ U[0.5, 8] Hz, depth U[0, 0.05]."""
    rng, f, amp, phase, gains = _voice_params(n, seed)
    fm = rng.uniform(0.5, 8.0, n)
    depth = rng.uniform(0.0, 0.05, n)
    pm = rng.uniform(0.0, 1.0, n)
    voices = []
    for i in range(n):
        lfo = mod.Sine(float(fm[i]), float(depth[i]), phase=float(pm[i]), samplerate=samplerate)
        voices.append(mod.Sine(float(f[i]), amplitude=float(amp[i]), phase=float(phase[i]), fm_lfo=lfo, samplerate=samplerate))
    return voices, gains


def staggered_notes(mod, slots, samplerate, seed=0, partials=16, period=1.0, notes=11, adsr=None):
    """Notes that do not move in lock-step: `slots` players, each re-triggering its note every `period` seconds; player s starts
    its k-th note at (s / slots + k) * period -- the onsets are spread uniformly over the period -- and every note is a new voice
    (phase and envelope start with it): Harmonics x `partials` under SURVEY 8(d)'s ADSR (attack 0.01, decay 0.05, sustain 0.5 at
    level 0.6, release 0.2: 0.76 s of sound per note), delayed to its onset (DelayFilter).  slots * notes voices, in the order they
    start within a round (all players' k-th notes are neighbours); gains[i] belongs to voices[i]."""
    _, f, amp, phase, gains = _voice_params(slots, seed)
    harm = [(k, 1.0 / k) for k in range(1, partials + 1)]
    e = dict(ADSR)
    if adsr:
        e.update(adsr)
    voices, vgains = [], []
    for k in range(notes):
        for s in range(slots):
            onset = (s / slots + k) * period
            osc = mod.Harmonics(float(f[s]), harm, amplitude=float(amp[s]), phase=float(phase[s]), samplerate=samplerate)
            osc = mod.EnvelopeFilter(osc, e["attack"], e["decay"], e["sustain"], e["sustain_level"], e["release"])
            voices.append(mod.DelayFilter(osc, onset) if onset else osc)
            vgains.append(gains[s])
    return voices, vgains
