"""The split launch of sh_bank_render (lean Harmonics kernel with the three-term recurrence at eight frames per lane +
general-lists kernel, csrc/osc.hip RENDER_LEAN_HARM_ONLY / RENDER_GENERAL_ONLY) against the combined kernel
(SYNTHHIP_NO_SPLIT=1, four frames per lane by rotations) and against the C oracle -- the same voice tables, blocks
that lie in the attack (every voice general), blocks whose voices cross a phase-table piece end, the steady state
(no general voice: the general kernel leaves at once), and a bank in which only SOME voice groups hold general voices."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from synthesizer_amd.workloads import additive_voices

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SR = 48000

_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_split import banks, BLOCKS
from synthesizer_amd.mixer import VoiceBank
out = {}
for name, (voices, gains) in banks().items():
    bank = VoiceBank(voices, gains=gains)
    for n, start in BLOCKS:
        out["%%s_%%d_%%d" %% (name, n, start)] = bank.render(n, start=start)
np.savez(sys.argv[1], **out)
"""

BLOCKS = [(48000, 0), (48000, 48000), (20000, 5 * 48000), (48000, 200 * 48000), (16384, 3 * 48000 + 11)]


def banks():
    from synthesizer_amd import oscillators as G
    v, g = additive_voices(G, 1024, SR, seed=3, adsr={"sustain": 1.0e6})
    out = {"additive1024": (v, g)}
    # general voices (a bias keeps a Harmonics voice out of the lean loop) in the first and the last voice group only
    v2, g2 = additive_voices(G, 640, SR, seed=4, adsr={"sustain": 1.0e6})
    harm = [(k, 1.0 / k) for k in range(1, 9)]
    for i in (3, 17, 600, 639):
        v2[i] = G.Harmonics(100.0 + i, harm, amplitude=0.01, bias=0.001, samplerate=SR)
    out["mixed640"] = (v2, g2)
    return out


def test_split_launch_equals_combined_kernel_and_oracle(gpu, tmp_path):
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd.mixer import VoiceBank
    # the combined kernel's output, from a process of its own (the switches are read once per process)
    ref = tmp_path / "nosplit.npz"
    env = dict(os.environ, SYNTHHIP_NO_SPLIT="1", SYNTHHIP_VARIANT="444")
    p = subprocess.run([sys.executable, "-c", _CHILD % str(ROOT), str(ref)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    want = np.load(ref)
    for name, (voices, gains) in banks().items():
        bank = VoiceBank(voices, gains=gains)
        for n, start in BLOCKS:
            got = bank.render(n, start=start)
            w = want["%s_%d_%d" % (name, n, start)]
            scale = max(1e-3, float(np.max(np.abs(w))))
            # other frames per lane, recurrence instead of rotations, another order of the general voices: float64 noise
            # before ONE rounding to float32
            assert np.max(np.abs(got.astype(np.float64) - w)) <= 1.3e-7 * scale, (name, n, start)
            assert np.mean(got != w) < 2e-3, (name, n, start, float(np.mean(got != w)))
    # and against the C restatement of the oracle on a window the CPU finishes quickly: 64 of the voices, 1 s
    from synthesizer_amd import oscillators as G
    v, g = additive_voices(G, 1024, SR, seed=3, adsr={"sustain": 1.0e6})
    ov, _ = additive_voices(O, 1024, SR, seed=3, adsr={"sustain": 1.0e6})
    sub = slice(0, 1024, 16)
    n = 48000
    bank = VoiceBank(v[sub] * 4, gains=g[sub] * 4)             # 256 voices: the split path (several voice groups)
    got = bank.render(n, start=0)
    rows = np.stack([CO.render(o, n) for o in ov[sub]])
    bus = np.array(CO.mix_bus(rows, g[sub]), dtype=np.float64) * 4.0
    assert np.sqrt(np.mean((got - bus) ** 2)) <= 1e-6 / 3


_CHILD_SEG = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_split import transition_bank, mixed_kinds_bank, TRANSITION_BLOCKS
from synthesizer_amd.mixer import VoiceBank
voices, gains = transition_bank()
bank = VoiceBank(voices, gains=gains)
out = {}
for n, start in TRANSITION_BLOCKS:
    out["%%d_%%d" %% (n, start)] = bank.render(n, start=start)
buf = bank.render_pcm_device(48000, 0)
out["pcm"] = buf.download(np.int16, 96000)
voices, gains = mixed_kinds_bank()
bank = VoiceBank(voices, gains=gains)
for n, start in TRANSITION_BLOCKS[:7]:
    out["mixed_%%d_%%d" %% (n, start)] = bank.render(n, start=start)
np.savez(sys.argv[1], **out)
"""

# attack and decay in block 0, the first releases in the block that starts at 2 s, silence after; a ragged block, a block
# from the middle of the attack, a long one over everything (cut into launches of 2^22 frames AND into segments)
TRANSITION_BLOCKS = [(48000, 0), (48000, 48000), (48000, 96000), (48000, 144000), (30001, 1234), (20000, 100000), (16384, 0),
                     (200000, 0), (5000000, 0)]


def transition_bank():
    """384 additive voices; most share one ADSR, some leave their sustain earlier or decay longer, some start with a negative
    phase (their phase sum passes zero: piece ends that do not lie an octave apart)."""
    from synthesizer_amd import oscillators as G
    v, g = additive_voices(G, 384, SR, seed=12, adsr={"sustain": 2.0})
    harm = [(k, 1.0 / k) for k in range(1, 17)]
    for i in (7, 130, 383):
        v[i] = G.EnvelopeFilter(G.Harmonics(200.0 + i, harm, amplitude=0.01, phase=-0.4, samplerate=SR), 0.01, 0.05, 2.0, 0.6, 0.2)
    v[50] = G.EnvelopeFilter(G.Harmonics(321.0, harm, amplitude=0.01, samplerate=SR), 0.02, 0.3, 1.2, 0.5, 0.1)      # longer decay, earlier release
    v[51] = G.EnvelopeFilter(G.Harmonics(123.4, harm, amplitude=0.01, samplerate=SR), 0.0, 0.0, 0.7, 1.0, 0.05)      # no attack, released at 0.7 s
    return v, g


def mixed_kinds_bank():
    """256 voices that are all lean candidates, of every lean kind: FM Sine carriers with closed-form Sine LFOs, plain waveforms,
    polynomial Harmonics -- with and without an envelope (the bank takes RENDER_LEAN_ALL_SEG)."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.workloads import fm_voices
    rng = np.random.default_rng(77)
    fv, fg = fm_voices(G, 96, SR, seed=5)
    av, ag = additive_voices(G, 64, SR, seed=6, adsr={"sustain": 2.0})
    voices, gains = list(fv) + list(av), list(fg) + list(ag)
    env = lambda o: G.EnvelopeFilter(o, 0.01, 0.05, 2.0, 0.6, 0.2)
    for k in range(96):
        f, a, ph = float(rng.uniform(60, 3000)), float(rng.uniform(0.005, 0.02)), float(rng.uniform(0, 1))
        kind = (G.Sine, G.Sawtooth, G.Square, G.Triangle)[k % 4]
        o = kind(f, a, phase=ph, samplerate=SR) if k % 8 < 4 else G.Pulse(f, a, phase=ph, pulsewidth=0.3, samplerate=SR)
        voices.append(env(o) if k % 3 == 0 else o)
        gains.append((float(rng.uniform(0, 1)), float(rng.uniform(0, 1))))
    return voices, gains


def test_segmented_transition_launches(gpu, tmp_path):
    """The first block of a note as a segmented launch (RENDER_LEAN_HARM_SEG / RENDER_GENERAL_SEG: one batched prepare, the lean
    kernel over all segments, the general kernel with the first segment's groups split further and k_seg_combine) against the
    same launches unsegmented (SYNTHHIP_NO_SEG=1: every voice through the general code) and against the C oracle."""
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd.mixer import VoiceBank
    ref = tmp_path / "noseg.npz"
    env = dict(os.environ, SYNTHHIP_NO_SEG="1")
    p = subprocess.run([sys.executable, "-c", _CHILD_SEG % str(ROOT), str(ref)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    want = np.load(ref)
    voices, gains = transition_bank()
    bank = VoiceBank(voices, gains=gains)
    for n, start in TRANSITION_BLOCKS:
        got = bank.render(n, start=start)
        w = want["%d_%d" % (n, start)]
        scale = max(1e-3, float(np.max(np.abs(w))))
        assert np.max(np.abs(got.astype(np.float64) - w)) <= 1.3e-7 * scale, (n, start)
        assert np.mean(got != w) < 2e-3, (n, start, float(np.mean(got != w)))
    mv, mg = mixed_kinds_bank()
    mbank = VoiceBank(mv, gains=mg)
    for n, start in TRANSITION_BLOCKS[:7]:
        got = mbank.render(n, start=start)
        w = want["mixed_%d_%d" % (n, start)]
        scale = max(1e-3, float(np.max(np.abs(w))))
        # (naive Square / Pulse / Sawtooth voices: both paths evaluate the exact accumulated phase, edges included)
        assert np.max(np.abs(got.astype(np.float64) - w)) <= 1.3e-7 * scale, ("mixed", n, start)
        # (the FM Sine voices of this bank change pieces of their LFO's table at the first tile boundary behind the piece's end, and
        # tiles count from the launch's -- here: the segment's -- first frame: a few float32 roundings more fall the other way)
        assert np.mean(got != w) < 3e-3, ("mixed", n, start, float(np.mean(got != w)))
    pcm = bank.render_pcm_device(48000, 0).download(np.int16, 96000)
    assert np.max(np.abs(pcm.astype(np.int32) - want["pcm"].astype(np.int32))) <= 1 and np.mean(pcm != want["pcm"]) < 2e-3
    # a run that starts with the segmented launch and goes on: blocks 0 .. 5 pipelined == the blocks one by one
    from synthesizer_amd import _native as N
    bufs = [N.DeviceBuffer(48000 * 8) for _ in range(6)]
    for k in range(6):
        bank.render_device(48000, k * 48000, bus_f32=bufs[k])
    for k in range(6):
        got = bufs[k].download(np.float32, 96000).reshape(48000, 2)
        assert np.array_equal(got, bank.render(48000, start=k * 48000)), k
    # the oracle on the first block and on the block of the releases: 48 of the voices, four times each
    ov, _ = additive_voices(O, 384, SR, seed=12, adsr={"sustain": 2.0})
    keep = [k for k in range(0, 384, 8) if k not in (7, 50, 51, 130, 383)]
    small = VoiceBank([voices[k] for k in keep] * 4, gains=[gains[k] for k in keep] * 4)
    rows = np.stack([CO.render(ov[k], 144000) for k in keep])
    bus = np.array(CO.mix_bus(rows, [gains[k] for k in keep]), dtype=np.float64) * 4.0
    for n, start in ((48000, 0), (48000, 96000)):
        got = small.render(n, start=start)
        assert np.abs(bus[start:start + n]).max() > 1e-3
        assert np.sqrt(np.mean((got - bus[start:start + n]) ** 2)) <= 1e-6 / 3, (n, start)


def test_segmented_materialisation_of_long_rows(gpu):
    """sh_bank_generate over rows longer than one 65 536-frame segment: one record set per segment from ONE prepare launch,
    the lean Harmonics kernel over all of them, the general / silent lists per segment where they can exist."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, n, start = 160, 150_001, 777                     # attack + decay lie in segment 0; the last segment is ragged
    v, g = additive_voices(G, nv, SR, seed=8, adsr={"sustain": 2.0})     # ... and the release (then silence) in segment 1 / 2
    bank = VoiceBank(v, gains=g)
    rows = bank.generate(n, start=start)
    assert rows.shape == (nv, n)
    for i in (0, 63, 64, 159):                           # the oscillator on its own (general code, one frame per lane)
        one = v[i].render(n, start=start)
        assert np.max(np.abs(rows[i] - one)) < 2e-7 and np.mean(rows[i] != one) < 0.01, i
    # silent after the release
    assert not rows[:, int(2.3 * SR):].any()
    # the same rows from short launches (one record set each)
    for lo in (0, 40_000, 100_000):
        part = bank.generate(30_000, start=start + lo)
        assert np.max(np.abs(part - rows[:, lo:lo + 30_000])) < 2e-7
    # mixed down: two-step == fused within float32 summation noise
    two = bank.render_two_step(n, start=start)
    fused = bank.render(n, start=start)
    assert np.sqrt(np.mean((two.astype(np.float64) - fused) ** 2)) <= 5e-7


def test_materialisation_from_the_start_of_the_notes(gpu):
    """sh_bank_generate over rows that start at frame 0: the head (attack, decay, a dozen binades of the phase sum) is cut into
    unequal segments with a record set each (plan_segments, k_prepare_segments_var<false>, the lean and the lists kernel driven
    by the segment table, the general voices of a chunk dealt to several workgroups), the rest follows with equal segments."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, n = 192, 300_000
    v, g = additive_voices(G, nv, SR, seed=21, adsr={"sustain": 3.0})
    harm = [(k, 1.0 / k) for k in range(1, 17)]
    v[3] = G.EnvelopeFilter(G.Harmonics(150.0, harm, amplitude=0.02, phase=-0.3, samplerate=SR), 0.01, 0.05, 3.0, 0.6, 0.2)
    v[70] = G.EnvelopeFilter(G.Harmonics(250.0, harm, amplitude=0.02, samplerate=SR), 0.0, 0.02, 0.1, 0.5, 0.05)       # released and silent early
    bank = VoiceBank(v, gains=g)
    rows = bank.generate(n, start=0)
    assert rows.shape == (nv, n)
    for i in (0, 3, 63, 64, 70, 191):
        one = v[i].render(n, start=0)
        assert np.max(np.abs(rows[i] - one)) < 2e-7 and np.mean(rows[i] != one) < 0.01, i
    assert not rows[70, int(0.2 * SR):].any()
    for start, cnt in ((0, 20_000), (0, 48_000), (100, 70_000), (40_000, 30_000), (65_000, 9000), (0, 8192)):
        part = bank.generate(cnt, start=start)
        assert np.max(np.abs(part - rows[:, start:start + cnt])) < 2e-7, (start, cnt)
    two = bank.render_two_step(n, start=0)
    fused = bank.render(n, start=0)
    assert np.sqrt(np.mean((two.astype(np.float64) - fused) ** 2)) <= 5e-7


def test_recurrence_at_frequencies_where_the_step_angle_is_degenerate(gpu):
    """The lean loop steps 64 samples by x[j] = 2cos(d) x[j-1] - x[j-2], d = 64*dt.  Where sin(d) is tiny -- 750 Hz at 48 kHz is
    d = 2 pi exactly, 375 Hz d = pi, and their neighbours -- 2cos(d) pins the step angle least precisely (error ~1e-16/|sin d|
    per step).  192 voices right at and around such frequencies (several voice groups: the eight-frames-per-lane lean kernel),
    16 partials each, against the C oracle: still orders of magnitude inside the contract."""
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    harm = [(k, 1.0 / k) for k in range(1, 17)]
    freqs = []
    for base in (750.0, 375.0, 1500.0, 93.75, 3000.0, 187.5):
        for eps in (0.0, 1e-9, -1e-9, 1e-6, -1e-6, 1e-4, 1e-3, -1e-2):
            freqs.append(base * (1.0 + eps))
    freqs = (freqs * 4)[:192]
    rng = np.random.default_rng(2)
    phases = rng.uniform(0, 1, len(freqs))
    gains = [(float(np.float32(a)), float(np.float32(b))) for a, b in rng.uniform(0.1, 1.0, (len(freqs), 2))]

    def build(m):
        return [m.Harmonics(f, harm, amplitude=0.01, phase=float(p), samplerate=SR) for f, p in zip(freqs, phases)]
    bank = VoiceBank(build(G), gains=gains)
    ov = build(O)
    for start, n in ((0, 48000), (48000 * 5, 20000)):           # the first second and a block five seconds into the note
        got = bank.render(n, start=start)
        rows = np.stack([CO.render(o, start + n)[start:] for o in ov])
        want = np.array(CO.mix_bus(rows, gains), dtype=np.float64)
        err = got.astype(np.float64) - want
        assert np.sqrt(np.mean(err ** 2)) <= 2e-8, (start, float(np.sqrt(np.mean(err ** 2))))
        assert np.max(np.abs(err)) <= 2e-7, start
