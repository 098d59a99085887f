"""GPU parity: the mixer sum bus over oscillator voices (VoiceBank) vs the oracle.

BASELINE configs[1] (64-voice additive Harmonics + ADSR, 48 kHz stereo) and configs[2] (1024 FM voices)
at durations the pure-Python oracle finishes in seconds; the full-size runs are covered by
size-independent properties (linearity of the bus in the voice set, fused == two-step, block
concatenation == one long render).
"""
import numpy as np
import pytest

from oracle import synth_oracle as O
from tests.helpers import rms
from synthesizer_amd.workloads import additive_voices, fm_voices

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-6
SR = 48000


def _oracle_bus(ovoices, gains, n):
    vs = [v.take(n) for v in ovoices]
    return np.array(O.mix_bus(vs, gains), dtype=np.float64)


def test_config2_additive_64_voices_adsr(gpu):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    gv, gains = additive_voices(G, 64, SR, seed=0)
    ov, _ = additive_voices(O, 64, SR, seed=0)
    n = 6000
    bank = VoiceBank(gv, gains=gains)
    fused = bank.render(n)
    want = _oracle_bus(ov, gains, n)
    assert fused.shape == (n, 2) and fused.dtype == np.float32
    assert rms(fused, want) <= RMS_TOL
    assert np.max(np.abs(fused - want)) < 5e-7
    # two-step (materialise, then HBM-bound mix) agrees with the fused kernel
    two = bank.render_two_step(n)
    assert rms(two, want) <= RMS_TOL
    # per-voice materialisation agrees with the single-oscillator path
    mat = bank.generate(n)
    assert mat.shape == (64, n)
    for i in (0, 17, 63):
        one = gv[i].render(n, start=0)       # another launch shape (frames per lane): the float64 values may differ in
        assert np.max(np.abs(mat[i] - one)) < 1.5e-7 and np.mean(mat[i] != one) < 0.01      # the last bit before rounding
    # frames past the release: voices are silent, bus is exactly zero
    late = bank.render(256, start=SR * 2)
    assert not late.any()


def test_config3_fm_voices(gpu):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, n = 96, 3000
    gv, gains = fm_voices(G, nv, SR, seed=1)
    ov, _ = fm_voices(O, nv, SR, seed=1)
    bank = VoiceBank(gv, gains=gains)
    got = bank.render(n)
    want = _oracle_bus(ov, gains, n)
    assert rms(got, want) <= RMS_TOL


def test_bank_properties_at_full_size(gpu):
    """1024 voices x 1 s: properties that need no oracle."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, n = 1024, 48000
    gv, gains = additive_voices(G, nv, SR, seed=3)
    full = VoiceBank(gv, gains=gains)
    a = VoiceBank(gv[:400], gains=gains[:400])
    b = VoiceBank(gv[400:], gains=gains[400:])
    whole = full.render(n)
    # linearity in the voice set
    parts = a.render(n).astype(np.float64) + b.render(n).astype(np.float64)
    assert rms(whole, parts) <= 2e-7
    # block concatenation == one render (stateless in the sample index)
    pieces = np.concatenate([full.render(12345, 0), full.render(n - 12345, 12345)])
    assert np.array_equal(pieces, whole)
    # fused == two-step within float32 summation noise
    two = full.render_two_step(n)
    assert rms(whole, two) <= 5e-7
    # odd sizes / tails
    odd = full.render(1001, start=77)
    assert np.array_equal(odd, whole[77:77 + 1001])
    # int16 epilogue: truncation toward zero of 32767*bus, saturating
    s = full.render_sample(4096)
    pcm = s.get_frames_numpy()
    want = np.clip(np.trunc(32767.0 * whole[:4096].astype(np.float64)), -32768, 32767).astype(np.int16)
    assert np.array_equal(pcm, want)


def test_mix_bus_kernel_shapes(gpu):
    """sh_mix_bus_f32 over materialised voices: ragged sizes, voice-group split, the direct kernel, gains."""
    from synthesizer_amd.mixer import mix_bus
    rng = np.random.default_rng(5)
    # the last two sizes are long enough (>= 1536 tiles of 256 frames) for the direct, unsplit kernel; one of them ragged
    for nv, nf in ((1, 1), (3, 7), (8, 255), (33, 1000), (64, 4099), (1024, 3000), (257, 1024), (9, 393216), (5, 500003)):
        v = rng.uniform(-1, 1, (nv, nf)).astype(np.float32)
        g = rng.uniform(0, 1, (nv, 2)).astype(np.float32)
        got = mix_bus(v, g)
        want = v.astype(np.float64).T @ g.astype(np.float64)
        assert got.shape == (nf, 2)
        assert np.max(np.abs(got - want)) <= 4e-6 * max(1.0, np.sqrt(nv))


def test_bank_argument_checks(gpu):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    # (a voice modulated by a non-Sine oscillator is a bank voice like any other: tests/test_gpu_modbank.py)
    assert VoiceBank([G.Sine(440, fm_lfo=G.Square(2, 0.1, samplerate=SR), samplerate=SR)]).render(64).shape == (64, 2)
    with pytest.raises(ValueError):
        VoiceBank([])
    with pytest.raises(ValueError):
        VoiceBank([G.Sine(440, samplerate=SR), G.Sine(440, samplerate=44100)])


def _c_oracle_bus(ovoices, gains, n):
    from oracle import c_oracle as CO
    voices = np.stack([CO.render(v, n) for v in ovoices])
    return CO.mix_bus(voices, gains)


def test_configs_at_full_size_vs_c_oracle(gpu):
    """BASELINE configs[1] and configs[2] at their full voice counts and 1 s of audio, and the headline
    1024-voice additive workload over the whole note, against the C restatement of the oracle
    (oracle/oracle.c, itself checked bit for bit against the Python oracle in tests/test_oracle_c.py)."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    # configs[1]: 64-voice additive + ADSR, 48 kHz stereo, 1 s
    gv, gains = additive_voices(G, 64, SR, seed=0)
    ov, _ = additive_voices(O, 64, SR, seed=0)
    got = VoiceBank(gv, gains=gains).render(SR)
    want = _c_oracle_bus(ov, gains, SR)
    assert rms(got, want) <= RMS_TOL and np.max(np.abs(got - want)) < 5e-7
    # configs[2]: 1024 FM voices, 1 s
    gv, gains = fm_voices(G, 1024, SR, seed=1)
    ov, _ = fm_voices(O, 1024, SR, seed=1)
    got = VoiceBank(gv, gains=gains).render(SR)
    want = _c_oracle_bus(ov, gains, SR)
    assert rms(got, want) <= RMS_TOL and np.max(np.abs(got - want)) < 5e-7
    # headline: 1024-voice additive, the attack/decay part and a window across the release end
    gv, gains = additive_voices(G, 1024, SR, seed=0)
    ov, _ = additive_voices(O, 1024, SR, seed=0)
    bank = VoiceBank(gv, gains=gains)
    n = 6000
    want = _c_oracle_bus(ov, gains, 37000)
    assert rms(bank.render(n), want[:n]) <= RMS_TOL
    tail = bank.render(1000, start=36000)                  # release ends at sample 36480
    assert rms(tail, want[36000:37000]) <= RMS_TOL and np.max(np.abs(tail - want[36000:37000])) < 5e-7
    two = bank.render_two_step(n)
    assert rms(two, want[:n]) <= RMS_TOL


def test_edge_sizes_and_positions(gpu):
    """Empty and ragged launches, single voices, very late start positions."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank, mix_bus
    from synthesizer_amd import _native as N
    v = [G.Sine(440.0, 0.5, samplerate=SR), G.Square(1000.0, 0.25, samplerate=SR), G.Harmonics(55.0, [(1, 1.0), (2, 0.5)], 0.25, samplerate=SR)]
    bank = VoiceBank(v, pans=[-1.0, 0.0, 1.0])
    assert bank.render(0).shape == (0, 2)
    assert bank.generate(0).shape == (3, 0)
    one = bank.render(1)
    assert one.shape == (1, 2)
    for n in (1, 63, 64, 65, 127, 128, 129, 255, 257, 1000):
        got = bank.render(n, start=5)
        assert np.array_equal(got, bank.render(2000, start=0)[5:5 + n]), n
        assert np.array_equal(bank.generate(n, start=5), bank.generate(2000, start=0)[:, 5:5 + n]), n
    # pan law: left-only, centre, right-only
    g = bank.gains
    assert g[0] == (1.0, 0.0) and g[1] == (0.5, 0.5) and g[2] == (0.0, 1.0)
    # a position far into the stream (2^40 samples = 265 days at 48 kHz): tables still cover it
    far = 1 << 40
    a = bank.render(512, start=far)
    b = np.concatenate([bank.render(200, start=far), bank.render(312, start=far + 200)])
    assert np.array_equal(a, b) and np.isfinite(a).all() and np.abs(a).max() <= 1.0
    sq = G.Square(1000.0, samplerate=SR).render(96, start=far)
    assert set(np.unique(sq)) <= {-1.0, 1.0} and np.array_equal(sq[:24], sq[48:72])      # 48-sample period, exact edges
    # single-voice bank == the oscillator itself
    solo = VoiceBank([v[0]], gains=[(1.0, 1.0)])
    x = v[0].render(3000, start=100)
    assert np.array_equal(solo.render(3000, start=100)[:, 0], x)
    # mix_bus with one frame / one voice
    assert np.allclose(mix_bus(np.array([[0.5]], dtype=np.float32), [(0.5, 2.0)]), [[0.25, 1.0]])
    small = N.DeviceBuffer(16)
    with pytest.raises(ValueError):
        N.check(N.lib().sh_bank_generate(bank._bank.handle, 0, 100, small.handle, 100))


def test_bank_with_released_voices(gpu):
    """Voices whose note has ended before the block are skipped by the kernel; the others are unaffected."""
    from oracle import c_oracle as CO
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    rng = np.random.default_rng(12)
    n_v = 24
    f = rng.uniform(80, 2000, n_v)
    sustain = rng.uniform(0.0, 0.5, n_v)           # notes end between 0.11 s and 0.61 s
    gains = [(float(np.float32(a)), float(np.float32(b))) for a, b in rng.uniform(0, 1, (n_v, 2))]

    def make(M):
        return [M.EnvelopeFilter(M.Harmonics(float(f[i]), [(1, 1.0), (2, 0.4), (3, 0.2)], 0.2, samplerate=SR),
                                 0.01, 0.05, float(sustain[i]), 0.5, 0.05) for i in range(n_v)]

    bank = VoiceBank(make(G), gains=gains)
    ov = make(O)
    total = int(0.7 * SR)
    want = CO.mix_bus(np.stack([CO.render(v, total) for v in ov]), gains)
    for start in (0, int(0.2 * SR), int(0.4 * SR), int(0.62 * SR)):
        n = min(6000, total - start)
        got = bank.render(n, start=start)
        assert rms(got, want[start:start + n]) <= RMS_TOL, start
        assert np.max(np.abs(got - want[start:start + n])) < 3e-7, start
    assert not bank.render(1000, start=int(0.65 * SR)).any()
    mat = bank.generate(2000, start=int(0.3 * SR))
    for i in range(n_v):
        one = make(G)[i].render(2000, start=int(0.3 * SR))
        assert np.max(np.abs(mat[i] - one)) < 1.5e-7 and np.mean(mat[i] != one) < 0.01


def test_lean_loop_classification_and_parity(gpu):
    """The render kernel sends steady voices (polynomial Harmonics, constant envelope gain, one phase-table piece per
    launch) through its lean loop and the rest through the general code; the split changes per block.  Streaming
    block by block (launch records of block s+1 prepared inside block s's kernel) must agree with the materialise +
    mix path, which knows no lean loop, and with the C oracle -- through attack/decay (nothing is lean), the first
    sustain blocks (many voices cross a phase-table piece end inside the block: the lean loop's two-piece and
    straddling-tile paths) and later ones, for a bank with several voice groups and a ragged last chunk."""
    import ctypes as C
    from oracle import c_oracle as CO
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    n_v, block = 333, 12000
    adsr = {"attack": 0.01, "decay": 0.05, "sustain": 5.0, "sustain_level": 0.6, "release": 0.2}
    gv, gains = additive_voices(G, n_v, SR, seed=3, adsr=adsr)
    ov, _ = additive_voices(O, n_v, SR, seed=3, adsr=adsr)
    bank = VoiceBank(gv, gains=gains)
    check = VoiceBank(additive_voices(G, n_v, SR, seed=3, adsr=adsr)[0], gains=gains)
    want = CO.mix_bus(np.stack([CO.render(v, 4 * block) for v in ov]), gains)
    stats = []
    for s in range(12):
        got = bank.render(block, start=s * block)
        a, b = C.c_uint32(), C.c_uint32()
        N.check(N.lib().sh_bank_launch_stats(bank._bank.handle, C.byref(a), C.byref(b)))
        stats.append((a.value, b.value))
        assert a.value + b.value == n_v                      # nobody is silent yet
        two = check.render_two_step(block, start=s * block)
        assert np.max(np.abs(got - two)) < 2e-7, s
        if s < 4:
            assert rms(got, want[s * block:(s + 1) * block]) <= RMS_TOL, s
            assert np.max(np.abs(got - want[s * block:(s + 1) * block])) < 5e-7, s
    assert stats[0][0] == 0                                   # attack and decay lie inside block 0: no constant gain
    assert stats[-1][0] > 0.8 * n_v                           # later almost everything is lean
    # random access (no speculation) gives the same block
    again = bank.render(block, start=7 * block)
    assert np.array_equal(again, VoiceBank(additive_voices(G, n_v, SR, seed=3, adsr=adsr)[0], gains=gains).render(block, start=7 * block))


def test_lean_loop_fm_and_mixed_bank(gpu):
    """FM Sine voices with a plain Sine LFO take the lean loop too (a second record kind); a bank that mixes them with
    additive voices, enveloped FM, plain Pulse voices, and voices that never go lean (biased Sine, Clenshaw Harmonics) is summed by
    both loops in one launch.  Against the C oracle over blocks that include the time table's piece ends (1 s, 2 s, 4 s
    of accumulated time at 48 kHz: frames 48000, 96000, 192000)."""
    import ctypes as C
    from oracle import c_oracle as CO
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    rng = np.random.default_rng(5)
    nv = 90
    f = rng.uniform(60, 3000, nv)
    fm = rng.uniform(0.5, 8, nv)
    depth = rng.uniform(0, 0.05, nv)
    gains = [(float(np.float32(a)), float(np.float32(b))) for a, b in rng.uniform(0, 1, (nv, 2)) / np.sqrt(nv)]

    def make(M):
        out = []
        for k in range(nv):
            lfo = M.Sine(float(fm[k]), float(depth[k]), phase=0.3, samplerate=SR)
            kind = k % 6
            if kind == 0:
                v = M.Sine(float(f[k]), 0.7, phase=0.1, fm_lfo=lfo, samplerate=SR)                       # lean FM
            elif kind == 1:
                v = M.EnvelopeFilter(M.Sine(float(f[k]), 0.5, fm_lfo=lfo, samplerate=SR), 0.01, 0.02, 9.0, 0.5, 0.1)   # lean FM after decay
            elif kind == 2:
                v = M.EnvelopeFilter(M.Harmonics(float(f[k]), [(j, 1.0 / j) for j in range(1, 9)], 0.5, samplerate=SR),
                                     0.01, 0.02, 9.0, 0.6, 0.1)                                           # lean Harmonics
            elif kind == 3:
                v = M.Sine(float(f[k]), 0.5, bias=0.1, fm_lfo=lfo, samplerate=SR)                        # biased: general
            elif kind == 4:
                v = M.Pulse(float(f[k]), 0.4, pulsewidth=0.3, samplerate=SR)                              # lean (plain waveform)
            else:
                v = M.Harmonics(float(f[k]), [(1, 1.0), (33, 0.2)], 0.5, samplerate=SR)                   # Clenshaw: general
            out.append(v)
        return out

    bank = VoiceBank(make(G), gains=gains)
    ov = make(O)
    block = 9000
    want_all = CO.mix_bus(np.stack([CO.render(v, 190000 + block) for v in ov]), gains)
    for start in (0, 2 * block, 47000, 95000, 190000):
        want = want_all[start:start + block]
        got = bank.render(block, start=start)
        a, b = C.c_uint32(), C.c_uint32()
        N.check(N.lib().sh_bank_launch_stats(bank._bank.handle, C.byref(a), C.byref(b)))
        assert a.value + b.value == nv
        if start:
            # kinds 0, 1, 2 and the Pulse voices (4); an FM voice whose LFO's table piece ends inside the block stays lean (the loop
            # changes pieces at a tile boundary)
            assert a.value == 4 * (nv // 6), (start, a.value)
        else:
            assert a.value == 0       # the first frames: the accumulated time runs through many binades, attack / decay
        assert rms(got, want) <= RMS_TOL, start
        assert np.max(np.abs(got - want)) < 5e-7, start
    # the whole FM config goes lean
    gv, g2 = fm_voices(G, 256, SR, seed=1)
    fmbank = VoiceBank(gv, gains=g2)
    fmbank.render(4000, start=12345)
    a, b = C.c_uint32(), C.c_uint32()
    N.check(N.lib().sh_bank_launch_stats(fmbank._bank.handle, C.byref(a), C.byref(b)))
    assert (a.value, b.value) == (256, 0)


def test_streaming_defers_the_fold_of_partial_buses(gpu):
    """With several voice groups a render leaves partial buses; in a stream of renders the next launch folds them
    (no kernel in between), any other API call folds them first.  Blocks rendered back to back into different buffers,
    into one reused buffer, with a change of block length in between, and a float64 bus, all read back afterwards,
    equal the same blocks rendered one at a time."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, block = 512, 6000
    gv, gains = additive_voices(G, nv, SR, seed=9)
    bank = VoiceBank(gv, gains=gains)
    ref = VoiceBank(additive_voices(G, nv, SR, seed=9)[0], gains=gains)
    single = [ref.render(block, start=s * block) for s in range(6)]          # each followed by a download: folded at once
    bufs = [N.DeviceBuffer(block * 8) for _ in range(6)]
    for s in range(6):
        bank.render_device(block, s * block, bus_f32=bufs[s])                   # nothing between the launches
    got = [b.download(np.float32, block * 2).reshape(block, 2) for b in bufs]
    for s in range(6):
        assert np.array_equal(got[s], single[s]), s
    # one buffer reused: after the stream it holds the last block
    one = N.DeviceBuffer(block * 8)
    for s in range(6):
        bank.render_device(block, s * block, bus_f32=one)
    assert np.array_equal(one.download(np.float32, block * 2).reshape(block, 2), single[5])
    # a different block length in the middle of a stream (other launch shape: folded by a kernel), then float64 buses
    short = N.DeviceBuffer(1000 * 8)
    bank.render_device(block, 0, bus_f32=bufs[0])
    bank.render_device(1000, block, bus_f32=short)
    bank.render_device(block, 2 * block, bus_f32=bufs[2])
    assert np.array_equal(bufs[0].download(np.float32, block * 2).reshape(block, 2), single[0])
    assert np.array_equal(short.download(np.float32, 2000).reshape(1000, 2), ref.render(1000, start=block))
    assert np.array_equal(bufs[2].download(np.float32, block * 2).reshape(block, 2), single[2])
    d64 = [N.DeviceBuffer(block * 16) for _ in range(3)]
    for s in range(3):
        bank.render_device(block, s * block, bus_f32=None, bus_f64=d64[s])
    for s in range(3):
        v = d64[s].download(np.float64, block * 2).reshape(block, 2)
        assert np.array_equal(v.astype(np.float32), single[s]), s


def test_fuzz_fused_vs_two_step_vs_oracle(gpu):
    """Random mixed banks, random starts and block lengths (piece ends of the phase tables inside the block, ragged tiles,
    ragged chunks, notes that end, sample indices beyond 2^32): the fused render (lean + general loops, deferred fold),
    the materialise + mix path (lists incl. silent rows) and the C oracle agree."""
    from oracle import c_oracle as CO
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    import os
    rng = np.random.default_rng(int(os.environ.get("SYNTHHIP_FUZZ_SEED", "2024")))
    for case in range(14):
        nv = int(rng.choice([1, 7, 63, 64, 65, 130, 257, 400]))
        f = rng.uniform(40, 4000, nv)
        kinds = rng.integers(0, 10, nv)
        sustain = rng.uniform(0.02, 1.5, nv)
        gains = [(float(np.float32(a)), float(np.float32(b))) for a, b in rng.uniform(0, 1, (nv, 2)) / np.sqrt(nv)]
        seeds = rng.uniform(0, 1, (nv, 3))

        def make(M):
            out = []
            for k in range(nv):
                lfo = M.Sine(0.5 + 7 * float(seeds[k, 0]), 0.04 * float(seeds[k, 1]), phase=float(seeds[k, 2]), samplerate=SR)
                kd = int(kinds[k])
                if kd == 0:
                    v = M.Harmonics(float(f[k]), [(j, 1.0 / j) for j in range(1, 13)], 0.5, phase=float(seeds[k, 2]), samplerate=SR)
                elif kd == 1:
                    v = M.EnvelopeFilter(M.Harmonics(float(f[k]), [(1, 1.0), (3, 0.3), (5, 0.2)], 0.6, samplerate=SR),
                                         0.01, 0.03, float(sustain[k]), 0.55, 0.05)
                elif kd == 2:
                    v = M.Sine(float(f[k]), 0.7, fm_lfo=lfo, samplerate=SR)
                elif kd == 3:
                    v = M.EnvelopeFilter(M.Sine(float(f[k]), 0.6, fm_lfo=lfo, samplerate=SR), 0.005, 0.02, float(sustain[k]), 0.4, 0.1)
                elif kd == 4:
                    v = M.Sawtooth(float(f[k]), 0.4, phase=float(seeds[k, 0]), samplerate=SR)
                elif kd == 5:
                    v = M.Harmonics(float(f[k]), [(1, 1.0), (2, 0.5)], 0.5, bias=0.05, samplerate=SR)
                elif kd == 6:
                    v = M.Square(float(f[k]), 0.3, samplerate=SR)
                elif kd == 7:
                    v = M.EnvelopeFilter(M.Triangle(float(f[k]), 0.4, phase=float(seeds[k, 1]), samplerate=SR),
                                         0.0, 0.01, float(sustain[k]), 0.8, 0.02)
                elif kd == 8:
                    v = M.Pulse(float(f[k]), 0.35, pulsewidth=0.05 + 0.9 * float(seeds[k, 0]), samplerate=SR)
                else:
                    v = M.Sine(float(f[k]), 0.5, phase=float(seeds[k, 2]), samplerate=SR)
                out.append(v)
            return out

        bank = VoiceBank(make(G), gains=gains)
        total = int(rng.choice([3000, 9000, 20000, 40000]))
        ov = make(O)
        want = CO.mix_bus(np.stack([CO.render(v, total) for v in ov]), gains)
        pos = 0
        while pos < total:
            n = int(min(total - pos, rng.choice([1, 63, 256, 257, 1000, 4096, 12345])))
            got = bank.render(n, start=pos)
            assert rms(got, want[pos:pos + n]) <= RMS_TOL, (case, pos, n)
            assert np.max(np.abs(got - want[pos:pos + n])) < 5e-7, (case, pos, n)
            two = bank.render_two_step(n, start=pos)
            assert np.max(np.abs(got - two)) < 1e-6, (case, pos, n)      # two-step: float32 voices, float32 accumulation
            pos += n
        # far into the stream (no oracle there: the sequential restatement cannot get to 2^33): fused == two-step
        far = (1 << 33) + int(rng.integers(0, 1 << 20))
        for n in (777, 9000):
            a = bank.render(n, start=far)
            b = bank.render_two_step(n, start=far)
            assert np.max(np.abs(a - b)) < 1e-6, (case, far, n)
            far += n


def test_buffer_pool_and_free_with_a_pending_fold(gpu):
    """Device buffers come from a pool (freed ones are handed out again without a device synchronisation); a buffer that
    is freed while a render still owes it the fold of its partial buses gets the fold first."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    gv, gains = additive_voices(G, 512, SR, seed=4)
    bank = VoiceBank(gv, gains=gains)
    ref = bank.render(6000, start=0)
    for _ in range(3):
        tmp = N.DeviceBuffer(6000 * 8)
        bank.render_device(6000, 0, bus_f32=tmp)
        tmp.free()                                    # nothing was called in between: the fold is still pending
        again = N.DeviceBuffer(6000 * 8)              # very likely the same memory, handed out again
        bank.render_device(6000, 0, bus_f32=again)
        assert np.array_equal(again.download(np.float32, 12000).reshape(6000, 2), ref)
        again.free()
    # sizes are the requested ones whatever the size class; contents of a recycled buffer are the new owner's
    for nbytes in (1, 255, 257, 4097, 1000003):
        a = N.DeviceBuffer(nbytes)
        assert a.nbytes == nbytes and N.lib().sh_buf_size(a.handle) == nbytes
        a.upload(np.full(nbytes, 7, dtype=np.uint8))
        a.free()
        b = N.DeviceBuffer(nbytes)
        b.zero()
        assert not b.download(np.uint8, nbytes).any()
        b.free()


def test_two_stream_render_pipeline(gpu):
    """Back-to-back renders alternate between two streams; launch n takes its records from launch n-2 and folds launch
    n-2's partial buses.  Long runs, runs that break (a jump in the start position, another bank in between, another
    block length, an unrelated API call, a bus freed while its fold is outstanding, a generate call), notes that end
    inside the run -- every block must equal the same block rendered alone (each followed by a download)."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    rng = np.random.default_rng(31)
    nv, block = 640, 5000
    mk = lambda seed: additive_voices(G, nv, SR, seed=seed, adsr={"sustain": 0.3})      # notes end after ~0.6 s = block 6
    (va, ga), (vb, gb) = mk(1), mk(2)
    bank_a, bank_b = VoiceBank(va, gains=ga), VoiceBank(vb, gains=gb)
    ref_a, ref_b = VoiceBank(mk(1)[0], gains=ga), VoiceBank(mk(2)[0], gains=gb)
    nblocks = 23
    alone_a = [ref_a.render(block, start=s * block) for s in range(nblocks)]
    alone_b = [ref_b.render(block, start=s * block) for s in range(nblocks)]

    def read(buf, n=block):
        return buf.download(np.float32, n * 2).reshape(n, 2)

    # 1. one long run, rotating buses
    bufs = [N.DeviceBuffer(block * 8) for _ in range(nblocks)]
    for s in range(nblocks):
        bank_a.render_device(block, s * block, bus_f32=bufs[s])
    for s in range(nblocks):
        assert np.array_equal(read(bufs[s]), alone_a[s]), s
    # 2. runs broken in every way, checked against the block-at-a-time results
    plan = []
    s = 0
    while len(plan) < 60:
        kind = int(rng.integers(0, 8))
        run = int(rng.integers(1, 6))
        s = int(rng.integers(0, nblocks - run))
        plan += [("a", s + k) for k in range(run)]
        if kind == 0:
            plan.append(("b", int(rng.integers(0, nblocks))))
        elif kind == 1:
            plan.append(("sync", 0))
        elif kind == 2:
            plan.append(("short", int(rng.integers(0, nblocks - 1))))
        elif kind == 3:
            plan.append(("free", int(rng.integers(0, nblocks))))
        elif kind == 4:
            plan.append(("generate", int(rng.integers(0, nblocks))))
        elif kind == 5:
            plan += [("b", k) for k in range(3)]
    outs = []
    for what, s in plan:
        if what in ("a", "b"):
            buf = N.DeviceBuffer(block * 8)
            (bank_a if what == "a" else bank_b).render_device(block, s * block, bus_f32=buf)
            outs.append((what, s, buf))
        elif what == "sync":
            N.sync()
        elif what == "short":
            buf = N.DeviceBuffer(777 * 8)
            bank_a.render_device(777, s * block, bus_f32=buf)
            outs.append(("short", s, buf))
        elif what == "free":
            buf = N.DeviceBuffer(block * 8)
            bank_a.render_device(block, s * block, bus_f32=buf)
            buf.free()                                           # its fold is still outstanding: must be folded (or dropped) safely
        elif what == "generate":
            v = N.DeviceBuffer(nv * 1024 * 4)
            bank_a.generate_device(1024, s * block, out=v)
            v.free()
    for what, s, buf in outs:
        if what == "short":
            assert np.array_equal(read(buf, 777), ref_a.render(777, start=s * block)), (what, s)
        else:
            assert np.array_equal(read(buf), (alone_a if what == "a" else alone_b)[s]), (what, s)
    # 3. one bus for a whole run: it holds the last block afterwards; the float64 bus likewise
    one, one64 = N.DeviceBuffer(block * 8), N.DeviceBuffer(block * 16)
    for s in range(9):
        bank_b.render_device(block, s * block, bus_f32=one, bus_f64=one64)
    assert np.array_equal(read(one), alone_b[8])
    assert np.array_equal(one64.download(np.float64, block * 2).reshape(block, 2).astype(np.float32), alone_b[8])


def test_render_run_with_another_thread_calling_in(gpu):
    """A run of pipelined renders in one thread while another thread keeps calling unrelated entry points: each of those
    calls ends the run wherever it happens to land (joins the streams, folds what is outstanding) and the next render
    starts a new one.  Both sides' results must be exact whatever the interleaving."""
    import audioop
    import threading
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.sample import Sample
    nv, block, nblocks = 512, 4000, 40
    voices, gains = additive_voices(G, nv, SR, seed=4)
    bank = VoiceBank(voices, gains=gains)
    ref = VoiceBank(additive_voices(G, nv, SR, seed=4)[0], gains=gains)
    alone = [ref.render(block, start=s * block) for s in range(nblocks)]
    bufs = [N.DeviceBuffer(block * 8) for _ in range(nblocks)]
    errors, done = [], threading.Event()

    def render():
        try:
            for rep in range(3):
                for s in range(nblocks):
                    bank.render_device(block, s * block, bus_f32=bufs[s])
        except Exception as e:                                   # pragma: no cover
            errors.append(e)
        done.set()

    rng = np.random.default_rng(8)
    x = rng.integers(-20000, 20000, 2 * 20000).astype(np.int16)
    y = rng.integers(-20000, 20000, 2 * 20000).astype(np.int16)
    want_add, want_peak = audioop.add(x.tobytes(), y.tobytes(), 2), audioop.max(x.tobytes(), 2)
    t = threading.Thread(target=render)
    t.start()
    rounds = 0
    while not done.is_set() or rounds < 3:
        a = Sample.from_raw_frames(x.tobytes(), 2, 8000, 2)
        assert a.peak() == want_peak
        assert bytes(a.mix(Sample.from_raw_frames(y.tobytes(), 2, 8000, 2)).view_frame_data()) == want_add
        rounds += 1
    t.join()
    assert not errors
    for s in range(nblocks):
        got = bufs[s].download(np.float32, block * 2).reshape(block, 2)
        assert np.array_equal(got, alone[s]), s


def test_pcm_blocks_straight_from_the_fold(gpu):
    """sh_bank_render_pcm: int16 stereo PCM produced by the fold of the partial buses (in the render kernel two launches
    on, or by the combine kernel when the run ends) equals quantising the float32 bus of the same block with
    sh_quantize_clip_f32 -- for a run of blocks into a ring of buffers, for a loud bank that saturates, for a small bank
    (one voice group: the kernel's own epilogue writes the PCM), and mixed with float renders in one run."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    L = N.lib()
    block = 4800
    for nv, loud in ((640, 1.0), (640, 40.0), (5, 1.0)):
        voices, gains = additive_voices(G, nv, SR, seed=nv)
        gains = [(gl * loud, gr * loud) for gl, gr in gains]
        bank = VoiceBank(voices, gains=gains)
        ref = VoiceBank(additive_voices(G, nv, SR, seed=nv)[0], gains=gains)
        want = []
        for s in range(9):
            bus = ref.render_device(block, s * block)
            pcm = N.DeviceBuffer(block * 4)
            N.check(L.sh_quantize_clip_f32(bus.handle, block * 2, 32767.0, pcm.handle))
            want.append(pcm.download_bytes(block * 4))
        if loud > 1:
            assert any(np.abs(np.frombuffer(w, np.int16)).max() == 32767 for w in want)      # it does saturate
        ring = [N.DeviceBuffer(block * 4) for _ in range(9)]
        for s in range(9):
            bank.render_pcm_device(block, s * block, pcm=ring[s])                            # nothing in between: one run
        for s in range(9):
            assert ring[s].download_bytes(block * 4) == want[s], (nv, loud, s)
        # float and PCM outputs alternating within one run, other scale
        f32 = N.DeviceBuffer(block * 8)
        p16 = N.DeviceBuffer(block * 4)
        bank.render_device(block, 0, bus_f32=f32)
        bank.render_pcm_device(block, block, scale=1000.0, pcm=p16)
        bank.render_device(block, 2 * block, bus_f32=f32)
        half = N.DeviceBuffer(block * 4)
        bus = ref.render_device(block, block)                  # (keep the buffer object alive across the call)
        N.check(L.sh_quantize_clip_f32(bus.handle, block * 2, 1000.0, half.handle))
        assert p16.download_bytes(block * 4) == half.download_bytes(block * 4)
        assert np.array_equal(f32.download(np.float32, block * 2), ref.render(block, 2 * block).reshape(-1))
        smp = bank.render_sample(block, 3 * block)
        assert bytes(smp.view_frame_data()) == want[3] and smp.nchannels == 2 and smp.samplewidth == 2
    with pytest.raises(ValueError):
        bank.render_pcm_device(block, 0, pcm=N.DeviceBuffer(16))


def test_banks_rendering_turn_by_turn_keep_their_pipelines(gpu):
    """Each bank owns its run of pipelined renders, its pending folds and its partial-bus ring: three banks rendering turn by
    turn (a, b, c, a, b, c, ...) stay pipelined -- and every block equals the block rendered alone.  Also: two banks writing
    the SAME bus buffer in turn (the later write must win), a bank destroyed while the others' runs go on, and
    sh_bank_launch_stats in the middle."""
    import ctypes
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, block, nblocks = 384, 6000, 11
    mk = lambda seed: additive_voices(G, nv, SR, seed=seed, adsr={"sustain": 100.0})
    banks = [VoiceBank(mk(s)[0], gains=mk(s)[1]) for s in (41, 42, 43)]
    alone = [[VoiceBank(mk(s)[0], gains=mk(s)[1]).render(block, start=k * block) for k in range(nblocks)] for s in (41, 42, 43)]
    read = lambda buf: buf.download(np.float32, block * 2).reshape(block, 2)
    bufs = [[N.DeviceBuffer(block * 8) for _ in range(nblocks)] for _ in banks]
    for k in range(nblocks):
        for i, bank in enumerate(banks):
            bank.render_device(block, k * block, bus_f32=bufs[i][k])
        if k == 5:
            nf, ng = ctypes.c_uint32(), ctypes.c_uint32()
            N.check(N.lib().sh_bank_launch_stats(banks[1]._bank.handle, ctypes.byref(nf), ctypes.byref(ng)))   # ends the runs
            assert nf.value + ng.value == nv
    for i in range(3):
        for k in range(nblocks):
            assert np.array_equal(read(bufs[i][k]), alone[i][k]), (i, k)
    # one bus buffer written by two banks in turn: what it holds in the end is the last render into it
    shared = N.DeviceBuffer(block * 8)
    for k in range(4):
        banks[0].render_device(block, k * block, bus_f32=shared)
        banks[1].render_device(block, k * block, bus_f32=shared)
    assert np.array_equal(read(shared), alone[1][3])
    for k in range(4):
        banks[1].render_device(block, k * block, bus_f32=shared)
        banks[0].render_device(block, k * block, bus_f32=shared)
    assert np.array_equal(read(shared), alone[0][3])
    # a bank goes away in the middle of everybody's runs
    outs = [N.DeviceBuffer(block * 8) for _ in range(6)]
    for k in range(3):
        banks[0].render_device(block, k * block, bus_f32=outs[k])
        banks[2].render_device(block, k * block, bus_f32=outs[3 + k])
    banks[2]._bank.free()
    banks[0].render_device(block, 3 * block, bus_f32=shared)
    assert np.array_equal(read(shared), alone[0][3])
    for k in range(3):
        assert np.array_equal(read(outs[k]), alone[0][k]) and np.array_equal(read(outs[3 + k]), alone[2][k]), k


def test_fm_only_lean_kernel_in_a_steady_window(gpu):
    """Round 4: a bank whose lean candidates are ALL FM Sine voices renders its steady blocks through the lean kernel instantiated for
    that kind alone (k_render_lean<.., LEAN_K_FM, false>): sines by sin_tab_n, the accumulated time of a one-piece tile by one addition
    per frame, one accumulation behind the four forms (straddle x LFO bias).  256 voices, a third of the LFOs with a bias, a window of
    two 16 384-frame blocks that holds the time table's piece end at n = 48 000 (t = 1.0: the tile there straddles it) against the C
    oracle; and the same blocks as a pipelined run and through the general code (SYNTHHIP_VARIANT shapes are not needed: 444 split)."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, blk, first = 256, 16384, 40000
    rng = np.random.default_rng(11)
    f = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), nv))
    amp = rng.uniform(0.1, 1.0, nv) / np.sqrt(nv)
    ph = rng.uniform(0.0, 1.0, nv)
    fm = rng.uniform(0.5, 8.0, nv)
    depth = rng.uniform(0.0, 0.05, nv)
    pm = rng.uniform(0.0, 1.0, nv)
    gains = [(float(np.float32(g)), float(np.float32(1.0 - g))) for g in rng.uniform(0.0, 1.0, nv)]

    def build(m):
        out = []
        for i in range(nv):
            lfo = m.Sine(float(fm[i]), float(depth[i]), phase=float(pm[i]), bias=(0.01 if i % 3 == 0 else 0.0), samplerate=SR)
            out.append(m.Sine(float(f[i]), amplitude=float(amp[i]), phase=float(ph[i]), fm_lfo=lfo, samplerate=SR))
        return out
    gv, ov = build(G), build(O)
    want = _c_oracle_bus(ov, gains, first + 2 * blk)
    bank = VoiceBank(gv, gains=gains)
    for k in range(2):
        got = bank.render(blk, start=first + k * blk)
        w = want[first + k * blk: first + (k + 1) * blk]
        assert rms(got, w) <= RMS_TOL and np.max(np.abs(got - w)) < 5e-7, k
        assert np.abs(w).max() > 0.05
    import ctypes as C
    a, b = C.c_uint32(), C.c_uint32()
    N.check(N.lib().sh_bank_launch_stats(bank._bank.handle, C.byref(a), C.byref(b)))
    assert (a.value, b.value) == (nv, 0)                    # every voice took the lean lists in the steady window
    # the pipelined run of the same blocks (records two launches ahead, folds taken over): bit-identical with the single renders
    ring = [N.DeviceBuffer(blk * 8) for _ in range(4)]
    singles = [bank.render(blk, start=first + k * blk) for k in range(4)]
    for k in range(4):
        bank.render_device(blk, first + k * blk, bus_f32=ring[k])
    N.sync()
    for k in range(4):
        assert np.array_equal(ring[k].download(np.float32, blk * 2).reshape(blk, 2), singles[k]), k


def test_lean_lists_in_runs_by_kind(gpu):
    """Round 4: every chunk's lean list is written in three runs (polynomial Harmonics, FM Sine, the rest) and the all-kinds lean kernel
    walks each run with a loop of its own.  Banks whose chunks hold every composition -- one kind only, runs of one entry, runs that a
    wavefront's stride of four skips entirely, a last chunk that is not full -- against the C oracle in a steady window, and the float64
    bus of the mixed bank against the sum of the buses of its three single-kind sub-banks (each rendered by another instantiation of
    the kernel)."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, blk, first = 333, 6000, 70000
    rng = np.random.default_rng(21)
    f = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), nv))
    amp = rng.uniform(0.1, 1.0, nv) / np.sqrt(nv)
    ph = rng.uniform(0.0, 1.0, nv)
    fm = rng.uniform(0.5, 8.0, nv)
    depth = rng.uniform(0.0, 0.05, nv)
    gains = [(float(np.float32(g)), float(np.float32(1.0 - g))) for g in rng.uniform(0.0, 1.0, nv)]
    # chunk 0: Harmonics only; chunk 1: FM Sine only; chunk 2: waveforms only; chunk 3: one Harmonics, one FM Sine, the rest Sawtooth;
    # chunk 4: random; chunk 5 (13 voices): random
    kinds = np.concatenate([np.zeros(64, int), np.ones(64, int), 2 + rng.integers(0, 5, 64), [0, 1] + [2] * 62,
                            rng.integers(0, 7, 64), rng.integers(0, 7, nv - 320)])

    def build(m, only=None):
        out = []
        for i in range(nv):
            k = int(kinds[i])
            cls = 0 if k == 0 else 1 if k == 1 else 2
            if only is not None and cls != only:
                continue
            fr, a, p = float(f[i]), float(amp[i]), float(ph[i])
            if k == 0:
                v = m.Harmonics(fr, [(j, 1.0 / j) for j in range(1, 1 + 1 + i % 16)], a, phase=p, samplerate=SR)
            elif k == 1:
                v = m.Sine(fr, a, phase=p, fm_lfo=m.Sine(float(fm[i]), float(depth[i]), samplerate=SR), samplerate=SR)
            elif k == 2:
                v = m.Sawtooth(fr, a, phase=p, samplerate=SR)
            elif k == 3:
                v = m.Square(fr, a, phase=p, samplerate=SR)
            elif k == 4:
                v = m.Triangle(fr, a, phase=p, samplerate=SR)
            elif k == 5:
                v = m.Pulse(fr, a, phase=p, pulsewidth=0.3, samplerate=SR)
            else:
                v = m.Sine(fr, a, phase=p, samplerate=SR)
            out.append(v)
        return out
    cls_of = [0 if k == 0 else 1 if k == 1 else 2 for k in kinds]
    bank = VoiceBank(build(G), gains=gains)
    want = _c_oracle_bus(build(O), gains, first + 2 * blk)
    for k in range(2):
        got = bank.render(blk, start=first + k * blk)
        w = want[first + k * blk: first + (k + 1) * blk]
        assert rms(got, w) <= RMS_TOL and np.max(np.abs(got - w)) < 5e-7, k
    import ctypes as C
    a, b = C.c_uint32(), C.c_uint32()
    N.check(N.lib().sh_bank_launch_stats(bank._bank.handle, C.byref(a), C.byref(b)))
    assert (a.value, b.value) == (nv, 0)                    # every voice took the lean lists
    whole = N.DeviceBuffer(blk * 16)
    bank.render_device(blk, first, bus_f64=whole)
    total = np.zeros(blk * 2)
    for c in range(3):
        sub = VoiceBank(build(G, only=c), gains=[g for g, cc in zip(gains, cls_of) if cc == c])
        part = N.DeviceBuffer(blk * 16)
        sub.render_device(blk, first, bus_f64=part)
        total += part.download(np.float64, blk * 2)
        part.free()
    got64 = whole.download(np.float64, blk * 2)
    whole.free()
    assert np.max(np.abs(got64 - total)) < 1e-12


def test_fm_follows_the_lfos_accumulated_phase_late_in_a_note(gpu):
    """Round 4: the LFO of an FM voice is an oscillator of its own in the reference -- its phase the accumulated t += d, which drifts from
    a + n d by half an ulp of t per sample -- and the carrier's angle is f_inc times the running SUM of it: summed along the ideal line
    (rounds 1-3) a 3.5 kHz carrier was 1e-5 off the oracle 30 s into a note and 1e-3 after 300 s.  The sum now follows the LFO's own
    phase table (oscillators.LfoTable).  100 s into the note: a single oscillator (general code; the window holds the frame at which the
    LFO's phase crosses 4096 rad for one of the rates) and a 160-voice bank (lean lists, the LFO's piece changing inside the launch for
    some voices) against the C oracle; what is left is the rounding noise of the reference's own running sums."""
    from oracle import c_oracle as CO
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    blk = 16384
    for secs, f, rate, depth, tol in ((100, 3520.0, 5.0, 0.5, 2e-7), (130, 880.0, 5.0133, 0.05, 5e-8),
                                      # (a very slow, deep LFO: the reference's own phase_correction sum -- 1.4e6 additions of slowly varying terms to a
                                      #  number of ~1e5, their roundings correlated -- is 4e-7 off the exact sum by now: inside the contract, not below)
                                      (30, 440.0, 0.01, 0.5, 1e-6)):
        first = secs * SR
        g = G.Sine(f, 1.0, phase=0.2, fm_lfo=G.Sine(rate, depth, phase=0.3, samplerate=SR), samplerate=SR)
        o = O.Sine(f, 1.0, phase=0.2, fm_lfo=O.Sine(rate, depth, phase=0.3, samplerate=SR), samplerate=SR)
        w = CO.render(o, first + blk)[first:]
        got = g.render_f64(blk, start=first)
        assert rms(got, w) <= tol, (secs, f, rate, float(rms(got, w)))
    nv, first = 160, 40 * SR
    rng = np.random.default_rng(3)
    fr = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), nv))
    rate = rng.uniform(4.0, 17.0, nv)          # 40 s in: phases of 1000 .. 4300 rad -- some cross 1024, 2048 or 4096 in the window
    gains = [(float(np.float32(a)), float(np.float32(1.0 - a))) for a in rng.uniform(0.0, 1.0, nv)]

    def build(m):
        return [m.Sine(float(fr[i]), 1.0 / np.sqrt(nv), phase=float(i % 7) / 7.0,
                       fm_lfo=m.Sine(float(rate[i]), 0.3, phase=0.1, bias=(0.01 if i % 4 == 0 else 0.0), samplerate=SR), samplerate=SR) for i in range(nv)]
    gv, ov = build(G), build(O)
    n = 3 * blk
    rows = np.stack([CO.render(v, first + n)[first:] for v in ov])
    want = CO.mix_bus(rows, gains)
    crossing = 0
    for i in range(nv):
        d = 2 * np.pi * rate[i] / SR
        t0, t1 = 0.1 * 2 * np.pi + first * d, 0.1 * 2 * np.pi + (first + n) * d
        crossing += int(np.floor(np.log2(t1)) > np.floor(np.log2(t0)))
    assert crossing >= 3                                       # the window does hold ends of LFO table pieces
    bank = VoiceBank(gv, gains=gains)
    for k in range(3):
        got = bank.render(blk, start=first + k * blk)
        w = want[k * blk:(k + 1) * blk]
        assert rms(got, w) <= 1e-7 and np.max(np.abs(got - w)) < 5e-7, (k, float(rms(got, w)))
    whole = bank.render(n, start=first)                        # one launch over all of it: piece ends inside the launch
    assert rms(whole, want) <= 1e-7
