"""GPU parity: voices with an onset (sh_voice::start_frame) -- upstream's DelayFilter(voice, seconds) fused into the voice's
record, so that a bank can hold notes that start at different times.  Oracle: oracle/synth_oracle.py DelayFilter over the same
voices (zeros first, then the source from ITS sample 0), and the C oracle with zero-padded rows for the larger banks."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import synth_oracle as O
from tests.helpers import rms

pytestmark = pytest.mark.gpu
SR = 48000
RMS_TOL = 1e-6


def _voices(m):
    harm = [(k, 1.0 / k) for k in range(1, 17)]
    env = lambda src: m.EnvelopeFilter(src, 0.01, 0.02, 0.05, 0.6, 0.03)
    specs = [
        (m.Sine(440.0, 0.2, samplerate=SR), 0.0),
        (env(m.Harmonics(220.0, harm, 0.3, phase=0.3, samplerate=SR)), 1 / SR),
        (env(m.Harmonics(330.0, harm, 0.3, phase=0.7, samplerate=SR)), 63 / SR),
        (m.Square(1000.0, 0.2, samplerate=SR), 64 / SR),                     # an edge on every 24th sample of ITS time base
        (m.Pulse(250.0, 0.2, pulsewidth=0.25, samplerate=SR), 100 / SR),
        (env(m.Sawtooth(150.0, 0.3, samplerate=SR)), 511 / SR),
        (m.Sine(330.0, 0.25, fm_lfo=m.Sine(5.0, 0.03, samplerate=SR), samplerate=SR), 512 / SR),     # closed-form FM restarts with the voice
        (env(m.Triangle(90.0, 0.3, samplerate=SR)), 4095 / SR),
        (m.WhiteNoise(800.0, 0.1, samplerate=SR, seed=5), 5000 / SR),
        (m.Linear(0.0, 1e-4, samplerate=SR), 7777 / SR),
        (env(m.Harmonics(110.0, harm, 0.3, samplerate=SR)), 9999 / SR),
        (env(m.Harmonics(55.0, harm, 0.3, samplerate=SR)), 0.5),              # beyond the rendered range: never sounds
    ]
    return [m.DelayFilter(v, d) if d else v for v, d in specs]


def test_voices_with_onsets_in_a_bank(gpu):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    gv, ov = _voices(G), _voices(O)
    rng = np.random.default_rng(9)
    gains = [(float(np.float32(a)), float(np.float32(b))) for a, b in rng.uniform(0.2, 1.0, (len(gv), 2))]
    n = 12000
    want = np.array(O.mix_bus([v.take(n) for v in ov], gains), dtype=np.float64)
    bank = VoiceBank(gv, gains=gains)
    assert bank._rows is None                               # every delayed voice became ONE record with an onset: no rows
    got = bank.render(n)
    assert rms(got, want) <= RMS_TOL and np.max(np.abs(got - want)) < 5e-7
    # block by block (onsets inside blocks, at block starts, blocks in front of onsets), and odd windows
    for block in (1000, 512, 64):
        parts = np.concatenate([bank.render(block, start=s) for s in range(0, 6144, block)])
        assert rms(parts, want[:len(parts)]) <= RMS_TOL, block
    for start, m in ((63, 2), (64, 1), (99, 3), (5000, 1), (4000, 4000), (9998, 1500)):
        assert rms(bank.render(m, start=start), want[start:start + m]) <= RMS_TOL, (start, m)
    # the reference-shaped two-step route: rows with leading zeros, exactly zero in front of the onset
    rows = bank.generate(n)
    for i, v in enumerate(ov):
        w = np.array(v.take(n), dtype=np.float64)
        assert np.max(np.abs(rows[i] - w)) < 2e-7, i
        lead = int(SR * v._seconds) if isinstance(v, O.DelayFilter) else 0          # (upstream's own expression)
        assert not rows[i][:min(lead, n)].any(), i
    assert rms(bank.render_two_step(n), want) <= RMS_TOL
    # Square / Pulse samples are EQUAL to float32(oracle): the accumulated phase restarts exactly at the onset
    for i in (3, 4):
        assert np.array_equal(rows[i], np.array(ov[i].take(n), dtype=np.float32)), i


def test_notes_that_start_at_different_times(gpu):
    """320 additive voices (several voice groups: split launches, segmented launches, the two-stream pipeline) whose notes start
    anywhere in the first 30 000 frames and end (ADSR without a held sustain): long blocks from frame 0, the same frames block by
    block in a pipelined run, against the C oracle's zero-padded rows."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, n = 320, 49152
    rng = np.random.default_rng(21)
    f = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), nv))
    amp = rng.uniform(0.1, 1.0, nv) / np.sqrt(nv)
    ph = rng.uniform(0, 1, nv)
    onset = rng.integers(0, 30000, nv)
    onset[:4] = (0, 16384, 16383, 512)
    gains = [(float(np.float32(a)), float(np.float32(b))) for a, b in rng.uniform(0.0, 1.0, (nv, 2))]
    harm = [(k, 1.0 / k) for k in range(1, 17)]

    def build(m):
        out = []
        for i in range(nv):
            v = m.EnvelopeFilter(m.Harmonics(float(f[i]), harm, float(amp[i]), phase=float(ph[i]), samplerate=SR), 0.01, 0.05, 0.2, 0.6, 0.1)
            out.append(m.DelayFilter(v, int(onset[i]) / SR) if onset[i] else v)
        return out
    gv, ov = build(G), build(O)
    rows = np.zeros((nv, n))
    for i, v in enumerate(ov):
        src = v._source if isinstance(v, O.DelayFilter) else v
        d = int(SR * v._seconds) if isinstance(v, O.DelayFilter) else 0             # upstream's expression: may be onset - 1 after the division
        rows[i, d:] = CO.render(src, n - d)
    want = CO.mix_bus(rows, gains)
    bank = VoiceBank(gv, gains=gains)
    assert rms(bank.render(n), want) <= RMS_TOL
    for block in (16384, 4096):
        bufs = [N.DeviceBuffer(block * 8) for _ in range(n // block)]
        for k in range(n // block):
            bank.render_device(block, k * block, bus_f32=bufs[k])
        parts = np.concatenate([b.download(np.float32, block * 2).reshape(block, 2) for b in bufs])
        assert rms(parts, want[:len(parts)]) <= RMS_TOL, block
    assert rms(bank.render_two_step(20000, start=10000), want[10000:30000]) <= RMS_TOL


_CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import staggered_notes
N.ensure_init(0)
v, g = staggered_notes(G, 256, 48000, seed=4, period=0.5, notes=4)
bank = VoiceBank(v, gains=g)
out = [bank.render(24000, start=k * 24000) for k in range(4)]
np.save(sys.argv[1], np.stack(out))
'''


def test_staggered_notes_tile_by_tile(gpu, tmp_path):
    """The bench's staggered workload at a quarter of its size (256 players x 4 rounds, a note every half second): blocks of half a
    second through the tile-classified launch (lean per (voice, tile) pair, general code for the pairs with an onset, a corner or a
    piece end) against the C oracle -- and against the same blocks with the classification switched off (SYNTHHIP_NO_TILES=1: every
    voice that holds a corner anywhere in the block goes through the general code)."""
    import ctypes
    import os
    import subprocess
    import sys
    from pathlib import Path
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.workloads import staggered_notes
    slots, notes, period, block = 256, 4, 0.5, 24000
    gv, gains = staggered_notes(G, slots, SR, seed=4, period=period, notes=notes)
    ov, _ = staggered_notes(O, slots, SR, seed=4, period=period, notes=notes)
    n = 4 * block
    rows = np.zeros((len(ov), n))
    for i, v in enumerate(ov):
        d = int(SR * v._seconds) if isinstance(v, O.DelayFilter) else 0
        if d < n:
            rows[i, d:] = CO.render(v._source if isinstance(v, O.DelayFilter) else v, n - d)
    want = CO.mix_bus(rows, gains)
    bank = VoiceBank(gv, gains=gains)
    before = N.debug_counters()["tiled_launches"]
    got = [bank.render(block, start=k * block) for k in range(4)]
    assert N.debug_counters()["tiled_launches"] - before == 4
    for k in range(4):
        assert rms(got[k], want[k * block:(k + 1) * block]) <= RMS_TOL, k
    assert np.abs(want).max() > 0.05
    # a pipelined run of the same blocks (records two launches ahead, folds taken over, tile sets per stream)
    ring = [N.DeviceBuffer(block * 8) for _ in range(4)]
    for k in range(4):
        bank.render_device(block, k * block, bus_f32=ring[k])
    for k in range(4):
        assert np.array_equal(ring[k].download(np.float32, block * 2).reshape(block, 2), got[k]), k
    # one long launch over everything (another tile count, a partial last tile)
    whole = bank.render(n - 100)
    assert rms(whole, want[:n - 100]) <= RMS_TOL
    # a call of several seconds is carried out as launches of 2^17 frames (a tile set per launch): the same frames as those launches
    long_n = (1 << 17) + 40000
    assert np.array_equal(bank.render(long_n), np.concatenate([bank.render(1 << 17), bank.render(40000, start=1 << 17)]))
    # the classification switched off: same buses up to float64 rounding (the lean pairs fold the envelope's line into the gains)
    ref = tmp_path / "notiles.npy"
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if not k.startswith("SYNTHHIP_") or k in ("SYNTHHIP_LIB", "SYNTHHIP_DEVICE")}
    p = subprocess.run([sys.executable, "-c", _CHILD % str(root), str(ref)], env=dict(env, SYNTHHIP_NO_TILES="1"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    other = np.load(ref)
    for k in range(4):
        assert rms(got[k], other[k]) <= 2e-8, k


def _mixed_notes(m, n, seed):
    """n polynomial-Harmonics notes with onsets over three seconds, in the order they start: a third under a short-attack ADSR of
    their own (the attack ends inside the onset's tile), a third without any envelope (the onset is a step), a third under the
    bench's ADSR; fundamentals up to 9 kHz (the phase sum of a high voice runs through a dozen binades in its first tile)."""
    rng = np.random.default_rng(seed)
    f = np.exp(rng.uniform(np.log(40.0), np.log(9000.0), n))
    amp = rng.uniform(0.1, 1.0, n) / np.sqrt(n)
    phase = rng.uniform(0.0, 1.0, n)
    onsets = np.sort(rng.integers(0, 3 * SR, n))
    onsets[:3] = (0, 1, 511)
    kinds = rng.integers(0, 3, n)
    gains = [(float(np.float32(g)), float(np.float32(1.0 - g))) for g in rng.uniform(0.0, 1.0, n)]
    voices = []
    for i in range(n):
        nh = int(rng.integers(1, 17))
        osc = m.Harmonics(float(f[i]), [(k, 1.0 / k) for k in range(1, nh + 1)], amplitude=float(amp[i]), phase=float(phase[i]), samplerate=SR)
        if kinds[i] == 0:
            osc = m.EnvelopeFilter(osc, float(rng.uniform(0.0002, 0.004)), float(rng.uniform(0.001, 0.02)), float(rng.uniform(0.0, 0.3)), 0.5,
                                   float(rng.uniform(0.001, 0.3)))
        elif kinds[i] == 2:
            osc = m.EnvelopeFilter(osc, 0.01, 0.05, 0.5, 0.6, 0.2)
        voices.append(m.DelayFilter(osc, int(onsets[i]) / SR) if onsets[i] else osc)
    return voices, gains, onsets


def test_walk_pairs_steps_and_chunk_ranges(gpu):
    """Tile-classified launches over a table of 1500 notes (24 chunks, a few of which sound in any block): onset tiles as walk pairs
    (many piece ends, the attack's end behind the onset, onsets without an envelope), the range of chunks per block, a stream of
    blocks, a jump back, one long launch -- against the C oracle, block by block."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    n_voices, block, nblocks = 1500, 16384, 7
    gv, gains, onsets = _mixed_notes(G, n_voices, 21)
    ov, _, _ = _mixed_notes(O, n_voices, 21)
    n = block * nblocks
    rows = np.zeros((n_voices, n))
    for i, v in enumerate(ov):
        d = int(SR * v._seconds) if isinstance(v, O.DelayFilter) else 0      # (DelayFilter's own count of frames: onsets[i] or one less)
        if d < n:
            rows[i, d:] = CO.render(v._source if isinstance(v, O.DelayFilter) else v, n - d)
    want = CO.mix_bus(rows, gains)
    assert np.abs(want).max() > 0.02
    bank = VoiceBank(gv, gains=gains)
    ring = [N.DeviceBuffer(block * 8) for _ in range(4)]
    read = lambda b: b.download(np.float32, block * 2).reshape(block, 2)
    before = N.debug_counters()
    for k in list(range(nblocks)) + [2, 3, 4, 0, 1]:              # a stream, a jump back into it, the start again
        bank.render_device(block, k * block, bus_f32=ring[k & 3])
        assert rms(read(ring[k & 3]), want[k * block:(k + 1) * block]) <= RMS_TOL, k
    # the stream without reading in between (the pipeline stays up: sets resolved two launches ahead, chunk ranges that move)
    for rep in range(2):
        for k in range(nblocks):
            bank.render_device(block, k * block, bus_f32=ring[k & 3])
    for k in range(nblocks - 4, nblocks):
        assert rms(read(ring[k & 3]), want[k * block:(k + 1) * block]) <= RMS_TOL, k
    after = N.debug_counters()
    assert after["tiled_launches"] - before["tiled_launches"] == nblocks + 5 + 2 * nblocks          # every one of them tile-classified
    assert after["tiled_predicted"] - before["tiled_predicted"] >= 2 * nblocks - 4                  # ... the stream on sets resolved ahead
    whole = bank.render(n - 77)
    assert rms(whole, want[:n - 77]) <= RMS_TOL
    assert np.max(np.abs(whole - want[:n - 77])) < 5e-6


def _notes_of_every_kind(m, n, seed):
    """n notes of the plain kinds -- Harmonics, Sine, Sawtooth, Square, Triangle, Pulse -- and Sine carriers with a Sine LFO, with
    onsets over two seconds in the order they start, two thirds of them under ADSRs of their own."""
    rng = np.random.default_rng(seed)
    onsets = np.sort(rng.integers(0, 2 * SR, n))
    onsets[:2] = (0, 300)
    gains = [(float(np.float32(g)), float(np.float32(1.0 - g))) for g in rng.uniform(0.0, 1.0, n)]
    voices = []
    for i in range(n):
        f = float(np.exp(rng.uniform(np.log(40.0), np.log(6000.0))))
        amp = float(rng.uniform(0.1, 1.0)) / np.sqrt(n)
        ph = float(rng.uniform(0.0, 1.0))
        k = int(rng.integers(0, 7))
        if k == 6:                                                   # a Sine carrier with a Sine LFO (closed-form FM: restarts with the note)
            lfo = m.Sine(float(rng.uniform(0.5, 9.0)), float(rng.uniform(0.0, 0.05)), phase=float(rng.uniform(0.0, 1.0)), samplerate=SR)
            osc = m.Sine(f, amp, phase=ph, fm_lfo=lfo, samplerate=SR)
        elif k == 0:
            nh = int(rng.integers(2, 17))
            osc = m.Harmonics(f, [(q, 1.0 / q) for q in range(1, nh + 1)], amplitude=amp, phase=ph, samplerate=SR)
        elif k == 1:
            osc = m.Sine(f, amp, phase=ph, samplerate=SR)
        elif k == 2:
            osc = m.Sawtooth(f, amp, phase=ph, samplerate=SR)
        elif k == 3:
            osc = m.Square(f, amp, phase=ph, samplerate=SR)
        elif k == 4:
            osc = m.Triangle(f, amp, phase=ph, samplerate=SR)
        else:
            osc = m.Pulse(f, amp, phase=ph, pulsewidth=float(rng.uniform(0.05, 0.95)), samplerate=SR)
        if rng.random() < 0.67:
            osc = m.EnvelopeFilter(osc, float(rng.uniform(0.0005, 0.02)), float(rng.uniform(0.002, 0.08)), float(rng.uniform(0.0, 0.5)), 0.6,
                                   float(rng.uniform(0.002, 0.3)))
        voices.append(m.DelayFilter(osc, int(onsets[i]) / SR) if onsets[i] else osc)
    return voices, gains


def test_notes_of_every_plain_kind_tile_by_tile(gpu):
    """A table of notes of all the plain kinds (the tiles kernel with the waveform branch): blocks of real-time length and of a
    third of a second, as a stream and with a jump, against the C oracle; the launches are tile-classified."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    n_voices = 900
    gv, gains = _notes_of_every_kind(G, n_voices, 33)
    ov, _ = _notes_of_every_kind(O, n_voices, 33)
    n = 6 * 16384
    rows = np.zeros((n_voices, n))
    for i, v in enumerate(ov):
        d = int(SR * v._seconds) if isinstance(v, O.DelayFilter) else 0
        if d < n:
            rows[i, d:] = CO.render(v._source if isinstance(v, O.DelayFilter) else v, n - d)
    want = CO.mix_bus(rows, gains)
    bank = VoiceBank(gv, gains=gains)
    for block in (16384, 2048):
        nblocks = n // block if block == 16384 else 24
        ring = [N.DeviceBuffer(block * 8) for _ in range(4)]
        before = N.debug_counters()["tiled_launches"]
        plan = list(range(nblocks)) + [1, 2]
        for k in plan:
            bank.render_device(block, k * block, bus_f32=ring[k & 3])
            got = ring[k & 3].download(np.float32, block * 2).reshape(block, 2)
            w = want[k * block:(k + 1) * block]
            assert rms(got, w) <= RMS_TOL, (block, k)
            assert np.max(np.abs(got - w)) < 2e-6, (block, k, float(np.max(np.abs(got - w))))
        assert N.debug_counters()["tiled_launches"] - before == len(plan)


def test_a_table_of_notes_is_the_same_nine_billion_frames_later(gpu):
    """Shift invariance at positions beyond 2^32 frames (two days into a stream at 48 kHz): a table of notes whose onsets are all K
    frames later, rendered K frames later, is the table itself bit for bit -- phases and envelopes count from a note's own first
    frame, tiles and chunk ranges from the launch's; nothing on the way may keep an absolute frame in 32 bits.  Streams of one-second
    blocks (the lean / general tile kernels) and of 2048-frame chunks (the merged kernel), and the counters say the tile path ran."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.workloads import staggered_notes
    voices, gains = staggered_notes(G, 192, SR, seed=4, period=0.5, notes=4)
    late_by = (2 ** 33 + 12345) / SR
    late = [G.DelayFilter(v, late_by) for v in voices]
    K = late[0]._shift
    assert K > 2 ** 33 and all(v._shift == K for v in late)
    assert late[5].spec().start_frame == voices[5].spec().start_frame + K
    a, b = VoiceBank(voices, gains=gains), VoiceBank(late, gains=gains)
    c0 = N.debug_counters()
    for blk, count in ((SR, 3), (2048, 12)):
        for s in range(count):
            want = a.render(blk, start=s * blk)
            got = b.render(blk, start=K + s * blk)
            assert np.array_equal(got, want), (blk, s, float(np.max(np.abs(got - want))))
            assert np.max(np.abs(want)) > 0.0
    # a launch across K itself: silence in front of the first onset, then the table's first frames (another launch shape than
    # the 96-frame render beside it: equal up to the order of the float64 sums)
    edge, head = np.asarray(b.render(4096, start=K - 4000)).reshape(-1), np.asarray(a.render(96, start=0)).reshape(-1)
    assert not edge[:8000].any() and np.max(np.abs(edge[8000:].astype(np.float64) - head)) < 1e-7 and head.any()
    c1 = N.debug_counters()
    assert c1["tiled_launches"] - c0["tiled_launches"] >= 2 * (3 + 12)


def test_fm_notes_in_runs_of_a_tile_list(gpu):
    """Round 4: the lean pairs of a (tile, chunk) list come in three runs -- Harmonics, FM Sine, plain waveforms -- walked by a loop each,
    and an FM Sine pair on one piece under one line takes the lean lists' arithmetic (lean_fm_frames: time by addition, the LFO's cosine
    by recurrence; every other LFO with a bias).  A table of 300 notes -- two thirds FM Sine, the rest Harmonics and Sawtooth, so that
    chunks hold every composition of runs -- under ADSRs with sustains long enough for whole tiles of sustain and release, as a stream of
    one-third-second blocks and of 2048-frame chunks, against the C oracle."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    n_voices, n = 300, 5 * 16384

    def notes(m):
        rng = np.random.default_rng(77)
        onsets = np.sort(rng.integers(0, SR, n_voices))
        onsets[0] = 0
        out = []
        for i in range(n_voices):
            f = float(np.exp(rng.uniform(np.log(40.0), np.log(6000.0))))
            amp = float(rng.uniform(0.1, 1.0)) / np.sqrt(n_voices)
            ph = float(rng.uniform(0.0, 1.0))
            k = int(rng.integers(0, 6))
            lfo = m.Sine(float(rng.uniform(0.5, 9.0)), float(rng.uniform(0.0, 0.05)), phase=float(rng.uniform(0.0, 1.0)),
                         bias=(0.02 if i % 2 else 0.0), samplerate=SR)
            if k < 4:
                osc = m.Sine(f, amp, phase=ph, fm_lfo=lfo, samplerate=SR)
            elif k == 4:
                osc = m.Harmonics(f, [(q, 1.0 / q) for q in range(1, 9)], amplitude=amp, phase=ph, samplerate=SR)
            else:
                osc = m.Sawtooth(f, amp, phase=ph, samplerate=SR)
            if i % 5:
                osc = m.EnvelopeFilter(osc, 0.005, 0.03, float(rng.uniform(0.2, 0.6)), 0.6, float(rng.uniform(0.05, 0.3)))
            out.append(m.DelayFilter(osc, int(onsets[i]) / SR) if onsets[i] else osc)
        return out
    gains = [(float(np.float32(g)), float(np.float32(1.0 - g))) for g in np.random.default_rng(78).uniform(0.0, 1.0, n_voices)]
    gv, ov = notes(G), notes(O)
    rows = np.zeros((n_voices, n))
    for i, v in enumerate(ov):
        d = int(SR * v._seconds) if isinstance(v, O.DelayFilter) else 0
        if d < n:
            rows[i, d:] = CO.render(v._source if isinstance(v, O.DelayFilter) else v, n - d)
    want = CO.mix_bus(rows, gains)
    bank = VoiceBank(gv, gains=gains)
    for block in (16384, 2048):
        nblocks = n // block
        ring = [N.DeviceBuffer(block * 8) for _ in range(4)]
        before = N.debug_counters()["tiled_launches"]
        for k in range(nblocks):
            bank.render_device(block, k * block, bus_f32=ring[k & 3])
            got = ring[k & 3].download(np.float32, block * 2).reshape(block, 2)
            w = want[k * block:(k + 1) * block]
            assert rms(got, w) <= RMS_TOL, (block, k)
            assert np.max(np.abs(got - w)) < 2e-6, (block, k, float(np.max(np.abs(got - w))))
        assert N.debug_counters()["tiled_launches"] - before == nblocks
