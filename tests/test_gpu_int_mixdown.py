"""The reference-shaped INTEGER mixdown, end to end (SURVEY.md section 8 rows a9 + a10 + a11 composed).

What upstream can actually produce for "N voices mixed" ([RECALL]; the tree is not mounted, /root/reference/README.md:1-2): every
oscillator's float64 block -> ``Sample.from_osc_block`` (``int(32767 * v)`` through ``array('h')``) -> optionally ``Sample.stereo(l, r)``
(``audioop.tostereo``) -> the mixer's ``mixed = audioop.add(mixed, voice, 2)`` down the voices in order.  The pieces are tested
elsewhere (the quantiser on one oscillator, the chain on random PCM); here the COMPOSITION runs at BASELINE config 2 / config 3 size
and is compared byte for byte with the oracle's float64 samples quantised and folded by the LIVE CPython ``audioop``:

* route A, the existing API: each voice's float64 block on the GPU -> ``Sample.from_osc_device`` -> (``stereo``) -> ``mix_samples``;
* route B, the int16 materialisation (``sh_bank_generate_i16``: the quantiser in the epilogue of the generate kernels, 2 bytes per
  voice-sample, no float rows in HBM) -> ``sh_mix_chain_i16`` / ``sh_mix_chain_pan_i16``.

Bit-exactness of an int16 sample hinges on a float64 value that the GPU reproduces to ~1e-15 relative, not bit for bit (table sine +
Horner against libm's sin per partial): a sample whose ``scale * v`` lies closer to an integer than that error may truncate to the
neighbour.  The tests COUNT the oracle's samples within 1e-9 of a truncation boundary, print the count, and accept a differing
int16 sample only there (and only by one step); everywhere else the bytes are equal.
"""
import audioop
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 48000
N1S = SR                     # one second
NEAR = 1e-9


def _oracle_rows_worker(args):
    kind, n_total, seed, lo, hi, n, scale = args
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd.workloads import additive_voices, fm_voices
    voices, _g = (additive_voices(O, n_total, SR, seed=seed, partials=16) if kind == "additive" else fm_voices(O, n_total, SR, seed=seed))
    rows = np.empty((hi - lo, n), dtype=np.int16)
    near = []
    for k, i in enumerate(range(lo, hi)):
        v = CO.render(voices[i], n)
        y = scale * v
        rows[k] = CO.quantise(v, scale).astype(np.int16)
        # distance of scale*v from the nearest point where int() changes its value: every integer except 0 approached from
        # inside (-1, 1) -- truncation toward zero maps the whole of (-1, 1) to 0
        r = np.rint(y)
        d = np.abs(y - r)
        d[r == 0] = 1.0
        for j in np.nonzero(d < NEAR)[0]:
            near.append((i, int(j), float(d[j])))
    return lo, rows, near


def oracle_int16_rows(kind, n_total, seed, n, scale=32767.0):
    """(int16 [n_total, n], [(voice, sample, distance)] of the samples within NEAR of a truncation boundary) by the C oracle +
    the oracle's quantiser, the voices dealt to the host's cores."""
    import multiprocessing as mp
    nproc = max(1, min(os.cpu_count() or 1, 64, n_total))
    per = -(-n_total // nproc)
    jobs = [(kind, n_total, seed, lo, min(n_total, lo + per), n, scale) for lo in range(0, n_total, per)]
    if len(jobs) == 1:
        parts = [_oracle_rows_worker(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(len(jobs)) as pool:
            parts = pool.map(_oracle_rows_worker, jobs, chunksize=1)
    rows = np.empty((n_total, n), dtype=np.int16)
    near = []
    for lo, r, nr in parts:
        rows[lo:lo + len(r)] = r
        near += nr
    return rows, near


def audioop_chain(rows_bytes):
    mixed = rows_bytes[0]
    for r in rows_bytes[1:]:
        mixed = audioop.add(mixed, r, 2)
    return mixed


def _check_rows(got, want, near, what, strict=False):
    """got == want except, at most, at samples the oracle itself puts within NEAR of a truncation boundary (by one step).
    strict (the additive banks, round 6: the int16 boundary guard redoes the samples in doubt term by term): no allowance at all."""
    diff = np.argwhere(got != want)
    allowed = set() if strict else {(v, j) for v, j, _d in near}
    print("%s: %d of %d oracle samples within %.0e of a truncation boundary; %d int16 samples differ"
          % (what, len(near), want.size, NEAR, len(diff)))
    for v, j in diff:
        assert (int(v), int(j)) in allowed, "%s: voice %d sample %d: %d != %d away from any boundary" % (what, v, j, got[v, j], want[v, j])
        assert abs(int(got[v, j]) - int(want[v, j])) == 1
    return len(diff)


def _gpu_voices(kind, n_total, seed):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.workloads import additive_voices, fm_voices
    return additive_voices(G, n_total, SR, seed=seed, partials=16) if kind == "additive" else fm_voices(G, n_total, SR, seed=seed)


@pytest.mark.parametrize("kind,nvoices", [("additive", 64), ("fm", 1024)], ids=["config2_64_harmonics_adsr", "config3_1024_fm"])
def test_reference_shaped_int16_mixdown(gpu, kind, nvoices):
    from synthesizer_amd.mixer import VoiceBank, mix_samples
    from synthesizer_amd.sample import Sample
    n = N1S
    want_rows, near = oracle_int16_rows(kind, nvoices, 0, n)
    gv, gains = _gpu_voices(kind, nvoices, 0)

    # ---- route A: float64 block on the GPU -> Sample.from_osc_block's quantiser -> Samples
    monos = [Sample.from_osc_device(v._render_f64_device(0, n), n, SR) for v in gv]
    got_a = np.stack([np.frombuffer(s.view_frame_data(), dtype=np.int16) for s in monos])
    ndiff = _check_rows(got_a, want_rows, near, "%s route A (float64 block -> quantise)" % kind, strict=kind == "additive")

    # ---- route B: the int16 materialisation: the same integers (it quantises the same float64 values), in one launch
    bank = VoiceBank(gv, gains=gains)
    vs = bank.voice_samples(n)
    got_b = np.stack([np.frombuffer(s.view_frame_data(), dtype=np.int16) for s in vs])
    _check_rows(got_b, want_rows, near, "%s route B (sh_bank_generate_i16)" % kind, strict=kind == "additive")
    # (the lean materialisation folds amplitude and envelope gain into the sine before the polynomial: its float64 value may differ
    #  from the general code's in the last place, so A and B are each held to the oracle, not to each other -- but they agree
    #  wherever both agree with the oracle, i.e. everywhere but at the printed boundary samples)
    assert np.count_nonzero(got_a != got_b) <= 2 * max(1, len(near))

    # ---- mono mixdown: the mixer's chain over the quantised voices, byte for byte
    want_mono = audioop_chain([r.tobytes() for r in want_rows])
    if ndiff == 0:
        assert bytes(mix_samples(monos).view_frame_data()) == want_mono
    dev = bank.mixdown_i16_device(n)
    got_mono = dev.download_bytes(n * 2)
    if np.array_equal(got_b, want_rows):
        assert got_mono == want_mono
    # in any case: the chain of the GPU's own rows is audioop's chain of them
    assert got_mono == audioop_chain([r.tobytes() for r in got_b])

    # ---- stereo: Sample.stereo(l, r) (audioop.tostereo) per voice, then the chain
    want_st = audioop_chain([audioop.tostereo(r.tobytes(), 2, gl, gr) for r, (gl, gr) in zip(want_rows, gains)])
    if ndiff == 0:
        stereos = [s.stereo(gl, gr) for s, (gl, gr) in zip(monos, gains)]
        assert bytes(mix_samples(stereos).view_frame_data()) == want_st
    got_st = bank.mixdown_stereo_i16_device(n).download_bytes(n * 4)
    assert got_st == audioop_chain([audioop.tostereo(r.tobytes(), 2, gl, gr) for r, (gl, gr) in zip(got_b, gains)])
    if np.array_equal(got_b, want_rows):
        assert got_st == want_st


def test_int16_mixdown_that_saturates_mid_chain(gpu):
    """Loud voices: the running sum hits the rails in the middle of the chain and later voices pull it back -- the result
    depends on the ORDER of the adds and is not the clamped exact sum."""
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    n = 24000
    rng = np.random.default_rng(5)
    f = rng.uniform(200.0, 900.0, 24)
    ph = rng.uniform(0.0, 1.0, 24)
    harm = [(1, 1.0), (2, 0.3), (3, 0.1)]

    def make(mod):
        return [mod.EnvelopeFilter(mod.Harmonics(float(f[i]), harm, amplitude=0.7, phase=float(ph[i]), samplerate=SR), 0.01, 0.05, 10.0, 0.8, 0.1)
                for i in range(24)]
    want_rows = np.stack([CO.quantise(CO.render(v, n)).astype(np.int16) for v in make(O)])
    exact = np.clip(want_rows.astype(np.int64).sum(axis=0), -32768, 32767).astype(np.int16)
    want = audioop_chain([r.tobytes() for r in want_rows])
    assert want != exact.tobytes()                               # the chain saturated on the way: order matters here
    gains = [((1.0 + 0.03 * i) / 2.0, (1.0 - 0.03 * i) / 2.0) for i in range(24)]
    bank = VoiceBank(make(G), gains=gains)
    rows, stride = bank.generate_i16_device(n)
    got_rows = rows.download(np.int16, 24 * stride).reshape(24, stride)[:, :n]
    assert np.array_equal(got_rows, want_rows)
    assert bank.mixdown_i16_device(n).download_bytes(n * 2) == want
    want_st = audioop_chain([audioop.tostereo(r.tobytes(), 2, gl, gr) for r, (gl, gr) in zip(want_rows, gains)])
    assert bank.mixdown_stereo_i16_device(n).download_bytes(n * 4) == want_st


def test_generate_i16_overflow_odd_lengths_and_windows(gpu):
    """OverflowError where from_osc_block raises; odd row lengths (a tail tile, two-byte stores); a window late in the notes; a scale
    other than 32767; mixed kinds (general lists) -- each equal to quantise(the voice's float64 block)."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.sample import Sample
    harm = [(k, 1.0 / k) for k in range(1, 17)]
    voices = [G.Harmonics(220.0 * (1 + 0.37 * i), harm, amplitude=0.4, phase=0.1 * i, samplerate=SR) for i in range(70)]
    voices += [G.Sine(330.0, 0.3713, samplerate=SR), G.Square(100.0, 0.25, samplerate=SR), G.Sawtooth(441.0, 0.3, bias=0.1, samplerate=SR),
               G.Sine(500.0, 0.4, fm_lfo=G.Sine(3.0, 0.2, samplerate=SR), samplerate=SR),
               G.EnvelopeFilter(G.Triangle(120.0, 0.6, samplerate=SR), 0.01, 0.02, 0.1, 0.5, 0.1)]
    bank = VoiceBank(voices)
    for start, n, scale in ((0, 70001, 32767.0), (3 * SR + 17, 9999, 20000.0), (12345, 333, 32767.0), (0, 131072 + 64, 32767.0)):
        rows, stride = bank.generate_i16_device(n, start, scale)
        got = rows.download(np.int16, len(voices) * stride).reshape(len(voices), stride)[:, :n]
        for i, v in enumerate(voices):
            blk = v._render_f64_device(start, n)
            want = np.frombuffer(Sample.from_osc_device(blk, n, SR, amplitude_scale=scale).view_frame_data(), dtype=np.int16)
            bad = np.nonzero(got[i] != want)[0]
            # (lean rows fold the gains into the sine: the last place of the float64 value may differ from the general code's)
            assert len(bad) <= 1 and all(abs(int(got[i, j]) - int(want[j])) == 1 for j in bad), (start, n, i, bad[:4])
        rows.free()
    loud = VoiceBank([G.Harmonics(220.0, harm, amplitude=0.4, samplerate=SR), G.Harmonics(330.0, harm, amplitude=1.2, samplerate=SR)])
    with pytest.raises(OverflowError):
        loud.generate_i16_device(20000)
    loud.generate_i16_device(20000, scale=10000.0)[0].free()     # fits at a smaller scale; the flag of the failed call is gone
    loud.generate_i16_device(20000, check=False)[0].free()       # the streaming form only enqueues ...
    loud.generate_i16_device(20000, scale=10000.0, check=False)[0].free()
    with pytest.raises(OverflowError):
        VoiceBank.overflow_check()                               # ... the flag stays up until it is asked for
    VoiceBank.overflow_check()                                   # and is down again afterwards
    with pytest.raises(ValueError):
        loud.generate_i16_device(100, stride=101)                # rows are written as 32-bit pairs


def test_sine_peaks_on_rational_frequencies(gpu):
    """Where int16 parity is decided by ONE ulp: a Sine whose frequency divides the sample rate puts samples exactly on its peaks; there
    the reference's accumulated t is within 1e-12 of pi/2 + 2 pi k, math.sin returns exactly +-1.0 (the true value is 1 - 1e-24), and with
    amplitude * scale an integer (the oscillators' default amplitude 1.0 at scale 32767) the sample sits ON a truncation boundary:
    32767 if the sine is 1.0, 32766 if it is one ulp short.  A sine made by rotating a neighbour's (the bank kernels' frames 2 .. of a
    lane) is an ulp off about half the time -- so next to a peak those paths take the sine from the cosine, +-fma(-c / 2, c, 1): one
    rounding of 1 - c^2 / 2, which is what math.sin returns there.  Both routes, against the oracle, early and minutes into the notes
    (where the accumulated t has drifted up to 1e-6 rad off the peak)."""
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.sample import Sample
    n = 60000
    freqs = [1000.0, 1500.0, 750.0, 250.0, 3000.0, 125.0, 6000.0, 2000.0, 12000.0, 375.0, 4000.0, 500.0]
    for start in (0, 7 * SR, 200 * SR):
        want = np.stack([CO.quantise(CO.render(O.Sine(f, samplerate=SR), start + n)[start:]).astype(np.int16) for f in freqs])
        if start < 100 * SR:        # the peaks are there (minutes in, the accumulated t has drifted 1e-4 rad off them: none reaches 32767)
            assert int(np.count_nonzero(np.abs(want.astype(np.int32)) == 32767)) > 1000
        gv = [G.Sine(f, samplerate=SR) for f in freqs]
        got_a = np.stack([np.frombuffer(Sample.from_osc_device(v._render_f64_device(start, n), n, SR).view_frame_data(), dtype=np.int16) for v in gv])
        assert np.array_equal(got_a, want), ("route A", start, np.argwhere(got_a != want)[:5])
        rows, stride = VoiceBank(gv).generate_i16_device(n, start)
        got_b = rows.download(np.int16, len(gv) * stride).reshape(len(gv), stride)[:, :n]
        assert np.array_equal(got_b, want), ("route B", start, np.argwhere(got_b != want)[:5])
        # the same voices under an envelope (another lean form) and in the fused float path's general code (a short block)
        rows, stride = VoiceBank(gv).generate_i16_device(1000, start)
        got_c = rows.download(np.int16, len(gv) * stride).reshape(len(gv), stride)[:, :1000]
        assert np.array_equal(got_c, want[:, :1000]), ("route B, short rows", start, np.argwhere(got_c != want[:, :1000])[:5])


def test_fused_mixdown_equals_rows_plus_chain(gpu):
    """sh_bank_mixdown_i16 folds the int16 samples into the mixer's chain where they are made (stretches in which every voice takes the
    lean Harmonics loop) -- the bytes are those of sh_bank_generate_i16's rows folded by sh_mix_chain_i16, and of audioop.add over them."""
    N = gpu
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.workloads import additive_voices

    def fused_stretches():
        return N.lib().sh_get_option(N.SH_INFO_LAST_MIXDOWN_FUSED)
    # the bench's bank on its plateau: three 65 536-frame segments, a ragged end
    gv, gains = additive_voices(G, 1024, SR, seed=0, partials=16, adsr={"sustain": 1.0e6})
    bank = VoiceBank(gv, gains=gains)
    start, n = 3 * SR + 11, 150001
    fused = bank.mixdown_i16_device(n, start).download_bytes(n * 2)
    assert fused_stretches() == 1
    two = bank.mixdown_i16_device(n, start, two_step=True).download_bytes(n * 2)
    assert fused == two
    # from frame 0: the attack and decay go through the rows (general voices), the plateau behind them is folded -- one result
    n0 = 3 * 65536 + 777
    a = bank.mixdown_i16_device(n0, 0).download_bytes(n0 * 2)
    assert fused_stretches() == 1
    assert a == bank.mixdown_i16_device(n0, 0, two_step=True).download_bytes(n0 * 2)
    # short calls and other kinds of bank: the rows route, same entry point
    for m in (1, 100, 8191, 8192, 12345):
        assert bank.mixdown_i16_device(m, start).download_bytes(m * 2) == bank.mixdown_i16_device(m, start, two_step=True).download_bytes(m * 2), m
    fm = VoiceBank(_gpu_voices("fm", 80, 2)[0])
    assert fm.mixdown_i16_device(20000, 1000).download_bytes(40000) == fm.mixdown_i16_device(20000, 1000, two_step=True).download_bytes(40000)
    assert fused_stretches() == 0
    # loud voices on their plateau: the chain saturates on the way and later voices pull it back -- against the oracle's samples
    # quantised and folded by the live audioop
    rng = np.random.default_rng(8)
    nv, start, n = 70, SR, 70001
    f, ph = rng.uniform(150.0, 1200.0, nv), rng.uniform(0.0, 1.0, nv)
    harm = [(1, 1.0), (2, 0.3), (5, 0.1)]

    def make(mod):
        return [mod.EnvelopeFilter(mod.Harmonics(float(f[i]), harm, amplitude=0.6, phase=float(ph[i]), samplerate=SR), 0.01, 0.05, 30.0, 0.8, 0.1)
                for i in range(nv)]
    want_rows = np.stack([CO.quantise(CO.render(v, start + n)[start:]).astype(np.int16) for v in make(O)])
    want = audioop_chain([r.tobytes() for r in want_rows])
    assert want != np.clip(want_rows.astype(np.int64).sum(axis=0), -32768, 32767).astype(np.int16).tobytes()      # order matters here
    loud = VoiceBank(make(G))
    got = loud.mixdown_i16_device(n, start).download_bytes(n * 2)
    assert fused_stretches() == 1
    rows, stride = loud.generate_i16_device(n, start)
    got_rows = rows.download(np.int16, nv * stride).reshape(nv, stride)[:, :n]
    assert got == audioop_chain([r.tobytes() for r in got_rows])
    if np.array_equal(got_rows, want_rows):
        assert got == want
    with pytest.raises(OverflowError):
        VoiceBank([G.EnvelopeFilter(G.Harmonics(300.0, harm, amplitude=1.3, samplerate=SR), 0.01, 0.05, 30.0, 0.9, 0.1) for _ in range(64)]).mixdown_i16_device(70000, SR)


def test_int16_rows_and_mixdown_of_a_table_of_notes(gpu):
    """Notes with onsets and envelopes of their own (DelayFilter fused into the records; silent voices, attacks, releases inside the
    range): the int16 rows equal quantise(every voice's float64 block), and the mixdown -- rows route and fused route alike -- is
    audioop's chain over them."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.sample import Sample
    from synthesizer_amd.workloads import staggered_notes
    voices, gains = staggered_notes(G, 48, SR, seed=4, partials=16, period=0.5, notes=4)
    bank = VoiceBank(voices, gains=gains)
    for start, n in ((0, 70000), (30000, 24001), (SR, 2 * 65536 + 100)):
        rows, stride = bank.generate_i16_device(n, start)
        got = rows.download(np.int16, len(voices) * stride).reshape(len(voices), stride)[:, :n]
        for i in range(0, len(voices), 7):
            want = np.frombuffer(Sample.from_osc_device(voices[i]._render_f64_device(start, n), n, SR).view_frame_data(), dtype=np.int16)
            bad = np.nonzero(got[i] != want)[0]
            assert len(bad) <= 1 and all(abs(int(got[i, j]) - int(want[j])) == 1 for j in bad), (start, n, i, bad[:4])
        want_mono = audioop_chain([r.tobytes() for r in got])
        assert bank.mixdown_i16_device(n, start).download_bytes(n * 2) == want_mono
        assert bank.mixdown_i16_device(n, start, two_step=True).download_bytes(n * 2) == want_mono
        want_st = audioop_chain([audioop.tostereo(r.tobytes(), 2, gl, gr) for r, (gl, gr) in zip(got, gains)])
        assert bank.mixdown_stereo_i16_device(n, start).download_bytes(n * 4) == want_st
        rows.free()
