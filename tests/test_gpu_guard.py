"""The int16 boundary guard (sh_voice::guard_* in include/synthhip.h, params.exact_harmonics / int16_guard).

A Harmonics voice in the polynomial or Clenshaw form is sum a_k sin(k t) of the EXACT products k t; the reference (oscillators.py class
Harmonics, [RECALL]: `for k, amp in harmonics: h += sin(t * k) * amp`) rounds every `t * k` first.  The float64 samples lie up to
guard_t |t| + guard_c apart -- four orders of magnitude inside the 1e-6 float contract -- but int(32767 v) of the two differ wherever
an integer lies between them: one int16 sample in 1e6 ten seconds into a note, one in 2e5 after five minutes (measured below with the
guard off).  With the guard a bank's int16 routes redo exactly those samples term by term: EQUAL integers, no allowance, at any time.

The C oracle enters the generator's loop at the window's first sample (c_oracle.render_window: `start` float64 additions of the
increment, then the samples) -- tests/test_oracle_c.py holds that equal to a render from sample 0.
"""
import audioop

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 48000
H16 = [(k, 1.0 / k) for k in range(1, 17)]


def _lists():
    rng = np.random.default_rng(5)
    return {
        "1/k x16": H16,
        "odd 1/n to 15 (SquareH)": [(n, 1.0 / n) for n in range(1, 16, 2)],
        "random x16": [(k, float(rng.uniform(-1, 1))) for k in range(1, 17)],
        "repeats and a negative k": [(3, 0.5), (1, 1.0), (3, 0.25), (-2, 0.4), (16, 0.1)],
        "1 + 33 (Clenshaw)": [(1, 1.0), (33, 0.3)],
        "dense to 40 (Clenshaw)": [(k, 1.0 / k) for k in range(1, 41)],
    }


@pytest.mark.parametrize("name", list(_lists()))
def test_the_bound_holds_the_fast_forms(gpu, name):
    """guard_t |t| + guard_c bounds |fast form - term-by-term| with room: the measured distance stays under half of it, at the start of
    a note and 10 s / 300 s / 3000 s in, for every list (the distance itself is printed: it is the polynomial form's whole error)."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import params
    harm = _lists()[name]
    n = 1 << 16
    worst = 0.0
    for f, amp, phase in ((440.0, 0.5, 0.2), (3520.0, 1.0, 0.7), (55.0, 0.05, 0.0)):
        for seconds in (0, 10, 300, 3000):
            start = seconds * SR
            got = {}
            for exact in (False, None):
                params.exact_harmonics = exact
                try:
                    g = G.Harmonics(f, harm, amp, phase=phase, samplerate=SR)
                    sp = g.spec()
                    assert sp.harm_guard is not None and (sp.harm_poly is not None or sp.harm_dense is not None)
                    got[exact] = g.render_f64(n, start=start)
                finally:
                    params.exact_harmonics = None
            voices = G.pack_voices([sp])[0]
            t_end = abs(sp.carrier.value(start + n))
            bound = float(voices["guard_t"][0]) * t_end + float(voices["guard_c"][0])
            dist = float(np.max(np.abs(got[False] - got[None])))
            worst = max(worst, dist / bound)
            assert dist <= 0.5 * bound, (name, f, seconds, dist, bound)
    print("%s: fast form vs term by term, worst distance / bound = %.3f" % (name, worst))


def _oracle_rows(make_o, start, n):
    from oracle import c_oracle as CO
    return np.stack([CO.quantise(CO.render_window(o, start, n)).astype(np.int16) for o in make_o()])


def _voices(mod, nv, seed=3, envelope=True, loud=False):
    rng = np.random.default_rng(seed)
    f = np.exp(rng.uniform(np.log(55.0), np.log(3520.0), nv))
    amp = rng.uniform(0.15, 0.5, nv) if loud else rng.uniform(0.1, 1.0, nv) / np.sqrt(nv)
    ph = rng.uniform(0.0, 1.0, nv)
    out = []
    for i in range(nv):
        o = mod.Harmonics(float(f[i]), H16, amplitude=float(amp[i]), phase=float(ph[i]), samplerate=SR)
        out.append(mod.EnvelopeFilter(o, 0.01, 0.05, 1.0e6, 0.6, 0.2) if envelope else o)
    return out


@pytest.mark.parametrize("seconds", [10, 300])
def test_int16_rows_and_mixdown_equal_the_oracle_late_in_the_notes(gpu, seconds):
    """BASELINE config 2's size (64 Harmonics x16 voices under an ADSR), a window of 2^20 frames 10 s and 300 s into the notes: every
    int16 row of generate_i16 and the fused mixdown equal the oracle's quantised samples / the live audioop chain over them -- 0 of
    67 M samples differ, no allowance.  The same bank built without the guard (params.int16_guard = False: rounds 1-5) differs in
    some: the test has teeth."""
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import params
    from synthesizer_amd.mixer import VoiceBank
    nv, n, start = 64, 1 << 20, seconds * SR
    want = _oracle_rows(lambda: _voices(O, nv), start, n)
    bank = VoiceBank(_voices(G, nv))
    rows, stride = bank.generate_i16_device(n, start)
    got = rows.download(np.int16, nv * stride).reshape(nv, stride)[:, :n]
    ndiff = int(np.count_nonzero(got != want))
    assert ndiff == 0, "%d of %d int16 samples differ %d s in (guard on)" % (ndiff, want.size, seconds)
    chain = want[0].tobytes()
    for r in want[1:]:
        chain = audioop.add(chain, r.tobytes(), 2)
    assert bank.mixdown_i16_device(n, start).download_bytes(n * 2) == chain
    params.int16_guard = False
    try:
        plain = VoiceBank(_voices(G, nv))
    finally:
        params.int16_guard = True
    prow, pstride = plain.generate_i16_device(n, start)
    pgot = prow.download(np.int16, nv * pstride).reshape(nv, pstride)[:, :n]
    pd = np.abs(pgot.astype(np.int32) - want.astype(np.int32))
    print("%d s in: guard on 0 of %d int16 samples differ; guard off %d (all by one step: %s)" % (seconds, want.size, int(np.count_nonzero(pd)), bool(pd.max() <= 1)))
    assert pd.max() <= 1
    if seconds >= 300:
        assert np.count_nonzero(pd) > 0          # (64 quiet voices: ~10 crossings expected at 300 s; at 10 s there may be none)


def test_loud_voices_short_rows_and_every_kernel_shape(gpu):
    """Loud voices (amplitude 0.15 .. 0.5 -- the 1/k series peaks at 1.85 -- the guard's reach grows with it) 300 s in, without an envelope, at row lengths that take the
    other materialisation kernels: 1000 frames (k_generate<1>), 3000 (<2>), 10 000 (lists / lean at four frames per lane), 70 000 (two
    segments, a partial last tile) -- and a bank that mixes the Harmonics with FM Sine voices (the lean lists of k_generate_lists).
    Every Harmonics row equals the oracle's quantised samples."""
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, start = 16, 300 * SR + 777
    for n in (1000, 3000, 10000, 70001):
        want = _oracle_rows(lambda: _voices(O, nv, seed=9, envelope=False, loud=True), start, n)
        bank = VoiceBank(_voices(G, nv, seed=9, envelope=False, loud=True))
        rows, stride = bank.generate_i16_device(n, start)
        got = rows.download(np.int16, nv * stride).reshape(nv, stride)[:, :n]
        assert np.array_equal(got, want), (n, int(np.count_nonzero(got != want)))
    n = 20000
    gv = _voices(G, nv, seed=9, envelope=False, loud=True)
    mixed = [x for pair in zip(gv, [G.Sine(200.0 + 10 * k, 0.5, fm_lfo=G.Sine(3.0, 0.02, samplerate=SR), samplerate=SR) for k in range(nv)]) for x in pair]
    want = _oracle_rows(lambda: _voices(O, nv, seed=9, envelope=False, loud=True), start, n)
    rows, stride = VoiceBank(mixed).generate_i16_device(n, start)
    got = rows.download(np.int16, 2 * nv * stride).reshape(2 * nv, stride)[0::2, :n]
    assert np.array_equal(got, want), int(np.count_nonzero(got != want))


def test_single_oscillator_blocks_are_the_term_by_term_sum(gpu):
    """A single Harmonics oscillator (blocks(), Sample.from_osc_block: launch-bound whatever its form) sums term by term by default:
    float64 within 1.5e-15 of the oracle and equal int16, 300 s in; params.exact_harmonics = False gives the polynomial form back."""
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import params
    from synthesizer_amd.sample import Sample
    start, n = 300 * SR, 1 << 18
    want = CO.render_window(O.Harmonics(440.0, H16, 0.5, phase=0.2, samplerate=SR), start, n)
    g = G.Harmonics(440.0, H16, 0.5, phase=0.2, samplerate=SR)
    blk = g._render_f64_device(start, n)
    assert float(np.max(np.abs(blk.download(np.float64, n) - want))) <= 1.5e-15
    got = np.frombuffer(Sample.from_osc_device(blk, n, SR).view_frame_data(), dtype=np.int16)
    assert np.array_equal(got, CO.quantise(want).astype(np.int16))
    params.exact_harmonics = False
    try:
        fast = G.Harmonics(440.0, H16, 0.5, phase=0.2, samplerate=SR).render_f64(n, start=start)
    finally:
        params.exact_harmonics = None
    assert 1e-12 < float(np.max(np.abs(fast - want))) < 1e-9


def test_a_nan_amplitude_is_an_overflow_on_every_route(gpu):
    """a NaN sample (NaN amplitude) converted to 0 silently on the lean int16 path while the general path raised; both raise now."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd import params
    for guard in (True, False):                    # (False: banks without guard lists take the int16 kernels without the check)
        params.int16_guard = guard
        try:
            for n in (1000, 20000):
                v = [G.Harmonics(440.0, H16, 0.1, samplerate=SR), G.Harmonics(550.0, H16, float("nan"), samplerate=SR)]
                bank = VoiceBank(v)
                with pytest.raises(OverflowError):
                    bank.generate_i16_device(n, 0)
                with pytest.raises(OverflowError):
                    bank.mixdown_i16_device(n, 0)
        finally:
            params.int16_guard = True
