"""What "held to the 1e-6 contract" means for an FM voice, pinned.

The reference's carrier angle is t * freq + phase_correction with phase_correction += (freq_previous - freq) * t per sample: a running
float64 sum whose additions round at ulp(f depth t).  The closed form (oscillators.LfoTable) follows the reference's ACCUMULATED phases
exactly, so what separates the two is that rounding walk alone: RMS <= c * ulp(f depth t) * sqrt(n).  A regression to sums along the
ideal lines (rounds 1-3: error ~ t^2, 5e-4 at 300 s for the loudest case) fails this test by two to three orders of magnitude.
Also here: the int16 boundary-crossing rate of the polynomial Harmonics form late in a note, and params.exact_harmonics, which removes it.
"""
import math

import numpy as np
import pytest

from tests.helpers import rms

SR = 48000
WINDOW = 16384


@pytest.mark.gpu
@pytest.mark.parametrize("f,depth", [(440.0, 0.05), (440.0, 0.5), (3520.0, 0.05), (3520.0, 0.5)])
def test_fm_error_stays_under_the_rounding_walk_bound(gpu, f, depth):
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    o = O.Sine(f, 0.9, phase=0.13, fm_lfo=O.Sine(5.0, depth, phase=0.31, samplerate=SR), samplerate=SR)
    g = G.Sine(f, 0.9, phase=0.13, fm_lfo=G.Sine(5.0, depth, phase=0.31, samplerate=SR), samplerate=SR)
    want_all = CO.render(o, 300 * SR + WINDOW)
    seen = []
    for seconds in (1, 30, 300):
        start = seconds * SR
        got = g.render_f64(WINDOW, start=start)
        err = rms(got, want_all[start:start + WINDOW])
        # the walk's bound, plus the per-sample rounding of the reference's own t * freq + phase_correction (not a walk: half an ulp of
        # an angle of 2 pi f t)
        floor = 0.5 * 2.0 ** -52 * (2.0 * math.pi * f * seconds) * 0.9
        bound = g.fm_error_bound(seconds + WINDOW / SR) + floor
        seen.append((seconds, err, bound))
        assert err <= bound, (f, depth, seen)
    print("FM %6.0f Hz depth %.2f: RMS vs the C oracle at 1 / 30 / 300 s: %s; contract horizon %.0f s"
          % (f, depth, "  ".join("%.1e (bound %.1e)" % (e, b) for _s, e, b in seen), g.fm_contract_horizon()))
    assert seen[-1][1] < 1e-6 or g.fm_contract_horizon() < 300.0        # inside the horizon the contract holds


def test_fm_contract_horizon_host_logic():
    from synthesizer_amd import oscillators as G
    plain = G.Sine(440.0, samplerate=SR)
    assert plain.fm_contract_horizon() == float("inf") and plain.fm_error_bound(1000.0) == 0.0
    loud = G.Sine(3520.0, fm_lfo=G.Sine(5.0, 0.5, samplerate=SR), samplerate=SR)
    soft = G.Sine(440.0, fm_lfo=G.Sine(5.0, 0.05, samplerate=SR), samplerate=SR)
    h_loud, h_soft = loud.fm_contract_horizon(), soft.fm_contract_horizon()
    assert 60.0 < h_loud < 1000.0 < h_soft
    assert abs(loud.fm_error_bound(h_loud) - 1e-6) < 1e-9                # the horizon is where the bound meets the contract
    assert abs(loud.fm_error_bound(2 * h_loud) / loud.fm_error_bound(h_loud) - 2.0 ** 1.5) < 1e-9     # t^1.5: a random walk of growing steps
    assert loud.fm_contract_horizon(1e-5) > h_loud
    saw = G.Sawtooth(880.0, fm_lfo=G.Sine(5.0, 0.1, bias=0.01, samplerate=SR), samplerate=SR)      # a turn-based carrier
    assert 0.0 < saw.fm_error_bound(10.0) < 1e-6


@pytest.mark.gpu
def test_harmonics_int16_boundary_crossings_late_in_a_note_and_exact_mode(gpu):
    """Harmonics x16 through the polynomial form against the reference's term-by-term sum: the float64 values part ways as t grows
    (the reference rounds every t * k), and with them one int16 sample in ~1e6 (10 s in) .. ~1e5 (300 s in) truncates to the neighbour.
    params.exact_harmonics = True: the term-by-term form on the device -- equal integers at any time."""
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import params
    from synthesizer_amd.sample import Sample
    h16 = [(k, 1.0 / k) for k in range(1, 17)]
    n = 1 << 20
    o = O.Harmonics(440.0, h16, 0.5, phase=0.2, samplerate=SR)
    want_all = CO.render(o, 300 * SR + n)
    rates = {}
    for seconds in (10, 300):
        start = seconds * SR
        want = CO.quantise(want_all[start:start + n]).astype(np.int16)
        g = G.Harmonics(440.0, h16, 0.5, phase=0.2, samplerate=SR)
        blk = g._render_f64_device(start, n)
        err = float(np.max(np.abs(blk.download(np.float64, n) - want_all[start:start + n])))
        got = np.frombuffer(Sample.from_osc_device(blk, n, SR).view_frame_data(), dtype=np.int16)
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1
        rates[seconds] = (int(np.count_nonzero(d)), err)
        assert np.count_nonzero(d) <= max(4, int(8.0 * 32767.0 * err * n))           # ~ 2 * scale * |error| crossings per sample, with room
    print("Harmonics x16, polynomial form -> int16: %d of %d samples off by one 10 s in (max float64 error %.1e), %d of %d 300 s in (%.1e)"
          % (rates[10][0], n, rates[10][1], rates[300][0], n, rates[300][1]))
    params.exact_harmonics = True
    try:
        g = G.Harmonics(440.0, h16, 0.5, phase=0.2, samplerate=SR)
        for seconds in (10, 300):
            start = seconds * SR
            m = 1 << 17
            want = CO.quantise(want_all[start:start + m]).astype(np.int16)
            blk = g._render_f64_device(start, m)
            assert float(np.max(np.abs(blk.download(np.float64, m) - want_all[start:start + m]))) <= 1.5e-15        # (sixteen table sines, each within an ulp of libm's)
            got = np.frombuffer(Sample.from_osc_device(blk, m, SR).view_frame_data(), dtype=np.int16)
            assert np.array_equal(got, want), seconds
    finally:
        params.exact_harmonics = False


@pytest.mark.gpu
def test_long_launch_over_several_lfo_piece_ends(gpu):
    """the lean FM loops change LFO-table pieces once per launch (at a 1024-frame boundary); a launch of 2^18 frames under a
    FAST, deep LFO (2 kHz: its accumulated phase crosses a binade every few ten thousand samples) holds several piece ends.  Bank render
    (lean FM loop, float32 bus) and materialised rows against the C oracle."""
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    nv, start, n = 128, 60000, 1 << 18
    rng = np.random.default_rng(11)
    fc = rng.uniform(800.0, 3520.0, nv)
    fl = rng.uniform(1500.0, 2500.0, nv)
    ph = rng.uniform(0.0, 1.0, nv)

    def make(mod):
        return [mod.Sine(float(fc[i]), 0.5, phase=float(ph[i]), fm_lfo=mod.Sine(float(fl[i]), 0.5, samplerate=SR), samplerate=SR) for i in range(nv)]
    gains = [(1.0 / nv, 1.0 / nv)] * nv
    rows = np.stack([CO.render(v, start + n)[start:] for v in make(O)])
    want_bus = CO.mix_bus(rows, gains)
    bank = VoiceBank(make(G), gains=gains)
    got_bus = bank.render(n, start)
    got_rows = bank.generate(n, start)
    per_voice = np.sqrt(np.mean((got_rows.astype(np.float64) - rows) ** 2, axis=1))
    print("fast-LFO bank, 2^18-frame launch from frame %d: bus RMS %.2e, worst voice RMS %.2e (float32 rows: rounding 1.7e-8)"
          % (start, rms(got_bus, want_bus), float(per_voice.max())))
    assert rms(got_bus, want_bus) <= 1e-6 / 3
    assert per_voice.max() <= 1e-6 / 3


@pytest.mark.gpu
def test_lfo_with_bias_minus_one(gpu):
    """the closed form folds the LFO's bias into the carrier's frequency, f (1 + bias), and divides the sine part by it -- a
    bias of -1 (the carrier sweeps through 0 Hz) or next to it takes the buffer path, which divides by nothing."""
    from oracle import synth_oracle as O
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    n = 6000
    for bias in (-1.0, -1.0 + 2.0 ** -12, -0.999):
        o = O.Sine(440.0, 0.8, phase=0.1, fm_lfo=O.Sine(5.0, 1.0, bias=bias, samplerate=SR), samplerate=SR)
        g = G.Sine(440.0, 0.8, phase=0.1, fm_lfo=G.Sine(5.0, 1.0, bias=bias, samplerate=SR), samplerate=SR)
        assert (g.spec().fm_mode == N.SH_FM_BUFFER) == (abs(1.0 + bias) < 2.0 ** -10)
        assert rms(g.render_f64(n), np.array(o.take(n))) <= 1e-9, bias
    gv = [G.Sine(300.0 + 10 * k, 0.5, fm_lfo=G.Sine(4.0 + k, 1.0, bias=-1.0, samplerate=SR), samplerate=SR) for k in range(4)]
    ov = [O.Sine(300.0 + 10 * k, 0.5, fm_lfo=O.Sine(4.0 + k, 1.0, bias=-1.0, samplerate=SR), samplerate=SR) for k in range(4)]
    want = np.array(O.mix_bus([v.take(n) for v in ov], [(0.5, 0.25)] * 4))
    assert rms(VoiceBank(gv, gains=[(0.5, 0.25)] * 4).render(n), want) <= 1e-7
