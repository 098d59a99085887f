"""Voice banks whose voices are modulated by ARBITRARY oscillators (fm_lfo= / pwm_lfo= anything, upstream composes them
freely) or are oscillator graphs themselves (filters, envelopes over filters, nested envelopes): the modulators / sources
are rendered as float64 rows of one matrix per launch (one launch per group of closed-form sources), the fm rows scanned
by one batched launch with the carries kept on the device, and the bank's general code reads its rows
(sh_bank_set_rows / sh_bank_generate_f64 / sh_scan_rows_f64 / sh_bank_render_rows).  Oracle: oracle/synth_oracle.py."""
import numpy as np
import pytest

from oracle import synth_oracle as O
from tests.helpers import rms
from synthesizer_amd.workloads import additive_voices

pytestmark = pytest.mark.gpu
SR = 48000
RMS_TOL = 1e-6


def _voices(m):
    harm = [(1, 1.0), (2, 0.5), (3, 0.25)]
    return [
        m.Sine(440.0, 0.3, fm_lfo=m.Square(3.0, 0.02, samplerate=SR), samplerate=SR),                     # fm row (bank of modulators)
        m.Pulse(220.0, 0.2, pulsewidth=0.3, pwm_lfo=m.Triangle(2.0, 0.2, bias=0.4, samplerate=SR), samplerate=SR),   # pwm row
        m.Sine(330.0, 0.25, fm_lfo=m.Sine(5.0, 0.03, samplerate=SR), samplerate=SR),                      # closed form: no row
        m.Harmonics(110.0, harm, 0.2, samplerate=SR),                                                     # plain
        m.Sawtooth(150.0, 0.2, fm_lfo=m.Sine(2.0, 0.05, fm_lfo=m.Sine(0.5, 0.5, samplerate=SR), samplerate=SR), samplerate=SR),  # modulator is FM itself
        m.Pulse(90.0, 0.2, pulsewidth=0.5, fm_lfo=m.Triangle(1.5, 0.04, samplerate=SR),
                pwm_lfo=m.Sine(0.7, 0.3, bias=0.5, samplerate=SR), samplerate=SR),                        # fm row AND pwm row
        m.Harmonics(70.0, harm, 0.2, fm_lfo=m.Sawtooth(1.0, 0.03, samplerate=SR), samplerate=SR),         # Harmonics with a buffer LFO
        m.Square(55.0, 0.15, fm_lfo=m.Sine(3.0, 0.02, fm_lfo=m.Square(0.7, 0.3, samplerate=SR), samplerate=SR), samplerate=SR),  # modulator needs a row itself
        m.MixingFilter(m.Sine(500.0, 0.1, samplerate=SR), m.Triangle(250.0, 0.1, samplerate=SR)),         # a filter graph as a voice
        m.EnvelopeFilter(m.MixingFilter(m.Sine(600.0, 0.2, samplerate=SR), m.Square(300.0, 0.1, samplerate=SR)), 0.01, 0.02, 0.03, 0.6, 0.02),
        m.EnvelopeFilter(m.EnvelopeFilter(m.Sine(700.0, 0.3, samplerate=SR), 0.0, 0.01, 0.05, 0.8, 0.01), 0.02, 0.0, 0.04, 1.0, 0.03),  # nested
        m.AmpModulationFilter(m.Sawtooth(800.0, 0.2, samplerate=SR), m.Sine(4.0, 0.5, bias=0.5, samplerate=SR)),
        m.EnvelopeFilter(m.Sine(900.0, 0.2, samplerate=SR), 0.01, 0.01, 0.02, 0.5, 0.02),                 # the usual fused envelope
    ]


def _oracle_bus(ov, gains, n):
    return np.array(O.mix_bus([v.take(n) for v in ov], gains), dtype=np.float64)


def test_bank_with_arbitrary_modulators_and_filter_voices(gpu):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    gv, ov = _voices(G), _voices(O)
    rng = np.random.default_rng(5)
    gains = [(float(np.float32(a)), float(np.float32(b))) for a, b in rng.uniform(0.2, 1.0, (len(gv), 2))]
    bank = VoiceBank(gv, gains=gains)
    n = 7000                                           # crosses three 2048-value scan tiles
    want = _oracle_bus(ov, gains, 2 * n + 100)
    got = bank.render(n)
    assert got.shape == (n, 2) and rms(got, want[:n]) <= RMS_TOL
    assert np.max(np.abs(got - want[:n])) < 5e-7
    # the next block continues the running sums from the carries kept on the device
    nxt = bank.render(n, start=n)
    assert rms(nxt, want[n:2 * n]) <= RMS_TOL
    # random access: back to an earlier position (replay from 0), a jump ahead, an odd short block
    again = bank.render(1000, start=2500)
    assert rms(again, want[2500:3500]) <= RMS_TOL
    ahead = bank.render(100, start=2 * n)
    assert rms(ahead, want[2 * n:2 * n + 100]) <= RMS_TOL
    # int16 PCM of such a bank: the float32 bus through the saturating quantiser
    pcm = bank.render_sample(2000).get_frames_numpy()
    ref = np.clip(np.trunc(32767.0 * bank.render(2000).astype(np.float64)), -32768, 32767).astype(np.int16)
    assert np.array_equal(pcm, ref)
    # the reference-shaped two-step route for such a bank: every voice as a float32 row (the general materialisation kernel reads
    # the same matrix of rows), then the HBM-bound mix -- rows equal the voices rendered one by one, the mix equals the fused bus
    rows = bank.generate(n)
    assert rows.shape == (len(gv), n)
    for i, v in enumerate(_voices(G)):
        one = v.render(n, start=0)
        assert np.max(np.abs(rows[i] - one)) < 1.5e-7, i
    two = bank.render_two_step(n)
    assert rms(two, want[:n]) <= RMS_TOL
    late = bank.generate(500, start=n)                # rows that start inside the running sums
    for i, ov_ in enumerate(_voices(O)):
        assert np.max(np.abs(late[i] - np.array(ov_.take(n + 500)[n:], dtype=np.float32))) < 2e-7, i


def test_voices_that_end_inside_a_bank_go_silent(gpu):
    """A voice that is no single record (an envelope with stop_at_end over a filter graph) and ends: from there on it is
    silent, like a fused stop_at_end voice beside it -- a block that straddles its end is zero-padded, a block past it renders
    (such a bank could not be rendered past that voice's end)."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank

    def build(m):
        return [m.EnvelopeFilter(m.MixingFilter(m.Sine(600.0, 0.2, samplerate=SR), m.Square(300.0, 0.1, samplerate=SR)),
                                 0.01, 0.02, 0.03, 0.6, 0.02, stop_at_end=True),            # 3841 samples, rendered into a row
                m.EnvelopeFilter(m.Sine(440.0, 0.3, samplerate=SR), 0.01, 0.01, 0.02, 0.5, 0.02, stop_at_end=True),   # fused, 2881 samples
                m.Sine(220.0, 0.2, samplerate=SR)]
    gv, ov = build(G), build(O)
    gains = [(0.5, 0.25), (0.25, 0.5), (0.5, 0.5)]
    bank = VoiceBank(gv, gains=gains)
    n = 6000
    rows = []
    for v in ov:
        x = v.take(n)
        rows.append(list(x) + [0.0] * (n - len(x)))
    want = np.array(O.mix_bus(rows, gains), dtype=np.float64)
    assert len(ov[0].take(n)) < 4000 < n
    assert rms(bank.render(n), want) <= RMS_TOL                       # one block over both ends
    assert rms(bank.render(1000, start=3500), want[3500:4500]) <= RMS_TOL     # straddles the row voice's end
    assert rms(bank.render(1000, start=5000), want[5000:]) <= RMS_TOL         # past it


def test_large_bank_with_modulated_voices_among_lean_ones(gpu):
    """192 voices (several voice groups, the split launch): mostly lean additive voices, every eighth voice modulated by a
    non-Sine oscillator -- the general-lists kernel reads the rows while the lean kernel does the rest."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank

    def build(m):
        v, g = additive_voices(m, 192, SR, seed=11, partials=6, adsr={"sustain": 100.0})
        for i in range(0, 192, 8):
            f = 100.0 + 7.0 * i
            if i % 16:
                v[i] = m.Sine(f, 0.02, fm_lfo=m.Triangle(0.5 + 0.01 * i, 0.03, samplerate=SR), samplerate=SR)
            else:
                v[i] = m.Pulse(f, 0.02, pulsewidth=0.4, pwm_lfo=m.Sawtooth(0.3 + 0.01 * i, 0.2, bias=0.5, samplerate=SR), samplerate=SR)
        return v, g
    gv, gains = build(G)
    ov, _ = build(O)
    bank = VoiceBank(gv, gains=gains)
    n = 3000
    want = _oracle_bus(ov, gains, 2 * n)
    assert rms(bank.render(n), want[:n]) <= RMS_TOL
    assert rms(bank.render(n, start=n), want[n:]) <= RMS_TOL
    # long block: eight frames per lane in the lean kernel, rows indexed by frame in the general one
    big = bank.render(20000)
    assert rms(big[:2 * n], want) <= RMS_TOL


def test_envelope_over_filters_and_nested_envelopes_as_oscillators(gpu):
    from synthesizer_amd import oscillators as G
    cases = [
        lambda m: m.EnvelopeFilter(m.MixingFilter(m.Sine(440.0, 0.4, samplerate=SR), m.Square(220.0, 0.3, samplerate=SR)), 0.01, 0.02, 0.03, 0.6, 0.02),
        lambda m: m.EnvelopeFilter(m.EnvelopeFilter(m.Sawtooth(330.0, 0.5, samplerate=SR), 0.005, 0.01, 0.05, 0.7, 0.01), 0.02, 0.0, 0.03, 1.0, 0.02),
        lambda m: m.EnvelopeFilter(m.EchoFilter(m.EnvelopeFilter(m.Sine(500.0, samplerate=SR), 0.0, 0.01, 0.01, 0.5, 0.01, stop_at_end=True), 0.01, 2, 0.02, 0.5),
                                   0.0, 0.01, 0.01, 0.5, 0.005, stop_at_end=True),      # shorter than its (finite) source
        lambda m: m.EnvelopeFilter(m.AbsFilter(m.Sine(100.0, samplerate=SR)), 0.0, 0.0, 0.02, 1.0, 0.01, stop_at_end=True),
    ]
    for k, make in enumerate(cases):
        g, o = make(G), make(O)
        n = 6000
        want = np.array(o.take(n), dtype=np.float64)
        got = g.render_f64(n, start=0)
        assert len(got) == len(want), k
        assert np.max(np.abs(got - want)) <= 1e-12, k
        assert g.length == (len(want) if len(want) < n else g.length), k
        # as someone's modulator and through the blocks() protocol
        blk_g, blk_o = next(g.blocks()), next(o.blocks())
        assert np.max(np.abs(np.array(blk_g) - np.array(blk_o))) <= 1e-12, k


def test_envelope_over_a_source_that_ends(gpu):
    """Upstream's EnvelopeFilter pulls one source sample per envelope sample up to the end of its release (+ the extra sample)
    and none after.  A finite source that outlasts those phases does not end the stream (silence follows unless stop_at_end);
    one that ends inside them ends the generator with next()'s StopIteration -- RuntimeError for the consumer (PEP 479), after
    every complete block before the missing sample has been delivered."""
    import itertools
    from synthesizer_amd import oscillators as G
    inner = lambda m, sustain: m.EnvelopeFilter(m.Sine(300.0, 0.7, samplerate=SR), 0.0, 0.01, sustain, 0.8, 0.01, stop_at_end=True)
    # the source (0.12 s) outlasts the outer phases (0.06 s): the stream goes on as silence
    make = lambda m: m.EnvelopeFilter(inner(m, 0.1), 0.01, 0.0, 0.03, 1.0, 0.02)
    g, o = make(G), make(O)
    n = 8000                                              # well past the end of the inner stream
    want = np.array(o.take(n), dtype=np.float64)
    got = g.render_f64(n, start=0)
    assert g.length is None and len(got) == len(want) == n
    assert np.max(np.abs(got - want)) <= 1e-12 and not got[int(0.061 * SR):].any() and got[100:200].any()
    assert np.array_equal(g.render_f64(500, start=7000), np.zeros(500))           # random access beyond the source's end
    # the source (0.03 s) ends inside the outer phases (0.12 s)
    make = lambda m: m.EnvelopeFilter(inner(m, 0.01), 0.01, 0.0, 0.1, 1.0, 0.01)
    g, o = make(G), make(O)
    src_len = len(inner(O, 0.01).take(100000))
    assert inner(G, 0.01).length == src_len
    with pytest.raises(RuntimeError):
        o.take(src_len + 1)
    with pytest.raises(RuntimeError):
        g.render_f64(src_len + 1, start=0)
    upto = g.render_f64(src_len, start=0)                                         # up to the last sample the source has
    whole = src_len // 512 * 512                                                  # (the oracle's take() pulls whole blocks)
    assert len(upto) == src_len and np.max(np.abs(upto[:whole] - np.array(make(O).take(whole)))) <= 1e-12
    # blocks(): the same whole blocks, then the same exception
    def drain(osc):
        out, it = [], osc.blocks()
        try:
            for blk in it:
                out.append(blk)
        except RuntimeError:
            return out, True
        return out, False
    bg, raised_g = drain(make(G))
    bo, raised_o = drain(make(O))
    assert raised_g and raised_o and len(bg) == len(bo) == src_len // 512
    assert np.max(np.abs(np.array(list(itertools.chain.from_iterable(bg))) - np.array(list(itertools.chain.from_iterable(bo))))) <= 1e-12
    # as a voice of a bank: the same error when the launch needs the missing sample, none before
    from synthesizer_amd.mixer import VoiceBank
    bank = VoiceBank([make(G), G.Sine(100.0, 0.1, samplerate=SR)], gains=[(1.0, 0.0), (0.0, 1.0)])
    assert bank.render(src_len, start=0)[:, 0].any()
    with pytest.raises(RuntimeError):
        bank.render(src_len + 10, start=0)


def test_buffer_path_fm_sums_over_the_accumulated_time_steps(gpu):
    """Round 4: the angle of an FM carrier is the sum of freq_j over the ACCUMULATED time steps; a modulator that is summed by
    a scan is weighted with (t_{j+1} - t_j) / inc first (oscillators.time_step_weights).  Twenty seconds into the notes a bank whose fm
    rows belong to carriers of both kinds of time step (Sine: 2 pi / sr; Sawtooth: 1 / sr) under biased non-Sine LFOs, whose block holds an
    end of a time table's piece, against the pure-Python oracle: without the weights the biased rows are 1e-7 off by now (a t^2 law)."""
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.oscillators import time_step_weights

    def voices(m):
        return [m.Sine(3520.0, 0.3, phase=0.2, fm_lfo=m.Triangle(5.0, 0.05, bias=0.01, samplerate=SR), samplerate=SR),
                m.Sawtooth(1760.0, 0.3, fm_lfo=m.Square(3.0, 0.02, bias=0.02, samplerate=SR), samplerate=SR),
                m.Sine(2000.0, 0.3, fm_lfo=m.Sawtooth(2.0, 0.3, samplerate=SR), samplerate=SR),
                m.Harmonics(880.0, [(1, 1.0), (2, 0.5)], 0.3, fm_lfo=m.Triangle(0.5, 0.2, bias=-0.01, samplerate=SR), samplerate=SR)]
    gains = [(0.5, 0.25), (0.25, 0.5), (0.4, 0.4), (0.3, 0.6)]
    # a block that holds the end of a piece of the time table t += 2 pi / sr (the accumulated time crosses 128 rad at sample 977 848)
    ends = [int(n) for n in G._table(0.0, 2 * np.pi / SR).records["n0"] if 900000 < int(n) < 1100000]
    assert ends
    first, n = ends[0] - 5000, 12000
    assert len(time_step_weights(2 * np.pi / SR, first, n)) >= 3
    bank = VoiceBank(voices(G), gains=gains)
    want = _oracle_bus(voices(O), gains, first + n)[first:]
    got = bank.render(n, start=first)
    assert rms(got, want) <= 2e-8, float(rms(got, want))              # (float32 rounding of the bus: ~1e-8)
    again = bank.render(4096, start=first + n)                        # the carries go on from there
    more = _oracle_bus(voices(O), gains, first + n + 4096)[first + n:]
    assert rms(again, more) <= 2e-8
