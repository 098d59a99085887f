"""GPU parity, SURVEY.md section 8(f) item 2 (continued): the editing operations of Sample -- clip / split / join /
add_silence / delay, speed, at_volume, echo, envelope, modulate_amp -- against oracle/sample_oracle.py, which
restates upstream's methods over bytes slicing and the live ``audioop`` module.  Bit-exact."""
import math

import numpy as np
import pytest

from oracle.sample_oracle import RefSample

pytestmark = pytest.mark.gpu
DT = {1: np.int8, 2: np.int16, 4: np.int32}


def _rand(rng, width, n, scale=1.0):
    info = np.iinfo(DT[width])
    x = (rng.integers(info.min, info.max + 1, n, dtype=np.int64) * scale).astype(DT[width])
    return x


def _pair(x, width, rate, nch):
    from synthesizer_amd.sample import Sample
    return Sample.from_raw_frames(x.tobytes(), width, rate, nch), RefSample(x.tobytes(), width, rate, nch)


def _same(s, r):
    assert (s.samplewidth, s.samplerate, s.nchannels) == (r.samplewidth, r.samplerate, r.nchannels)
    assert bytes(s.view_frame_data()) == r.frames


@pytest.mark.parametrize("width,nch", [(2, 1), (2, 2), (1, 2), (4, 1)])
def test_clip_split_join_silence_delay(gpu, width, nch):
    rng = np.random.default_rng(width * 10 + nch)
    rate = 8000
    x = _rand(rng, width, 8000 * nch)             # 1 s
    s, r = _pair(x, width, rate, nch)
    _same(s.clip(0.1, 0.9), r.clip(0.1, 0.9))
    rest_s, rest_r = s.split(0.5), r.split(0.5)
    _same(s, r)
    _same(rest_s, rest_r)
    _same(s.add_silence(0.0125), r.add_silence(0.0125))
    _same(s.add_silence(0.02, at_start=True), r.add_silence(0.02, at_start=True))
    _same(s.join(rest_s), r.join(rest_r))
    _same(s.delay(0.03), r.delay(0.03))
    _same(s.delay(0.05, keep_length=True), r.delay(0.05, keep_length=True))
    _same(s.delay(-0.04), r.delay(-0.04))
    _same(s.delay(-0.02, keep_length=True), r.delay(-0.02, keep_length=True))
    # split beyond the end and at zero; clip to nothing
    e_s, e_r = s.split(10.0), r.split(10.0)
    _same(e_s, e_r)
    assert len(e_s) == 0
    z_s, z_r = s.copy().split(0.0), r.copy().split(0.0)
    _same(z_s, z_r)
    _same(s.copy().clip(0.2, 0.2), r.copy().clip(0.2, 0.2))
    # joining an empty sample changes nothing
    _same(s.join(e_s), r.join(e_r))


@pytest.mark.parametrize("width,nch", [(2, 1), (2, 2), (1, 1), (4, 2)])
@pytest.mark.parametrize("speed", [0.5, 0.7937, 1.0, 1.25992, 2.0, 9.5])
def test_speed(gpu, width, nch, speed):
    rng = np.random.default_rng(int(speed * 1000) + width)
    x = _rand(rng, width, 20011 * nch)
    s, r = _pair(x, width, 44100, nch)
    _same(s.speed(speed), r.speed(speed))
    with pytest.raises(ValueError):
        s.speed(11.0)
    with pytest.raises(ValueError):
        s.speed(0.05)


def test_at_volume_leaves_original(gpu):
    rng = np.random.default_rng(5)
    x = _rand(rng, 2, 5000)
    s, r = _pair(x, 2, 22050, 1)
    s.lock()
    _same(s.at_volume(0.3), r.at_volume(0.3))
    _same(s, r)
    with pytest.raises(RuntimeError):
        s.amplify(0.3)


@pytest.mark.parametrize("width,nch", [(2, 1), (2, 2), (1, 1), (4, 1)])
def test_echo(gpu, width, nch):
    rng = np.random.default_rng(width + nch)
    x = _rand(rng, width, 16000 * nch, scale=0.4)      # 2 s at 8 kHz
    for length, amount, delay, decay in ((0.5, 4, 0.3, 0.6), (1.0, 3, 0.05, 0.5), (0.2, 0, 0.1, 0.5),
                                         (5.0, 2, 0.7, 0.9), (0.3, 40, 0.01, 0.5), (0.25, 3, 0.125, 1.2)):
        s, r = _pair(x, width, 8000, nch)
        _same(s.echo(length, amount, delay, decay), r.echo(length, amount, delay, decay))


@pytest.mark.parametrize("width,nch", [(2, 1), (2, 2), (1, 1), (4, 1)])
def test_envelope(gpu, width, nch):
    rng = np.random.default_rng(width * 3 + nch)
    x = _rand(rng, width, 8000 * nch)
    for a, d, sl, rel in ((0.1, 0.2, 0.5, 0.3), (0.0, 0.1, 0.8, 0.1), (0.05, 0.0, 1.0, 0.0), (0.3, 0.3, 0.0, 0.3),
                          (0.01, 0.02, 0.25, 0.9), (0.4, 0.4, 0.5, 0.4)):
        s, r = _pair(x, width, 8000, nch)
        _same(s.envelope(a, d, sl, rel), r.envelope(a, d, sl, rel))


@pytest.mark.parametrize("width", [2, 1, 4])
def test_modulate_amp_sample_and_sequences(gpu, width):
    rng = np.random.default_rng(width)
    x = _rand(rng, width, 30011)
    # another sample as the modulator: cycled, scaled to a peak of 1.0 (includes the most negative value)
    m = _rand(rng, 2, 777)
    m[3] = -32768
    s, r = _pair(x, width, 8000, 1)
    ms, mr = _pair(m, 2, 8000, 1)
    _same(s.modulate_amp(ms), r.modulate_amp(mr))
    # a modulator longer than the sample
    m2 = _rand(rng, 1, 50000)
    s, r = _pair(x, width, 8000, 1)
    ms, mr = _pair(m2, 1, 8000, 1)
    _same(s.modulate_amp(ms), r.modulate_amp(mr))
    # a list of numbers (normalised and cycled) and a plain iterator of factors (used as they are)
    seq = [0.5, 2.0, -1.0, 0.25, 0.0]
    s, r = _pair(x, width, 8000, 1)
    _same(s.modulate_amp(seq), r.modulate_amp(list(seq)))
    fac = rng.uniform(-1, 1, len(x)).tolist()
    s, r = _pair(x, width, 8000, 1)
    _same(s.modulate_amp(iter(fac)), r.modulate_amp(iter(fac)))
    # empty sample: nothing happens
    s, r = _pair(x[:0], width, 8000, 1)
    _same(s.modulate_amp(seq), r)


def test_modulate_amp_oscillator_and_overflow(gpu):
    import itertools
    from oracle import synth_oracle as O
    from synthesizer_amd.oscillators import Sine
    rng = np.random.default_rng(9)
    x = _rand(rng, 2, 12000)
    s, r = _pair(x, 2, 8000, 1)
    osc = Sine(3.0, amplitude=0.8, bias=0.1, samplerate=8000)
    ref_osc = O.Sine(3.0, amplitude=0.8, bias=0.1, samplerate=8000)
    s.modulate_amp(osc)
    r.modulate_amp(itertools.chain.from_iterable(ref_osc.blocks()))
    got = np.frombuffer(bytes(s.view_frame_data()), dtype=np.int16).astype(np.int64)
    want = np.frombuffer(r.frames, dtype=np.int16).astype(np.int64)
    # the modulator is a float oscillator (device sin vs libm sin agree to ~1 ulp): a product within 1e-9 of an
    # integer may truncate either way
    assert np.abs(got - want).max() <= 1
    assert (got != want).mean() < 1e-4
    # a factor that pushes a sample out of range raises, as the array store does upstream
    s, r = _pair(np.array([30000, -30000], dtype=np.int16), 2, 8000, 1)
    with pytest.raises(OverflowError):
        s.modulate_amp(iter([1.5, 1.5]))
    with pytest.raises(OverflowError):
        r.modulate_amp(iter([1.5, 1.5]))


def test_fuzz_random_chains_of_sample_operations(gpu):
    """Random chains of Sample operations (arithmetic, editing, resampling, channel conversions) on random PCM, against
    the same chain on RefSample / the live audioop -- bit-exact after every step."""
    import audioop
    import os
    from synthesizer_amd.sample import Sample

    class Ref(RefSample):
        def bias(self, b):
            self.frames = audioop.bias(self.frames, self.samplewidth, b)
            return self

        def reverse(self):
            self.frames = audioop.reverse(self.frames, self.samplewidth)
            return self

        def mono(self, lf=1.0, rf=1.0):
            if self.nchannels == 2:
                self.frames = audioop.tomono(self.frames, self.samplewidth, lf, rf)
                self.nchannels = 1
            return self

        def stereo(self, lf=1.0, rf=1.0):
            if self.nchannels == 1:
                self.frames = audioop.tostereo(self.frames, self.samplewidth, lf, rf)
                self.nchannels = 2
            return self

    rng = np.random.default_rng(int(os.environ.get("SYNTHHIP_FUZZ_SEED", "77")))
    for case in range(12):
        width = int(rng.choice([1, 2, 2, 4]))
        nch = int(rng.choice([1, 2]))
        rate = int(rng.choice([8000, 11025, 22050]))
        frames = int(rng.integers(200, 6000))
        x = _rand(rng, width, frames * nch, scale=0.3)
        s = Sample.from_raw_frames(x.tobytes(), width, rate, nch)
        r = Ref(x.tobytes(), width, rate, nch)
        for step in range(8):
            op = int(rng.integers(0, 17))
            dur = r.duration
            if op == 0:
                f = float(rng.uniform(-1.2, 1.2)); s.amplify(f); r.amplify(f)
            elif op == 1:
                b = int(rng.integers(-50, 50)); s.bias(b); r.bias(b)
            elif op == 2:
                s.reverse(); r.reverse()
            elif op == 3:
                t = float(rng.uniform(0, dur)); v = float(rng.uniform(0, 0.5)); s.fadeout(t, v); r.fadeout(t, v)
            elif op == 4:
                t = float(rng.uniform(0, dur)); v = float(rng.uniform(0, 0.5)); s.fadein(t, v); r.fadein(t, v)
            elif op == 5:
                y = _rand(rng, width, int(rng.integers(1, 3000)) * r.nchannels, scale=0.3)
                at = float(rng.uniform(0, dur * 1.2))
                s.mix_at(at, Sample.from_raw_frames(y.tobytes(), width, r.samplerate, r.nchannels))
                r.mix_at(at, Ref(y.tobytes(), width, r.samplerate, r.nchannels))
            elif op == 6:
                a = float(rng.uniform(0, dur)); b = float(rng.uniform(a, dur)); s.clip(a, b); r.clip(a, b)
            elif op == 7:
                t = float(rng.uniform(-0.2, 0.2) * dur); k = bool(rng.integers(0, 2)); s.delay(t, k); r.delay(t, k)
            elif op == 8 and len(r) > 10:
                sp = float(rng.choice([0.5, 0.8, 1.25, 2.0])); s.speed(sp); r.speed(sp)
            elif op == 9 and len(r) > 10:
                nr = int(rng.choice([8000, 11025, 16000, 22050, 44100])); s.resample(nr); r.resample(nr)
            elif op == 10:
                lf, rf = float(rng.uniform(0, 1)), float(rng.uniform(0, 1))
                if r.nchannels == 2:
                    s.mono(lf, rf); r.mono(lf, rf)
                else:
                    s.stereo(lf, rf); r.stereo(lf, rf)
            elif op == 11 and dur > 0:
                a = (float(rng.uniform(0, dur * 0.8)), int(rng.integers(0, 4)), float(rng.uniform(0.001, 0.05)), float(rng.uniform(0.2, 0.9)))
                s.echo(*a); r.echo(*a)
            elif op == 12 and dur > 0.05:
                a = (dur * 0.1, dur * 0.2, float(rng.uniform(0, 1)), dur * 0.3)
                s.envelope(*a); r.envelope(*a)
            elif op == 13:
                pn = float(rng.uniform(-1, 1)); s.pan(pn); r.pan(pn)
            elif op == 14 and len(r) > 0:
                pos = rng.uniform(-1, 1, len(r)).tolist(); s.pan(lfo=iter(pos)); r.pan(lfo=iter(pos))
            elif op == 15:
                y = _rand(rng, width, int(rng.integers(1, 3000)), scale=0.3)
                a = (str(rng.choice(["L", "R"])), float(rng.uniform(0, 1.5)), float(rng.uniform(0, dur)))
                s.stereo_mix(Sample.from_raw_frames(y.tobytes(), width, r.samplerate, 1), *a)
                r.stereo_mix(Ref(y.tobytes(), width, r.samplerate, 1), *a)
            elif op == 16 and r.nchannels <= 2:
                assert s.level_db_peak == r.level_db_peak and s.level_db_peak_mono == r.level_db_peak_mono
                if width < 4:
                    assert s.level_db_rms == r.level_db_rms
            assert (s.samplewidth, s.samplerate, s.nchannels) == (r.samplewidth, r.samplerate, r.nchannels), (case, step, op)
            assert bytes(s.view_frame_data()) == r.frames, (case, step, op, width, nch, len(r))


@pytest.mark.parametrize("width", [1, 2, 4])
@pytest.mark.parametrize("nch", [1, 2])
def test_level_db_matches_audioop(gpu, width, nch):
    """level_db_peak / level_db_rms: the per-channel maxima and root mean squares come from one device pass over the
    interleaved PCM; upstream takes them from audioop.tomono copies.  Same floats, exactly (integer statistics; the
    width-4 sum of squares is float64 in a different order, so the truncated RMS may differ by one unit there)."""
    rng = np.random.default_rng(width * 7 + nch)
    for n, scale in ((1, 1.0), (7, 1.0), (1000, 0.01), (50021, 0.5), (262144 + 3, 1.0)):
        x = _rand(rng, width, n * nch, scale)
        if nch == 2:
            x[1::2] = (x[1::2] * 0.25).astype(DT[width])          # channels at different levels
        s, r = _pair(x, width, 8000, nch)
        assert s.level_db_peak == r.level_db_peak
        assert s.level_db_peak_mono == r.level_db_peak_mono
        if width < 4:
            assert s.level_db_rms == r.level_db_rms
            assert s.level_db_rms_mono == r.level_db_rms_mono
        else:
            assert np.allclose(s.level_db_rms, r.level_db_rms, rtol=0, atol=1e-6)
            assert abs(s.level_db_rms_mono - r.level_db_rms_mono) < 1e-6
    # extremes: silence bottoms out at -60 (8-bit: one count of 128 = -42 dB); full-scale negative is above 0 dB by one count
    z, rz = _pair(np.zeros(64 * nch, DT[width]), width, 8000, nch)
    assert z.level_db_peak == rz.level_db_peak == ((-60.0, -60.0) if width > 1 else (20.0 * math.log(1 / 128, 10),) * 2)
    lo = np.full(64 * nch, np.iinfo(DT[width]).min, DT[width])
    s, r = _pair(lo, width, 8000, nch)
    assert s.level_db_peak == r.level_db_peak and s.level_db_peak[0] > 0.0
    if width < 4:
        assert s.level_db_rms == r.level_db_rms
    e, re_ = _pair(np.zeros(0, DT[width]), width, 8000, nch)
    assert e.level_db_peak == re_.level_db_peak and e.level_db_rms == re_.level_db_rms


def test_level_db_unaligned_device_views(gpu):
    """Stereo statistics of a sample whose device storage does not start on a 16-byte boundary (after clip)."""
    rng = np.random.default_rng(99)
    x = _rand(rng, 2, 2 * 30011)
    s, r = _pair(x, 2, 8000, 2)
    s.to_device()
    _same(s.clip(0.000125, 3.5), r.clip(0.000125, 3.5))          # drops one frame: 4-byte offset
    assert s.level_db_peak == r.level_db_peak and s.level_db_rms == r.level_db_rms


def test_level_meter_tracks_like_upstream(gpu):
    """LevelMeter over a run of chunks with rising and falling level: same (level, peak) sequence as the oracle's
    restatement over audioop, for both modes -- the hold (0.4 s) and fall (30 dB/s) logic is host arithmetic."""
    from oracle.sample_oracle import RefLevelMeter
    from synthesizer_amd.sample import LevelMeter
    rng = np.random.default_rng(5)
    rate, chunk = 8000, 800                                       # 0.1 s chunks
    for rms_mode in (False, True):
        meter, ref = LevelMeter(rms_mode=rms_mode), RefLevelMeter(rms_mode=rms_mode)
        for k in range(30):
            level = (0.9, 0.5, 0.02, 0.001, 0.0, 0.3)[k % 6] * (1.0 if k < 12 else 0.1)
            x = _rand(rng, 2, 2 * chunk, level)
            x[1::2] = (x[1::2] * 0.5).astype(np.int16)
            s, r = _pair(x, 2, rate, 2)
            assert meter.update(s) == ref.update(r)
        assert meter.peak_left >= meter.level_left and meter.peak_right >= meter.level_right
        meter.reset()
        assert (meter.peak_left, meter.level_right) == (-60.0, -60.0)
    with pytest.raises(ValueError):
        from synthesizer_amd.sample import Sample
        Sample.from_raw_frames(bytes(12), 2, rate, 3).level_db_peak


@pytest.mark.parametrize("width", [1, 2, 4])
@pytest.mark.parametrize("nch", [1, 2])
def test_pan_with_lfo(gpu, width, nch):
    """Sample.pan(lfo=...): per-frame position from an iterable or an oscillator; both channels of a stereo source are
    kept apart.  Bit-exact against the oracle's per-sample Python loop; plain pan() on the same sources as well."""
    import itertools
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    rng = np.random.default_rng(width + 10 * nch)
    n = 3001
    x = _rand(rng, width, n * nch)
    for p in (-1.0, -0.3, 0.0, 0.5, 1.0):
        s, r = _pair(x, width, 8000, nch)
        _same(s.pan(p), r.pan(p))
    pos = rng.uniform(-1, 1, n)
    pos[:4] = (-1.0, 1.0, 0.0, 0.999999)
    s, r = _pair(x, width, 8000, nch)
    _same(s.pan(lfo=iter(pos.tolist())), r.pan(lfo=iter(pos.tolist())))
    assert s.nchannels == 2 and len(s) == n
    s, r = _pair(x, width, 8000, nch)
    lfo = G.Sine(3.0, amplitude=0.8, samplerate=8000)
    ref_lfo = O.Sine(3.0, amplitude=0.8, samplerate=8000)
    _same(s.pan(lfo=lfo), r.pan(lfo=itertools.chain.from_iterable(ref_lfo.blocks())))
    # positions outside [-1, 1] can leave the sample range: Python raises, so does the device path
    s, r = _pair(np.full(8 * nch, np.iinfo(DT[width]).max, DT[width]), width, 8000, nch)
    with pytest.raises(OverflowError):
        r.pan(lfo=iter([-1.5] * 8))
    with pytest.raises(OverflowError):
        s.pan(lfo=iter([-1.5] * 8))
    with pytest.raises(ValueError):
        _pair(x, width, 8000, nch)[0].pan(lfo=iter([0.0] * 5))     # ran out
    e, re_ = _pair(np.zeros(0, DT[width]), width, 8000, nch)
    _same(e.pan(lfo=iter([])), re_.pan(lfo=iter([])))


@pytest.mark.parametrize("width", [1, 2, 4])
def test_stereo_mix(gpu, width):
    """Sample.stereo_mix: a mono sample into the left or right channel of a mono or stereo one, scaled, at an offset."""
    rng = np.random.default_rng(width)
    base_m, base_s, other = _rand(rng, width, 4000), _rand(rng, width, 2 * 4000), _rand(rng, width, 2500)
    for base, nch in ((base_m, 1), (base_s, 2)):
        for channel, factor, at, secs in (("L", 1.0, 0.0, None), ("R", 0.5, 0.1, None), ("L", 2.0, 0.4, 0.2), ("R", -1.0, 0.0, 0.05)):
            s, r = _pair(base, width, 8000, nch)
            o, ro = _pair(other, width, 8000, 1)
            _same(s.stereo_mix(o, channel, factor, at, secs), r.stereo_mix(ro, channel, factor, at, secs))
            _same(o, ro)                                             # the mixed-in sample is left alone
    with pytest.raises(AssertionError):
        _pair(base_m, width, 8000, 1)[0].stereo_mix(_pair(base_s, width, 8000, 2)[0], "L")
