"""GPU parity, SURVEY.md section 8(f) item 2: the elementwise Sample operations upstream implements with
audioop (mul, bias, reverse, tomono, tostereo, lin2lin, max, rms) -- bit-exact against the live module --
and the fade ramps against the oracle's restatement of upstream's per-sample expression."""
import audioop

import numpy as np
import pytest

from oracle import pcm_oracle as P

pytestmark = pytest.mark.gpu
DT = {1: np.int8, 2: np.int16, 4: np.int32}


def _rand(rng, width, n):
    info = np.iinfo(DT[width])
    x = rng.integers(info.min, info.max + 1, n, dtype=np.int64).astype(DT[width])
    k = min(n, 8)
    x[:k] = np.array([info.max, info.min, info.min + 1, info.max - 1, 0, -1, 3, -3], dtype=DT[width])[:k]
    return x


def _sample(arr, width, rate, nch):
    from synthesizer_amd.sample import Sample
    return Sample.from_raw_frames(arr.tobytes(), width, rate, nch)


def _bytes(s):
    return bytes(s.view_frame_data())


@pytest.mark.parametrize("width", [1, 2, 4])
def test_amplify_bias_reverse(gpu, width):
    rng = np.random.default_rng(width)
    for n in (0, 1, 7, 4096, 100003):
        x = _rand(rng, width, n)
        raw = x.tobytes()
        for factor in (1.5, 0.5, -1.0, 1.00001, 0.0, -0.333):
            assert _bytes(_sample(x, width, 8000, 1).amplify(factor)) == audioop.mul(raw, width, factor)
        assert _bytes(_sample(x, width, 8000, 1).invert()) == audioop.mul(raw, width, -1)
        for b in (1, -1, 12345):
            assert _bytes(_sample(x, width, 8000, 1).bias(b)) == audioop.bias(raw, width, b)
        assert _bytes(_sample(x, width, 8000, 1).reverse()) == audioop.reverse(raw, width)


@pytest.mark.parametrize("width", [1, 2, 4])
def test_mono_stereo_pan(gpu, width):
    rng = np.random.default_rng(10 + width)
    for frames in (1, 5, 3000):
        x = _rand(rng, width, frames * 2)
        raw = x.tobytes()
        for lf, rf in ((1.0, 1.0), (0.5, 0.25), (-1.0, 0.7)):
            s = _sample(x, width, 8000, 2).mono(lf, rf)
            assert s.nchannels == 1 and _bytes(s) == audioop.tomono(raw, width, lf, rf)
        assert _bytes(_sample(x, width, 8000, 2).left()) == audioop.tomono(raw, width, 1.0, 0)
        assert _bytes(_sample(x, width, 8000, 2).right()) == audioop.tomono(raw, width, 0, 1.0)
        m = _rand(rng, width, frames)
        for lf, rf in ((1.0, 1.0), (0.5, 0.25), (2.0, -1.0)):
            s = _sample(m, width, 8000, 1).stereo(lf, rf)
            assert s.nchannels == 2 and _bytes(s) == audioop.tostereo(m.tobytes(), width, lf, rf)
        s = _sample(m, width, 8000, 1).pan(0.5)
        assert _bytes(s) == audioop.tostereo(m.tobytes(), width, 0.25, 0.75)
        # stereo source: each channel scaled, (fbound(L*lf), fbound(R*rf))
        s = _sample(x, width, 8000, 2).stereo(0.5, 0.25)
        l = audioop.mul(audioop.tomono(raw, width, 1.0, 0), width, 0.5)
        r = audioop.mul(audioop.tomono(raw, width, 0, 1.0), width, 0.25)
        want = audioop.add(audioop.tostereo(l, width, 1.0, 0), audioop.tostereo(r, width, 0, 1.0), width)
        assert _bytes(s) == want and s.nchannels == 2


def test_width_conversions_and_normalize(gpu):
    from synthesizer_amd import params
    rng = np.random.default_rng(3)
    x = _rand(rng, 4, 5000)
    s = _sample(x, 4, 22050, 1)
    ref = audioop.ratecv(x.tobytes(), 4, 1, 22050, params.norm_samplerate, None)[0]
    ref = audioop.lin2lin(ref, 4, 2)
    ref = audioop.tostereo(ref, 2, 1, 1)
    s.normalize()
    assert (s.samplerate, s.samplewidth, s.nchannels) == (params.norm_samplerate, 2, 2) and _bytes(s) == ref
    y = _rand(rng, 2, 3001)
    assert _bytes(_sample(y, 2, 8000, 1).make_32bit()) == audioop.lin2lin(y.tobytes(), 2, 4)
    assert _bytes(_sample(y, 2, 8000, 1).make_32bit(False)) == audioop.mul(audioop.lin2lin(y.tobytes(), 2, 4), 4, 1.0 / 65536)
    z = (x // 7).astype(np.int32)
    mx = audioop.max(z.tobytes(), 4)
    want = audioop.lin2lin(audioop.mul(z.tobytes(), 4, (2 ** 31 - 2) / mx), 4, 2)
    assert _bytes(_sample(z, 4, 8000, 1).make_16bit()) == want
    for w in (1, 2, 4):
        for nw in (1, 2, 4):
            from synthesizer_amd import _native as N
            v = _rand(rng, w, 1000)
            src = N.DeviceBuffer.from_array(v)
            dst = N.DeviceBuffer(1000 * nw)
            N.check(N.lib().sh_pcm_lin2lin(src.handle, 1000, w, nw, dst.handle))
            assert dst.download_bytes(1000 * nw) == audioop.lin2lin(v.tobytes(), w, nw)


@pytest.mark.parametrize("width", [1, 2, 4])
def test_peak_rms_amplify_max(gpu, width):
    rng = np.random.default_rng(20 + width)
    for n in (0, 1, 1000, 300001):
        x = _rand(rng, width, n)
        s = _sample(x, width, 8000, 1)
        assert s.peak() == audioop.max(x.tobytes(), width)
        if width < 4 or n <= 1:
            assert s.rms() == audioop.rms(x.tobytes(), width)
        else:       # width 4: float64 sums in a different order than audioop's sequential loop
            assert abs(s.rms() - audioop.rms(x.tobytes(), width)) <= 1
    q = (_rand(rng, width, 5000) // 3).astype(DT[width])
    mx = audioop.max(q.tobytes(), width)
    want = audioop.mul(q.tobytes(), width, (2 ** (8 * width - 1) - 2) / mx)
    assert _bytes(_sample(q, width, 8000, 1).amplify_max()) == want
    silent = np.zeros(100, dtype=DT[width])
    assert _bytes(_sample(silent, width, 8000, 1).amplify_max()) == silent.tobytes()


def test_fades(gpu):
    rng = np.random.default_rng(30)
    x = _rand(rng, 2, 8000 * 2)           # 1 s stereo at 8 kHz
    s = _sample(x, 2, 8000, 2).fadeout(0.25, 0.1)
    cut = 2 * 2 * int(8000 * 0.75)
    want = x.tobytes()[:cut] + P.fade(x.tobytes()[cut:], 2, True, 0.9, 0.0)
    assert _bytes(s) == want
    s = _sample(x, 2, 8000, 2).fadein(0.5, 0.2)
    cut = 2 * 2 * int(8000 * 0.5)
    want = P.fade(x.tobytes()[:cut], 2, False, 0.8, 0.2) + x.tobytes()[cut:]
    assert _bytes(s) == want
    # longer than the sample: the whole sample fades
    s = _sample(x, 2, 8000, 2).fadeout(5.0)
    assert _bytes(s) == P.fade(x.tobytes(), 2, True, 1.0, 0.0)
    assert s.get_frame_array()[-1] == 0 or abs(s.get_frame_array()[-1]) <= 1


def test_elementwise_and_editing_golden(gpu):
    """The committed golden vectors (live audioop outputs, tests/golden/audioop_ops.npz) through the GPU path."""
    g = np.load("tests/golden/audioop_ops.npz")
    for width in (1, 2, 4):
        x = g["x%d" % width]
        assert _bytes(_sample(x, width, 8000, 1).amplify(1.5)) == g["mul%d_1p5" % width].tobytes()
        assert _bytes(_sample(x, width, 8000, 1).amplify(-0.333)) == g["mul%d_m0p333" % width].tobytes()
        assert _bytes(_sample(x, width, 8000, 1).bias(1000)) == g["bias%d_1000" % width].tobytes()
        assert _bytes(_sample(x, width, 8000, 1).reverse()) == g["reverse%d" % width].tobytes()
        assert _bytes(_sample(x, width, 8000, 2).mono(0.75, 0.5)) == g["tomono%d" % width].tobytes()
        assert _bytes(_sample(x, width, 8000, 1).stereo(0.3, 1.2)) == g["tostereo%d" % width].tobytes()
        s = _sample(x, width, 8000, 1)
        peak, rms = g["max_rms%d" % width].tolist()
        assert s.peak() == peak and abs(s.rms() - rms) <= (1 if width == 4 else 0)
    y = g["edit_in"]
    assert _bytes(_sample(y, 2, 8000, 2).echo(0.1, 3, 0.05, 0.6)) == g["edit_echo"].tobytes()
    assert _bytes(_sample(y, 2, 8000, 2).envelope(0.05, 0.05, 0.5, 0.08)) == g["edit_envelope"].tobytes()
    assert _bytes(_sample(y, 2, 8000, 2).speed(1.26)) == g["edit_speed_1p26"].tobytes()
    assert _bytes(_sample(y, 2, 8000, 2).modulate_amp(_sample(g["edit_mod"], 2, 8000, 1))) == g["edit_modulate"].tobytes()


def test_stats_long_ragged_buffers(gpu):
    """peak / sum of squares over buffers long enough for several turns of the 512-workgroup sweep, lengths that end
    inside a turn, inside a vector and on an odd sample; mono and per-channel entry points; exact."""
    import ctypes as C
    from synthesizer_amd import _native as N
    rng = np.random.default_rng(11)
    for n in (512 * 8192 * 2 + 1024 * 8 * 3 + 5, 512 * 8192 + 8, 4_500_001 * 2):
        x = rng.integers(-32768, 32768, n).astype(np.int16)
        x[n - 1] = -32768                                          # the maximum sits in the ragged tail
        buf = N.DeviceBuffer.from_array(x)
        mx, sq = C.c_uint32(), C.c_double()
        N.check(N.lib().sh_pcm_stats(buf.handle, n * 2, 2, C.byref(mx), C.byref(sq)))
        x64 = x.astype(np.int64)
        assert mx.value == 32768 and int(sq.value) == int((x64 * x64).sum())
        if n % 2 == 0:
            mx2, sq2 = (C.c_uint32 * 2)(), (C.c_double * 2)()
            N.check(N.lib().sh_pcm_stats_stereo(buf.handle, n // 2, 2, mx2, sq2))
            for c in range(2):
                ch = x64[c::2]
                assert mx2[c] == int(np.abs(ch).max()) and int(sq2[c]) == int((ch * ch).sum())
