"""24-bit PCM (audioop width 3) on the device path: every Sample operation upstream delegates to audioop -- add (mix, mix_at),
mul (amplify, invert, amplify_max), bias, reverse, tomono / tostereo (mono, left, right, stereo, pan), lin2lin (make_16bit,
make_32bit, normalize), max / rms (peak, rms, level meters), ratecv (resample, speed; also sharded by output range) -- bit-exact
against the LIVE CPython 3.10 module on random 3-byte samples with the corner values, plus the C entry points with odd
offsets / unaligned buffers and a load_wav(24-bit) -> resample -> mix -> write_wav round trip.  (24-bit samples are unpacked to
int32 << 8, run through the 32-bit kernels and packed again: csrc/common.hpp explains why that is audioop's arithmetic.)"""
import audioop
import ctypes
import io
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CORNERS = [0x7FFFFF, -0x800000, -0x7FFFFF, 0x7FFFFE, 0, -1, 3, -3, 1, 255, 256, -256, 65535, -65536]


def rand24(rng, n, scale=1.0):
    v = (rng.integers(-0x800000, 0x800000, n, dtype=np.int64) * scale).astype(np.int64)
    k = min(n, len(CORNERS))
    v[:k] = CORNERS[:k]
    b = np.empty((n, 3), dtype=np.uint8)
    u = v & 0xFFFFFF
    b[:, 0], b[:, 1], b[:, 2] = u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF
    return b.tobytes()


def sample(raw, rate=8000, nch=1):
    from synthesizer_amd.sample import Sample
    return Sample.from_raw_frames(raw, 3, rate, nch)


def data(s):
    return bytes(s.view_frame_data())


def test_add_mul_bias_reverse_24bit(gpu):
    rng = np.random.default_rng(24)
    for n in (1, 3, 4, 5, 64, 4097, 100003):
        a, b = rand24(rng, n), rand24(rng, n)
        assert data(sample(a).mix(sample(b))) == audioop.add(a, b, 3), n
        for f in (1.5, 0.5, -1.0, 1.00001, 0.0, -0.333, 255.99, 1e-9):
            assert data(sample(a).amplify(f)) == audioop.mul(a, 3, f), (n, f)
        assert data(sample(a).invert()) == audioop.mul(a, 3, -1)
        for bias in (1, -1, 12345, 0x7FFFFF, -0x800000, 1 << 24):
            assert data(sample(a).bias(bias)) == audioop.bias(a, 3, bias), (n, bias)
        assert data(sample(a).reverse()) == audioop.reverse(a, 3)
    # the host-pointer entry point and odd byte offsets into device buffers (unaligned 3-byte streams)
    N = gpu
    a, b = rand24(rng, 1001), rand24(rng, 1001)
    out = ctypes.create_string_buffer(len(a))
    N.check(N.lib().sh_pcm_add_host(a, b, len(a), 3, out))
    assert out.raw == audioop.add(a, b, 3)
    da, db = N.DeviceBuffer.from_bytes(b"\x00\x00\x00" + a), N.DeviceBuffer.from_bytes(b"\x00" * 6 + b)
    do = N.DeviceBuffer(len(a) + 9)
    N.check(N.lib().sh_pcm_add(da.handle, 3, db.handle, 6, len(a), 3, do.handle, 9))
    assert do.download_bytes(len(a), 9) == audioop.add(a, b, 3)
    N.check(N.lib().sh_pcm_mul(da.handle, 3, len(a), 3, 0.7, do.handle, 3))
    assert do.download_bytes(len(a), 3) == audioop.mul(a, 3, 0.7)
    with pytest.raises(ValueError):
        N.check(N.lib().sh_pcm_add(da.handle, 0, db.handle, 0, 1000, 3, do.handle, 0))     # not a whole number of samples
    # equal lengths are audioop's rule
    with pytest.raises(ValueError):
        sample(a).mix(sample(a[:300]), pad_shortest=False)
    s = sample(a).mix_at(0.05, sample(b))
    start = 3 * int(8000 * 0.05)
    base = bytearray(a + b"\0" * (start + len(b) - len(a)))
    base[start:start + len(b)] = audioop.add(bytes(base[start:start + len(b)]), b, 3)
    assert data(s) == bytes(base)


def test_channels_and_widths_24bit(gpu):
    rng = np.random.default_rng(25)
    for frames in (1, 5, 3000):
        x = rand24(rng, frames * 2)
        for lf, rf in ((1.0, 1.0), (0.5, 0.25), (-1.0, 0.7)):
            s = sample(x, nch=2).mono(lf, rf)
            assert s.nchannels == 1 and data(s) == audioop.tomono(x, 3, lf, rf)
        assert data(sample(x, nch=2).left()) == audioop.tomono(x, 3, 1.0, 0)
        m = rand24(rng, frames)
        for lf, rf in ((1.0, 1.0), (0.5, 0.25), (2.0, -1.0)):
            s = sample(m).stereo(lf, rf)
            assert s.nchannels == 2 and data(s) == audioop.tostereo(m, 3, lf, rf)
        assert data(sample(m).pan(0.5)) == audioop.tostereo(m, 3, 0.25, 0.75)
    from synthesizer_amd import _native as N
    v = rand24(rng, 1000)
    src = N.DeviceBuffer.from_bytes(v)
    for nw in (1, 2, 3, 4):
        dst = N.DeviceBuffer(1000 * nw)
        N.check(N.lib().sh_pcm_lin2lin(src.handle, 1000, 3, nw, dst.handle))
        assert dst.download_bytes(1000 * nw) == audioop.lin2lin(v, 3, nw), nw
    for w, dt in ((1, np.int8), (2, np.int16), (4, np.int32)):
        info = np.iinfo(dt)
        y = rng.integers(info.min, info.max + 1, 1000, dtype=np.int64).astype(dt).tobytes()
        s2 = N.DeviceBuffer.from_bytes(y)
        dst = N.DeviceBuffer(3000)
        N.check(N.lib().sh_pcm_lin2lin(s2.handle, 1000, w, 3, dst.handle))
        assert dst.download_bytes(3000) == audioop.lin2lin(y, w, 3), w
    assert data(sample(v).make_32bit()) == audioop.lin2lin(v, 3, 4)
    q = rand24(rng, 5000, scale=0.3)
    mx = audioop.max(q, 3)
    assert data(sample(q).amplify_max()) == audioop.mul(q, 3, (2 ** 23 - 2) / mx)
    assert data(sample(q).make_16bit()) == audioop.lin2lin(audioop.mul(q, 3, (2 ** 23 - 2) / mx), 3, 2)


def test_peak_rms_levels_24bit(gpu):
    rng = np.random.default_rng(26)
    for n in (1, 1000, 300001):
        x = rand24(rng, n)
        s = sample(x)
        assert s.peak() == audioop.max(x, 3)
        assert abs(s.rms() - audioop.rms(x, 3)) <= 1            # float64 sums in another order than audioop's loop
    st = rand24(rng, 2 * 5000, scale=0.5)
    s = sample(st, nch=2)
    import math
    lp, rp = s.level_db_peak
    l, r = audioop.tomono(st, 3, 1, 0), audioop.tomono(st, 3, 0, 1)
    assert lp == max(20.0 * math.log((audioop.max(l, 3) + 1) / 2 ** 23, 10), -60.0)
    assert rp == max(20.0 * math.log((audioop.max(r, 3) + 1) / 2 ** 23, 10), -60.0)


@pytest.mark.parametrize("rates", [(96000, 44100), (44100, 48000), (48000, 16000), (8000, 8001), (22050, 44100)])
@pytest.mark.parametrize("nch", [1, 2, 5])
def test_resample_24bit(gpu, rates, nch):
    rng = np.random.default_rng(27)
    inr, outr = rates
    for frames in (1, 2, 17, 5000):
        x = rand24(rng, frames * nch)
        want = audioop.ratecv(x, 3, nch, inr, outr, None)[0]
        s = sample(x, rate=inr, nch=nch).resample(outr)
        assert data(s) == want and s.samplerate == outr, (frames, nch)
    # sharded by output range: the ranks' byte strings concatenate to ratecv of the whole input
    from synthesizer_amd import dist
    x = rand24(rng, 70001 * nch)
    want = audioop.ratecv(x, 3, nch, inr, outr, None)[0]
    for world in (1, 3):
        parts = [dist.resample_shard(x, 3, nch, inr, outr, r, world)[1] for r in range(world)]
        assert b"".join(parts) == want, world


def test_wav_24bit_load_resample_mix_write(gpu, tmp_path):
    """A 24-bit WAV file: load -> resample -> mix with another -> amplify -> write, every step on the GPU, against audioop."""
    from synthesizer_amd.sample import Sample
    rng = np.random.default_rng(28)
    a, b = rand24(rng, 2 * 9000, scale=0.4), rand24(rng, 2 * 7000, scale=0.4)
    for name, raw in (("a.wav", a), ("b.wav", b)):
        with wave.open(str(tmp_path / name), "wb") as w:
            w.setnchannels(2); w.setsampwidth(3); w.setframerate(44100); w.writeframes(raw)
    sa, sb = Sample(str(tmp_path / "a.wav")), Sample(str(tmp_path / "b.wav"))
    assert sa.samplewidth == 3 and sa.nchannels == 2 and len(sa) == 9000
    sa.resample(48000).mix(sb.resample(48000)).amplify(0.8)
    ra, rb = audioop.ratecv(a, 3, 2, 44100, 48000, None)[0], audioop.ratecv(b, 3, 2, 44100, 48000, None)[0]
    want = audioop.mul(audioop.add(ra, rb + b"\0" * (len(ra) - len(rb)), 3), 3, 0.8)
    assert data(sa) == want
    out = io.BytesIO()
    sa.write_wav(out)
    out.seek(0)
    with wave.open(out) as w:
        assert (w.getsampwidth(), w.getnchannels(), w.getframerate()) == (3, 2, 48000) and w.readframes(w.getnframes()) == want
    # what has no 24-bit form upstream either stays refused, loudly
    with pytest.raises(NotImplementedError):
        sa.fadeout(0.1)
    assert sa.get_frames_numpy().shape == (len(sa), 2) and sa.get_frames_numpy().dtype == np.int32
