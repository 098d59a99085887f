"""GPU parity, SURVEY.md section 8(a) row a11 -- the real-time mixer's chunk loop: RealTimeMixer.chunks() against the
oracle's restatement over ``audioop.add`` (oracle/sample_oracle.py RefRealTimeMixer), byte for byte, chunk for chunk,
with samples of ragged lengths, repeating samples, delayed starts, samples added and removed while it runs, and
levels that saturate (so the order of the fold matters)."""
import numpy as np
import pytest

from oracle.sample_oracle import RefRealTimeMixer, RefSample

pytestmark = pytest.mark.gpu


def _pair(x, rate=8000, nch=1, name=""):
    from synthesizer_amd.sample import Sample
    s = Sample.from_raw_frames(x.tobytes(), 2, rate, nch)
    s.name = name
    return s, RefSample(x.tobytes(), 2, rate, nch)


def _rand(rng, n, scale=1.0):
    return (rng.integers(-32768, 32768, n) * scale).astype(np.int16)


@pytest.mark.parametrize("chunksize", [512, 4096, 1000])
def test_chunk_stream_matches_audioop_fold(gpu, chunksize):
    from synthesizer_amd.mixer import RealTimeMixer
    rng = np.random.default_rng(chunksize)
    mixer, ref = RealTimeMixer(chunksize), RefRealTimeMixer(chunksize)
    played = []
    mixer.all_played_callback = lambda: played.append(mixer.chunks_mixed)
    specs = [(3000, 1.0, False, 0), (9001, 0.9, False, 2), (257, 1.0, True, 0), (20000, 0.7, False, 1), (5, 1.0, True, 3),
             (chunksize // 2, 1.0, False, 0), (chunksize // 2 * 3, 1.0, True, 0), (0, 1.0, False, 0)]
    for n, scale, repeat, delay in specs:                       # n int16 samples each, loud: the sum saturates
        s, r = _pair(_rand(rng, n, scale))
        assert mixer.add_sample(s, repeat=repeat, chunk_delay=delay) == ref.add_sample(r, repeat=repeat, chunk_delay=delay)
    got, want = mixer.chunks(), ref.chunks()
    for turn in range(12):
        a, b = bytes(next(got)), next(want)
        assert len(a) == chunksize and a == b, "turn %d" % turn
        if turn == 4:                                           # a late joiner and a removal while running
            s, r = _pair(_rand(rng, 7000))
            sid = mixer.add_sample(s)
            assert sid == ref.add_sample(r)
            mixer.remove_sample(3)
            ref.remove_sample(3)
    assert mixer.chunks_mixed == 12
    # only the repeating samples are left; drop them: silence from then on, and the callback fires
    for sid in list(mixer.active_samples):
        mixer.remove_sample(sid)
        ref.remove_sample(sid)
    assert bytes(next(got)) == next(want) == bytes(chunksize)
    mixer.clear_sources()
    assert played


def test_many_sources_and_device_chunks(gpu):
    """200 quiet sources (the 8-wave fold) + the device-resident stream feeding the level meter."""
    from synthesizer_amd.mixer import RealTimeMixer
    from synthesizer_amd.sample import LevelMeter, Sample
    from oracle.sample_oracle import RefLevelMeter
    rng = np.random.default_rng(77)
    chunksize = 8192
    mixer, ref = RealTimeMixer(chunksize), RefRealTimeMixer(chunksize)
    for k in range(200):
        s, r = _pair(_rand(rng, int(rng.integers(1, 30000)), 0.02 if k % 10 else 1.0))
        mixer.add_sample(s, repeat=k % 7 == 0, chunk_delay=k % 3)
        ref.add_sample(r, repeat=k % 7 == 0, chunk_delay=k % 3)
    meter, ref_meter = LevelMeter(), RefLevelMeter()
    want = ref.chunks()
    for turn, dev in zip(range(10), mixer.chunks_device()):
        chunk = Sample(samplerate=8000, nchannels=2, samplewidth=2)
        chunk._set_device(dev, chunksize)
        b = next(want)
        assert meter.update(chunk) == ref_meter.update(RefSample(b, 2, 8000, 2))
        assert dev.download_bytes(chunksize) == b, "turn %d" % turn


def test_mix_samples_reads_sources_in_place(gpu):
    """mix_samples over ragged, device-resident samples (no staging copy) equals the chain of audioop.add."""
    import audioop
    from synthesizer_amd.mixer import mix_samples
    rng = np.random.default_rng(3)
    xs = [_rand(rng, n) for n in (10001 * 2, 8 * 2, 4097 * 2, 1 * 2, 10001 * 2)]
    samples = [_pair(x, nch=2)[0].to_device() for x in xs]
    longest = max(len(x) for x in xs) * 2
    mixed = bytes(longest)
    for x in xs:
        mixed = audioop.add(mixed, x.tobytes() + bytes(longest - 2 * len(x)), 2)
    out = mix_samples(samples)
    assert bytes(out.view_frame_data()) == mixed
    # a source that does not start on a 16-byte boundary: a clipped sample (view into its parent buffer, if it is one)
    s, r = _pair(xs[0], nch=2)
    s.to_device().clip(0.000125, 1.0)
    r.clip(0.000125, 1.0)
    out = mix_samples([s, samples[2]])
    want = audioop.add(r.frames + bytes(max(0, len(xs[2]) * 2 - len(r.frames))), xs[2].tobytes() + bytes(max(0, len(r.frames) - len(xs[2]) * 2)), 2)
    assert bytes(out.view_frame_data()) == want


def test_mixer_argument_checks(gpu):
    from synthesizer_amd.mixer import RealTimeMixer
    from synthesizer_amd.sample import Sample
    with pytest.raises(ValueError):
        RealTimeMixer(0)
    with pytest.raises(ValueError):
        RealTimeMixer(512, samplewidth=5)
    with pytest.raises(ValueError):
        RealTimeMixer(514, samplewidth=4)            # not a whole number of 32-bit samples
    with pytest.raises(ValueError):
        RealTimeMixer(512).add_sample(Sample.from_raw_frames(bytes(16), 4, 8000, 1))


def test_two_threads_share_the_library(gpu):
    """The way upstream drives the mixer: one thread pulls chunks, another adds samples and does its own Sample work
    meanwhile.  Every ctypes call drops the GIL, so the native calls really interleave; results on both sides must
    still be exact.  (Samples of silence are added at arbitrary moments: they cannot change the mix, whenever they
    land, so the chunk stream stays comparable with the oracle's.)"""
    import audioop
    import threading
    from synthesizer_amd.mixer import RealTimeMixer
    from synthesizer_amd.sample import Sample
    rng = np.random.default_rng(2024)
    chunksize, turns = 2048, 300
    mixer, ref = RealTimeMixer(chunksize), RefRealTimeMixer(chunksize)
    for k in range(40):
        s, r = _pair(_rand(rng, int(rng.integers(1000, 200000)), 0.3))
        mixer.add_sample(s, repeat=k % 4 == 0)
        ref.add_sample(r, repeat=k % 4 == 0)
    want_chunks = [c for _, c in zip(range(turns), ref.chunks())]
    got_chunks, errors = [], []

    def drain():
        try:
            for _, c in zip(range(turns), mixer.chunks()):
                got_chunks.append(bytes(c))
        except Exception as e:                                   # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=drain)
    x = _rand(rng, 2 * 30000)
    y = _rand(rng, 2 * 30000)
    want_add = audioop.add(x.tobytes(), y.tobytes(), 2)
    want_rate = audioop.ratecv(x.tobytes(), 2, 2, 8000, 11025, None)[0]
    want_peak = audioop.max(x.tobytes(), 2)
    t.start()
    rounds = 0
    while t.is_alive() or rounds < 5:
        a = Sample.from_raw_frames(x.tobytes(), 2, 8000, 2)
        b = Sample.from_raw_frames(y.tobytes(), 2, 8000, 2)
        assert a.peak() == want_peak
        assert bytes(a.copy().mix(b).view_frame_data()) == want_add
        assert bytes(a.resample(11025).view_frame_data()) == want_rate
        mixer.add_sample(Sample.from_raw_frames(bytes(2 * int(rng.integers(1, 5000))), 2, 8000, 1), repeat=bool(rounds & 1))
        rounds += 1
    t.join()
    assert not errors
    assert got_chunks == want_chunks


def _audioop_fold(rows, nsamples):
    import audioop
    mixed = bytes(2 * nsamples)
    for x in rows:
        mixed = audioop.add(mixed, x.tobytes() + bytes(2 * (nsamples - len(x))), 2)
    return mixed


@pytest.mark.parametrize("nsamples", [262144 + 40, 786432 + 8 + 3, 1200000])
def test_long_buffers_every_kernel_shape(gpu, nsamples):
    """The fold picks its kernel by buffer length (voices split over waves / two columns per workgroup / the direct
    loop): the three shapes, through both entry points (padded array, pointer table), loud voices so that the order of
    the saturating adds shows, ragged ends and a source off the 16-byte grid."""
    import ctypes as C
    from synthesizer_amd import _native as N
    from synthesizer_amd.mixer import mix_samples
    rng = np.random.default_rng(nsamples)
    nv = 9
    rows = [_rand(rng, nsamples, 0.6) for _ in range(nv)]
    want = _audioop_fold(rows, nsamples)
    # padded array
    stride = (nsamples + 7) // 8 * 8
    chunks = N.DeviceBuffer(nv * stride * 2)
    chunks.zero()
    for v, x in enumerate(rows):
        chunks.upload(x, v * stride * 2)
    out = N.DeviceBuffer(nsamples * 2)
    N.check(N.lib().sh_mix_chain_i16(chunks.handle, nv, stride, nsamples, out.handle))
    assert out.download_bytes(nsamples * 2) == want
    # pointer table, ragged: sources shorter than the output, one empty, one starting 2 bytes into its buffer
    lens = [nsamples, nsamples - 5, 1000, 0, nsamples // 2 + 1, nsamples, 8, nsamples - 1, nsamples]
    ragged = [x[:n] for x, n in zip(rows, lens)]
    want = _audioop_fold(ragged, nsamples)
    samples = [_pair(x)[0].to_device() for x in ragged]
    assert bytes(mix_samples(samples).view_frame_data()) == want
    shifted = N.DeviceBuffer.from_array(np.concatenate([np.zeros(1, np.int16), ragged[0]]))
    bufs = (C.c_void_p * 2)(shifted.handle, samples[1]._device().handle)
    offs = (C.c_size_t * 2)(1, 0)
    cnt = (C.c_uint32 * 2)(len(ragged[0]), len(ragged[1]))
    N.check(N.lib().sh_mix_chain_gather_i16(bufs, offs, cnt, 2, nsamples, out.handle, 0))
    assert out.download_bytes(nsamples * 2) == _audioop_fold(ragged[:2], nsamples)


def _rand_pcm(rng, nsamples, width, scale=1.0):
    """nsamples random samples of `width` bytes, little endian, loud enough to saturate sums."""
    bits = 8 * width
    x = (rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), nsamples) * scale).astype(np.int64)
    if width == 3:
        b = np.zeros((nsamples, 3), dtype=np.uint8)
        u = x & 0xFFFFFF
        b[:, 0], b[:, 1], b[:, 2] = u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF
        return b.tobytes()
    return x.astype({1: np.int8, 2: np.int16, 4: np.int32}[width]).tobytes()


@pytest.mark.parametrize("width", [1, 2, 3, 4])
def test_mixer_of_every_sample_width(gpu, width):
    """a11 for the widths audioop.add takes: RealTimeMixer(samplewidth=w) chunk for chunk against the oracle's loop of
    audioop.add(mixed, chunk, w), mix_samples against the same fold over whole samples, and the strided entry point
    (sh_mix_chain) against a loop of live audioop.add."""
    import audioop
    from synthesizer_amd import _native as N
    from synthesizer_amd.mixer import RealTimeMixer, mix_samples
    from synthesizer_amd.sample import Sample
    rng = np.random.default_rng(100 + width)
    chunksize = 1200 * width
    mixer, ref = RealTimeMixer(chunksize, samplewidth=width), RefRealTimeMixer(chunksize, samplewidth=width)
    raws = []
    for n, scale, repeat, delay in [(3000, 1.0, False, 0), (9001, 0.6, False, 2), (257, 1.0, True, 0), (5000, 0.7, False, 1),
                                    (600, 1.0, False, 0), (7, 1.0, True, 1), (0, 1.0, False, 0)]:
        raw = _rand_pcm(rng, n, width, scale)
        raws.append(raw)
        s = Sample.from_raw_frames(raw, width, 8000, 1)
        assert mixer.add_sample(s, repeat=repeat, chunk_delay=delay) == ref.add_sample(RefSample(raw, width, 8000, 1), repeat=repeat, chunk_delay=delay)
    got, want = mixer.chunks(), ref.chunks()
    for turn in range(10):
        a, b = bytes(next(got)), next(want)
        assert len(a) == chunksize and a == b, "width %d turn %d" % (width, turn)
    with pytest.raises(ValueError):
        mixer.add_sample(Sample.from_raw_frames(_rand_pcm(rng, 10, 2 if width != 2 else 1), 2 if width != 2 else 1, 8000, 1))
    # whole samples: pad with silence to the longest, fold in order
    samples = [Sample.from_raw_frames(r, width, 8000, 1) for r in raws[:5]]
    longest = max(len(r) for r in raws[:5])
    acc = raws[0] + bytes(longest - len(raws[0]))
    for r in raws[1:5]:
        acc = audioop.add(acc, r + bytes(longest - len(r)), width)
    assert bytes(mix_samples(samples).view_frame_data()) == acc
    # the strided form: nv rows of one buffer
    nv, ns = 37, 5003
    rows = [_rand_pcm(rng, ns, width, 0.2) for _ in range(nv)]
    buf = N.DeviceBuffer.from_bytes(b"".join(rows))
    out = N.DeviceBuffer(ns * width)
    N.check(N.lib().sh_mix_chain(buf.handle, nv, ns, ns, width, out.handle))
    acc = rows[0]
    for r in rows[1:]:
        acc = audioop.add(acc, r, width)
    assert out.download_bytes(ns * width) == acc


def test_a_sample_the_mixer_streams_from_is_never_mixed_into_in_place(gpu):
    """Sample.mix adds in place when nothing grows -- but not into a buffer a RealTimeMixer reads (add_sample keeps the
    Sample's device buffer by reference), and not when a sample is mixed into itself at an offset."""
    import audioop
    from synthesizer_amd.mixer import RealTimeMixer
    from synthesizer_amd.sample import Sample
    rng = np.random.default_rng(3)
    a = rng.integers(-9000, 9000, 4096, dtype=np.int16)
    b = rng.integers(-9000, 9000, 4096, dtype=np.int16)
    s = Sample.from_raw_frames(a.tobytes(), 2, 8000, 1).to_device()
    mixer = RealTimeMixer(2048)
    mixer.add_sample(s)
    s.mix(Sample.from_raw_frames(b.tobytes(), 2, 8000, 1))            # the Sample changes ...
    assert bytes(s.view_frame_data()) == audioop.add(a.tobytes(), b.tobytes(), 2)
    chunks = mixer.chunks()
    got = bytes(next(chunks)) + bytes(next(chunks))
    assert got == a.tobytes()[:4096]                                   # ... what the mixer plays does not (two chunks of 2048 bytes)
    # a sample mixed into itself at an offset: source and destination overlap at shifted positions
    t = Sample.from_raw_frames(a.tobytes(), 2, 8000, 1).to_device()
    t.mix_at(0.1, t, other_seconds=0.2)                                # frames 800 .. 2400 += frames 0 .. 1600
    want = a.copy().astype(np.int32)
    want[800:2400] = np.clip(want[800:2400] + a[:1600].astype(np.int32), -32768, 32767)
    assert np.array_equal(np.frombuffer(t.view_frame_data(), dtype=np.int16), want.astype(np.int16))


def test_the_real_time_lane_beside_a_streaming_bank(gpu):
    """a thread that drains RealTimeMixer.chunks() and a thread that streams a VoiceBank's blocks.  chunks() runs on the
    mixer's own lane (stream, lock, buffers: sh_rt_*), so a turn neither waits for the library's lock nor queues behind the bank's launches,
    and it does not end the bank's run of pipelined renders.  Checked: the chunks are the oracle's; the bank's blocks are what it renders
    alone (bit for bit); and the two overlap -- the bank renders, beside the mixer, at no less than 0.7 of its rate alone (measured 0.98;
    through the one lock and stream of rounds 1-5, timed below for the record: 0.50, every mixer turn drains the bank's pipeline), while the
    mixer makes its turns (12 390 per second, median 74 us, against 7 107 and 140 us)."""
    import threading
    import time
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import RealTimeMixer, VoiceBank
    from synthesizer_amd.workloads import additive_voices
    rng = np.random.default_rng(7)
    chunksize = 4096
    mixer, ref = RealTimeMixer(chunksize), RefRealTimeMixer(chunksize)
    for k in range(64):
        s, r = _pair(_rand(rng, int(rng.integers(200000, 400000)), 0.2))
        mixer.add_sample(s, repeat=True)
        ref.add_sample(r, repeat=True)
    SR, nv, F = 48000, 1024, 48000
    gv, gains = additive_voices(G, nv, SR, seed=0, adsr={"sustain": 1.0e6})
    bank = VoiceBank(gv, gains=gains)
    ring = [N.DeviceBuffer(F * 8) for _ in range(4)]

    def stream_bank(seconds, first):
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(16):
                bank.render_device(F, (first + n) * F, bus_f32=ring[n & 3])
                n += 1
            if n % 256 == 0:
                N.sync()                      # (bounded queue depth: a real producer paces itself)
        N.sync()
        return n, time.perf_counter() - t0
    stream_bank(0.3, 10)                       # clocks up, shapes seen
    alone_n, alone_t = stream_bank(1.0, 1000)
    got, lat, stop, errors = [], [], threading.Event(), []

    def drain():
        try:
            it = mixer.chunks()
            while not stop.is_set():
                t0 = time.perf_counter()
                got.append(bytes(next(it)))
                lat.append(time.perf_counter() - t0)
        except Exception as e:                                   # pragma: no cover
            errors.append(e)
    t = threading.Thread(target=drain)
    t.start()
    both_n, both_t = stream_bank(1.0, 100000)
    stop.set()
    t.join()
    assert not errors, errors
    want = [c for _, c in zip(range(len(got)), ref.chunks())]
    assert got == want
    # the same pair of threads through the library's one lock and stream (rounds 1-5), for the record
    old_mixer = RealTimeMixer(chunksize)
    old_mixer.use_lane = False
    for k in range(64):
        old_mixer.add_sample(_pair(_rand(rng, 250000, 0.2))[0], repeat=True)
    old_lat, stop2 = [], threading.Event()

    def drain_old():
        it = old_mixer.chunks()
        while not stop2.is_set():
            t0 = time.perf_counter()
            next(it)
            old_lat.append(time.perf_counter() - t0)
    t2 = threading.Thread(target=drain_old)
    t2.start()
    old_n, old_t = stream_bank(1.0, 200000)
    stop2.set()
    t2.join()
    ol = sorted(x * 1e6 for x in old_lat)
    print("rounds 1-5 (one lock, one stream pair): bank beside the mixer %.0f blocks/s (%.2f x alone); mixer %d turns, median %.0f us, 90 %% %.0f us"
          % (old_n / old_t, old_n / old_t / (alone_n / alone_t), len(ol), ol[len(ol) // 2], ol[int(len(ol) * 0.9)]))
    rate_alone, rate_both = alone_n / alone_t, both_n / both_t
    lat_us = sorted(x * 1e6 for x in lat)
    print("bank alone %.0f blocks/s, beside the mixer %.0f (%.2f x); mixer: %d turns beside the bank, median %.0f us, 90 %% %.0f us per turn (chunk on the host)"
          % (rate_alone, rate_both, rate_both / rate_alone, len(lat), lat_us[len(lat_us) // 2], lat_us[int(len(lat_us) * 0.9)]))
    assert len(got) >= 200 and rate_both >= 0.7 * rate_alone          # (measured 0.98; 0.50 without the lane)
    # the bank's block beside the mixer == the same block rendered alone
    a = N.DeviceBuffer(F * 8)
    bank.render_device(F, 100003 * F, bus_f32=a)
    assert np.array_equal(a.download(np.float32, F * 2), VoiceBank(additive_voices(G, nv, SR, seed=0, adsr={"sustain": 1.0e6})[0], gains=gains).render(F, 100003 * F).reshape(-1))
