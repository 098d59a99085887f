"""Buffers beyond 2^32 bytes and 2^31 samples: the 64-bit index arithmetic of the PCM kernels (audioop.add, audioop.max / rms,
audioop.ratecv for 16-bit PCM and for the float32 shape of BASELINE configs[4]) at sizes no test with a CPU-side expectation
of the whole result could hold.  Inputs are one tile of noise repeated with an ODD period (an index that wrapped at 2^32 would
land on other data), results are checked in windows -- at the head, across the 2^31-sample and 2^32-byte marks, at the tail --
against CPython's audioop / the oracle's closed form on just the input those windows read."""
import audioop

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fill(buf, tile: np.ndarray, total_elems: int) -> None:
    """buf[e] = tile[e % len(tile)] for e < total_elems (uploads of at most one tile)."""
    n, size = len(tile), tile.itemsize
    for e0 in range(0, total_elems, n):
        m = min(n, total_elems - e0)
        buf.upload(tile[:m], e0 * size)


def test_pcm_add_and_stats_beyond_4_gib(gpu):
    from synthesizer_amd import _native as N
    L = N.lib()
    n = (1 << 31) + (1 << 29) + 12345                       # 2.68 G samples of int16: 5.4 GB per operand
    rng = np.random.default_rng(2026)
    ta = rng.integers(-32768, 32768, (1 << 24) - 3, dtype=np.int64).astype(np.int16)
    tb = rng.integers(-32768, 32768, (1 << 24) - 13, dtype=np.int64).astype(np.int16)
    a, b, o = N.DeviceBuffer(2 * n), N.DeviceBuffer(2 * n), N.DeviceBuffer(2 * n)
    _fill(a, ta, n)
    _fill(b, tb, n)
    N.check(L.sh_pcm_add(a.handle, 0, b.handle, 0, 2 * n, 2, o.handle, 0))
    w = 1 << 16
    for first in (0, (1 << 31) - w // 2, (1 << 32) // 2 - w // 2 + 1, (1 << 31) + (1 << 28) + 7, n - w):
        got = o.download(np.int16, w, 2 * first)
        idx = np.arange(first, first + w)
        want = np.frombuffer(audioop.add(ta[idx % len(ta)].tobytes(), tb[idx % len(tb)].tobytes(), 2), dtype=np.int16)
        assert np.array_equal(got, want), first
    # in place at an offset past 4 GiB (Sample.mix_at): out = a, both ranges start beyond 2^32 bytes
    off = (1 << 32) + 2 * 4097
    cnt = 2 * n - off
    N.check(L.sh_pcm_add(a.handle, off, b.handle, off, cnt, 2, a.handle, off))
    first = off // 2
    for lo in (first, n - w):
        got = a.download(np.int16, w, 2 * lo)
        idx = np.arange(lo, lo + w)
        want = np.frombuffer(audioop.add(ta[idx % len(ta)].tobytes(), tb[idx % len(tb)].tobytes(), 2), dtype=np.int16)
        assert np.array_equal(got, want), lo
    assert np.array_equal(a.download(np.int16, w, 2 * (first - w)), ta[np.arange(first - w, first) % len(ta)])   # untouched below
    # audioop.max and the sum of squares behind audioop.rms over all of b: whole tiles + the ragged rest, exactly
    mx, ss = N.C.c_uint32(0), N.C.c_double(0.0)
    N.check(L.sh_pcm_stats(b.handle, 2 * n, 2, N.C.byref(mx), N.C.byref(ss)))
    reps, rest = divmod(n, len(tb))
    sq = tb.astype(np.int64) ** 2
    want_ss = reps * int(sq.sum()) + int(sq[:rest].sum())
    assert mx.value == int(np.abs(tb.astype(np.int64)).max())
    assert abs(ss.value - want_ss) <= want_ss * 2e-16 * 4
    for x in (a, b, o):
        x.free()


def test_resample_int16_mono_beyond_4_gib(gpu):
    """44.1 kHz -> 48 kHz, 2.2 G frames of 16-bit mono (4.4 GB in, 4.8 GB out).  An output frame m = 160 k sits exactly on
    input frame 147 k, so audioop.ratecv started on the input from there reproduces the outputs from m on."""
    from synthesizer_amd import _native as N
    L = N.lib()
    inrate, outrate = 44100, 48000
    in_frames = (1 << 31) + (1 << 26) + 999
    tile = np.random.default_rng(7).integers(-32768, 32768, (1 << 24) - 5, dtype=np.int64).astype(np.int16)
    src = N.DeviceBuffer(2 * in_frames)
    _fill(src, tile, in_frames)
    nout = L.sh_resample_out_frames(in_frames, inrate, outrate)
    dst = N.DeviceBuffer(2 * nout)
    made = N.C.c_size_t(0)
    N.check(L.sh_resample(src.handle, in_frames, 1, 2, 0, inrate, outrate, dst.handle, N.C.byref(made)))
    assert made.value == nout
    w = 8000
    marks = [0, ((1 << 31) // 160) * 160 - 160 * 20, ((1 << 31) * 160 // 147 // 160) * 160 - 160 * 10, ((nout - w - 200) // 160) * 160]
    for m in marks:
        q = m // 160 * 147                                   # the input frame output m coincides with
        need = w * 147 // 160 + 8
        idx = np.arange(q, min(q + need, in_frames))
        want = np.frombuffer(audioop.ratecv(tile[idx % len(tile)].tobytes(), 2, 1, inrate, outrate, None)[0], dtype=np.int16)
        cnt = min(w, len(want), nout - m)
        got = dst.download(np.int16, cnt, 2 * m)
        assert np.array_equal(got, want[:cnt]), m
    # the very last output frames
    tail = dst.download(np.int16, 64, 2 * (nout - 64))
    m = ((nout - 64) // 160) * 160
    q = m // 160 * 147
    idx = np.arange(q, in_frames)
    want = np.frombuffer(audioop.ratecv(tile[idx % len(tile)].tobytes(), 2, 1, inrate, outrate, None)[0], dtype=np.int16)
    assert len(want) == nout - m and np.array_equal(tail, want[-64:])
    src.free()
    dst.free()


def test_resample_float32_8ch_beyond_4_gib(gpu):
    """The shape of BASELINE configs[4] at 2.4 times its length: 8 channels x 1450 s x 96 kHz float32 = 4.45 GB -> 44.1 kHz."""
    from oracle import pcm_oracle as P
    from synthesizer_amd import _native as N
    L = N.lib()
    nch, inrate, outrate = 8, 96000, 44100
    in_frames = 96000 * 1450 + 77
    assert in_frames * nch * 4 > (1 << 32)
    tile_frames = (1 << 21) - 3
    tile = np.random.default_rng(56).uniform(-1, 1, (tile_frames, nch)).astype(np.float32)
    src = N.DeviceBuffer(in_frames * nch * 4)
    for f0 in range(0, in_frames, tile_frames):
        k = min(tile_frames, in_frames - f0)
        src.upload(tile[:k].reshape(-1), f0 * nch * 4)
    nout = L.sh_resample_out_frames(in_frames, inrate, outrate)
    assert nout == P.ratecv_out_frames(in_frames, inrate, outrate)
    dst = N.DeviceBuffer(nout * nch * 4)
    N.check(L.sh_resample(src.handle, in_frames, nch, 4, 1, inrate, outrate, dst.handle, None))
    get = lambda idx: tile[idx % tile_frames]
    w = 4096
    in_mark = (1 << 32) // (nch * 4)                          # the input frame at byte 2^32
    for first in (0, in_mark * outrate // inrate - w // 2, nout // 2 + 3, nout - w):
        got = dst.download(np.float32, w * nch, first * nch * 4).reshape(w, nch)
        assert np.array_equal(got, P.ratecv_f32_window(get, inrate, outrate, first, w)), first
    src.free()
    dst.free()


def test_bank_render_of_two_billion_frames_in_one_launch(gpu):
    """sh_bank_render with nframes beyond 2^31 (12.4 hours of audio at 48 kHz, a 17 GB float32 bus) for a small bank of mixed
    kinds: windows of the one launch -- head, either side of frame 2^31, tail -- against the same frames rendered as short
    launches of their own (other launch shapes, other launch records: float64 rounding before the one rounding to float32),
    and the first window against the oracle.  The library cuts such a render into launches of 2^22 frames: a launch's grid
    counts work-items in 32 bits per dimension, and a bank with several voice groups keeps 32 B of partial buses per frame
    and group."""
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.workloads import additive_voices
    SR = 48000
    harm = [(k, 1.0 / k) for k in range(1, 17)]

    def voices(M):
        return [M.Sine(440.0, 0.1, samplerate=SR), M.Harmonics(97.3, harm, amplitude=0.05, phase=0.2, samplerate=SR),
                M.Square(311.0, 0.03, samplerate=SR), M.Sawtooth(55.5, 0.04, phase=0.7, samplerate=SR),
                M.Sine(220.0, 0.08, fm_lfo=M.Sine(3.0, 0.01, samplerate=SR), samplerate=SR),
                M.Triangle(1234.5, 0.05, samplerate=SR), M.Pulse(77.0, 0.02, pulsewidth=0.2, samplerate=SR),
                M.Harmonics(1500.1, harm[:5], amplitude=0.03, samplerate=SR)]

    gains = [(0.9, 0.1), (0.5, 0.5), (0.2, 0.8), (1.0, 0.0), (0.3, 0.7), (0.6, 0.4), (0.0, 1.0), (0.7, 0.7)]
    bank = VoiceBank(voices(G), gains=gains)
    n = (1 << 31) + (1 << 20) + 321
    bus = N.DeviceBuffer(8 * n)
    bank.render_device(n, 0, bus_f32=bus)
    w = 20000
    for first in (0, (1 << 31) - w // 2, (1 << 31) + 4097, 1_000_000_007, n - w):
        got = bus.download(np.float32, 2 * w, 8 * first).reshape(w, 2)
        want = bank.render(w, start=first)
        assert np.max(np.abs(got.astype(np.float64) - want)) <= 2e-7, first
        assert np.mean(got != want) < 0.01, first
    rows = np.stack([CO.render(o, w) for o in voices(O)])
    ref = np.array(CO.mix_bus(rows, gains), dtype=np.float64)
    assert np.sqrt(np.mean((bus.download(np.float32, 2 * w).reshape(w, 2) - ref) ** 2)) <= 1e-6
    bus.free()
    # 256 voices -> several voice groups -> 2 x groups x 16 B of partial buses per frame: hundreds of GB at this length in one
    # launch; as a run of RENDER_MAX_FRAMES launches (two-stream pipeline, partial buses of one launch each) it is 0.4 s of GPU
    big, bg = additive_voices(G, 256, SR, seed=1, adsr={"sustain": 44000.0})     # released at frame 2.112e9: the last windows are silent
    bank2 = VoiceBank(big, gains=bg)
    out = N.DeviceBuffer(8 * n)
    bank2.render_device(n, 0, bus_f32=out)
    for first in (0, (1 << 22) - w // 2, 1_900_000_000, 2_112_000_000 + 2000, (1 << 31) - w // 2, n - w):   # 2^22: a seam between two launches
        got = out.download(np.float32, 2 * w, 8 * first).reshape(w, 2)
        want = bank2.render(w, start=first)
        assert (np.abs(want).max() > 1e-3) == (first < 2_120_000_000), first
        assert np.max(np.abs(got.astype(np.float64) - want)) <= 2e-7, first
        assert np.mean(got != want) < 0.01, first
    out.free()

def test_one_sample_per_thread_kernels_beyond_2_32_samples(gpu):
    """The elementwise Sample operations whose kernels take ONE sample per thread (bias, reverse, lin2lin, fade) on more than
    2^32 samples of 8-bit PCM: a dispatch counts work-items in 32 bits per grid dimension, so these launches fold their
    blocks into two dimensions (csrc/common.hpp grid1d / block_id) -- unfolded, the launch ran 2^20 threads and reported
    success."""
    from synthesizer_amd import _native as N
    L = N.lib()
    n = (1 << 32) + (1 << 20) + 77
    tile = np.random.default_rng(99).integers(-128, 128, (1 << 24) - 7, dtype=np.int64).astype(np.int8)
    src, dst = N.DeviceBuffer(n), N.DeviceBuffer(n)
    _fill(src, tile, n)
    w = 1 << 16
    marks = (0, (1 << 31) - w // 2, (1 << 32) - w // 2, (1 << 32) + 4099, n - w)
    at = lambda first: tile[np.arange(first, first + w) % len(tile)]
    N.check(L.sh_pcm_bias(src.handle, n, 1, 3, dst.handle))
    for first in marks:
        assert dst.download(np.int8, w, first).tobytes() == audioop.bias(at(first).tobytes(), 1, 3), first
    N.check(L.sh_pcm_reverse(src.handle, n, 1, dst.handle))
    for first in marks:
        want = tile[np.arange(n - 1 - first, n - 1 - first - w, -1) % len(tile)]
        assert np.array_equal(dst.download(np.int8, w, first), want), first
    N.check(L.sh_pcm_fade(src.handle, 0, n, 1, 1, 1.0, 1.0, dst.handle, 0))         # Sample.fadeout to silence over the whole length
    for first in marks:
        i = np.arange(first, first + w, dtype=np.float64)
        ref = np.trunc(at(first).astype(np.float64) * (1.0 - i * 1.0 / float(n))).astype(np.int8)
        assert np.array_equal(dst.download(np.int8, w, first), ref), first
    dst.free()
    wide = N.DeviceBuffer(2 * n)
    N.check(L.sh_pcm_lin2lin(src.handle, n, 1, 2, wide.handle))
    for first in marks:
        assert wide.download(np.int16, w, 2 * first).tobytes() == audioop.lin2lin(at(first).tobytes(), 1, 2), first
    src.free()
    wide.free()


def test_mix_chain_beyond_2_30_samples(gpu):
    """sh_mix_chain_i16 on rows of more than 2^30 samples with a stride that rules out the 16-byte path: the split kernel's grid
    (one workgroup per 512 samples) is folded into two dimensions beyond 2^21 workgroups."""
    from synthesizer_amd import _native as N
    L = N.lib()
    ns = (1 << 30) + (1 << 20) + 3
    stride = ns + 5                                          # odd: rows 1 and 2 start off the 16-byte grid
    nv = 3
    tiles = [np.random.default_rng(40 + v).integers(-20000, 20000, (1 << 23) - 3 - 2 * v, dtype=np.int64).astype(np.int16) for v in range(nv)]
    src = N.DeviceBuffer(2 * (stride * (nv - 1) + ns))
    for v in range(nv):
        t = tiles[v]
        for e0 in range(0, ns, len(t)):
            m = min(len(t), ns - e0)
            src.upload(t[:m], 2 * (v * stride + e0))
    dst = N.DeviceBuffer(2 * ns)
    N.check(L.sh_mix_chain_i16(src.handle, nv, stride, ns, dst.handle))
    w = 1 << 15
    for first in (0, (1 << 30) - w // 2, (1 << 30) + 512 * 3 + 1, ns - w):
        idx = np.arange(first, first + w)
        want = tiles[0][idx % len(tiles[0])].tobytes()
        for v in range(1, nv):
            want = audioop.add(want, tiles[v][idx % len(tiles[v])].tobytes(), 2)
        assert dst.download(np.int16, w, 2 * first).tobytes() == want, first
    src.free()
    dst.free()


def test_24bit_add_beyond_2_31_samples(gpu):
    """audioop.add on 2^31 + 2^22 samples of 24-bit PCM (6.5 GB per operand): the unpack / 32-bit add / pack route with its int32
    temporaries of 8.6 GB each, whose kernels take four samples per thread in folded grids."""
    from synthesizer_amd import _native as N
    L = N.lib()
    n = (1 << 31) + (1 << 22) + 8
    rng = np.random.default_rng(24)
    ta = rng.integers(0, 256, 3 * ((1 << 22) - 5), dtype=np.int64).astype(np.uint8)          # whole 3-byte samples
    tb = rng.integers(0, 256, 3 * ((1 << 22) - 11), dtype=np.int64).astype(np.uint8)
    a, b, o = N.DeviceBuffer(3 * n), N.DeviceBuffer(3 * n), N.DeviceBuffer(3 * n)
    _fill(a, ta, 3 * n)
    _fill(b, tb, 3 * n)
    N.check(L.sh_pcm_add(a.handle, 0, b.handle, 0, 3 * n, 3, o.handle, 0))
    w = 1 << 15                                               # samples per window
    for first in (0, (1 << 31) - w // 2, ((1 << 32) // 3) - w // 2, (1 << 31) + (1 << 21) + 5, n - w):
        idx = np.arange(3 * first, 3 * (first + w))
        want = audioop.add(ta[idx % len(ta)].tobytes(), tb[idx % len(tb)].tobytes(), 3)
        assert o.download(np.uint8, 3 * w, 3 * first).tobytes() == want, first
    for x in (a, b, o):
        x.free()


def test_quantise_beyond_2_32_samples(gpu):
    """sh_quantize_f32 / sh_quantize_clip_f32 on 2^32 + 2^20 float32 samples (17 GB in, 8.6 GB out): the vector kernel's grid (2048
    samples per workgroup) folds exactly there; windows against int(scale * v)."""
    from oracle import synth_oracle as O
    from synthesizer_amd import _native as N
    L = N.lib()
    n = (1 << 32) + (1 << 20) + 4
    tile = np.random.default_rng(31).uniform(-1.0, 1.0, (1 << 24) - 9).astype(np.float32)
    src, dst = N.DeviceBuffer(4 * n), N.DeviceBuffer(2 * n)
    _fill(src, tile, n)
    N.check(L.sh_quantize_f32(src.handle, 0, n, 32767.0, 2, dst.handle, 0))
    w = 1 << 14
    marks = (0, (1 << 31) - w // 2, (1 << 32) - w // 2, (1 << 32) + 2048 * 3 + 1, n - w)
    for first in marks:
        idx = np.arange(first, first + w)
        want = np.trunc(32767.0 * tile[idx % len(tile)].astype(np.float64)).astype(np.int16)
        assert np.array_equal(dst.download(np.int16, w, 2 * first), want), first
    assert list(dst.download(np.int16, 64, 0)) == O.quantise([float(x) for x in tile[:64]])
    N.check(L.sh_quantize_clip_f32(src.handle, n, 40000.0, dst.handle))
    for first in marks:
        idx = np.arange(first, first + w)
        want = np.clip(np.trunc(40000.0 * tile[idx % len(tile)].astype(np.float64)), -32768, 32767).astype(np.int16)
        assert np.array_equal(dst.download(np.int16, w, 2 * first), want), first
    src.free()
    dst.free()
