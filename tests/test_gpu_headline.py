"""GPU parity on the workloads the bench line quotes, in the launches it times.

* the headline: 1024 additive voices (Harmonics x16 + the bench's ADSR with the held sustain), seed 0, a 48 000-frame block
  five seconds into the notes -- the eight-frames-per-lane lean kernel of a split launch, as the third launch of a pipelined
  run (records resolved two launches ahead, the fold taken over by a later launch) -- against the C oracle run from frame 0;
* BASELINE configs[3] as far as one GPU goes: the 8192-voice table -- eight voice shards' float64 partial buses summed in
  rank order against the whole bank's, and a 512-voice stride subset against the C oracle.
"""
import numpy as np
import pytest

from tests.helpers import oracle_bus_window, rms

pytestmark = pytest.mark.gpu

SR = 48000
ADSR_BENCH = {"sustain": 1.0e6}           # bench.py's: attack 0.01, decay 0.05, the sustain spans any run, release 0.2
WINDOW = 4096


def test_headline_block_in_the_timed_window_vs_oracle(gpu):
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.workloads import additive_voices
    import ctypes
    gv, gains = additive_voices(G, 1024, SR, seed=0, partials=16, adsr=ADSR_BENCH)
    bank = VoiceBank(gv, gains=gains)
    block = SR
    ring = [N.DeviceBuffer(block * 8) for _ in range(4)]
    pcm_ring = [N.DeviceBuffer(block * 4) for _ in range(4)]
    for k in (3, 4, 5):                                      # a run: block 5 is its third launch
        bank.render_device(block, k * block, bus_f32=ring[k & 3])
    nf, ng = ctypes.c_uint32(), ctypes.c_uint32()
    N.check(N.lib().sh_bank_launch_stats(bank._bank.handle, ctypes.byref(nf), ctypes.byref(ng)))
    assert (nf.value, ng.value) == (1024, 0)                  # every voice through the lean loop: the launch the bench times
    got = ring[5 & 3].download(np.float32, block * 2).reshape(block, 2)
    for k in (3, 4, 5):
        bank.render_pcm_device(block, k * block, pcm=pcm_ring[k & 3])
    pcm = pcm_ring[5 & 3].download(np.int16, block * 2).reshape(block, 2)
    want = oracle_bus_window("additive", 1024, 0, ADSR_BENCH, range(1024), 5 * block, WINDOW)
    err = rms(got[:WINDOW], want)
    assert err <= 1e-6 / 3, err
    assert np.max(np.abs(got[:WINDOW] - want)) < 2e-7
    # int16: the rule is exact -- the PCM block IS trunc(32767 * float32 bus) of the library's own bus, saturated --
    own = np.clip(np.trunc(32767.0 * got.astype(np.float64)), -32768, 32767).astype(np.int16)
    assert np.array_equal(pcm, own)
    # ... and against quantise(float32(oracle bus)) it can differ by one step where the two float64 sums (1024 voices, another
    # order, ~1e-8 apart) round to neighbouring float32 values on either side of an integer: rare, never more than 1
    ref = np.clip(np.trunc(32767.0 * want.astype(np.float32).astype(np.float64)), -32768, 32767).astype(np.int16)
    d = np.abs(pcm[:WINDOW].astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 2e-3, (int(d.max()), float(np.mean(d != 0)))
    # the same frames by a single render of just the window (another launch shape: four frames per lane, no run)
    alone = VoiceBank(additive_voices(G, 1024, SR, seed=0, partials=16, adsr=ADSR_BENCH)[0], gains=gains).render(WINDOW, start=5 * block)
    assert rms(alone, want) <= 1e-6 / 3
    assert np.max(np.abs(alone.astype(np.float64) - got[:WINDOW])) < 1e-7


def test_config4_voice_table_on_one_gpu(gpu):
    """8192 voices (BASELINE configs[3]): what each of eight ranks would render, on one GPU -- the shards' float64 partial buses
    summed on the host in rank order equal the whole bank's float64 bus (1e-12), in the steady state and from the start of the
    notes; the float32 bus of a 512-voice stride subset equals the C oracle's."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.dist import shard_range
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.workloads import additive_voices
    nv, world = 8192, 8
    gv, gains = additive_voices(G, nv, SR, seed=0, partials=16, adsr=ADSR_BENCH)
    whole = VoiceBank(gv, gains=gains)
    shards = []
    for r in range(world):
        lo, hi = shard_range(nv, r, world)
        assert hi - lo == 1024
        shards.append(VoiceBank(gv[lo:hi], gains=gains[lo:hi]))
    for start, n in ((5 * SR, 48000), (0, 12000), (7 * SR + 123, 8192)):
        b64 = N.DeviceBuffer(n * 16)
        b32 = N.DeviceBuffer(n * 8)
        whole.render_device(n, start, bus_f32=b32, bus_f64=b64)
        w64 = b64.download(np.float64, n * 2).reshape(n, 2)
        w32 = b32.download(np.float32, n * 2).reshape(n, 2)
        acc = np.zeros((n, 2), dtype=np.float64)
        for sh in shards:                                    # rank order, like ncclReduce's result up to the order of the sum
            sh.render_device(n, start, bus_f32=None, bus_f64=b64)
            acc += b64.download(np.float64, n * 2).reshape(n, 2)
        assert np.max(np.abs(acc - w64)) <= 1e-12, (start, float(np.max(np.abs(acc - w64))))
        assert np.array_equal(w64.astype(np.float32), w32)     # the float32 bus is ONE rounding of the float64 bus
        b64.free()
        b32.free()
    # 512 voices of the table (every 16th) against the C oracle, a window five seconds in
    sub = list(range(0, nv, 16))
    bank = VoiceBank([gv[i] for i in sub], gains=[gains[i] for i in sub])
    got = bank.render(WINDOW, start=5 * SR)
    want = oracle_bus_window("additive", nv, 0, ADSR_BENCH, range(0, nv, 16), 5 * SR, WINDOW)
    assert rms(got, want) <= 1e-6 / 3
