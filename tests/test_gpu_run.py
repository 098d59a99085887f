"""A run of blocks per call (sh_bank_render_run / VoiceBank.render_run): one crossing of the ABI, one launch per stretch of ring
buffers that lie back to back -- the blocks are the ones block-by-block renders deliver."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 48000


def _bank(nv, seed=3):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.workloads import additive_voices
    gv, gains = additive_voices(G, nv, SR, seed=seed, partials=16, adsr={"sustain": 1.0e6})
    return VoiceBank(gv, gains=gains)


@pytest.mark.parametrize("nv,nframes", [(64, 48000), (1024, 12000), (5, 4096)])
def test_run_of_blocks_equals_block_by_block(gpu, nv, nframes):
    N = gpu
    nblocks, start = 7, 3 * nframes + 128
    single = _bank(nv)
    want = [single.render(nframes, start + k * nframes) for k in range(nblocks)]

    # ring buffers that are separate allocations: one launch per block, the very launches of render() -- bit for bit
    bank = _bank(nv)
    ring = [N.DeviceBuffer(nframes * 8) for _ in range(3)]
    bank.render_run(nframes, 3, start, ring=ring)
    N.sync()
    for k in range(3):
        assert np.array_equal(ring[k].download(np.float32, nframes * 2).reshape(nframes, 2), want[k]), k

    # a contiguous ring of 4 slots, 7 blocks: stretches [0..3], [4..6] (the wrap ends a stretch) -- merged launches may sum the voice
    # groups' partial buses in another order: equal to float32 rounding of a float64 sum taken in another order
    cont = bank.make_ring(nframes, 4)
    bank.render_run(nframes, nblocks, start, ring=cont)
    N.sync()
    for slot in range(4):
        k = slot + 4 if slot + 4 < nblocks else slot           # what the slot holds after the wrap
        got = cont[slot].download(np.float32, nframes * 2).reshape(nframes, 2)
        assert np.max(np.abs(got.astype(np.float64) - want[k])) <= 1.5e-7, (slot, k)

    # the PCM ring (saturated int16 stereo), contiguous: within one step of the quantised float32 bus at the rare sample whose
    # float32 rounding differs (another summation order)
    pcm = bank.make_ring(nframes, 4, bytes_per_frame=4)
    bank.render_run(nframes, 4, start, pcm_ring=pcm)
    N.sync()
    for k in range(4):
        got = pcm[k].download(np.int16, nframes * 2).reshape(nframes, 2).astype(np.int32)
        ref = np.clip(np.trunc(32767.0 * want[k].astype(np.float64)), -32768, 32767).astype(np.int32)
        d = np.abs(got - ref)
        assert d.max() <= 1 and np.mean(d != 0) < 2e-3, (k, int(d.max()), float(np.mean(d != 0)))


def test_run_arguments(gpu):
    N = gpu
    bank = _bank(8)
    with pytest.raises(ValueError):
        bank.render_run(1000, 2, ring=[N.DeviceBuffer(1000 * 8 - 8)])          # a slot too small
    with pytest.raises(ValueError):
        bank.render_run(1000, 2, ring=bank.make_ring(1000, 2), pcm_ring=bank.make_ring(1000, 3, 4))
    out = bank.render_run(1000, 3)                                              # no ring given: a fresh contiguous one
    N.sync()
    assert len(out) == 3
    want = np.concatenate([_bank(8).render(1000, k * 1000) for k in range(3)])
    got = np.concatenate([b.download(np.float32, 2000).reshape(1000, 2) for b in out])
    assert np.max(np.abs(got.astype(np.float64) - want)) <= 1.5e-7
    assert bank.render_run(1000, 0, ring=bank.make_ring(1000, 1)) is not None    # nothing to do
