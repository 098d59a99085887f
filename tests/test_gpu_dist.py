"""GPU: the RCCL leg of the voice-sharded path, as far as ONE GPU can exercise it -- librccl is dlopen()ed,
a 1-rank communicator is created on the library's stream, and the reduce / all-reduce / barrier entry
points run on the float64 partial bus.  (world_size 2 data flow: tests/test_dist_gloo.py on CPU; the
8-GPU run is the driver's.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_reduce(gpu):
    from synthesizer_amd import _native as N
    from synthesizer_amd import dist
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.workloads import additive_voices
    L = N.lib()
    dist.init(0, 1, broadcast=lambda payload, rank, world, n: payload)
    try:
        assert L.sh_dist_rank() == 0 and L.sh_dist_world() == 1
        voices, gains = additive_voices(G, 32, 48000, seed=2)
        bank = dist.DistVoiceBank(voices, gains, 0, 1)
        n = 4096
        ref = bank.local.render(n)
        bank._buffers(n)
        bank.local.render_device(n, 0, bus_f32=None, bus_f64=bank._bus64[0])
        N.check(L.sh_dist_reduce_bus(bank._bus64[0].handle, n * 2, 0))
        N.check(L.sh_dist_allreduce_bus(bank._bus64[0].handle, n * 2))
        N.check(L.sh_dist_barrier())
        N.check(L.sh_bus_finalize(bank._bus64[0].handle, n * 2, bank._bus32[0].handle))
        got = bank._bus32[0].download(np.float32, n * 2).reshape(n, 2)
        assert np.array_equal(got, ref)
        assert np.array_equal(bank.render(n), ref)
        with pytest.raises(ValueError):
            N.check(L.sh_dist_reduce_bus(bank._bus64[0].handle, n * 2, 3))
        # the pipelined form: reduce of block s on the communication stream while block s+1 renders
        nslots = L.sh_dist_slots()
        b64 = [N.DeviceBuffer(n * 16) for _ in range(nslots)]
        b32 = [N.DeviceBuffer(n * 8) for _ in range(nslots)]
        outs = []
        for s in range(2 * nslots + 1):
            k = s % nslots
            N.check(L.sh_dist_wait_slot(k))
            if s >= nslots:                      # slot reuse: fetch the block rendered nslots steps ago first
                N.sync()
                outs.append(b32[k].download(np.float32, n * 2).reshape(n, 2))
            bank.local.render_device(n, s * n, bus_f32=None, bus_f64=b64[k])
            N.check(L.sh_dist_reduce_bus_async(b64[k].handle, n * 2, 0, b32[k].handle, k))
        N.sync()
        for j, got_j in enumerate(outs):
            assert np.array_equal(got_j, bank.local.render(n, j * n)), j
        with pytest.raises(ValueError):
            N.check(L.sh_dist_reduce_bus_async(b64[0].handle, n * 2, 0, b32[0].handle, nslots))
        # DistVoiceBank's batched ring, forced through the multi-rank code path on this 1-rank communicator
        ring = dist.DistVoiceBank(voices, gains, 0, 1, batch=3)
        ring.world, ring.batch = 2, 3          # pretend there are peers: same calls, the reduce is the identity
        got_blocks = [ring.render_device(1000, s * 1000) for s in range(3 * nslots + 2)]    # wraps the ring, ends mid-slot
        ring.flush()
        N.sync()
        for s in (3 * nslots + 1, 3 * nslots, 3 * nslots - 1, 2 * 3):           # blocks whose slots have not been reused
            assert np.array_equal(got_blocks[s].download(np.float32, 2000).reshape(1000, 2), bank.local.render(1000, s * 1000)), s
        # the same with a bank large enough for several voice groups: the renders of a slot then form one run of the
        # two-stream pipeline (partial buses folded two launches on) which the slot's reduce has to end first
        big_v, big_g = additive_voices(G, 640, 48000, seed=6)
        big = dist.DistVoiceBank(big_v, big_g, 0, 1, batch=5)
        big.world, big.batch = 2, 5
        alone = dist.DistVoiceBank(additive_voices(G, 640, 48000, seed=6)[0], big_g, 0, 1).local
        blocks = [big.render_device(3000, s * 3000) for s in range(5 * nslots + 3)]
        big.flush()
        N.sync()
        for s in range(5 * nslots + 2, 5 * nslots + 2 - 5 * (nslots - 1), -1):    # blocks whose slots have not been reused
            assert np.array_equal(blocks[s].download(np.float32, 6000).reshape(3000, 2), alone.render(3000, s * 3000)), s
        # a table of notes (tile-classified launches, float64 partial bus out, blocks of real-time and of one-second length) down
        # the same ring: the lagged reduce keeps the run of renders -- and the tile sets resolved two launches ahead -- alive
        from synthesizer_amd.workloads import staggered_notes
        nv, ng = staggered_notes(G, 256, 48000, seed=3, period=0.5, notes=4)
        for blk in (2048, 24000):
            notes = dist.DistVoiceBank(nv, ng, 0, 1, batch=4)
            notes.world, notes.batch = 2, 4
            ref_bank = dist.DistVoiceBank(nv, ng, 0, 1).local
            before = N.debug_counters()
            nblk = 4 * nslots + 2
            got = [notes.render_device(blk, s * blk) for s in range(nblk)]
            notes.flush()
            N.sync()
            after = N.debug_counters()
            assert after["tiled_launches"] - before["tiled_launches"] == nblk
            assert after["tiled_predicted"] - before["tiled_predicted"] >= nblk - 4
            for s in range(nblk - 1, nblk - 1 - 4 * (nslots - 1), -1):
                assert np.array_equal(got[s].download(np.float32, blk * 2).reshape(blk, 2), ref_bank.render(blk, s * blk)), (blk, s)
    finally:
        dist.shutdown()
    assert L.sh_dist_world() == 0
