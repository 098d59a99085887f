"""Multi-process path on CPU (gloo, world_size 2): the voice table is partitioned with
synthesizer_amd.dist.shard_range, every rank renders the partial bus of its shard (here with the oracle --
no GPU in this container), the partial buses are summed by the collective, and the result must equal the
single-process bus.  This is the data flow of dist.DistVoiceBank with RCCL replaced by gloo -- and
test_dist_voice_bank_ring_over_gloo drives dist.DistVoiceBank ITSELF (slot ring, batching, back-pressure, flush)
with the oracle as the renderer and gloo as the collective.  (RCCL with >= 2 ranks: tests/test_gpu_multi.py,
which runs wherever two GPUs are visible.)"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nvoices, nframes, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as td
    from oracle import synth_oracle as O
    from synthesizer_amd.dist import shard_range
    from synthesizer_amd.workloads import additive_voices
    td.init_process_group("gloo", rank=rank, world_size=world)
    voices, gains = additive_voices(O, nvoices, 48000, seed=5, partials=4)
    lo, hi = shard_range(nvoices, rank, world)
    part = np.array(O.mix_bus([v.take(nframes) for v in voices[lo:hi]], gains[lo:hi]), dtype=np.float64)
    t = torch.from_numpy(part.copy())
    td.reduce(t, dst=0, op=td.ReduceOp.SUM)            # float64 partial buses summed to rank 0
    # the rendezvous used for the RCCL unique id: 128 opaque bytes broadcast from rank 0 (the channel bench.py hands to dist.init:
    # synthesizer_amd.dist.Rendezvous, the TCP star beside MASTER_PORT -- gloo's own store sits ON MASTER_PORT)
    from synthesizer_amd.dist import Rendezvous
    rdzv = Rendezvous(rank, world)
    ident = rdzv.as_broadcast()(bytes(range(128)) if rank == 0 else None, rank, world, 128)
    rdzv.barrier()
    rdzv.close()
    td.barrier()
    if rank == 0:
        q.put((t.numpy().copy(), ident, (lo, hi)))
    else:
        q.put((None, ident, (lo, hi)))
    td.destroy_process_group()


def test_voice_sharded_bus_equals_single_process():
    import torch.multiprocessing as mp
    from oracle import synth_oracle as O
    from synthesizer_amd.workloads import additive_voices
    nvoices, nframes, world = 13, 600, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nvoices, nframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bus = next(r[0] for r in results if r[0] is not None)
    assert all(r[1] == bytes(range(128)) for r in results)
    assert sorted(r[2] for r in results) == [(0, 7), (7, 13)]
    voices, gains = additive_voices(O, nvoices, 48000, seed=5, partials=4)
    want = np.array(O.mix_bus([v.take(nframes) for v in voices], gains), dtype=np.float64)
    assert np.max(np.abs(bus - want)) < 1e-14


class _OracleGlooBackend:
    """DistVoiceBank backend for CPU: the oracle renders this rank's shard, gloo sums the partial buses."""

    def __init__(self, voices, gains, td, torch, rank, total_frames):
        from oracle import synth_oracle as O
        self.td, self.torch, self.rank = td, torch, rank
        self.log, self.results = [], []
        # the shard's float64 partial bus for the whole run (the oracle's generators start at sample 0)
        self.full = np.array(O.mix_bus([v.take(total_frames) for v in voices], gains), dtype=np.float64).reshape(-1, 2)

    def nslots(self):
        return 4

    def alloc(self, nbytes):
        return np.zeros(nbytes, dtype=np.uint8)

    def view(self, buf, offset, nbytes):
        return buf[offset:offset + nbytes]

    def render(self, nframes, start, bus_f32, bus_f64):
        part = self.full[start:start + nframes].reshape(-1)
        self.log.append(("render", start))
        if bus_f64 is not None:
            bus_f64.view(np.float64)[:] = part
        if bus_f32 is not None:
            bus_f32.view(np.float32)[:] = part.astype(np.float32)

    def wait_slot(self, slot):
        self.log.append(("wait", slot))

    def reduce_async(self, bus_f64, nvalues, root, bus_f32, slot):
        self.log.append(("reduce", slot, nvalues, root))
        t = self.torch.from_numpy(bus_f64.view(np.float64)[:nvalues])
        self.td.reduce(t, dst=root, op=self.td.ReduceOp.SUM)
        if self.rank == root:
            bus_f32.view(np.float32)[:nvalues] = bus_f64.view(np.float64)[:nvalues].astype(np.float32)
            self.results.append(bus_f32.view(np.float32)[:nvalues].copy())

    def sync(self):
        pass

    def download(self, buf, nvalues):
        return buf.view(np.float32)[:nvalues].copy()


def _ring_worker(rank, world, port, nvoices, nframes, batch, nblocks, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as td
    from oracle import synth_oracle as O
    from synthesizer_amd import dist
    from synthesizer_amd.workloads import additive_voices
    td.init_process_group("gloo", rank=rank, world_size=world)
    voices, gains = additive_voices(O, nvoices, 48000, seed=5, partials=4)
    lo, hi = dist.shard_range(nvoices, rank, world)
    backend = _OracleGlooBackend(voices[lo:hi], gains[lo:hi], td, torch, rank, (nblocks + 1) * nframes)
    bank = dist.DistVoiceBank(voices, gains, rank, world, batch=batch, backend=backend)
    assert (bank.lo, bank.hi) == (lo, hi) and bank.batch == batch
    for s in range(nblocks):
        bank.render_device(nframes, s * nframes)
    bank.flush()
    bank.sync()
    one = bank.render(nframes, nblocks * nframes)                 # the blocking form: root gets the array, the others None
    td.barrier()
    q.put((rank, backend.log, np.concatenate(backend.results) if backend.results else None, one))
    td.destroy_process_group()


def test_dist_voice_bank_ring_over_gloo():
    """dist.DistVoiceBank itself, world 2: four slots x batch 3, fourteen blocks (the ring wraps and ends mid-slot)."""
    import torch.multiprocessing as mp
    from oracle import synth_oracle as O
    from synthesizer_amd.workloads import additive_voices
    nvoices, nframes, world, batch, nblocks = 9, 200, 2, 3, 14
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, nvoices, nframes, batch, nblocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=240) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    voices, gains = additive_voices(O, nvoices, 48000, seed=5, partials=4)
    total = (nblocks + 1) * nframes
    want = np.array(O.mix_bus([v.take(total) for v in voices], gains), dtype=np.float64)
    (_, log0, bus0, one0), (_, log1, bus1, one1) = results
    assert bus1 is None and one1 is None
    got = bus0.reshape(-1, 2)
    assert got.shape == (total, 2)
    assert np.max(np.abs(got - want.astype(np.float32))) <= 1.2e-7              # float64 sum in another order, one rounding
    assert np.array_equal(one0, got[nblocks * nframes:])
    for log in (log0, log1):
        reduces = [e for e in log if e[0] == "reduce"]
        # 4 full slots of 3 blocks, the partly filled fifth (2 blocks, slot 0 again), then render()'s single block
        assert [e[1] for e in reduces] == [0, 1, 2, 3, 0, 1]
        assert [e[2] for e in reduces] == [batch * nframes * 2] * 4 + [2 * nframes * 2, nframes * 2]
        # back-pressure: the first render into a slot is preceded by wait_slot(slot)
        slot_of_block = [(s // batch) % 4 for s in range(nblocks)] + [1]
        it = iter(log)
        for s, k in enumerate(slot_of_block):
            if s % batch == 0 or s == nblocks:
                prev = None
                for e in it:
                    if e[0] == "render":
                        assert prev == ("wait", k), (s, k, prev)
                        break
                    prev = e
            else:
                for e in it:
                    if e[0] == "render":
                        break


def test_tcp_rendezvous_broadcast():
    """The torch-free rendezvous (MASTER_ADDR:MASTER_PORT+1) used when no process group exists."""
    import multiprocessing as mp
    from synthesizer_amd.dist import _tcp_broadcast
    port = _free_port()
    os.environ["SYNTHHIP_RDZV_PORT"] = str(port)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    payload = os.urandom(128)

    def client(q):
        q.put(_tcp_broadcast(None, 1, 2, 128))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=client, args=(q,))
    p.start()
    assert _tcp_broadcast(payload, 0, 2, 128) == payload
    assert q.get(timeout=60) == payload
    p.join(timeout=30)
    del os.environ["SYNTHHIP_RDZV_PORT"]


def test_resample_ranges_and_spans_cover_exactly():
    """Output-range sharding of Sample.resample: the ranges partition the output; every range's input span holds
    exactly the frames its outputs interpolate (index arithmetic of the oracle), starts on the 16-frame grid and
    overlaps its neighbour by the halo only."""
    import math
    from oracle import pcm_oracle as P
    from synthesizer_amd import dist
    for in_frames, inrate, outrate in ((50021, 96000, 44100), (50021, 44100, 48000), (1000, 8000, 8001), (777, 3, 7),
                                       (5, 48000, 44100), (123457, 48000, 16000)):
        nout = P.ratecv_out_frames(in_frames, inrate, outrate)
        for world in (1, 2, 3, 8):
            ranges = dist.resample_ranges(nout, world)
            assert len(ranges) == world and ranges[0][0] == 0
            assert sum(c for _, c in ranges) == nout
            for (f0, c0), (f1, _c1) in zip(ranges, ranges[1:]):
                assert f0 + c0 == f1 and (f1 % 256 == 0 or f1 == nout)
            g = math.gcd(inrate, outrate)
            inr, outr = inrate // g, outrate // g
            for first, count in ranges:
                if not count:
                    continue
                lo, n = dist.resample_span(in_frames, inrate, outrate, first, count)
                j_first = -(-first * inr // outr)
                j_last = -(-(first + count - 1) * inr // outr)
                assert lo % 16 == 0 and lo <= max(0, j_first - 1) < lo + 16
                assert lo + n - 1 == j_last and lo + n <= in_frames
    with pytest.raises(ValueError):
        dist.resample_span(1000, 48000, 44100, 900, 100)       # beyond the 919 output frames


class _LaggedFakeBackend:
    """The held-back reduce path of DistVoiceBank (reduce_lagged / mark_slot) without a GPU: records the calls."""

    def __init__(self):
        self.log = []

    def nslots(self):
        return 4

    def alloc(self, nbytes):
        return np.zeros(nbytes, dtype=np.uint8)

    def view(self, buf, offset, nbytes):
        return buf[offset:offset + nbytes]

    def render(self, nframes, start, bus_f32, bus_f64):
        self.log.append(("render", start))

    def wait_slot(self, slot):
        self.log.append(("wait", slot))

    def mark_slot(self, slot):
        self.log.append(("mark", slot))

    def reduce_lagged(self, bus_f64, nvalues, root, bus_f32, slot):
        self.log.append(("reduce", slot, nvalues))

    def sync(self):
        self.log.append(("sync",))


def test_sync_releases_the_full_slots_whose_reduce_is_held_back():
    """render whole batches, then sync() WITHOUT flush(): the held reduces must have been enqueued in front of the wait."""
    from oracle import synth_oracle as O
    from synthesizer_amd import dist
    from synthesizer_amd.workloads import additive_voices
    voices, gains = additive_voices(O, 4, 48000, seed=1, partials=2)
    be = _LaggedFakeBackend()
    bank = dist.DistVoiceBank(voices, gains, 0, 2, batch=2, backend=be)
    for s in range(4):                                   # two full slots, nothing partly filled
        bank.render_device(100, s * 100)
    assert [e for e in be.log if e[0] == "reduce"] == [] or len([e for e in be.log if e[0] == "reduce"]) < 2     # (held back)
    bank.sync()
    reduces = [e for e in be.log if e[0] == "reduce"]
    assert [e[1] for e in reduces] == [0, 1] and all(e[2] == 2 * 100 * 2 for e in reduces)
    assert be.log[-1] == ("sync",) and be.log.index(reduces[-1]) < len(be.log) - 1
    assert bank._held == []
    # a partly filled slot is flush()'s business: sync() alone leaves it
    bank.render_device(100, 400)
    bank.sync()
    assert len([e for e in be.log if e[0] == "reduce"]) == 2
    bank.flush()
    assert len([e for e in be.log if e[0] == "reduce"]) == 3
