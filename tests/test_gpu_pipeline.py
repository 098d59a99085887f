"""GPU: the pipeline of consecutive renders (two HIP streams, records resolved two launches ahead, folds deferred) as seen
from outside -- whatever the interleaving of banks, buffers and block lengths, a bus buffer holds exactly what the LAST render
into it produces when rendered alone; and the render path neither allocates nor synchronises once it has seen its shapes.

Round 3: single-group (small) banks are pipelined too -- their launches write the caller's buffers themselves, so the library
has to keep two launches off the same memory -- and the alias check between banks covers folds that are in flight
(a ring of three bus buffers shared by two banks)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from synthesizer_amd.workloads import additive_voices, fm_voices

pytestmark = pytest.mark.gpu

SR = 48000
ROOT = Path(__file__).resolve().parent.parent


def _bank(G, VoiceBank, nv, seed, fm=False):
    v, g = (fm_voices(G, nv, SR, seed=seed) if fm else additive_voices(G, nv, SR, seed=seed, adsr={"sustain": 100.0}))
    return VoiceBank(v, gains=g)


@pytest.mark.parametrize("nv", [5, 64, 100])
def test_small_banks_are_pipelined_and_exact(gpu, nv):
    """A single-group bank: runs over rings of 4, 3, 2 buffers and over ONE buffer, float32 / float64 / PCM outputs, broken by
    jumps and by other calls -- every buffer equals the block rendered alone."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    block, nblocks = 3000, 14
    bank, ref = _bank(G, VoiceBank, nv, 7), _bank(G, VoiceBank, nv, 7)
    alone = [ref.render(block, start=s * block) for s in range(nblocks)]
    alone_pcm = [ref.render_pcm_device(block, s * block).download_bytes(block * 4) for s in range(nblocks)]
    read = lambda buf: buf.download(np.float32, block * 2).reshape(block, 2)
    for ring_n in (4, 3, 2, 1):
        ring = [N.DeviceBuffer(block * 8) for _ in range(ring_n)]
        holds = [None] * ring_n
        for s in range(nblocks):
            bank.render_device(block, s * block, bus_f32=ring[s % ring_n])
            holds[s % ring_n] = s
            if s == 8:                                             # a jump back: the run breaks, the buffers keep their blocks
                bank.render_device(block, 2 * block, bus_f32=ring[(s + 1) % ring_n])
                holds[(s + 1) % ring_n] = 2
        for k in range(ring_n):
            assert np.array_equal(read(ring[k]), alone[holds[k]]), (ring_n, k, holds[k])
    # float64 and PCM outputs through the same pipeline
    r64 = [N.DeviceBuffer(block * 16) for _ in range(4)]
    r16 = [N.DeviceBuffer(block * 4) for _ in range(4)]
    for s in range(8):
        bank.render_device(block, s * block, bus_f32=None, bus_f64=r64[s & 3])
    for k in range(4):
        got = r64[k].download(np.float64, block * 2).reshape(block, 2)
        assert np.array_equal(got.astype(np.float32), alone[4 + k]), k
    for s in range(8):
        bank.render_pcm_device(block, s * block, pcm=r16[s & 3])
    for k in range(4):
        assert r16[k].download_bytes(block * 4) == alone_pcm[4 + k], k
    # a buffer freed right after the render into it, zero-fills and uploads between the renders of a run
    for s in range(6):
        buf = N.DeviceBuffer(block * 8)
        bank.render_device(block, s * block, bus_f32=buf)
        if s % 2:
            buf.free()
        else:
            assert np.array_equal(read(buf), alone[s])
    a, b = N.DeviceBuffer(block * 8), N.DeviceBuffer(block * 8)
    bank.render_device(block, 0, bus_f32=a)
    b.zero()                                                       # main-stream work between two launches of what could be a run
    bank.render_device(block, block, bus_f32=b)                    # ... must not be overtaken by the render on the second stream
    assert np.array_equal(read(b), alone[1]) and np.array_equal(read(a), alone[0])


def test_banks_sharing_rings_of_buffers_last_write_wins(gpu):
    """Random plans over a ring of three buffers shared by four banks (two with several voice groups: deferred folds, folds
    taken over by a launch on the other stream; two single-group: direct writes): every render goes to the next ring buffer
    whoever issues it.  Whenever the plan reads a buffer it must hold the block of the last render into it, nothing else."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    block, nblocks = 5000, 12
    specs = [(640, 11, False), (384, 12, False), (48, 13, False), (96, 14, True)]
    banks = [_bank(G, VoiceBank, nv, seed, fm) for nv, seed, fm in specs]
    refs = [_bank(G, VoiceBank, nv, seed, fm) for nv, seed, fm in specs]
    alone = [[r.render(block, start=s * block) for s in range(nblocks)] for r in refs]
    read = lambda buf: buf.download(np.float32, block * 2).reshape(block, 2)
    rng = np.random.default_rng(77)
    for ring_n in (3, 4, 2):
        ring = [N.DeviceBuffer(block * 8) for _ in range(ring_n)]
        holds = [None] * ring_n
        pos = [0, 0, 0, 0]
        slot = 0
        for step in range(160):
            i = int(rng.integers(0, len(banks)))
            run = int(rng.integers(1, 5))
            for _ in range(run):
                s = pos[i] % nblocks
                banks[i].render_device(block, s * block, bus_f32=ring[slot])
                holds[slot] = (i, s)
                pos[i] = s + 1
                slot = (slot + 1) % ring_n
            if rng.integers(0, 6) == 0:
                k = int(rng.integers(0, ring_n))
                if holds[k] is not None:
                    assert np.array_equal(read(ring[k]), alone[holds[k][0]][holds[k][1]]), (ring_n, step, k, holds[k])
        for k in range(ring_n):
            if holds[k] is not None:
                assert np.array_equal(read(ring[k]), alone[holds[k][0]][holds[k][1]]), (ring_n, "end", k, holds[k])


def test_render_path_neither_allocates_nor_synchronises_after_warm_up(gpu):
    """sh_debug_counters: a first pass over mixed block lengths (512 .. 48 000 frames, from the start of the notes -- segmented
    transition launches -- and from their steady state, float32 and PCM outputs, a small bank beside a large one) may grow the
    partial-bus ring and the record sets; a second pass over the same shapes must leave the driver alone: no hipMalloc, no
    hipFree, no host-side stream synchronisation of the library's own."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    big, small = _bank(G, VoiceBank, 1024, 3), _bank(G, VoiceBank, 64, 4)
    lengths = (512, 4096, 48000, 777, 16384, 48000, 30000)
    ring = [N.DeviceBuffer(48000 * 8) for _ in range(4)]
    pcm = [N.DeviceBuffer(48000 * 4) for _ in range(4)]

    def one_pass():
        k = 0
        for base in (0, 5 * SR):                       # from frame 0 (attack, decay: segmented launches) and from the sustain
            for n in lengths:
                for rep in range(3):
                    big.render_device(n, base + rep * n, bus_f32=ring[k & 3])
                    small.render_device(n, base + rep * n, bus_f32=ring[(k + 1) & 3])
                    big.render_pcm_device(n, base + rep * n, pcm=pcm[k & 3])
                    k += 1
        N.sync()

    one_pass()
    before = N.debug_counters()
    one_pass()
    one_pass()
    after = N.debug_counters()
    for key in ("device_allocs", "device_frees", "stream_syncs"):
        assert after[key] == before[key], (key, before, after)


_CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices
N.ensure_init(0)
out = {}
for nv in (64, 640):
    v, g = additive_voices(G, nv, 48000, seed=5, adsr={"sustain": 100.0})
    bank = VoiceBank(v, gains=g)
    ring = [N.DeviceBuffer(6000 * 8) for _ in range(4)]
    blocks = []
    for s in range(9):
        bank.render_device(6000, s * 6000, bus_f32=ring[s & 3])
        if s >= 3:
            pass
    for s in range(9, 13):
        bank.render_device(6000, s * 6000, bus_f32=ring[s & 3])
    out["nv%%d" %% nv] = np.stack([ring[s & 3].download(np.float32, 12000) for s in range(9, 13)])
# a table of notes (tile-classified launches): real-time chunks -- the merged kernel -- and longer blocks
from synthesizer_amd.workloads import staggered_notes
v, g = staggered_notes(G, 256, 48000, seed=2, period=0.5, notes=4)
bank = VoiceBank(v, gains=g)
for blk in (3000, 20000):
    ring = [N.DeviceBuffer(blk * 8) for _ in range(4)]
    for s in range(9):
        bank.render_device(blk, s * blk, bus_f32=ring[s & 3])
    out["notes%%d" %% blk] = np.stack([ring[s & 3].download(np.float32, blk * 2) for s in range(5, 9)])
np.savez(sys.argv[1], **out)
'''


def test_knobs_change_no_result(gpu, tmp_path):
    """SYNTHHIP_NO_SMALL_PIPELINE / SYNTHHIP_NO_OVERLAP / SYNTHHIP_NO_SPECULATION / SYNTHHIP_NO_LADDER select
    other schedules of the same work: the buses -- of small and large lock-step banks and of a table of notes -- are bit-identical
    with the default's."""
    outs = {}
    for name, env in (("default", {}), ("no_small", {"SYNTHHIP_NO_SMALL_PIPELINE": "1"}), ("no_overlap", {"SYNTHHIP_NO_OVERLAP": "1"}),
                      ("no_spec", {"SYNTHHIP_NO_SPECULATION": "1"}), ("no_ladder", {"SYNTHHIP_NO_LADDER": "1"})):
        path = tmp_path / (name + ".npz")
        clean = {k: v for k, v in os.environ.items() if not k.startswith("SYNTHHIP_") or k in ("SYNTHHIP_LIB", "SYNTHHIP_DEVICE")}
        p = subprocess.run([sys.executable, "-c", _CHILD % str(ROOT), str(path)], env=dict(clean, **env),
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[name] = np.load(path)
    for name in outs:
        for key in ("nv64", "nv640", "notes3000", "notes20000"):
            assert np.array_equal(outs[name][key], outs["default"][key]), (name, key)
