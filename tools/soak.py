"""Soak: a long stream (speculative launch records, deferred folds, lean/general classification changing block by block)
against independent random-access renders of sampled blocks -- must be bit-identical."""
import sys

sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices, fm_voices

N.ensure_init(0)
SR = 48000
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 600
block = 48000
rng = np.random.default_rng(1)
for name, (voices, gains) in (("additive", additive_voices(G, 1024, SR, seed=0, adsr={"sustain": nblocks * 0.7, "release": nblocks * 0.2})),
                              ("fm", fm_voices(G, 512, SR, seed=1))):
    bank = VoiceBank(voices, gains=gains)
    keep = sorted(set(rng.integers(0, nblocks, 24).tolist() + [0, 1, 2, nblocks - 1]))
    bufs = {s: N.DeviceBuffer(block * 8) for s in keep}
    scratch = N.DeviceBuffer(block * 8)
    for s in range(nblocks):
        bank.render_device(block, s * block, bus_f32=bufs.get(s, scratch))
    N.sync()
    bad = 0
    for s in keep:
        got = bufs[s].download(np.float32, block * 2)
        ref = bank.render(block, start=s * block).reshape(-1)          # random access: records prepared by a kernel
        if not np.array_equal(got, ref):
            bad += 1
            print(name, "block", s, "differs: max", float(np.abs(got - ref).max()))
    print(name, "blocks checked", len(keep), "mismatches", bad)

# a table of notes (tile-classified launches): a stream of one-second blocks and one of real-time chunks (sets resolved two launches
# ahead, chunk ranges that move, the merged kernel) against random-access renders of sampled blocks (sets resolved in front)
from synthesizer_amd.workloads import staggered_notes
voices, gains = staggered_notes(G, 1024, SR, seed=0, period=1.0, notes=12)
bank = VoiceBank(voices, gains=gains)
for blk, n in ((48000, 11), (4096, 11 * 48000 // 4096)):
    keep = sorted(set(rng.integers(0, n, 16).tolist() + [0, 1, 2, n - 1]))
    bufs = {s: N.DeviceBuffer(blk * 8) for s in keep}
    scratch = N.DeviceBuffer(blk * 8)
    for s in range(n):
        bank.render_device(blk, s * blk, bus_f32=bufs.get(s, scratch))
    N.sync()
    bad = 0
    for s in keep:
        got = bufs[s].download(np.float32, blk * 2)
        ref = bank.render(blk, start=s * blk).reshape(-1)
        if not np.array_equal(got, ref):
            bad += 1
            print("notes", blk, "block", s, "differs: max", float(np.abs(got - ref).max()))
    print("notes, blocks of", blk, "checked", len(keep), "mismatches", bad, "tile-classified launches so far", N.debug_counters()["tiled_launches"])
