"""One rank of the >= 2-GPU RCCL test (tests/test_gpu_multi.py spawns WORLD_SIZE of these, one per GPU).

    RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment; argv[1] = output directory.

    python multi_gpu_worker.py --plumbing      prints what this process WOULD do (rank, world, GPU ordinal) and exits: no GPU touched

This script is the one place of the multi-GPU test that picks a device: the ordinal is LOCAL_RANK (dist.device_for_rank), selected
explicitly before anything else touches the library.  Every rank: TCP rendezvous of the ncclUniqueId (no torch), a DistVoiceBank with
batch 8 over several wraps of the slot ring, the blocking render, and its share of a range-sharded resample.  Root
writes the reduced buses (float32 and the float64 sums) for the parent to compare with the single-GPU render."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

NVOICES_PER_RANK, EXTRA, NFRAMES, BATCH, SEED = 96, 5, 3000, 8, 9
RESAMPLE = dict(frames=200_003, nch=2, width=2, inrate=44100, outrate=48000, seed=4)


def workload(world):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.workloads import additive_voices
    return additive_voices(G, NVOICES_PER_RANK * world + EXTRA, 48000, seed=SEED)       # uneven shards on purpose


def resample_input():
    r = RESAMPLE
    rng = np.random.default_rng(r["seed"])
    return rng.integers(-32768, 32768, r["frames"] * r["nch"], dtype=np.int16).tobytes()


def plumbing(env=None):
    """(rank, world, GPU ordinal) from the launcher's environment -- host logic, no GPU."""
    from synthesizer_amd import dist
    env = os.environ if env is None else env
    return {"rank": int(env.get("RANK", "0")), "world": int(env.get("WORLD_SIZE", "1")), "device": dist.device_for_rank(env)}


def main():
    if sys.argv[1] == "--plumbing":
        import json
        print(json.dumps(plumbing()))
        return
    out = Path(sys.argv[1])
    import json
    from synthesizer_amd import _native as N
    from synthesizer_amd import dist
    pl = plumbing()
    rank, world, device = pl["rank"], pl["world"], pl["device"]
    N.ensure_init(device)                             # THE device choice of this process
    info = N.device_info()
    assert info["device"] == device, (info, device)
    if world > 1:
        dist.init(rank, world)
    L = N.lib()
    ci = dist.comm_info()
    assert world == 1 or (L.sh_dist_rank() == rank and L.sh_dist_world() == world)
    assert world == 1 or (ci["communicator"] and ci["world"] == world and ci["rank"] == rank), ci     # what RCCL itself saw
    (out / ("rank_%d.json" % rank)).write_text(json.dumps({"rank": rank, "world": world, "device": device, "pci": N.device_pci(),
                                                           "name": info["name"], "rccl": ci}))
    voices, gains = workload(world)
    bank = dist.DistVoiceBank(voices, gains, rank, world, batch=BATCH)
    nslots = L.sh_dist_slots()
    nblocks = BATCH * nslots * 2 + 3                 # two wraps of the ring, ending mid-slot
    got32 = np.zeros((nblocks, NFRAMES, 2), dtype=np.float32)
    got64 = np.zeros((nblocks, NFRAMES, 2), dtype=np.float64)
    pending = []                                     # (block, slot, index in slot) not yet fetched

    def fetch():
        bank.flush()
        bank.sync()
        if rank == 0:
            for s, k, j in pending:
                v64, v32 = bank._views[k][j]
                got32[s] = v32.download(np.float32, NFRAMES * 2).reshape(NFRAMES, 2)
                if world > 1:                       # (a lone rank renders straight to float32: nothing to reduce)
                    got64[s] = v64.download(np.float64, NFRAMES * 2).reshape(NFRAMES, 2)
        pending.clear()

    for s in range(nblocks):
        k, j = bank._slot, bank._fill
        bank.render_device(NFRAMES, s * NFRAMES)
        pending.append((s, k, j))
        if len(pending) == (2 * BATCH if world > 1 else 1):   # two slots in flight while the next ones render, then collect
            fetch()
    fetch()
    one = bank.render(NFRAMES, nblocks * NFRAMES)     # the blocking form
    assert (one is None) == (rank != 0)
    if world > 1:
        N.check(L.sh_dist_barrier())
    # Sample.resample sharded by output range: no rank talks to another
    r = RESAMPLE
    first, pcm = dist.resample_shard(resample_input(), r["width"], r["nch"], r["inrate"], r["outrate"], rank, world)
    (out / ("resample_%d.bin" % rank)).write_bytes(pcm)
    (out / ("resample_%d.first" % rank)).write_text(str(first))
    if rank == 0:
        np.save(out / "bus32.npy", got32)
        np.save(out / "bus64.npy", got64)
        np.save(out / "one.npy", one)
    if world > 1:
        dist.shutdown()
    (out / ("done_%d" % rank)).write_text("ok")


if __name__ == "__main__":
    main()
