"""RCCL with MORE THAN ONE rank: runs wherever >= 2 GPUs are visible (the driver's 8-GPU box), skips -- with the
reason -- on a 1-GPU lease.  World sizes 2, 4 and 8 as the box allows: one process per GPU (tests/multi_gpu_worker.py),
TCP rendezvous, DistVoiceBank(batch=8) over two wraps of the slot ring; root's bus must equal the single-GPU bus up to
the float64 summation order (<= 1e-12 before the one rounding to float32), and the range-sharded resample must
concatenate to audioop.ratecv of the whole input, bit for bit."""
import audioop
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worlds(ngpus):
    return [w for w in (2, 4, 8) if w <= ngpus]


def test_worker_dry_run_on_one_gpu(gpu, tmp_path):
    """The worker script itself with WORLD_SIZE=1 (no communicator): everything but the multi-rank exchange runs on any
    box, so a typo in the worker cannot hide until an 8-GPU node shows up."""
    sys.path.insert(0, str(ROOT / "tests"))
    import multi_gpu_worker as W
    from synthesizer_amd.mixer import VoiceBank
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "multi_gpu_worker.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    import json
    me = json.loads((tmp_path / "rank_0.json").read_text())
    assert me["device"] == 0 and me["world"] == 1 and me["rccl"]["world"] == 1 and not me["rccl"]["communicator"] and len(me["pci"]) >= 7, me
    got32, one = np.load(tmp_path / "bus32.npy"), np.load(tmp_path / "one.npy")
    voices, gains = W.workload(1)
    alone = VoiceBank(voices, gains=gains)
    n = W.NFRAMES
    for s in (0, 1, got32.shape[0] - 1):
        assert np.array_equal(got32[s], alone.render(n, s * n)), s
    assert np.array_equal(one, alone.render(n, got32.shape[0] * n))
    r = W.RESAMPLE
    want_pcm = audioop.ratecv(W.resample_input(), r["width"], r["nch"], r["inrate"], r["outrate"], None)[0]
    assert (tmp_path / "resample_0.bin").read_bytes() == want_pcm


def test_voice_shards_reduced_by_rccl_across_gpus(gpu, tmp_path):
    ngpus = gpu.lib().sh_device_count()
    if ngpus < 2:
        pytest.skip("needs >= 2 GPUs for a multi-rank RCCL communicator; this box shows %d (1-rank RCCL: tests/test_gpu_dist.py, "
                    "world-2 ring logic over gloo: tests/test_dist_gloo.py)" % ngpus)
    sys.path.insert(0, str(ROOT / "tests"))
    import multi_gpu_worker as W
    from synthesizer_amd import _native as N
    from synthesizer_amd.mixer import VoiceBank
    for world in _worlds(ngpus):
        out = tmp_path / ("w%d" % world)
        out.mkdir()
        port = _free_port()
        procs = []
        for r in range(world):
            from synthesizer_amd import dist
            env = dist.rank_env(r, world, port)               # (what bench.py's own launcher gives its ranks)
            procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "multi_gpu_worker.py"), str(out)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        logs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            logs.append(o)
        assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
        assert all((out / ("done_%d" % r)).exists() for r in range(world))
        # every rank on a GPU of its own, and RCCL saw all of them
        import json
        ranks = [json.loads((out / ("rank_%d.json" % r)).read_text()) for r in range(world)]
        assert [q["device"] for q in ranks] == list(range(world)), ranks
        assert len({q["pci"] for q in ranks}) == world, ranks
        assert all(q["rccl"]["communicator"] and q["rccl"]["world"] == world and q["rccl"]["rank"] == q["rank"] for q in ranks), ranks
        # the single-GPU render of the same voice table
        voices, gains = W.workload(world)
        alone = VoiceBank(voices, gains=gains)
        got32, got64, one = np.load(out / "bus32.npy"), np.load(out / "bus64.npy"), np.load(out / "one.npy")
        nblocks, n = got32.shape[0], W.NFRAMES
        b64 = N.DeviceBuffer(n * 16)
        for s in range(nblocks + 1):
            alone.render_device(n, s * n, bus_f32=None, bus_f64=b64)
            want = b64.download(np.float64, n * 2).reshape(n, 2)
            if s == nblocks:
                assert np.max(np.abs(one.astype(np.float64) - want)) <= 6e-8 * max(1.0, float(np.max(np.abs(want)))), world
                continue
            assert np.max(np.abs(got64[s] - want)) <= 1e-12, (world, s)                  # another summation order, float64
            assert np.array_equal(got32[s], got64[s].astype(np.float32)), (world, s)      # ... and ONE rounding, on root
        b64.free()
        # resample: the ranks' byte strings in rank order == audioop.ratecv of the whole input
        r = W.RESAMPLE
        firsts = [int((out / ("resample_%d.first" % k)).read_text()) for k in range(world)]
        parts = [(out / ("resample_%d.bin" % k)).read_bytes() for k in range(world)]
        fb = r["width"] * r["nch"]
        assert firsts == [sum(len(p) for p in parts[:k]) // fb for k in range(world)]
        want_pcm = audioop.ratecv(W.resample_input(), r["width"], r["nch"], r["inrate"], r["outrate"], None)[0]
        assert b"".join(parts) == want_pcm, world
