"""bench.py's multi-rank plumbing on CPU: the launch convention, the rendezvous, the device check, the gather -- everything of an
N-GPU run except the GPUs (SYNTHHIP_BENCH_DRY_RUN=1: renders are 20-us pauses, the line says "data": "dry-run").

Why this exists: the scaling curve is the one part of BASELINE.json's north_star this build cannot measure on its one-GPU leases.  The first
8-GPU node the driver gets must produce a curve whether it starts the bench as `python bench.py --gpus 8` or under
`python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`; both forms are exercised here with 2 ranks.
"""
import json
import multiprocessing as mp
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(cmd, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SYNTHHIP_RDZV_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=timeout)


def _line(p):
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])        # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_more_gpus_than_the_node_has_is_a_one_line_refusal():
    p = _run([sys.executable, "bench.py", "--gpus", "64"])
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])
    assert p.stdout.strip() == ""
    err = [x for x in p.stderr.splitlines() if x.strip()]
    assert len(err) == 1 and err[0].startswith("bench.py: --gpus 64, but this node shows"), p.stderr
    assert "Traceback" not in p.stderr


def test_gpus_must_match_the_launchers_world():
    p = _run([sys.executable, "bench.py", "--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "SYNTHHIP_BENCH_DRY_RUN": "1"})
    assert p.returncode == 2 and "WORLD_SIZE is 2" in p.stderr and "Traceback" not in p.stderr


def test_rank_env_maps_eight_ranks_to_eight_ordinals():
    from synthesizer_amd import dist
    port = dist.free_port()
    envs = [dist.rank_env(r, 8, port, base={"PATH": "/bin", "SYNTHHIP_DEVICE": "3"}) for r in range(8)]
    assert [dist.device_for_rank(e) for e in envs] == list(range(8))
    assert all(e["WORLD_SIZE"] == "8" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == str(port) for e in envs)
    assert all(e["RANK"] == e["LOCAL_RANK"] == str(r) for r, e in enumerate(envs))
    assert all("SYNTHHIP_DEVICE" not in e and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    # a launcher that shows every rank one GPU of its own (HIP_VISIBLE_DEVICES per process): each rank's device is ordinal 0
    assert [dist.device_for_rank(e, visible=1) for e in envs] == [0] * 8
    assert [dist.device_for_rank(e, visible=8) for e in envs] == list(range(8))
    assert dist.device_for_rank(envs[5], visible=4) == 5            # fewer GPUs than ranks, more than one: sh_init refuses ordinal 5 loudly


def _check_two_rank_line(d, launcher_prefix):
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 2
    assert d["rccl"]["world"] == 2 and d["rccl"]["rank"] == 0
    ranks = d["rccl"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["device"] for r in ranks] == [0, 1]
    assert len({r["pci"] for r in ranks}) == 2 and all(r["rccl_world"] == 2 for r in ranks)
    assert d["config"]["voices_total"] == 2048 and d["config"]["voices_this_rank"] == 1024 and d["scaling"] == "weak"
    assert d["launcher"].startswith(launcher_prefix), d["launcher"]
    assert d["control_channel"]["barrier"].startswith("shared memory") or d["control_channel"]["barrier"] == "TCP"
    assert d["data"].startswith("dry-run") and d["passes"]["count"] >= 3 and d["value"] > 0


def test_two_rank_dry_run_started_plainly():
    p = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "2", "--min-seconds", "0.05"], {"SYNTHHIP_BENCH_DRY_RUN": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    _check_two_rank_line(_line(p), "bench.py itself")


def test_two_rank_dry_run_under_the_torch_launcher():
    pytest.importorskip("torch")
    from synthesizer_amd import dist
    p = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(dist.free_port()), "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "2", "--min-seconds", "0.05"],
             {"SYNTHHIP_BENCH_DRY_RUN": "1"}, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    _check_two_rank_line(_line(p), "an external launcher")


def test_a_failing_rank_takes_the_job_down():
    """A rank that cannot pass the device check (two ranks told to use the same ordinal) ends the job with its status, promptly."""
    p = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1", "--min-seconds", "0.01"],
             {"SYNTHHIP_BENCH_DRY_RUN": "1", "SYNTHHIP_BENCH_DRY_SAME_PCI": "1"}, timeout=120)
    assert p.returncode == 3 and "RCCL / device check failed" in p.stderr and p.stdout.strip() == ""


def _rdzv_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.pop("SYNTHHIP_RDZV_PORT", None)
    from synthesizer_amd.dist import Rendezvous
    r = Rendezvous(rank, world, shm_barrier=(world != 3))          # (world 3: the barrier over the sockets)
    b = r.broadcast(bytes(range(128)) if rank == 0 else None, 128)
    g = r.gather({"rank": rank, "x": rank * rank})
    m = r.allmax(rank * 1.5, -float(rank), 7.0)
    order = []
    for k in range(50):
        r.barrier()
        order.append(k)
    q.put((rank, b == bytes(range(128)), [x["x"] for x in g], m, len(order), r._slots is not None))
    r.close()


@pytest.mark.parametrize("world", [1, 3, 4])
def test_rendezvous_collectives(world):
    from synthesizer_amd.dist import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for r, (rank, bc_ok, xs, m, nb, shm) in enumerate(got):
        assert rank == r and bc_ok and xs == [k * k for k in range(world)] and nb == 50
        assert m == ((world - 1) * 1.5, 0.0, 7.0)
        assert shm == (world == 4)
