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
    return _check_compact(lines[0])


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
    assert d["config"]["voice_shards"] == [[0, 1024], [1024, 2048]]


def _check_compact(text):
    """the one stdout line is <= 4 KB and json.loads()-able with the contract's keys."""
    assert len(text.encode()) <= 4096, len(text.encode())
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    for k in ("workload", "voices_total", "frames_per_step", "samplerate"):
        assert k in d["config"], k
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    return d


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_eight_rank_dry_run_prints_the_compact_line(scaling, tmp_path):
    """The first real N = 8 run's plumbing: 8 ranks, voice shards tiling the table, one compact line from rank 0."""
    detail = tmp_path / "detail.json"
    p = _run([sys.executable, "bench.py", "--gpus", "8", "--scaling", scaling, "--steps", "10", "--warmup", "2", "--min-seconds", "0.05", "--detail", str(detail)],
             {"SYNTHHIP_BENCH_DRY_RUN": "1"}, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = _check_compact(lines[0])
    total = 8192
    assert d["n_gpus"] == 8 and d["scaling"] == scaling and d["config"]["voices_total"] == total and d["config"]["voices_this_rank"] == 1024
    shards = d["config"]["voice_shards"]
    assert len(shards) == 8 and shards[0][0] == 0 and shards[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(shards, shards[1:])) and all(hi - lo == 1024 for lo, hi in shards)
    ranks = d["rccl"]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8)) and [r["device"] for r in ranks] == list(range(8))
    assert len({r["pci"] for r in ranks}) == 8 and all(r["rccl_world"] == 8 for r in ranks) and d["rccl"]["world"] == 8
    full = json.loads(detail.read_text())
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and "passes" in full and full["config"]["voice_shards"] == shards


def test_compact_line_of_a_full_single_gpu_record_fits():
    """A record with every side row a real N = 1 run produces (synthetic numbers, the r05 run's shapes and its long notes) still gives a
    line of <= 4 KB that carries roofline.frac, cpu_baseline.value and the HBM-regime pair."""
    sys.path.insert(0, str(ROOT))
    import bench
    note = "x" * 700
    row = {"ms": 0.44, "bytes": 2689920000, "GBps": 6096.39, "frac_hbm": 0.762, "note": note}
    out = {
        "metric": "Msamples/sec mixed to stereo bus, 1024-voice additive @48kHz", "value": 1327000.123456789, "unit": "Msamples/s", "n_gpus": 1, "steps": 20,
        "warmup": 5, "ms_per_step": 0.037, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "1024-voice additive (Harmonics x16 partials + ADSR) -> float32 stereo bus, 48 kHz, fused generate-and-mix, block 48000 frames",
                   "voices_total": 1024, "voices_this_rank": 1024, "frames_per_step": 48000, "samplerate": 48000, "voice_shards": [[0, 1024]], "adsr": note, "parallelism": "single GPU"},
        "passes": {"count": 7000, "steps_per_pass": 20, "timed_region_s": 5.5, "median_ms_per_step": 0.037, "min_ms_per_step": 0.0365, "max_ms_per_step": 0.05, "note": note},
        "roofline": {"kernel": "k_render_lean<4, 16, 3, 0, false>", "bound": "valu_f64", "kernel_note": note, "clock_note": note, "achieved": 28.5, "peak": 39.3216,
                     "unit": "T f64 lane-ops/s", "frac": 0.7251234, "ops_per_voice_sample": 21.25, "traffic": 29024800.0, "avg_launch_ms": 0.0366, "launches_in_flight": 2,
                     "why": note, "hbm": {"bound": "hbm", "achieved": 10.4, "peak": 8000.0, "unit": "GB/s", "frac": 0.0013, "algorithmic_bytes": 384000.0,
                                          "traffic_over_algorithmic": 75.6, "note": note}, "profile_stale": False, "profile_note": note, "timing_note": note},
        "two_step": {"value": 700000.0, "roofline_mix": {"frac": 0.90, "achieved": 7200.0, "note": note}, "roofline_generate": {"frac": 0.55, "achieved": 4400.0}},
        "two_step_i16": {"note": note, "rows": [row] * 8},
        "configs": {"config1_sine_440Hz_1s_44k1_mono_to_host": {"ms": 0.043, "note": note}, "config2_additive_64v_adsr_48k_stereo": {"ms_per_1s_block": 0.0033, "note": note},
                    "config3_fm_1024v_48k_stereo": {"ms_per_1s_block": 0.037, "note": note}, "config4_8192v_8gpu": {"note": note}},
        "pcm_rows": dict({"resample_f32_8ch_600s_96k_to_44k1": row, "resample_i16_mono_44k1_to_48k_900MB": row, "mix_chain_i16_1024v_10s_stereo": row},
                         **{"row%d" % k: row for k in range(30)}),
        "int16_stream": {"ms_per_step": 0.037, "note": note}, "run_of_blocks": {"ms_per_step": 0.0366, "note": note}, "staggered_notes": {"ms_per_step": 0.043, "note": note},
        "job_from_frame_0": {"ms": 0.5, "note": note},
        "cpu_baseline": {"value": 1.0208, "unit": "Msamples/s", "cores": 1, "kind": "port", "sample": note, "sustain_window": {"note": note},
                         "c_port": {"value": 8.02, "cores": 1}, "c_port_all_cores": {"value": 118.8, "cores": 64, "note": note}, "all_cores": {"note": note}},
        "speedup_vs_cpu_baseline": 1300089.5, "verified": {"ok": True, "block_start_frame": 1, "abs_max": 0.5, "checksum_f64": 1.25, "note": note},
        "rccl": {"world": 1, "rank": 0, "version": "2.27.7", "communicator": False, "ranks": [{"rank": 0, "local_rank": 0, "device": 0, "pci": "0000:05:00.0", "rccl_rank": 0, "rccl_world": 1}], "note": note},
        "library": "synthhip 0.6 (gfx950) src:0123456789abcdef", "launcher": "none (one process)", "control_channel": {"data": None, "barrier": None},
    }
    assert len(json.dumps(out)) > 20000
    d = _check_compact(bench.compact_line(out, "bench_detail.json"))
    assert d["roofline"]["frac"] == pytest.approx(0.725123, rel=1e-5) and d["cpu_baseline"]["value"] == pytest.approx(1.0208)
    assert d["cpu_baseline"]["c_port_all_cores"] == {"value": 118.8, "cores": 64} and d["speedup_vs_cpu_baseline"] > 1e6
    assert d["roofline_hbm_regime"]["mix_frac"] == 0.9 and d["roofline_hbm_regime"]["generate_frac"] == 0.55
    assert d["verified"]["ok"] is True and d["rccl"]["world"] == 1 and d["roofline"]["hbm"]["frac"] == 0.0013
    assert d["side"]["config5_resample_f32_frac_hbm"] == 0.762 and "no reference oracle" in d["side"]["config5_oracle"]
    assert "note" not in json.dumps(d["roofline"]) and d["detail"] == "bench_detail.json"


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


def _token_rank0(port, token, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"], os.environ["SYNTHHIP_RDZV_TOKEN"] = "127.0.0.1", str(port), token
    os.environ.pop("SYNTHHIP_RDZV_PORT", None)
    from synthesizer_amd.dist import Rendezvous
    r = Rendezvous(0, 2, shm_barrier=False, timeout=60)
    q.put(r.gather("zero"))
    r.close()


def test_rendezvous_never_unpickles_and_drops_strangers():
    """the control channel is JSON behind a capped length; a connector without the job's token (or with a pickle, or with
    a huge length prefix) is dropped and the job's own rank still gets through."""
    import pickle
    import socket
    import time
    from synthesizer_amd import dist
    src = (ROOT / "synthesizer_amd" / "dist.py").read_text()
    assert "pickle.loads" not in src and "import pickle" not in src
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = dist.free_port()
    p0 = ctx.Process(target=_token_rank0, args=(port, "s3cret", q))
    p0.start()

    def connect():
        for _ in range(200):
            try:
                return socket.create_connection(("127.0.0.1", port + 1), timeout=5)
            except OSError:
                time.sleep(0.05)
        raise AssertionError("rank 0 never listened")

    class Boom:
        def __reduce__(self):
            return (os.system, ("touch /tmp/synthhip_pwned",))
    evil = pickle.dumps(Boom())
    for payload in (len(evil).to_bytes(8, "little") + evil,                       # the old framing with a pickle inside
                    (1 << 31).to_bytes(4, "little") + b"x" * 16,                  # a length prefix far over the cap
                    (lambda b: len(b).to_bytes(4, "little") + b)(json.dumps({"rank": 1, "token": "wrong"}).encode())):
        c = connect()
        c.sendall(payload)
        c.settimeout(10)
        try:
            assert c.recv(1) == b""           # dropped
        except (ConnectionError, socket.timeout):
            pass
        c.close()
    assert not os.path.exists("/tmp/synthhip_pwned")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"], os.environ["SYNTHHIP_RDZV_TOKEN"] = "127.0.0.1", str(port), "s3cret"
    try:
        r1 = dist.Rendezvous(1, 2, shm_barrier=False, timeout=60)
        assert r1.gather(b"\x00\xff one") == ["zero", b"\x00\xff one"]
        r1.close()
        assert q.get(timeout=60) == ["zero", b"\x00\xff one"]
    finally:
        for k in ("MASTER_ADDR", "MASTER_PORT", "SYNTHHIP_RDZV_TOKEN"):
            os.environ.pop(k, None)
        p0.join(30)
    assert p0.exitcode == 0
    big = {"x": "y" * (dist.Rendezvous.MAX_MESSAGE + 1)}
    with pytest.raises(ValueError):
        dist.Rendezvous._send_obj(None, big)
