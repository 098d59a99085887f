"""bench.py on the GPU, briefly: the ONE stdout line the driver parses from a real run, not only from the dry runs of
tests/test_bench_launch.py -- <= 4 KB, the contract's keys, a live roofline (HIP-event launch time x committed lane-ops per voice-sample),
`verified.ok`, and the detail rows in the file beside it."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_a_short_real_run_prints_the_compact_line(gpu, tmp_path):
    detail = tmp_path / "detail.json"
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--min-seconds", "0.2", "--no-configs",
                        "--no-pcm-rows", "--cpu-frames", "2048", "--no-cpu-all-cores", "--detail", str(detail)],
                       cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, p.stdout[-1000:]
    assert len(lines[0].encode()) <= 4096
    d = json.loads(lines[0])
    assert d["metric"].startswith("Msamples/sec mixed to stereo bus") and d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["value"] > 1e5 and 0.01 < d["ms_per_step"] < 1.0 and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["voices_total"] == 1024 and d["config"]["frames_per_step"] == 48000 and d["config"]["voice_shards"] == [[0, 1024]]
    r = d["roofline"]
    assert r["bound"] == "valu_f64" and 0.3 < r["frac"] < 1.0 and r["peak"] == pytest.approx(39.3216) and r["traffic"] and r["algorithmic_bytes"] == 384000.0
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-4 and 0.0 < r["hbm"]["frac"] < 0.01
    h = d["roofline_hbm_regime"]
    assert 0.7 < h["mix_frac"] < 1.0 and 0.3 < h["generate_frac"] < 1.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0 and d["speedup_vs_cpu_baseline"] > 1e4
    assert d["verified"]["ok"] is True and d["rccl"]["world"] == 1
    full = json.loads(detail.read_text())
    assert full["two_step_i16"]["int16_guard"]["rows_x"] > 0.9 and "passes" in full and full["value"] == pytest.approx(d["value"], rel=1e-5)
