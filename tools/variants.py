"""A/B the k_bank_render template variants on the GPU box (one process per variant, same workload)."""
import json
import os
import subprocess
import sys

variants = sys.argv[1:] or ["811", "821", "841", "411", "421", "441", "1611", "1621", "814", "424"]
for v in variants:
    var, _, groups = v.partition(":")
    env = dict(os.environ, SYNTHHIP_VARIANT=var)
    if groups:
        env["SYNTHHIP_GROUPS"] = groups
    p = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "3", "--no-two-step", "--cpu-frames", "0"],
                       env=env, capture_output=True, text=True)
    try:
        j = json.loads(p.stdout.strip().splitlines()[-1])
        print("variant %s: %.0f Msamples/s, %.4f ms/launch" % (v, j["value"], j["roofline"]["avg_launch_ms"]), flush=True)
    except Exception as e:
        print("variant %s failed: %s %s" % (v, p.stdout[-300:], p.stderr[-300:]), flush=True)
