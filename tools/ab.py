#!/usr/bin/env python
"""A/B plumbing for kernel work (GPU box, from the repo root):

    python tools/ab.py build NAME [extra hipcc flags ...]   a VARIANT of libsynthhip.so beside the shipped one: synthesizer_amd/build/libsynthhip_NAME.so
                                                           (load it with SYNTHHIP_LIB=<that path>; e.g. -DSH_DIAG for tools/headline_phases.py)
    python tools/ab.py time LIB [LIB ...]                   the headline (us per block: median pass, best pass) under each library, two streams and one
    python tools/ab.py config CONFIG LIB [LIB ...]          bench.py --only-config CONFIG (config2 / config3 / staggered) under each library
    python tools/ab.py shapes [WFM[:groups] ...]            the headline under SYNTHHIP_VARIANT (and SYNTHHIP_GROUPS) settings of ONE library
    python tools/ab.py diff                                 how far the shape switches move a result: float64 bus of one block, headline and config 3

(Round 5: build_variant.py, variant_diff.py, variants.py and the ab_*.sh wrappers in one file.)"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def build(argv):
    from synthesizer_amd import build as B
    name, extra = argv[0], argv[1:]
    out = B.HERE / "build" / ("libsynthhip_%s.so" % name)
    objdir = B.HERE / "build" / ("obj_" + name)
    objdir.mkdir(parents=True, exist_ok=True)
    defs = ['-DSH_SOURCE_HASH="%s"' % B.source_hash()] + extra
    jobs = []
    for src in B.SOURCES:
        obj = objdir / (src + ".o")
        jobs.append((src, obj, subprocess.Popen([B.HIPCC] + B.FLAGS + defs + ["-c", str(B.CSRC / src), "-o", str(obj)])))
    bad = [s for s, _o, p in jobs if p.wait() != 0]
    if bad:
        sys.exit("failed: " + " ".join(bad))
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [str(o) for _s, o, _p in jobs] + ["-o", str(out), "-ldl"], check=True)
    import shutil
    shutil.rmtree(objdir, ignore_errors=True)
    print(out)


def _bench(args, env):
    p = subprocess.run([sys.executable, "bench.py"] + args, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    try:
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:
        print("bench.py failed: %s %s" % (p.stdout[-300:], p.stderr[-300:]), flush=True)
        return None


def time_libs(argv):
    for lib in argv:
        for no_overlap in ("0", "1"):
            d = _bench(["--no-pcm-rows", "--no-two-step", "--no-configs", "--cpu-frames", "0", "--min-seconds", "0.6"],
                       {"SYNTHHIP_ALLOW_STALE": "1", "SYNTHHIP_LIB": lib, "SYNTHHIP_NO_OVERLAP": no_overlap})
            if d:
                print("%s no_overlap %s: %.2f %.2f us per block (median pass, best pass)" % (lib, no_overlap, d["ms_per_step"] * 1e3, d["passes"]["min_ms_per_step"] * 1e3), flush=True)


def config(argv):
    cfg, libs = argv[0], argv[1:]
    for lib in libs:
        d = _bench(["--only-config", cfg], {"SYNTHHIP_ALLOW_STALE": "1", "SYNTHHIP_LIB": lib})
        if d:
            for name, row in d["configs"].items():
                print("%s %s: %s" % (lib, name, {k: v for k, v in row.items() if isinstance(v, (int, float))}), flush=True)


def shapes(argv):
    for v in argv or ["4163", "484", "444", "844", "484:8", "484:16"]:
        var, _, groups = v.partition(":")
        env = {"SYNTHHIP_VARIANT": var}
        if groups:
            env["SYNTHHIP_GROUPS"] = groups
        d = _bench(["--steps", "40", "--warmup", "3", "--no-two-step", "--no-configs", "--no-pcm-rows", "--cpu-frames", "0", "--min-seconds", "0.5"], env)
        if d:
            print("variant %s: %.0f Msamples/s, %.4f ms/launch" % (v, d["value"], d["roofline"]["avg_launch_ms"]), flush=True)


_DIFF_CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G, workloads as W
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
out = {}
for name, (v, g) in (("add", W.additive_voices(G, 1024, 48000, seed=0, partials=16, adsr={"sustain": 1e9})), ("fm", W.fm_voices(G, 1024, 48000, seed=1))):
    bank = VoiceBank(v, gains=g)
    b = N.DeviceBuffer(48000 * 16)
    bank.render_device(48000, 100 * 48000, bus_f64=b)
    out[name] = b.download(np.float64, 96000)
np.savez(sys.argv[1], **out)
'''


def diff(argv):
    import numpy as np
    res = {}
    for name, env in (("default", {}), ("v484", {"SYNTHHIP_VARIANT": "484"}), ("v444", {"SYNTHHIP_VARIANT": "444"}), ("nosplit", {"SYNTHHIP_NO_SPLIT": "1"}),
                      ("g8", {"SYNTHHIP_GROUPS": "8"})):
        path = "/tmp/vd_%s.npz" % name
        subprocess.run([sys.executable, "-c", _DIFF_CHILD, path], env=dict(os.environ, **env), check=True)
        res[name] = np.load(path)
    for name in res:
        for k in ("add", "fm"):
            print(name, k, "max |diff| %.3e" % np.max(np.abs(res[name][k] - res["default"][k])), "scale %.3f" % np.max(np.abs(res["default"][k])))


COMMANDS = {"build": build, "time": time_libs, "config": config, "shapes": shapes, "diff": diff}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        sys.exit(__doc__)
    COMMANDS[sys.argv[1]](sys.argv[2:])
