"""How far the shape switches move a result: the float64 bus of one block of the headline's bank and of config 3's FM bank under SYNTHHIP_VARIANT,
SYNTHHIP_NO_SPLIT and SYNTHHIP_GROUPS against the default (a child process per setting: the switches are read once by sh_init).
usage (GPU box): python tools/variant_diff.py"""
import os, subprocess, sys, numpy as np
CHILD = r'''
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
res = {}
for name, env in (("default", {}), ("v484", {"SYNTHHIP_VARIANT": "484"}), ("v444", {"SYNTHHIP_VARIANT": "444"}), ("nosplit", {"SYNTHHIP_NO_SPLIT": "1"}), ("g8", {"SYNTHHIP_GROUPS": "8"})):
    path = "/tmp/vd_%s.npz" % name
    subprocess.run([sys.executable, "-c", CHILD, path], env=dict(os.environ, **env), check=True)
    res[name] = np.load(path)
for name in res:
    for k in ("add", "fm"):
        d = np.max(np.abs(res[name][k] - res["default"][k]))
        print(name, k, "max |diff| %.3e" % d, "scale %.3f" % np.max(np.abs(res["default"][k])))
