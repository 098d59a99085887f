import sys, os
sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices
from oracle import c_oracle as CO
from oracle import synth_oracle as O
N.ensure_init(0)
SR = 48000
n = 48000
gv, gains = additive_voices(G, 1024, SR, seed=3)
ov, _ = additive_voices(O, 1024, SR, seed=3)
for lo, hi in ((0, 1024), (0, 400), (400, 1024), (0, 384), (0, 64), (0, 128)):
    bank = VoiceBank(gv[lo:hi], gains=gains[lo:hi])
    got = bank.render(n)
    m = 2048
    want = CO.mix_bus(np.stack([CO.render(v, m) for v in ov[lo:hi]]), gains[lo:hi])
    err = np.abs(got[:m].astype(np.float64) - want)
    bad = np.nonzero(err.max(axis=1) > 1e-5)[0]
    print(os.environ.get("LABEL", ""), lo, hi, "max err first %d frames: %.3e; bad frames: %d first %s" % (m, err.max(), len(bad), bad[:8]), flush=True)
    got2 = bank.render(n)
    print("   rerender equal:", np.array_equal(got, got2), " late frames |got| max", np.abs(got[m:]).max())
