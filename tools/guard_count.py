import os, sys, ctypes
sys.path.insert(0, os.getcwd())
os.environ["SYNTHHIP_ALLOW_STALE"] = "1"
os.environ["SYNTHHIP_LIB"] = "synthesizer_amd/build/libsynthhip_count.so"
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd import workloads as W
from synthesizer_amd.mixer import VoiceBank
N.ensure_init(0)
L = ctypes.CDLL(os.path.abspath(os.environ["SYNTHHIP_LIB"]))
SR, NV, F2 = 48000, 1024, 480000
v, g = W.additive_voices(G, NV, SR, seed=0, partials=16, adsr={"sustain": 1.0e6})
bank = VoiceBank(v, gains=g)
rows = N.DeviceBuffer(NV * F2 * 2)
out = (ctypes.c_int * 16)()
L.sh_debug_flag_words(out, 16)
for start in (5 * SR, 300 * SR):
    bank.generate_i16_device(F2, start, out=rows, stride=F2, check=False)
    N.sync()
    L.sh_debug_flag_words(out, 16)
    quarters = NV * F2 // 256
    import struct
    vv = struct.unpack("d", struct.pack("II", out[10] & 0xFFFFFFFF, out[11] & 0xFFFFFFFF))[0]
    ww = struct.unpack("d", struct.pack("II", out[8] & 0xFFFFFFFF, out[9] & 0xFFFFFFFF))[0]
    print("first near sample in the careful path: v = %r, scale v = %r, w = %r (low word %d), near_lo %d, lane %d, frame %d; samples the careful path itself found near: %d" % (vv, 32767.0 * vv, ww, out[8] & 0xFFFFFFFF, out[12], out[13], out[14], out[7]))
    print("start %d s: careful quarters %d of %d wave-quarters (%.3g), near lanes %d -> P(sample near) ~ %.3g" % (start // SR, out[5], quarters, out[5] / quarters, out[6], out[6] / (NV * F2)))
