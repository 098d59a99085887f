"""How many voices of the headline workload take the lean loop, block by block, and what a block costs."""
import ctypes as C
import sys

sys.path.insert(0, ".")
import bench
from synthesizer_amd import _native as N
from synthesizer_amd.mixer import VoiceBank

N.ensure_init(0)
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
voices, gains = bench.build_voices(1024, nblocks + 1.0)
bank = VoiceBank(list(voices), gains=list(gains))
bus = N.DeviceBuffer(48000 * 8)
L = N.lib()
for s in range(nblocks):
    N.timer_start()
    bank.render_device(48000, s * 48000, bus_f32=bus)
    ms = N.timer_stop()
    if s < 12 or s % 25 == 0:
        a, b = C.c_uint32(), C.c_uint32()
        N.check(L.sh_bank_launch_stats(bank._bank.handle, C.byref(a), C.byref(b)))
        print("block %4d: fast %4d general %4d  %.1f us" % (s, a.value, b.value, ms * 1e3))
