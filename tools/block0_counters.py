"""Block 0 of the bench bank twenty times (every voice through the general code), for rocprofv3 --pmc runs."""
import sys
sys.path.insert(0, ".")
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.mixer import VoiceBank
from synthesizer_amd.workloads import additive_voices
N.ensure_init(0)
SR = 48000
env = len(sys.argv) > 1 and sys.argv[1] == "env"
bus = N.DeviceBuffer(SR * 8)
voices, gains = additive_voices(G, 1024, SR, seed=0, envelope=env, adsr={"sustain": 100.0})
bank = VoiceBank(voices, gains=gains)
for rep in range(20):
    bank.render_device(SR, 0, bus_f32=bus)
    N.sync()
