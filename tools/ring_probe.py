"""One rank's share of BASELINE configs[3] through DistVoiceBank's slot ring on a 1-rank RCCL communicator: microseconds per
block by batch size (HIP events and wall clock), beside the same renders without the exchange.  NCCL_* environment variables
apply (e.g. NCCL_MAX_NCHANNELS)."""
import json
import os
import sys
import time

sys.path.insert(0, ".")
import bench
from synthesizer_amd import _native as N
from synthesizer_amd import dist

N.ensure_init(0)
SR = 48000
voices, gains = bench.build_voices(1024)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "SYNTHHIP_"))}}
with bench._stdout_to_stderr():
    dist.init(0, 1, broadcast=lambda payload, rank, world, n: payload)
local = dist.DistVoiceBank(voices, gains, 0, 1).local
bufs = [N.DeviceBuffer(SR * 16) for _ in range(4)]
p = [5]


def local_step():
    local.render_device(SR, p[0] * SR, bus_f32=None, bus_f64=bufs[p[0] & 3])
    p[0] += 1


out["render_only_us"] = bench.steady(N, local_step, min_seconds=0.1, reps=40) * 1e3
# (A) the same renders into 32 views of four big buffers (the ring's memory layout), nothing else
big = [N.DeviceBuffer(8 * SR * 16) for _ in range(4)]
views = [b_.view(j * SR * 16, SR * 16) for b_ in big for j in range(8)]
q = [5]


def view_step():
    local.render_device(SR, q[0] * SR, bus_f32=None, bus_f64=views[q[0] % 32])
    q[0] += 1


out["render_into_32_views_us"] = bench.steady(N, view_step, min_seconds=0.1, reps=64) * 1e3


# (B) the ring with the device side of the exchange stubbed out (marks, waits and collectives do nothing)
class _NoComm(dist._HipBackend):
    def mark_slot(self, slot):
        pass

    def reduce_lagged(self, *a):
        pass

    def wait_slot_keep(self, slot):
        pass

    def reduce_async(self, *a):
        pass

    def wait_slot(self, slot):
        pass


def ring_us(batch, backend=None):
    ring = dist.DistVoiceBank(voices, gains, 0, 1, batch=batch, backend=backend)
    if backend is not None:
        ring.local = backend.local
    ring.world, ring.batch = 2, batch
    pos = [5]

    def step():
        ring.render_device(SR, pos[0] * SR)
        pos[0] += 1
    for _ in range(2 * batch):
        step()
    us = bench.steady(N, step, min_seconds=0.15, reps=4 * batch) * 1e3       # (timer_stop ends the run: one drain per 4 batches stays in)
    ring.flush()
    N.sync()
    return us


for batch in (8, 32):
    out["nocomm_batch%d_us" % batch] = ring_us(batch, _NoComm(voices, gains))
for batch in (8, 16, 32):
    out["comm_batch%d_us" % batch] = ring_us(batch)
dist.shutdown()
print(json.dumps(out))
