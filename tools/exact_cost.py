#!/usr/bin/env python
"""What params.exact_harmonics costs (GPU box, repo root): the bank routes with the polynomial Harmonics form against the term-by-term form
(the reference's own loop: sin(fl(t k)) a_k in list order), at config 2 size (64 voices) and at the headline's (1024).

    python tools/exact_cost.py
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def steady(N, call, min_seconds=0.15, reps=5):
    call()
    N.sync()
    loops, total = [], 0.0
    while total < min_seconds or len(loops) < 3:
        N.timer_start()
        for _ in range(reps):
            call()
        ms = N.timer_stop()
        loops.append(ms / reps)
        total += ms / 1e3
    return statistics.median(loops)


def main():
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import params
    from synthesizer_amd import workloads as W
    from synthesizer_amd.mixer import VoiceBank
    N.ensure_init(0)
    SR = 48000
    print("%-8s %-10s %-34s %12s %12s %8s" % ("voices", "frames", "route", "poly ms", "exact ms", "x"))
    for nv, n in ((64, 48000), (64, 480000), (256, 48000), (1024, 48000), (1024, 480000)):
        banks = {}
        for exact in (False, True):
            params.exact_harmonics = exact
            try:
                v, g = W.additive_voices(G, nv, SR, seed=0, partials=16, adsr={"sustain": 1.0e6})
                banks[exact] = VoiceBank(v, gains=g)
            finally:
                params.exact_harmonics = False
        stride = (n + 63) & ~63
        rows = N.DeviceBuffer(nv * stride * 2)
        mono = N.DeviceBuffer(n * 2)
        ring = [N.DeviceBuffer(n * 8) for _ in range(4)]
        pos = [0]
        start = 5 * SR

        def routes(b):
            def render():
                b.render_device(n, start + (pos[0] % 16) * n, bus_f32=ring[pos[0] & 3])
                pos[0] += 1
            return (("render -> float32 bus (fused)", render),
                    ("generate_i16 rows", lambda: b.generate_i16_device(n, start, out=rows, stride=stride, check=False)),
                    ("mixdown_i16 (saturating chain)", lambda: b.mixdown_i16_device(n, start, out=mono, check=False)))
        for (name, fp), (_n, fe) in zip(routes(banks[False]), routes(banks[True])):
            a = steady(N, fp)
            e = steady(N, fe)
            print("%-8d %-10d %-34s %12.4f %12.4f %8.2f" % (nv, n, name, a, e, e / a), flush=True)
        for b in banks.values():
            b.overflow_check()
        for x in [rows, mono] + ring:
            x.free()


if __name__ == "__main__":
    main()
