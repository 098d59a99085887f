#!/usr/bin/env python
"""16-bit mono Sample.resample rows, timed (GPU box, repo root): ms and fraction of 8 TB/s per rate pair, and every output checked against
live audioop.ratecv on a head / middle / tail window.  SYNTHHIP_NO_PERIOD=1: k_resample_small (rounds 1-5) instead of k_resample_period_i16."""
import audioop
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from synthesizer_amd import _native as N  # noqa: E402


def main():
    N.ensure_init(0)
    L = N.lib()
    frames = 450_000_000
    rng = np.random.default_rng(5)
    host = rng.integers(-32768, 32768, size=1 << 22, dtype=np.int16)
    src = N.DeviceBuffer(frames * 2)
    # (the device buffer: the 8 MB pattern repeated -- uploads of 900 MB would dominate the call)
    reps = frames // host.size + 1
    big = np.tile(host, reps)[:frames]
    src.upload(big)
    pairs = [(44100, 48000), (48000, 44100), (96000, 44100), (44100, 96000), (22050, 48000), (48000, 96000), (8000, 44100), (44100, 32000)]
    if len(sys.argv) > 1:
        pairs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]          # in:out or in:out:channels
    total = frames
    for spec in pairs:
        inr, outr = spec[0], spec[1]
        nch = spec[2] if len(spec) > 2 else 1
        frames = total // nch
        nout = L.sh_resample_out_frames(frames, inr, outr)
        dst = N.DeviceBuffer(nout * 2 * nch)
        call = lambda: N.check(L.sh_resample(src.handle, frames, nch, 2, 0, inr, outr, dst.handle, None))
        for _ in range(30):                       # (clocks up: the first dozen launches of a memory-bound kernel run 10-20 % slow)
            call()
        N.sync()
        runs = []
        for _ in range(12):
            N.timer_start()
            for _ in range(10):
                call()
            runs.append(N.timer_stop() / 10)
        runs.sort()
        best, tot, n = runs[0], runs[len(runs) // 2], 1     # (tot / n: the median run)
        nbytes = (frames + nout) * 2 * nch
        # parity on three windows of the result against the live module (the reference's own arithmetic)
        ok = True
        got_all = None
        for name, first in (("head", 0), ("middle", (nout // 2) & ~15), ("tail", max(0, nout - 70000) & ~15)):
            cnt = min(65536, nout - first)
            got = dst.view(first * 2 * nch, cnt * 2 * nch).download(np.int16, cnt * nch)
            # the input span those outputs read, from an aligned start: ratecv from scratch on the span gives the same values when the
            # span starts at a period boundary of the position sequence: use the library's own range logic instead -- whole-prefix oracle
            # for the head, and for the others the period trick (start at an output index that is a multiple of outr_reduced)
            g = np.gcd(inr, outr)
            ri, ro = inr // g, outr // g
            m0 = (first // ro) * ro
            j0 = m0 // ro * ri
            skip = first - m0
            need_in = ((first + cnt) * ri) // ro - j0 + 2
            seg = big[j0 * nch:(j0 + need_in) * nch].tobytes()
            ref = np.frombuffer(audioop.ratecv(seg, 2, nch, inr, outr, None)[0], dtype=np.int16)
            ref = ref[skip * nch:(skip + cnt) * nch]
            m = min(len(ref), len(got))
            if m < (cnt - 2) * nch or not np.array_equal(ref[:m], got[:m]):
                ok = False
                bad = np.nonzero(ref[:m] != got[:m])[0]
                print("   MISMATCH %s: %d of %d differ, first at %s" % (name, len(bad), m, bad[:5]), flush=True)
        print("%6d -> %6d x%d: %.4f ms median, %.4f best, %.3f of 8 TB/s (best %.3f)  parity %s" %
              (inr, outr, nch, tot / n, best, nbytes / (tot / n * 1e-3) / 8e12, nbytes / (best * 1e-3) / 8e12, "ok" if ok else "FAILED"), flush=True)
        dst.free()


if __name__ == "__main__":
    main()
