#!/usr/bin/env python
"""Where does one launch of the headline render kernel spend its time?  (measure first.)

    python tools/ab.py build diag -DSH_DIAG
    SYNTHHIP_LIB=synthesizer_amd/build/libsynthhip_diag.so python tools/headline_phases.py [--serial] [--tag NAME] > out.json

A -DSH_DIAG library leaves, per wavefront of the render kernel, the 100 MHz timestamps of its phases (entry, fold of the launch two
back, trig table in LDS + barrier, voice loop, LDS reduce, store) and the SIMD it ran on.  This script renders the bench's headline
stream (1024 additive voices, 48 000-frame blocks, block 6804 on: the steady state the timed passes sit in), reads the records of the
last launches and prints a JSON summary: phase durations, the spread of starts and ends over the chip, wavefronts per SIMD.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

DIAG_WAVES, DIAG_SLOTS = 4096, 10


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(q * len(xs)))]


def summarize(rec, waves_per_wg=4):
    """rec: [nwaves, DIAG_SLOTS] uint64 of ONE launch (rendering wavefronts only)."""
    t = rec[:, :6].astype(np.int64) * 10          # ns
    t0 = int(t[:, 0].min())
    rel = t - t0
    hw = rec[:, 6]
    xcc = (hw >> 32).astype(np.int64) & 0xF
    hwid = hw.astype(np.int64) & 0xFFFFFFFF
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    simd_key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
    cu_key = simd_key // 4
    per_simd = np.bincount(np.unique(simd_key, return_inverse=True)[1])
    per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1])
    clk = (rec[:, 8].astype(np.int64) - rec[:, 7].astype(np.int64)) / np.maximum(1, (t[:, 5] - t[:, 0]))      # core cycles per ns

    def ph(a, b):
        d = (t[:, b] - t[:, a]) / 1e3
        return {"median_us": float(np.median(d)), "p95_us": float(np.percentile(d, 95)), "max_us": float(d.max()), "mean_us": float(d.mean())}
    # by the number of wavefronts that shared the wave's SIMD
    inv = np.unique(simd_key, return_inverse=True)[1]
    share = per_simd[inv]
    loop = (t[:, 3] - t[:, 2]) / 1e3
    by_share = {int(k): {"waves": int((share == k).sum()), "loop_median_us": float(np.median(loop[share == k])),
                         "end_median_us": float(np.median(rel[share == k, 5]) / 1e3), "end_max_us": float(rel[share == k, 5].max() / 1e3)}
                for k in sorted(set(share.tolist()))}
    wg0 = np.arange(len(rec)) // waves_per_wg
    return {
        "waves": int(len(rec)),
        "launch_span_us": float(rel[:, 5].max() / 1e3),
        "starts_us": {"first": 0.0, "median": float(np.median(rel[:, 0]) / 1e3), "p95": float(np.percentile(rel[:, 0], 95) / 1e3), "last": float(rel[:, 0].max() / 1e3)},
        "ends_us": {"first": float(rel[:, 5].min() / 1e3), "median": float(np.median(rel[:, 5]) / 1e3), "p95": float(np.percentile(rel[:, 5], 95) / 1e3),
                    "last": float(rel[:, 5].max() / 1e3)},
        "phases": {"entry_to_fold_done": ph(0, 1), "table_and_barrier": ph(1, 2), "voice_loop": ph(2, 3), "wait_for_the_workgroup": ph(3, 4), "reduce_and_store": ph(4, 5),
                   "whole_wave": ph(0, 5)},
        "simds_used": int(len(per_simd)), "cus_used": int(len(per_cu)),
        "waves_per_simd_histogram": {int(k): int(v) for k, v in zip(*np.unique(per_simd, return_counts=True))},
        "waves_per_cu_histogram": {int(k): int(v) for k, v in zip(*np.unique(per_cu, return_counts=True))},
        "by_waves_sharing_the_simd": by_share,
        "core_clock_GHz": {"median": float(np.median(clk)), "min": float(clk.min()), "max": float(clk.max())},
        "idle_tail_us": {"median_wave_end_to_launch_end": float((rel[:, 5].max() - np.median(rel[:, 5])) / 1e3),
                         "simd_time_idle_frac": float(1.0 - (t[:, 5] - t[:, 0]).sum() / (per_simd.max() * 1.0) / max(1, len(per_simd)) / max(1, rel[:, 5].max()))},
        "fold_workgroups": {"entry_to_fold_done_median_us_group0": float(np.median(((t[:, 1] - t[:, 0]) / 1e3)[: len(rec) // 8])),
                            "others": float(np.median(((t[:, 1] - t[:, 0]) / 1e3)[len(rec) // 8:]))},
        "end_by_group_median_us": [float(np.median(rel[g * (len(rec) // 8):(g + 1) * (len(rec) // 8), 5]) / 1e3) for g in range(8)] if len(rec) % 8 == 0 else None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=400)
    ap.add_argument("--voices", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=48000)
    ap.add_argument("--first-block", type=int, default=6804)
    ap.add_argument("--dump", default=None, help="write the raw records of the last four launches here (.npy)")
    args = ap.parse_args()
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import workloads as W
    from synthesizer_amd.mixer import VoiceBank
    N.ensure_init(0)
    L = N.lib()
    if not hasattr(L, "sh_debug_diag"):
        sys.exit("this library was not built with -DSH_DIAG (tools/ab.py build diag -DSH_DIAG; SYNTHHIP_LIB=...)")
    L.sh_debug_diag.restype = C.c_int
    L.sh_debug_diag.argtypes = [C.c_void_p, C.c_size_t]
    voices, gains = W.additive_voices(G, args.voices, 48000, seed=0, partials=16, adsr={"sustain": 1.0e6})
    bank = VoiceBank(voices, gains=gains)
    F = args.frames
    ring = [N.DeviceBuffer(F * 8) for _ in range(4)]
    k0 = args.first_block
    N.timer_start()
    for k in range(args.blocks):
        bank.render_device(F, (k0 + k) * F, bus_f32=ring[k & 3])
    ms = N.timer_stop()
    N.sync()
    # a second, timed pass of the same length at steady clocks
    k1 = k0 + args.blocks
    N.timer_start()
    for k in range(args.blocks):
        bank.render_device(F, (k1 + k) * F, bus_f32=ring[k & 3])
    ms2 = N.timer_stop()
    N.sync()
    raw = np.zeros(4 * DIAG_WAVES * DIAG_SLOTS, dtype=np.uint64)
    N.check(L.sh_debug_diag(raw.ctypes.data, raw.nbytes))
    raw = raw.reshape(4, DIAG_WAVES, DIAG_SLOTS)
    if args.dump:
        np.save(args.dump, raw)
    out = {"library": L.sh_version().decode(), "serial": os.environ.get("SYNTHHIP_NO_OVERLAP") == "1", "voices": args.voices, "frames": F,
           "us_per_block_first_pass": ms / args.blocks * 1e3, "us_per_block": ms2 / args.blocks * 1e3, "launches": {}}
    last = k1 + args.blocks - 1
    got = ring[(args.blocks - 1) & 3].download(np.float32, F * 2).astype(np.float64)
    out["last_block_checksum"] = {"sum": float(got.sum()), "abs_sum": float(np.abs(got).sum())}
    for b in range(4):
        blk = last - b
        rec = raw[blk & 3]
        gx, gy = int(rec[0, 9] >> 32), int(rec[0, 9] & 0xFFFFFFFF)
        if gx == 0:
            continue
        tiles = gx
        groups = 8 if args.voices >= 1024 else gy
        # rendering wavefronts: the first tiles * groups workgroups (rows behind them resolve the next block's records)
        nrender = min(DIAG_WAVES, tiles * min(gy, groups) * 4)
        out["launches"]["block_%d" % blk] = dict({"grid": [gx, gy]}, **summarize(rec[:nrender]))
    # how the last two launches lie against each other (two streams)
    a, b_ = raw[last & 3], raw[(last - 1) & 3]
    na = min(DIAG_WAVES, int(a[0, 9] >> 32) * 8 * 4)
    if na:
        sa, ea = int(a[:na, 0].min()), int(a[:na, 5].max())
        sb, eb = int(b_[:na, 0].min()), int(b_[:na, 5].max())
        out["last_two_launches"] = {"start_gap_us": (sa - sb) / 100.0, "prev_end_minus_this_start_us": (eb - sa) / 100.0, "this_span_us": (ea - sa) / 100.0,
                                    "prev_span_us": (eb - sb) / 100.0}
    if hasattr(L, "sh_debug_diag2"):
        L.sh_debug_diag2.restype = C.c_int
        L.sh_debug_diag2.argtypes = [C.c_void_p, C.c_size_t]
        d2 = np.zeros(64 * 8, dtype=np.uint64)
        N.check(L.sh_debug_diag2(d2.ctypes.data, d2.nbytes))
        d2 = d2.reshape(64, 8).astype(np.int64)
        rows = []
        for it in range(12):
            t = d2[it, :5]
            if t[0] == 0:
                continue
            rows.append({"voice": it, "ticks_top_to_angle": int(t[1] - t[0]), "angle_to_sincos": int(t[2] - t[1]), "sincos_to_rotation": int(t[3] - t[2]),
                         "rotation_to_end": int(t[4] - t[3]), "whole": int(t[4] - t[0]),
                         "top_to_next_top": int(d2[it + 1, 0] - t[0]) if it + 1 < 12 and d2[it + 1, 0] else None})
        out["one_wavefront_voice_timeline_ticks"] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
