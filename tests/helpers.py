"""Shared helpers for the parity tests (oracle side)."""
import math

import numpy as np


def rms(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(np.sqrt(np.mean((a - b) ** 2)))


def accumulated(t0, inc, start, n, chunk=1 << 22):
    """t_start .. t_{start+n-1} of the float64 running sum t += inc (sequential, exact):
    numpy's cumsum is a plain left-to-right loop."""
    t = float(t0)
    done = 0
    while done < start:
        m = min(chunk, start - done)
        a = np.full(m + 1, inc, dtype=np.float64)
        a[0] = t
        t = float(np.cumsum(a)[-1])
        done += m
    a = np.full(n, inc, dtype=np.float64)
    a[0] = t
    return np.cumsum(a)


def ulp32_diff(a, b):
    """distance in float32 ulps between two float32 arrays"""
    a = np.asarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


# ---- the C oracle over many voices, in parallel (a window late in the notes costs the oracle every sample before it) ----

def _oracle_window_worker(args):
    kind, n_total, seed, adsr, lo, hi, step, start, n = args
    from oracle import c_oracle as CO
    from oracle import synth_oracle as O
    from synthesizer_amd.workloads import additive_voices, fm_voices
    if kind == "additive":
        voices, gains = additive_voices(O, n_total, 48000, seed=seed, partials=16, adsr=adsr)
    else:
        voices, gains = fm_voices(O, n_total, 48000, seed=seed)
    idx = list(range(lo, hi, step))
    rows = np.stack([CO.render(voices[i], start + n)[start:] for i in idx])
    return CO.mix_bus(rows, [gains[i] for i in idx])


def oracle_bus_window(kind, n_total, seed, adsr, indices, start, n, max_procs=64):
    """float64 stereo bus [n, 2] of the voices `indices` (a range) of the workload (`kind`, n_total, seed, adsr) over the frames
    [start, start + n), by the C oracle run from frame 0, the voices dealt to up to max_procs processes (the partial buses of
    the processes are added in voice order: float64 summation order differs from one long sum by ~1e-16)."""
    import multiprocessing as mp
    import os
    idx = list(indices)
    lo, hi, step = idx[0], idx[-1] + 1, (idx[1] - idx[0]) if len(idx) > 1 else 1
    assert idx == list(range(lo, hi, step))
    nproc = max(1, min(os.cpu_count() or 1, max_procs, len(idx)))
    per = -(-len(idx) // nproc)
    jobs = []
    for p in range(nproc):
        part = idx[p * per:(p + 1) * per]
        if part:
            jobs.append((kind, n_total, seed, adsr, part[0], part[-1] + 1, step, start, n))
    if len(jobs) == 1:
        return _oracle_window_worker(jobs[0])
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        parts = pool.map(_oracle_window_worker, jobs, chunksize=1)
    bus = parts[0].copy()
    for p in parts[1:]:
        bus += p
    return bus
