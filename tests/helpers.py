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
