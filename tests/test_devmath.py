"""devmath.hpp (the float64 trigonometry and waveform formulas of the kernels) compiled for the host with
g++ and compared with libm / the Python expressions of the reference.  No GPU involved: this checks the
arithmetic the kernels will do, not the kernels."""
import ctypes
import math
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def dm(tmp_path_factory):
    out = tmp_path_factory.mktemp("devmath") / "libdevmath.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-shared", "-fPIC",
                    str(ROOT / "tests" / "cpu_devmath.cpp"), "-o", str(out)], check=True)
    return ctypes.CDLL(str(out))


def _call(fn, t, nout, *extra):
    t = np.ascontiguousarray(t, dtype=np.float64)
    outs = [np.empty_like(t) for _ in range(nout)]
    P = ctypes.POINTER(ctypes.c_double)
    fn.restype = None
    fn(t.ctypes.data_as(P), ctypes.c_int(len(t)), *[ctypes.c_double(e) for e in extra], *[o.ctypes.data_as(P) for o in outs])
    return outs


def test_table_sincos_accuracy(dm):
    rng = np.random.default_rng(0)
    t = np.concatenate([rng.uniform(-10, 10, 200000), rng.uniform(-1e6, 1e6, 200000), rng.uniform(-1e9, 1e9, 100000),
                        np.arange(-2000, 2000) * (math.pi / 512), [0.0, -0.0, 1e-300, 5e-324]])
    s, c = _call(dm.dm_sincos, t, 2)
    ls = np.sin(t.astype(np.longdouble)).astype(np.float64)
    lc = np.cos(t.astype(np.longdouble)).astype(np.float64)
    assert np.max(np.abs(s - ls)) < 4e-16 and np.max(np.abs(c - lc)) < 4e-16
    # the float32 the kernel stores is the correctly rounded one except in a vanishing fraction of cases
    assert np.mean(s.astype(np.float32) != ls.astype(np.float32)) < 1e-5


def test_polynomial_sincos_accuracy(dm):
    rng = np.random.default_rng(1)
    t = np.concatenate([rng.uniform(-10, 10, 100000), rng.uniform(-1e8, 1e8, 100000)])
    s, c = _call(dm.dm_sincos_poly, t, 2)
    (s1,) = _call(dm.dm_sin, t, 1)
    (c1,) = _call(dm.dm_cos, t, 1)
    ls = np.sin(t.astype(np.longdouble)).astype(np.float64)
    lc = np.cos(t.astype(np.longdouble)).astype(np.float64)
    for got, want in ((s, ls), (s1, ls), (c, lc), (c1, lc)):
        assert np.max(np.abs(got - want)) < 6e-16


def test_waveform_formulas_match_the_python_expressions(dm):
    rng = np.random.default_rng(2)
    t = np.concatenate([rng.uniform(-5, 5, 50000), np.arange(-40, 40) * 0.25, np.arange(-40, 40) * 0.25 + 1e-17,
                        np.nextafter(np.arange(-20, 20) * 0.5, 100), np.nextafter(np.arange(-20, 20) * 0.5, -100),
                        [-1e-20, 1e-20, -0.0, 0.0]])
    amp, bias, pw = 0.8, 0.05, 0.3
    (saw,) = _call(dm.dm_saw, t, 1, amp * 2.0, bias)
    (sq,) = _call(dm.dm_square, t, 1, amp, bias)
    (pu,) = _call(dm.dm_pulse, t, 1, pw, amp, bias)
    for i, x in enumerate(t.tolist()):
        assert saw[i] == bias + amp * 2.0 * (x - math.floor(0.5 + x))
        assert sq[i] == (-amp if int(x * 2) % 2 else amp) + bias
        assert pu[i] == (amp if x % 1.0 < pw else -amp) + bias
