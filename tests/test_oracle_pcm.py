"""The oracle itself must be right before anything is compared with it: oracle/pcm_oracle.py (restatement
of CPython 3.10 Modules/audioop.c add/ratecv) against the live ``audioop`` module and the golden vectors."""
import audioop
import sys

import numpy as np
import pytest

from oracle import pcm_oracle as P

DT = {1: np.int8, 2: np.int16, 4: np.int32}


def _rand_bytes(rng, width, nsamples):
    if width == 3:
        return rng.integers(0, 256, nsamples * 3, dtype=np.uint8).tobytes()
    info = np.iinfo(DT[width])
    return rng.integers(info.min, info.max + 1, nsamples, dtype=np.int64).astype(DT[width]).tobytes()


def test_python_version_is_the_pinned_one():
    assert sys.version_info[:2] == (3, 10), "audioop golden vectors are pinned to CPython 3.10"


@pytest.mark.parametrize("width", [1, 2, 3, 4])
def test_add_matches_audioop(width):
    rng = np.random.default_rng(width)
    for n in (0, 1, 17, 4096):
        a, b = _rand_bytes(rng, width, n), _rand_bytes(rng, width, n)
        assert P.add(a, b, width) == audioop.add(a, b, width)
    with pytest.raises(ValueError):
        P.add(b"\0\0", b"\0\0\0\0", 2)
    with pytest.raises(audioop.error):
        audioop.add(b"\0\0", b"\0\0\0\0", 2)


def test_add_golden():
    g = np.load("tests/golden/audioop_add.npz")
    for w in (1, 2, 4):
        assert P.add(g["a%d" % w].tobytes(), g["b%d" % w].tobytes(), w) == g["sum%d" % w].tobytes()
    assert P.add_chain([c.tobytes() for c in g["chain_in"]], 2) == g["chain_out"].tobytes()
    # saturation corners really are in the vectors
    assert g["sum2"][0] == 32767 and g["sum2"][1] == -32768


@pytest.mark.parametrize("width", [1, 2, 3, 4])
@pytest.mark.parametrize("nch", [1, 2, 8])
def test_ratecv_matches_audioop(width, nch):
    rng = np.random.default_rng(100 + width * 10 + nch)
    for (i, o) in ((96000, 44100), (48000, 44100), (44100, 48000), (8000, 8000), (44100, 22050), (22050, 44100), (3, 7)):
        for frames in (0, 1, 2, 5, 777):
            raw = _rand_bytes(rng, width, frames * nch)
            ref = audioop.ratecv(raw, width, nch, i, o, None)[0]
            assert P.ratecv(raw, width, nch, i, o) == ref
            if frames <= 5 or (width == 2 and nch == 2):
                assert P.ratecv_sequential(raw, width, nch, i, o) == ref
            assert len(ref) == P.ratecv_out_frames(frames, i, o) * width * nch


def test_ratecv_golden():
    g = np.load("tests/golden/audioop_ratecv.npz")
    n = 0
    while "case%d_meta" % n in g:
        i, o, nch, width = (int(v) for v in g["case%d_meta" % n])
        assert P.ratecv(g["case%d_in" % n].tobytes(), width, nch, i, o) == g["case%d_out" % n].tobytes()
        n += 1
    assert n >= 30
    assert P.ratecv(g["ramp_in"].tobytes(), 2, 1, 96000, 44100) == g["ramp_out"].tobytes()


def test_ratecv_f32_is_linear_interpolation():
    x = np.linspace(-1, 1, 1000, dtype=np.float32).reshape(-1, 1)
    y = P.ratecv_f32(x, 96000, 44100)
    pos = np.arange(len(y)) * 96000 / 44100
    want = np.interp(pos, np.arange(1000), x[:, 0].astype(np.float64))
    assert np.max(np.abs(y[:, 0] - want)) < 1e-6
    assert P.ratecv_f32(np.zeros((0, 2), np.float32), 3, 7).shape == (0, 2)


@pytest.mark.parametrize("width", [1, 2, 4])
def test_elementwise_ops_match_audioop(width):
    rng = np.random.default_rng(50 + width)
    raw = _rand_bytes(rng, width, 2000)
    info = np.iinfo(DT[width])
    corners = np.array([info.max, info.min, info.min + 1, info.max - 1, 0, -1, 1, 3, -3, 5, -5, 100], dtype=DT[width]).tobytes()
    for frames in (raw, corners, b""):
        for factor in (1.5, 0.5, -1.0, 1.00001, 0.0, 2.0, -0.333, 1.0 / 65536):
            assert P.mul(frames, width, factor) == audioop.mul(frames, width, factor)
        for b in (1, -1, 12345, -70000 if width > 1 else -70):
            assert P.bias(frames, width, b) == audioop.bias(frames, width, b)
        assert P.reverse(frames, width) == audioop.reverse(frames, width)
        for lf, rf in ((1.0, 1.0), (0.5, 0.25), (1.0, 0.0), (0.0, 1.0), (-1.0, 0.7)):
            assert P.tomono(frames, width, lf, rf) == audioop.tomono(frames, width, lf, rf)
            assert P.tostereo(frames, width, lf, rf) == audioop.tostereo(frames, width, lf, rf)
        for nw in (1, 2, 4):
            assert P.lin2lin(frames, width, nw) == audioop.lin2lin(frames, width, nw)
        assert P.amax(frames, width) == audioop.max(frames, width)
        assert P.rms(frames, width) == audioop.rms(frames, width)


def test_elementwise_ops_golden():
    """oracle restatements of audioop.mul / bias / reverse / tomono / tostereo / lin2lin / max / rms against the committed
    outputs of the live module (tests/golden/audioop_ops.npz)."""
    g = np.load("tests/golden/audioop_ops.npz")
    for width in (1, 2, 4):
        raw = g["x%d" % width].tobytes()
        assert P.mul(raw, width, 1.5) == g["mul%d_1p5" % width].tobytes()
        assert P.mul(raw, width, -0.333) == g["mul%d_m0p333" % width].tobytes()
        assert P.bias(raw, width, 1000) == g["bias%d_1000" % width].tobytes()
        assert P.reverse(raw, width) == g["reverse%d" % width].tobytes()
        assert P.tomono(raw, width, 0.75, 0.5) == g["tomono%d" % width].tobytes()
        assert P.tostereo(raw, width, 0.3, 1.2) == g["tostereo%d" % width].tobytes()
        assert [P.amax(raw, width), P.rms(raw, width)] == g["max_rms%d" % width].tolist()
        for nw in (1, 2, 4):
            if nw != width:
                assert P.lin2lin(raw, width, nw) == g["lin2lin%d_%d" % (width, nw)].tobytes()
