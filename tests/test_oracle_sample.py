"""CPU: oracle/sample_oracle.py (the restatement of upstream's Sample editing methods) on hand-checked cases."""
import array

import numpy as np
import pytest

from oracle.sample_oracle import RefSample


def _mk(vals, width=2, rate=10, nch=1):
    return RefSample(array.array({1: "b", 2: "h", 4: "i"}[width], vals).tobytes(), width, rate, nch)


def _vals(r):
    return r.get_frame_array().tolist()


def test_clip_split_join_delay_known_answers():
    r = _mk(list(range(10)))                       # 1 s at 10 Hz
    assert _vals(r.copy().clip(0.2, 0.5)) == [2, 3, 4]
    rest = r.split(0.7)
    assert _vals(r) == [0, 1, 2, 3, 4, 5, 6] and _vals(rest) == [7, 8, 9]
    assert _vals(r.join(rest)) == list(range(10))
    assert _vals(r.copy().delay(0.2)) == [0, 0] + list(range(10))
    assert _vals(r.copy().delay(0.2, keep_length=True)) == [0, 0] + list(range(8))
    assert _vals(r.copy().delay(-0.3)) == list(range(3, 10))
    assert _vals(r.copy().delay(-0.3, keep_length=True)) == list(range(3, 10)) + [0, 0, 0]
    assert _vals(r.copy().add_silence(0.2, at_start=True))[:3] == [0, 0, 0]
    assert len(r.split(5.0)) == 0 and len(r) == 10


def test_echo_known_answer():
    r = _mk([1000] * 4 + [0] * 6)
    # the last 1.0 s (everything), 2 echos 0.2 s apart: the first at 0.5, the second is the first one at 0.25
    # (each echo is the previous echo amplified by the running factor)
    r.echo(1.0, 2, 0.2, 0.5)
    assert _vals(r) == [1000, 1000, 1500, 1500, 625, 625, 125, 125, 0, 0, 0, 0, 0, 0]
    # inaudible echos are skipped
    q = _mk([100] * 10, width=1)
    assert _vals(q.echo(1.0, 5, 0.1, 0.001)) == [100] * 10


def test_envelope_keeps_length_and_shapes():
    r = _mk([10000] * 100, rate=100)
    r.envelope(0.1, 0.2, 0.5, 0.3)
    v = _vals(r)
    assert len(v) == 100
    assert v[0] == 0 and v[5] == 5000 and v[10] == 10000          # attack ramp, decay starts at full level
    assert v[20] == 7500 and v[30] == 5000 and v[69] == 5000       # decay to the sustain level, sustain
    assert v[70] == 5000 and v[85] == 2500 and v[99] == int(5000 * (1.0 - 29 * 1.0 / 30))


def test_modulate_amp_and_speed():
    r = _mk([1000, -1000, 2000, -2000, 3000])
    assert _vals(r.copy().modulate_amp([1, -2])) == [500, 1000, 1000, 2000, 1500]      # scaled by 2, cycled
    assert _vals(r.copy().modulate_amp(iter([0.5, 0.5, 0.5, 0.5, 0.999]))) == [500, -500, 1000, -1000, 2997]
    m = _mk([-32768, 16384], rate=10)
    assert _vals(r.copy().modulate_amp(m)) == [-1000, -500, -2000, -1000, -3000]
    with pytest.raises(OverflowError):
        _mk([30000]).modulate_amp(iter([2.0]))
    s = _mk(list(range(0, 1000, 10)), rate=100)
    assert len(s.copy().speed(2.0)) == 50 and len(s.copy().speed(0.5)) in (199, 200)
    assert s.copy().speed(1.0).frames == s.frames
    with pytest.raises(ValueError):
        s.speed(20)
